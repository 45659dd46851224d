#!/bin/bash
# Round 5, visit g: phase stamps of the shipped kernel (probe build), A/B of deferred hidden-row stores.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for cfg in "4096 8" "2048 4"; do set -- $cfg
  echo "== probe_bps, envs $1, $2 waves per workgroup"
  SMZ_LIB_PATH=$R/gpurun_variants/libsmz_probe_bps.so python3 tools/bps_probe.py $1 $2 2>&1 | grep -v amdgpu.ids
done | tee $O/r05_g_bps_probe.txt
run() { python bench.py $2 --min-timed-seconds 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 | $2 |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4))"; }
SMZ_LIB_PATH=$R/gpurun_variants/libsmz_defer.so timeout 900 python -m pytest "tests/test_gpu_fullsize_parity.py::test_production_search_kernel_equals_oracle_on_every_tree" tests/test_gpu_episode_parity.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2 3; do for w in "" "--rng philox"; do
  unset SMZ_LIB_PATH;                                       run "rows stored inside the pass (shipped)" "$w"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_defer.so;  run "rows stored after the staging        " "$w"
done; done 2>&1 | tee $O/r05_g_defer_rows_ab.txt
