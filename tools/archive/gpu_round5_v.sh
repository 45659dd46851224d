#!/bin/bash
# Round 5, visit v: (1) phase stamps of the kernel after the staging changes (probe sets A and B); (2) the paired descent for four
# actions (SMZ_PAIR_A4): parity with the variant library, A/B on the four-action workloads.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in probe_a probe_b; do for cfg in "4096 8" "2048 4"; do set -- $cfg
  echo "== $v, envs $1, $2 waves per workgroup"
  SMZ_LIB_PATH=$R/gpurun_variants/libsmz_$v.so python3 tools/bps_probe.py $1 $2 2>&1 | grep -v amdgpu.ids
done; done 2>&1 | tee $O/r05_v_bps_probe.txt
export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_pair4.so
timeout 2400 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_end_to_end.py tests/test_gpu_episode_parity.py tests/test_gpu_tree_parity.py -m gpu -q -x 2>&1 | tail -3
unset SMZ_LIB_PATH
run() { python bench.py $2 --min-timed-seconds 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 | $2 |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), r['kernel_launched'], 'chain', round((r.get('bound_actual') or {}).get('frac',0),3))"; }
for rep in 1 2; do for w in "--workload lunarlander_mlp_4096x50" "--workload lunarlander_mlp_4096x50 --rng philox" "--workload lunarlander_mlp_4096x50 --envs 8192" ""; do
  unset SMZ_LIB_PATH;                                      run "sequential descent (four actions)" "$w"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_pair4.so;  run "paired descent                   " "$w"
done; done 2>&1 | tee $O/r05_v_pair4_ab.txt
