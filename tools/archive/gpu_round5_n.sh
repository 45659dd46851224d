#!/bin/bash
# Round 5, visit n: block-parallel selection for FOUR actions (root on a quad, SMZ_BPS_A4): parity, A/B on the LunarLander-shaped workload.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_episode_parity.py tests/test_gpu_end_to_end.py -m gpu -q -x 2>&1 | tail -4
run() { python bench.py $2 --min-timed-seconds 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 | $2 |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), r['kernel_launched'], 'chain', round((r.get('bound_actual') or {}).get('frac',0),3))"; }
for rep in 1 2 3; do for w in "--workload lunarlander_mlp_4096x50" "--workload lunarlander_mlp_4096x50 --rng philox"; do
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_bpshbm.so;  run "sequential descent for four actions (before)" "$w"
  unset SMZ_LIB_PATH;                                        run "block-parallel selection, root on a quad     " "$w"
done; done 2>&1 | tee $O/r05_n_bps_a4_ab.txt
unset SMZ_LIB_PATH; run "headline (control)" ""
