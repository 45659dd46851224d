#!/bin/bash
# Round 5, visit aa: select_block requests everything its branches read together with the children's fields
# (SMZ_SELECT_LOADS_FIRST): parity with the variant, A/B on the block-parallel-selection workloads.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_lf.so
timeout 2400 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_end_to_end.py tests/test_gpu_episode_parity.py -m gpu -q -x 2>&1 | tail -3
unset SMZ_LIB_PATH
run() { python bench.py $2 --min-timed-seconds 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 | $2 |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), r['kernel_launched'], 'chain', round((r.get('bound_actual') or {}).get('frac',0),3))"; }
for rep in 1 2 3; do for w in "" "--workload cartpole_mlp_4096x100" "--rng philox" "--rng philox --workload cartpole_mlp_4096x100"; do
  unset SMZ_LIB_PATH;                                   run "shipped (r05_w)     " "$w"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_lf.so;  run "block loads up front" "$w"
done; done 2>&1 | tee $O/r05_aa_select_loads_ab.txt
