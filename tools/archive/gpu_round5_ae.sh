#!/bin/bash
# Round 5, visit ae: trees in global memory -- the leaf's action word requested before the path records' words (SMZ_LEAF_FIRST);
# three selection passes in flight (-DSMZ_SELECT_TWO_PASSES=3).  Parity with the variants, A/B against the shipped library.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in lfst np3; do
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_$v.so
  timeout 2400 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_end_to_end.py -m gpu -q -x 2>&1 | tail -1
done
unset SMZ_LIB_PATH
run() { python bench.py $2 --min-timed-seconds 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 | $2 |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), r['kernel_launched'], 'chain', round((r.get('bound_actual') or {}).get('frac',0),3))"; }
for rep in 1 2 3; do for w in "--workload cartpole_mlp_4096x100" "--rng philox --workload cartpole_mlp_4096x100"; do
  unset SMZ_LIB_PATH;                                     run "shipped (r05_ad)              " "$w"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_lfst.so;  run "leaf word first               " "$w"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_np3.so;   run "leaf word first + three passes" "$w"
done; done 2>&1 | tee $O/r05_ae_leaf_first_ab.txt
