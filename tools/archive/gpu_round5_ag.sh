#!/bin/bash
# Round 5, visit ag: the leaf's word before the path records also for trees in LDS (-DSMZ_LEAF_FIRST_LDS=1): A/B on the headline.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_tl1.so
timeout 2400 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_end_to_end.py -m gpu -q -x 2>&1 | tail -1
unset SMZ_LIB_PATH
run() { python bench.py $2 --min-timed-seconds 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 | $2 |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), r['kernel_launched'])"; }
for rep in 1 2 3 4; do for w in "" "--rng philox"; do
  unset SMZ_LIB_PATH;                                    run "shipped (r05_af)   " "$w"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_tl1.so;  run "leaf first in LDS  " "$w"
done; done 2>&1 | tee $O/r05_ag_leaf_first_lds_ab.txt
