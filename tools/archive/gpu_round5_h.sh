#!/bin/bash
# Round 5, visit h: the leaf named from the record lanes (SMZ_LEAF_LANES): parity tests, A/B against the paired-tails library.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_episode_parity.py tests/test_gpu_end_to_end.py -m gpu -q -x 2>&1 | tail -4
run() { python bench.py $2 --min-timed-seconds 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 | $2 |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4))"; }
for rep in 1 2 3; do for w in "" "--rng philox"; do
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_pair.so;  run "leaf read back by the tree's lane (round 5, paired tails)" "$w"
  unset SMZ_LIB_PATH;                                       run "leaf named from the record lanes (new)                  " "$w"
done; done 2>&1 | tee $O/r05_h_leaf_lanes_ab.txt
