#!/bin/bash
# Same-box A/B of the block-parallel selection (the shipped library) against the level-by-level descent
# (make -C stochastic-muzero_amd/csrc variant VARIANT=noblocks EXTRA=-DSMZ_SELECT_BLOCKS=0), three repetitions.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py $2 --min-timed-seconds 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 | $2 |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4))"; }
for rep in 1 2 3; do for w in "" "--rng philox" "--workload lunarlander_mlp_4096x50"; do
  unset SMZ_LIB_PATH; run "block-parallel (shipped)" "$w"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_noblocks.so; run "level-by-level          " "$w"
done; done
