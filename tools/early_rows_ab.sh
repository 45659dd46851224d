#!/bin/bash
# Same-box A/B of the early parent-row loads (the shipped library) against the loads where the network inputs are assembled
# (make -C stochastic-muzero_amd/csrc variant VARIANT=noearly EXTRA=-DSMZ_EARLY_ROWS=0), three repetitions.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py $2 --min-timed-seconds 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 | $2 |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4))"; }
for rep in 1 2 3; do for w in "" "--rng philox"; do
  unset SMZ_LIB_PATH; run "early rows (shipped)   " "$w"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_noearly.so; run "rows at the input stage" "$w"
done; done
unset SMZ_LIB_PATH
