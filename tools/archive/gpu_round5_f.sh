#!/bin/bash
# Round 5, visit f: scalar-register chase: parity tests, then the same-box A/B  per-row tails | paired tails | paired tails + scalar chase.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_episode_parity.py tests/test_gpu_end_to_end.py -m gpu -q -x 2>&1 | tail -6
run() { python bench.py $2 --min-timed-seconds 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 | $2 |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4))"; }
for rep in 1 2 3; do for w in "" "--rng philox"; do
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_nopair.so; run "early rows, per-row tails, LDS chase        " "$w"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_pair.so;   run "early rows, paired tails, LDS chase         " "$w"
  unset SMZ_LIB_PATH;                                        run "early rows, paired tails, scalar chase (new)" "$w"
done; done 2>&1 | tee $O/r05_f_pair_chase_ab.txt
unset SMZ_LIB_PATH
for w in "--workload lunarlander_mlp_4096x50" "--workload cartpole_mlp_4096x100" "--workload lunarlander_mlp_4096x50_K4"; do
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_nopair.so; run "per-row tails" "$w"; unset SMZ_LIB_PATH; run "shipped      " "$w"; done 2>&1 | tee -a $O/r05_f_pair_chase_ab.txt
