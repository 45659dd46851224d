#!/bin/bash
# Round 5, visit s: the staging's waits.  (1) stage_finish stores the twisted words after all trees' words went to LDS
# (SMZ_STAGE_STORES_LAST); (2) the next round's source words are requested with the parent rows (SMZ_EARLY_STAGE).  Parity, A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_stage_c.so
timeout 2400 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_episode_parity.py tests/test_gpu_end_to_end.py tests/test_gpu_tree_parity.py tests/test_gpu_decode_floor.py -m gpu -q -x 2>&1 | tail -4
unset SMZ_LIB_PATH
run() { python bench.py $2 --min-timed-seconds 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 | $2 |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), r['kernel_launched'], 'chain', round((r.get('bound_actual') or {}).get('frac',0),3))"; }
for rep in 1 2; do for w in "" "--workload cartpole_mlp_4096x100" "--workload lunarlander_mlp_4096x50" "--workload lunarlander_mlp_4096x50_K4" "--rng philox"; do
  unset SMZ_LIB_PATH;                                        run "shipped (r05_h)            " "$w"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_stage_a.so;  run "stores last                " "$w"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_stage_b.so;  run "early source words         " "$w"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_stage.so;    run "both                       " "$w"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_stage_c.so;  run "both + explicit vmcnt(0)   " "$w"
done; done 2>&1 | tee $O/r05_s_stage_ab.txt
# vision: tower biases staged in LDS + stores last (stage_c) against the shipped kernel
for rep in 1 2; do
  unset SMZ_LIB_PATH;                                        run "shipped (r05_h)            " "--workload vision_resnet_1024x50"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_stage.so;    run "stores last                " "--workload vision_resnet_1024x50"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_stage_c.so;  run "stores last + biases in LDS" "--workload vision_resnet_1024x50"
done 2>&1 | tee $O/r05_s_vision_ab.txt
