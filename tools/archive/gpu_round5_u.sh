#!/bin/bash
# Round 5, visit u: the rebuilt library (stores last + early source words + vision biases in LDS + vision offsets in lanes): parity,
# vision A/B against the previous library and the biases-only variant.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_end_to_end.py tests/test_gpu_episode_parity.py -m gpu -q -x 2>&1 | tail -3
run() { python bench.py $2 --min-timed-seconds 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 | $2 |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), r['kernel_launched'])"; }
for rep in 1 2 3; do
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_r05h.so;    run "r05_h library                   " "--workload vision_resnet_1024x50"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_stage_c.so; run "biases in LDS                   " "--workload vision_resnet_1024x50"
  unset SMZ_LIB_PATH;                                       run "biases in LDS + offsets in lanes" "--workload vision_resnet_1024x50"
done 2>&1 | tee $O/r05_u_vision_ab.txt
for w in "" "--workload cartpole_mlp_4096x100"; do
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_r05h.so;    run "r05_h library" "$w"
  unset SMZ_LIB_PATH;                                       run "rebuilt      " "$w"
done 2>&1 | tee $O/r05_u_mlp_check.txt
