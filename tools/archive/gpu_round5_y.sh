#!/bin/bash
# Round 5, visit y: (1) Philox handles compute the next round's words in the shadow of the parent rows' round trip; (2) the
# representation kernel's pad_store picks its batch-norm pair from registers.  Parity with the variant, A/B against the shipped library.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_shadow.so
timeout 2400 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_end_to_end.py tests/test_gpu_episode_parity.py tests/test_gpu_records.py tests/test_gpu_frames.py tests/test_gpu_decode_floor.py -m gpu -q -x 2>&1 | tail -3
unset SMZ_LIB_PATH
run() { python bench.py $2 --min-timed-seconds 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 | $2 |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), r['kernel_launched'])"; }
for rep in 1 2 3; do for w in "--rng philox" "--rng philox --workload cartpole_mlp_4096x100" "--workload vision_resnet_1024x50" ""; do
  unset SMZ_LIB_PATH;                                       run "shipped (r05_w)" "$w"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_shadow.so;  run "variant        " "$w"
done; done 2>&1 | tee $O/r05_y_shadow_ab.txt
