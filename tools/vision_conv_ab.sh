#!/bin/bash
# A/B on one box: k_search_vision with its 3x3 convolutions on the matrix cores (default library) against the vector-unit
# version (gpurun_variants/libsmz_convvalu.so = -DSMZ_VISION_CONV_MFMA=0).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python3 bench.py --workload vision_resnet_1024x50 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('$1', round(d['value']/1e6,2), 'M sims/s', round(d['ms_per_step'],4), 'ms/step', 'kernel us', r.get('mean_launch_us'))"; }
for rep in 1 2 3; do
  SMZ_LIB_PATH=$R/gpurun_variants/libsmz_convvalu.so run "convolutions on the vector unit "
  run "convolutions on the matrix cores"
done
SMZ_DEBUG_SKIP=16 python3 bench.py --workload vision_resnet_1024x50 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "k_search_vision phases" | tail -1
