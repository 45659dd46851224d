#!/bin/bash
# A/B on one box: the descent touches the hidden rows it may need (default library) against a build without that
# (gpurun_variants/libsmz_notouch.so = -DSMZ_TOUCH_ROWS=0), headline + vision workloads.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline $2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6,1), 'M sims/s', round(d['ms_per_step'],4), 'ms/step')"; }
for rep in 1 2 3; do
  SMZ_LIB_PATH=$R/gpurun_variants/libsmz_notouch.so run "headline, no touch"
  run "headline, touch   "
done
for rep in 1 2; do
  SMZ_LIB_PATH=$R/gpurun_variants/libsmz_notouch.so run "vision (kernel unchanged), variant lib" "--workload vision_resnet_1024x50 --steps 8 --warmup 2"
  run "vision (kernel unchanged), default lib" "--workload vision_resnet_1024x50 --steps 8 --warmup 2"
  SMZ_LIB_PATH=$R/gpurun_variants/libsmz_notouch.so run "4096x100, no touch" "--workload cartpole_mlp_4096x100"
  run "4096x100, touch   " "--workload cartpole_mlp_4096x100"
done
