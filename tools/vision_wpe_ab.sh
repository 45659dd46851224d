#!/bin/bash
# VERDICT r3 item 6, measured: k_search_vision register-allocated for TWO wavefronts per SIMD (gpurun_variants/libsmz_vis2.so =
# `make -C stochastic-muzero_amd/csrc variant VARIANT=vis2 EXTRA=-DSMZ_VISION_WPE=2`: 256 registers a lane, the tower weights spill)
# against the shipped one-wavefront-per-SIMD kernel (512 registers, weights resident), at 1024 / 2048 / 4096 image envs -- from 2048
# envs on a CU holds two workgroups of the variant at once, which is the regime the withdrawn eight-wave design was meant for.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --workload vision_resnet_1024x50 --envs $1 --steps 8 --warmup 2 --min-timed-seconds 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$2 envs $1: %.2f M simulations/s, %.3f ms per env step, search kernel %.1f us' % (d['value']/1e6, d['ms_per_step'], r['mean_launch_us']))"; }
for n in 1024 2048 4096; do
  unset SMZ_LIB_PATH; run $n "shipped (1 wavefront per SIMD)  "
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_vis2.so; run $n "waves_per_eu(2, 2) variant      "
done
