#!/bin/bash
# Does a third wavefront per SIMD raise the search kernel's throughput per CU?  libsmz_w12.so = k_search_mlp register-allocated
# for 768-thread workgroups (168 VGPRs, 3 waves per SIMD); 12 waves x 2 trees = 24 trees per CU need 6144 envs to fill the chip.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python3 bench.py --envs $1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --min-timed-seconds 0.3 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', $1, 'envs', round(d['value']/1e6,1), 'M sims/s', round(d['ms_per_step'],4), 'ms/step', d['config']['search'][:40])"; }
for rep in 1 2; do
  run 4096 "default(8 waves)"
  SMZ_LIB_PATH=$R/gpurun_variants/libsmz_w12.so run 4096 "w12 lib, 8 waves (168 VGPR)"
  SMZ_LIB_PATH=$R/gpurun_variants/libsmz_w12.so SMZ_SEARCH_WAVES=12 run 6144 "w12 lib, 12 waves"
  SMZ_LIB_PATH=$R/gpurun_variants/libsmz_w12.so SMZ_SEARCH_WAVES=12 SMZ_SEARCH_TPW=1 run 3072 "w12 lib, 12 waves tpw1"
done
