#!/bin/bash
# Same-box A/B of library builds at large batches: tools/ab_libs.sh <envs> <name> [<name> ...]  (gpurun_variants/libsmz_<name>.so)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
envs=$1; shift
for rep in 1 2; do
for v in "$@"; do
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_$v.so
  python3 bench.py --envs $envs --steps 2 --warmup 1 --no-cpu-baseline --min-timed-seconds 0.1 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$envs $v', round(d['value']/1e6,1), round(d['ms_per_step'],3), round(d['roofline']['mean_launch_us'],1))"
done; done
