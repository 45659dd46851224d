#!/bin/bash
# A/B on one box: the library built from HEAD (gpurun_variants/libsmz_head.so) against the working tree's library.
# Usage: tools/ab_head.sh [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "${@:2}" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6,1), 'M sims/s', round(d['ms_per_step'],4), 'ms/step')"; }
for rep in 1 2 3; do
  SMZ_LIB_PATH=$R/gpurun_variants/libsmz_head.so run "HEAD        " "$@"
  run "working tree" "$@"
done
