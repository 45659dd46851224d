#!/bin/bash
# Same-box A/B of two builds of libsmz.so on the headline workload: tools/ab_lib.sh <other.so> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OTHER=$1; shift
run() { python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), 'M sims/s', round(d['ms_per_step'],4), 'ms/step')"; }
for rep in 1 2; do
  echo -n "other  "; SMZ_LIB_PATH=$R/$OTHER run "$@"
  echo -n "tree   "; run "$@"
done
