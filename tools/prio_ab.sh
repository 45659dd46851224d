#!/bin/bash
# s_setprio of the tree phases / the network phase of k_search_mlp (default 0 / 3) against the builds in gpurun_variants/
# (libsmz_prio<X>.so: part 2 compiled with -DSMZ_PRIO_TREE=.. -DSMZ_PRIO_HEADS=..).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6,1), 'M sims/s', round(d['ms_per_step'],4), 'ms/step')"; }
for rep in 1 2; do
  run "default tree 0 / heads 3"
  for f in $R/gpurun_variants/libsmz_prio*.so; do v=$(basename $f .so); SMZ_LIB_PATH=$f run "$v        "; done
done
