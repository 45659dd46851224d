#!/bin/bash
# A/B on one box: trees in LDS (k_search_mlp<..., TLDS>) against trees in global memory, headline workload.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('$1', round(d['value']/1e6,1), 'M sims/s', round(d['ms_per_step'],4), 'ms/step', r.get('kernel_launched'), r.get('mean_launch_us'))"; }
for rep in 1 2 3; do
  SMZ_SEARCH_TLDS=0 run "trees in global memory"
  SMZ_SEARCH_TLDS=1 run "trees in LDS          "
done
SMZ_SEARCH_TLDS=1 run "lunarlander, trees in LDS" "--workload lunarlander_mlp_4096x50"
SMZ_SEARCH_TLDS=0 run "lunarlander, global     " "--workload lunarlander_mlp_4096x50"
