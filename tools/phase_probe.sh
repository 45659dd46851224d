#!/bin/bash
# s_memtime phase stamps of the specialised k_search_mlp (SMZ_DEBUG_SKIP=48) with two wavefronts per SIMD (the production
# geometry: 4096 envs, 8-wave workgroups) and with ONE wavefront per SIMD (2048 envs, 4-wave workgroups): the second gives the
# phases' uncontended durations -- the dependent-chain floor of a round (profiles/r03_ceiling.md).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in "4096 8" "2048 4" "1024 2"; do
  set -- $cfg
  echo "== envs $1, waves per workgroup $2 (per SIMD: $(( $2 / 4 )).$(( ($2 % 4) * 25 )))"
  SMZ_SEARCH_WAVES=$2 SMZ_DEBUG_SKIP=48 timeout 300 python3 bench.py --envs $1 --steps 2 --warmup 1 --no-cpu-baseline --heads hip 2>&1 | grep "phase cycles" | tail -1
  SMZ_SEARCH_WAVES=$2 timeout 300 python3 bench.py --envs $1 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', round(d['value']/1e6,1), 'M sims/s', round(d['ms_per_step'],4), 'ms/step')"
done
