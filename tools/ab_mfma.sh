#!/bin/bash
# Same-box A/B of the heads evaluation inside the single-launch search:
#   pk    = round-1 arithmetic (packed even/odd FMA accumulators; tools/libsmz_pk.so built with -DSMZ_DENSE_PK=1)
#   chain = k-ordered fma chain on the vector units (SMZ_SEARCH_MFMA=0)
#   mfma  = the same chain on the matrix cores, 16 leaves per workgroup (SMZ_SEARCH_MFMA=1)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
run() { python3 $R/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-roofline $EXTRA 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6,1), 'M sims/s', round(d['ms_per_step'],4), 'ms/step  p10-p90', [round(x,2) for x in d['timing']['block_ms_p10_p90']])"; }
for rep in 1 2; do
  SMZ_LIB_PATH=$R/tools/libsmz_pk.so SMZ_SEARCH_MFMA=0 run pk
  SMZ_SEARCH_MFMA=0 run chain
  SMZ_SEARCH_MFMA=1 run mfma
done
