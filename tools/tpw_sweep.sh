#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for m in 16 48; do SMZ_DEBUG_SKIP=$m timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --heads hip 2>&1 | grep "phase cycles" | tail -1; done
for i in 1 2; do timeout 120 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('4096x50:', round(d['value']/1e6,2),'M sims/s', round(d['ms_per_step'],3))"; done
