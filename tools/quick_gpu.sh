#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -5
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --heads hip 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2),'M sims/s', round(d['ms_per_step'],3),'ms/step', d['config']['heads'])"
