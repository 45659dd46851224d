#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in 1 2 3; do
python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-roofline --heads hip 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2),'M sims/s', round(d['ms_per_step'],3),'ms/step', d['config']['heads'])"
done
git stash -q 2>/dev/null
