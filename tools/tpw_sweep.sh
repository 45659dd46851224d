#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for w in cartpole_mlp_4096x50 lunarlander_mlp_4096x50 cartpole_mlp_4096x100; do python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'], round(d['value']/1e6,2),'M sims/s', round(d['ms_per_step'],3),'ms/step', d['config']['heads'], d['roofline']['kernel'][:16], round(d['roofline']['frac'],4))"; done
SMZ_SEARCH_WAVES=4 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('4 waves x 4 trees:', round(d['value']/1e6,2),'M sims/s')"
