#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -2
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('4096x50:', round(d['value']/1e6,2),'M sims/s', round(d['ms_per_step'],3))"
rm -f gpurun_out/sweep_hip.jsonl
./tools/sweep_envs.sh hip 2>&1 | tail -5
