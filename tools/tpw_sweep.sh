#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -4
rm -f gpurun_out/sweep_hip.jsonl
./tools/sweep_envs.sh hip 2>&1 | tail -5
for w in lunarlander_mlp_4096x50 cartpole_mlp_4096x100; do python bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'], round(d['value']/1e6,2),'M sims/s', round(d['ms_per_step'],3),'ms/step')"; done
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --heads torch 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('torch heads + HIP graph:', round(d['value']/1e6,2),'M sims/s', round(d['ms_per_step'],3),'ms/step')"
