#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for st in 0 1; do echo "== lds_stage $st"; for n in 65536 1048576; do SMZ_LDS_STAGE=$st python bench.py --envs $n --steps 2 --warmup 1 --no-cpu-baseline --heads hip 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; t=r['tree_kernel_alone']
print(dict(envs=d['config']['envs_per_gpu'], Msims=round(d['value']/1e6,1), tree_us=round(t['mean_launch_us'],1), tree_GBs=round(t['achieved'],1), tree_frac=round(t['frac'],4)))"; done; done
