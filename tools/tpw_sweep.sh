#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -4
for g in 1 2 4 8 16; do
 for t in 4 16; do
  echo "== groups $g tpw $t"; SMZ_TREES_PER_WAVE=$t python bench.py --steps 8 --warmup 2 --no-cpu-baseline --groups $g 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2),'M sims/s', round(d['ms_per_step'],3),'ms/step  tree kernel', round(d['roofline']['mean_launch_us'],1),'us')"
 done
done
