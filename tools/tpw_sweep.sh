#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -8
for hd in hip torch; do
 for t in 4 8; do
  echo "== heads $hd tpw $t"; SMZ_TREES_PER_WAVE=$t python bench.py --steps 8 --warmup 2 --no-cpu-baseline --heads $hd 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2),'M sims/s', round(d['ms_per_step'],3),'ms/step  tree kernel', round(d['roofline']['mean_launch_us'],1),'us')"
 done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hip -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --heads hip > $R/gpurun_out/prof_hip.log 2>&1
f=$(find $R/gpurun_out/prof_hip -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-220
