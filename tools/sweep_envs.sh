#!/bin/bash
# Tree-count sweep: where the tree kernel leaves the latency-bound regime.  Prints one line per size.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for n in 4096 16384 65536 262144 1048576; do
  python bench.py --envs $n --steps 2 --warmup 1 --min-timed-seconds 1 --no-cpu-baseline --heads ${1:-hip} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; t=r['tree_kernel_alone']
print(json.dumps(dict(envs=d['config']['envs_per_gpu'], sims_per_s=round(d['value']), ms_per_step=round(d['ms_per_step'],3), dominant=r['kernel'][:14], dom_us=round(r['mean_launch_us'],1), dom_GBs=round(r['achieved'],1), tree_us=round(t['mean_launch_us'],1), tree_GBs=round(t['achieved'],1), tree_frac=round(t['frac'],4), depth=round(r['mean_depth'],2))))" | tee -a ${SWEEP_OUT:-gpurun_out/sweep_${1:-hip}.jsonl}
done
