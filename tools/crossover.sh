#!/bin/bash
# Where the step-wise kernels (matrix-core network kernel, rows left in the tree) overtake the single-launch search:
# simulations/s of both at a few tree counts on one box.  Sets mcts.single_launch_max_trees.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for n in ${@:-8192 12288 16384 20480 24576 28672 32768}; do
  for mode in "" "--stepwise" "--stepwise forced"; do
    if [ "$mode" = "--stepwise forced" ]; then export SMZ_MLP_IN_PLACE_MIN=0 SMZ_MLP_MFMA_MIN=0; mode="--stepwise"; tag=forced; else unset SMZ_MLP_IN_PLACE_MIN SMZ_MLP_MFMA_MIN; tag=${mode:-single}; fi
    python3 bench.py --envs $n --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --min-timed-seconds 0.2 $mode 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($n, '$tag', round(d['value']/1e6,1), round(d['ms_per_step'],3))"
  done
done
