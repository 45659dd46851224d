#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_vis
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vis -- python3 $R/bench.py --steps 4 --warmup 2 --workload vision_resnet_1024x50 --no-cpu-baseline --no-roofline --min-timed-seconds 0.01 $1 > /dev/null 2>&1
f=$(find $R/gpurun_out/prof_vis -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-200
