#!/bin/bash
# Per-kernel time of one vision env step (rocprofv3 kernel stats of the vision workload).  Usage: tools/vision_trace.sh <tag>
TAG=${1:-vt}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python3 $R/bench.py --workload vision_resnet_1024x50 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_$TAG.log 2>&1
f=$(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats_vision.csv && head -8 "$f" | cut -c1-200
rm -rf $O/prof_$TAG
