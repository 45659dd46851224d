#!/bin/bash
# rocprofv3 kernel statistics of one env step at 1 M trees (step-wise kernels + matrix-core heads): tools/profile_1m.sh <tag>
TAG=${1:-r02_h}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof1m_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1m_$TAG -- python3 $R/bench.py --envs 1048576 --groups 1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --heads hip --min-timed-seconds 0.01 > /dev/null 2>&1
f=$(find $O/prof1m_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats_1Mtrees.csv && head -8 "$f" | cut -c1-90,200-330
