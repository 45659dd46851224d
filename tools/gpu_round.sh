#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, kernel-trace profile.  Usage: tools/gpu_round.sh <tag> [pytest args]
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -m gpu -q ${2:-} 2>&1 | tail -${3:-15} | tee gpurun_out/pytest_$TAG.log
python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke_$TAG.log
python bench.py --steps 16 --warmup 3 2>gpurun_out/bench_$TAG.err | tee gpurun_out/bench_$TAG.json
tail -5 gpurun_out/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
tail -3 $R/gpurun_out/prof_$TAG.log
f=$(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-300 | grep -v "^\"Cijk\|elementwise_kernel\|CatArray\|rocclr"
