#!/bin/bash
# One GPU-box visit of round 2: parity tests, smoke, bench lines, head-error report, kernel-trace profile.
# Usage: tools/gpu_round2.sh <tag> [pytest -k expression]
TAG=${1:-r02_a}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
if [ -n "$2" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x -k "$2" 2>&1 | tail -25 | tee gpurun_out/pytest_$TAG.log
else
  timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 | tee gpurun_out/pytest_$TAG.log
fi
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke_$TAG.log
timeout 300 python tools/head_error_report.py gpurun_out/head_errors_$TAG.json 2>&1 | tail -14
timeout 600 python bench.py --steps 16 --warmup 3 2>gpurun_out/bench_$TAG.err | tee gpurun_out/bench_$TAG.json
tail -3 gpurun_out/bench_$TAG.err
