#!/bin/bash
# Phase stamps with parts of the round switched off (results are meaningless then; only the timing of what is left counts):
# SMZ_DEBUG_SKIP bit 1 = no network evaluation (the "networks" phase is then the gather of the parent rows + LDS hand-off alone),
# bit 2 = no descent, bit 4 = no expansion / backup, bit 8 = no random-word staging.  One wavefront per SIMD (2048 envs, 4 waves).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for skip in 48 49 50 52 56; do
  echo "== SMZ_DEBUG_SKIP=$skip"
  SMZ_SEARCH_WAVES=4 SMZ_DEBUG_SKIP=$skip timeout 300 python3 bench.py --envs 2048 --steps 2 --warmup 1 --no-cpu-baseline --heads hip 2>&1 | grep "phase cycles" | tail -1
done
