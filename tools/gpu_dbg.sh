#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for lib in "" gpurun_variants/libsmz_dec1.so; do
  echo "=== lib: ${lib:-shipped}"
  if [ -n "$lib" ]; then export SMZ_LIB_PATH=$R/$lib; else unset SMZ_LIB_PATH; fi
  timeout 900 python -m pytest "tests/test_gpu_episode_parity.py::test_every_tree_of_every_step_of_the_timed_loop_equals_the_oracle[reset-4096-70-3]" -m gpu -q -x 2>&1 | grep -E "^E  |^tests/.*Error|passed|failed|^>" | head -12
done
