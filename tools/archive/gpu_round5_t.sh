#!/bin/bash
# Round 5, visit t: phase stamps of the vision search kernel, shipped library against the round's variant.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for lib in "" $R/gpurun_variants/libsmz_stage_c.so; do
  if [ -n "$lib" ]; then export SMZ_LIB_PATH=$lib; else unset SMZ_LIB_PATH; fi
  echo "== ${lib:-shipped}"; python tools/vision_phase_probe.py 1024 2>&1 | grep -v "amdgpu.ids" | tail -4
done 2>&1 | tee $O/r05_t_vision_phases.txt
