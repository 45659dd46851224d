#!/bin/bash
# Round 5, visit ah: k_search_vision with the block-parallel selection + parent planes requested behind the chase (SMZ_VISION_BPS):
# parity with the variant (every test that touches the vision family), phase stamps, A/B against the shipped library.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_vyv.so
timeout 2400 python -m pytest tests -m gpu -q -x -k "vision or image or frame" 2>&1 | tail -3
unset SMZ_LIB_PATH
for lib in "" $R/gpurun_variants/libsmz_vbps.so $R/gpurun_variants/libsmz_vyv.so; do
  if [ -n "$lib" ]; then export SMZ_LIB_PATH=$lib; else unset SMZ_LIB_PATH; fi
  echo "== ${lib:-shipped}"; python tools/vision_phase_probe.py 1024 2>&1 | grep -v "amdgpu.ids" | tail -2
done 2>&1 | tee $O/r05_ah_vision_bps_phases.txt
run() { python bench.py $2 --min-timed-seconds 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 | $2 |', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), r['kernel_launched'])"; }
for rep in 1 2 3; do
  unset SMZ_LIB_PATH;                                     run "shipped (r05_af)         " "--workload vision_resnet_1024x50"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_vbps.so;  run "block-parallel selection " "--workload vision_resnet_1024x50"
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_vyv.so;   run "... + stored value terms " "--workload vision_resnet_1024x50"
done 2>&1 | tee $O/r05_ah_vision_bps_ab.txt
