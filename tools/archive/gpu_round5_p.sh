#!/bin/bash
# Round 5, visit p: phase stamps of the final kernel, sets A and B (variant builds -DSMZ_BPS_PROBE=1 / =2), one and two wavefronts per SIMD.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in probe_a probe_b; do for cfg in "4096 8" "2048 4"; do set -- $cfg
  echo "== $v, envs $1, $2 waves per workgroup"
  SMZ_LIB_PATH=$R/gpurun_variants/libsmz_$v.so python3 tools/bps_probe.py $1 $2 2>&1 | grep -v amdgpu.ids | tail -2
done; done | tee $O/r05_p_bps_probe.txt
