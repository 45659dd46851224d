#!/bin/bash
# Phase stamps of the production LDS-resident kernel with the block-parallel selection and with the level-by-level descent
# (variant builds: make -C stochastic-muzero_amd/csrc variant VARIANT=probe_bps EXTRA=-DSMZ_BPS_PROBE ;
#                  make ... variant VARIANT=probe_seq EXTRA="-DSMZ_BPS_PROBE -DSMZ_SELECT_BLOCKS=0"), two wavefronts per SIMD
# (4096 envs, 8-wave workgroups) and one (2048 envs, 4-wave workgroups: the uncontended chain).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in probe_bps probe_seq; do for cfg in "4096 8" "2048 4"; do set -- $cfg
  echo "== $v, envs $1, $2 waves per workgroup"
  SMZ_LIB_PATH=$R/gpurun_variants/libsmz_$v.so python3 tools/bps_probe.py $1 $2 2>&1 | grep -v amdgpu.ids
done; done
