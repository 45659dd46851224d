#!/bin/bash
# Round 5, visit l: large batches, the fused tree kernel with Philox streams in its specialised form (k_expand_backup<.., AEX, PHX>) against MT19937 and against the
# generic Philox kernels of the previous library; parity of the Philox step-wise path.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest "tests/test_gpu_fullsize_parity.py::test_philox_mode_equals_the_oracle_drawing_from_the_same_counter_stream" tests/test_gpu_tree_parity.py -m gpu -q -x 2>&1 | tail -3
run() { python bench.py --envs $2 $3 --steps 4 --warmup 2 --no-cpu-baseline --min-timed-seconds 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; t=r['tree_kernel_alone']; print('$1 | envs $2 $3 |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],3), 'ms/step | tree kernel', round(t['mean_launch_us'],1), 'us', round(t['achieved'],1), 'GB/s frac', round(t['frac'],4))"; }
for envs in 262144 1048576; do
  unset SMZ_LIB_PATH;                                        run "MT19937 (parity mode)              " $envs ""
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_bpshbm.so;  run "Philox, generic step-wise kernels  " $envs "--rng philox"
  unset SMZ_LIB_PATH;                                        run "Philox, specialised fused kernel   " $envs "--rng philox"
done 2>&1 | tee $O/r05_l_philox_large.txt
