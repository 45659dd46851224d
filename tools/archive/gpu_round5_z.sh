#!/bin/bash
# Round 5, visit z: the batch staging functions of the step-wise kernels store their twisted words last too (8 serial store round
# trips per batch of 8 trees before): parity of the step-wise paths with the variant, tree-count sweep A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_stw.so
timeout 2400 python -m pytest tests/test_gpu_tree_parity.py tests/test_gpu_end_to_end.py tests/test_gpu_fullsize_parity.py tests/test_gpu_selfplay_seam.py -m gpu -q -x 2>&1 | tail -3
unset SMZ_LIB_PATH
rm -f $O/r05_z_sweep_*.jsonl
for rep in 1 2; do
  unset SMZ_LIB_PATH;                                    SWEEP_OUT=$O/r05_z_sweep_shipped.jsonl tools/sweep_envs.sh > /dev/null
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_stw.so;  SWEEP_OUT=$O/r05_z_sweep_variant.jsonl tools/sweep_envs.sh > /dev/null
done
for f in shipped variant; do echo "== $f"; cat $O/r05_z_sweep_$f.jsonl | cut -c1-230; done 2>&1 | tee $O/r05_z_stores_last_stepwise_ab.txt
