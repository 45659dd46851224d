#!/bin/bash
# Round 5, visit e: paired tails (all branch combinations): probe dump comparison + timing, then the parity tests on the library.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in nopair pair; do tools/spec_rows_probe_$v /tmp/dump_$v.bin | tail -1; done
cmp /tmp/dump_nopair.bin /tmp/dump_pair.bin && echo "paired tails == per-row tails on 16384 row evaluations (bit for bit)"
for v in nopair pair; do echo "==== $v"; tools/spec_rows_probe_$v | grep -A1 -E "^2 rows|wavefronts" | grep -v "^--"; done 2>&1 | tee $O/r05_e_pair_tails_probe.txt
timeout 2400 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_episode_parity.py tests/test_gpu_end_to_end.py tests/test_gpu_mlp_heads.py tests/test_gpu_tree_parity.py tests/test_gpu_decode_floor.py -m gpu -q 2>&1 | tail -8
python bench.py --min-timed-seconds 4 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('headline', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), 'chain', round(r['bound_actual']['frac'],3), 'launch us', round(r['mean_launch_us'],1))"
