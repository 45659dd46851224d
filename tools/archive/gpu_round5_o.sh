#!/bin/bash
# Round 5, visit o: the narrow afterstate layer of a pair on lane halves (SMZ_DENSE_HALVES) + the one-element Q part without chains: probe dump comparison, parity, A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in pair1 h0 h1; do tools/spec_rows_probe_$v /tmp/dump_$v.bin | tail -1; done
cmp /tmp/dump_pair1.bin /tmp/dump_h0.bin && cmp /tmp/dump_pair1.bin /tmp/dump_h1.bin && echo "lane-halves layer + chain-free Q part == before on 16384 row evaluations (bit for bit)"
for v in h0 h1; do echo "==== $v"; tools/spec_rows_probe_$v | grep -A1 -E "^2 rows|wavefronts" | grep -v "^--"; done 2>&1 | tee $O/r05_o_halves_probe.txt
timeout 2400 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_episode_parity.py tests/test_gpu_end_to_end.py tests/test_gpu_mlp_heads.py -m gpu -q -x 2>&1 | tail -4
run() { python bench.py $2 --min-timed-seconds 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 | $2 |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4))"; }
for rep in 1 2 3; do for w in "" "--rng philox"; do
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_bpshbm.so;  run "before (set r05_g)                      " "$w"
  unset SMZ_LIB_PATH;                                        run "lane-halves layer + chain-free Q (new)  " "$w"
done; done 2>&1 | tee $O/r05_o_halves_ab.txt
