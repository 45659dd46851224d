#!/bin/bash
# Round 5, third GPU visit: parity of the early-parent-row kernel, its same-box A/B, the loopback gather sweep.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r05_c
timeout 2400 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_episode_parity.py tests/test_gpu_rccl_loopback.py tests/test_gpu_end_to_end.py -m gpu -q 2>&1 | tail -15 > $O/${T}_pytest.log
tail -8 $O/${T}_pytest.log
echo "== early rows A/B"; tools/early_rows_ab.sh 2>&1 | tee $O/${T}_early_rows_ab.txt
S="--min-timed-seconds 3 --no-cpu-baseline --no-roofline --rccl-loopback"
echo "== loopback gather sweep"
for rep in 1 2; do
for m in "--gather-mode plain" "--gather-slices 1" "--gather-slices 2" "--gather-slices 4"; do
  python bench.py $S $m 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['timing']; print('$m |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), 'ms/step | exchange alone', round(t['gather_ms_median'],3), 'ms | exposed', t['gather_overlap']['exposed_ms_median'])"
done; done 2>&1 | tee $O/${T}_loopback_gather_sweep.txt
python bench.py --min-timed-seconds 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('no process group |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4))" | tee -a $O/${T}_loopback_gather_sweep.txt
