#!/bin/bash
# Round 5, second GPU visit: the GPU tests the first visit did not reach, the loopback bench line (with its exit status), the
# pipelined learning_cycle test and lines.  Outputs: gpurun_out/r05_b_*
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r05_b
timeout 2400 python -m pytest tests/test_gpu_rccl_loopback.py tests/test_gpu_host_envs.py tests/test_gpu_records.py tests/test_gpu_multirank.py tests/test_gpu_loop.py \
    "tests/test_gpu_fullsize_parity.py::test_production_search_kernel_equals_oracle_on_every_tree" -m gpu -q -s 2>&1 | tail -60 > $O/${T}_pytest_new.log
tail -25 $O/${T}_pytest_new.log
S="--min-timed-seconds 3 --no-cpu-baseline"
line() { python3 -c "
import json,sys
try:
    d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; b=r.get('bound_actual') or {}
    print('$1'.split('/')[-1], round(d['value']/1e6,2),'M', round(d['ms_per_step'],4),'ms/step', (r.get('kernel_launched') or '')[:60], 'chain', round(b.get('frac',0),3), (d.get('timing') or {}).get('gather_overlap'))
except Exception as e: print('$1', 'FAILED', e)"; }
echo "== rccl loopback line"
python bench.py $S --rccl-loopback --no-roofline > $O/${T}_bench_rccl_loopback.json 2>$O/${T}_loop.err; echo "rc=$?"; line $O/${T}_bench_rccl_loopback.json; tail -12 $O/${T}_loop.err
python bench.py $S --rccl-loopback --no-roofline --gather-mode plain > $O/${T}_bench_rccl_loopback_plain.json 2>$O/${T}_loop_plain.err; echo "rc=$?"; line $O/${T}_bench_rccl_loopback_plain.json; tail -5 $O/${T}_loop_plain.err
echo "== end to end"
python bench.py $S --end-to-end > $O/${T}_bench_end_to_end.json 2>/dev/null; line $O/${T}_bench_end_to_end.json
python bench.py $S --end-to-end --pipeline 8 > $O/${T}_bench_end_to_end_pipelined.json 2>/dev/null; line $O/${T}_bench_end_to_end_pipelined.json
python bench.py $S --end-to-end --learning-cycle --pipeline 1 > $O/${T}_bench_learning_cycle_sync.json 2>$O/${T}_lc.err; line $O/${T}_bench_learning_cycle_sync.json; tail -3 $O/${T}_lc.err
python bench.py $S --end-to-end --learning-cycle --pipeline 8 > $O/${T}_bench_learning_cycle_pipelined.json 2>$O/${T}_lc.err; line $O/${T}_bench_learning_cycle_pipelined.json; tail -3 $O/${T}_lc.err
echo "== host env defaults"
python bench.py $S --no-roofline --host-env python > $O/${T}_bench_hostenv_python.json 2>/dev/null; line $O/${T}_bench_hostenv_python.json
python bench.py $S --no-roofline --host-env python --per-env-step > $O/${T}_bench_hostenv_python_per_env.json 2>/dev/null; line $O/${T}_bench_hostenv_python_per_env.json
