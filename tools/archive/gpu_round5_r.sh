#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 2400 python -m pytest "tests/test_gpu_end_to_end.py::test_single_launch_search_equals_stepwise_search" tests/test_gpu_multirank.py::test_bench_starts_its_own_ranks tests/test_gpu_records.py -m gpu -q 2>&1 | tail -8
for a in "--end-to-end" "--end-to-end --learning-cycle --pipeline 1"; do python bench.py --min-timed-seconds 3 --no-cpu-baseline --no-roofline $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a |', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), (d.get('end_to_end') or {}))"; done
