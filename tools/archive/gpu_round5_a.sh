#!/bin/bash
# Round 5, first GPU visit: the new GPU tests (RCCL loopback, K = 4 at full size, host envs, records, multirank), the stage-1
# probe of the speculative-evaluation study, the K = 4 / C5-in-LDS / host-env / loopback bench lines.  Outputs: gpurun_out/r05_a_*
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r05_a
timeout 1500 python -m pytest tests/test_gpu_rccl_loopback.py tests/test_gpu_host_envs.py tests/test_gpu_records.py tests/test_gpu_multirank.py \
    "tests/test_gpu_fullsize_parity.py::test_production_search_kernel_equals_oracle_on_every_tree" -m gpu -q -x -s 2>&1 | tail -40 > $O/${T}_pytest_new.log
tail -15 $O/${T}_pytest_new.log
echo "== stage-1 probe"; timeout 300 tools/spec_rows_probe > $O/${T}_spec_rows_probe.txt 2>&1; cat $O/${T}_spec_rows_probe.txt
S="--min-timed-seconds 3 --no-cpu-baseline"
line() { python3 -c "
import json,sys
try:
    d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; b=r.get('bound_actual') or {}
    print('$1'.split('/')[-1], round(d['value']/1e6,2),'M', round(d['ms_per_step'],4),'ms/step', (r.get('kernel_launched') or '')[:60], 'chain', round(b.get('frac',0),3))
except Exception as e: print('$1', 'FAILED', e)"; }
echo "== K4"; python bench.py $S --workload lunarlander_mlp_4096x50_K4 2>$O/${T}_k4.err > $O/${T}_bench_lunar_K4.json; line $O/${T}_bench_lunar_K4.json
python bench.py $S --workload lunarlander_mlp_4096x50 2>/dev/null > $O/${T}_bench_lunar.json; line $O/${T}_bench_lunar.json
echo "== C5 per-rank shape: trees in HBM (shipped) against LDS-resident in four-wave workgroups"
for rep in 1 2; do
  python bench.py $S --workload cartpole_mlp_4096x100 2>/dev/null > $O/${T}_bench_c100_hbm_$rep.json; line $O/${T}_bench_c100_hbm_$rep.json
  SMZ_SEARCH_WAVES=4 python bench.py $S --workload cartpole_mlp_4096x100 2>/dev/null > $O/${T}_bench_c100_lds4w_$rep.json; line $O/${T}_bench_c100_lds4w_$rep.json
done
python bench.py $S --workload cartpole_mlp_4096x100 --rng philox 2>/dev/null > $O/${T}_bench_c100_hbm_philox.json; line $O/${T}_bench_c100_hbm_philox.json
SMZ_SEARCH_WAVES=4 python bench.py $S --workload cartpole_mlp_4096x100 --rng philox 2>/dev/null > $O/${T}_bench_c100_lds4w_philox.json; line $O/${T}_bench_c100_lds4w_philox.json
echo "== headline"; python bench.py --min-timed-seconds 4 2>/dev/null > $O/${T}_bench.json; line $O/${T}_bench.json
echo "== host envs (batched slices)"
for cfg in "0 1" "4 1" "8 1" "16 1" "8 2" "16 2"; do set -- $cfg
  python bench.py $S --no-roofline --host-env python --host-workers $1 --groups $2 2>$O/${T}_he.err > $O/${T}_bench_hostenv_python_w$1_g$2.json; line $O/${T}_bench_hostenv_python_w$1_g$2.json
done
python bench.py $S --no-roofline --host-env native 2>/dev/null > $O/${T}_bench_hostenv_native.json; line $O/${T}_bench_hostenv_native.json
echo "== rccl loopback line"; python bench.py $S --rccl-loopback --no-roofline 2>$O/${T}_loop.err > $O/${T}_bench_rccl_loopback.json; line $O/${T}_bench_rccl_loopback.json; tail -3 $O/${T}_loop.err
