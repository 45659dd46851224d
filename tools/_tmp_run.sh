python bench.py --workload cartpole_mlp_4096x100 --min-timed-seconds 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('c100', round(d['value']/1e6,1), r['kernel_launched'], r['bound_actual']['how'][:60])"
SMZ_SEARCH_WAVES=4 python bench.py --workload cartpole_mlp_4096x100 --envs 2048 --min-timed-seconds 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_episode_parity.py tests/test_gpu_end_to_end.py -x -q 2>&1 | tail -2
tools/bps_ab.sh 2>&1 | head -6
