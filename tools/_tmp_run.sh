python -m pytest tests/test_gpu_records.py -x -q 2>&1 | grep -E "Error|error|^E|passed|failed" | head -20
for m in 1 4 8; do python bench.py --end-to-end --pipeline $m --min-timed-seconds 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pipeline', $m, round(d['value']/1e6,1), 'M', round(d['ms_per_step']*64,2), 'ms per 64-step iteration', d['timing']['blocks'], 'blocks')"; done
