#!/bin/bash
# Round 5, visit q: the vision search kernel's tails on lane halves (SMZ_VISION_TAIL_HALVES): parity of the vision family, A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -q -x -k "vision or image or frame" 2>&1 | tail -4
run() { python bench.py --workload vision_resnet_1024x50 --steps 8 --warmup 2 --min-timed-seconds 3 --no-cpu-baseline $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 |', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), 'ms/step | search kernel', round(r['mean_launch_us'],1), 'us')"; }
for rep in 1 2 3; do
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_halves.so;  run "three wave-wide tails (before)      " ""
  unset SMZ_LIB_PATH;                                        run "decodes on lane halves + quad softmax" ""
done 2>&1 | tee $O/r05_q_vision_tails_ab.txt
