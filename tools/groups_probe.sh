#!/bin/bash
# Independent env groups on their own streams at large batches (the tree kernel of one group overlaps the network kernel of
# another), with the network kernel leaving room on the CU (fewer wavefronts per workgroup): tools/groups_probe.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for envs in ${@:-1048576 262144}; do
for w in 12 8 6; do
for g in 1 2 3; do
  SMZ_MLP_MFMA_WAVES=$w python3 bench.py --envs $((envs / g * g)) --groups $g --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --min-timed-seconds 0.1 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$envs waves $w groups $g', round(d['value']/1e6,1), round(d['ms_per_step'],3))"
done; done; done
