# Large-batch tree kernel against occupancy: dynamic-LDS padding (SMZ_DEBUG_LDS_PAD) leaves 9 / 6 / 5 / 4 / 2 wavefronts per CU (17 KB of LDS per one-wavefront workgroup + the pad)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for pad in 0 9000 12000 19000 38000; do
  export SMZ_DEBUG_LDS_PAD=$pad
  python3 bench.py --envs 1048576 --steps 2 --warmup 1 --no-cpu-baseline --min-timed-seconds 0.1 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pad $pad', round(d['value']/1e6,1), round(d['ms_per_step'],3), round(d['roofline']['mean_launch_us'],1))"
done
