R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for envs in 16384 20480 32768; do for rep in 1 2; do for v in base n16up; do
  export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_$v.so
  python3 bench.py --envs $envs --stepwise --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --min-timed-seconds 0.2 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$envs $v', round(d['value']/1e6,1), round(d['ms_per_step'],3))"
done; done; done
