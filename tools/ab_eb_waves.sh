# A/B of the fused step-wise tree kernel register-allocated for 6/7/8 waves per SIMD (libs built with -DSMZ_EB_WAVES=N into gpurun_variants/)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for envs in 1048576 262144; do
for v in tree w6 w7 w8 tree; do
  if [ $v = tree ]; then unset SMZ_LIB_PATH; else export SMZ_LIB_PATH=$R/gpurun_variants/libsmz_$v.so; fi
  echo "== $envs $v"
  python3 bench.py --envs $envs --steps 2 --warmup 1 --no-cpu-baseline --min-timed-seconds 0.1 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['ms_per_step'], d.get('roofline'))"
done; done
