#!/bin/bash
# Same-box rocprofv3 kernel statistics of two builds of libsmz.so: tools/ab_prof.sh <other.so>
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for which in other tree; do
  rm -rf $R/gpurun_out/abprof_$which
  if [ $which = other ]; then export SMZ_LIB_PATH=$R/$1; else unset SMZ_LIB_PATH; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abprof_$which -- python3 $R/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-roofline --min-timed-seconds 0.1 > /dev/null 2>&1
  f=$(find $R/gpurun_out/abprof_$which -name "*kernel_stats.csv" | head -1)
  echo "== $which"; head -4 "$f" | cut -c1-60,150-260
done
