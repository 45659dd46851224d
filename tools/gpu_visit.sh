#!/bin/bash
# One GPU visit, parameterised (round 6; replaces the per-visit scripts of round 5, tools/archive/gpu_round5_*.sh):
#   gpurun --timeout S -- tools/gpu_visit.sh <tag> <step> [<step> ...]
# steps (each writes gpurun_out/<tag>_<step>.*, tails go to stdout):
#   tests:<pytest -k expression or file list>   pytest -m gpu on a selection ("tests:all" = the whole GPU suite)
#   smoke                                       __graft_entry__.smoke()
#   bench[:<extra bench.py args>]               python3 bench.py <args>  -> <tag>_bench<n>.json
#   ab:<variant>[:<bench args>]                 same-box A/B: shipped library vs gpurun_variants/libsmz_<variant>.so, 2 x 2 runs
#   stats:<name>:<bench args>                   rocprofv3 --kernel-trace --stats of bench.py <args> -> <tag>_kernel_stats_<name>.csv
#   sh:<command>                                anything else
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=$1; shift
n=0
line() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline') or {}
        print(round(d['value']/1e6,1),'M sims/s',round(d['ms_per_step'],4),'ms/step', r.get('kernel_launched'), 'frac', r.get('frac'), 'tree_frac', (r.get('tree_kernel_alone') or {}).get('frac'))
        for a in d.get('also',[]): print('   also', a.get('workload'), round(a.get('value',0)/1e6,1) if 'value' in a else a.get('error'), a.get('ms_per_step'), a.get('kernel'), a.get('frac'))
"; }
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}; [ "$kind" = "$step" ] && rest=""
  echo "=== $TAG $step"
  case $kind in
    tests)
      if [ "$rest" = "all" ]; then timeout 3300 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $O/${TAG}_pytest.log
      else timeout 3000 python3 -m pytest -m gpu -q -x $rest 2>&1 | tail -8 | tee -a $O/${TAG}_pytest.log; fi ;;
    smoke) python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    bench) n=$((n+1)); python3 bench.py $rest > $O/${TAG}_bench$n.json 2> $O/${TAG}_bench$n.err; line < $O/${TAG}_bench$n.json; tail -2 $O/${TAG}_bench$n.err ;;
    ab)
      v=${rest%%:*}; a=${rest#*:}; [ "$v" = "$rest" ] && a=""
      for rep in 1 2; do
        echo -n "variant $v : "; SMZ_LIB_PATH=$R/gpurun_variants/libsmz_$v.so python3 bench.py --no-cpu-baseline --no-roofline --also-seconds 0 --min-timed-seconds 4 $a 2>/dev/null | line
        echo -n "shipped      : "; python3 bench.py --no-cpu-baseline --no-roofline --also-seconds 0 --min-timed-seconds 4 $a 2>/dev/null | line
      done | tee -a $O/${TAG}_ab_$v.txt ;;
    stats)
      name=${rest%%:*}; a=${rest#*:}
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $O/prof_$name && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 $R/bench.py $a > $O/${TAG}_stats_$name.log 2>&1 )
      f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/${TAG}_kernel_stats_$name.csv 2>/dev/null; head -12 $O/${TAG}_kernel_stats_$name.csv | cut -c1-200; rm -rf $O/prof_$name ;;
    sh) bash -c "$rest" 2>&1 | tail -40 ;;
  esac
done
