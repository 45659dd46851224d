#!/bin/bash
# Round-2 evidence set: bench lines of every workload (roofline + cpu_baseline), rocprofv3 kernel stats of the headline
# command, TCC traffic passes and SQ counter passes on the dominant kernel.  Usage: tools/profile_round2.sh <tag>
TAG=${1:-r02_b}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python bench.py --steps 16 --warmup 3 2>$O/${TAG}_bench.err > $O/${TAG}_bench.json; tail -c 300 $O/${TAG}_bench.json; echo
python bench.py --steps 16 --warmup 3 --workload lunarlander_mlp_4096x50 2>/dev/null > $O/${TAG}_bench_lunar.json
python bench.py --steps 16 --warmup 3 --workload cartpole_mlp_4096x100 2>/dev/null > $O/${TAG}_bench_c100.json
python bench.py --steps 8 --warmup 2 --workload vision_resnet_1024x50 2>/dev/null > $O/${TAG}_bench_vision.json
python bench.py --steps 16 --warmup 3 --rng philox --no-cpu-baseline 2>/dev/null > $O/${TAG}_bench_philox.json
python bench.py --steps 8 --warmup 2 --host-env --no-cpu-baseline --no-roofline 2>/dev/null > $O/${TAG}_bench_hostenv.json
for f in $O/${TAG}_bench*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read()); r=d.get('roofline') or {}; c=d.get('cpu_baseline') or {}
print('$f'.split('/')[-1], round(d['value']/1e6,2),'M sims/s', round(d['ms_per_step'],4),'ms/step', 'roofline frac', round(r.get('frac',0),4), 'cpu', round(c.get('value',0)/1e6,3),'M on', c.get('cores'))"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python3 $R/bench.py --steps 16 --warmup 3 --no-cpu-baseline > $O/prof_$TAG.log 2>&1
f=$(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats.csv && head -6 "$f" | cut -c1-260
rm -rf $O/pmc_traffic
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --pmc $c --kernel-include-regex "k_search_mlp" --output-format csv -d $O/pmc_traffic -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --min-timed-seconds 0.01 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, json, os
O="$O"
agg=collections.defaultdict(list)
for f in glob.glob(O+"/pmc_traffic/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
m={c: sum(x)/len(x) for c,x in agg.items()}
out={"kernel": "k_search_mlp<2,2,1,false,true>", "workload": "cartpole_mlp_4096x50",
     "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum> --kernel-include-regex k_search_mlp -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline (three separate passes; tools/profile_round2.sh)",
     "FETCH_SIZE_KB_per_launch": m.get("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": m.get("WRITE_SIZE"),
     "TCC_HIT_sum": m.get("TCC_HIT_sum"), "TCC_MISS_sum": m.get("TCC_MISS_sum"), "launches": {c: len(x) for c,x in agg.items()}}
if m.get("FETCH_SIZE") and m.get("WRITE_SIZE"):
    out["hbm_bytes_per_launch_raw"]=(m["FETCH_SIZE"]+m["WRITE_SIZE"])*1024
    out["hbm_bytes_per_launch_read_x2"]=(2*m["FETCH_SIZE"]+m["WRITE_SIZE"])*1024
    out["l2_hit_rate"]=m["TCC_HIT_sum"]/(m["TCC_HIT_sum"]+m["TCC_MISS_sum"]) if m.get("TCC_HIT_sum") else None
    out["note"]="gfx950: FETCH_SIZE reads exactly half of a wide coalesced stream (MI355X_MICROARCH.md, HBM); this kernel's reads are mostly 4-16 B gathers, for which the counter is uncalibrated, so the raw sum is reported as traffic and the x2-read figure as an upper bound"
open(O+"/${TAG}_traffic_k_search_mlp.json","w").write(json.dumps(out, indent=1)); print(json.dumps(out)[:400])
PY
bash $R/tools/pmc_search.sh > $O/${TAG}_pmc.log 2>&1; cp $O/pmc_k_search_mlp.json $O/${TAG}_pmc_k_search_mlp.json 2>/dev/null; tail -2 $O/${TAG}_pmc.log | cut -c1-300
