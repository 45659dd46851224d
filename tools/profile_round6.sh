#!/bin/bash
# Round-6 evidence set (round 5's + the K = 4 instantiation's counters; the counter files are collected FIRST so that the bench lines carry roofline.traffic and roofline.compute): bench lines of every workload (roofline incl. bound_actual, cpu_baseline), rocprofv3 kernel stats of
# the headline command, TCC traffic passes on the dominant kernel keyed on the instantiation that ran and on the kernel
# sources' hash (bench.py find_traffic).  Usage: tools/profile_round6.sh <tag>   (outputs under gpurun_out/)
TAG=${1:-r06_z}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
cd /tmp && export TMPDIR=/tmp
traffic() {   # $1 = kernel regex, $2 = workload key, $3 = extra bench args, $4 = output name
  rm -rf $O/pmc_traffic
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    rocprofv3 --pmc $c --kernel-include-regex "$1" --output-format csv -d $O/pmc_traffic -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --also-seconds 0 --min-timed-seconds 0.01 $3 > /dev/null 2>&1
  done
  python3 - "$1" "$2" "$3" "$4" <<PY
import csv, glob, collections, json, os, sys, re
sys.path.insert(0, "$R")
import bench
O="$O"; rx, workload, extra, name = sys.argv[1:5]
agg=collections.defaultdict(list); kernels=collections.Counter()
for f in glob.glob(O+"/pmc_traffic/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Counter_Name"]].append(float(row["Counter_Value"])); kernels[row["Kernel_Name"]]+=1
m={c: sum(x)/len(x) for c,x in agg.items()}
kn=kernels.most_common(1)[0][0] if kernels else ""
mm=re.search(r"(k_search_\w+<[^>]*>)", kn)
name=mm.group(1) if mm else kn
if name.startswith("k_search_vision<") and name.endswith(", false>") and name.count(",") == 2:
    name=name[:-len(", false>")]+">"        # smz_last_kernel names the MT19937 instantiation without its defaulted PHX parameter
out={"kernel": name, "kernel_name_as_profiled": kn, "workload": workload, "source_sha16": bench.kernel_source_sha16(),
     "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum> --kernel-include-regex %s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline %s (three separate passes; tools/profile_round6.sh)" % (rx, extra),
     "FETCH_SIZE_KB_per_launch": m.get("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": m.get("WRITE_SIZE"),
     "TCC_HIT_sum": m.get("TCC_HIT_sum"), "TCC_MISS_sum": m.get("TCC_MISS_sum"), "launches": {c: len(x) for c,x in agg.items()}}
if m.get("FETCH_SIZE") and m.get("WRITE_SIZE"):
    out["hbm_bytes_per_launch_raw"]=(m["FETCH_SIZE"]+m["WRITE_SIZE"])*1024
    out["hbm_bytes_per_launch_read_x2"]=(2*m["FETCH_SIZE"]+m["WRITE_SIZE"])*1024
    out["l2_hit_rate"]=m["TCC_HIT_sum"]/(m["TCC_HIT_sum"]+m["TCC_MISS_sum"]) if m.get("TCC_HIT_sum") else None
    out["note"]="gfx950: FETCH_SIZE reads exactly half of a wide coalesced stream (MI355X_MICROARCH.md, HBM); this kernel's reads are mostly 4-16 B gathers, for which the counter is uncalibrated, so the raw sum is reported as traffic and the x2-read figure as an upper bound"
open(O+"/"+name,"w").write(json.dumps(out, indent=1)); print(json.dumps(out)[:500])
PY
}
traffic "k_search_mlp" "cartpole_mlp_4096x50" "" "${TAG}_traffic_k_search_mlp.json"
traffic "k_search_vision" "vision_resnet_1024x50" "--workload vision_resnet_1024x50" "${TAG}_traffic_k_search_vision.json"
traffic "k_search_mlp" "cartpole_mlp_4096x50+philox" "--rng philox" "${TAG}_traffic_k_search_mlp_philox.json"
traffic "k_search_mlp" "cartpole_mlp_4096x100" "--workload cartpole_mlp_4096x100" "${TAG}_traffic_k_search_mlp_c100.json"
traffic "k_search_mlp" "lunarlander_mlp_4096x50" "--workload lunarlander_mlp_4096x50" "${TAG}_traffic_k_search_mlp_lunar.json"
traffic "k_search_mlp" "lunarlander_mlp_4096x50_K4" "--workload lunarlander_mlp_4096x50_K4" "${TAG}_traffic_k_search_mlp_lunar_K4.json"
cp $O/${TAG}_traffic_*.json $R/profiles/ 2>/dev/null; cd $R
# SQ counters of the headline kernel (separate --pmc passes; tools/pmc_search.sh)
cd $R; tools/pmc_search.sh > /dev/null 2>&1; python3 - <<PY
import json, sys
sys.path.insert(0, "$R")
import bench
d = json.load(open("$O/pmc_k_search_mlp.json"))
d = {"kernel": "k_search_mlp<2, 2, 1, false, true, false, false, true>", "workload": "cartpole_mlp_4096x50", "source_sha16": bench.kernel_source_sha16(),
     "command": "tools/pmc_search.sh (rocprofv3 --pmc <group> --kernel-include-regex k_search_mlp -- python3 bench.py --steps 2 --warmup 1 ...; four separate passes)",
     "per_launch_mean": d}
open("$O/${TAG}_pmc_k_search_mlp.json", "w").write(json.dumps(d, indent=1)); print(json.dumps(d)[:300])
PY
cp $O/${TAG}_pmc_k_search_mlp.json $R/profiles/ 2>/dev/null

# ---- bench lines (counter files of THIS build are in profiles/ now: traffic and the VALU-issue bound are filled)
cd $R
python bench.py 2>$O/${TAG}_bench.err > $O/${TAG}_bench.json; tail -c 300 $O/${TAG}_bench.json; echo
S="--min-timed-seconds 3"
python bench.py $S --workload lunarlander_mlp_4096x50 2>/dev/null > $O/${TAG}_bench_lunar.json
python bench.py $S --workload lunarlander_mlp_4096x50_K4 2>/dev/null > $O/${TAG}_bench_lunar_K4.json
python bench.py $S --workload cartpole_mlp_4096x100 2>/dev/null > $O/${TAG}_bench_c100.json
python bench.py $S --steps 8 --warmup 2 --workload vision_resnet_1024x50 2>/dev/null > $O/${TAG}_bench_vision.json
python bench.py $S --rng philox --no-cpu-baseline 2>/dev/null > $O/${TAG}_bench_philox.json
python bench.py $S --end-to-end --no-cpu-baseline 2>/dev/null > $O/${TAG}_bench_end_to_end.json
python bench.py $S --end-to-end --pipeline 8 --no-cpu-baseline 2>/dev/null > $O/${TAG}_bench_end_to_end_pipelined.json
python bench.py $S --end-to-end --learning-cycle --pipeline 1 --no-cpu-baseline 2>/dev/null > $O/${TAG}_bench_learning_cycle_sync.json
python bench.py $S --end-to-end --learning-cycle --pipeline 8 --no-cpu-baseline 2>/dev/null > $O/${TAG}_bench_learning_cycle_pipelined.json
python bench.py $S --rccl-loopback --no-roofline 2>/dev/null > $O/${TAG}_bench_rccl_loopback.json
python bench.py $S --rccl-loopback --no-roofline --gather-mode overlapped 2>/dev/null > $O/${TAG}_bench_rccl_loopback_overlapped.json
python bench.py $S --host-env python --per-env-step --no-cpu-baseline --no-roofline 2>/dev/null > $O/${TAG}_bench_hostenv_python_per_env.json
python bench.py $S --steps 8 --warmup 2 --host-env native --no-cpu-baseline --no-roofline 2>/dev/null > $O/${TAG}_bench_hostenv_native.json
python bench.py $S --steps 8 --warmup 2 --host-env native --groups 2 --no-cpu-baseline --no-roofline 2>/dev/null > $O/${TAG}_bench_hostenv_native_2groups.json
python bench.py $S --host-env python --no-cpu-baseline --no-roofline 2>/dev/null > $O/${TAG}_bench_hostenv_python.json
python bench.py $S --host-env python --per-env-step --host-workers 0 --groups 1 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null > $O/${TAG}_bench_hostenv_python_serial.json
python bench.py $S --workload vision_resnet_1024x50 --steps 8 --warmup 2 --host-env python --no-cpu-baseline --no-roofline 2>/dev/null > $O/${TAG}_bench_vision_hostenv.json
for f in $O/${TAG}_bench*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read()); r=d.get('roofline') or {}; c=d.get('cpu_baseline') or {}; b=r.get('bound_actual') or {}
print('$f'.split('/')[-1], round(d['value']/1e6,2),'M sims/s', round(d['ms_per_step'],4),'ms/step', r.get('bound'), 'frac', round(r.get('frac',0),4), 'chain frac', round(b.get('frac',0),3), 'cpu', round(c.get('value',0)/1e6,3),'M on', c.get('cores'))"; done
rm -f $O/${TAG}_env_sweep.jsonl; SWEEP_OUT=$O/${TAG}_env_sweep.jsonl tools/sweep_envs.sh hip > /dev/null 2>&1; cat $O/${TAG}_env_sweep.jsonl
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python3 $R/bench.py --no-cpu-baseline --also-seconds 0 --min-timed-seconds 2 > $O/prof_$TAG.log 2>&1
f=$(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats.csv && head -6 "$f" | cut -c1-260
rm -rf $O/prof_${TAG}_v
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_v -- python3 $R/bench.py --workload vision_resnet_1024x50 --steps 8 --warmup 2 --no-cpu-baseline > $O/prof_${TAG}_v.log 2>&1
f=$(find $O/prof_${TAG}_v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats_vision.csv && head -5 "$f" | cut -c1-260
cp $O/prior_exactness.jsonl $O/${TAG}_prior_exactness.jsonl 2>/dev/null
