#!/bin/bash
# TCC traffic passes on the single-launch vision search kernel (three separate --pmc runs): tools/profile_vision_traffic.sh <tag>
TAG=${1:-r02_k}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_traffic_vis
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --pmc $c --kernel-include-regex "k_search_vision" --output-format csv -d $O/pmc_traffic_vis -- python3 $R/bench.py --workload vision_resnet_1024x50 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --min-timed-seconds 0.01 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, json
O="$O"
agg=collections.defaultdict(list)
for f in glob.glob(O+"/pmc_traffic_vis/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
m={c: sum(x)/len(x) for c,x in agg.items()}
out={"kernel": "k_search_vision<2,true>", "workload": "vision_resnet_1024x50",
     "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum> --kernel-include-regex k_search_vision -- python3 bench.py --workload vision_resnet_1024x50 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline (three separate passes; tools/profile_vision_traffic.sh)",
     "FETCH_SIZE_KB_per_launch": m.get("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": m.get("WRITE_SIZE"),
     "TCC_HIT_sum": m.get("TCC_HIT_sum"), "TCC_MISS_sum": m.get("TCC_MISS_sum"), "launches": {c: len(x) for c,x in agg.items()}}
if m.get("FETCH_SIZE") and m.get("WRITE_SIZE"):
    out["hbm_bytes_per_launch_raw"]=(m["FETCH_SIZE"]+m["WRITE_SIZE"])*1024
    out["hbm_bytes_per_launch_read_x2"]=(2*m["FETCH_SIZE"]+m["WRITE_SIZE"])*1024
    out["l2_hit_rate"]=m["TCC_HIT_sum"]/(m["TCC_HIT_sum"]+m["TCC_MISS_sum"]) if m.get("TCC_HIT_sum") else None
open(O+"/${TAG}_traffic_k_search_vision.json","w").write(json.dumps(out, indent=1)); print(json.dumps(out)[:700])
PY
