#!/bin/bash
# HBM traffic of the dominant kernel from the TCC counters (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
K=${1:-k_search_mlp}; EXTRA=${2:-}
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --pmc $c --kernel-include-regex "$K" --output-format csv -d $R/gpurun_out/pmc_traffic -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline $EXTRA > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, json, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
agg=collections.defaultdict(list)
for f in glob.glob(R+"/gpurun_out/pmc_traffic/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
out={c: sum(x)/len(x) for c,x in agg.items()}
out["launches"]={c: len(x) for c,x in agg.items()}
out["kernel"]="$K"
print(json.dumps(out))
open(R+"/gpurun_out/traffic_$K.json","w").write(json.dumps(out))
PY
