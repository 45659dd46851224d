#!/bin/bash
# Round 5, visit m: TCC traffic of the fused tree kernel at 1 M trees, MT19937 against Philox streams (three separate --pmc passes each).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in mt19937 philox; do
  rm -rf $O/pmc_traffic_1m
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    rocprofv3 --pmc $c --kernel-include-regex "k_expand_backup" --output-format csv -d $O/pmc_traffic_1m -- python3 $R/bench.py --envs 1048576 --groups 1 --steps 1 --warmup 1 --rng $mode --no-cpu-baseline --no-roofline --min-timed-seconds 0.01 > /dev/null 2>&1
  done
  python3 - $mode <<PY
import csv, glob, collections, json, sys
O="$O"; mode=sys.argv[1]
agg=collections.defaultdict(list); names=collections.Counter()
for f in glob.glob(O+"/pmc_traffic_1m/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_expand_backup" in row["Kernel_Name"] and ", true, " in row["Kernel_Name"]:      # the fused (FUSE_SELECT) launches
            agg[row["Counter_Name"]].append(float(row["Counter_Value"])); names[row["Kernel_Name"].split("(")[0]]+=1
m={c: sum(x)/len(x) for c,x in agg.items()}
out={"kernel": names.most_common(1)[0][0] if names else None, "rng": mode, "trees": 1048576,
     "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum> --kernel-include-regex k_expand_backup -- python3 bench.py --envs 1048576 --groups 1 --steps 1 --warmup 1 --rng %s (three passes; tools/gpu_round5_m.sh)" % mode,
     "FETCH_SIZE_KB_per_launch": m.get("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": m.get("WRITE_SIZE"), "TCC_HIT_sum": m.get("TCC_HIT_sum"), "TCC_MISS_sum": m.get("TCC_MISS_sum"),
     "launches": {c: len(x) for c, x in agg.items()}}
if m.get("FETCH_SIZE") and m.get("WRITE_SIZE"):
    out["hbm_bytes_per_launch_raw"] = (m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
    out["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) if m.get("TCC_HIT_sum") else None
open(O+"/r05_m_traffic_k_expand_backup_1Mtrees_%s.json" % mode, "w").write(json.dumps(out, indent=1)); print(json.dumps(out)[:700])
PY
done
