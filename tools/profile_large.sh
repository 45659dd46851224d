#!/bin/bash
# Large-batch regime: trees-per-GPU sweep + TCC traffic of the fused tree kernel at 1 M trees.  Usage: tools/profile_large.sh <tag>
TAG=${1:-r02_c}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/sweep_hip.jsonl; bash tools/sweep_envs.sh hip > /dev/null 2>&1; cp $O/sweep_hip.jsonl $O/${TAG}_env_sweep.jsonl; cat $O/${TAG}_env_sweep.jsonl
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_traffic_1m
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --pmc $c --kernel-include-regex "k_expand_backup" --output-format csv -d $O/pmc_traffic_1m -- python3 $R/bench.py --envs 1048576 --groups 1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --min-timed-seconds 0.01 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, json
O="$O"
agg=collections.defaultdict(list)
for f in glob.glob(O+"/pmc_traffic_1m/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_expand_backup" in row["Kernel_Name"] and "true" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
m={c: sum(x)/len(x) for c,x in agg.items()}
out={"kernel":"k_expand_backup<2,2,true,true> (expand + backup + next select)","trees":1048576,
     "command":"rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum> --kernel-include-regex k_expand_backup -- python3 bench.py --envs 1048576 --steps 1 --warmup 1 (three passes; tools/profile_large.sh)",
     "FETCH_SIZE_KB_per_launch":m.get("FETCH_SIZE"),"WRITE_SIZE_KB_per_launch":m.get("WRITE_SIZE"),"TCC_HIT_sum":m.get("TCC_HIT_sum"),"TCC_MISS_sum":m.get("TCC_MISS_sum"),
     "launches":{c:len(x) for c,x in agg.items()}}
if m.get("FETCH_SIZE") and m.get("WRITE_SIZE"):
    out["hbm_bytes_per_launch_raw"]=(m["FETCH_SIZE"]+m["WRITE_SIZE"])*1024
    out["hbm_bytes_per_launch_read_x2"]=(2*m["FETCH_SIZE"]+m["WRITE_SIZE"])*1024
open(O+"/${TAG}_traffic_k_expand_backup_1Mtrees.json","w").write(json.dumps(out,indent=1)); print(json.dumps(out)[:600])
PY
