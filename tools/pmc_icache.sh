#!/bin/bash
# Instruction-cache counters of the single-launch search kernel (its round loop is ~45 KB of code; the instruction cache is
# 64 KB per two CUs).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmci_*
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-include-regex "k_search_mlp" --output-format csv -d $R/gpurun_out/pmci_$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmci_$i.log 2>&1
  tail -1 $R/gpurun_out/pmci_$i.log | cut -c1-120
done
python3 - <<'PY'
import csv, glob, collections, os, json
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
out={}
for f in sorted(glob.glob(R+"/gpurun_out/pmci_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for c,x in agg.items(): out[c]=sum(x)/len(x)
print(json.dumps(out, indent=1))
open(R+"/gpurun_out/pmc_icache_k_search_mlp.json","w").write(json.dumps(out, indent=1))
PY
