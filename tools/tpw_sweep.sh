#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for w in 2 4 8; do
  echo "== search waves $w"; SMZ_SEARCH_WAVES=$w python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --heads hip 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2),'M sims/s', round(d['ms_per_step'],3),'ms/step')"
done
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
rocprofv3 --pmc $grp --kernel-include-regex "k_search_mlp" --output-format csv -d $R/gpurun_out/pmc_mega -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --heads hip > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for f in glob.glob(R+"/gpurun_out/pmc_mega/**/*counter_collection.csv", recursive=True):
    agg=collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print({c: round(sum(x)/len(x)) for c,x in agg.items()})
PY
