#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_mlp
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE"; do
rocprofv3 --pmc $grp --kernel-include-regex "k_mlp_recurrent" --output-format csv -d $R/gpurun_out/pmc_mlp -- python3 $R/bench.py --envs 65536 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --heads hip --stepwise > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for f in glob.glob(R+"/gpurun_out/pmc_mlp/**/*counter_collection.csv", recursive=True):
    agg=collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print({c: round(sum(x)/len(x)) for c,x in agg.items()})
PY
