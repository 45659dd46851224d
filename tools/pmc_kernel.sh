#!/bin/bash
# SQ counter passes on ONE kernel of a bench.py command (separate --pmc runs, never with trace domains):
#   tools/pmc_kernel.sh <kernel regex> <out name> <bench.py args...>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
K=$1; NAME=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmck_*
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-include-regex "$K" --output-format csv -d $O/pmck_$i -- python3 $R/bench.py "$@" > $O/pmck_$i.log 2>&1 || tail -2 $O/pmck_$i.log
done
python3 - "$K" "$NAME" <<'PY'
import csv, glob, collections, os, json, sys, re
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); O=R+"/gpurun_out"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(O+"/pmck_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        agg[(re.search(r"k_\w+(<[^>]*>)?", row["Kernel_Name"]) or re.search(r"\w+", row["Kernel_Name"])).group(0)][row["Counter_Name"]].append(float(row["Counter_Value"]))
out={k: {c: sum(v)/len(v) for c,v in cs.items()} for k,cs in agg.items()}
for k,c in out.items():
    wc=c.get("SQ_WAVE_CYCLES",0) or 1
    print(k, "waves", c.get("SQ_WAVES"), "VALU/wave", round(c.get("SQ_INSTS_VALU",0)/max(1,c.get("SQ_WAVES",1)),1),
          "valu_active/wavecyc", round(c.get("SQ_ACTIVE_INST_VALU",0)/wc,3), "wait_any", round(c.get("SQ_WAIT_ANY",0)/wc,3),
          "wait_lds", round(c.get("SQ_WAIT_INST_LDS",0)/wc,3), "mfma_busy", c.get("SQ_VALU_MFMA_BUSY_CYCLES"), "busy_cycles", c.get("SQ_BUSY_CYCLES"), "grbm", c.get("GRBM_GUI_ACTIVE"))
json.dump({"kernel_regex": sys.argv[1], "per_launch_mean": out}, open(O+"/pmc_"+sys.argv[2]+".json","w"), indent=1)
PY
rm -rf $O/pmck_*
