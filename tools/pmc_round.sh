#!/bin/bash
# PMC passes on the tree kernels (separate runs per counter group; no trace domains combined with --pmc).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
TAG=${1:-pmc}
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-include-regex "k_expand_backup|k_select|k_root_init" --output-format csv -d $R/gpurun_out/${TAG}_$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/${TAG}_$i.log 2>&1
  tail -2 $R/gpurun_out/${TAG}_$i.log | cut -c1-200
done
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for d in sorted(glob.glob(R+"/gpurun_out/%s_*/" % os.environ.get("TAG_","pmc"))):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"].split("(smz::Params")[0][-36:]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k,v in agg.items():
            print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, "n=",len(next(iter(v.values()))))
PY
