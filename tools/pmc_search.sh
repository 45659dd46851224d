#!/bin/bash
# PMC passes on the single-launch search kernel (separate runs per counter group; no trace domains with --pmc).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcs_*
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-include-regex "k_search_mlp" --output-format csv -d $R/gpurun_out/pmcs_$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --also-seconds 0 --min-timed-seconds 0.05 > $R/gpurun_out/pmcs_$i.log 2>&1
  tail -1 $R/gpurun_out/pmcs_$i.log | cut -c1-160
done
python3 - <<'PY'
import csv, glob, collections, os, json
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
out={}
for f in sorted(glob.glob(R+"/gpurun_out/pmcs_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for c,x in agg.items(): out[c]=sum(x)/len(x)
print(json.dumps(out))
open(R+"/gpurun_out/pmc_k_search_mlp.json","w").write(json.dumps(out, indent=1))
PY
