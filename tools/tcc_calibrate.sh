#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / TCC request counters per access for scattered small pieces (tools/tcc_calibrate.hip); separate --pmc passes.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  n=$(echo $c | tr ' ' '_')
  rm -rf $O/cal_$n
  rocprofv3 --pmc $c --output-format csv -d $O/cal_$n -- $R/tools/tcc_calibrate > $O/cal_$n.log 2>&1 || tail -3 $O/cal_$n.log
done
rm -rf $O/cal_time; rocprofv3 --kernel-trace --stats --output-format csv -d $O/cal_time -- $R/tools/tcc_calibrate > $O/cal_time.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os, json
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); O=R+"/gpurun_out"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/cal_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
N=1<<22
out={}
for k,cs in sorted(agg.items()):
    rec={c: sum(v)/len(v) for c,v in cs.items()}
    line={"fetch_B_per_access": rec.get("FETCH_SIZE",0)*1024/N, "write_B_per_access": rec.get("WRITE_SIZE",0)*1024/N}
    for c in rec:
        if c not in ("FETCH_SIZE","WRITE_SIZE"): line[c+"_per_access"]=rec[c]/N
    out[k]=line
for f in glob.glob(O+"/cal_time/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Name"].split("(")[0]
        if k.startswith("void cal_descent"):
            out[k.replace("void ","")]={"launch_us": float(row["AverageNs"])/1e3, "trees": 1<<20, "levels": 6, "ns_per_tree": float(row["AverageNs"])/(1<<20)}
        if k in out:
            us=float(row["AverageNs"])/1e3
            out[k]["launch_us"]=us; out[k]["G_accesses_per_s"]=N/us/1e3
for k,line in sorted(out.items()):
    print(k, {a: round(b,2) for a,b in line.items()})
json.dump(out, open(O+"/tcc_calibration.json","w"), indent=1)
PY
