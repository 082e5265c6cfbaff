#!/bin/bash
# issue-side and traffic counters of the C3 street workload's kernels: bash tools/pmc_c3.sh [c3|c2|m1]   (outputs gpurun_out/pmcc)
set -o pipefail
W=${1:-c3}; R=$PWD; O=$R/gpurun_out/pmcc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/a -o a -- python3 $R/tools/raster_only.py $W 5 > $O/a.log 2>&1; echo "a rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 $R/tools/raster_only.py $W 5 > $O/f.log 2>&1; echo "f rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/b -o b -- python3 $R/tools/raster_only.py $W 5 > $O/b.log 2>&1; echo "b rc=$?"
cd $R
python - <<PY
import csv, collections, os, re
for f in ("gpurun_out/pmcc/a/a_counter_collection.csv","gpurun_out/pmcc/f/f_counter_collection.csv","gpurun_out/pmcc/b/b_counter_collection.csv"):
    if not os.path.exists(f):
        print("missing", f); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        m=re.search(r"(blend_\w+(<[^>]*>)?)", k)
        if m:
            key=m.group(1)
            agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(key, r["Counter_Name"])]+=1
    for k,v in agg.items():
        print(k, {c: "%.4g"%(x/max(n[(k,c)],1)) for c,x in v.items()})
PY
find $O -name "*.csv" -size +1M -delete
