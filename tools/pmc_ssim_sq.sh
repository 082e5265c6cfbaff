#!/bin/bash
# issue-side counters of the SSIM kernels: bash tools/pmc_ssim_sq.sh   (outputs gpurun_out/pmcs)
set -o pipefail
R=$PWD; O=$R/gpurun_out/pmcs; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/a -o a -- python3 $R/tools/ssim_pmc.py > $O/a.log 2>&1; echo "a rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $O/b -o b -- python3 $R/tools/ssim_pmc.py > $O/b.log 2>&1; echo "b rc=$?"
cd $R
python - <<PY
import csv, collections, os, re
for f in ("gpurun_out/pmcs/a/a_counter_collection.csv","gpurun_out/pmcs/b/b_counter_collection.csv"):
    if not os.path.exists(f):
        print("missing", f); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        m=re.search(r"ssim_\w+", k)
        if m:
            key=m.group(0)
            agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(key, r["Counter_Name"])]+=1
    for k,v in agg.items():
        print(k, {c: "%.4g"%(x/max(n[(k,c)],1)) for c,x in v.items()})
PY
find $O -name "*.csv" -size +1M -delete
