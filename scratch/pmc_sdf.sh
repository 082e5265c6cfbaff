set -o pipefail
R=$PWD; O=$R/gpurun_out/pmcs; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM --output-format csv -d $O/a -o a -- python3 $R/scratch/sdf_pmc.py 1000000 131072 > $O/a.log 2>&1; echo "a rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $O/b -o b -- python3 $R/scratch/sdf_pmc.py 1000000 131072 > $O/b.log 2>&1; echo "b rc=$?"
cd $R
python - <<PY
import csv, collections
for f in ("gpurun_out/pmcs/a/a_counter_collection.csv","gpurun_out/pmcs/b/b_counter_collection.csv"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "sdf_forward" in k or "knn_search" in k:
            key=k.split("(")[0][-40:]
            agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(key,r["Counter_Name"])]+=1
    for k,v in agg.items():
        print(k, {c: "%.4g"%(x/n[(k,c)]) for c,x in v.items()})
PY
find $O -name "*.csv" -size +1M -delete
