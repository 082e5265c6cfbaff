set -o pipefail
R=$PWD; O=$R/gpurun_out/pmcg; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM --output-format csv -d $O/a -o a -- python3 $R/scratch/sdf_step_big.py > $O/a.log 2>&1; echo "a rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VMEM_WR --output-format csv -d $O/b -o b -- python3 $R/scratch/sdf_step_big.py > $O/b.log 2>&1; echo "b rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --output-format csv -d $O/c -o c -- python3 $R/scratch/sdf_step_big.py > $O/c.log 2>&1; echo "c rc=$?"
cd $R
python - <<PY
import csv, collections, os
for f in ("gpurun_out/pmcg/a/a_counter_collection.csv","gpurun_out/pmcg/b/b_counter_collection.csv","gpurun_out/pmcg/c/c_counter_collection.csv"):
    if not os.path.exists(f): print("missing", f); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        for key in ("sdf_grad_mfma", "sdf_forward_mfma", "sdf_grad_kernel", "gather_sum"):
            if key in k:
                agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(key,r["Counter_Name"])]+=1
    for k,v in agg.items():
        print(k, {c: "%.4g"%(x/n[(k,c)]) for c,x in v.items()})
PY
tail -3 $O/c.log
find $O -name "*kernel_trace.csv" -delete
