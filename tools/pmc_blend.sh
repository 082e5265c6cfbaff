set -o pipefail
R=$PWD; O=$R/gpurun_out/pmcb; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/a -o a -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sdf > $O/a.log 2>&1; echo "a rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE --output-format csv -d $O/b -o b -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sdf > $O/b.log 2>&1; echo "b rc=$?"
cd $R
python - <<PY
import csv, collections
for f in ("gpurun_out/pmcb/a/a_counter_collection.csv","gpurun_out/pmcb/b/b_counter_collection.csv"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "blend_bwd_kernel" in k or "blend_fwd_wave" in k or "occl_budget" in k:
            key=k.split("(")[0][-40:]
            agg[key][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in agg.items():
        print(k, {c: "%.3g"%x for c,x in v.items()})
PY
find $O -name "*.csv" -size +1M -delete
