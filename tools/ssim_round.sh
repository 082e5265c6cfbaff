#!/bin/bash
# SSIM alone: tests, the two PMC passes and the bench leg: bash tools/ssim_round.sh <outdir>
R=$PWD; O=$R/${1:-gpurun_out/ssim}; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_ssim.py -m gpu -x -q > $O/pytest.log 2>&1; tail -1 $O/pytest.log
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/ssim_$ctr -o p -- python3 $R/tools/ssim_pmc.py > $O/ssim_pmc.log 2>&1; echo "ssim pmc $ctr rc=$?"
done
cd $R
python profiles/pmc_summary.py $O/ssim_FETCH_SIZE/p_counter_collection.csv $O/ssim_WRITE_SIZE/p_counter_collection.csv $O/ssim_pmc_traffic.json > $O/ssim_pmc_summary.txt 2>&1
cat $O/ssim_pmc_summary.txt
python - <<'PY'
import json, torch, bench
r = bench.bench_ssim(torch.device("cuda"), 20, 3) if hasattr(bench, "bench_ssim") else None
print(json.dumps(r)[:1500])
PY
