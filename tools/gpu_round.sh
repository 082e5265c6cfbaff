#!/bin/bash
# One GPU-box pass of a round (outputs under gpurun_out/r04<tag>): usage (repo root on the box): bash tools/gpu_round.sh <tag> <part>
# part 1: tests + smoke + bench;  part 2: rocprofv3 kernel stats + PMC passes + 2-rank rehearsal
set -o pipefail
tag=${1:-x}; part=${2:-1}
R=$PWD
O=$R/gpurun_out/r04$tag
mkdir -p $O
export TMPDIR=/tmp
if [ "$part" = "1" ]; then
  timeout -k 10 800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
  tail -3 $O/pytest.log
  timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
  timeout -k 10 500 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"
  tail -c 400 $O/bench.log
else
  cd /tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o s -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1; echo "prof rc=$?"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sdf > $O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sdf > $O/pmc_write.log 2>&1; echo "pmc write rc=$?"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/pmc_sq -o q -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sdf > $O/pmc_sq.log 2>&1; echo "pmc sq rc=$?"
  # SKIP_SDF_PMC=1: the SDF forward kernels did not change since the last sdf_pmc_traffic.json was taken
  cfgs=("1000000 131072" "1000000 16384" "200000 131072" "5000000 131072")
  [ -n "$SKIP_SDF_PMC" ] && cfgs=()
  for cfg in "${cfgs[@]}"; do
    set -- $cfg
    for ctr in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 150 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/sdf_${1}_${2}_$ctr -o p -- python3 $R/tools/sdf_pmc.py $1 $2 > $O/sdf_pmc.log 2>&1; echo "sdf pmc $1 $2 $ctr rc=$?"
    done
  done
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/ssim_$ctr -o p -- python3 $R/tools/ssim_pmc.py > $O/ssim_pmc.log 2>&1; echo "ssim pmc $ctr rc=$?"
  done
  cd $R
  python profiles/pmc_summary.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv $O/pmc_traffic.json $O/pmc_sq/q_counter_collection.csv '{"gaussians": 1000000, "width": 1920, "height": 1080, "mode": "surfel"}' > $O/pmc_summary.txt 2>&1; echo "pmc summary rc=$?"
  python profiles/pmc_summary.py $O/ssim_FETCH_SIZE/p_counter_collection.csv $O/ssim_WRITE_SIZE/p_counter_collection.csv $O/ssim_pmc_traffic.json > $O/ssim_pmc_summary.txt 2>&1; echo "ssim summary rc=$?"
  cp $O/prof/s_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
  # C3 street workload alone (kernel statistics) and the decoder kernels' matrix-pipe counters
  cd /tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o c -- python3 $R/tools/raster_only.py c3 40 > $O/c3.log 2>&1; echo "c3 stats rc=$?"
  cd $R
  cp $O/c3/c_kernel_stats.csv $O/c3_kernel_stats.csv 2>/dev/null
  bash tools/pmc_mlp.sh > $O/mlp_pmc.txt 2>&1; echo "mlp pmc rc=$?"
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --single-device --no-sdf --no-cpu-baseline --exchange-leg > $O/bench_2rank_gloo.log 2>&1; echo "2-rank rehearsal rc=$?"
  tail -c 300 $O/bench_2rank_gloo.log
  find $O -type f -size +12M -delete
  find $O -name "*kernel_trace.csv" -size +3M -delete
  du -sh $O
fi
