#!/bin/bash
# One GPU-box pass of round 2: usage (repo root on the box): bash tools/gpu_round.sh <tag> <part>
# part 1: tests + smoke + bench;  part 2: rocprofv3 kernel stats + PMC passes + 2-rank rehearsal
set -o pipefail
tag=${1:-x}; part=${2:-1}
R=$PWD
O=$R/gpurun_out/r03$tag
mkdir -p $O
export TMPDIR=/tmp
if [ "$part" = "1" ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
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
  for cfg in "1000000 131072" "1000000 16384" "200000 131072" "5000000 131072"; do
    set -- $cfg
    for ctr in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 150 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/sdf_${1}_${2}_$ctr -o p -- python3 $R/tools/sdf_pmc.py $1 $2 > $O/sdf_pmc.log 2>&1; echo "sdf pmc $1 $2 $ctr rc=$?"
    done
  done
  cd $R
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --single-device --no-sdf --no-cpu-baseline > $O/bench_2rank_gloo.log 2>&1; echo "2-rank rehearsal rc=$?"
  tail -c 300 $O/bench_2rank_gloo.log
  find $O -type f -size +12M -delete
  find $O -name "*kernel_trace.csv" -size +3M -delete
  du -sh $O
fi
