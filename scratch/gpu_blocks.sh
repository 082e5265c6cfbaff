#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02c; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_sdf.py tests/test_tracker.py tests/test_map.py tests/test_mesher.py tests/test_map_io.py -m gpu -q -x > $O/pytest_sdf.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_sdf.log
for idx in table blocks; do
  PINGS_KNN_INDEX=$idx timeout -k 10 300 python scratch/sdf_index_ab.py 1000000 > $O/ab_$idx.log 2>&1; echo "ab $idx rc=$?"; cat $O/ab_$idx.log
done
PINGS_KNN_INDEX=blocks timeout -k 10 300 python scratch/sdf_index_ab.py 5000000 > $O/ab_blocks_5m.log 2>&1; echo "rc=$?"; cat $O/ab_blocks_5m.log
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/sdfb_1000000_131072_$ctr -o p -- python3 $R/scratch/sdf_pmc.py 1000000 131072 > $O/sdf_pmc.log 2>&1; echo "sdf pmc $ctr rc=$?"
done
find $O -name "*kernel_trace.csv" -size +3M -delete
