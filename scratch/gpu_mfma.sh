#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02h; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_sdf.py tests/test_tracker.py tests/test_mesher.py -m gpu -q -x > $O/pytest_sdf.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_sdf.log
for v in mfma vector; do
  echo "== $v"; PINGS_SDF_FWD=$v timeout -k 10 300 python scratch/sdf_index_ab.py 1000000 2>&1 | grep -E "sdf_forward|knn_search|train" | tee -a $O/ab_$v.log
done
