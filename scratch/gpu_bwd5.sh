#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02m; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_sdf.py tests/test_tracker.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in mfma vector mfma vector; do
  echo "== bwd $v"; PINGS_SDF_BWD=$v timeout -k 10 300 python scratch/sdf_prof.py 1000000 2>&1 | grep -E "fused"
done
