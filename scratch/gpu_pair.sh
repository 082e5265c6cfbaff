#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02u; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_sdf.py tests/test_tracker.py tests/test_mesher.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2; do timeout -k 10 300 python scratch/sdf_index_ab.py 1000000 2>&1 | grep -E "sdf_forward"; done
