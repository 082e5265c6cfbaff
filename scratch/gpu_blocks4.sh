#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02f; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_sdf.py tests/test_tracker.py tests/test_mesher.py -m gpu -q -x > $O/pytest_sdf.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sdf.log
echo "== w4 (default)"; timeout -k 10 300 python scratch/sdf_index_ab.py 1000000 2>&1 | grep -v amdgpu.ids | tee $O/ab_w4.log
echo "== w3"; PINGS_HIP_LIB=$R/pings_amd/lib/libpings_hip_w3.so timeout -k 10 300 python scratch/sdf_index_ab.py 1000000 2>&1 | grep -v amdgpu.ids | tee $O/ab_w3.log
