#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02e; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_sdf.py tests/test_tracker.py tests/test_map.py tests/test_mesher.py tests/test_map_io.py -m gpu -q -x > $O/pytest_sdf.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_sdf.log
PINGS_KNN_INDEX=blocks timeout -k 10 300 python scratch/sdf_index_ab.py 1000000 2>&1 | grep -v amdgpu.ids | tee $O/ab_blocks.log
timeout -k 10 300 python scratch/sdf_hostprof.py 2>&1 | grep -v amdgpu.ids | tee $O/hostprof.log
timeout -k 10 300 python scratch/sdf_phase.py 2>&1 | grep -v amdgpu.ids | tee $O/phase.log
