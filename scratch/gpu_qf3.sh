#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02qa; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_sdf.py tests/test_tracker.py tests/test_mesher.py tests/test_map.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python scratch/qf_hostprof.py 2>&1 | grep -E "host issue"
timeout -k 10 300 python scratch/sdf_prof.py 1000000 2>&1 | grep -E "query_feature\+"
