#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02q; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_sdf.py tests/test_tracker.py tests/test_mesher.py tests/test_map.py tests/test_map_io.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
