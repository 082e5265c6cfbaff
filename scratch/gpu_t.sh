#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02n; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_sdf.py tests/test_raster.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
