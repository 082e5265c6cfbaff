#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02i; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_mlp.py tests/test_render.py tests/test_imgloss.py tests/test_sdf.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 300 python scratch/sdf_index_ab.py 1000000 2>&1 | grep -E "train" | tee $O/ab.log
