#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02s; mkdir -p $O
for m in buckets table; do
PINGS_ROWS_GATHER=$m timeout -k 10 500 python -m pytest tests/test_sdf.py -m gpu -q -x -k "scatter or backward or gradient or double" > $O/pytest_$m.log 2>&1; echo "pytest $m rc=$?"; tail -2 $O/pytest_$m.log
done
for m in table buckets table buckets; do
  echo "== $m"; PINGS_ROWS_GATHER=$m timeout -k 10 300 python scratch/sdf_prof.py 1000000 2>&1 | grep -E "fused|query_feature\+"
done
