#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02g; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do
for v in default p0 w3; do
  if [ $v = default ]; then unset PINGS_HIP_LIB; else export PINGS_HIP_LIB=$R/pings_amd/lib/libpings_hip_$v.so; fi
  echo "== $v"; timeout -k 10 300 python scratch/sdf_index_ab.py 1000000 2>&1 | grep -E "sdf_forward|knn_search" | tee -a $O/ab_$v.log
done
done
