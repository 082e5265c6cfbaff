#!/bin/bash
R=$PWD
for v in default s4 default s4; do
  if [ $v = default ]; then unset PINGS_HIP_LIB; else export PINGS_HIP_LIB=$R/pings_amd/lib/libpings_hip_$v.so; fi
  echo "== $v"; timeout -k 10 300 python scratch/sdf_index_ab.py 1000000 2>&1 | grep -E "sdf_forward"
done
