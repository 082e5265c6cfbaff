#!/bin/bash
R=$PWD
for cfg in "" "PINGS_SDF_GRAD_BLOCKS=512" "PINGS_SDF_GRAD_BLOCKS=256" "PINGS_SDF_GRAD_SKIPROWS=1" "PINGS_SDF_GRAD_BLOCKS=512 PINGS_SDF_GRAD_SKIPROWS=1"; do
  echo "== $cfg"; env $cfg timeout -k 10 300 python scratch/sdf_prof.py 1000000 2>&1 | grep -E "fused" 
done
