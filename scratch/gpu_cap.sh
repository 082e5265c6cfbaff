#!/bin/bash
for cap in 0 768 512 384 256; do
  echo "== cap $cap"; PINGS_KNN_GRID_CAP=$cap timeout -k 10 300 python scratch/sdf_index_ab.py 1000000 2>&1 | grep -E "B=16384 sdf_forward"
done
