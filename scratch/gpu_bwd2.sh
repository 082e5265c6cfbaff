#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02k; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o s -- python3 $R/scratch/sdf_step.py 1000000 > $O/p.log 2>&1; echo "rc=$?"
cut -d, -f1-4 $O/p/s_kernel_stats.csv | grep -i "sdf\|rows\|gather_sum" | cut -c1-160
