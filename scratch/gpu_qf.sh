#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02t; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o s -- python3 $R/scratch/qf_step_big.py 131072 > $O/p.log 2>&1; echo "rc=$?"; grep "wall ms" $O/p.log
cd $R
python - <<PY
import csv,re
rows=list(csv.DictReader(open('gpurun_out/r02t/p/s_kernel_stats.csv')))
for r in rows[:22]:
    n=r['Name']; m=re.search(r'(\w+_kernel\w*|fillBuffer\w*|copyBuffer)',n)
    print(f"{(m.group(1) if m else n[:50]):50s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
