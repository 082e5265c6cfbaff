#!/bin/bash
# how much of the Metric-1 step is kernel time and how much the gaps between launches: tools/gap_m1.sh <outdir>
out=${1:-gpurun_out/gap}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o m1 -- python3 $GRAFT_REPO_ROOT/tools/raster_only.py ${2:-m1} 40 > $GRAFT_REPO_ROOT/$out/run.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $out/run.log | cut -c1-200
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print("kernel time total ms", tot / 1e6, "launches", calls)
for r in rows[:12]:
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
