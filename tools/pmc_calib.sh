#!/bin/bash
# PMC calibration + a fresh default bench line (repo root on the box): bash tools/pmc_calib.sh <tag>
set -o pipefail
tag=${1:-x}
R=$PWD
O=$R/gpurun_out/r03$tag
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 120 $R/profiles/_build/pmc_calib > $O/calib_stdout.txt 2>&1; echo "calib plain rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_f -o f -- $R/profiles/_build/pmc_calib > $O/calib_f.log 2>&1; echo "calib fetch rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_w -o w -- $R/profiles/_build/pmc_calib > $O/calib_w.log 2>&1; echo "calib write rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/calib_s -o s -- $R/profiles/_build/pmc_calib > $O/calib_s.log 2>&1; echo "calib stats rc=$?"
cd $R
python profiles/pmc_calib_summary.py $O/calib_f/f_counter_collection.csv $O/calib_w/w_counter_collection.csv $O/calib_stdout.txt $O/pmc_calibration.json
timeout -k 10 500 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"
tail -c 300 $O/bench.log
find $O -name "*kernel_trace.csv" -size +3M -delete
