#!/bin/bash
# Diagnostic build of the library with the blend-loop counters compiled in (never the product .so):
#   bash tools/build_stats_lib.sh  ->  profiles/_build/libpings_hip_stats.so   (use with PINGS_HIP_LIB=...)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/profiles/_build/stats_obj; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I$R/include -I$R/pings_amd/csrc -DPINGS_BUILDING_DLL -DPINGS_BWD_STATS -DPINGS_MLP_STATS"
for f in $R/pings_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  # reuse a product object only if it is newer than its source and than every header (a stale object would be
  # linked into a library that passes the ABI-version check only by luck)
  o=$R/pings_amd/csrc/_obj/$b.o
  stale=0
  for d in $f $R/pings_amd/csrc/*.hpp $R/include/*.h; do [ -f $o ] && [ $o -nt $d ] || stale=1; done
  if [ "$b" = "raster_bwd" ] || [ "$b" = "mlp" ] || [ $stale = 1 ]; then
    /opt/rocm/bin/hipcc $FLAGS -c $f -o $O/$b.o &
  else
    cp $R/pings_amd/csrc/_obj/$b.o $O/$b.o
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/profiles/_build/libpings_hip_stats.so $O/*.o
echo built $R/profiles/_build/libpings_hip_stats.so
