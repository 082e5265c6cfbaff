#!/bin/bash
# diagnostic build of the library with blend-loop counters (never shipped)
set -e
cd /root/repo
python -m pings_amd.build > /dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Iinclude -Ipings_amd/csrc -DPINGS_BUILDING_DLL"
hipcc $F -DPINGS_BLEND_STATS -c pings_amd/csrc/raster_fwd.hip -o scratch/lib_stats/raster_fwd.o
objs=$(ls pings_amd/csrc/_obj/*.o | grep -v raster_fwd.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib_stats/libpings_hip.so $objs scratch/lib_stats/raster_fwd.o
echo built
