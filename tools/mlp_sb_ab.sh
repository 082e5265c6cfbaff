#!/bin/bash
# A/B builds of the decoder backward with some of its scheduling barriers left to the compiler:
#   bash tools/mlp_sb_ab.sh build   (here)      ->  profiles/_build/libpings_hip_sb<mask>.so
#   bash tools/mlp_sb_ab.sh run     (GPU box)   ->  one dec_bench line per variant
R=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I$R/include -I$R/pings_amd/csrc -DPINGS_BUILDING_DLL"
MASKS=${MASKS:-"0 1 5 13 7"}
if [ "$1" = "build" ]; then
  mkdir -p $R/profiles/_build/sb
  for m in $MASKS; do
    ( /opt/rocm/bin/hipcc $FLAGS -DPINGS_MLP_SB=$m -c $R/pings_amd/csrc/mlp.hip -o $R/profiles/_build/sb/mlp_$m.o &&
      objs=$(ls $R/pings_amd/csrc/_obj/*.o | grep -v "/mlp.o") &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/profiles/_build/libpings_hip_sb$m.so $objs $R/profiles/_build/sb/mlp_$m.o && echo built $m ) &
  done
  wait
else
  echo "product (mask 15):"; python $R/tools/dec_bench.py 2>/dev/null | tail -1 | cut -c1-260
  for m in $MASKS; do
    echo "mask $m:"; PINGS_HIP_LIB=$R/profiles/_build/libpings_hip_sb$m.so python $R/tools/dec_bench.py 2>/dev/null | tail -1 | cut -c1-260
  done
fi
