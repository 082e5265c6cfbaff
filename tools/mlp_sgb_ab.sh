#!/bin/bash
# A/B builds of the decoder backward with scheduling-group chains (PINGS_MLP_SGB, csrc/mlp.hip):
#   bash tools/mlp_sgb_ab.sh build   (here)      ->  profiles/_build/libpings_hip_sgb<v>.so
#   bash tools/mlp_sgb_ab.sh run     (GPU box)   ->  one dec_bench line per variant
R=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I$R/include -I$R/pings_amd/csrc -DPINGS_BUILDING_DLL"
VARS=${VARS:-"1 2 3 4"}
if [ "$1" = "build" ]; then
  mkdir -p $R/profiles/_build/sgb
  for m in $VARS; do
    /opt/rocm/bin/hipcc $FLAGS $EXTRA -DPINGS_MLP_SGB=$m -c $R/pings_amd/csrc/mlp.hip -o $R/profiles/_build/sgb/mlp_$m.o &
  done
  wait
  objs=$(ls $R/pings_amd/csrc/_obj/*.o | grep -v "/mlp.o")
  for m in $VARS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/profiles/_build/libpings_hip_sgb$m.so $objs $R/profiles/_build/sgb/mlp_$m.o && echo built $m
  done
else
  echo "product:"; python $R/tools/dec_bench.py 2>/dev/null | tail -1 | cut -c1-200
  for m in $VARS; do
    echo "variant $m:"; PINGS_HIP_LIB=$R/profiles/_build/libpings_hip_sgb$m.so python $R/tools/dec_bench.py 2>/dev/null | tail -1 | cut -c1-200
    PINGS_HIP_LIB=$R/profiles/_build/libpings_hip_sgb$m.so timeout -k 10 200 python -m pytest $R/tests/test_mlp.py -x -q -m gpu 2>&1 | tail -1
  done
fi
