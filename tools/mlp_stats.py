"""Shader-clock ticks per phase of the grouped decoder backward (diagnostic library, tools/build_stats_lib.sh):
PINGS_HIP_LIB=profiles/_build/libpings_hip_stats.so python tools/mlp_stats.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from pings_amd import _lib

torch.autograd.set_multithreading_enabled(False)
L = _lib.lib()
L.pings_debug_mlp_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 8)()
dev = torch.device("cuda")
r = bench.bench_decoder(dev, 5, 2)
L.pings_debug_mlp_stats(buf, 1)
cap = {}
orig = bench._timeit
bench._timeit = lambda fn, s, w, repeats=3: (cap.setdefault("n", 0), orig(fn, 1, 0, 1))[1]
r = bench.bench_decoder(dev, 1, 0)
L.pings_debug_mlp_stats(buf, 0)
names = ["prologue", "tile start (LDS views)", "A/B issued", "mask", "transposes + C issued", "D issued", "stage next tile",
         "tail"]
tot = float(sum(buf))
for n, v in zip(names, buf):
    print(f"{n:26s} {v:16d}  {100.0 * v / max(tot, 1):5.1f} %")
print(r)
clk = (C.c_ulonglong * 2)()
L.pings_debug_mlp_clock.argtypes = [C.POINTER(C.c_ulonglong)]
L.pings_debug_mlp_clock(clk)
print({"workgroup0_shader_cycles": clk[0], "realtime_ticks_100MHz": clk[1], "shader_clock_GHz": round(0.1 * clk[0] / max(clk[1], 1), 3)})
