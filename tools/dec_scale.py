"""Decoder leg at several point counts (fixed cost vs per-tile cost of the backward): python tools/dec_scale.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

torch.autograd.set_multithreading_enabled(False)
for n in (31_250, 62_500, 125_000, 250_000, 500_000):
    r = bench.bench_decoder(torch.device("cuda"), 20, 3, n_points=n)
    print(json.dumps({"neural_points": n, "fwd_ms": r["fwd_ms"], "bwd_ms": r["bwd_ms"], "bwd_frac": r["bwd_frac_of_fp32_mfma_peak"]}))
