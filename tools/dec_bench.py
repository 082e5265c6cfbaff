"""The five spawn decoders alone (bench.py decoder leg): python tools/dec_bench.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

torch.autograd.set_multithreading_enabled(False)
print(json.dumps(bench.bench_decoder(torch.device("cuda"), 20, 3)))
