"""One SDF forward configuration under rocprofv3 --pmc: python sdf_pmc.py <N_np> <B> (20 launches of pings_sdf_forward)."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch, bench
from pings_amd import neural_points as hnp
N, B = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda")
npm, dec = bench.sdf_synth_map(N, dev)
x = bench.sdf_queries(npm, B, dev)
for _ in range(20):
    hnp.sdf_fused(npm, dec, x, use_only_measured_points=False)
torch.cuda.synchronize()
print("done", N, B)
