"""SDF forward / query_feature / knn search kernel times for the current PINGS_KNN_INDEX: python sdf_index_ab.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pings_amd import neural_points as hnp
dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = bench._lib_handle()
npm, dec = bench.sdf_synth_map(N, dev)
print("index", hnp.KNN_INDEX, "N", N, end=" ")
if hnp.KNN_INDEX == "blocks":
    t0 = time.perf_counter(); bi = hnp._block_index(npm); torch.cuda.synchronize()
    print("build wall ms %.2f" % ((time.perf_counter() - t0) * 1e3), "status", bi.status.cpu().tolist(), end=" ")
    npm._pings_blocks = None
    print("build kernels", bench._prof_run(L, lambda: (setattr(npm, "_pings_blocks", None), hnp._block_index(npm)), 3))
else:
    print()
for B in (16384, 131072):
    x = bench.sdf_queries(npm, B, dev)
    fwd = lambda: hnp.sdf_fused(npm, dec, x, use_only_measured_points=False)
    srch = lambda: hnp.radius_neighborhood_topk(npm, x, query_locally=True)
    fwdg = lambda: hnp.sdf_fused(npm, dec, x, need_grad=True, use_only_measured_points=False)
    for name, fn, stage in (("sdf_forward", fwd, "sdf_forward"), ("sdf_forward+grad", fwdg, "sdf_forward"), ("knn_search", srch, "knn_search")):
        tw = bench._timeit(fn, 30, 5)
        tk = bench._prof_run(L, fn, 20)[stage]
        print(f"  B={B} {name}: kernel {tk:.4f} ms  wall {tw*1e3:.4f} ms  {B/tk/1e3:.1f} Msamples/s (kernel)")
    r = bench.sdf_train_rates(npm, dec, x, 30, 5)
    print("  train rates", r)
