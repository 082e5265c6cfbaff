import sys, time, torch, ctypes as C
sys.path.insert(0,'/root/repo')
import bench
from pings_amd import neural_points as hnp, _lib
dev=torch.device('cuda')
L=_lib.lib(); L.pings_prof_enable.argtypes=[C.c_int]; L.pings_prof_report.argtypes=[C.c_char_p,C.c_size_t]
npm,dec=bench.sdf_synth_map(1_000_000,dev)
for B in (16384,131072):
    x=bench.sdf_queries(npm,B,dev)
    def t(f,n=20):
        for _ in range(3): f()
        torch.cuda.synchronize(); L.pings_prof_enable(1); t0=time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/n*1e3
        L.pings_prof_enable(0); buf=C.create_string_buffer(4096); L.pings_prof_report(buf,4096)
        return dt, {l.split()[0]: float(l.split()[2])/int(l.split()[1]) for l in buf.value.decode().strip().splitlines()}
    for compact in (True, False):
        hnp.USE_COMPACT_TABLE=compact
        print("B",B,"compact",compact,"search",t(lambda: hnp.radius_neighborhood_topk(npm,x,query_locally=True)), "fused",t(lambda: hnp.sdf_fused(npm,dec,x,use_only_measured_points=False)), "fused+grad", t(lambda: hnp.sdf_fused(npm,dec,x,need_grad=True,use_only_measured_points=False)))
