import sys, json, torch
sys.path.insert(0, '/root/repo')
import bench
print(json.dumps(bench.bench_decoder(torch.device('cuda:0'), 20, 3)))
