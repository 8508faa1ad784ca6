import json, sys, torch
sys.path.insert(0, '/root/repo')
import bench
print(json.dumps(bench.cglow_timing(torch.device('cuda:0'), cpu_steps=2)))
