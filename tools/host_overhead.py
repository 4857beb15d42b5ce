#!/usr/bin/env python3
"""Host enqueue time vs device time of one eager train step (are we launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msau_amd import MSAUWrapper, TrainEngine
import bench
kw = dict(scale_space_num=4, res_depth=2, featRoot=8, final_act="softmax", num_blocks=3, dtype="bf16", seed=0)
m = MSAUWrapper(64, 5, kw).cuda()
eng = TrainEngine(m)
x, label = bench.synthetic(16, 64, 336, 256, 5, 1234, torch.device("cuda"))
for _ in range(3):
    eng.step(x, label)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    eng.step(x, label)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/n:.3f} ms/step, total {1e3*(t2-t0)/n:.3f} ms/step")
