#!/usr/bin/env python3
"""Host enqueue time of one training step against its device time: sync, step() (returns when everything is enqueued),
sync.  If the first number is close to the second the step is host-bound.   python tools/host_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msau_amd import MSAUWrapper, TrainEngine
from oracle import msau_oracle as O

B, Cin, H, W, ncls = 16, 64, 336, 256, 5
m = MSAUWrapper(Cin, ncls, dict(scale_space_num=4, res_depth=2, featRoot=8, filter_size=3, pool_size=2, final_act="softmax", num_blocks=3, dtype="bf16", seed=0)).cuda()
eng = TrainEngine(m)
x, label = O.synthetic_batch(B, Cin, H, W, ncls, seed=1)
x, label = x.cuda(), label.cuda()
for _ in range(10):
    eng.step(x, label)
torch.cuda.synchronize()
host, tot = [], []
for _ in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(x, label)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0); tot.append(t2 - t0)
host.sort(); tot.sort()
print(f"host enqueue median {host[len(host)//2]*1e3:.3f} ms, step (from idle) median {tot[len(tot)//2]*1e3:.3f} ms")
t0 = time.perf_counter()
for _ in range(100):
    eng.step(x, label)
torch.cuda.synchronize()
print(f"back-to-back {1e3*(time.perf_counter()-t0)/100:.3f} ms/step")
