#!/usr/bin/env python3
"""Print loss / grad-norm per step for the bench configuration (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msau_amd import MSAUWrapper, TrainEngine
import bench

B = int(os.environ.get("B", "16")); dtype = os.environ.get("DTYPE", "bf16"); steps = int(os.environ.get("STEPS", "14"))
kw = dict(scale_space_num=4, res_depth=2, featRoot=8, final_act="softmax", num_blocks=3, dtype=dtype, seed=0)
m = MSAUWrapper(64, 5, kw).cuda()
eng = TrainEngine(m, use_graph=os.environ.get("GRAPH", "0") == "1")
x, label = bench.synthetic(B, 64, 336, 256, 5, 1234, torch.device("cuda"))
for i in range(steps):
    loss = eng.step(x, label)
    print(i, float(loss), float(eng.grad_norm), flush=True)
