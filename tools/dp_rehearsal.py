#!/usr/bin/env python3
"""One-GPU rehearsal of the data-parallel step (MSAU_FORCE_DIST=1: world-size-1 RCCL all-reduces stay in the
sequence).  Prints ms/step for: no exchange, per-stage buckets behind the backward (default), all buckets after the
backward, one bucket after the backward -- the cost of the exchange machinery itself, without any wire time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")

import torch
import torch.distributed as dist

import bench
from msau_amd.model import MSAUWrapper, TrainEngine
from msau_amd.dp import GradSync

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
kw = dict(scale_space_num=4, res_depth=2, featRoot=8, final_act="softmax", num_blocks=3, dtype="bf16", seed=0)
x, label = bench.synthetic(16, 64, 336, 256, 5, 1234, dev)


def measure(tag, force, mode):
    os.environ["MSAU_FORCE_DIST"] = "1" if force else "0"
    m = MSAUWrapper(64, 5, kw).to(dev)
    eng = TrainEngine(m)
    if mode == "after":
        eng._fwd_bwd_orig = eng._fwd_bwd
        def fb(plan, x, l):
            eng.model  # noqa
            plan.forward(eng.model._flat, x, export=False)
            loss = plan.loss_grads(l)
            plan.backward(eng.flat_grad)
            eng._ar_started = False
            return loss
        eng._fwd_bwd = fb
    if mode == "one":
        eng.sync = GradSync(eng.flat_grad, None, None)
        def fb(plan, x, l):
            plan.forward(eng.model._flat, x, export=False)
            loss = plan.loss_grads(l)
            plan.backward(eng.flat_grad)
            eng._ar_started = False
            return loss
        eng._fwd_bwd = fb
    for _ in range(5):
        eng.step(x, label)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        eng.step(x, label)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{tag:48s} {1e3 * (t2 - t0) / n:.3f} ms/step (host enqueue {1e3 * (t1 - t0) / n:.3f})", flush=True)


for _ in range(3):
    measure("no exchange", False, "stage")
    measure("per-stage buckets behind backward (default)", True, "stage")
    measure("4 buckets after backward", True, "after")
    measure("1 bucket after backward", True, "one")
dist.destroy_process_group()
