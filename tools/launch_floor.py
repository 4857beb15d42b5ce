#!/usr/bin/env python3
"""Launch overheads on the box: (1) GPU-side cadence of back-to-back tiny kernels, (2) host enqueue cost of a tiny
kernel, (3) host enqueue cost and GPU-side cadence of the forward-only launch sequence at batch 1 (118 dependent
kernels with almost no work each): what a launch-bound chain costs per kernel on each side."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from msau_amd import _lib
from msau_amd.model import MSAUWrapper

_lib.load()
x = torch.zeros(1024, device="cuda")
s = torch.cuda.current_stream().cuda_stream


def tiny(n):
    for _ in range(n):
        _lib.call("msau_fill_zero", s, x.data_ptr(), 4096)


tiny(100)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
_lib.call("msau_spin", s, 30000)          # park the stream so the host runs ahead
e0.record(); tiny(2000); e1.record(); torch.cuda.synchronize()
print("tiny kernel, GPU cadence      : %.2f us/launch" % (e0.elapsed_time(e1) * 1000 / 2000))
t = time.perf_counter(); tiny(2000); t1 = time.perf_counter(); torch.cuda.synchronize()
print("tiny kernel, host enqueue     : %.2f us/launch (through ctypes)" % ((t1 - t) * 1e6 / 2000))

for (H, W) in ((70, 128), (336, 256)):
    m = MSAUWrapper(64, 5, dict(featRoot=8, scale_space_num=4, res_depth=2, filter_size=3, pool_size=2, dtype="bf16", seed=0)).cuda().eval()
    plan = m._plan_for_shape(1, H, W, torch.device("cuda", 0), False)
    n = plan._fwd_seq[1]
    for _ in range(3):
        plan._run_seq(plan._fwd_seq, s)
    torch.cuda.synchronize()
    reps = 20
    _lib.call("msau_spin", s, 60000)
    t = time.perf_counter()
    e0.record()
    for _ in range(reps):
        plan._run_seq(plan._fwd_seq, s)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("forward sweep B=1 %dx%d (%d launches): host enqueue %.2f us/launch, GPU cadence %.2f us/launch"
          % (H, W, n, (t1 - t) * 1e6 / (reps * n), e0.elapsed_time(e1) * 1000 / (reps * n)))
