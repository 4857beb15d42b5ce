#!/usr/bin/env python3
"""Micro-benchmark of the memory-bound normalisation / pooling / boundary passes north_star names, at the cfg-2 shapes
(B = 16): LocalResponseNorm forward / backward (layers.py:145,161-162) and MaxPool2d(2,2) forward / backward
(model/model.py:158-160) at 8 and 16 channels, the fp32 NCHW -> NHWC conversion of the input.  One launch of each per
iteration, in a fixed order (tools/make_traffic.py keys its PMC passes on that order).

    python tools/norm_bench.py [iters]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msau_amd import _lib as L

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda")
s = torch.cuda.current_stream().cuda_stream
B = 16
# name -> callable; ORDER matters (see above)
ops = []
for (H, W, Cc) in ((336, 256, 8), (168, 128, 16)):
    npix = B * H * W
    t = lambda *shape: (torch.randn(*shape, device=dev) * 0.5).to(torch.bfloat16)
    a, dy, out = t(B, H, W, Cc), t(B, H, W, Cc), t(B, H, W, Cc)
    py, pidx = t(B, (H + 1) // 2, (W + 1) // 2, Cc), torch.zeros(B, (H + 1) // 2, (W + 1) // 2, Cc, dtype=torch.uint8, device=dev)
    ops.append((f"lrn_fwd<bf16,C{Cc}>", 2 * npix * Cc * 2,
                lambda a=a, out=out, npix=npix, Cc=Cc: L.call("msau_lrn_fwd", s, L.BF16, a.data_ptr(), out.data_ptr(), npix, Cc, Cc, Cc, 1e-4, 0.75, 1.0)))
    ops.append((f"lrn_bwd<bf16,C{Cc}>", 3 * npix * Cc * 2,
                lambda a=a, dy=dy, out=out, npix=npix, Cc=Cc: L.call("msau_lrn_bwd", s, L.BF16, a.data_ptr(), dy.data_ptr(), out.data_ptr(), npix, Cc, Cc, Cc, 1e-4, 0.75, 1.0)))
    ops.append((f"pool_fwd<bf16,C{Cc}>", npix * Cc * 2 + (npix // 4) * Cc * 3,
                lambda a=a, py=py, pidx=pidx, H=H, W=W, Cc=Cc: L.call("msau_maxpool2x2_fwd", s, L.BF16, a.data_ptr(), py.data_ptr(), pidx.data_ptr(), B, H, W, Cc)))
    ops.append((f"pool_bwd<bf16,C{Cc}>", npix * Cc * 2 + (npix // 4) * Cc * 3,
                lambda py=py, pidx=pidx, out=out, H=H, W=W, Cc=Cc: L.call("msau_maxpool2x2_bwd", s, L.BF16, py.data_ptr(), pidx.data_ptr(), out.data_ptr(), None, B, H, W, Cc, 0)))
xin = torch.randn(B, 64, 336, 256, device=dev)
xout = torch.empty(B, 336, 256, 64, dtype=torch.bfloat16, device=dev)
ops.append(("msau_nchw_to_nhwc", B * 64 * 336 * 256 * 6,
            lambda: L.call("msau_nchw_to_nhwc", s, L.BF16, xin.data_ptr(), xout.data_ptr(), B, 64, 64, 336, 256)))

if __name__ == "__main__":
    for name, nbytes, fn in ops:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        print(f"{name:22s} {nbytes / 1e6:7.1f} MB  {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)
