#!/usr/bin/env python3
"""Micro-benchmark of the small-image convolutions of cfg 2 (levels 1-3, B=16): a chain of dependent msau_conv2d launches
(x -> y -> x ...), device time per launch by events, nothing else on the device.  These launches are latency chains
(weights -> LDS, input tile, epilogue operands, store), not bandwidth: this is the tool that prices a change to that chain.

    python tools/small_bench.py [reps]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msau_amd import _lib as L

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda")
s = torch.cuda.current_stream().cuda_stream
B = 16
lib = L.load()


def timed(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L.call("msau_spin", s, 3000)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def t(H, W, Cc):
    return (torch.randn(B, H, W, Cc, device=dev) * 0.5).to(torch.bfloat16)


rows_out = []
for (H, W, Cc, K) in ((168, 128, 16, 3), (84, 64, 32, 3), (84, 64, 32, 1), (42, 32, 64, 3), (42, 32, 64, 1), (336, 256, 8, 3), (336, 256, 8, 4), (336, 256, 8, 1), (336, 256, 16, 1)):
    x, y, m = t(H, W, Cc), t(H, W, Cc), t(H, W, Cc)
    kchunk = -(-K * K * Cc // 32) * 32
    rows = max(16, Cc)
    w = (torch.randn(rows * kchunk, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.zeros(rows, device=dev)

    def desc(xi, yo, flags, add=None, mask_b=None):
        d = L.ConvDesc()
        d.B, d.Hin, d.Win, d.Hout, d.Wout = B, H, W, H, W
        d.C1, d.C2, d.Cout = Cc, 0, Cc
        d.KH = d.KW = K
        d.dil, d.pad_t, d.pad_l, d.stride, d.ups = 1, (K - 1) // 2, (K - 1) // 2, 1, 1
        d.flags = flags
        d.x1, d.wpack, d.bias, d.y = xi.data_ptr(), w.data_ptr(), b.data_ptr(), yo.data_ptr()
        d.add = add.data_ptr() if add is not None else None
        d.mask_b = mask_b.data_ptr() if mask_b is not None else None
        return d

    for name, fl, kw in (("relu_in|relu_out", L.CONV_RELU_IN | L.CONV_RELU_OUT, {}),
                         ("mask_b", L.CONV_MASK_B, dict(mask_b=m)),
                         ("add|relu_out", L.CONV_ADD | L.CONV_RELU_OUT, dict(add=m))):
        d1, d2 = desc(x, y, fl, **kw), desc(y, x, fl, **kw)

        def chain():
            L.check(lib.msau_conv2d(s, L.BF16, d1), "conv")
            L.check(lib.msau_conv2d(s, L.BF16, d2), "conv")
        us = timed(chain) / 2
        mb = 2 * B * H * W * Cc * 2 / 1e6 * (1.5 if kw else 1.0)
        print(f"{H}x{W}x{Cc} k{K} {name:18s} {us:7.2f} us/launch   {mb / us / 1e3:6.2f} TB/s", flush=True)

# the floor: a dependent chain of the smallest kernel the library has
us = timed(lambda: L.call("msau_spin", s, 1))
print(f"msau_spin(1) chain: {us:.2f} us/launch")
