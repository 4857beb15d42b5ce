#!/usr/bin/env python3
"""Micro-benchmark of single conv / wgrad launches at the BASELINE cfg-2 layer shapes (GPU only).

    python tools/kbench.py [--dtype bf16] [--batch 16] [--iters 50]
Prints per-shape: launch time (HIP events, back-to-back launches), algorithmic GB/s and TFLOP/s.
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msau_amd import _lib as L
from msau_amd.plan import Act, ConvOp, Plan

SHAPES = [  # name, H, W, Cin (x1), Cin2, Cout, k, dil, kind
    ("L0 8->8 3x3", 336, 256, 8, 0, 8, 3, 1, "conv"),
    ("L0 64->8 3x3 (first)", 336, 256, 64, 0, 8, 3, 1, "conv"),
    ("L0 8+8->8 3x3 (merge)", 336, 256, 8, 8, 8, 3, 1, "conv"),
    ("L0 8+8->8 1x1 (couple)", 336, 256, 8, 8, 8, 1, 1, "conv"),
    ("L0 8->5 4x4 (end)", 336, 256, 8, 0, 5, 4, 1, "conv"),
    ("L1 16->16 3x3", 168, 128, 16, 0, 16, 3, 1, "conv"),
    ("L1 8->16 3x3 d2", 168, 128, 8, 0, 16, 3, 2, "conv"),
    ("L2 32->32 3x3", 84, 64, 32, 0, 32, 3, 1, "conv"),
    ("L3 64->64 3x3", 42, 32, 64, 0, 64, 3, 1, "conv"),
    ("L3 32->64 3x3 d8", 42, 32, 32, 0, 64, 3, 8, "conv"),
    ("L1->L0 deconv 16->8", 168, 128, 16, 0, 8, 3, 1, "deconv"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dt = L.BF16 if args.dtype == "bf16" else L.F32
    dev = torch.device("cuda:0")
    for name, H, W, c1, c2, co, k, dil, kind in SHAPES:
        if args.only and args.only not in name:
            continue
        cin = c1 + c2
        wshape = (co, cin, k, k) if kind == "conv" else (c1, co, k, k)
        poff = {"w": 0, "b": -(-int(torch.tensor(wshape).prod()) // 4) * 4}
        pshape = {"w": wshape, "b": (co,)}
        flat = torch.randn(poff["b"] + 64, device=dev) * 0.1
        ops = {}

        def build(plan):
            x = plan.x_in
            x2 = Act(plan, "x2", H, W, c2) if c2 else None
            if kind == "conv":
                y = Act(plan, "y", H, W, co)
            else:
                y = Act(plan, "y", 2 * H, 2 * W, co)
            ops["op"] = ConvOp(plan, "c", x, x2, "w", "b", y, k, dil=dil, kind=kind)
            plan.logits = y
        plan = Plan(dict(channels=c1, input_grad=True), args.batch, H, W, dt, dev, poff, pshape, training=True, builder=build)
        plan.pack(flat)
        for a in plan.acts:
            a.data.normal_()
            if a.grad is not None:
                a.grad.normal_()
        op = ops["op"]
        s = torch.cuda.current_stream().cuda_stream

        def timeit(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.iters * 1e3

        t_f = timeit(lambda: L.call("msau_conv2d", s, dt, C.byref(op.fdesc)))
        line = f"{name:26s} fwd {t_f:7.1f} us {op.fbytes / t_f / 1e3:7.0f} GB/s {op.flops / t_f / 1e6:6.1f} TF"
        if op.ddesc[0] is not None:
            t_d = timeit(lambda: L.call("msau_conv2d", s, dt, C.byref(op.ddesc[0])))
            line += f" | dgrad {t_d:7.1f} us {op.dmeta[0][1] / t_d / 1e3:7.0f} GB/s"
        t_w = timeit(lambda: L.call("msau_conv2d_wgrad", s, dt, C.byref(op.wdesc)))
        line += f" | wgrad {t_w:7.1f} us {op.wbytes / t_w / 1e3:7.0f} GB/s (nslabs {op.wdesc.nslabs})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
