#!/usr/bin/env python3
"""Micro-benchmark: msau_conv_pair (two chained 3x3 convs, one launch) against the two msau_conv2d launches it replaces,
forward flags and data-gradient flags, on the residual-block shapes of cfg 2 (B=16).  Serialised, device time by events.

    python tools/pair_bench.py [reps]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msau_amd import _lib as L

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
only = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # channel count to run (0: all)
dev = torch.device("cuda")
s = torch.cuda.current_stream().cuda_stream
B = 16


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


for (H, W, Cc) in ((336, 256, 8), (168, 128, 16)):
    if only and Cc != only:
        continue
    n = B * H * W * Cc
    t = lambda: (torch.randn(B, H, W, Cc, device=dev) * 0.5).to(torch.bfloat16)
    x, r1, out, g, gr1, gx = t(), t(), t(), t(), t(), t()
    kchunk = -(-9 * Cc // 32) * 32
    rows = max(16, Cc)
    w1 = (torch.randn(rows * kchunk, device=dev) * 0.05).to(torch.bfloat16)
    w2 = (torch.randn(rows * kchunk, device=dev) * 0.05).to(torch.bfloat16)
    for w_ in (w1, w2):                  # as msau_pack_params writes it: the k padding and the rows beyond C are zero
        w_.view(rows, kchunk)[:, 9 * Cc:] = 0
        w_.view(rows, kchunk)[Cc:] = 0
    b1 = torch.zeros(rows, device=dev)
    b2 = torch.zeros(rows, device=dev)

    def conv_desc(xi, wi, bi, yo, flags, add=None, mask_a=None, mask_b=None):
        d = L.ConvDesc()
        d.B, d.Hin, d.Win, d.Hout, d.Wout = B, H, W, H, W
        d.C1, d.C2, d.Cout = Cc, 0, Cc
        d.KH = d.KW = 3
        d.dil, d.pad_t, d.pad_l, d.stride, d.ups = 1, 1, 1, 1, 1
        d.flags = flags
        d.x1, d.wpack, d.bias, d.y = xi.data_ptr(), wi.data_ptr(), bi.data_ptr() if bi is not None else None, yo.data_ptr()
        d.add = add.data_ptr() if add is not None else None
        d.mask_a = mask_a.data_ptr() if mask_a is not None else None
        d.mask_b = mask_b.data_ptr() if mask_b is not None else None
        return d

    probe = L.ConvPairDesc()
    probe.B, probe.H, probe.W, probe.C = B, H, W, Cc
    probe.flags1, probe.flags2 = L.PAIR_MASK_MID, L.CONV_MASK_A | L.CONV_ADD
    probe.x = probe.add = g.data_ptr()
    nbits = int(L.load().msau_conv_pair_bits_bytes(L.BF16, C.byref(probe)))
    bits_mid = torch.zeros(nbits, dtype=torch.uint8, device=dev)     # ReLU masks as bit planes, as the plan runs it
    bits_a = torch.zeros_like(bits_mid)

    def pair_desc(fwd):
        p = L.ConvPairDesc()
        p.B, p.H, p.W, p.C = B, H, W, Cc
        if os.environ.get("MSAU_PAIR_BITS", "1") != "0":
            p.bits_mid, p.bits_a = bits_mid.data_ptr(), bits_a.data_ptr()
        if fwd:
            p.flags1, p.flags2 = L.PAIR_RELU_IN | L.PAIR_RELU_MID, L.CONV_ADD | L.CONV_RELU_OUT
            p.x, p.w1, p.b1, p.mid = x.data_ptr(), w1.data_ptr(), b1.data_ptr(), r1.data_ptr()
            p.w2, p.b2, p.add, p.y = w2.data_ptr(), b2.data_ptr(), x.data_ptr(), out.data_ptr()
        else:
            p.flags1, p.flags2 = L.PAIR_MASK_MID, L.CONV_MASK_A | L.CONV_ADD
            p.x, p.w1, p.b1, p.mask_mid, p.mid = g.data_ptr(), w2.data_ptr(), None, r1.data_ptr(), gr1.data_ptr()
            p.w2, p.b2, p.add, p.mask_a, p.y = w1.data_ptr(), None, g.data_ptr(), x.data_ptr(), gx.data_ptr()
        return p

    f1 = conv_desc(x, w1, b1, r1, L.CONV_RELU_IN | L.CONV_RELU_OUT)
    f2 = conv_desc(r1, w2, b2, out, L.CONV_ADD | L.CONV_RELU_OUT, add=x)
    d2 = conv_desc(g, w2, None, gr1, L.CONV_MASK_B, mask_b=r1)
    d1 = conv_desc(gr1, w1, None, gx, L.CONV_MASK_A | L.CONV_ADD, add=g, mask_a=x)
    pf, pb = pair_desc(True), pair_desc(False)
    lib = L.load()
    assert lib.msau_conv_pair_applicable(L.BF16, C.byref(pf))

    def two(a, b):
        L.check(lib.msau_conv2d(s, L.BF16, C.byref(a)))
        L.check(lib.msau_conv2d(s, L.BF16, C.byref(b)))

    # the fused launches against the two launches they replace (same packed weights, same inputs)
    def rel(a_, b_):
        return float((a_.float() - b_.float()).abs().max() / b_.float().abs().max().clamp_min(1e-9))
    two(f1, f2)
    torch.cuda.synchronize()
    r1_ref, out_ref = r1.clone(), out.clone()
    r1.zero_(); out.zero_()
    L.check(lib.msau_conv_pair(s, L.BF16, C.byref(pf)))
    torch.cuda.synchronize()
    print(f"  check fwd: mid rel {rel(r1, r1_ref):.2e} (equal {bool(torch.equal(r1, r1_ref))}), out rel {rel(out, out_ref):.2e} (equal {bool(torch.equal(out, out_ref))})")
    r1.copy_(r1_ref)
    two(d2, d1)
    torch.cuda.synchronize()
    gr1_ref, gx_ref = gr1.clone(), gx.clone()
    gr1.zero_(); gx.zero_()
    L.check(lib.msau_conv_pair(s, L.BF16, C.byref(pb)))
    torch.cuda.synchronize()
    print(f"  check bwd: gmid rel {rel(gr1, gr1_ref):.2e} (equal {bool(torch.equal(gr1, gr1_ref))}), gx rel {rel(gx, gx_ref):.2e} (equal {bool(torch.equal(gx, gx_ref))})")
    res = {}
    res["fwd 2 launches"] = timed(lambda: two(f1, f2))
    res["fwd pair"] = timed(lambda: L.check(lib.msau_conv_pair(s, L.BF16, C.byref(pf))))
    res["bwd 2 launches"] = timed(lambda: two(d2, d1))
    res["bwd pair"] = timed(lambda: L.check(lib.msau_conv_pair(s, L.BF16, C.byref(pb))))
    if Cc == 8:
        # riders of the 8-channel data-gradient launch: LRN backward in the epilogue, the first conv's weight gradient
        la, lda = t(), t()
        for tag, fl in (("bwd pair +lrn", L.PAIR_LRN_BWD), ("bwd pair +wg1", L.PAIR_WGRAD1), ("bwd pair +lrn+wg1", L.PAIR_LRN_BWD | L.PAIR_WGRAD1)):
            q = pair_desc(False)
            q.flags1 |= fl
            q.lrn_a, q.lrn_da, q.lrn_alpha_over_n, q.lrn_beta, q.lrn_k = la.data_ptr(), lda.data_ptr(), 1e-4 / 8, 0.75, 1.0
            slabs = torch.zeros(4096 * 640, device=dev)
            q.wg1_x, q.wg1_slabs = x.data_ptr(), slabs.data_ptr()
            q.wg1_nslabs = int(lib.msau_conv_pair_wgrad_slabs(L.BF16, C.byref(q))) if fl & L.PAIR_WGRAD1 else 0
            if lib.msau_conv_pair_applicable(L.BF16, C.byref(q)):
                res[tag] = timed(lambda: L.check(lib.msau_conv_pair(s, L.BF16, C.byref(q))))
    mb = n * 2 / 1e6
    print(f"C={Cc} {H}x{W}: tensor {mb:.1f} MB | " + " | ".join(f"{k} {v:.1f} us" for k, v in res.items()))
