"""GPU: every launch of a bf16 training plan checked ELEMENT BY ELEMENT against float64 arithmetic on the launch's OWN stored
operands (round-4 verdict, item 2a).

The network-level parity tests compare tensors in norms, because a ReLU that flips within bf16 rounding legitimately moves single
elements downstream.  A norm cannot see one wrong pixel per image or a last-ulp error, though.  Here nothing propagates: the
reference of every op is computed in float64 from the inputs that op actually read (the stored NHWC tensors of the plan, the
bf16-rounded weights of the packed image), so its result must be the CORRECTLY ROUNDED one up to the order of the fp32 sums:

    |stored - exact| <= 1/2 ulp_bf16 + c * sum |terms|        (c: the fp32 accumulation slack, 3e-5 for K <= 1200 terms)

for every element of every forward output (row-streaming pairs / single convs incl. the LRN second output and the 4x4 end conv,
tile pairs, lean and generic instances, transposed convs), exact equality for the max pool, and the same bound with one half-ulp
per accumulated contribution for every stored gradient tensor (two-output data gradients, strided / zero-stuffed instances,
residual adds, ReLU masks from bit planes, LRN and pool backward, the loss gradient).  Weight gradients (fp32) are checked against
float64 sums with a bound relative to sum |x||g|.  The riders of the 8-channel pair's data-gradient launch (LRN backward, first
conv's weight gradient) are checked against the same step with the riders off.  Reference semantics: model/model.py:37-50,129-164,
224-259, model/layers/layers.py:82-102,152-164,249-250 (through oracle/msau_oracle.py's formulae)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from msau_amd import _lib as L
from msau_amd.model import MSAUWrapper, TrainEngine
from msau_amd.plan import AttnCoreOp, ConvOp, LrnOp, PoolOp
from oracle import msau_oracle as O

pytestmark = pytest.mark.gpu
ACC = 3e-5                 # fp32 accumulation slack, relative to the sum of |terms| of an output element


def nchw(t, C, samples=None):
    if samples is not None:
        t = t[list(samples)]
    return t[..., :C].permute(0, 3, 1, 2).double().cpu().contiguous()


def ulp_bf16(v):
    a = v.abs().clamp_min(2.0 ** -120)
    return torch.pow(2.0, torch.floor(torch.log2(a)) - 7)


def assert_rounded(got, ref, slack, what, half_ulps=1, mag=None):
    """`mag`: the largest magnitude an intermediate (stored, rounded) partial sum can have -- contributions that cancel are each
    rounded at THEIR size, not at the size of what is left"""
    big = torch.maximum(got.abs(), ref.abs())
    if mag is not None:
        big = torch.maximum(big, mag)
    tol = half_ulps * 0.5 * ulp_bf16(big) * (1 + 1e-6) + slack + 1e-30
    d = (got - ref).abs()
    bad = d > tol
    if bad.any():
        i = int(torch.argmax((d / tol).flatten()))
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.numel()} elements beyond the rounding bound; worst: got "
                             f"{float(got.flatten()[i])!r} exact {float(ref.flatten()[i])!r} tol {float(tol.flatten()[i])!r}")


def conv_ref(op, sd, inp, w=None, b=None):
    """the conv / transposed conv of `op` applied to `inp` (NCHW float64, concatenated sources) with the given weight"""
    w = sd[op.wname] if w is None else w
    b = sd[op.bname] if b is None else b
    if op.kind == "conv":
        k, d = op.k, op.dil
        pb, pr = (k - 1) * d - op.pad_t, (k - 1) * d - op.pad_l
        return F.conv2d(F.pad(inp, (op.pad_l, pr, op.pad_t, pb)), w, b, dilation=d)
    return O.deconv(inp, w, b, (op.out.H, op.out.W))


def lrn_bwd_ref(a, dy):
    """-> (da, sum of |terms|): da_c = dy_c d_c^-b - a_c (2 alpha b / n) sum_{c' : c in window(c')} dy_c' a_c' d_c'^(-b-1); the kernels
    take the window sums as differences of fp32 prefix sums over ALL channels, so the terms of the bound are those of the whole sum"""
    n, alpha, beta = a.shape[1], 1e-4, 0.75
    ar = a.clone().requires_grad_(True)
    O.lrn(ar, n).backward(dy)
    # d = |a / y|^(1/beta)   (|.| BEFORE the clamp: a negative y used to be clamped to 1e-300, which made d huge and the bound of
    # every element with a < 0 -- half of an LRN input -- collapse to the bare half ulp; the bench-shape case then tripped over two
    # exact ties among 1.4 M elements)
    d = (a.abs() / O.lrn(a, n).abs().clamp_min(1e-300)).where(a != 0, torch.ones_like(a)) ** (1.0 / beta)
    t1 = dy.abs() * d ** -beta
    t2 = a.abs() * (2 * alpha * beta / n) * (dy.abs() * a.abs() * d ** (-beta - 1)).sum(1, keepdim=True)
    return ar.grad, t1 + t2


class Checker:
    def __init__(self, m, eng, plan, sd0, labels, max_h, min_h=0, samples=None):
        """`samples`: check the forward outputs and stored gradients of these batch elements only (every op is per sample;
        the float64 references of all 16 bench-size images would take minutes) -- weight gradients always sum the whole batch.
        `min_h` < H <= `max_h`: the levels whose tensors are checked."""
        self.m, self.eng, self.plan, self.labels, self.max_h, self.min_h = m, eng, plan, labels, max_h, min_h
        self.samples = None if samples is None else sorted(set(samples))
        self.nb = plan.B if samples is None else len(self.samples)
        # the packed images hold bf16(fp32 master weight); biases stay fp32
        self.sd = {k: (v.bfloat16().double() if k.endswith("weight") else v.double()) for k, v in sd0.items()}
        self.nchecked = 0

    def small(self, act):
        return self.min_h < act.H <= self.max_h

    def n(self, t, C):
        return nchw(t, C, self.samples)

    def source(self, op, whole_batch=False):
        n = nchw if whole_batch else self.n
        xs = [n(op.x1.data, op.x1.C)] + ([n(op.x2.data, op.x2.C)] if op.x2 is not None else [])
        inp = torch.cat(xs, 1)
        return torch.relu(inp) if op.relu_in else inp

    # ---- forward ------------------------------------------------------------------------------------------------
    def forward(self):
        P = self.plan
        for op in P.ops:
            if isinstance(op, ConvOp) and self.small(op.out):
                inp = self.source(op)
                y = conv_ref(op, self.sd, inp)
                S = conv_ref(op, self.sd, inp.abs(), self.sd[op.wname].abs(), self.sd[op.bname].abs())
                if op.fwd_add is not None:
                    add = self.n(op.fwd_add.data, op.fwd_add.C)
                    y, S = y + add, S + add.abs()
                if op.relu_out:
                    y = torch.relu(y)
                assert_rounded(self.n(op.out.data, op.out.C), y, ACC * S, f"forward {op.name}")
                self.nchecked += 1
            elif isinstance(op, LrnOp) and self.small(op.y):
                a = self.n(op.a.data, op.a.C)
                y = O.lrn(a, op.a.C)
                assert_rounded(self.n(op.y.data, op.y.C), y, 2e-6 * y.abs(), f"forward {op.name}")
                self.nchecked += 1
            elif isinstance(op, PoolOp) and self.small(op.x):
                x = self.n(op.x.data, op.x.C)
                y = F.max_pool2d(F.pad(x, (0, op.x.W % 2, 0, op.x.H % 2)), 2, 2)
                assert torch.equal(self.n(op.y.data, op.y.C), y), f"forward {op.name}"
                self.nchecked += 1
            elif isinstance(op, AttnCoreOp) and self.small(op.y):
                B, N = self.nb, op.N
                f, g, h, x = (self.n(t.data, t.C).reshape(B, t.C, N) for t in (op.f, op.g, op.h, op.x))
                beta = torch.softmax(torch.matmul(g.transpose(1, 2), f), dim=-1)
                y = torch.matmul(h, beta) + x
                # the probabilities enter the second product rounded to bf16: 2^-9 of every |term|
                S = torch.matmul(h.abs(), beta)
                assert_rounded(self.n(op.y.data, op.y.C).reshape(B, -1, N), y, 2.0 ** -8 * S + ACC * x.abs(), f"forward {op.name}")
                self.nchecked += 1

    # ---- backward: the stored gradient of an activation = sum of its consumers' contributions ---------------------------
    def conv_dgrad(self, op, g, src_index, absolute=False):
        """d(conv output)/d(source src_index) applied to g: float64 autograd through conv_ref (a linear map)"""
        C1 = op.x1.C
        Cin = C1 + (op.x2.C if op.x2 is not None else 0)
        z = torch.zeros((self.nb, Cin, op.x1.H, op.x1.W), dtype=torch.float64, requires_grad=True)
        w = self.sd[op.wname]
        conv_ref(op, self.sd, z, w.abs() if absolute else w, torch.zeros_like(self.sd[op.bname])).backward(g.abs() if absolute else g)
        return z.grad[:, :C1] if src_index == 0 else z.grad[:, C1:]

    def ce_grad(self, act):
        lg = self.n(act.data, act.C)
        lab = self.labels.cpu() if self.samples is None else self.labels[self.samples].cpu()
        B = lg.shape[0]
        p = torch.softmax(lg, 1)
        oh = F.one_hot(lab, act.C).permute(0, 3, 1, 2).double()
        cnt = (lab != 0).reshape(B, -1).sum(1).clamp_min(1).double().view(B, 1, 1, 1)
        return (p - oh) * (lab != 0).unsqueeze(1).double() / cnt / self.plan.B             # (mean over the WHOLE batch)

    def unwritten(self, act):
        """gradient buffers no launch writes: an LRN output whose backward rides on the pair launch (dy never stored) and the
        intermediate of a pair whose first weight gradient rides on it"""
        lrn = getattr(act, "lrn_producer", None)
        if lrn is not None and lrn.bwd_fused_into is not None:
            return True
        for pr in self.plan.pairs:
            if pr.active and pr.c1.out is act and pr.c1.wg_fused:
                return True
        return False

    def backward(self):
        P = self.plan
        for t in P.acts:
            if t.grad is None or not self.small(t) or self.unwritten(t) or t.n_contrib == 0:
                continue
            tot = torch.zeros((self.nb, t.C, t.H, t.W), dtype=torch.float64)
            S = torch.zeros_like(tot)
            mag = torch.zeros_like(tot)
            n, ok = 0, True
            data = self.n(t.data, t.C)
            for op in P.ops:
                if isinstance(op, ConvOp):
                    for si, x in enumerate((op.x1, op.x2)):
                        if x is t and op.slots[si] is not None and op.out.grad is not None:
                            if self.unwritten(op.out):
                                ok = False
                                continue
                            g = self.n(op.out.grad, op.out.C)
                            part, pabs = self.conv_dgrad(op, g, si), self.conv_dgrad(op, g, si, True)
                            if op.relu_in:
                                part, pabs = part * (data > 0), pabs * (data > 0)
                            tot, S, n, mag = tot + part, S + pabs, n + 1, mag + part.abs()
                            if si == 0 and op.bwd_add is not None:
                                ga = self.n(op.bwd_add.grad, op.bwd_add.C)
                                tot, S, mag = tot + ga, S + ga.abs(), mag + ga.abs()
                elif isinstance(op, LrnOp) and op.a is t:
                    if self.unwritten(op.y):
                        ok = False
                        continue
                    dy = self.n(op.y.grad, op.y.C)
                    part, terms = lrn_bwd_ref(data, dy)
                    tot, S, n, mag = tot + part, S + 0.2 * terms, n + 1, mag + part.abs()
                elif isinstance(op, PoolOp) and op.x is t and op.y.grad is not None:
                    dy = self.n(op.y.grad, op.y.C)
                    idx = (op.idx if self.samples is None else op.idx[self.samples])[..., :t.C].permute(0, 3, 1, 2).cpu().long()
                    part = torch.zeros((self.nb, t.C, 2 * op.y.H, 2 * op.y.W), dtype=torch.float64)
                    for pos in range(4):
                        part[:, :, pos // 2::2, pos % 2::2] = dy * (idx == pos)
                    tot, n, mag = tot + part[:, :, :t.H, :t.W], n + 1, mag + part[:, :, :t.H, :t.W].abs()
                elif isinstance(op, AttnCoreOp) and t in (op.f, op.g, op.h):
                    ok = False                      # the attention core's gradients: norm-checked in test_ops_gpu.py (bf16 probabilities)
            if t.name in P.ext_slot:
                ce = self.ce_grad(t)
                tot, n, mag = tot + ce, n + 1, mag + ce.abs()
                # the loss gradient is fp32 softmax arithmetic (exp, sum, divide, weight, 1 / count): a few fp32 ulps of its own size.
                # (The slack is ACC * S: round 4 added 1e-5 |ce| here, i.e. 3e-10 relative -- less than ONE fp32 ulp -- and the check
                #  tripped over exact bf16 ties as soon as the kernel's operation order changed.)
                S = S + (2e-6 / ACC) * ce.abs() + 1e-12
            if not ok or n == 0:
                continue
            if t.relu_out:
                tot, S = tot * (data > 0), S * (data > 0)
            assert_rounded(self.n(t.grad, t.C), tot, ACC * S, f"gradient of {t.name} ({n} contributions)", half_ulps=n, mag=mag)
            self.nchecked += 1

    def weight_grads(self, names=None, bound=2e-5):
        P, m = self.plan, self.m
        out = {}
        for op in P.ops:
            if not isinstance(op, ConvOp) or op.wdesc is None or op.out.grad is None or not self.small(op.out):
                continue
            if self.unwritten(op.out) or (names is not None and op.name not in names):
                continue
            g = nchw(op.out.grad, op.out.C)
            inp = self.source(op, whole_batch=True)
            res = []
            for absolute in (False, True):
                w = self.sd[op.wname].clone().requires_grad_(True)
                b = self.sd[op.bname].clone().requires_grad_(True)
                conv_ref(op, self.sd, inp.abs() if absolute else inp, w, b).backward(g.abs() if absolute else g)
                res.append((w.grad, b.grad))
            for (ref, S), key in zip(zip(res[0], res[1]), (op.wname, op.bname)):
                off = m._poff[key]
                got = self.eng.flat_grad[off:off + ref.numel()].view(ref.shape).double().cpu()
                d = (got - ref).abs()
                tol = bound * S + 1e-12
                assert bool((d <= tol).all()), (f"weight gradient {op.name} {key.rsplit('.', 1)[-1]}: worst error / sum|x||g| = "
                                                f"{float((d / S.clamp_min(1e-30)).max()):.2e} (bound {bound})")
                out[key] = got
            self.nchecked += 1
        return out


def _run(B, H, W, C, seed, env, max_h, monkeypatch, min_h=0, samples=None):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    L.load().msau_reload_env()
    cfg = dict(O.DEFAULT_CFG, channels=C, num_blocks=3)
    sd0 = O.init_params(cfg, seed=seed)
    x, label = O.synthetic_batch(B, C, H, W, 5, seed=seed + 1)
    kw = dict(scale_space_num=4, res_depth=2, featRoot=8, final_act="softmax", num_blocks=3, dtype="bf16")
    m = MSAUWrapper(C, 5, kw)
    m.load_state_dict(sd0)
    m = m.cuda()
    eng = TrainEngine(m)
    eng.step(x.cuda(), label.cuda())
    torch.cuda.synchronize()
    plan = m._plan_for_shape(B, H, W, torch.device("cuda", 0), True)
    return Checker(m, eng, plan, sd0, label.cuda().reshape(B, H, W), max_h, min_h, samples)


# (B, H, W, min_h, max_h, samples)
CASES = [pytest.param(3, 176, 144, 0, 10 ** 6, None, id="3x176x144-all-levels"),
         pytest.param(16, 336, 256, 0, 84, None, id="bench-shape-levels-2-3"),
         pytest.param(16, 336, 256, 84, 10 ** 6, (0, 15), id="bench-shape-levels-0-1-samples-0-15")]


@pytest.mark.parametrize("B,H,W,min_h,max_h,samples", CASES)
def test_every_launch_is_correctly_rounded(monkeypatch, B, H, W, min_h, max_h, samples):
    """(3, 176x144): the row-streaming kernels at levels 0-1, lean instances at level 1, generic kernels below -- every op checked.
    (16, 336x256), levels 2-3: the BENCH shape; the float64 references are computed for 84x64 and 42x32 (the lean / split /
    tile-pair / strided instances the headline runs there, which no smaller image reaches), all 16 samples.
    (16, 336x256), levels 0-1: the launches that carry 60 % of the headline's bytes (row pairs, row convs incl. LRN output, dual
    3x3 / 1x1, 4x4, the 16-channel pairs with the pooled output, the first conv fed with the NCHW tensor) AT the bench shape --
    segment heights, strip counts and XCD task runs differ from the small image's; forward outputs and stored gradients of the
    first and the last sample of the batch element by element, weight gradients (they sum the batch) over all 16."""
    rows = max_h > 84                                  # the case reaches the levels the row-streaming kernels take
    try:
        # riders off: every intermediate gradient is stored, every launch can be checked on its own operands
        ck = _run(B, H, W, 64, 51, {"MSAU_FUSE_LRN_BWD": "0", "MSAU_PAIR_WGRAD": "0", "MSAU_ROWS_MIN_TASKS": "1"}, max_h, monkeypatch,
                  min_h, samples)
        assert not any(pr.c1.wg_fused for pr in ck.plan.pairs)
        ck.forward()
        ck.backward()
        wg_off = ck.weight_grads()
        assert ck.nchecked > (250 if min_h == 0 and rows else 120), ck.nchecked
        inst = {ck.plan.rec_meta[a][0] for a in ck.plan.rec_meta}
        if rows:
            assert any(k.startswith("rowpair_kernel<bf16,C8>") for k in inst) and any(k.startswith("rowconv_kernel") for k in inst), inst
        else:
            assert any(k.startswith("conv_pair_kernel<bf16,C32>") for k in inst) and any("CIN64,CT4,K3" in k for k in inst), inst
        if not rows:
            return
        grads_off = {t.name: t.grad.clone() for t in ck.plan.acts if t.grad is not None and not ck.unwritten(t)}
        # (1) the weight-gradient rider alone: it changes NO data gradient -- every stored gradient bit for bit, the riding weight
        #     gradients within the fp32 bound
        ck1 = _run(B, H, W, 64, 51, {"MSAU_FUSE_LRN_BWD": "0", "MSAU_PAIR_WGRAD": "1", "MSAU_ROWS_MIN_TASKS": "1"}, max_h, monkeypatch,
                   min_h, samples)
        fused = [pr for pr in ck1.plan.pairs if pr.active and pr.c1.wg_fused]
        assert fused and not any(isinstance(op, LrnOp) and op.bwd_fused_into is not None for op in ck1.plan.ops)
        for t in ck1.plan.acts:
            if t.grad is not None and not ck1.unwritten(t) and t.name in grads_off:
                assert torch.equal(t.grad, grads_off[t.name]), t.name               # (whole batch, on the device)
        ck1.weight_grads(names={pr.c1.name for pr in fused} | {pr.c2.name for pr in fused})
        # (2) both riders (the shipped default).  The LRN backward in the pair's epilogue runs the stand-alone pass's arithmetic on
        #     the same storage-rounded dy, but in another instruction context (other fma contractions): at 3x176x144 it came out bit
        #     for bit, at the bench shape a handful of its 22 M outputs land on the other side of a bf16 rounding boundary -- and every
        #     gradient computed AFTER it (the earlier stages' whole backward) legitimately inherits that.  So: the rider of the LAST
        #     stage (the first to run: everything before it is bit-identical to the riders-off run, whose stored dy is therefore ITS dy)
        #     is held to the float64 bound; every other stored gradient to <= 2 bf16 ulps of the riders-off value on all but 1e-2 of
        #     its elements and 3e-3 in rel-L2 (a wrong rider term would be off by orders of magnitude more).
        ck2 = _run(B, H, W, 64, 51, {"MSAU_FUSE_LRN_BWD": "1", "MSAU_PAIR_WGRAD": "1", "MSAU_ROWS_MIN_TASKS": "1"}, max_h, monkeypatch,
                   min_h, samples)
        fused = [pr for pr in ck2.plan.pairs if pr.active and pr.c1.wg_fused]
        riders = [op for op in ck2.plan.ops if isinstance(op, LrnOp) and op.bwd_fused_into is not None]
        assert fused and riders
        last_stage = max(op.stage for op in riders)
        # (the rider that runs FIRST in the backward: the last stage's deepest level -- since round 5 the 16-channel pair carries one too)
        first_rider = min((op for op in riders if op.stage == last_stage), key=lambda op: op.a.H)
        exact = 0
        for t in ck2.plan.acts:
            if t.grad is None or ck2.unwritten(t) or t.name not in grads_off:
                continue
            rider = next((op for op in riders if op.a is t), None)
            if rider is not None and rider is first_rider:
                ref, terms = lrn_bwd_ref(ck2.n(t.data, t.C), ck2.n(grads_off[rider.y.name], rider.y.C))
                assert_rounded(ck2.n(t.grad, t.C), ref, ACC * 0.2 * terms, f"LRN-backward rider {t.name}")
                exact += 1
                continue
            a, b = t.grad.float(), grads_off[t.name].float()
            if torch.equal(a, b):
                exact += 1
                continue
            ulp = torch.pow(2.0, torch.floor(torch.log2(torch.maximum(a.abs(), b.abs()).clamp_min(2.0 ** -120))) - 7)
            far = ((a - b).abs() > 2 * ulp).float().mean()
            rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
            assert float(far) < 1e-2 and rel < 3e-3, (t.name, float(far), rel)     # (observed at the net's first tensor, where everything has piled up: 4.4e-3, 8e-4)
        assert exact >= 3, exact                     # (at least the last stage's tensors ahead of its rider are bit-equal)
        ck2.weight_grads(names={pr.c1.name for pr in fused} | {pr.c2.name for pr in fused})
    finally:
        monkeypatch.undo()
        L.load().msau_reload_env()


@pytest.mark.parametrize("c,hw", [(8, (57, 61)), (8, (90, 29)), (16, (71, 250)), (16, (40, 15))])
def test_row_pair_forward_with_zero_bias_is_bit_identical_to_the_tile_kernels(monkeypatch, c, hw):
    """with zero biases the row-streaming pair sums the same products in the same order as the two single
    launches (the bias is the only thing that enters differently: as the MFMA's C operand)"""
    from tests.hip_harness import Act, run_graph
    from msau_amd.plan import PairOp
    torch.manual_seed(5)
    H, W = hw
    B = 2
    x = torch.randn(B, c, H, W)
    p = {"w": 0.2 * torch.randn(c, c, 3, 3), "b": torch.zeros(c), "w2": 0.2 * torch.randn(c, c, 3, 3), "b2": torch.zeros(c)}
    gy = torch.randn(B, c, H, W)
    mids = []

    def build(plan):
        x0 = plan.x_in
        r1 = Act(plan, "r1", H, W, c, relu_out=True)
        c1 = ConvOp(plan, "c1", x0, None, "w", "b", r1, 3, relu_in=True, relu_out=True)
        out = Act(plan, "out", H, W, c, relu_out=True)
        c2 = ConvOp(plan, "c2", r1, None, "w2", "b2", out, 3, relu_out=True, fwd_add=x0)
        c1.bwd_add = out
        PairOp(plan, c1, c2)
        plan.logits = out
        mids.append((plan, r1))
    monkeypatch.setenv("MSAU_ROWS_MIN_TASKS", "1")
    res = {}
    try:
        for mode, env in (("rows", {"MSAU_PAIR_ROWS": "1"}), ("single", {"MSAU_FUSE_PAIR": "0", "MSAU_CONV_ROWS": "0", "MSAU_PAIR_ROWS": "0"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            L.load().msau_reload_env()
            y = run_graph(build, p, x, gy, L.BF16)[0]
            res[mode] = (y, mids[-1][1].data.clone())
            for k in env:
                monkeypatch.delenv(k)
    finally:
        monkeypatch.undo()
        L.load().msau_reload_env()
    assert mids[0][0].pairs[0].active and not mids[1][0].pairs[0].active
    assert torch.equal(res["rows"][1], res["single"][1]), "intermediate"
    assert torch.equal(res["rows"][0], res["single"][0]), "block output"
