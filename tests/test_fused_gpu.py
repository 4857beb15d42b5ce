"""GPU: the fused forms added in round 2 against the launches they replace, at sizes where the specialised ("lean")
instances take the launch (>= 64 pixel tiles) -- the golden op fixtures are too small for that and go through the generic
kernels.  Everything through the C ABI (msau_amd.plan -> msau_run_ops / direct ctypes calls).

  * MSAU_CONV_POOL   zero pad + MaxPool2d(2,2) in the producing conv's epilogue: bit-exact (values, positions -> gradients)
  * MSAU_CONV_LRN    LocalResponseNorm(size=C) in the level-entry conv's epilogue: same arithmetic, other summation order
  * msau_conv2d_wgrad_group   several weight gradients of one shape in one grid: the same slabs, bit for bit
  * the role-swapped 64 -> 8 weight gradient (the net's first conv) against autograd
"""
import ctypes as C
import os

import pytest
import torch

from msau_amd import _lib as L
from oracle import msau_oracle as O
from tests.golden_util import err
from tests.hip_harness import Act, ConvOp, LrnOp, PoolOp, run_graph

pytestmark = pytest.mark.gpu
DT = [pytest.param(L.F32, id="f32"), pytest.param(L.BF16, id="bf16")]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c,k,dual,hw", [(8, 1, True, (65, 63)), (16, 1, True, (49, 70)), (32, 1, True, (49, 64)), (32, 3, False, (64, 61))])
def test_pool_in_the_conv_epilogue_is_bit_exact(monkeypatch, dtype, c, k, dual, hw):
    """the instances that carry the pooled output: the coupling 1x1 conv over concat(prev, cur) (model/model.py:143-148) and
    the channel-split 32 -> 32 3x3 (second conv of the level-2 residual block)"""
    if dtype == L.F32 and k == 3:
        pytest.skip("fp32 32 -> 32 3x3 stages its K in two chunks and runs the generic kernel: nothing to fuse into")
    torch.manual_seed(11)
    B, (H, W) = 4, hw
    x = torch.randn(B, c, H, W)
    p = {"w": 0.2 * torch.randn(c, 2 * c if dual else c, k, k), "b": 0.1 * torch.randn(c), "w0": 0.3 * torch.randn(c, c, 1, 1),
         "b0": 0.1 * torch.randn(c)}
    gy = torch.randn(B, c, (H + 1) // 2, (W + 1) // 2)
    took = []

    def build(plan):
        xi = plan.x_in
        y = Act(plan, "y", H, W, c, relu_out=True)
        if dual:
            other = Act(plan, "o", H, W, c)
            ConvOp(plan, "c0", xi, None, "w0", "b0", other, 1)
            ConvOp(plan, "c", xi, other, "w", "b", y, k, relu_out=True)
        else:
            ConvOp(plan, "c", xi, None, "w", "b", y, k, relu_out=True)
        q = Act(plan, "q", (H + 1) // 2, (W + 1) // 2, c)
        po = PoolOp(plan, "p", y, q)
        plan.logits = q
        took.append(po)
    if not dual:
        p.pop("w0"); p.pop("b0")
    out = {}
    # the pooled epilogue lives in the tile kernels: keep the un-pooled mode on them too (the row-streaming instance of the
    # 8-channel coupling conv sums in another order and has its own test)
    monkeypatch.setenv("MSAU_CONV_ROWS", "0")
    L.load().msau_reload_env()
    try:
        for mode in ("1", "0"):
            monkeypatch.setenv("MSAU_FUSE_POOL", mode)
            out[mode] = run_graph(build, p, x, gy, dtype)
    finally:
        monkeypatch.undo()
        L.load().msau_reload_env()
    assert took[0].fused_into is not None and took[1].fused_into is None
    for a, b in zip(out["1"], out["0"]):
        if isinstance(a, dict):
            for n in a:
                assert torch.equal(a[n], b[n]), n
        elif a is not None:
            assert torch.equal(a, b)
    # and against autograd
    xr = x.clone()
    src = torch.cat([xr, O.conv_same(xr, p["w0"], p["b0"])], 1) if dual else xr
    yr = torch.nn.functional.max_pool2d(torch.nn.functional.pad(O.conv_same(src, p["w"], p["b"], relu=True), (0, W % 2, 0, H % 2)), 2, 2)
    assert err(out["1"][0], yr, dtype == L.BF16) < (3e-2 if dtype == L.BF16 else 1e-4)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C8,hw", [(8, (57, 61)), (16, (60, 129)), (8, (70, 250))])
def test_pool_in_the_residual_pair_epilogue_is_bit_exact(monkeypatch, dtype, C8, hw):
    """conv -> ReLU -> conv -> add -> ReLU (one msau_conv_pair launch) followed by the pool: fused vs stand-alone pool"""
    torch.manual_seed(12)
    B, (H, W), c = 4, hw, C8
    x = torch.randn(B, c, H, W)
    p = {"w": 0.2 * torch.randn(c, c, 3, 3), "b": 0.1 * torch.randn(c), "w2": 0.2 * torch.randn(c, c, 3, 3), "b2": 0.1 * torch.randn(c)}
    gy = torch.randn(B, c, (H + 1) // 2, (W + 1) // 2)
    took = []

    def build(plan):
        x0 = plan.x_in
        r1 = Act(plan, "r1", H, W, c, relu_out=True)
        c1 = ConvOp(plan, "c1", x0, None, "w", "b", r1, 3, relu_in=True, relu_out=True)
        out = Act(plan, "out", H, W, c, relu_out=True)
        c2 = ConvOp(plan, "c2", r1, None, "w2", "b2", out, 3, relu_out=True, fwd_add=x0)
        c1.bwd_add = out                      # d(x0) += g(block output), as Plan._res_block
        from msau_amd.plan import PairOp
        PairOp(plan, c1, c2)
        q = Act(plan, "q", (H + 1) // 2, (W + 1) // 2, c)
        took.append((PoolOp(plan, "p", out, q), plan))
        plan.logits = q
    out = {}
    monkeypatch.setenv("MSAU_PAIR_POOL_MINC", "8")          # the 8-channel pooled pair instance is off by default (slower than the pool launch)
    if c == 8:
        # only the tile kernels pool at 8 channels; without this the un-pooled mode would run the row-streaming kernel, which
        # sums in another order (16 channels: the row kernel pools too, both modes run it)
        monkeypatch.setenv("MSAU_PAIR_ROWS", "0")
        L.load().msau_reload_env()
    try:
        for mode in ("1", "0"):
            monkeypatch.setenv("MSAU_FUSE_POOL", mode)
            out[mode] = run_graph(build, p, x, gy, dtype)
    finally:
        monkeypatch.undo()
        L.load().msau_reload_env()
    pool1, plan1 = took[0]
    assert plan1.pairs[0].active and pool1.fused_into is plan1.pairs[0] and took[1][0].fused_into is None
    for a, b in zip(out["1"], out["0"]):
        if isinstance(a, dict):
            for n in a:
                assert torch.equal(a[n], b[n]), n
        elif a is not None:
            assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c,hw", [(8, (57, 61)), (16, (60, 129)), (8, (71, 250))])
def test_relu_masks_as_bit_planes_give_the_same_gradients(monkeypatch, dtype, c, hw):
    """the residual pair's backward reads (mid > 0) and (x > 0) from bit planes its forward wrote, instead of the tensors"""
    torch.manual_seed(16)
    B, (H, W) = 4, hw
    x = torch.randn(B, c, H, W)
    p = {"w": 0.2 * torch.randn(c, c, 3, 3), "b": 0.1 * torch.randn(c), "w2": 0.2 * torch.randn(c, c, 3, 3), "b2": 0.1 * torch.randn(c)}
    gy = torch.randn(B, c, H, W)
    plans = []

    def build(plan):
        x0 = plan.x_in
        r1 = Act(plan, "r1", H, W, c, relu_out=True)
        c1 = ConvOp(plan, "c1", x0, None, "w", "b", r1, 3, relu_in=True, relu_out=True)
        out = Act(plan, "out", H, W, c, relu_out=True)
        c2 = ConvOp(plan, "c2", r1, None, "w2", "b2", out, 3, relu_out=True, fwd_add=x0)
        c1.bwd_add = out
        from msau_amd.plan import PairOp
        PairOp(plan, c1, c2)
        plan.logits = out
        plans.append(plan)
    monkeypatch.setenv("MSAU_PAIR_ROWS", "0")      # the tile kernels in both modes (the row kernel has its own test below)
    L.load().msau_reload_env()
    out = {}
    try:
        for mode in ("1", "0"):
            monkeypatch.setenv("MSAU_PAIR_BITS", mode)
            out[mode] = run_graph(build, p, x, gy, dtype)
    finally:
        monkeypatch.undo()
        L.load().msau_reload_env()
    assert plans[0].pairs[0].bdesc is not None and plans[0].pairs[0].bdesc.bits_mid and not plans[1].pairs[0].bdesc.bits_mid
    for a, b in zip(out["1"], out["0"]):
        if isinstance(a, dict):
            for n in a:
                assert torch.equal(a[n], b[n]), n
        elif a is not None:
            assert torch.equal(a, b)


@pytest.mark.parametrize("c", [8, 16])
@pytest.mark.parametrize("hw,B,sh", [((57, 61), 3, 0), ((71, 250), 2, 0), ((33, 30), 2, 10), ((16, 91), 2, 4), ((90, 29), 2, 16), ((64, 64), 2, 22),
                                     ((45, 14), 2, 0), ((40, 15), 1, 7)])
def test_row_streaming_pair_matches_the_tile_kernels_and_the_oracle(monkeypatch, c, hw, B, sh):
    """conv_rows.hip (8 / 16 channels, bf16: a wave walks a 30- / 14-column strip row by row, pixel-pair packed MFMA rows at 8
    channels, ReLU masks as lane ballots) against the tile kernels of conv_pair.hip on the same rounded inputs, and both against the fp32 oracle:
    forward, input gradient, weight / bias gradients; strips that end inside / at / beyond the image edge, segments of
    every height (MSAU_ROWS_SH), images shorter than one segment."""
    torch.manual_seed(21)
    H, W = hw
    x = torch.randn(B, c, H, W)
    p = {"w": 0.2 * torch.randn(c, c, 3, 3), "b": 0.1 * torch.randn(c), "w2": 0.2 * torch.randn(c, c, 3, 3), "b2": 0.1 * torch.randn(c)}
    gy = torch.randn(B, c, H, W)
    plans = []

    def build(plan):
        x0 = plan.x_in
        r1 = Act(plan, "r1", H, W, c, relu_out=True)
        c1 = ConvOp(plan, "c1", x0, None, "w", "b", r1, 3, relu_in=True, relu_out=True)
        out = Act(plan, "out", H, W, c, relu_out=True)
        c2 = ConvOp(plan, "c2", r1, None, "w2", "b2", out, 3, relu_out=True, fwd_add=x0)
        c1.bwd_add = out
        from msau_amd.plan import PairOp
        PairOp(plan, c1, c2)
        plan.logits = out
        plans.append(plan)
    monkeypatch.setenv("MSAU_ROWS_MIN_TASKS", "1")
    if sh:
        monkeypatch.setenv("MSAU_ROWS_SH", str(sh))
    out = {}
    try:
        for mode in ("1", "0"):
            monkeypatch.setenv("MSAU_PAIR_ROWS", mode)
            L.load().msau_reload_env()
            out[mode] = run_graph(build, p, x, gy, L.BF16)
    finally:
        monkeypatch.undo()
        L.load().msau_reload_env()
    assert plans[0].pairs[0].active and plans[0].pairs[0].bdesc is not None
    # the planes follow the instance: 32 bytes of ballots per (row, 30-column strip) against a byte per pixel (small images:
    # mode "0" has no tile instance either and runs the two convs as separate launches)
    assert plans[0].pairs[0].bits_mid.numel() == B * H * -(-W // (30 if c == 8 else 14)) * 32
    assert not plans[1].pairs[0].active or plans[1].pairs[0].bits_mid.numel() == B * H * W * (c // 8)
    # fp32 reference on the bf16-rounded inputs and weights
    xr = x.bfloat16().float().requires_grad_(True)
    pr = {k: v.bfloat16().float().requires_grad_(True) for k, v in p.items()}
    F = torch.nn.functional
    r1 = F.relu(F.conv2d(F.relu(xr), pr["w"], pr["b"], padding=1))
    ref = F.relu(F.conv2d(r1, pr["w2"], pr["b2"], padding=1) + xr)
    ref.backward(gy.bfloat16().float())
    for mode in ("1", "0"):
        y, _, dx, gr = out[mode]
        assert err(y, ref.detach(), True) < 2e-2, mode
        assert err(dx, xr.grad, True) < 6e-2, mode
        for n in p:
            assert err(gr[n], pr[n].grad, True) < 8e-2, (mode, n)      # sums through two ReLU masks of bf16-rounded gradients
    # and against each other: the same products in another summation order, then the same rounding
    for a, b in zip(out["1"], out["0"]):
        if isinstance(a, dict):
            for n in a:
                assert err(a[n], b[n], True) < 2e-2, n
        elif a is not None:
            assert err(a, b, True) < 1e-2


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cin,cout,dil,hw", [(8, 8, 1, (130, 140)), (8, 16, 2, (66, 65)), (16, 32, 4, (52, 70)), (32, 64, 8, (70, 83))])
def test_lrn_in_the_conv_epilogue_matches_the_standalone_pass(monkeypatch, dtype, cin, cout, dil, hw):
    torch.manual_seed(13)
    B, (H, W) = 4, hw
    x = torch.randn(B, cin, H, W)
    p = {"w": 0.3 * torch.randn(cout, cin, 3, 3), "b": 0.1 * torch.randn(cout)}
    gy = torch.randn(B, cout, H, W)
    took = []

    def build(plan):
        a = Act(plan, "a", H, W, cout)
        ConvOp(plan, "c", plan.x_in, None, "w", "b", a, 3, dil=dil)
        y = Act(plan, "y", H, W, cout)
        took.append(LrnOp(plan, "l", a, y))
        plan.logits = y
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MSAU_FUSE_LRN", mode)
        out[mode] = run_graph(build, p, x, gy, dtype)
    fits = cout != 64                                       # (the 64-channel LRN stays its own launch: the fused instance measured slower)
    assert (took[0].fused_into is not None) == fits and took[1].fused_into is None
    bf = dtype == L.BF16
    # same inputs (the storage-rounded conv result), same formula, different order of the window sums: one bf16 ulp at most
    assert err(out["1"][0], out["0"][0], bf) < (4e-3 if bf else 1e-6)
    assert torch.equal(out["1"][2], out["0"][2])           # the backward does not depend on who wrote y
    xr = x.clone().requires_grad_(True)
    yr = O.lrn(O.conv_same(xr, p["w"], p["b"], dilation=dil), cout)
    if True:
        assert err(out["1"][0], yr.detach(), bf) < (3e-2 if bf else 1e-4)


def _wdesc(x, g, slabs, cin, cout, k, nslabs):
    B, H, W, _ = x.shape
    w = L.WgradDesc()
    w.B, w.Hin, w.Win, w.Hout, w.Wout = B, H, W, H, W
    w.C1, w.C2, w.Cout, w.KH, w.KW = cin, 0, cout, k, k
    w.dil, w.pad_t, w.pad_l, w.stride, w.flags = 1, (k - 1) // 2, (k - 1) // 2, 1, 0      # SAME: 4x4 pads 1 before, 2 after (layers/utils.py:10-19)
    w.x1, w.x2, w.g, w.slabs, w.nslabs = x.data_ptr(), None, g.data_ptr(), slabs.data_ptr(), nslabs
    return w


@pytest.mark.parametrize("cin,cout,k,hw", [(32, 32, 3, (42, 32)), (64, 64, 1, (21, 16)), (8, 8, 3, (64, 80))])
def test_grouped_weight_gradients_write_the_same_slabs(cin, cout, k, hw, monkeypatch):
    torch.manual_seed(14)
    lib = L.load()
    monkeypatch.setenv("MSAU_WGRAD_ROWS", "0")           # grouping belongs to the tile kernels (the row kernel runs one layer per launch)
    lib.msau_reload_env()
    s = torch.cuda.current_stream().cuda_stream
    B, (H, W), n = 4, hw, 3
    xs = [(torch.randn(B, H, W, cin, device="cuda") * 0.5).to(torch.bfloat16) for _ in range(n)]
    gs = [(torch.randn(B, H, W, cout, device="cuda") * 0.5).to(torch.bfloat16) for _ in range(n)]
    probe = _wdesc(xs[0], gs[0], xs[0], cin, cout, k, 1)
    geom = L.WgradGeom()
    L.check(lib.msau_wgrad_geometry(L.BF16, C.byref(probe), C.byref(geom)), "geometry")
    nslabs = min(geom.max_slabs, 48)
    elems = nslabs * geom.slab_bytes // 4
    one = [torch.full((elems,), float("nan"), device="cuda") for _ in range(n)]
    grp = [torch.full((elems,), float("nan"), device="cuda") for _ in range(n)]
    d1 = [_wdesc(xs[i], gs[i], one[i], cin, cout, k, nslabs) for i in range(n)]
    d2 = [_wdesc(xs[i], gs[i], grp[i], cin, cout, k, nslabs) for i in range(n)]
    assert lib.msau_conv2d_wgrad_groupable(L.BF16, C.byref(d2[0]), C.byref(d2[1])) == 1
    for d in d1:
        L.check(lib.msau_conv2d_wgrad(s, L.BF16, C.byref(d)), "wgrad")
    arr = (C.POINTER(L.WgradDesc) * n)(*[C.pointer(d) for d in d2])
    L.check(lib.msau_conv2d_wgrad_group(s, L.BF16, arr, n), "wgrad_group")
    torch.cuda.synchronize()
    monkeypatch.undo()
    lib.msau_reload_env()
    for a, b in zip(one, grp):
        assert not torch.isnan(a).any() and torch.equal(a, b)
    # a launch of another shape is refused, not mis-grouped
    other = _wdesc(xs[0], gs[0], grp[0], cin, cout, k, max(1, nslabs - 1))
    assert lib.msau_conv2d_wgrad_groupable(L.BF16, C.byref(d2[0]), C.byref(other)) == 0


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cin,hw,B", [(64, (37, 45), 3), (128, (33, 18), 2), (64, (16, 16), 1)])
def test_first_conv_weight_gradient_vs_autograd(dtype, cin, hw, B):
    """Cin in 64-channel chunks -> 8, 3x3: bf16 takes the role-swapped instance (wgrad_lean.hip, wgrad_in64_kernel), partial
    tiles and the bias column included; fp32 the generic one"""
    torch.manual_seed(15)
    H, W = hw
    x = (torch.rand(B, cin, H, W) < 0.3).float()            # chargrid-like occupancy
    p = {"w": 0.1 * torch.randn(8, cin, 3, 3), "b": 0.1 * torch.randn(8)}
    gy = torch.randn(B, 8, H, W)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    yr = O.conv_same(x, leaves["w"], leaves["b"])
    yr.backward(gy)

    def build(plan):
        y = Act(plan, "y", H, W, 8)
        ConvOp(plan, "c", plan.x_in, None, "w", "b", y, 3)
        plan.logits = y
    y, _, _, grads = run_graph(build, p, x, gy, dtype)
    bf = dtype == L.BF16
    assert err(y, yr.detach(), bf) < (3e-2 if bf else 1e-4)
    assert err(grads["w"], leaves["w"].grad, bf) < (3e-2 if bf else 1e-4)
    assert err(grads["b"], leaves["b"].grad, bf) < (3e-2 if bf else 1e-4)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cin,cout,hw,B", [(128, 8, (65, 63), 4), (192, 16, (33, 70), 5), (768, 8, (48, 64), 6)])
def test_chunked_first_conv_vs_autograd(dtype, cin, cout, hw, B):
    """many input channels -> one 16-row tile, 3x3 (the 768 -> 8 first conv of cfg 4): conv_chunked_kernel walks the 64-channel
    chunks of the packed image with the accumulators in registers; forward against autograd, the weight gradient too"""
    torch.manual_seed(17)
    H, W = hw
    x = torch.randn(B, cin, H, W)
    p = {"w": 0.05 * torch.randn(cout, cin, 3, 3), "b": 0.1 * torch.randn(cout)}
    gy = torch.randn(B, cout, H, W)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    yr = O.conv_same(x, leaves["w"], leaves["b"])
    yr.backward(gy)

    def build(plan):
        y = Act(plan, "y", H, W, cout)
        ConvOp(plan, "c", plan.x_in, None, "w", "b", y, 3)
        plan.logits = y
    y, _, _, grads = run_graph(build, p, x, gy, dtype)
    bf = dtype == L.BF16
    assert err(y, yr.detach(), bf) < (3e-2 if bf else 1e-4)
    assert err(grads["w"], leaves["w"].grad, bf) < (3e-2 if bf else 1e-4)
    assert err(grads["b"], leaves["b"].grad, bf) < (3e-2 if bf else 1e-4)


@pytest.mark.parametrize("cin,cout,k,hw", [(32, 32, 3, (42, 32)), (64, 64, 3, (21, 16)), (64, 64, 1, (21, 16)), (16, 32, 3, (84, 64)), (32, 64, 3, (33, 20)),
                                           (8, 8, 3, (64, 80)), (16, 16, 3, (50, 37)),
                                           # the row-streaming instance (conv_rows.hip, bf16): strips ending inside / at the image edge,
                                           # fewer rows than one segment, one column
                                           (8, 8, 3, (57, 61)), (8, 8, 3, (90, 29)), (8, 8, 3, (17, 30)), (8, 8, 3, (200, 1)),
                                           # ... and its 4x4 form (the end conv: two x fragments per row, four rows carried)
                                           (8, 8, 4, (64, 80)), (8, 8, 4, (57, 61)), (8, 8, 4, (90, 29)), (8, 8, 4, (19, 31)), (8, 8, 4, (200, 1))])
@pytest.mark.parametrize("dtype", DT)
def test_weight_gradient_slabs_vs_fp32_autograd_on_the_same_rounded_inputs(cin, cout, k, hw, dtype, monkeypatch):
    """bf16 x bf16 products are exact in fp32, so the lean weight-gradient instances (pixel-split, two-wave-set, plain) must
    agree with an fp32 autograd weight gradient of the SAME bf16-rounded tensors up to summation order: 2e-5 of the largest
    entry -- three orders of magnitude tighter than the bf16 network tolerances, tight enough to see a lost or doubled tile"""
    torch.manual_seed(18)
    lib = L.load()
    if k == 4:
        monkeypatch.setenv("MSAU_WGRAD_ROWS4", "1")        # (the 4x4 row instance is off by default: measured slower in the step)
        lib.msau_reload_env()
    s = torch.cuda.current_stream().cuda_stream
    B, (H, W) = 3, hw
    td = torch.bfloat16 if dtype == L.BF16 else torch.float32
    x = (torch.randn(B, H, W, cin, device="cuda") * 0.5).to(td)
    g = (torch.randn(B, H, W, cout, device="cuda") * 0.5).to(td)
    probe = _wdesc(x, g, x, cin, cout, k, 1)
    geom = L.WgradGeom()
    L.check(lib.msau_wgrad_geometry(dtype, C.byref(probe), C.byref(geom)), "geometry")
    if not (geom.lean and geom.nchunks == 1):
        pytest.skip("fp32 stages this shape in several K chunks (generic kernel: covered by the golden op tests)")
    nslabs = min(geom.max_slabs, 40)
    slab_elems = geom.slab_bytes // 4
    slabs = torch.full((nslabs * slab_elems,), float("nan"), device="cuda")
    d = _wdesc(x, g, slabs, cin, cout, k, nslabs)
    if (cin, cout) == (8, 8) and k in (3, 4) and dtype == L.BF16 and B * -(-W // 30) * -(-H // 8) >= 16:
        L.check(lib.msau_wgrad_geometry(dtype, C.byref(d), C.byref(geom)), "geometry")
        assert geom.lean == 2                             # the row-streaming instance takes it
    L.check(lib.msau_conv2d_wgrad(s, dtype, C.byref(d)), "wgrad")
    torch.cuda.synchronize()
    tot = slabs.view(nslabs, cout, geom.kext).sum(0).cpu()
    assert not torch.isnan(tot[:, :k * k * cin + 1]).any()
    xr = x.float().permute(0, 3, 1, 2).cpu()
    gr = g.float().permute(0, 3, 1, 2).cpu()
    w = torch.zeros(cout, cin, k, k, requires_grad=True)
    b = torch.zeros(cout, requires_grad=True)
    O.conv_same(xr, w, b).backward(gr)
    got_w = tot[:, :k * k * cin].view(cout, k * k, cin).permute(0, 2, 1).reshape(cout, cin, k, k)      # slab k = [tap][channel]
    got_b = tot[:, k * k * cin]
    if k == 4:
        monkeypatch.delenv("MSAU_WGRAD_ROWS4")
        lib.msau_reload_env()
    assert float((got_w - w.grad).abs().max()) < 2e-5 * float(w.grad.abs().max())
    assert float((got_b - b.grad).abs().max()) < 2e-5 * float(b.grad.abs().max())


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c,cs,hw", [(64, 64, (16, 24)), (768, 768, (8, 16)), (100, 128, (8, 8)), (300, 320, (16, 8)), (13, 16, (6, 6)),
                                     (72, 72, (8, 8)), (128, 128, (5, 7)), (7, 8, (4, 4))])
def test_nchw_to_nhwc_every_mapping(dtype, c, cs, hw):
    """the NCHW fp32 -> NHWC conversion at the boundary (train_chargrid_funsd_msau.py:50-53 hands over NCHW fp32): the
    narrow and the many-channel ("wide") thread mappings of the register kernel, the LDS-tiled kernel (H*W % 4 != 0) and
    the scalar one, with and without padded channels -- a cast and a transpose, so bit-exact"""
    torch.manual_seed(5)
    B, (H, W) = 3, hw
    x = torch.randn(B, c, H, W)
    tdt = torch.float32 if dtype == L.F32 else torch.bfloat16
    out = torch.full((B, H, W, cs), 7.0, dtype=tdt, device="cuda")
    xd = x.cuda()
    L.call("msau_nchw_to_nhwc", torch.cuda.current_stream().cuda_stream, dtype, xd.data_ptr(), out.data_ptr(), B, c, cs, H, W)
    torch.cuda.synchronize()
    want = torch.zeros(B, H, W, cs, dtype=tdt)
    want[..., :c] = x.permute(0, 2, 3, 1).to(tdt)
    assert torch.equal(out.cpu(), want)


def _pack_image(w, c1, c2, rows=16):
    """OIHW weight [Co][c1 + c2][k][k] -> the packed image msau_pack_params writes: [chunk = source][row][k = tap * 8 + ci],
    k padded to a multiple of 32, bf16 (one 8-channel chunk per source)"""
    co, _, k, _ = w.shape
    kchunk = -(-k * k * 8 // 32) * 32
    nch = 1 if c2 == 0 else 2
    img = torch.zeros(nch, rows, kchunk)
    for s in range(nch):
        ws = w[:, s * 8:(s + 1) * 8]                                  # [co][8][k][k]
        img[s, :co, :k * k * 8] = ws.permute(0, 2, 3, 1).reshape(co, k * k * 8)
    return img.to(torch.bfloat16).reshape(-1).cuda()


@pytest.mark.parametrize("k,dual,flags", [(3, False, 0), (3, False, L.CONV_ACCUM), (3, False, L.CONV_LRN), (3, True, 0),
                                          (1, True, L.CONV_RELU_OUT), (4, False, 0), (4, False, L.CONV_MASK_B),
                                          (4, False, L.CONV_ACCUM | L.CONV_MASK_B)])
@pytest.mark.parametrize("hw,B", [((40, 64), 2), ((37, 45), 3), ((9, 130), 2), ((130, 140), 4)])
def test_row_streaming_single_convs_match_the_tile_kernels_and_torch(k, dual, flags, hw, B):
    """rowconv8_kernel (conv_rows.hip): the 8-channel level's single convolutions -- 3x3 (plain, accumulate, LRN second output),
    3x3 over concat(x1, x2), 1x1 over the concat + ReLU, 4x4 with the asymmetric SAME pad (plain, masked, accumulate + masked) --
    through msau_conv2d, against the tile kernels on the same packed weights and against torch on the rounded operands"""
    torch.manual_seed(5)
    H, W = hw
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    bf = lambda t: t.to(torch.bfloat16)
    nhwc = lambda t: bf(t).permute(0, 2, 3, 1).contiguous().cuda()
    x1, x2 = torch.randn(B, 8, H, W), torch.randn(B, 8, H, W)
    w = 0.2 * torch.randn(8, 16 if dual else 8, k, k)
    bias = 0.1 * torch.randn(8)
    yold, mb = torch.randn(B, 8, H, W), torch.randn(B, 8, H, W)
    d = L.ConvDesc()
    d.B, d.Hin, d.Win, d.Hout, d.Wout = B, H, W, H, W
    d.C1, d.C2, d.Cout, d.KH, d.KW, d.dil, d.stride, d.ups = 8, 8 if dual else 0, 8, k, k, 1, 1, 1
    d.pad_t = d.pad_l = {1: 0, 3: 1, 4: 1}[k]
    d.flags = flags
    X1, X2, Wp = nhwc(x1), nhwc(x2), _pack_image(w, 8, 8 if dual else 0)
    bias16 = torch.zeros(16); bias16[:8] = bias
    Bd, MB = bias16.cuda(), nhwc(mb)
    d.x1, d.x2, d.wpack, d.bias = X1.data_ptr(), (X2.data_ptr() if dual else None), Wp.data_ptr(), Bd.data_ptr()
    d.mask_b = MB.data_ptr() if flags & L.CONV_MASK_B else None
    d.lrn_alpha_over_n, d.lrn_beta, d.lrn_k = 1e-4 / 8, 0.75, 1.0
    out = {}
    try:
        for mode in ("1", "0"):
            os.environ["MSAU_CONV_ROWS"] = mode
            os.environ["MSAU_ROWS_MIN_TASKS"] = "1"
            lib.msau_reload_env()
            Y, Y2 = nhwc(yold), torch.zeros(B, H, W, 8, dtype=torch.bfloat16, device="cuda")
            d.y, d.y2 = Y.data_ptr(), (Y2.data_ptr() if flags & L.CONV_LRN else None)
            info = (L.i32 * 8)()
            L.check(lib.msau_conv2d_launch_info(L.BF16, C.byref(d), info))
            assert (info[6] == 3) == (mode == "1"), (mode, list(info))
            if mode == "0" and flags & L.CONV_LRN and not info[7] & 4:
                continue                          # small image: no tile instance carries the LRN epilogue
            L.check(lib.msau_conv2d(s, L.BF16, C.byref(d)))
            torch.cuda.synchronize()
            out[mode] = (Y.float().cpu(), Y2.float().cpu())
    finally:
        os.environ.pop("MSAU_CONV_ROWS", None)
        os.environ.pop("MSAU_ROWS_MIN_TASKS", None)
        lib.msau_reload_env()
    # torch on the rounded operands (SAME padding: 4x4 pads top/left 1, bottom/right 2 -- model/layers/utils.py:10-19)
    F = torch.nn.functional
    xin = torch.cat([bf(x1).float(), bf(x2).float()], 1) if dual else bf(x1).float()
    pad = {1: (0, 0, 0, 0), 3: (1, 1, 1, 1), 4: (1, 2, 1, 2)}[k]
    ref = F.conv2d(F.pad(xin, pad), bf(w).float(), bias)
    if flags & L.CONV_ACCUM:
        ref = ref + bf(yold).float()
    if flags & L.CONV_RELU_OUT:
        ref = ref.relu()
    if flags & L.CONV_MASK_B:
        ref = ref * (bf(mb).float() > 0)
    ref = ref.permute(0, 2, 3, 1)
    for mode in out:
        assert err(out[mode][0], ref, True) < 1e-2, mode
    if "0" in out:
        assert err(out["1"][0], out["0"][0], True) < 5e-3
    if flags & L.CONV_LRN:
        yr = out["1"][0].permute(0, 3, 1, 2)
        lrn = F.local_response_norm(yr, 8, alpha=1e-4, beta=0.75, k=1.0).permute(0, 2, 3, 1)
        assert err(out["1"][1], lrn, True) < 1e-2
        if "0" in out:
            assert err(out["1"][1], out["0"][1], True) < 5e-3


@pytest.mark.parametrize("k,flags,flags2", [(3, 0, 0), (3, L.CONV_ACCUM, 0), (1, 0, L.CONV_MASK_B),
                                            (3, L.CONV_ADD | L.CONV_ACCUM | L.CONV_MASK_B, L.CONV_ACCUM | L.CONV_MASK_B)])
@pytest.mark.parametrize("hw,B", [((40, 64), 2), ((37, 45), 3), ((9, 130), 2), ((130, 140), 4)])
def test_row_streaming_two_output_data_gradients_match_the_tile_kernels_and_torch(k, flags, flags2, hw, B):
    """rowconv8_kernel<.., MSAU_CONV_DOUT, flags2>: the data gradient of an 8-channel conv over concat(x1, x2) -- g [8] -> (dx1 [8],
    dx2 [8]) -- for the flag sets the reference's nets produce (3x3: plain / y accumulates; 1x1: y2 masked); the last case has
    no row instance and must stay on the tile kernel.  Through msau_conv2d, against the tile kernel on the same packed weights
    (where the image is large enough for it) and against torch on the rounded operands"""
    rows_have = (k, flags, flags2) in ((3, 0, 0), (3, L.CONV_ACCUM, 0), (1, 0, L.CONV_MASK_B))
    torch.manual_seed(7)
    H, W = hw
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    bf = lambda t: t.to(torch.bfloat16)
    nhwc = lambda t: bf(t).permute(0, 2, 3, 1).contiguous().cuda()
    g = torch.randn(B, 8, H, W)
    w = 0.2 * torch.randn(16, 8, k, k)
    add, y1old, y2old, mb1, mb2 = (torch.randn(B, 8, H, W) for _ in range(5))
    d = L.ConvDesc()
    d.B, d.Hin, d.Win, d.Hout, d.Wout = B, H, W, H, W
    d.C1, d.C2, d.Cout, d.KH, d.KW, d.dil, d.stride, d.ups = 8, 0, 16, k, k, 1, 1, 1
    d.pad_t = d.pad_l = k // 2
    d.flags, d.flags2 = flags | L.CONV_DOUT, flags2
    G, Wp, AD, MB1, MB2 = nhwc(g), _pack_image(w, 8, 0), nhwc(add), nhwc(mb1), nhwc(mb2)
    d.x1, d.wpack, d.bias = G.data_ptr(), Wp.data_ptr(), None
    d.add = AD.data_ptr() if flags & L.CONV_ADD else None
    d.mask_b = MB1.data_ptr() if flags & L.CONV_MASK_B else None
    d.mask_b2 = MB2.data_ptr() if flags2 & L.CONV_MASK_B else None
    out = {}
    try:
        for mode in ("1", "0"):
            os.environ["MSAU_DOUT_ROWS"] = mode
            os.environ["MSAU_ROWS_MIN_TASKS"] = "1"
            lib.msau_reload_env()
            Y1, Y2 = nhwc(y1old), nhwc(y2old)
            d.y, d.y2 = Y1.data_ptr(), Y2.data_ptr()
            info = (L.i32 * 8)()
            L.check(lib.msau_conv2d_launch_info(L.BF16, C.byref(d), info))
            assert (info[6] == 3) == (mode == "1" and rows_have), (mode, list(info))
            if not info[7] & 2:
                continue                          # small image and no row instance: the caller issues one launch per output
            L.check(lib.msau_conv2d(s, L.BF16, C.byref(d)))
            torch.cuda.synchronize()
            out[mode] = (Y1.float().cpu(), Y2.float().cpu())
    finally:
        os.environ.pop("MSAU_DOUT_ROWS", None)
        os.environ.pop("MSAU_ROWS_MIN_TASKS", None)
        lib.msau_reload_env()
    F = torch.nn.functional
    both = F.conv2d(bf(g).float(), bf(w).float(), None, padding=k // 2)
    r1, r2 = both[:, :8], both[:, 8:]
    if flags & L.CONV_ADD:
        r1 = r1 + bf(add).float()
    if flags & L.CONV_ACCUM:
        r1 = r1 + bf(y1old).float()
    if flags & L.CONV_MASK_B:
        r1 = r1 * (bf(mb1).float() > 0)
    if flags2 & L.CONV_ACCUM:
        r2 = r2 + bf(y2old).float()
    if flags2 & L.CONV_MASK_B:
        r2 = r2 * (bf(mb2).float() > 0)
    for mode in out:
        assert err(out[mode][0], r1.permute(0, 2, 3, 1), True) < 1e-2, mode
        assert err(out[mode][1], r2.permute(0, 2, 3, 1), True) < 1e-2, mode
    assert rows_have <= ("1" in out)
    if len(out) == 2:
        assert err(out["1"][0], out["0"][0], True) < 5e-3 and err(out["1"][1], out["0"][1], True) < 5e-3


@pytest.mark.parametrize("B,Cr,hw,relu", [(2, 64, (24, 32), 0), (3, 61, (17, 100), 1), (1, 64, (9, 16), 0), (2, 64, (40, 256), 1),
                                          (2, 5, (33, 288), 0), (1, 64, (8, 132), 0), (5, 64, (50, 60), 1), (16, 64, (336, 256), 0)])
def test_first_conv_fed_with_the_nchw_tensor_is_bit_identical_to_conversion_plus_conv(B, Cr, hw, relu):
    """MSAU_CONV_NCHW (csrc/conv_first.hip): the net's first conv reads the API's fp32 NCHW tensor itself and writes the NHWC bf16
    copy the weight gradient needs -- same products, order and roundings as msau_nchw_to_nhwc followed by the dense launch.
    Widths that are no multiple of 16 / 32 / 64, fewer real channels than the 64 stored, bands with a ragged last one, the
    bench's own size; with and without the copy."""
    H, W = hw
    dev = torch.device("cuda")
    s = torch.cuda.current_stream().cuda_stream
    lib = L.load()
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + H + W)
    x = (torch.randn(B, Cr, H, W, generator=g) * (torch.rand(B, 1, H, W, generator=g) > 0.5)).to(dev)
    w = (torch.randn(16 * 576, generator=g) * 0.1).to(dev).to(torch.bfloat16)
    w.view(16, 576)[8:] = 0                                         # rows beyond the 8 output channels, as msau_pack_params writes them
    bias = torch.randn(16, generator=g).to(dev)
    xn = torch.zeros(B, H, W, 64, device=dev, dtype=torch.bfloat16)
    y_ref = torch.zeros(B, H, W, 8, device=dev, dtype=torch.bfloat16)
    L.call("msau_nchw_to_nhwc", s, L.BF16, x.data_ptr(), xn.data_ptr(), B, Cr, 64, H, W)
    d = L.ConvDesc()
    d.B, d.Hin, d.Win, d.Hout, d.Wout = B, H, W, H, W
    d.C1, d.C2, d.Cout = 64, 0, 8
    d.KH = d.KW = 3
    d.dil, d.pad_t, d.pad_l, d.stride, d.ups = 1, 1, 1, 1, 1
    d.flags = L.CONV_RELU_OUT if relu else 0
    d.x1, d.wpack, d.bias, d.y = xn.data_ptr(), w.data_ptr(), bias.data_ptr(), y_ref.data_ptr()
    L.check(lib.msau_conv2d(s, L.BF16, d), "conv")
    info = (L.i32 * 8)()
    L.call("msau_conv2d_launch_info", L.BF16, C.byref(d), info)
    assert info[7] & 64, list(info)
    xn2, y2 = torch.full_like(xn, 7.0), torch.full_like(y_ref, 7.0)
    d.flags |= L.CONV_NCHW
    d.x1, d.y, d.y2, d.head_classes = x.data_ptr(), y2.data_ptr(), xn2.data_ptr(), Cr
    L.check(lib.msau_conv2d(s, L.BF16, d), "first conv, NCHW input")
    torch.cuda.synchronize()
    assert torch.equal(y2, y_ref) and torch.equal(xn2, xn)
    y3 = torch.full_like(y_ref, 7.0)
    d.y, d.y2 = y3.data_ptr(), None                                  # forward-only plans: no copy
    L.check(lib.msau_conv2d(s, L.BF16, d), "first conv, NCHW input, no copy")
    torch.cuda.synchronize()
    assert torch.equal(y3, y_ref)


def test_first_conv_nchw_refuses_what_it_does_not_implement():
    lib = L.load()
    d = L.ConvDesc()
    d.B, d.Hin, d.Win, d.Hout, d.Wout = 1, 16, 30, 16, 30                # width no multiple of 4
    d.C1, d.C2, d.Cout, d.KH, d.KW = 64, 0, 8, 3, 3
    d.dil, d.pad_t, d.pad_l, d.stride, d.ups = 1, 1, 1, 1, 1
    info = (L.i32 * 8)()
    L.call("msau_conv2d_launch_info", L.BF16, C.byref(d), info)
    assert not info[7] & 64
    d.Win = d.Wout = 32
    L.call("msau_conv2d_launch_info", L.BF16, C.byref(d), info)
    assert info[7] & 64
    L.call("msau_conv2d_launch_info", L.F32, C.byref(d), info)       # bf16 storage only
    assert not info[7] & 64
    d.Win = d.Wout = 304                                             # the three-row ring would not fit the LDS
    L.call("msau_conv2d_launch_info", L.BF16, C.byref(d), info)
    assert not info[7] & 64
    x = torch.zeros(8, device="cuda")
    d.flags = L.CONV_NCHW
    d.x1 = d.wpack = d.y = x.data_ptr()
    assert lib.msau_conv2d(torch.cuda.current_stream().cuda_stream, L.BF16, d) != 0       # refused, not launched


@pytest.mark.parametrize("c,hw,B,pool", [(8, (57, 61), 3, False), (8, (64, 90), 2, True), (8, (71, 250), 2, True), (8, (33, 31), 4, False),
                                          (16, (60, 129), 2, False), (16, (49, 70), 3, True), (16, (40, 15), 2, True), (16, (16, 28), 4, False),
                                          (32, (84, 64), 2, True), (32, (29, 33), 3, False), (32, (21, 16), 2, True), (32, (14, 14), 1, False)])
def test_coupling_conv_in_the_residual_pairs_forward_launch(monkeypatch, c, hw, B, pool):
    """MSAU_PAIR_COUPLE (round 5): z = ReLU(conv1x1(concat(prev, y))) of a coupled stage (model/model.py:143-148,246-252) computed by
    the row-streaming pair's forward launch from the finished row of y, optionally with the zero-padded 2x2 max pool of z
    (model/model.py:158-160), against the stand-alone launches (MSAU_PAIR_COUPLE=0, pool as its own launch).  8 channels: the
    stand-alone coupling launch is rowconv8_kernel<2, 1, 1> -- the same single MFMA with the same k order -- so EVERYTHING is
    bit-identical: y, z, pooled z, positions (through the pool's gradient), every gradient.  16 channels: the stand-alone launch is a
    tile kernel that sums the two sources in two MFMAs and adds the bias last, so z may differ in the last bit of a few elements:
    <= 1 bf16 ulp everywhere, bit-equal in > 99.5 %, gradients within bf16 noise.  Both against torch on the CPU."""
    from msau_amd.plan import PairOp
    torch.manual_seed(21)
    H, W = hw
    x = torch.randn(B, c, H, W)
    p = {"w": 0.2 * torch.randn(c, c, 3, 3), "b": 0.1 * torch.randn(c), "w2": 0.2 * torch.randn(c, c, 3, 3), "b2": 0.1 * torch.randn(c),
         "w0": 0.3 * torch.randn(c, c, 1, 1), "b0": 0.1 * torch.randn(c), "wc": 0.25 * torch.randn(c, 2 * c, 1, 1), "bc": 0.1 * torch.randn(c)}
    Ho, Wo = ((H + 1) // 2, (W + 1) // 2) if pool else (H, W)
    gy = torch.randn(B, c, Ho, Wo)
    seen = []

    def build(plan):
        x0 = plan.x_in
        prev = Act(plan, "prev", H, W, c, relu_out=True)
        ConvOp(plan, "c0", x0, None, "w0", "b0", prev, 1, relu_out=True)
        r1 = Act(plan, "r1", H, W, c, relu_out=True)
        c1 = ConvOp(plan, "c1", x0, None, "w", "b", r1, 3, relu_in=True, relu_out=True)
        out = Act(plan, "out", H, W, c, relu_out=True)
        c2 = ConvOp(plan, "c2", r1, None, "w2", "b2", out, 3, relu_out=True, fwd_add=x0)
        c1.bwd_add = out
        PairOp(plan, c1, c2)
        z = Act(plan, "z", H, W, c, relu_out=True)
        cp = ConvOp(plan, "cpl", prev, out, "wc", "bc", z, 1, relu_out=True)
        if pool:
            q = Act(plan, "q", Ho, Wo, c)
            PoolOp(plan, "p", z, q)
            plan.logits = q
        else:
            plan.logits = z
        seen.append((plan, cp, z, out))
    monkeypatch.setenv("MSAU_ROWS_MIN_TASKS", "1")
    res, mids = {}, {}
    try:
        for mode in ("1", "0"):
            monkeypatch.setenv("MSAU_PAIR_COUPLE", mode)
            monkeypatch.setenv("MSAU_FUSE_POOL", mode)               # stand-alone mode: the pool as its own launch (bit-exact by its own test)
            L.load().msau_reload_env()
            res[mode] = run_graph(build, p, x, gy, L.BF16)
            plan, cp, z, out = seen[-1]
            mids[mode] = (z.data.clone(), out.data.clone())
    finally:
        monkeypatch.undo()
        L.load().msau_reload_env()
    (plan1, cp1, _, _), (plan0, cp0, _, _) = seen
    assert plan1.pairs[0].active and cp1.cpl_fused_into is plan1.pairs[0] and cp0.cpl_fused_into is None
    # (32 channels: the tile pair conv_pair_kernel<C32> carries the rider -- the epilogue's result layout is the 1x1 conv's B fragment)
    assert plan1.pairs[0].key.startswith("rowpair_kernel" if c < 32 else "conv_pair_kernel") and plan0.pairs[0].key == plan1.pairs[0].key
    if pool:
        assert any(isinstance(o, PoolOp) and o.fused_into is plan1.pairs[0] for o in plan1.ops)
    assert torch.equal(mids["1"][1], mids["0"][1]), "the block's own output"
    z1, z0 = mids["1"][0].float(), mids["0"][0].float()
    if c == 8:
        assert torch.equal(z1, z0), "z"
        for a, b in zip(res["1"], res["0"]):
            if isinstance(a, dict):
                for n in a:
                    assert torch.equal(a[n], b[n]), n
            elif a is not None:
                assert torch.equal(a, b)
    else:
        ulp = torch.pow(2.0, torch.floor(torch.log2(torch.maximum(z1.abs(), z0.abs()).clamp_min(2.0 ** -120))) - 7)
        assert bool(((z1 - z0).abs() <= ulp).all()), float(((z1 - z0).abs() / ulp).max())
        assert float((z1 != z0).float().mean()) < 5e-3
        assert err(res["1"][0], res["0"][0], True) < 1e-2
        assert err(res["1"][2], res["0"][2], True) < 3e-2
        for n in res["1"][3]:
            assert err(res["1"][3][n], res["0"][3][n], True) < 3e-2, n
    # against torch (fp32, CPU)
    prev_r = O.conv_same(x, p["w0"], p["b0"], relu=True)
    r = O.conv_same(torch.relu(x), p["w"], p["b"], relu=True)
    out_r = torch.relu(O.conv_same(r, p["w2"], p["b2"]) + x)
    z_r = O.conv_same(torch.cat([prev_r, out_r], 1), p["wc"], p["bc"], relu=True)
    y_r = torch.nn.functional.max_pool2d(torch.nn.functional.pad(z_r, (0, W % 2, 0, H % 2)), 2, 2) if pool else z_r
    assert err(res["1"][0], y_r, True) < 3e-2


@pytest.mark.parametrize("cin", [16, 32])
@pytest.mark.parametrize("hw_in,odd,B", [((20, 32), (0, 0), 2), ((21, 19), (1, 1), 3), ((9, 70), (1, 0), 2), ((65, 33), (0, 1), 4), ((168, 128), (0, 0), 2)])
def test_row_streaming_transposed_conv_matches_the_zero_stuffed_launch_and_torch(monkeypatch, hw_in, odd, B, cin):
    """rowdeconv8_kernel / rowdeconv16_kernel (conv_rows.hip, round 5): the level-1 -> level-0 and level-2 -> level-1 transposed convs
    (16 -> 8 and 32 -> 16 channels, k 3, stride 2, padding 1,
    output_size even or odd per axis: model/model.py:230, model/layers/layers.py:249-250) computed from the live taps only, against the
    tile kernel's conv over the zero-stuffed input (MSAU_DECONV_ROWS=0: same packed image, other summation order) and against
    torch.nn.functional.conv_transpose2d on the rounded operands; the backward (strided data gradient, weight gradient) is untouched
    and must give the same bits in both modes."""
    torch.manual_seed(31)
    Hi, Wi = hw_in
    Ho, Wo = 2 * Hi - odd[0], 2 * Wi - odd[1]
    cout = cin // 2
    x = torch.randn(B, cin, Hi, Wi)
    p = {"w": 0.2 * torch.randn(cin, cout, 3, 3), "b": 0.1 * torch.randn(cout)}     # ConvTranspose2d weight: [Cin][Cout][k][k]
    gy = torch.randn(B, cout, Ho, Wo)
    seen = []

    def build(plan):
        y = Act(plan, "y", Ho, Wo, cout)
        seen.append(ConvOp(plan, "dc", plan.x_in, None, "w", "b", y, 3, kind="deconv"))
        plan.logits = y
    monkeypatch.setenv("MSAU_ROWS_MIN_TASKS", "1")
    res = {}
    try:
        for mode in ("1", "0"):
            monkeypatch.setenv("MSAU_DECONV_ROWS", "2" if mode == "1" else "0")
            L.load().msau_reload_env()
            res[mode] = run_graph(build, p, x, gy, L.BF16)
    finally:
        monkeypatch.undo()
        L.load().msau_reload_env()
    assert seen[0].fkey.startswith("rowconv_kernel") and "ups2" in seen[0].fkey and not seen[1].fkey.startswith("rowconv_kernel"), (seen[0].fkey, seen[1].fkey)
    bf = lambda t: t.to(torch.bfloat16).float()
    ref = torch.nn.functional.conv_transpose2d(bf(x), bf(p["w"]), p["b"], stride=2, padding=1, output_padding=(1 - odd[0], 1 - odd[1]))
    assert tuple(ref.shape[2:]) == (Ho, Wo)
    for mode in ("1", "0"):
        assert err(res[mode][0], ref, True) < 1e-2, mode
    a, b = res["1"][0], res["0"][0]
    ulp = torch.pow(2.0, torch.floor(torch.log2(torch.maximum(a.abs(), b.abs()).clamp_min(2.0 ** -120))) - 7)
    # one bf16 ulp (a sum that lands on the other side of a rounding boundary) + the fp32 order noise of a result that cancels to ~0
    assert bool(((a - b).abs() <= ulp + 1e-5 * float(ref.abs().max())).all()) and float((a != b).float().mean()) < 2e-2, float((a != b).float().mean())
    assert torch.equal(res["1"][2], res["0"][2]), "input gradient"
    for n in res["1"][3]:
        assert torch.equal(res["1"][3][n], res["0"][3][n]), n


@pytest.mark.parametrize("hw,B", [((64, 128), 4), ((97, 75), 6), ((130, 140), 4)])      # (sizes at which the plan fuses the two data gradients into one launch)
def test_coupling_conv_weight_gradient_in_its_data_gradient_launch(monkeypatch, hw, B):
    """MSAU_CONV_WGRAD (round 5): the weight / bias gradient of the 8-channel coupling conv z = ReLU(Wc concat(prev, y) + bc)
    (model/model.py:143-148) computed by its own two-output data-gradient launch (rowconv8_kernel<1, 1, 1, DOUT, MASK_B, WG>) against
    the stand-alone weight-gradient launch (MSAU_COUPLE_WGRAD=0): every data gradient and every other parameter gradient bit for
    bit, wc / bc within the fp32 summation-order bound, and against torch autograd on the rounded operands."""
    torch.manual_seed(41)
    H, W = hw
    c = 8
    x = torch.randn(B, c, H, W)
    p = {"w0": 0.3 * torch.randn(c, c, 1, 1), "b0": 0.1 * torch.randn(c), "w1": 0.2 * torch.randn(c, c, 3, 3), "b1": 0.1 * torch.randn(c),
         "wc": 0.25 * torch.randn(c, 2 * c, 1, 1), "bc": 0.1 * torch.randn(c)}
    gy = torch.randn(B, c, H, W)
    seen = []

    def build(plan):
        x0 = plan.x_in
        # (prev without a ReLU of its own: in the net the previous stage's tensor has later contributions, so this launch writes
        #  its gradient un-masked -- the flag set the rider's instance is compiled for)
        prev = Act(plan, "prev", H, W, c)
        ConvOp(plan, "c0", x0, None, "w0", "b0", prev, 1)
        cur = Act(plan, "cur", H, W, c, relu_out=True)
        ConvOp(plan, "c1", x0, None, "w1", "b1", cur, 3, relu_out=True)
        z = Act(plan, "z", H, W, c, relu_out=True)
        seen.append(ConvOp(plan, "cpl", prev, cur, "wc", "bc", z, 1, relu_out=True))
        plan.logits = z
    monkeypatch.setenv("MSAU_ROWS_MIN_TASKS", "1")
    res = {}
    try:
        for mode in ("1", "0"):
            monkeypatch.setenv("MSAU_COUPLE_WGRAD", mode)
            L.load().msau_reload_env()
            res[mode] = run_graph(build, p, x, gy, L.BF16)
    finally:
        monkeypatch.undo()
        L.load().msau_reload_env()
    assert seen[0].wg_fused and not seen[1].wg_fused
    assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][2], res["0"][2])
    for n in res["1"][3]:
        if n in ("wc", "bc"):
            d = (res["1"][3][n] - res["0"][3][n]).abs().max()
            assert float(d) <= 2e-5 * float(res["0"][3][n].abs().max()) * (H * W * B) ** 0.5 + 1e-6, (n, float(d))
        else:
            assert torch.equal(res["1"][3][n], res["0"][3][n]), n
    # torch autograd on bf16-rounded operands (the stored activations are bf16; accumulation fp32)
    bf = lambda t: t.to(torch.bfloat16).float()
    prev_r = bf(O.conv_same(bf(x), bf(p["w0"]), p["b0"]))
    cur_r = bf(O.conv_same(bf(x), bf(p["w1"]), p["b1"], relu=True))
    wc = bf(p["wc"]).requires_grad_(True)
    bc = p["bc"].clone().requires_grad_(True)
    z_r = O.conv_same(torch.cat([prev_r, cur_r], 1), wc, bc, relu=True)
    gz = bf(bf(gy) * (bf(z_r.detach()) > 0))                  # the gradient tensor the launch reads: masked, stored in bf16
    O.conv_same(torch.cat([prev_r, cur_r], 1), wc, bc).backward(gz)
    assert err(res["1"][3]["wc"], wc.grad, False) < 2e-3 and err(res["1"][3]["bc"], bc.grad, False) < 2e-3


@pytest.mark.parametrize("hw,B,tail", [((21, 19), 3, False), ((42, 32), 2, True), ((16, 16), 1, True)])
def test_attention_projection_data_gradients_in_one_launch(monkeypatch, hw, B, tail):
    """msau_attn_proj_bwd (round 5): d(x) of the attention block's f, g, h projections (model/layers/attention.py:152-154) + the
    residual's dy + x's ReLU mask in ONE launch instead of three accumulating msau_conv2d launches (MSAU_ATTN_PROJ_FUSE=0).  The
    fused launch rounds once where the three round three times, so the two agree to a few bf16 ulps, not bit for bit; the sharp check
    is against float64 on the launch's own stored operands (half a bf16 ulp + fp32 accumulation).  `tail`: x has a later reader, so
    the launch accumulates onto an earlier contribution (the net's case: the decoder's transposed conv reads the same tensor)."""
    from msau_amd.plan import AttnCoreOp, ProjBwd
    torch.manual_seed(43)
    H, W = hw
    c = 64
    x = torch.randn(B, c, H, W)
    p = {"w0": 0.06 * torch.randn(c, c, 3, 3), "b0": 0.1 * torch.randn(c),
         "wf": 0.15 * torch.randn(8, c, 1, 1), "bf": 0.1 * torch.randn(8), "wg": 0.15 * torch.randn(8, c, 1, 1), "bg": 0.1 * torch.randn(8),
         "wh": 0.12 * torch.randn(c, c, 1, 1), "bh": 0.1 * torch.randn(c),
         "wt": 0.12 * torch.randn(c, c, 1, 1), "bt": 0.1 * torch.randn(c), "wz": 0.1 * torch.randn(8, 2 * c, 1, 1), "bz": 0.1 * torch.randn(8)}
    if not tail:
        p["wz"] = 0.1 * torch.randn(8, c, 1, 1)
        del p["wt"], p["bt"]
    gy = torch.randn(B, 8, H, W)
    keep = {}

    def build(plan):
        x2 = Act(plan, "x2", H, W, c, relu_out=True)
        ConvOp(plan, "c0", plan.x_in, None, "w0", "b0", x2, 3, relu_out=True)
        fa, ga, ha = Act(plan, "f", H, W, 8), Act(plan, "g", H, W, 8), Act(plan, "h", H, W, c)
        fop = ConvOp(plan, "f", x2, None, "wf", "bf", fa, 1)
        gop = ConvOp(plan, "g", x2, None, "wg", "bg", ga, 1)
        hop = ConvOp(plan, "h", x2, None, "wh", "bh", ha, 1)
        pj = ProjBwd(plan, fop, gop, hop)
        y = Act(plan, "y", H, W, c)
        AttnCoreOp(plan, "attn", fa, ga, ha, x2, y)
        hop.bwd_add = y
        z = Act(plan, "z", H, W, 8)
        if tail:
            t = Act(plan, "t", H, W, c)
            ConvOp(plan, "t", x2, None, "wt", "bt", t, 1)
            ConvOp(plan, "z", y, t, "wz", "bz", z, 1)
        else:
            ConvOp(plan, "z", y, None, "wz", "bz", z, 1)
        plan.logits = z
        keep[os.environ["MSAU_ATTN_PROJ_FUSE"]] = (pj, x2, fa, ga, ha, y)
    res = {}
    try:
        for mode in ("1", "0"):
            monkeypatch.setenv("MSAU_ATTN_PROJ_FUSE", mode)
            res[mode] = run_graph(build, p, x, gy, L.BF16)
    finally:
        monkeypatch.undo()
    assert keep["1"][0].active and not keep["0"][0].active
    assert bool(keep["1"][0].args.accumulate) == tail and keep["1"][0].args.add and keep["1"][0].args.mask_b
    assert torch.equal(res["1"][0], res["0"][0])                                   # (the forward is the same launches)
    # fused against the three launches: a few bf16 ulps on d(x2), carried through c0's data / weight gradient
    assert err(res["1"][2], res["0"][2], True) < 6e-3
    for n in res["1"][3]:
        if n in ("w0", "b0"):
            assert err(res["1"][3][n], res["0"][3][n], True) < 6e-3, n
        else:
            assert torch.equal(res["1"][3][n], res["0"][3][n]), n
    if not tail:
        # float64 on the stored operands of the launch: d(x2) = (Wf^T df + Wg^T dg + Wh^T dh + dy) * (x2 > 0)
        pj, x2, fa, ga, ha, y = keep["1"]
        bfw = lambda k: p[k].to(torch.bfloat16).double().view(p[k].shape[0], c)
        d = lambda a: a.grad.double().cpu().view(-1, a.Cs)
        terms = [d(fa) @ bfw("wf"), d(ga) @ bfw("wg"), d(ha) @ bfw("wh"), d(y)]
        S = d(fa).abs() @ bfw("wf").abs() + d(ga).abs() @ bfw("wg").abs() + d(ha).abs() @ bfw("wh").abs() + d(y).abs()
        mask = (x2.data.double().cpu().view(-1, c) > 0)
        ref = sum(terms) * mask
        got = x2.grad.double().cpu().view(-1, c)
        ulp = torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 7)
        excess = (got - ref).abs() - (0.5 * ulp * (1 + 1e-6) + 3e-6 * S + 1e-30)
        assert float(excess.max()) <= 0, float(excess.max())
        assert float((got != 0).double().mean()) > 0.2


@pytest.mark.parametrize("hw,B,later", [((21, 19), 3, False), ((42, 32), 2, True), ((96, 96), 8, False)])   # (the last: more tiles than the grid has waves)
def test_both_data_gradients_of_the_64_channel_coupling_conv_in_one_launch(monkeypatch, hw, B, later):
    """msau_dgrad2_1x1 (round 5): d(prev), d(y) of z = ReLU(Wc concat(prev, y) + bc) (model/model.py:143-148) at 64 + 64 channels in
    one launch against the two msau_conv2d launches (MSAU_DGRAD2=0): the same rounded weights, the same k order per output, one
    rounding each -> bit for bit, masks and accumulation included.  `later`: both sources have a later reader (their gradient
    buffers already hold a contribution: the launch accumulates)."""
    torch.manual_seed(47)
    H, W = hw
    c = 64
    x = torch.randn(B, c, H, W)
    p = {"w0": 0.12 * torch.randn(c, c, 1, 1), "b0": 0.1 * torch.randn(c), "w1": 0.05 * torch.randn(c, c, 3, 3), "b1": 0.1 * torch.randn(c),
         "wc": 0.1 * torch.randn(c, 2 * c, 1, 1), "bc": 0.1 * torch.randn(c), "wz": 0.1 * torch.randn(8, (3 if later else 1) * c, 1, 1), "bz": 0.1 * torch.randn(8)}
    gy = torch.randn(B, 8, H, W)
    seen = []

    def build(plan):
        prev = Act(plan, "prev", H, W, c)
        ConvOp(plan, "c0", plan.x_in, None, "w0", "b0", prev, 1)
        cur = Act(plan, "cur", H, W, c, relu_out=True)
        ConvOp(plan, "c1", plan.x_in, None, "w1", "b1", cur, 3, relu_out=True)
        z = Act(plan, "z", H, W, c, relu_out=True)
        seen.append(ConvOp(plan, "cpl", prev, cur, "wc", "bc", z, 1, relu_out=True))
        out = Act(plan, "out", H, W, 8)
        if later:
            cat = Act(plan, "pc", H, W, 2 * c)                   # a later reader of prev and cur: 128 -> ... via two more convs
            ConvOp(plan, "pc", prev, cur, "wpc", "bpc", cat, 1)
            ConvOp(plan, "zc", z, cat, "wz", "bz", out, 1)
        else:
            ConvOp(plan, "zc", z, None, "wz", "bz", out, 1)
        plan.logits = out
    if later:
        p["wpc"], p["bpc"] = 0.1 * torch.randn(2 * c, 2 * c, 1, 1), 0.1 * torch.randn(2 * c)
    res = {}
    try:
        for mode in ("1", "0"):
            monkeypatch.setenv("MSAU_DGRAD2", mode)
            res[mode] = run_graph(build, p, x, gy, L.BF16)
    finally:
        monkeypatch.undo()
    assert seen[0].dgrad2 is not None and seen[1].dgrad2 is None
    assert bool(seen[0].dgrad2.accumulate1) == later and bool(seen[0].dgrad2.accumulate2) == later and seen[0].dgrad2.mask2 and not seen[0].dgrad2.mask1
    assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][2], res["0"][2])
    for n in res["1"][3]:
        assert torch.equal(res["1"][3][n], res["0"][3][n]), n
    # and against torch autograd in fp32 (bf16 storage: relative L2)
    xs = x.clone().requires_grad_(True)
    ps = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    prev = O.conv_same(xs, ps["w0"], ps["b0"])
    cur = O.conv_same(xs, ps["w1"], ps["b1"], relu=True)
    z = O.conv_same(torch.cat([prev, cur], 1), ps["wc"], ps["bc"], relu=True)
    if later:
        out = O.conv_same(torch.cat([z, O.conv_same(torch.cat([prev, cur], 1), ps["wpc"], ps["bpc"])], 1), ps["wz"], ps["bz"])
    else:
        out = O.conv_same(z, ps["wz"], ps["bz"])
    out.backward(gy)
    # (a sanity bound: two ReLU layers in bf16 storage against fp32 -- the sharp statement is the bit equality above)
    assert err(res["1"][2], xs.grad, True) < 1e-1
    for n in ("w0", "w1", "wc"):
        assert err(res["1"][3][n], ps[n].grad, True) < 1e-1, n


@pytest.mark.parametrize("hw,B,pool", [((84, 64), 2, True), ((29, 33), 3, False), ((21, 16), 2, True), ((14, 14), 1, False), ((45, 100), 2, False),
                                        ((45, 100), 6, False), ((84, 64), 8, True)])       # (the last two: sizes at which the stand-alone form is ONE two-output launch)
def test_coupling_conv_data_gradients_as_the_prologue_of_the_pairs_backward_launch(monkeypatch, hw, B, pool):
    """MSAU_PAIR_DCOUPLE (round 5): the two-output data gradient of the coupling conv z = ReLU(Wc concat(prev, y) + bc)
    (model/model.py:143-148,246-252) at 32 channels as the prologue of the residual pair's backward launch -- the launch reads d(z),
    computes g = d(y) = (Wc[:, y]^T d(z)) . [y > 0] for its tile incl. the halo in LDS, writes g and d(prev) of its own pixels --
    against the stand-alone MSAU_CONV_DOUT launch (MSAU_PAIR_DCOUPLE=0): one k-step with the same rounded weights either way, so
    every gradient is bit-identical."""
    from msau_amd.plan import PairOp
    torch.manual_seed(23)
    H, W = hw
    c = 32
    x = torch.randn(B, c, H, W)
    p = {"w": 0.1 * torch.randn(c, c, 3, 3), "b": 0.1 * torch.randn(c), "w2": 0.1 * torch.randn(c, c, 3, 3), "b2": 0.1 * torch.randn(c),
         "w0": 0.2 * torch.randn(c, c, 1, 1), "b0": 0.1 * torch.randn(c), "wc": 0.15 * torch.randn(c, 2 * c, 1, 1), "bc": 0.1 * torch.randn(c)}
    Ho, Wo = ((H + 1) // 2, (W + 1) // 2) if pool else (H, W)
    gy = torch.randn(B, c, Ho, Wo)
    seen = []

    def build(plan):
        x0 = plan.x_in
        prev = Act(plan, "prev", H, W, c)                   # (no ReLU of its own: this launch then writes its gradient un-masked, as in the net)
        ConvOp(plan, "c0", x0, None, "w0", "b0", prev, 1)
        r1 = Act(plan, "r1", H, W, c, relu_out=True)
        c1 = ConvOp(plan, "c1", x0, None, "w", "b", r1, 3, relu_in=True, relu_out=True)
        out = Act(plan, "out", H, W, c, relu_out=True)
        c2 = ConvOp(plan, "c2", r1, None, "w2", "b2", out, 3, relu_out=True, fwd_add=x0)
        c1.bwd_add = out
        PairOp(plan, c1, c2)
        z = Act(plan, "z", H, W, c, relu_out=True)
        cp = ConvOp(plan, "cpl", prev, out, "wc", "bc", z, 1, relu_out=True)
        if pool:
            q = Act(plan, "q", Ho, Wo, c)
            PoolOp(plan, "p", z, q)
            plan.logits = q
        else:
            plan.logits = z
        seen.append((plan, cp, out, prev))
    res, grads = {}, {}
    try:
        for mode in ("1", "0"):
            monkeypatch.setenv("MSAU_PAIR_DCOUPLE", mode)
            res[mode] = run_graph(build, p, x, gy, L.BF16)
            plan, cp, out, prev = seen[-1]
            grads[mode] = (out.grad.clone(), prev.grad.clone())
    finally:
        monkeypatch.undo()
    (plan1, cp1, _, _), (plan0, cp0, _, _) = seen
    assert plan1.pairs[0].dcp is cp1 and cp1.dgrad_in_pair is plan1.pairs[0] and plan0.pairs[0].dcp is None and cp0.dgrad_in_pair is None
    assert plan1.pairs[0].key.startswith("conv_pair_kernel")
    assert torch.equal(grads["1"][0], grads["0"][0]), "d(y)"
    assert torch.equal(grads["1"][1], grads["0"][1]), "d(prev)"
    for a, b in zip(res["1"], res["0"]):
        if isinstance(a, dict):
            for n in a:
                assert torch.equal(a[n], b[n]), n
        elif a is not None:
            assert torch.equal(a, b)
    assert float((grads["1"][0] != 0).float().mean()) > 0.1 and float((grads["1"][1] != 0).float().mean()) > 0.5
