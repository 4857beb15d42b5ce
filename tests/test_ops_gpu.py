"""GPU: every HIP kernel, called through the C ABI, against the reference's golden vectors
(tests/golden/ops.npz) and against the CPU oracle on seeded inputs.  fp32 storage: tolerance 1e-4
relative to the tensor's max (fp32 MFMA sums in a different order than MKL-DNN); bf16 storage: 3e-2."""
import numpy as np
import pytest
import torch

from msau_amd import _lib as L
from oracle import msau_oracle as O
from tests.golden_util import err, load_ops, rel_err
from tests.hip_harness import Act, AttnCoreOp, ConvOp, LrnOp, PoolOp, run_graph

pytestmark = pytest.mark.gpu

TOL = {L.F32: 1e-4, L.BF16: 3e-2}
# bf16 gradients through a ReLU: ~0.4 % of the masks flip within bf16 rounding -> ~6 % relative L2
TOLG = {L.F32: 1e-4, L.BF16: 1e-1}
DT = [pytest.param(L.F32, id="f32"), pytest.param(L.BF16, id="bf16")]


def _conv_builder(co, k, relu, dil=1, lrn=False, kind="conv", out_hw=None):
    def build(plan):
        x = plan.x_in
        H, W = out_hw or (x.H, x.W)
        if lrn:
            a = Act(plan, "a", H, W, co)
            ConvOp(plan, "c", x, None, "w", "b", a, k, dil=dil)
            y = Act(plan, "y", H, W, co)
            LrnOp(plan, "l", a, y)
        else:
            y = Act(plan, "y", H, W, co, relu_out=relu)
            ConvOp(plan, "c", x, None, "w", "b", y, k, dil=dil, relu_out=relu, kind=kind)
        plan.logits = y
    return build


def _check(g, tag, y, dx, grads, wkey, bkey, tol):
    bf = tol > 1e-3
    tg = TOLG[L.BF16 if bf else L.F32]
    assert err(y, g[f"{tag}.y"], bf) < tol, "fwd"
    assert err(dx, g[f"{tag}.gx"], bf) < tg, "dgrad"
    assert err(grads["w"], g[f"{tag}.g.{wkey}"], bf) < tg, "wgrad"
    assert err(grads["b"], g[f"{tag}.g.{bkey}"], bf) < tg, "bgrad"


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("tag,k,relu", [("conv3", 3, True), ("conv3lin", 3, False), ("conv1", 1, True),
                                        ("conv4", 4, False), ("conv3c13", 3, False)])
def test_conv_golden(tag, k, relu, dtype):
    g = load_ops()
    w, b = g[f"{tag}.p.custom_conv.weight"], g[f"{tag}.p.custom_conv.bias"]
    y, _, dx, grads = run_graph(_conv_builder(w.shape[0], k, relu), {"w": torch.tensor(w), "b": torch.tensor(b)},
                                g[f"{tag}.x"], g[f"{tag}.gy"], dtype)
    _check(g, tag, y, dx, grads, "custom_conv.weight", "custom_conv.bias", TOL[dtype])


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("d", [1, 2, 4, 8])
def test_dilconv_lrn_golden(d, dtype):
    g = load_ops(); tag = f"dil{d}"
    w, b = g[f"{tag}.p.conv.weight"], g[f"{tag}.p.conv.bias"]
    y, _, dx, grads = run_graph(_conv_builder(w.shape[0], 3, False, dil=d, lrn=True),
                                {"w": torch.tensor(w), "b": torch.tensor(b)}, g[f"{tag}.x"], g[f"{tag}.gy"], dtype)
    _check(g, tag, y, dx, grads, "conv.weight", "conv.bias", TOL[dtype])


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C", [8, 16, 64])
def test_lrn_golden(C, dtype):
    g = load_ops(); tag = f"lrn{C}"

    def build(plan):
        y = Act(plan, "y", plan.x_in.H, plan.x_in.W, C)
        LrnOp(plan, "l", plan.x_in, y)
        plan.logits = y
    y, _, dx, _ = run_graph(build, {}, g[f"{tag}.x"], g[f"{tag}.gy"], dtype)
    assert err(y, g[f"{tag}.y"], dtype == L.BF16) < TOL[dtype] and err(dx, g[f"{tag}.gx"], dtype == L.BF16) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C,n", [(5, 5), (12, 12), (24, 24), (40, 7)])
def test_lrn_generic_vs_oracle(C, n, dtype):
    """channel counts that take the generic (non power-of-two / padded) LRN path"""
    torch.manual_seed(C)
    x = 10 * torch.randn(2, C, 6, 5); gy = torch.randn(2, C, 6, 5)
    xr = x.clone().requires_grad_(True)
    yr = O.lrn(xr, n); yr.backward(gy)

    def build(plan):
        y = Act(plan, "y", 6, 5, C)
        op = LrnOp(plan, "l", plan.x_in, y)
        plan.logits = y
        op.n = n
    # LrnOp uses size = C; emulate other sizes only when n == C
    if n != C:
        pytest.skip("plan-level LRN always uses size == channels (the only form the reference instantiates)")
    y, _, dx, _ = run_graph(build, {}, x, gy, dtype)
    assert err(y, yr.detach(), dtype == L.BF16) < TOL[dtype] and err(dx, xr.grad, dtype == L.BF16) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("tag", ["deconv_even", "deconv_odd"])
def test_deconv_golden(tag, dtype):
    g = load_ops()
    w, b = g[f"{tag}.p.conv.weight"], g[f"{tag}.p.conv.bias"]
    hw = tuple(g[f"{tag}.y"].shape[2:])
    y, _, dx, grads = run_graph(_conv_builder(w.shape[1], 3, False, kind="deconv", out_hw=hw),
                                {"w": torch.tensor(w), "b": torch.tensor(b)}, g[f"{tag}.x"], g[f"{tag}.gy"], dtype)
    _check(g, tag, y, dx, grads, "conv.weight", "conv.bias", TOL[dtype])


@pytest.mark.parametrize("dtype", DT)
def test_res_block_golden(dtype):
    g = load_ops(); tag = "res"
    params = {f"r.conv_res_list.{i}.custom_conv.{n}": torch.tensor(g[f"{tag}.p.conv_res_list.{i}.custom_conv.{n}"])
              for i in range(2) for n in ("weight", "bias")}

    def build(plan):
        plan.cfg.update(res_depth=2, filter_size=3)
        plan.logits = plan._res_block(plan.x_in, "r", "t")
    y, _, dx, grads = run_graph(build, params, g[f"{tag}.x"], g[f"{tag}.gy"], dtype)
    tol, bf = TOL[dtype], dtype == L.BF16
    assert err(y, g[f"{tag}.y"], bf) < tol and err(dx, g[f"{tag}.gx"], bf) < tol
    for k, v in grads.items():
        assert err(v, g[f"{tag}.g.{k[2:]}"], bf) < tol, k


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("tag", ["attn64", "attn32"])
def test_attention_golden(tag, dtype):
    g = load_ops()
    params = {f"{m}.{n}": torch.tensor(g[f"{tag}.p.attention_block.{m}.conv.{n}"]) for m in "fgh" for n in ("weight", "bias")}
    C = g[f"{tag}.x"].shape[1]

    def build(plan):
        x = plan.x_in
        f = Act(plan, "f", x.H, x.W, C // 8); gg = Act(plan, "g", x.H, x.W, C // 8); h = Act(plan, "h", x.H, x.W, C)
        ConvOp(plan, "f", x, None, "f.weight", "f.bias", f, 1)
        ConvOp(plan, "g", x, None, "g.weight", "g.bias", gg, 1)
        hop = ConvOp(plan, "h", x, None, "h.weight", "h.bias", h, 1)
        y = Act(plan, "y", x.H, x.W, C)
        AttnCoreOp(plan, "a", f, gg, h, x, y)
        hop.bwd_add = y
        plan.logits = y
    y, _, dx, grads = run_graph(build, params, g[f"{tag}.x"], g[f"{tag}.gy"], dtype)
    tol, bf = TOL[dtype] * (3 if dtype == L.BF16 else 1), dtype == L.BF16
    assert err(y, g[f"{tag}.y"], bf) < tol and err(dx, g[f"{tag}.gx"], bf) < tol
    for k, v in grads.items():
        m, n = k.split(".")
        ref = g[f"{tag}.g.attention_block.{m}.conv.{n}"]
        # the f-bias gradient is identically 0 (softmax is invariant to it): absolute check
        assert err(v, ref, bf) < tol or float(np.abs(ref).max()) < 1e-5, k


@pytest.mark.parametrize("dtype", DT)
def test_pool_golden(dtype):
    g = load_ops()
    x = g["pool.x"]

    def build(plan):
        xi = plan.x_in
        y = Act(plan, "y", (xi.H + 1) // 2, (xi.W + 1) // 2, xi.C)
        PoolOp(plan, "p", xi, y)
        plan.logits = y
    y, _, dx, _ = run_graph(build, {}, x, g["pool.gy"], dtype)
    if dtype == L.F32:
        assert np.array_equal(y.numpy(), g["pool.y"]) and np.array_equal(dx.numpy(), g["pool.gx"])   # bit exact
    else:
        assert rel_err(y, g["pool.y"]) < 1e-2


@pytest.mark.parametrize("dtype", DT)
def test_dual_source_conv_vs_oracle(dtype):
    """concat([x1, x2]) -> 3x3 conv and 1x1+ReLU conv (model/model.py:147-148,242-244) with two source pointers"""
    torch.manual_seed(5)
    B, c, H, W = 2, 16, 13, 21
    x = torch.randn(B, 8, H, W)
    p = {"w0": 0.2 * torch.randn(c, 8, 3, 3), "b0": 0.1 * torch.randn(c),
         "w1": 0.2 * torch.randn(c, 8, 1, 1), "b1": 0.1 * torch.randn(c),
         "wm": 0.1 * torch.randn(c, 2 * c, 3, 3), "bm": 0.1 * torch.randn(c),
         "wc": 0.2 * torch.randn(c, 2 * c, 1, 1), "bc": 0.1 * torch.randn(c)}
    gy = torch.randn(B, c, H, W)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = x.clone().requires_grad_(True)
    a = O.conv_same(xr, leaves["w0"], leaves["b0"], relu=True)
    b2 = O.conv_same(xr, leaves["w1"], leaves["b1"])
    m = O.conv_same(torch.cat([a, b2], 1), leaves["wm"], leaves["bm"])
    yr = O.conv_same(torch.cat([b2, m], 1), leaves["wc"], leaves["bc"], relu=True)
    yr.backward(gy)

    def build(plan):
        xi = plan.x_in
        A = Act(plan, "A", H, W, c, relu_out=True); Bt = Act(plan, "B", H, W, c)
        ConvOp(plan, "a", xi, None, "w0", "b0", A, 3, relu_out=True)
        ConvOp(plan, "b", xi, None, "w1", "b1", Bt, 1)
        M = Act(plan, "M", H, W, c)
        ConvOp(plan, "m", A, Bt, "wm", "bm", M, 3)
        Y = Act(plan, "Y", H, W, c, relu_out=True)
        ConvOp(plan, "c", Bt, M, "wc", "bc", Y, 1, relu_out=True)
        plan.logits = Y
    y, _, dx, grads = run_graph(build, p, x, gy, dtype)
    tol, bf = TOL[dtype], dtype == L.BF16
    assert err(y, yr.detach(), bf) < tol and err(dx, xr.grad, bf) < TOLG[dtype]
    for k in p:
        assert err(grads[k], leaves[k].grad, bf) < TOLG[dtype], k


@pytest.mark.parametrize("dtype", DT)
def test_wide_input_chunked_K(dtype):
    """Cin = 120 (K chunked) -> 24: exercises channel chunking and non power-of-two channel counts"""
    torch.manual_seed(6)
    x = torch.randn(1, 120, 9, 17)
    p = {"w": 0.05 * torch.randn(24, 120, 3, 3), "b": 0.1 * torch.randn(24)}
    gy = torch.randn(1, 24, 9, 17)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = x.clone().requires_grad_(True)
    yr = O.conv_same(xr, leaves["w"], leaves["b"]); yr.backward(gy)
    y, _, dx, grads = run_graph(_conv_builder(24, 3, False), p, x, gy, dtype)
    tol, bf = TOL[dtype], dtype == L.BF16
    assert err(y, yr.detach(), bf) < tol and err(dx, xr.grad, bf) < tol
    assert err(grads["w"], leaves["w"].grad, bf) < tol and err(grads["b"], leaves["b"].grad, bf) < tol


def test_masked_ce_golden():
    g = load_ops()
    from msau_amd.model import _MaskedCEFunction
    lg = torch.tensor(g["ce.logits"], device="cuda", requires_grad=True)
    ax = torch.tensor(g["ce.aux"], device="cuda", requires_grad=True)
    lab = torch.tensor(g["ce.label"], device="cuda")
    loss = _MaskedCEFunction.apply(lg, ax, lab)
    loss.backward()
    assert abs(float(loss) - float(g["ce.loss"])) < 1e-5
    assert rel_err(lg.grad.cpu(), g["ce.glogits"]) < 1e-5 and rel_err(ax.grad.cpu(), g["ce.gaux"]) < 1e-5


def test_masked_ce_batch_and_empty_sample():
    """batch rule of SURVEY 8(e): per-sample masked mean, mean over samples; a sample without labels contributes 0"""
    from msau_amd.model import _MaskedCEFunction
    torch.manual_seed(3)
    lg = torch.randn(3, 5, 9, 7); ax = torch.randn(3, 5, 9, 7)
    lab = torch.randint(0, 5, (3, 9, 7)); lab[1] = 0
    lr_, ar_ = lg.clone().requires_grad_(True), ax.clone().requires_grad_(True)
    ref = O.msau_loss(lr_, ar_, lab); ref.backward()
    lgc, axc = lg.cuda().requires_grad_(True), ax.cuda().requires_grad_(True)
    loss = _MaskedCEFunction.apply(lgc, axc, lab.cuda()); loss.backward()
    assert abs(float(loss) - float(ref)) < 1e-5
    assert rel_err(lgc.grad.cpu(), lr_.grad) < 1e-5 and rel_err(axc.grad.cpu(), ar_.grad) < 1e-5


def test_clip_adam_vs_oracle():
    torch.manual_seed(9)
    n = 10007
    p0, g0 = torch.randn(n), 3 * torch.randn(n)
    P = {"p": p0.clone()}; M = {"p": torch.zeros(n)}; V = {"p": torch.zeros(n)}
    dev = "cuda"
    p, m, v = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    state = torch.zeros(8, device=dev); ws = torch.zeros(int(L.load().msau_adam_ws_floats(n)), device=dev)
    for step in range(1, 4):
        g = g0 * step
        gn = O.clip_adam_step(P, {"p": g.clone()}, M, V, step)
        gd = g.to(dev)
        L.call("msau_clip_adam_step", torch.cuda.current_stream().cuda_stream, p.data_ptr(), gd.data_ptr(), m.data_ptr(),
               v.data_ptr(), state.data_ptr(), ws.data_ptr(), n, 1e-4, 0.9, 0.999, 1e-8, 1.0, 1.0)
        assert abs(float(state[1]) - gn) < 1e-4 * gn and int(state[0]) == step
        assert float((p.cpu() - P["p"]).abs().max()) < 2e-7


@pytest.mark.parametrize("N,C", [(1344, 64), (200, 32), (77, 64)])
def test_attention_mfma_bf16_vs_fp32_kernels(N, C):
    """the bf16 MFMA attention kernels against the fp32 VALU kernels on the same (bf16-representable) inputs,
    at the real bottleneck size N = 42*32 and at sizes that are not multiples of the 16/32/128 tiling"""
    torch.manual_seed(N)
    B, D = 2, 8
    dev = "cuda"
    mk = lambda *s, sc=1.0: (sc * torch.randn(*s, device=dev)).bfloat16()
    f, g, h, x, dy = mk(B, N, D, sc=0.7), mk(B, N, D, sc=0.7), mk(B, N, C), mk(B, N, C), mk(B, N, C)
    s = torch.cuda.current_stream().cuda_stream
    outs = {}
    for name, dt, cast in (("ref", L.F32, torch.float32), ("mfma", L.BF16, torch.bfloat16)):
        ff, gg, hh, xx, dd = (t.to(cast).contiguous() for t in (f, g, h, x, dy))
        y = torch.empty_like(xx); stats = torch.zeros(B, N, 2, device=dev); ws = torch.zeros(B, N, device=dev)
        df, dg, dh = torch.empty_like(ff), torch.empty_like(gg), torch.empty_like(hh)
        L.call("msau_selfattn_fwd", s, dt, ff.data_ptr(), gg.data_ptr(), hh.data_ptr(), xx.data_ptr(), y.data_ptr(),
               stats.data_ptr(), B, N, D, C)
        L.call("msau_selfattn_bwd", s, dt, ff.data_ptr(), gg.data_ptr(), hh.data_ptr(), dd.data_ptr(), stats.data_ptr(),
               df.data_ptr(), dg.data_ptr(), dh.data_ptr(), ws.data_ptr(), B, N, D, C)
        torch.cuda.synchronize()
        outs[name] = [t.float().cpu() for t in (y, stats, df, dg, dh)]
    from tests.golden_util import rel_l2
    for nm, a, b in zip(("y", "stats", "df", "dg", "dh"), outs["mfma"], outs["ref"]):
        assert torch.isfinite(a).all(), nm
        assert rel_l2(a, b) < (1e-4 if nm == "stats" else 2e-2), (nm, rel_l2(a, b))


def test_device_chargrid_rasteriser_matches_the_reference_chargrids(tmp_path):
    """N1: box lists painted on the device against tests/golden/funsd/chargrid.npz DIRECTLY -- the arrays the reference's own
    funsd_preprocessing_word_level.py + data_generator_funsd_bert.py:149-186 painted for the committed documents (bit exact,
    train and test split, label map of the fixture) -- and, as before, against the product's CPU painter."""
    import json, os, pickle
    import numpy as np
    from msau_amd.data import funsd as F
    from msau_amd.data.raster import document_boxes, rasterize
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "funsd")
    g = np.load(os.path.join(G, "chargrid.npz"), allow_pickle=True)
    labels_map = json.loads(str(g["labels_json"]))
    train, inv = F.get_preprocessed_list_word_msau(os.path.join(G, "train"))
    test, _ = F.get_preprocessed_list_word_msau(os.path.join(G, "test"), inv_dict_charset=inv)
    C = len(inv)
    checked = 0
    for split, docs in (("train", train), ("test", test)):
        docs.sort(key=lambda d: d["file_path"])
        pickle.dump(docs, open(tmp_path / f"{split}.pkl", "wb"))
        ds = F.FUNSDCharGridDataLoaderBoxMaskBoxLabel(str(tmp_path / f"{split}.pkl"), labels_map, write_labels_file=False)
        for i in range(len(ds)):
            want_mask, want_label = g[f"{split}{i}.mask"], g[f"{split}{i}.label"]          # uint8 [1,C,H,W], [1,H,W]: the reference's arrays
            cb, lb, H, W = document_boxes(ds.inp_list[i])
            assert (1, C, H, W) == tuple(want_mask.shape), (split, i)
            ref = ds[i]
            for dtype in ("fp32", "bf16"):
                grid, labels = rasterize(cb, lb, 1, H, W, C, dtype)
                got = grid[..., :C].permute(0, 3, 1, 2).float().cpu()
                assert np.array_equal(got.numpy(), want_mask.astype(np.float32)), (split, i, dtype)
                assert np.array_equal(labels.cpu().numpy(), want_label.astype(np.int64)), (split, i, dtype)
                assert float(grid[..., C:].abs().sum()) == 0.0
                assert torch.equal(got, ref["mask"]) and torch.equal(labels.cpu(), ref["label"].long())
                checked += 1
    assert checked >= 4


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("C,with_aux", [(5, True), (5, False), (12, True)])
def test_masked_ce_multi_equals_two_single_launches(dtype, C, with_aux):
    """final + aux CE in one launch: gradients bit-identical to msau_masked_ce per tensor, loss equal to their sum"""
    torch.manual_seed(3)
    B, H, W = 3, 37, 29
    Cs = -(-C // 8) * 8
    td = torch.float32 if dtype == "fp32" else torch.bfloat16
    dt = L.F32 if dtype == "fp32" else L.BF16
    lg = (torch.randn(B, H, W, Cs) * 2).to(td).cuda()
    ax = (torch.randn(B, H, W, Cs) * 2).to(td).cuda()
    labels = torch.randint(0, C, (B, H, W), dtype=torch.int64)
    labels[1] = 0                                              # a sample without labelled pixels contributes 0
    labels = labels.cuda()
    s = torch.cuda.current_stream().cuda_stream
    counts = torch.zeros(B, dtype=torch.int32, device="cuda")
    L.call("msau_label_counts", s, labels.data_ptr(), counts.data_ptr(), B, H * W)
    ws1 = torch.zeros(int(L.load().msau_ce_ws_floats(B * H * W)), device="cuda")
    ws2 = torch.zeros(int(L.load().msau_ce_multi_ws_floats(B * H * W)), device="cuda")
    ref_loss = torch.zeros(1, device="cuda")
    d_ref = []
    for t in ([lg, ax] if with_aux else [lg]):
        d = torch.empty_like(t)
        L.call("msau_masked_ce", s, dt, t.data_ptr(), labels.data_ptr(), counts.data_ptr(), d.data_ptr(), ref_loss.data_ptr(),
               ws1.data_ptr(), B, H * W, C, Cs, 1.0 / B)
        d_ref.append(d)
    loss = torch.full((1,), 123.0, device="cuda")                # overwritten, not accumulated
    d0, d1 = torch.empty_like(lg), torch.empty_like(ax)
    for _ in range(2):                                           # the scratch needs no re-initialisation between calls
        L.call("msau_masked_ce_multi", s, dt, lg.data_ptr(), ax.data_ptr() if with_aux else None, labels.data_ptr(),
               counts.data_ptr(), d0.data_ptr(), d1.data_ptr() if with_aux else None, loss.data_ptr(), ws2.data_ptr(),
               B, H * W, C, Cs, 1.0 / B, 1)
    # the label counts as K partial counts per sample (msau_label_counts_split), added up by the CE launch: the same bits
    for K in (2, 7, 16):
        part = torch.full((B * K,), -5, dtype=torch.int32, device="cuda")
        L.call("msau_label_counts_split", s, labels.data_ptr(), part.data_ptr(), B, H * W, K)
        assert torch.equal(part.view(B, K).sum(1).to(torch.int32), counts)
        loss_k = torch.full((1,), 321.0, device="cuda")
        dk0, dk1 = torch.empty_like(lg), torch.empty_like(ax)
        L.call("msau_masked_ce_multi", s, dt, lg.data_ptr(), ax.data_ptr() if with_aux else None, labels.data_ptr(),
               part.data_ptr(), dk0.data_ptr(), dk1.data_ptr() if with_aux else None, loss_k.data_ptr(), ws2.data_ptr(),
               B, H * W, C, Cs, 1.0 / B, K)
        torch.cuda.synchronize()
        assert torch.equal(dk0, d0) and torch.equal(loss_k, loss) and (not with_aux or torch.equal(dk1, d1))
    assert torch.equal(d0, d_ref[0])
    if with_aux:
        assert torch.equal(d1, d_ref[1])
    assert abs(float(loss) - float(ref_loss)) <= 2e-6 * abs(float(ref_loss))
    assert float(d0[1].float().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_masked_ce_more_than_16_classes(dtype):
    """the 17-class key-value head (inference/postprocess.py CLASS_NAMES): CE value and gradient against torch"""
    torch.manual_seed(11)
    B, H, W, C = 2, 21, 17, 17
    Cs = 24
    td = torch.float32 if dtype == "fp32" else torch.bfloat16
    dt = L.F32 if dtype == "fp32" else L.BF16
    lg = torch.zeros(B, H, W, Cs)
    lg[..., :C] = torch.randn(B, H, W, C) * 2
    lg = lg.to(td).cuda()
    labels = torch.randint(0, C, (B, H, W), dtype=torch.int64).cuda()
    s = torch.cuda.current_stream().cuda_stream
    counts = torch.zeros(B, dtype=torch.int32, device="cuda")
    L.call("msau_label_counts", s, labels.data_ptr(), counts.data_ptr(), B, H * W)
    ws = torch.zeros(int(L.load().msau_ce_ws_floats(B * H * W)), device="cuda")
    loss = torch.zeros(1, device="cuda")
    d = torch.empty_like(lg)
    L.call("msau_masked_ce", s, dt, lg.data_ptr(), labels.data_ptr(), counts.data_ptr(), d.data_ptr(), loss.data_ptr(), ws.data_ptr(),
           B, H * W, C, Cs, 1.0 / B)
    x = lg[..., :C].float().clone().requires_grad_(True)
    ref = 0.0
    for b in range(B):
        sel = labels[b] != 0
        ref = ref + torch.nn.functional.cross_entropy(x[b][sel], labels[b][sel]) / B
    ref.backward()
    tol = 1e-5 if dtype == "fp32" else 1e-2
    assert abs(float(loss) - float(ref)) < max(tol, 1e-5) * abs(float(ref)) + 1e-6
    assert float((d[..., :C].float() - x.grad).abs().max()) < tol * float(x.grad.abs().max()) + 1e-7
    assert float(d[..., C:].float().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("k,cin,cout", [(1, 192, 64), (3, 192, 64), (1, 136, 8)])
def test_data_gradient_wider_than_128_channels(dtype, k, cin, cout):
    """a conv whose INPUT has more than 128 channels: its data gradient is a launch with > 128 output channels, cut into
    128-channel slices by msau_conv2d (the 1x1 convs 3c -> c of the box variant at c = 64; the reference's 256-channel
    levels); includes a remainder slice (136 = 128 + 8)"""
    torch.manual_seed(9)
    x = torch.randn(2, cin, 10, 19)
    p = {"w": 0.05 * torch.randn(cout, cin, k, k), "b": 0.1 * torch.randn(cout)}
    gy = torch.randn(2, cout, 10, 19)
    leaves = {kk: v.clone().requires_grad_(True) for kk, v in p.items()}
    xr = x.clone().requires_grad_(True)
    yr = O.conv_same(xr, leaves["w"], leaves["b"]); yr.backward(gy)
    y, _, dx, grads = run_graph(_conv_builder(cout, k, False), p, x, gy, dtype)
    tol, bf = TOL[dtype], dtype == L.BF16
    assert err(y, yr.detach(), bf) < tol and err(dx, xr.grad, bf) < tol
    assert err(grads["w"], leaves["w"].grad, bf) < tol and err(grads["b"], leaves["b"].grad, bf) < tol


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cin,cout,k,dil,hw", [(64, 192, 1, 1, (10, 19)), (128, 256, 3, 32, (11, 8)), (64, 128, 3, 16, (21, 40)),
                                                (256, 256, 3, 1, (6, 9))])
def test_wide_outputs_and_large_dilations(dtype, cin, cout, k, dil, hw):
    """the 256-channel levels of the reference's constructor defaults (model/model.py:406-408: 6 scales, dilation 2^l up to 32):
    more than 128 OUTPUT channels (forward and weight gradient in channel slices) and dilations whose halo rectangle would
    not fit the LDS (the tile then holds one block per tap)"""
    torch.manual_seed(11)
    H, W = hw
    x = torch.randn(2, cin, H, W)
    p = {"w": 0.05 * torch.randn(cout, cin, k, k), "b": 0.1 * torch.randn(cout)}
    gy = torch.randn(2, cout, H, W)
    leaves = {kk: v.clone().requires_grad_(True) for kk, v in p.items()}
    xr = x.clone().requires_grad_(True)
    yr = O.conv_same(xr, leaves["w"], leaves["b"], dilation=dil); yr.backward(gy)
    y, _, dx, grads = run_graph(_conv_builder(cout, k, False, dil=dil), p, x, gy, dtype)
    tol, bf = TOL[dtype], dtype == L.BF16
    assert err(y, yr.detach(), bf) < tol and err(dx, xr.grad, bf) < tol
    assert err(grads["w"], leaves["w"].grad, bf) < tol and err(grads["b"], leaves["b"].grad, bf) < tol


@pytest.mark.parametrize("dtype", DT)
def test_lrn_and_attention_at_256_channels(dtype):
    torch.manual_seed(12)
    x = 5.0 * torch.randn(2, 256, 5, 7)
    gy = torch.randn(2, 256, 5, 7)
    xr = x.clone().requires_grad_(True)
    yr = O.lrn(xr, 256); yr.backward(gy)

    def build(plan):
        y = Act(plan, "y", plan.x_in.H, plan.x_in.W, 256)
        LrnOp(plan, "l", plan.x_in, y)
        plan.logits = y
    y, _, dx, _ = run_graph(build, {}, x, gy, dtype)
    bf = dtype == L.BF16
    assert err(y, yr.detach(), bf) < TOL[dtype] and err(dx, xr.grad, bf) < TOL[dtype]
