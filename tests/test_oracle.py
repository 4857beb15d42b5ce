"""CPU: the oracle restatement (oracle/msau_oracle.py) against golden vectors produced by the
reference itself (oracle/gen_goldens.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import msau_oracle as O
from tests.golden_util import NET_CASES, load_net_case, load_ops, rel_err, summarize

TOL = 2e-5      # fp32 restatement vs fp32 reference, same backend ops: only summation-order noise


def _grad(fn, x, gy, params):
    x = torch.tensor(x, requires_grad=True)
    ps = {k: torch.tensor(v, requires_grad=True) for k, v in params.items()}
    y = fn(x, ps)
    y.backward(torch.tensor(gy))
    return y.detach(), x.grad, {k: p.grad for k, p in ps.items()}


def _params(g, tag):
    pre = f"{tag}.p."
    return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}


@pytest.mark.parametrize("tag,relu,dil", [("conv3", True, 1), ("conv3lin", False, 1), ("conv1", True, 1),
                                          ("conv4", False, 1), ("conv3c13", False, 1)])
def test_conv_same(tag, relu, dil):
    g = load_ops()
    y, gx, gp = _grad(lambda x, p: O.conv_same(x, p["custom_conv.weight"], p["custom_conv.bias"], dil, relu),
                      g[f"{tag}.x"], g[f"{tag}.gy"], _params(g, tag))
    assert rel_err(y, g[f"{tag}.y"]) < TOL
    assert rel_err(gx, g[f"{tag}.gx"]) < TOL
    for k, v in gp.items():
        assert rel_err(v, g[f"{tag}.g.{k}"]) < TOL


@pytest.mark.parametrize("d", [1, 2, 4, 8])
def test_dilconv_lrn(d):
    g = load_ops(); tag = f"dil{d}"
    y, gx, gp = _grad(lambda x, p: O.dilconv_lrn(x, p["conv.weight"], p["conv.bias"], d),
                      g[f"{tag}.x"], g[f"{tag}.gy"], _params(g, tag))
    assert rel_err(y, g[f"{tag}.y"]) < TOL
    assert rel_err(gx, g[f"{tag}.gx"]) < TOL
    for k, v in gp.items():
        assert rel_err(v, g[f"{tag}.g.{k}"]) < TOL


@pytest.mark.parametrize("C", [8, 16, 64])
def test_lrn(C):
    g = load_ops(); tag = f"lrn{C}"
    y, gx, _ = _grad(lambda x, p: O.lrn(x, C), g[f"{tag}.x"], g[f"{tag}.gy"], {})
    assert rel_err(y, g[f"{tag}.y"]) < TOL
    assert rel_err(gx, g[f"{tag}.gx"]) < TOL


@pytest.mark.parametrize("tag", ["deconv_even", "deconv_odd"])
def test_deconv(tag):
    g = load_ops()
    hw = tuple(g[f"{tag}.y"].shape[2:])
    y, gx, gp = _grad(lambda x, p: O.deconv(x, p["conv.weight"], p["conv.bias"], hw),
                      g[f"{tag}.x"], g[f"{tag}.gy"], _params(g, tag))
    assert rel_err(y, g[f"{tag}.y"]) < TOL and rel_err(gx, g[f"{tag}.gx"]) < TOL
    for k, v in gp.items():
        assert rel_err(v, g[f"{tag}.g.{k}"]) < TOL


def test_res_block():
    g = load_ops(); tag = "res"
    y, gx, gp = _grad(lambda x, p: O.res_block(x, {"r." + k: v for k, v in p.items()}, "r", 2),
                      g[f"{tag}.x"], g[f"{tag}.gy"], _params(g, tag))
    assert rel_err(y, g[f"{tag}.y"]) < TOL and rel_err(gx, g[f"{tag}.gx"]) < TOL
    for k, v in gp.items():
        assert rel_err(v, g[f"{tag}.g.{k}"]) < TOL


@pytest.mark.parametrize("tag", ["attn64", "attn32"])
def test_attention(tag):
    g = load_ops()
    y, gx, gp = _grad(lambda x, p: O.self_attention(x, p, "attention_block"),
                      g[f"{tag}.x"], g[f"{tag}.gy"], _params(g, tag))
    assert rel_err(y, g[f"{tag}.y"]) < TOL and rel_err(gx, g[f"{tag}.gx"]) < TOL
    for k, v in gp.items():
        assert rel_err(v, g[f"{tag}.g.{k}"]) < TOL


def test_pool():
    g = load_ops()
    x = torch.tensor(g["pool.x"], requires_grad=True)
    y = torch.nn.functional.max_pool2d(O.pad_same(x, 2, 2, 2, 2), 2, 2)
    y.backward(torch.tensor(g["pool.gy"]))
    assert np.array_equal(y.detach().numpy(), g["pool.y"]) and np.array_equal(x.grad.numpy(), g["pool.gx"])


def test_masked_ce():
    g = load_ops()
    lg = torch.tensor(g["ce.logits"], requires_grad=True); ax = torch.tensor(g["ce.aux"], requires_grad=True)
    loss = O.msau_loss(lg, ax, torch.tensor(g["ce.label"]))
    loss.backward()
    assert abs(float(loss) - float(g["ce.loss"])) < 1e-6
    assert rel_err(lg.grad, g["ce.glogits"]) < TOL and rel_err(ax.grad, g["ce.gaux"]) < TOL


def test_same_pads():
    assert O.same_pads(9, 4) == (1, 2)          # 4x4 -> top1/bottom2
    assert O.same_pads(9, 3, 1, 8) == (8, 8)
    assert O.same_pads(7, 2, 2) == (0, 1) and O.same_pads(8, 2, 2) == (0, 0)


def test_param_count_cfg2():
    n = sum(int(np.prod(s)) for s in O.param_shapes(O.DEFAULT_CFG).values())
    assert n == 636167 and len(O.param_shapes(O.DEFAULT_CFG)) == 196


@pytest.mark.parametrize("name", NET_CASES)
def test_net_forward_loss_grads_step(name):
    g, cfg, sd, x, label = load_net_case(name)
    big = "logits_sub" in g.files
    with_step = "loss" in g.files
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v = {k: torch.zeros_like(t) for k, t in sd.items()}
    before = {k: t.clone() for k, t in sd.items()}
    if with_step:
        loss, logits, aux, grads, gn = O.train_step(sd, m, v, 1, x, label, cfg)
    else:
        with torch.no_grad():
            logits, aux = O.msau_forward(sd, x, cfg)
    pred = O.predictor(logits)
    for nm, t in (("logits", logits), ("aux", aux), ("pred", pred)):
        if t is None:
            assert nm == "aux" and "aux" not in g.files and "aux_sub" not in g.files
            continue
        if big:
            assert rel_err(t[:, :, ::7, ::5], g[nm + "_sub"]) < 1e-4
            assert abs(summarize(t)[0][0] - g[nm + "_summary"][0]) < 1e-4 * g[nm + "_summary"][0]
        else:
            assert rel_err(t, g[nm]) < 1e-4, nm
    if not with_step:
        return
    assert abs(loss - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    assert abs(gn - float(g["grad_norm"])) < 1e-3 * float(g["grad_norm"])
    names = [str(s) for s in g["param_names"]]
    dead = set(str(s) for s in g["dead_params"])
    # the reference leaves exactly the last stage's attention parameters without a gradient
    assert dead == {k for k in names if f"blocks.{cfg['num_blocks'] - 1}.downsamplingblock.layer_attentions" in k}
    gmax = max(float(s[0]) for s in g["grad_summary"])
    for i, k in enumerate(names):
        if k in dead:
            assert grads[k] is None or float(grads[k].abs().max()) == 0.0
            assert torch.equal(sd[k], before[k])
            continue
        s, smp = summarize(grads[k])
        ref_s, ref_smp = g["grad_summary"][i], g["grad_samples"][i]
        # fp32 on the CPU both sides, but not the same kernels (oneDNN picks its algorithms per call site): 2e-3 of a tensor's
        # largest sampled gradient; 1e-2 for the reference's default geometry (6 scales, res_depth 3: ~110 convs deep,
        # observed 5.3e-3 on a transposed-conv weight while the logits agree to 2e-6)
        gt = 1e-2 if "defaults" in name else 2e-3
        assert abs(s[0] - ref_s[0]) <= gt * ref_s[0] + 1e-6 * gmax, k
        assert np.abs(smp - ref_smp).max() <= gt * np.abs(ref_smp).max() + 1e-5 * gmax, k
        # Adam's first step moves every live element by ~lr * sign(g): compare the norm of the move
        # (skipped where |g| is within noise of Adam's eps, e.g. the attention f-bias whose true grad is 0)
        if ref_s[0] / np.sqrt(sd[k].numel()) > 1e-6:
            ds, dsmp = summarize(sd[k] - before[k])
            assert abs(ds[0] - g["delta_summary"][i][0]) <= 2e-2 * g["delta_summary"][i][0] + 1e-9, k


def test_bf16_storage_mode_is_the_same_function_with_rounding_points():
    """`msau_forward(..., storage="bf16")` (the checker of the device path's throughput mode, tests/test_net_gpu.py) is the
    pinned fp32 restatement plus roundings: "fp32" / None leave it untouched bit for bit, "bf16" stays within bf16 noise of it,
    its outputs ARE bf16 values, and the weights it was handed are not modified."""
    g, cfg, sd, x, label = load_net_case("net_f4_c13_b2_64x48")
    sd0 = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        ref, ref_aux = O.msau_forward(sd, x, cfg)
        same, same_aux = O.msau_forward(sd, x, cfg, storage="fp32")
        low, low_aux = O.msau_forward(sd, x, cfg, storage="bf16")
    assert torch.equal(ref, same) and torch.equal(ref_aux, same_aux)
    assert rel_err(ref, g["logits"]) < 2e-5                                   # (the pinned path)
    assert all(torch.equal(sd[k], sd0[k]) for k in sd)
    assert torch.equal(low, low.bfloat16().float()) and torch.equal(low_aux, low_aux.bfloat16().float())
    from tests.golden_util import rel_l2
    assert 1e-4 < rel_l2(low, ref) < 3e-2 and rel_l2(low_aux, ref_aux) < 3e-2, (rel_l2(low, ref), rel_l2(low_aux, ref_aux))


def test_bf16_oracle_rounding_noise_envelope():
    """the premise of tests/test_net_gpu.py::test_bf16_logits_cost_no_more_than_bf16_storage_itself, kept honest: two evaluations
    with the SAME rounding points (fp32 against float64 sums in between) decorrelate -- they end up as far from each other as
    either is from the fp32 reference, so a network-level bf16 bound is a radius around the reference, not a few ulps"""
    from tests.golden_util import rel_l2
    g, cfg, sd, x, label = load_net_case("net_r3_s3_c8_21x35")
    with torch.no_grad():
        a32, _ = O.msau_forward(sd, x, cfg, storage="bf16")
        a64, _ = O.msau_forward(sd, x.double(), cfg, storage="bf16")
        r32, _ = O.msau_forward(sd, x, cfg)
    assert a64.dtype == torch.float64
    assert 0.3 * rel_l2(a32, r32) < rel_l2(a32, a64.float()) < 1.5 * rel_l2(a32, r32)
