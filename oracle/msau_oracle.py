"""CPU oracle for the MSAU train path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  Nothing under ``msau_amd/`` imports it;
the product path has no CPU fallback and fails loudly without its HIP library.

This is a restatement (plain PyTorch-CPU fp32 functional ops + explicit
formulae, no ``torch.nn`` modules) of the reference's Multi-Stage Attentional
U-Net forward / loss / clip+Adam step.  Each function cites the reference
file:line it follows (paths relative to the reference root).

Parity status: PINNED.  ``oracle/gen_goldens.py`` imports the reference's own
``model.model.MSAUWrapper`` in the build container and writes golden vectors
to ``tests/golden/``; ``tests/test_oracle.py`` checks this restatement against
those vectors (forward logits, loss, per-parameter gradients, parameters after
one clip+Adam step, op-level tensors).

Weights are addressed by the reference's ``state_dict`` keys.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

DEFAULT_CFG = dict(channels=64, n_class=5, scale_space_num=4, res_depth=2,
                   featRoot=8, filter_size=3, pool_size=2, num_blocks=3)


# ---------------------------------------------------------------------------
# storage="bf16": the reference's arithmetic with the ROUNDING POINTS of the
# device path's throughput mode (activations and packed
# weights stored as bf16, accumulation fp32).  `msau_forward(..., storage="bf16")`
# rounds exactly the tensors the device plan stores -- every conv / LRN /
# residual-block / coupling / attention / transposed-conv output, the packed
# weights, the bf16 attention probabilities -- and nothing else, so what is
# left between it and the device's bf16 logits is the order of the fp32 sums.
# The reference itself (model/model.py) is fp32 throughout; storage=None (the
# default, and the only mode the golden vectors pin) is that path unchanged.
# ---------------------------------------------------------------------------
def _ident(t: Tensor) -> Tensor:
    return t


def _to_bf16(t: Tensor) -> Tensor:
    return t.bfloat16().to(t.dtype)                    # (float64 tensors stay float64: exact sums between the same rounding points)


def rounder(storage: Optional[str]):
    assert storage in (None, "fp32", "bf16"), storage
    return _to_bf16 if storage == "bf16" else _ident


# ---------------------------------------------------------------------------
# A1  pad_2d                                        model/layers/utils.py:5-28
# ---------------------------------------------------------------------------
def same_pads(in_size: int, k: int, stride: int = 1, dilation: int = 1) -> Tuple[int, int]:
    """TF "SAME" padding (before, after) for one spatial dim.
    conv/pool: utils.py:8-19; atrous: effective kernel k+(k-1)(d-1), utils.py:21-26."""
    k_eff = k + (k - 1) * (dilation - 1)
    out = int(math.ceil(float(in_size) / float(stride)))
    pad = max((out - 1) * stride + k_eff - in_size, 0)
    return pad // 2, pad - pad // 2


def pad_same(x: Tensor, kh: int, kw: int, sh: int = 1, sw: int = 1, dilation: int = 1) -> Tensor:
    pt, pb = same_pads(x.shape[2], kh, sh, dilation)
    pl, pr = same_pads(x.shape[3], kw, sw, dilation)
    return F.pad(x, (pl, pr, pt, pb))


# ---------------------------------------------------------------------------
# A2  Conv2dBnLrnDrop (bn/lrn/dropout inactive)      model/layers/layers.py:82-102
# ---------------------------------------------------------------------------
def activation_of(cfg: dict):
    """MSAUWrapper's `activation_name` (model/model.py:412-416): "relu" -> torch.nn.ReLU, "elu" -> torch.nn.ELU (alpha 1)."""
    name = cfg.get("activation", "relu")
    assert name in ("relu", "elu"), name
    return F.elu if name == "elu" else torch.relu


def conv_same(x: Tensor, w: Tensor, b: Tensor, dilation: int = 1, relu: bool = False, act=None) -> Tensor:
    """(`relu=True` applies `act`, torch.relu unless the caller passes the net's activation)"""
    kh, kw = w.shape[2], w.shape[3]
    y = F.conv2d(pad_same(x, kh, kw, 1, 1, dilation), w, b, dilation=dilation)
    return (act or torch.relu)(y) if relu else y


# ---------------------------------------------------------------------------
# A3  LocalResponseNorm(size=C)                      model/layers/layers.py:145,161-162
#     torch semantics: div = (k + alpha/n * sum_{c' in [c-n//2, c+(n-1)//2]} x^2)^beta
# ---------------------------------------------------------------------------
def lrn(x: Tensor, size: int, alpha: float = 1e-4, beta: float = 0.75, k: float = 1.0) -> Tensor:
    C = x.shape[1]
    sq = x * x
    csum = torch.cumsum(sq, dim=1)
    csum = torch.cat([torch.zeros_like(csum[:, :1]), csum], dim=1)      # S[t] = sum_{c<t}
    idx = torch.arange(C)
    lo = (idx - size // 2).clamp(min=0)
    hi = (idx + (size - 1) // 2 + 1).clamp(max=C)
    win = csum[:, hi] - csum[:, lo]
    div = (k + (alpha / size) * win) ** beta
    return x / div


def dilconv_lrn(x: Tensor, w: Tensor, b: Tensor, rate: int, q=_ident) -> Tensor:
    """DilConv2dBnLrnDrop.forward with activation=None, use_lrn=True: layers.py:152-164.
    (q: the conv output is a stored tensor of the device plan, the LRN reads it back)"""
    return q(lrn(q(conv_same(x, w, b, dilation=rate)), w.shape[0]))


# ---------------------------------------------------------------------------
# A4  Deconv2DBnLrnDrop                              model/layers/layers.py:221-226,249-250
# ---------------------------------------------------------------------------
def deconv(x: Tensor, w: Tensor, b: Tensor, out_hw: Tuple[int, int]) -> Tensor:
    k = w.shape[2]
    p = k // 2
    oph = out_hw[0] - ((x.shape[2] - 1) * 2 - 2 * p + k)
    opw = out_hw[1] - ((x.shape[3] - 1) * 2 - 2 * p + k)
    assert 0 <= oph < 2 and 0 <= opw < 2, (oph, opw)
    return F.conv_transpose2d(x, w, b, stride=2, padding=p, output_padding=(oph, opw))


# ---------------------------------------------------------------------------
# A5  MultiConvResidualBlock                         model/model.py:37-50
# ---------------------------------------------------------------------------
def res_block(x: Tensor, sd: Dict[str, Tensor], prefix: str, R: int, q=_ident, act=torch.relu) -> Tensor:
    r = torch.relu(x)                                  # model.py:35,39: `self.relu = torch.nn.ReLU()` whatever activation_name says
    for i in range(R):
        w = sd[f"{prefix}.conv_res_list.{i}.custom_conv.weight"]
        b = sd[f"{prefix}.conv_res_list.{i}.custom_conv.bias"]
        r = conv_same(r, w, b, relu=(i < R - 1), act=act)
        if i < R - 1:
            r = q(r)                                   # the inner tensors are stored (the backward reads them)
    return q(act(r + x))


# ---------------------------------------------------------------------------
# A6  SelfAttentionBlock                             model/layers/attention.py:152-162
# ---------------------------------------------------------------------------
def self_attention(x: Tensor, sd: Dict[str, Tensor], prefix: str, q=_ident) -> Tensor:
    B, C, H, W = x.shape
    f = q(F.conv2d(x, sd[f"{prefix}.f.conv.weight"], sd[f"{prefix}.f.conv.bias"])).reshape(B, -1, H * W)
    g = q(F.conv2d(x, sd[f"{prefix}.g.conv.weight"], sd[f"{prefix}.g.conv.bias"])).reshape(B, -1, H * W)
    h = q(F.conv2d(x, sd[f"{prefix}.h.conv.weight"], sd[f"{prefix}.h.conv.bias"])).reshape(B, C, H * W)
    s = torch.matmul(g.transpose(1, 2), f)              # s[i, j] = g_i . f_j
    beta = q(torch.softmax(s, dim=-1))                  # rows normalised (q: the probabilities are the bf16 operand of the next product)
    o = torch.matmul(h, beta)                           # o[:, j] = sum_i h[:, i] beta[i, j]
    return q(o.reshape(B, C, H, W) + x)


# ---------------------------------------------------------------------------
# A7-A9  one U-Net stage                             model/model.py:129-164, 224-259, 328-344
# ---------------------------------------------------------------------------
def stage_forward(inp: Tensor, sd: Dict[str, Tensor], b: int, cfg: dict,
                  prev_dw: Optional[Dict[int, Tensor]], prev_up: Optional[Dict[int, Tensor]],
                  need_attention: bool = True, q=_ident):
    S, R = cfg["scale_space_num"], cfg["res_depth"]
    k, ps = cfg["filter_size"], cfg["pool_size"]
    act = activation_of(cfg)
    assert ps == 2
    coupled = b > 0
    pd = f"msau_net.blocks.{b}.downsamplingblock"
    pu = f"msau_net.blocks.{b}.upsamplingblock"
    dw: Dict[int, Tensor] = {}
    x_in = inp
    x = inp
    for l in range(S):                                                  # model.py:136-162
        x = dilconv_lrn(x_in, sd[f"{pd}.conv1s.{l}.conv.weight"], sd[f"{pd}.conv1s.{l}.conv.bias"], 2 ** l, q)
        x = res_block(x, sd, f"{pd}.conv_res_list.{l}", R, q, act)
        if coupled:                                                     # model.py:143-148
            x = q(conv_same(torch.cat([prev_dw[l], x], dim=1),
                            sd[f"{pd}.conv1_1s.{l}.custom_conv.weight"],
                            sd[f"{pd}.conv1_1s.{l}.custom_conv.bias"], relu=True, act=act))
        if l > S - 2:                                                   # model.py:149-150
            dw[l] = self_attention(x, sd, f"{pd}.layer_attentions.attention_block", q) if need_attention else x
        else:
            dw[l] = x
        if l < S - 1:                                                   # model.py:158-160
            x_in = F.max_pool2d(pad_same(x, ps, ps, ps, ps), ps, ps)
    cur = x                                                             # pre-attention tensor, model.py:162-164
    up: Dict[int, Tensor] = {}
    for l in range(S - 2, -1, -1):                                      # model.py:226-254
        d = q(deconv(cur, sd[f"{pu}.deconvs.{l}.conv.weight"], sd[f"{pu}.deconvs.{l}.conv.bias"],
                     tuple(dw[l].shape[2:])))
        x = q(conv_same(torch.cat([dw[l], d], dim=1),
                        sd[f"{pu}.conv1s.{l}.custom_conv.weight"], sd[f"{pu}.conv1s.{l}.custom_conv.bias"]))
        x = res_block(x, sd, f"{pu}.conv_res_list.{l}", R, q, act)
        if coupled:
            x = q(conv_same(torch.cat([prev_up[l], x], dim=1),
                            sd[f"{pu}.conv1_1s.{l}.custom_conv.weight"],
                            sd[f"{pu}.conv1_1s.{l}.custom_conv.bias"], relu=True, act=act))
        up[l] = x
        cur = x
    return cur, dw, up


# ---------------------------------------------------------------------------
# A10  MSAUNet.forward                               model/model.py:378-396
# ---------------------------------------------------------------------------
def msau_forward(sd: Dict[str, Tensor], inp: Tensor, cfg: dict, storage: Optional[str] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """storage=None / "fp32": the reference's fp32 forward.  storage="bf16": the same arithmetic with the device plan's
    rounding points (see `rounder`): weights (not biases: they stay fp32 on the device too) and the input are rounded once,
    every stored activation where it is stored."""
    nb = cfg.get("num_blocks", 3)
    q = rounder(storage)
    if storage == "bf16":
        sd = {k: (q(v) if k.endswith("weight") else v) for k, v in sd.items()}
    if inp.dtype != torch.float32:                     # float64 input: the sums between the rounding points in float64 too
        sd = {k: v.to(inp.dtype) for k, v in sd.items()}
    prev_dw = prev_up = None
    aux = None
    x = q(inp)
    out = None
    for b in range(nb):
        # the last stage's attention output is never consumed (model.py:149-150,226-227)
        out, prev_dw, prev_up = stage_forward(x, sd, b, cfg, prev_dw, prev_up, need_attention=(b < nb - 1), q=q)
        out = q(conv_same(out, sd[f"msau_net.end_convs.{b}.custom_conv.weight"],
                          sd[f"msau_net.end_convs.{b}.custom_conv.bias"]))   # 4x4, asym SAME pad
        x = out
        if b == nb - 2:
            aux = out
    return out, aux


def predictor(logits: Tensor) -> Tensor:
    """final_act="softmax": model/model.py:426-427,437."""
    return torch.softmax(logits, dim=1)


# ---------------------------------------------------------------------------
# A11  MSAUWrapper.loss                              model/model.py:446-459
# ---------------------------------------------------------------------------
def masked_ce_sample(logits: Tensor, label: Tensor) -> Tensor:
    """logits [C,H,W], label [H,W] long; CE mean over pixels with label != 0.
    A sample with no labelled pixel contributes 0 (the reference would give NaN)."""
    m = label != 0
    n = int(m.sum())
    if n == 0:
        return logits.sum() * 0.0
    lg = logits[:, m].t()                       # [n, C]
    return F.cross_entropy(lg, label[m], reduction="mean")


def msau_loss(logits: Tensor, aux: Optional[Tensor], label: Tensor) -> Tensor:
    """Batch rule: per-sample masked mean CE(final)+CE(aux), then mean over samples.
    For batch 1 this is exactly model/model.py:446-459."""
    B = logits.shape[0]
    tot = logits.new_zeros(())
    for i in range(B):
        tot = tot + masked_ce_sample(logits[i], label[i])
        if aux is not None:
            tot = tot + masked_ce_sample(aux[i], label[i])
    return tot / B


# ---------------------------------------------------------------------------
# A13  clip_grad_norm(max_norm=1.0) + Adam(lr=1e-4)  train_chargrid_funsd_msau.py:24-26,57-59
# ---------------------------------------------------------------------------
def clip_adam_step(params: Dict[str, Tensor], grads: Dict[str, Optional[Tensor]],
                   m: Dict[str, Tensor], v: Dict[str, Tensor], step: int,
                   lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 1.0):
    """In-place.  Parameters whose grad is None are skipped by both clip and Adam
    (torch semantics).  `step` is the 1-based step count.  Returns the pre-clip norm."""
    live = [k for k in params if grads.get(k) is not None]
    total = torch.sqrt(sum((grads[k].double() ** 2).sum() for k in live)).float()
    coef = min(1.0, float(max_norm / (total + 1e-6)))     # torch.nn.utils.clip_grad_norm_
    b1, b2 = betas
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    for k in live:
        g = grads[k] * coef
        m[k].mul_(b1).add_(g, alpha=1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].addcdiv_(m[k], denom, value=-lr / bc1)
    return float(total)


# ---------------------------------------------------------------------------
# parameter construction (A2 init formulae; used for synthetic benches/tests)
# ---------------------------------------------------------------------------
def param_shapes(cfg: dict) -> "Dict[str, Tuple[int, ...]]":
    """state_dict key -> shape, in the reference's registration order
    (model/model.py:98-127,197-222,356-376)."""
    S, R, Fr = cfg["scale_space_num"], cfg["res_depth"], cfg["featRoot"]
    k, nb = cfg["filter_size"], cfg.get("num_blocks", 3)
    out: Dict[str, Tuple[int, ...]] = {}
    for b in range(nb):
        cin = cfg["channels"] if b == 0 else cfg["n_class"]
        coupled = b > 0
        pd = f"msau_net.blocks.{b}.downsamplingblock"
        pu = f"msau_net.blocks.{b}.upsamplingblock"
        for l in range(S):
            c = Fr * 2 ** l
            for r in range(R):
                out[f"{pd}.conv_res_list.{l}.conv_res_list.{r}.custom_conv.weight"] = (c, c, k, k)
                out[f"{pd}.conv_res_list.{l}.conv_res_list.{r}.custom_conv.bias"] = (c,)
        last = cin
        for l in range(S):
            c = Fr * 2 ** l
            out[f"{pd}.conv1s.{l}.conv.weight"] = (c, last, k, k)
            out[f"{pd}.conv1s.{l}.conv.bias"] = (c,)
            last = c
        if coupled:
            for l in range(S):
                c = Fr * 2 ** l
                out[f"{pd}.conv1_1s.{l}.custom_conv.weight"] = (c, 2 * c, 1, 1)
                out[f"{pd}.conv1_1s.{l}.custom_conv.bias"] = (c,)
        C = Fr * 2 ** (S - 1)
        pa = f"{pd}.layer_attentions.attention_block"
        for nm, co in (("f", C // 8), ("g", C // 8), ("h", C)):
            out[f"{pa}.{nm}.conv.weight"] = (co, C, 1, 1)
            out[f"{pa}.{nm}.conv.bias"] = (co,)
        for l in range(S - 1):
            c = Fr * 2 ** l
            for r in range(R):
                out[f"{pu}.conv_res_list.{l}.conv_res_list.{r}.custom_conv.weight"] = (c, c, k, k)
                out[f"{pu}.conv_res_list.{l}.conv_res_list.{r}.custom_conv.bias"] = (c,)
        for l in range(S - 1):
            c = Fr * 2 ** l
            out[f"{pu}.conv1s.{l}.custom_conv.weight"] = (c, 2 * c, k, k)
            out[f"{pu}.conv1s.{l}.custom_conv.bias"] = (c,)
        if coupled:
            for l in range(S - 1):
                c = Fr * 2 ** l
                out[f"{pu}.conv1_1s.{l}.custom_conv.weight"] = (c, 2 * c, 1, 1)
                out[f"{pu}.conv1_1s.{l}.custom_conv.bias"] = (c,)
        for l in range(S - 1):
            c = Fr * 2 ** l
            out[f"{pu}.deconvs.{l}.conv.weight"] = (2 * c, c, k, k)
            out[f"{pu}.deconvs.{l}.conv.bias"] = (c,)
    for b in range(nb):
        out[f"msau_net.end_convs.{b}.custom_conv.weight"] = (cfg["n_class"], Fr, 4, 4)
        out[f"msau_net.end_convs.{b}.custom_conv.bias"] = (cfg["n_class"],)
    return out


def init_params(cfg: dict, seed: int = 0) -> Dict[str, Tensor]:
    """W ~ N(0, sqrt(2/(kh*kw*K2+K3))), b ~ N(0.1, 1e-5) with the kernel_shape each call
    site writes (layers.py:33-36,59-60,111-114,130-131,216,227-228); attention 1x1 convs keep
    torch's Conv2d default init (attention.py:19-21).  Statistics match the reference; the
    draw order does not (goldens ship weights instead)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    for key, shp in param_shapes(cfg).items():
        if key.endswith(".bias"):
            if ".attention_block." in key:
                wshape = param_shapes(cfg)[key[:-4] + "weight"]
                bound = 1.0 / math.sqrt(wshape[1] * wshape[2] * wshape[3])
                sd[key] = (torch.rand(shp, generator=g) * 2 - 1) * bound
            else:
                sd[key] = 0.1 + 1e-5 * torch.randn(shp, generator=g)
            continue
        if ".attention_block." in key:
            bound = 1.0 / math.sqrt(shp[1] * shp[2] * shp[3])       # kaiming_uniform(a=sqrt(5))
            sd[key] = (torch.rand(shp, generator=g) * 2 - 1) * bound
        elif ".deconvs." in key:
            # kernel_shape = [k, k, c, 2c] -> K2 = shp[1], K3 = shp[0]   (model.py:198-201)
            std = math.sqrt(2.0 / (shp[2] * shp[3] * shp[1] + shp[0]))
            sd[key] = std * torch.randn(shp, generator=g)
        else:
            std = math.sqrt(2.0 / (shp[2] * shp[3] * shp[1] + shp[0]))
            sd[key] = std * torch.randn(shp, generator=g)
    return sd


def synthetic_batch(B: int, C: int, H: int, W: int, n_class: int, seed: int,
                    dense: bool = False, occupancy: float = 0.3):
    """SURVEY 8(d) synthetic chargrids: one-hot with 30 % occupancy (or N(0,1) vectors at
    occupied pixels when `dense`), labels = occupied * U{1..n_class-1}."""
    g = torch.Generator().manual_seed(seed)
    occ = torch.rand((B, H, W), generator=g) < occupancy
    if dense:
        x = torch.randn((B, C, H, W), generator=g) * occ[:, None].float()
    else:
        ch = torch.randint(0, C, (B, H, W), generator=g)
        x = torch.zeros((B, C, H, W))
        x.scatter_(1, ch[:, None], occ[:, None].float())
    label = torch.randint(1, n_class, (B, H, W), generator=g) * occ.long()
    return x, label


def train_step(sd: Dict[str, Tensor], m: Dict[str, Tensor], v: Dict[str, Tensor], step: int,
               x: Tensor, label: Tensor, cfg: dict, lr: float = 1e-4):
    """One reference training step (train_chargrid_funsd_msau.py:46-59) on a batch.
    Returns (loss, logits, aux, grads, grad_norm).  `sd` tensors are updated in place."""
    leaves = {k: t.detach().clone().requires_grad_(True) for k, t in sd.items()}
    logits, aux = msau_forward(leaves, x, cfg)
    loss = msau_loss(logits, aux, label)
    loss.backward()
    grads = {k: leaves[k].grad for k in leaves}
    with torch.no_grad():
        gn = clip_adam_step(sd, grads, m, v, step, lr=lr)
    return float(loss), logits.detach(), (aux.detach() if aux is not None else None), grads, gn
