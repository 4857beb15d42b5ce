"""CPU restatement of the box-convolution variant of MSAU  --  TEST INFRASTRUCTURE ONLY.

**PARITY UNPINNED.**  `model/model_box.py` (reference) builds its residual blocks from `box_convolution.BoxConv2d`
(github shrubb/box-convolutions; no version pinned anywhere in the reference, not in its requirements.txt, not vendored,
not installed in this image) and the reference tree holds no tests or fixtures for that path, so nothing here could be
checked against the reference's own arithmetic.  What follows restates

  * the network skeleton from the reference's own file -- model/model_box.py:51-59 (MultiBoxConvBlock.forward),
    :129-157 (encoder), :224-247 (decoder), :341-358 (BMSAUNet.forward) -- reusing the pinned pieces of
    oracle/msau_oracle.py (SAME convs, dilated conv + LRN, transposed conv, attention, pooling), and
  * the box filter from its published definition (Burkov & Lempitsky, "Deep Neural Networks with Box Convolutions",
    NeurIPS 2018): a normalised integral of the (piecewise-constant, zero-extended) image over a real-valued box,

and is the checker for msau_amd's HIP kernels (csrc/boxconv.hip): "self-consistent only".  Conventions chosen here
(they are the documented semantics of msau_amd.BMSAUWrapper):
    out[b, c*F + f, y, x] = 1/A * integral_{rows [y+hmin, y+hmax+1)} integral_{cols [x+wmin, x+wmax+1)} in[b, c]
    A = (hmax - hmin + 1) (wmax - wmin + 1);  (hmin, hmax, wmin, wmax) = stored parameter x max box size;
    stored parameters `x_min, x_max` (rows) and `y_min, y_max` (columns), each [in_planes, num_filters], as BoxConv2d
    names them; before use the box is clamped to |edge| <= max size and max >= min.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from . import msau_oracle as O

DEFAULT_BOX_CFG = dict(O.DEFAULT_CFG, num_box_convs=3, num_box_per_channels=3, max_box_sizes=28)


def box_pixels(sd: Dict[str, Tensor], prefix: str, max_h: float, max_w: float):
    """stored parameters -> valid boxes in pixels (differentiable: clamp passes the gradient inside the range)"""
    hmin = (sd[prefix + ".x_min"] * max_h).clamp(-max_h, max_h)
    hmax = (sd[prefix + ".x_max"] * max_h).clamp(-max_h, max_h)
    wmin = (sd[prefix + ".y_min"] * max_w).clamp(-max_w, max_w)
    wmax = (sd[prefix + ".y_max"] * max_w).clamp(-max_w, max_w)
    hmax = torch.where(hmax >= hmin, hmax, hmin)
    wmax = torch.where(wmax >= wmin, wmax, wmin)
    return hmin, hmax, wmin, wmax


def _overlap(n: int, lo: Tensor, hi: Tensor) -> Tensor:
    """[..., n_out = n, n_in = n]: length of [u, u+1) intersected with [y + lo, y + hi) for output y, input u"""
    y = torch.arange(n, dtype=lo.dtype).view(n, 1)
    u = torch.arange(n, dtype=lo.dtype).view(1, n)
    a = y + lo[..., None, None]
    b = y + hi[..., None, None]
    return (torch.minimum(u + 1, b) - torch.maximum(u, a)).clamp_min(0)


def box_conv(x: Tensor, hmin: Tensor, hmax: Tensor, wmin: Tensor, wmax: Tensor) -> Tensor:
    """x [B,C,H,W]; boxes [C,F] in pixels -> [B, C*F, H, W] (channel c*F + f), normalised box integrals"""
    B, C, H, W = x.shape
    Fn = hmin.shape[1]
    Wh = _overlap(H, hmin, hmax + 1)                          # [C,F,H,H]
    Ww = _overlap(W, wmin, wmax + 1)                          # [C,F,W,W]
    area = (hmax - hmin + 1) * (wmax - wmin + 1)
    # out[b,c,f,y,x] = sum_u sum_v Wh[c,f,y,u] x[b,c,u,v] Ww[c,f,x,v]
    t = torch.einsum("cfyu,bcuv->bcfyv", Wh, x)
    out = torch.einsum("bcfyv,cfxv->bcfyx", t, Ww) / area[None, :, :, None, None]
    return out.reshape(B, C * Fn, H, W)


def box_block(x: Tensor, sd: Dict[str, Tensor], prefix: str, cfg: dict) -> Tensor:
    """MultiBoxConvBlock.forward, model/model_box.py:51-59"""
    n, mb = cfg["num_box_convs"], float(cfg["max_box_sizes"])
    r = torch.relu(x)
    for i in range(n):
        hmin, hmax, wmin, wmax = box_pixels(sd, f"{prefix}.conv_list.{2 * i}", mb, mb)
        r = box_conv(r, hmin, hmax, wmin, wmax)
        w, b = sd[f"{prefix}.conv_list.{2 * i + 1}.custom_conv.weight"], sd[f"{prefix}.conv_list.{2 * i + 1}.custom_conv.bias"]
        r = O.conv_same(r, w, b, relu=(i < n - 1))
    return torch.relu(r + x)


def stage_forward(inp, sd, b, cfg, prev_dw, prev_up, need_attention=True):
    """BUNetBlock: model/model_box.py:129-157 (encoder), :224-247 (decoder) -- msau_oracle.stage_forward with box blocks"""
    S, k, ps = cfg["scale_space_num"], cfg["filter_size"], cfg["pool_size"]
    coupled = b > 0
    pd = f"msau_net.blocks.{b}.downsamplingblock"
    pu = f"msau_net.blocks.{b}.upsamplingblock"
    dw, x_in, x = {}, inp, inp
    for l in range(S):
        x = O.dilconv_lrn(x_in, sd[f"{pd}.conv1s.{l}.conv.weight"], sd[f"{pd}.conv1s.{l}.conv.bias"], 2 ** l)
        x = box_block(x, sd, f"{pd}.conv_box_list.{l}", cfg)
        if coupled:
            x = O.conv_same(torch.cat([prev_dw[l], x], dim=1), sd[f"{pd}.conv1_1s.{l}.custom_conv.weight"],
                            sd[f"{pd}.conv1_1s.{l}.custom_conv.bias"], relu=True)
        if l > S - 2:
            dw[l] = O.self_attention(x, sd, f"{pd}.layer_attentions.attention_block") if need_attention else x
        else:
            dw[l] = x
        if l < S - 1:
            x_in = F.max_pool2d(O.pad_same(x, ps, ps, ps, ps), ps, ps)
    cur, up = x, {}
    for l in range(S - 2, -1, -1):
        d = O.deconv(cur, sd[f"{pu}.deconvs.{l}.conv.weight"], sd[f"{pu}.deconvs.{l}.conv.bias"], tuple(dw[l].shape[2:]))
        x = O.conv_same(torch.cat([dw[l], d], dim=1), sd[f"{pu}.conv1s.{l}.custom_conv.weight"], sd[f"{pu}.conv1s.{l}.custom_conv.bias"])
        x = box_block(x, sd, f"{pu}.conv_box_list.{l}", cfg)
        if coupled:
            x = O.conv_same(torch.cat([prev_up[l], x], dim=1), sd[f"{pu}.conv1_1s.{l}.custom_conv.weight"],
                            sd[f"{pu}.conv1_1s.{l}.custom_conv.bias"], relu=True)
        up[l] = x
        cur = x
    return cur, dw, up


def bmsau_forward(sd: Dict[str, Tensor], inp: Tensor, cfg: dict) -> Tuple[Tensor, Optional[Tensor]]:
    """BMSAUNet.forward, model/model_box.py:341-358"""
    nb = cfg.get("num_blocks", 3)
    prev_dw = prev_up = None
    aux, x, out = None, inp, None
    for b in range(nb):
        out, prev_dw, prev_up = stage_forward(x, sd, b, cfg, prev_dw, prev_up, need_attention=(b < nb - 1))
        out = O.conv_same(out, sd[f"msau_net.end_convs.{b}.custom_conv.weight"], sd[f"msau_net.end_convs.{b}.custom_conv.bias"])
        x = out
        if b == nb - 2:
            aux = out
    return out, aux


def param_shapes(cfg: dict) -> "Dict[str, Tuple[int, ...]]":
    """state_dict key -> shape of BMSAUWrapper (model/model_box.py:109-131,195-218,320-339; BoxConv2d holds four
    [in_planes, num_filters] parameters x_min / x_max / y_min / y_max)"""
    S, Fr, k, nb = cfg["scale_space_num"], cfg["featRoot"], cfg["filter_size"], cfg.get("num_blocks", 3)
    n, Fn = cfg["num_box_convs"], cfg["num_box_per_channels"]
    out: Dict[str, Tuple[int, ...]] = {}

    def block(prefix, c):
        for i in range(n):
            for nm in ("x_min", "x_max", "y_min", "y_max"):
                out[f"{prefix}.conv_list.{2 * i}.{nm}"] = (c, Fn)
            out[f"{prefix}.conv_list.{2 * i + 1}.custom_conv.weight"] = (c, Fn * c, 1, 1)
            out[f"{prefix}.conv_list.{2 * i + 1}.custom_conv.bias"] = (c,)

    for b in range(nb):
        cin = cfg["channels"] if b == 0 else cfg["n_class"]
        pd = f"msau_net.blocks.{b}.downsamplingblock"
        pu = f"msau_net.blocks.{b}.upsamplingblock"
        for l in range(S):
            block(f"{pd}.conv_box_list.{l}", Fr * 2 ** l)
        last = cin
        for l in range(S):
            c = Fr * 2 ** l
            out[f"{pd}.conv1s.{l}.conv.weight"] = (c, last, k, k)
            out[f"{pd}.conv1s.{l}.conv.bias"] = (c,)
            last = c
        if b > 0:
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
            out[f"{pu}.conv1s.{l}.custom_conv.weight"] = (c, 2 * c, k, k)
            out[f"{pu}.conv1s.{l}.custom_conv.bias"] = (c,)
        if b > 0:
            for l in range(S - 1):
                c = Fr * 2 ** l
                out[f"{pu}.conv1_1s.{l}.custom_conv.weight"] = (c, 2 * c, 1, 1)
                out[f"{pu}.conv1_1s.{l}.custom_conv.bias"] = (c,)
        for l in range(S - 1):
            c = Fr * 2 ** l
            out[f"{pu}.deconvs.{l}.conv.weight"] = (2 * c, c, k, k)
            out[f"{pu}.deconvs.{l}.conv.bias"] = (c,)
        for l in range(S - 1):
            block(f"{pu}.conv_box_list.{l}", Fr * 2 ** l)
    for b in range(nb):
        out[f"msau_net.end_convs.{b}.custom_conv.weight"] = (cfg["n_class"], Fr, 4, 4)
        out[f"msau_net.end_convs.{b}.custom_conv.bias"] = (cfg["n_class"],)
    return out


def init_params(cfg: dict, seed: int = 0) -> Dict[str, Tensor]:
    """conv / attention parameters as msau_oracle.init_params; boxes: centre ~ U(-1/4, 1/4), half extent ~ U(1/28, 1/4) of the
    max box size (stored units).  The third-party package's own initialiser is unavailable (unpinned)."""
    g = torch.Generator().manual_seed(seed)
    shapes = param_shapes(cfg)
    sd: Dict[str, Tensor] = {}
    pending: Dict[str, Tuple[Tensor, Tensor]] = {}
    for key, shp in shapes.items():
        leaf = key.rsplit(".", 1)[1]
        if leaf in ("x_min", "x_max", "y_min", "y_max"):
            base, axis = key.rsplit(".", 1)[0], leaf[0]
            if (base, axis) not in pending:
                centre = (torch.rand(shp, generator=g) - 0.5) * 0.5
                half = 1.0 / 28 + torch.rand(shp, generator=g) * (0.25 - 1.0 / 28)
                pending[(base, axis)] = (centre - half, centre + half)
            lo, hi = pending[(base, axis)]
            sd[key] = lo if leaf.endswith("min") else hi
        elif key.endswith(".bias"):
            if ".attention_block." in key:
                w = shapes[key[:-4] + "weight"]
                sd[key] = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(w[1] * w[2] * w[3])
            else:
                sd[key] = 0.1 + 1e-5 * torch.randn(shp, generator=g)
        elif ".attention_block." in key:
            sd[key] = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(shp[1] * shp[2] * shp[3])
        else:
            sd[key] = math.sqrt(2.0 / (shp[2] * shp[3] * shp[1] + shp[0])) * torch.randn(shp, generator=g)
    return sd
