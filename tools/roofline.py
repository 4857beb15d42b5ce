#!/usr/bin/env python3
"""Algorithmic work of one MSAU training step, derived from the architecture alone.

Walks the layer list of the reference network (model/model.py:98-127 encoder, :197-222 decoder, :356-376 stages
and end convs; attention.py:138-162) and prints, per tile (= one [C,H,W] sample):

  * forward conv / deconv FLOPs, attention FLOPs, the train-step FLOPs  3*(conv_fwd + attn_live) - first_conv_fwd
  * the compulsory activation traffic (each conv reads its un-padded input once and writes its output once,
    everything else fused; backward charged 2x forward), in elements and bytes
  * the resulting upper bounds in tiles/s for the HBM and MFMA peaks of one MI355X.

Known answers for BASELINE.json configs[1] (checked by tests/test_host_cpu.py): 98 convs incl. 9 deconvs, 636 167
parameters, conv_fwd 8.434 GF, attn_live 0.520 GF, first conv 0.793 GF, train 26.07 GF/tile, 76.75 M activation
elements => 460 MB/tile fwd+bwd at bf16.
"""
from __future__ import annotations

import argparse
import json
from dataclasses import dataclass
from typing import Dict, List

HBM_PEAK = 8.0e12          # B/s   (MI355X_MICROARCH.md: HBM3E ~8 TB/s)
MFMA_PEAK_BF16 = 2.5e15    # FLOP/s dense bf16

CONFIGS: Dict[str, dict] = {
    "cfg1": dict(H=128, W=128, channels=32, num_blocks=1),
    "cfg2": dict(H=336, W=256, channels=64, num_blocks=3),
    "cfg4": dict(H=336, W=256, channels=768, num_blocks=2),
    "cfg5-plain": dict(H=512, W=384, channels=64, num_blocks=3),
}
DEFAULTS = dict(n_class=5, featRoot=8, scale_space_num=4, res_depth=2, filter_size=3)


@dataclass
class Conv:
    name: str
    cin: int
    cout: int
    k: int
    hin: int
    win: int
    hout: int
    wout: int
    transposed: bool = False
    live: bool = True      # False: last-stage attention convs, no gradient

    @property
    def flops(self) -> int:
        # a transposed conv scatters every input pixel through the k*k taps; a conv gathers per output pixel
        px = self.hin * self.win if self.transposed else self.hout * self.wout
        return 2 * px * self.cin * self.cout * self.k * self.k

    @property
    def elems(self) -> int:
        return self.hin * self.win * self.cin + self.hout * self.wout * self.cout

    @property
    def params(self) -> int:
        return self.cin * self.cout * self.k * self.k + self.cout


def level_sizes(H: int, W: int, S: int):
    out = [(H, W)]
    for _ in range(S - 1):
        H, W = (H + 1) // 2, (W + 1) // 2            # SAME 2x2/2 max-pool: ceil
        out.append((H, W))
    return out


def conv_list(cfg: dict) -> List[Conv]:
    c = dict(DEFAULTS, **cfg)
    S, R, F, k, nb, ncls = c["scale_space_num"], c["res_depth"], c["featRoot"], c["filter_size"], c["num_blocks"], c["n_class"]
    hw = level_sizes(c["H"], c["W"], S)
    convs: List[Conv] = []

    def add(name, cin, cout, kk, l, **kw):
        h, w = hw[l]
        convs.append(Conv(name, cin, cout, kk, h, w, h, w, **kw))

    for b in range(nb):
        last = c["channels"] if b == 0 else ncls
        for l in range(S):
            cl = F << l
            add(f"s{b}.enc{l}.dil", last, cl, k, l)
            for r in range(R):
                add(f"s{b}.enc{l}.res{r}", cl, cl, k, l)
            if b > 0:
                add(f"s{b}.enc{l}.couple", 2 * cl, cl, 1, l)
            last = cl
        cb = F << (S - 1)
        live = b < nb - 1
        for nm, co in (("f", cb // 8), ("g", cb // 8), ("h", cb)):
            add(f"s{b}.attn.{nm}", cb, co, 1, S - 1, live=live)
        for l in range(S - 2, -1, -1):
            cl = F << l
            (hi, wi), (ho, wo) = hw[l + 1], hw[l]
            convs.append(Conv(f"s{b}.dec{l}.deconv", 2 * cl, cl, k, hi, wi, ho, wo, transposed=True))
            add(f"s{b}.dec{l}.merge", 2 * cl, cl, k, l)
            for r in range(R):
                add(f"s{b}.dec{l}.res{r}", cl, cl, k, l)
            if b > 0:
                add(f"s{b}.dec{l}.couple", 2 * cl, cl, 1, l)
        add(f"s{b}.end", F, ncls, 4, 0)
    return convs


def attention_flops(cfg: dict) -> Dict[str, float]:
    """s = g^T f (N x N x C/8) and o = h beta (C x N x N) per stage (attention.py:149-157)."""
    c = dict(DEFAULTS, **cfg)
    S, F, nb = c["scale_space_num"], c["featRoot"], c["num_blocks"]
    h, w = level_sizes(c["H"], c["W"], S)[-1]
    N, C = h * w, F << (S - 1)
    per_stage = 2 * N * N * (C // 8) + 2 * N * N * C
    return {"per_stage": per_stage, "all": per_stage * nb, "live": per_stage * (nb - 1), "N": N, "C": C}


def work(cfg: dict, act_bytes: int = 2) -> dict:
    convs = conv_list(cfg)
    attn = attention_flops(cfg)
    conv_fwd = sum(cv.flops for cv in convs)
    first = convs[0].flops
    elems = sum(cv.elems for cv in convs)
    train = 3 * (conv_fwd + attn["live"]) - first
    bytes_fwd = elems * act_bytes
    bytes_train = 3 * bytes_fwd
    return {
        "convs": len(convs), "deconvs": sum(cv.transposed for cv in convs),
        "params": sum(cv.params for cv in convs),
        "conv_fwd_flops": conv_fwd, "attn_all_flops": attn["all"], "attn_live_flops": attn["live"],
        "first_conv_flops": first, "train_flops": train,
        "act_elems": elems, "bytes_fwd": bytes_fwd, "bytes_train": bytes_train,
        "intensity_flop_per_byte": train / bytes_train,
        "bound": "hbm" if train / bytes_train < MFMA_PEAK_BF16 / HBM_PEAK else "mfma",
        "tiles_per_s_hbm": HBM_PEAK / bytes_train, "tiles_per_s_mfma": MFMA_PEAK_BF16 / train,
    }


def achieved(cfg: dict, tiles_per_s: float, act_bytes: int = 2) -> dict:
    w = work(cfg, act_bytes)
    return {"hbm_frac": tiles_per_s * w["bytes_train"] / HBM_PEAK, "mfma_frac": tiles_per_s * w["train_flops"] / MFMA_PEAK_BF16}


def main() -> None:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--act-bytes", type=int, default=2, help="bytes per activation element (2 = bf16 storage)")
    ap.add_argument("--tiles-per-s", type=float, default=None, help="also print the achieved whole-step fractions")
    ap.add_argument("--layers", action="store_true", help="print the per-layer table")
    a = ap.parse_args()
    cfg = CONFIGS[a.config]
    if a.layers:
        for cv in conv_list(cfg):
            print(f"{cv.name:18s} {cv.cin:4d}->{cv.cout:<4d} k{cv.k} {cv.hin}x{cv.win}->{cv.hout}x{cv.wout} "
                  f"{cv.flops / 1e6:8.1f} MF {cv.elems / 1e6:7.3f} Melem{'' if cv.live else '  (dead)'}")
    w = work(cfg, a.act_bytes)
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in w.items()}, indent=1))
    print(f"peaks: HBM {HBM_PEAK / 1e12:.1f} TB/s, MFMA bf16 {MFMA_PEAK_BF16 / 1e15:.2f} PFLOP/s (dense); "
          f"ridge {MFMA_PEAK_BF16 / HBM_PEAK:.0f} flop/B")
    if a.tiles_per_s:
        print(json.dumps(achieved(cfg, a.tiles_per_s, a.act_bytes)))


if __name__ == "__main__":
    main()
