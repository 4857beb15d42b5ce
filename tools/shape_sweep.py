#!/usr/bin/env python3
"""Robustness sweep: one train step and one forward-only prediction for a range of batch / image / channel sizes in both
storage types; every launch must fit its resources, losses must be finite and the two storage types must agree."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msau_amd.model import MSAUWrapper, TrainEngine
from oracle import msau_oracle as O

SHAPES = [(1, 70, 128, 60, 17), (3, 333, 251, 13, 5), (5, 200, 200, 64, 5), (2, 512, 384, 64, 5), (16, 168, 128, 64, 5),
          (8, 336, 256, 64, 5), (1, 1024, 768, 8, 3), (4, 97, 61, 32, 5), (2, 336, 256, 768, 5), (32, 128, 128, 32, 2)]
bad = 0
for B, H, W, C, ncls in SHAPES:
    x, label = O.synthetic_batch(B, C, H, W, ncls, seed=B + H)
    x, label = x.cuda(), label.cuda()
    res = {}
    for dtype in ("fp32", "bf16"):
        try:
            kw = dict(scale_space_num=4, res_depth=2, featRoot=8, final_act="softmax", num_blocks=3, dtype=dtype, seed=1)
            m = MSAUWrapper(C, ncls, kw).cuda()
            eng = TrainEngine(m)
            loss = eng.step(x, label)
            probs, amax = m.eval().predict_nhwc(inp=x)
            torch.cuda.synchronize()
            res[dtype] = (float(loss), float(probs.sum()) / (B * H * W))
            del eng, m
        except Exception as e:                                   # noqa: BLE001
            res[dtype] = ("FAIL", str(e)[:160])
            bad += 1
        torch.cuda.empty_cache()
    ok = all(isinstance(v[0], float) and v[0] == v[0] for v in res.values())
    agree = ok and abs(res["fp32"][0] - res["bf16"][0]) < 3e-2 * abs(res["fp32"][0]) and abs(res["bf16"][1] - 1) < 1e-2
    bad += 0 if agree else 1
    print(f"B={B:2d} {H}x{W}x{C} classes={ncls}:", {k: (round(v[0], 5) if isinstance(v[0], float) else v) for k, v in res.items()},
          "OK" if agree else "MISMATCH", flush=True)
print("problems:", bad)
sys.exit(1 if bad else 0)
