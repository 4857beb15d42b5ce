#!/usr/bin/env python3
"""Forward-only throughput of the inference path: MSAUWrapper.predict_nhwc at the bench size and at
the size KVModel really runs at (text lines scaled to 3 px), dense input vs the device-painted id mask.
Not the headline metric (bench.py is); numbers are quoted in DESIGN.md."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from msau_amd.model import MSAUWrapper


def run(B, C, H, W, n_class, dtype, use_ids, iters, warmup, graph=False):
    m = MSAUWrapper(C, n_class, dict(featRoot=8, scale_space_num=4, res_depth=2, filter_size=3, pool_size=2,
                                     final_act="softmax", dtype=dtype, seed=0)).cuda().eval()
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, C, (B, H, W), generator=g, dtype=torch.int32)
    if use_ids:
        arg = dict(ids=ids.cuda())
    else:
        arg = dict(inp=torch.nn.functional.one_hot(ids.long(), C).permute(0, 3, 1, 2).float().cuda())
    for _ in range(warmup):
        m.predict_nhwc(graph=graph, **arg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        m.predict_nhwc(graph=graph, **arg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    plan = next(iter(m._plans.values()))
    return {"B": B, "C": C, "H": H, "W": W, "n_class": n_class, "dtype": dtype, "input": "ids" if use_ids else "dense",
            "ms_per_call": round(dt * 1e3, 3), "tiles_per_s": round(B / dt, 1), "head_fused": plan.head_fused,
            "activation_MB": round(sum(b.numel() * b.element_size() for b in plan.buffers) / 1e6, 1),
            "launches": plan._fwd_seq[1], "graph": graph}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    for cfg in ((16, 64, 336, 256, 5, "bf16"), (1, 64, 336, 256, 5, "bf16"), (1, 60, 70, 128, 17, "bf16"), (1, 60, 70, 128, 17, "fp32")):
        for use_ids, graph in ((False, False), (True, False), (True, True)):
            print(json.dumps(run(*cfg, use_ids, a.iters, a.warmup, graph)), flush=True)


if __name__ == "__main__":
    main()
