#!/usr/bin/env python3
"""The reference's own operating point: batch 1, a different H x W per document (train_chargrid_funsd_msau.py:45-59,
data_generator_funsd_bert.py:216-222).  N synthetic documents with FUNSD-like chargrid sizes, the engine loop, every
document's plan cached; reports, as one JSON object:

  first_epoch_ms_per_doc   epoch 0: every step builds its plan (buffers, descriptors, launch lists)
  docs_per_s / ms_per_doc  epochs >= 1: plans cached, steps back to back, one host sync at the end of the epoch
  host_ms_per_doc          host time to ENQUEUE a step (the loop is launch-bound when this is close to ms_per_doc)
  launches_per_step        native launch records of the median-size plan
  graph                    the same with TrainEngine(use_graph=True): one HIP-graph replay per cached plan

    python tools/funsd_loop.py [--docs 120] [--epochs 3] [--channels 64] [--dtype bf16] [--graph]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def doc_shapes(n, seed=0):
    """FUNSD forms are A4 scans; the chargrid cell is the smallest word box (data_generator_funsd_bert.py:153-155): grids of
    roughly 60-170 rows x 40-130 columns.  Distinct shapes, as in a real epoch (almost every document has its own)."""
    import random
    rng = random.Random(seed)
    out = set()
    while len(out) < n:
        out.add((rng.randint(60, 170), rng.randint(40, 130)))
    return sorted(out, key=lambda s: rng.random())


def run(args, use_graph):
    import torch
    from msau_amd import MSAUWrapper, TrainEngine
    dev = torch.device("cuda", 0)
    kw = dict(scale_space_num=4, res_depth=2, featRoot=8, filter_size=3, pool_size=2, final_act="softmax", num_blocks=3,
              dtype=args.dtype, seed=0)
    m = MSAUWrapper(args.channels, 5, kw).to(dev)
    shapes = doc_shapes(args.docs)
    m.max_cached_plans = 2 * len(shapes) + 2
    eng = TrainEngine(m, lr=1e-4, use_graph=use_graph)
    g = torch.Generator(device="cpu").manual_seed(1)
    docs = []
    for (H, W) in shapes:
        occ = torch.rand((1, H, W), generator=g) < 0.1
        ids = torch.randint(0, args.channels, (1, H, W), generator=g)
        x = torch.zeros((1, args.channels, H, W))
        x.scatter_(1, ids.unsqueeze(1), occ.unsqueeze(1).float())
        lab = (occ * torch.randint(1, 5, (1, H, W), generator=g)).long()
        docs.append((x.to(dev), lab.to(dev)))
    torch.cuda.synchronize()
    res = {}
    t0 = time.perf_counter()
    for x, lab in docs:
        eng.step(x, lab)
    torch.cuda.synchronize()
    res["first_epoch_ms_per_doc"] = round(1e3 * (time.perf_counter() - t0) / len(docs), 3)
    times, host = [], []
    for ep in range(args.epochs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = 0.0
        for x, lab in docs:
            h0 = time.perf_counter()
            loss = eng.step(x, lab)
            th += time.perf_counter() - h0
        he = time.perf_counter()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        times.append((t1 - t0) / len(docs))
        host.append(th / len(docs))
    best = min(times)
    res.update(docs=len(docs), epochs=args.epochs, docs_per_s=round(1.0 / best, 1), ms_per_doc=round(1e3 * best, 3),
               host_ms_per_doc=round(1e3 * min(host), 3), loss=round(float(loss), 4),
               plan_MB=round(sum(p.activation_bytes() for p in m._plans.values()) / 1e6, 1))
    med = sorted(shapes, key=lambda s: s[0] * s[1])[len(shapes) // 2]
    plan = m._plan_for_shape(1, med[0], med[1], dev, True)
    res["median_shape"] = list(med)
    res["launches_per_step"] = sum(seq[1] for seq in (plan._fwd_seq, plan._bwd_seq) if seq is not None) + 6   # + pack, convert, counts, CE, clip+Adam
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=120)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--graph", action="store_true", help="also run with one HIP-graph replay per cached plan")
    args = ap.parse_args()
    out = {"eager": run(args, False)}
    if args.graph:
        out["graph"] = run(args, True)
        out["graph_speedup"] = round(out["graph"]["docs_per_s"] / out["eager"]["docs_per_s"], 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
