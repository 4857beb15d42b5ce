#!/usr/bin/env python3
"""Reproducibility across fresh processes: one cold train step per process, exact fingerprints of every activation and
gradient, compared afterwards.  This is the tool that found the one-ulp finding of DESIGN.md section 2.

    for i in 1 2 3 4 5 6 7 8; do python tools/det_check.py run /tmp/r$i.pt; done; python tools/det_check.py cmp /tmp/r*.pt
    MSAU_DETERMINISTIC=1 ...      # the same with the deterministic mode of the plan
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def run(out):
    from msau_amd.model import MSAUWrapper, TrainEngine
    from tests.golden_util import load_net_case
    g, cfg, sd, x, label = load_net_case("net_cfg2_336x256x64")
    kw = dict(scale_space_num=cfg["scale_space_num"], res_depth=cfg["res_depth"], featRoot=cfg["featRoot"],
              filter_size=cfg["filter_size"], pool_size=cfg["pool_size"], final_act="softmax", num_blocks=cfg["num_blocks"], dtype="bf16")
    m = MSAUWrapper(cfg["channels"], cfg["n_class"], kw)
    m.load_state_dict(sd)
    m = m.cuda()
    eng = TrainEngine(m)
    loss = eng.step(x.cuda(), label.cuda())
    torch.cuda.synchronize()
    plan = m._plan_for(x.cuda(), True)
    bits = lambda t: (t.contiguous().view(torch.int16).to(torch.int64) * 31 + 7).remainder(1000003).sum().cpu()   # exact
    d = {"loss": loss.cpu(), "grad": eng.flat_grad.cpu()}
    for a in plan.acts:
        d["act:" + a.name] = bits(a.data)
        if a.grad is not None:
            d["grad:" + a.name] = bits(a.grad)
    torch.save(d, out)


def cmp(files):
    """group the processes into classes of bit-identical results; report every minority class against the largest one"""
    ds = [torch.load(f) for f in files]
    keys = list(ds[0])
    classes = []                                   # [(representative index, [member indices])]
    for i, d in enumerate(ds):
        for rep, members in classes:
            if all(torch.equal(ds[rep][k], d[k]) for k in keys):
                members.append(i)
                break
        else:
            classes.append((i, [i]))
    classes.sort(key=lambda c: -len(c[1]))
    ref = ds[classes[0][0]]
    print(f"{len(files)} processes, {len(classes)} distinct result(s); class sizes {[len(m) for _, m in classes]}")
    for rep, members in classes[1:]:
        d = ds[rep]
        bad = [k for k in keys if not torch.equal(ref[k], d[k])]
        print(f"  class of {[files[m] for m in members]}: {len(bad)} tensors differ from the majority; activations: "
              f"{[k for k in bad if k.startswith('act:')][:4]}; last gradients in backward order: {[k for k in bad if k.startswith('grad:')][-2:]}")
    print(f"{len(files) - len(classes[0][1])} of {len(files)} processes differ from the majority")


if __name__ == "__main__":
    run(sys.argv[2]) if sys.argv[1] == "run" else cmp(sys.argv[2:])
