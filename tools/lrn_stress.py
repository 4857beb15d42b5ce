#!/usr/bin/env python3
"""Does msau_lrn_bwd give the same bits when another kernel runs beside it?  Fixed inputs (taken from a real train step),
repeated launches on the main stream with / without a weight-gradient kernel running concurrently on a side stream."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msau_amd import _lib as L
from msau_amd.model import MSAUWrapper, TrainEngine
from msau_amd.plan import ConvOp
from tests.golden_util import load_net_case

g, cfg, sd, x, label = load_net_case("net_cfg2_336x256x64")
kw = dict(scale_space_num=cfg["scale_space_num"], res_depth=cfg["res_depth"], featRoot=cfg["featRoot"], filter_size=cfg["filter_size"],
          pool_size=cfg["pool_size"], final_act="softmax", num_blocks=cfg["num_blocks"], dtype="bf16")
m = MSAUWrapper(cfg["channels"], cfg["n_class"], kw)
m.load_state_dict(sd)
m = m.cuda()
eng = TrainEngine(m)
eng.step(x.cuda(), label.cuda())
torch.cuda.synchronize()
plan = m._plan_for(x.cuda(), True)
acts = {a.name: a for a in plan.acts}
a_, y_ = acts["s2.d0.a"], acts["s2.d0.lrn"]
convs = [op for op in plan.ops if isinstance(op, ConvOp) and op.wdesc is not None and op.name.startswith("s2.d0")]
main = torch.cuda.current_stream()
side = L.concurrent_stream(torch.device("cuda", 0))
out = torch.empty_like(a_.grad)


def lrn(dst):
    L.call("msau_lrn_bwd", main.cuda_stream, L.BF16, a_.data.data_ptr(), y_.grad.data_ptr(), dst.data_ptr(), a_.npix, a_.C, a_.Cs, a_.C, 1e-4, 0.75, 1.0)


ref = torch.empty_like(out)
lrn(ref)
torch.cuda.synchronize()
for mode in ("alone", "beside wgrad"):
    bad = 0
    for it in range(200):
        if mode != "alone":
            for op in convs:
                L.call("msau_conv2d_wgrad", side.cuda_stream, L.BF16, C.byref(op.wdesc))
        lrn(out)
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            if bad <= 3:
                d = (out.float() != ref.float())
                print("   iter", it, "elements differing", int(d.sum()))
    print(mode, ": differing repeats", bad, "of 200")
