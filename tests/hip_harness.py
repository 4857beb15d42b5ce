"""Run small custom op graphs through the C ABI (via msau_amd.plan) for the op-level parity tests."""
import math

import torch

from msau_amd import _lib as L
from msau_amd.plan import Act, AttnCoreOp, ConvOp, LrnOp, Plan, PoolOp  # noqa: F401

DEV = "cuda:0"


def run_graph(builder, params, x, gy, dtype=L.F32, gy2=None):
    """builder(plan) builds ops from plan.x_in and sets plan.logits (and optionally plan.aux).
    params: name -> CPU tensor.  Returns (y, y2, dx, {name: grad}) as CPU fp32 tensors."""
    poff, pshape, off = {}, {}, 0
    for k, v in params.items():
        poff[k], pshape[k] = off, tuple(v.shape)
        off += -(-v.numel() // 4) * 4
    flat = torch.zeros(max(off, 4), dtype=torch.float32)
    for k, v in params.items():
        flat[poff[k]:poff[k] + v.numel()] = v.reshape(-1).float()
    flat = flat.to(DEV)
    x = torch.as_tensor(x).float().contiguous().to(DEV)
    B, Cc, H, W = x.shape
    plan = Plan(dict(channels=Cc, input_grad=True), B, H, W, dtype, torch.device(DEV), poff, pshape,
                training=True, builder=builder)
    y, y2 = plan.forward(flat, x)
    y, y2 = y.clone(), (y2.clone() if y2 is not None else None)
    plan.set_external_grads(torch.as_tensor(gy).float().to(DEV), None if gy2 is None else torch.as_tensor(gy2).float().to(DEV))
    fg = torch.zeros_like(flat)
    plan.backward(fg)
    dx = plan.input_grad_nchw() if plan.x_in.grad is not None else None
    torch.cuda.synchronize()
    grads = {k: fg[poff[k]:poff[k] + math.prod(pshape[k])].view(pshape[k]).cpu() for k in params}
    return y.cpu(), (y2.cpu() if y2 is not None else None), (dx.cpu() if dx is not None else None), grads
