"""`UNetLoss` counterpart (reference: model/training/cost.py:6-65): plain cross entropy over every pixel on
one-hot targets, 0.5 * final + 0.5 * auxiliary when auxiliary logits are given, accuracy over pixels whose
target class is not 0.  The CE and its gradient are one HIP kernel (`msau_softmax_ce`)."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib as L


class _SoftmaxCE(torch.autograd.Function):
    """mean over all B*H*W pixels of -log softmax(logits)[target] on NCHW fp32 logits"""

    @staticmethod
    def forward(ctx, logits, target):
        B, C, H, W = logits.shape
        dev = logits.device
        Cs = -(-C // 8) * 8
        s = torch.cuda.current_stream().cuda_stream
        lg = logits.contiguous().float()
        tg = target.reshape(B, H, W).contiguous().long()
        nhwc = torch.empty((B, H, W, Cs), dtype=torch.float32, device=dev)
        dn = torch.empty_like(nhwc)
        loss = torch.zeros((1,), dtype=torch.float32, device=dev)
        ws = torch.zeros((int(L.load().msau_ce_ws_floats(B * H * W)),), dtype=torch.float32, device=dev)
        L.call("msau_nchw_to_nhwc", s, L.F32, lg.data_ptr(), nhwc.data_ptr(), B, C, Cs, H, W)
        L.call("msau_softmax_ce", s, L.F32, nhwc.data_ptr(), tg.data_ptr(), dn.data_ptr(), loss.data_ptr(), ws.data_ptr(),
               B, H * W, C, Cs, 1.0 / (B * H * W))
        g = torch.empty((B, C, H, W), dtype=torch.float32, device=dev)
        L.call("msau_nhwc_to_nchw", s, L.F32, dn.data_ptr(), g.data_ptr(), B, C, Cs, H, W)
        ctx.save_for_backward(g)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        return g * go, None


class _WeightedSoftmaxCE(torch.autograd.Function):
    """torch.nn.CrossEntropyLoss(weight) on NCHW fp32 logits (model/training/cost.py:24-31): sum_p w[t_p] nll_p / sum_p w[t_p].
    One pass writes the un-normalised gradient and both sums (`msau_softmax_ce_weighted`); the denominator, known only after
    it, is applied where autograd multiplies by the incoming gradient anyway."""

    @staticmethod
    def forward(ctx, logits, target, class_w):
        B, C, H, W = logits.shape
        dev = logits.device
        Cs = -(-C // 8) * 8
        s = torch.cuda.current_stream().cuda_stream
        lg = logits.contiguous().float()
        tg = target.reshape(B, H, W).contiguous().long()
        nhwc = torch.empty((B, H, W, Cs), dtype=torch.float32, device=dev)
        dn = torch.empty_like(nhwc)
        sums = torch.zeros((2,), dtype=torch.float32, device=dev)
        ws = torch.zeros((int(L.load().msau_ce_ws_floats(B * H * W)),), dtype=torch.float32, device=dev)
        L.call("msau_nchw_to_nhwc", s, L.F32, lg.data_ptr(), nhwc.data_ptr(), B, C, Cs, H, W)
        L.call("msau_softmax_ce_weighted", s, L.F32, nhwc.data_ptr(), tg.data_ptr(), class_w.data_ptr(), dn.data_ptr(),
               sums.data_ptr(), ws.data_ptr(), B, H * W, C, Cs)
        g = torch.empty((B, C, H, W), dtype=torch.float32, device=dev)
        L.call("msau_nhwc_to_nchw", s, L.F32, dn.data_ptr(), g.data_ptr(), B, C, Cs, H, W)
        ctx.save_for_backward(g, sums)
        return sums[0] / sums[1]

    @staticmethod
    def backward(ctx, go):
        g, sums = ctx.saved_tensors
        return g * (go / sums[1]), None, None


class UNetLoss(torch.nn.Module):
    def __init__(self, kwargs):
        super().__init__()
        self.cost_name = kwargs.get("cost_name", "cross_entropy")
        self.act_name = kwargs.get("act_name", "softmax")
        self.class_weights = kwargs.get("class_weights", None)
        if self.class_weights is not None and self.cost_name == "cross_entropy":
            # the reference keeps them as a CPU tensor inside torch.nn.CrossEntropyLoss (cost.py:26-29); here a buffer that
            # follows the module to the device
            self.register_buffer("class_weights_torch", torch.from_numpy(np.array(self.class_weights, dtype=np.float32)))

    def _ce(self, logits, tgt_idx):
        if self.class_weights is None:
            return _SoftmaxCE.apply(logits, tgt_idx)
        cw = self.class_weights_torch.to(logits.device)
        assert cw.numel() == logits.shape[1], "class_weights: one weight per class"
        return _WeightedSoftmaxCE.apply(logits, tgt_idx, cw.contiguous())

    def forward(self, logits, tgt, kwargs):
        """-> (acc, loss, final_loss) exactly as the reference; `tgt` / `aux_tgt` are one-hot [B,C,H,W]"""
        aux_logits = kwargs.get("aux_logits", None)
        aux_tgt = kwargs.get("aux_tgt", None)
        if self.cost_name != "cross_entropy":
            return torch.softmax(logits, dim=1) if self.act_name == "softmax" else logits
        tgt = torch.argmax(tgt, dim=1)
        with torch.no_grad():
            pred = torch.argmax(logits, dim=1)
            nz = tgt != 0
            acc = float((pred[nz] == tgt[nz]).sum()) / max(int(nz.sum()), 1) if bool(nz.any()) else float("nan")
        loss_map = self._ce(logits, tgt)
        if aux_logits is not None:
            aux = self._ce(aux_logits, torch.argmax(aux_tgt, dim=1))
            return acc, 0.5 * loss_map + 0.5 * aux, loss_map
        return acc, loss_map, None
