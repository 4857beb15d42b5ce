"""Static execution plan of the MSAU forward / backward for one (batch, H, W) shape.

The plan is the MI355X-side counterpart of the reference's module graph
(model/model.py:129-164 encoder, :224-259 decoder, :328-344 stage, :378-396 net): it is built once
per shape, owns every activation / gradient buffer (NHWC, padded to 8 channels, fp32 or bf16
storage), and is a flat list of kernel launches through the C ABI of libmsau_hip.so.  Because the
launch sequence and all pointers are static it can be captured into a HIP graph and replayed.

Gradient bookkeeping: every activation knows how many consumers contribute to its gradient.
Contributions run in reverse forward order; the first one to run writes, later ones accumulate
(MSAU_CONV_ACCUM) and the last one multiplies by the ReLU mask of the activation's producer
(MSAU_CONV_MASK_B), so that a gradient buffer always ends up holding dL/d(pre-activation).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L


def same_pads(in_size: int, k: int, stride: int = 1, dilation: int = 1) -> Tuple[int, int]:
    """TF "SAME" padding (before, after) -- host restatement of model/layers/utils.py:5-28."""
    k_eff = k + (k - 1) * (dilation - 1)
    out = -(-in_size // stride)
    pad = max((out - 1) * stride + k_eff - in_size, 0)
    return pad // 2, pad - pad // 2


def _ru(a: int, b: int) -> int:
    return -(-a // b) * b


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class Act:
    """An activation tensor [B,H,W,Cs] (+ its gradient buffer)."""

    def __init__(self, plan: "Plan", name: str, H: int, W: int, C: int, relu_out: bool = False,
                 needs_grad: bool = True):
        self.plan, self.name, self.H, self.W, self.C = plan, name, H, W, C
        self.Cs = _ru(C, 8)
        self.relu_out = relu_out
        self.needs_grad = needs_grad and plan.training
        # forward-only plans get their buffers from Plan._assign_buffers (liveness-based reuse), after the graph is known
        self.data = None if plan.reuse else torch.zeros((plan.B, H, W, self.Cs), dtype=plan.tdtype, device=plan.device)
        self.grad: Optional[torch.Tensor] = None
        self.n_contrib = 0
        plan.acts.append(self)

    @property
    def npix(self) -> int:
        return self.plan.B * self.H * self.W

    def register(self) -> int:
        k = self.n_contrib
        self.n_contrib += 1
        return k

    def slot_flags(self, k: int) -> Tuple[bool, bool]:
        """(accumulate, apply producer's ReLU mask) for contribution k (forward-order index)."""
        return k != self.n_contrib - 1, (k == 0 and self.relu_out)

    def alloc_grad(self):
        if self.needs_grad and self.n_contrib > 0 and self.grad is None:
            self.grad = torch.zeros_like(self.data)


class Op:
    name = "op"

    def reads(self) -> List["Act"]:
        return []

    def writes(self) -> List["Act"]:
        return []

    def finalize(self):
        pass

    def note(self):
        pass

    def fwd(self, s):
        raise NotImplementedError

    def bwd(self, s):
        pass


class ConvOp(Op):
    """SAME conv (3x3 / dilated / 1x1 / 4x4; layers.py:82-102,152-164) or transposed conv
    (layers.py:249-250), with the concat of its two sources, input ReLU, bias, residual add and
    output ReLU fused."""

    def __init__(self, plan: "Plan", name: str, x1: Act, x2: Optional[Act], wname: str, bname: str, out: Act,
                 k: int, dil: int = 1, relu_in: bool = False, relu_out: bool = False,
                 fwd_add: Optional[Act] = None, kind: str = "conv", head: bool = False):
        self.plan, self.name, self.x1, self.x2, self.out = plan, name, x1, x2, out
        self.head = head            # forward-only: also emit softmax probabilities + argmax (MSAU_CONV_HEAD)
        self.wname, self.bname, self.k, self.dil, self.kind = wname, bname, k, dil, kind
        self.relu_in, self.relu_out, self.fwd_add = relu_in, relu_out, fwd_add
        self.bwd_add: Optional[Act] = None
        self.pair: Optional["PairOp"] = None        # set when this conv runs as half of a fused two-conv launch
        self.cpl: Optional["ConvOp"] = None         # on a pair's second conv: the coupling 1x1 conv that reads its output (may ride on the launch)
        self.cpl_fused_into: Optional["PairOp"] = None   # on a coupling conv: the pair whose forward launch computes it (MSAU_PAIR_COUPLE)
        self.proj: Optional["ProjBwd"] = None       # on an attention projection: the fused data-gradient launch of f, g, h
        self.dgrad2 = None                          # a 64 + 64-channel 1x1 conv: both data gradients in one launch (msau_dgrad2_1x1)
        self.dgrad_in_pair: Optional["PairOp"] = None   # on a coupling conv: the pair whose BACKWARD launch computes its two data gradients (MSAU_PAIR_DCOUPLE)
        if kind == "conv" and k == 1 and x2 is not None and relu_out and not relu_in and fwd_add is None and plan.ops:
            prod = plan.ops[-1]
            if isinstance(prod, ConvOp) and prod.out is x2 and prod.pair is not None and prod is prod.pair.c2 \
                    and x1.Cs == x2.Cs == out.Cs == x2.C == out.C:
                prod.cpl = self                      # (structure only: PairOp.bind decides whether an instance takes it)
        assert out.relu_out == relu_out
        if kind == "conv":
            assert out.H == x1.H and out.W == x1.W
            self.pad_t = same_pads(x1.H, k, 1, dil)[0]
            self.pad_l = same_pads(x1.W, k, 1, dil)[0]
        else:
            assert x2 is None and dil == 1 and not relu_in
            p = k // 2
            for o, i in ((out.H, x1.H), (out.W, x1.W)):
                op = o - ((i - 1) * 2 - 2 * p + k)
                assert 0 <= op < 2, f"{name}: output size {o} unreachable from {i}"
            self.pad_t = self.pad_l = k - 1 - p
        self.slots = []
        for x in (x1, x2):
            self.slots.append(x.register() if (x is not None and x.needs_grad) else None)
        self.stage = plan._cur_stage
        plan.ops.append(self)

    def reduce_group(self):
        """slab-reduction group: the stage, except for the level-0 encoder convs of stage 0.  They close the backward
        sweep, and whatever is reduced behind them is exposed; with a group of their own the bulk of stage 0 (the
        level-1..3 layers own most of the slab bytes) is reduced earlier, while the level-0 launches still run."""
        # (moving the stage's own reduction further up -- a group for levels 0..1, 0..2 ... -- measured 0.0 %: the side queue is busy
        #  throughout the backward sweep, so its END does not move; profiles/HISTORY_r03_r04.md)
        return -1 if (self.stage == 0 and self.name.startswith("s0.d0.")) else self.stage

    def reads(self):
        return [t for t in (self.x1, self.x2, self.fwd_add) if t is not None]

    def writes(self):
        lrn, pool = getattr(self, "lrn", None), getattr(self, "pool", None)     # fused outputs are written by this launch too
        w = [self.out] + ([lrn.y] if lrn is not None else []) + ([pool.y] if pool is not None else [])
        if self.cpl is not None:         # a coupling conv that may ride on this launch: its outputs must not alias this launch's inputs
            w += [t for t in self.cpl.writes() if t not in w]
        return w

    # ---- helpers -------------------------------------------------------------------------
    def _geom(self, C1, C2, Cout, dil, stride, ups):
        g = L.ConvPackGeom()
        L.call("msau_conv_pack_geometry", self.plan.dtype, C1, C2, Cout, self.k, self.k, dil, stride, ups, C.byref(g))
        return g

    def _pack_entry(self, g, off, row_is_dim0, flip, row_off, rows_real, k1, k2):
        shp = self.plan.pshape[self.wname]
        e = L.PackEntry()
        e.src_off = self.plan.poff[self.wname]
        e.dst_off = off
        e.kind, e.dim0, e.dim1, e.KH, e.KW = 0, shp[0], shp[1], self.k, self.k
        e.row_is_dim0, e.flip, e.row_off, e.rows_real, e.rows_pad = int(row_is_dim0), int(flip), row_off, rows_real, g.rows
        e.k1_real, e.k1_store = k1
        e.k2_real, e.k2_store = k2
        e.cch, e.nchunks, e.kchunk, e.dtype = g.cch, g.nchunks, g.kchunk, self.plan.dtype
        return e, g.nchunks * g.rows * g.kchunk

    def finalize(self):
        P = self.plan
        x1, x2, out = self.x1, self.x2, self.out
        C2s = x2.Cs if x2 is not None else 0
        C2r = x2.C if x2 is not None else 0
        conv = self.kind == "conv"
        # ---- forward image + bias
        g = self._geom(x1.Cs, C2s, out.Cs, self.dil, 1, 1 if conv else 2)
        self.w_off = P.alloc_pack(g.bytes)
        e, n = self._pack_entry(g, self.w_off, row_is_dim0=conv, flip=not conv, row_off=0, rows_real=out.C,
                                k1=(x1.C, x1.Cs), k2=(C2r, C2s))
        P.add_pack_entry(e, n)
        self.b_off = P.alloc_pack(out.Cs * 4)
        be = L.PackEntry()
        be.src_off, be.dst_off, be.kind = P.poff[self.bname], self.b_off, 1
        be.row_off, be.rows_real, be.rows_pad = 0, out.C, out.Cs
        P.add_pack_entry(be, out.Cs)
        # ---- data-gradient images
        self.d_off = [None, None]
        self.wg_fused = False        # the weight gradient rides on the residual pair's data-gradient launch (MSAU_PAIR_WGRAD1)
        self.uentry = None
        self.dd_off = None           # one launch for both sources of a concat conv (MSAU_CONV_DOUT) when an instance has it
        if P.training and conv and x2 is not None and None not in self.slots and x1.C == x1.Cs == x2.C == x2.Cs \
                and x1.Cs + x2.Cs <= 128 and os.environ.get("MSAU_FUSE_DGRAD", "1") != "0" and not P.act_flag:
            proto = L.ConvDesc()
            proto.B, proto.Hin, proto.Win, proto.Hout, proto.Wout = P.B, out.H, out.W, x1.H, x1.W
            proto.C1, proto.C2, proto.Cout = out.Cs, 0, x1.Cs + x2.Cs
            proto.KH = proto.KW = self.k
            proto.dil, proto.stride, proto.ups = self.dil, 1, 1
            proto.pad_t = (self.k - 1) * self.dil - self.pad_t
            proto.pad_l = (self.k - 1) * self.dil - self.pad_l
            info = (L.i32 * 8)()
            L.call("msau_conv2d_launch_info", P.dtype, C.byref(proto), info)
            if info[7] & 2:
                gd = self._geom(out.Cs, 0, x1.Cs + x2.Cs, self.dil, 1, 1)
                self.dd_off = P.alloc_pack(gd.bytes)
                e, n = self._pack_entry(gd, self.dd_off, row_is_dim0=False, flip=True, row_off=0,
                                        rows_real=x1.C + x2.C, k1=(out.C, out.Cs), k2=(0, 0))
                P.add_pack_entry(e, n)
        # the same image for MSAU_PAIR_DCOUPLE where no two-output instance takes the launch (small images): the prologue of the
        # residual pair's backward launch only needs the packed weights
        self.dcp_off = self.dd_off
        if P.training and conv and self.k == 1 and x2 is not None and self.dd_off is None and None not in self.slots \
                and x1.C == x1.Cs == x2.C == x2.Cs == out.Cs == 32 and P.dtype == L.BF16 and not P.act_flag \
                and any(pr.c2.cpl is self for pr in P.pairs):
            gd = self._geom(out.Cs, 0, x1.Cs + x2.Cs, self.dil, 1, 1)
            self.dcp_off = P.alloc_pack(gd.bytes)
            e, n = self._pack_entry(gd, self.dcp_off, row_is_dim0=False, flip=True, row_off=0,
                                    rows_real=x1.C + x2.C, k1=(out.C, out.Cs), k2=(0, 0))
            P.add_pack_entry(e, n)
        if P.training and self.dd_off is None:
            for si, x in enumerate((x1, x2)):
                if x is None or self.slots[si] is None:
                    continue
                if conv:
                    gd = self._geom(out.Cs, 0, x.Cs, self.dil, 1, 1)
                    self.d_off[si] = P.alloc_pack(gd.bytes)
                    e, n = self._pack_entry(gd, self.d_off[si], row_is_dim0=False, flip=True,
                                            row_off=(0 if si == 0 else x1.C), rows_real=x.C,
                                            k1=(out.C, out.Cs), k2=(0, 0))
                else:
                    gd = self._geom(out.Cs, 0, x.Cs, 1, 2, 1)
                    self.d_off[si] = P.alloc_pack(gd.bytes)
                    e, n = self._pack_entry(gd, self.d_off[si], row_is_dim0=True, flip=False, row_off=0,
                                            rows_real=x.C, k1=(out.C, out.Cs), k2=(0, 0))
                P.add_pack_entry(e, n)

    def bind(self):
        """Create the launch descriptors (all buffers exist now)."""
        P = self.plan
        x1, x2, out = self.x1, self.x2, self.out
        conv = self.kind == "conv"
        d = L.ConvDesc()
        d.B, d.Hin, d.Win, d.Hout, d.Wout = P.B, x1.H, x1.W, out.H, out.W
        d.C1, d.C2, d.Cout = x1.Cs, (x2.Cs if x2 is not None else 0), out.Cs
        d.KH = d.KW = self.k
        d.dil, d.pad_t, d.pad_l, d.stride, d.ups = self.dil, self.pad_t, self.pad_l, 1, (1 if conv else 2)
        d.flags = (L.CONV_RELU_IN if self.relu_in else 0) | ((L.CONV_RELU_OUT | P.act_flag) if self.relu_out else 0) | \
                  (L.CONV_ADD if self.fwd_add is not None else 0)
        d.x1, d.x2 = _ptr(x1.data), _ptr(x2.data if x2 is not None else None)
        d.wpack, d.bias = P.pack_ptr(self.w_off), P.pack_ptr(self.b_off)
        d.add = _ptr(self.fwd_add.data) if self.fwd_add is not None else None
        d.y = _ptr(out.data)
        if self.head:
            # softmax + argmax in the end conv's epilogue when the instance taking the launch implements it,
            # otherwise Plan.predict runs the stand-alone kernel on the stored logits (same arithmetic)
            d.head_probs, d.head_argmax, d.head_classes = _ptr(P.head_probs), _ptr(P.head_argmax), out.C
            info = (L.i32 * 8)()
            L.call("msau_conv2d_launch_info", P.dtype, C.byref(d), info)
            P.head_fused = bool(info[7] & 1) and d.flags == 0 and out.C <= 16
            if P.head_fused:
                d.flags = L.CONV_HEAD
        lrn = getattr(self, "lrn", None)
        if lrn is not None and conv and d.flags & ~L.CONV_RELU_IN == 0 and lrn.a.C == lrn.a.Cs:
            info = (L.i32 * 8)()
            L.call("msau_conv2d_launch_info", P.dtype, C.byref(d), info)
            if info[7] & 4:                  # LocalResponseNorm(size=C) in this conv's epilogue: layers.py:145,161-162
                d.flags |= L.CONV_LRN
                d.y2 = _ptr(lrn.y.data)
                d.lrn_alpha_over_n, d.lrn_beta, d.lrn_k = 1e-4 / lrn.a.C, 0.75, 1.0
                lrn.fused_into = self
        pool = getattr(self, "pool", None)
        if pool is not None and conv and not (d.flags & (L.CONV_LRN | L.CONV_HEAD)):
            info = (L.i32 * 8)()
            L.call("msau_conv2d_launch_info", P.dtype, C.byref(d), info)
            if info[7] & 8:                  # zero pad + MaxPool2d(2,2) in this conv's epilogue: model/model.py:158-160
                d.flags |= L.CONV_POOL
                d.pool_y, d.pool_idx = _ptr(pool.y.data), _ptr(pool.idx)
                pool.fused_into = self
        self.fdesc = d
        self.ddesc = [None, None]
        self.wdesc = None
        if not P.training or out.grad is None:
            return
        if self.dd_off is not None:
            dd = L.ConvDesc()
            dd.B = P.B
            dd.Hin, dd.Win, dd.Hout, dd.Wout = out.H, out.W, x1.H, x1.W
            dd.C1, dd.C2, dd.Cout = out.Cs, 0, x1.Cs + x2.Cs
            dd.KH = dd.KW = self.k
            dd.dil, dd.stride, dd.ups = self.dil, 1, 1
            dd.pad_t = (self.k - 1) * self.dil - self.pad_t
            dd.pad_l = (self.k - 1) * self.dil - self.pad_l
            assert not self.relu_in
            fl = [0, 0]
            for si, x in enumerate((x1, x2)):
                accum, maskb = x.slot_flags(self.slots[si])
                fl[si] = (L.CONV_ACCUM if accum else 0) | (L.CONV_MASK_B if maskb else 0)
                if maskb:
                    setattr(dd, "mask_b" if si == 0 else "mask_b2", _ptr(x.data))
            if self.bwd_add is not None:
                fl[0] |= L.CONV_ADD
                dd.add = _ptr(self.bwd_add.grad)
            dd.flags, dd.flags2 = fl[0] | L.CONV_DOUT, fl[1]
            dd.x1, dd.wpack, dd.bias = _ptr(out.grad), P.pack_ptr(self.dd_off), None
            dd.y, dd.y2 = _ptr(x1.grad), _ptr(x2.grad)
            self.ddesc[0] = dd
        for si, x in enumerate((x1, x2)):
            if x is None or self.slots[si] is None or self.dd_off is not None:
                continue
            accum, maskb = x.slot_flags(self.slots[si])
            dd = L.ConvDesc()
            dd.B = P.B
            dd.Hin, dd.Win, dd.Hout, dd.Wout = out.H, out.W, x.H, x.W
            dd.C1, dd.C2, dd.Cout = out.Cs, 0, x.Cs
            dd.KH = dd.KW = self.k
            if conv:
                dd.dil, dd.stride, dd.ups = self.dil, 1, 1
                dd.pad_t = (self.k - 1) * self.dil - self.pad_t
                dd.pad_l = (self.k - 1) * self.dil - self.pad_l
            else:
                dd.dil, dd.stride, dd.ups = 1, 2, 1
                dd.pad_t = dd.pad_l = self.k // 2
            fl = 0
            if self.relu_in:
                fl |= L.CONV_MASK_A
                dd.mask_a = _ptr(x.data)
            if si == 0 and self.bwd_add is not None:
                fl |= L.CONV_ADD
                dd.add = _ptr(self.bwd_add.grad)
            if accum:
                fl |= L.CONV_ACCUM
            if maskb:
                fl |= L.CONV_MASK_B | P.act_flag
                dd.mask_b = _ptr(x.data)
            dd.flags = fl
            dd.x1, dd.wpack, dd.bias, dd.y = _ptr(out.grad), P.pack_ptr(self.d_off[si]), None, _ptr(x.grad)
            self.ddesc[si] = dd
        # a 1x1 conv over concat(x1, x2) at 64 + 64 channels (the bottleneck level's coupling conv): both data gradients in one launch
        # (msau_dgrad2_1x1, csrc/pointwise.hip) -- MSAU_CONV_DOUT has instances up to 32 + 32
        self.dgrad2 = None
        d1, d2 = self.ddesc
        if conv and self.k == 1 and x2 is not None and self.dd_off is None and d1 is not None and d2 is not None \
                and P.dtype == L.BF16 and not P.act_flag and x1.C == x1.Cs == x2.C == x2.Cs == out.Cs == 64 \
                and not ((d1.flags | d2.flags) & ~(L.CONV_ACCUM | L.CONV_MASK_B)) and os.environ.get("MSAU_DGRAD2", "1") != "0":
            geo = [self._geom(out.Cs, 0, x.Cs, 1, 1, 1) for x in (x1, x2)]
            if all((gq.nchunks, gq.rows, gq.kchunk) == (1, 64, 64) for gq in geo):
                a = L.Dgrad2Args()
                a.g, a.w1_pack, a.w2_pack = _ptr(out.grad), P.pack_ptr(self.d_off[0]), P.pack_ptr(self.d_off[1])
                a.dx1, a.dx2 = _ptr(x1.grad), _ptr(x2.grad)
                a.mask1 = d1.mask_b if d1.flags & L.CONV_MASK_B else None
                a.mask2 = d2.mask_b if d2.flags & L.CONV_MASK_B else None
                a.npix, a.C = out.npix, 64
                a.accumulate1, a.accumulate2 = int(bool(d1.flags & L.CONV_ACCUM)), int(bool(d2.flags & L.CONV_ACCUM))
                self.dgrad2 = a
        # ---- weight gradient
        w = L.WgradDesc()
        w.B = P.B
        w.KH = w.KW = self.k
        if conv:
            w.Hin, w.Win, w.Hout, w.Wout = x1.H, x1.W, out.H, out.W
            w.C1, w.C2, w.Cout = x1.Cs, (x2.Cs if x2 is not None else 0), out.Cs
            w.dil, w.pad_t, w.pad_l, w.stride = self.dil, self.pad_t, self.pad_l, 1
            w.flags = L.CONV_RELU_IN if self.relu_in else 0
            w.x1, w.x2, w.g = _ptr(x1.data), _ptr(x2.data if x2 is not None else None), _ptr(out.grad)
            rows_real, k1, k2 = out.C, (x1.C, x1.Cs), ((x2.C, x2.Cs) if x2 is not None else (0, 0))
        else:
            # transposed conv: dWt[ci][co][ky][kx] = sum x[iy,ix,ci] * dy[2iy-1+ky, 2ix-1+kx, co]
            # = the wgrad of the stride-2 conv that maps dy -> dx, i.e. roles swapped.
            w.Hin, w.Win, w.Hout, w.Wout = out.H, out.W, x1.H, x1.W
            w.C1, w.C2, w.Cout = out.Cs, 0, x1.Cs
            w.dil, w.pad_t, w.pad_l, w.stride = 1, self.k // 2, self.k // 2, 2
            w.flags = 0
            w.x1, w.x2, w.g = _ptr(out.grad), None, _ptr(x1.data)
            rows_real, k1, k2 = x1.C, (out.C, out.Cs), (0, 0)
        wg = L.WgradGeom()
        w.nslabs = 1
        L.call("msau_wgrad_geometry", P.dtype, C.byref(w), C.byref(wg))
        nslabs = max(1, min(wg.max_slabs, int(os.environ.get('MSAU_SLAB_CAP', '384')), max(int(os.environ.get('MSAU_SLAB_MIN', '64')), (int(os.environ.get('MSAU_SLAB_MB', '3')) << 20) // max(wg.slab_bytes, 1))))
        w.nslabs = nslabs
        slab_elems = wg.slab_bytes // 4
        self.slab_off = P.alloc_slab(nslabs * slab_elems)
        self.wdesc, self.wgeom = w, wg
        shp = P.pshape[self.wname]
        taps = self.k * self.k
        u = L.UnpackEntry()
        u.slab_off, u.w_off, u.b_off = self.slab_off, P.poff[self.wname], P.poff[self.bname]
        u.nslabs, u.slab_elems, u.kext = nslabs, slab_elems, wg.kext
        u.dim0, u.dim1, u.KH, u.KW, u.row_is_dim0, u.rows_real = shp[0], shp[1], self.k, self.k, 1, rows_real
        u.k1_real, u.k1_store = k1
        u.k2_real, u.k2_store = k2
        u.cch, u.nchunks, u.accumulate = wg.cch, wg.nchunks, 0
        if conv:
            u.b_src_off, u.b_slab_stride, u.b_elem_stride, u.b_nslabs = self.slab_off + taps * wg.cch, slab_elems, wg.kext, nslabs
            u.b_count = out.C
        else:
            self.csum_blocks = max(1, min(256, out.npix // 1024))
            self.csum_off = P.alloc_slab(self.csum_blocks * out.Cs)
            u.b_src_off, u.b_slab_stride, u.b_elem_stride, u.b_nslabs = self.csum_off, out.Cs, 1, self.csum_blocks
            u.b_count = out.C
        P.add_unpack_entry(u, slab_elems, self.reduce_group())
        self.uentry = u
        # the coupling 1x1 conv's weight gradient rides on its two-output data-gradient launch where an instance has it (the
        # row-streaming 8-channel one): g is that launch's input, the second source its ReLU-mask operand; MSAU_CONV_WGRAD
        dd = self.ddesc[0]
        if conv and self.k == 1 and x2 is not None and self.dd_off is not None and dd is not None and not self.relu_in \
                and dd.flags == L.CONV_DOUT and dd.flags2 == L.CONV_MASK_B and u.kext == 16 and u.slab_elems == 256 and u.cch == 8 \
                and u.nchunks == 2 and os.environ.get("MSAU_COUPLE_WGRAD", "1") != "0" and P.cfg.get("couple_wgrad", True):
            dd.flags |= L.CONV_WGRAD
            dd.wg_x1, dd.wg_slabs = _ptr(x1.data), 1                 # (placeholder: the slab arena does not exist yet)
            ns = int(L.load().msau_conv2d_rider_slabs(P.dtype, C.byref(dd)))
            if ns > 0:
                dd.wg_nslabs = ns
                self.wg_slab_off = P.alloc_slab(ns * 256)
                u.slab_off, u.nslabs = self.wg_slab_off, ns
                u.b_src_off, u.b_nslabs = self.wg_slab_off + 8, ns
                self.wg_fused = True
            else:
                dd.flags &= ~L.CONV_WGRAD
                dd.wg_x1 = dd.wg_slabs = None

    def late_bind(self):
        P = self.plan
        if self.wdesc is not None:
            self.wdesc.slabs = P.slab_ptr(self.slab_off)
        if self.wg_fused and self.pair is None and self.ddesc[0] is not None and self.ddesc[0].flags & L.CONV_WGRAD:
            self.ddesc[0].wg_slabs = P.slab_ptr(self.wg_slab_off)
        # ---- launch metadata for profiling / roofline accounting (bench.py)
        T = "f32" if P.dtype == L.F32 else "bf16"
        esz = 4 if P.dtype == L.F32 else 2
        taps = self.k * self.k
        cin_real = self.x1.C + (self.x2.C if self.x2 is not None else 0)
        self.flops = 2.0 * P.B * (self.out.H * self.out.W if self.kind == "conv" else self.x1.H * self.x1.W) * \
            taps * cin_real * self.out.C

        def conv_meta(d):
            info = (L.i32 * 8)()
            L.call("msau_conv2d_launch_info", P.dtype, C.byref(d), info)
            nin = d.B * d.Hin * d.Win * (d.C1 + d.C2)
            nout = d.B * d.Hout * d.Wout * d.Cout
            extra = sum(1 for f in (L.CONV_ADD, L.CONV_ACCUM, L.CONV_MASK_A, L.CONV_MASK_B, L.CONV_LRN) if d.flags & f)
            if d.flags & L.CONV_DOUT:
                # both halves are written once; the epilogue operands are per half
                half = nout // 2
                ex1 = sum(1 for f in (L.CONV_ADD, L.CONV_ACCUM, L.CONV_MASK_B) if d.flags & f)
                ex2 = sum(1 for f in (L.CONV_ACCUM, L.CONV_MASK_B) if d.flags2 & f)
                rider = (half * esz + d.wg_nslabs * 256 * 4) if d.flags & L.CONV_WGRAD else 0       # the first source read, the slabs written
                return f"conv_lean_kernel<{T},CIN{d.C1},CT{info[0]},K{d.KH},dout{',wgrad' if rider else ''}>", (nin + half * (2 + ex1 + ex2)) * esz + rider
            if info[6] == 3:
                name = f"rowconv_kernel<{T},CIN{d.C1 + d.C2},CO{d.Cout},K{d.KH}{',lrn' if d.flags & L.CONV_LRN else ''}{',ups2' if d.ups == 2 else ''}>"
            elif info[6] == 2:
                name = f"conv_chunked_kernel<{T},CIN{d.C1},K{d.KH}>"
            elif info[6]:
                var = ",dual" if d.C2 else ",ups2" if d.ups == 2 else ",s2" if d.stride == 2 else ""
                var += ",lrn" if d.flags & L.CONV_LRN else ""
                name = f"conv_lean_kernel<{T},CIN{d.C1 + d.C2},CT{info[0]},K{d.KH}{var}>"
            else:
                name = f"conv_kernel<{T},CT{info[0]},PT{info[1]}>"
            pooled = (nout // 4) * (esz + 1) if d.flags & L.CONV_POOL else 0       # pooled tensor + 1-byte positions
            return name, (nin + nout * (1 + extra)) * esz + pooled

        self.fkey, self.fbytes = conv_meta(self.fdesc)
        P.note_launch(self.fkey, self.fbytes, self.flops)
        self.dmeta = [None, None]
        for si, dd in enumerate(self.ddesc):
            if dd is not None:
                src_c = (self.x1.C + self.x2.C) if self.dd_off is not None else (self.x1, self.x2)[si].C
                fl = 2.0 * P.B * (self.out.H * self.out.W if self.kind == "conv" else self.x1.H * self.x1.W) * taps * src_c * self.out.C
                self.dmeta[si] = conv_meta(dd)
                if (self.proj is None or not self.proj.active) and self.dgrad2 is None and self.dgrad_in_pair is None:
                    P.note_launch(self.dmeta[si][0], self.dmeta[si][1], fl)
        if self.dgrad2 is not None:
            a = self.dgrad2
            nops = 3 + a.accumulate1 + a.accumulate2 + (1 if a.mask1 else 0) + (1 if a.mask2 else 0)
            self.d2meta = ("dgrad2_1x1<bf16,C64>", self.out.npix * 64 * nops * esz)
            P.note_launch(self.d2meta[0], self.d2meta[1], 2.0 * self.out.npix * 128 * self.out.C)
        if self.wdesc is not None:
            w, wg = self.wdesc, self.wgeom
            ctn = -(-w.Cout // 16)
            ctn = 4 if ctn == 3 else (8 if ctn > 4 else ctn)
            nkw = -(-(wg.kext // 16) // 4)
            nkw = 1 if nkw <= 1 else 2 if nkw <= 2 else 3 if nkw <= 3 else 5 if nkw <= 5 else 10
            if wg.lean == 2:
                self.wkey = f"rowwgrad_kernel<{T},C{wg.cch},CO{w.Cout},K{w.KH}>"
            elif wg.lean:
                self.wkey = f"wgrad_lean_kernel<{T},C{wg.cch},CO{w.Cout},K{w.KH}>"
            else:
                self.wkey = f"wgrad_kernel<{T},CT{ctn},NK{nkw}>"
            self.wbytes = (w.B * w.Hin * w.Win * (w.C1 + w.C2) * wg.nchunks // wg.nchunks + w.B * w.Hout * w.Wout * w.Cout) * esz \
                + w.nslabs * wg.slab_bytes
            if not self.wg_fused:
                P.note_launch(self.wkey, self.wbytes, self.flops)
            if self.kind != "conv":                      # bias gradient of the transposed conv: one pass over its output gradient
                P.note_launch("msau_channel_sum", self.out.npix * self.out.Cs * esz, 0.0)

    def fwd(self, s):
        if self.pair is not None and self.pair.active:
            if self is self.pair.c2:
                L.call("msau_conv_pair", s, self.plan.dtype, C.byref(self.pair.fdesc), key=self.pair.key)
            return
        if self.cpl_fused_into is not None:
            return
        L.call("msau_conv2d", s, self.plan.dtype, C.byref(self.fdesc), key=self.fkey)

    def fwd_recs(self):
        rm = self.plan.rec_meta
        if self.cpl_fused_into is not None:             # computed by the residual pair's forward launch
            return []
        if self.pair is not None and self.pair.active:
            if self is not self.pair.c2:
                return []
            rm[C.addressof(self.pair.fdesc)] = (self.pair.key, self.pair.fbytes)
            return [(L.OP_CONV_PAIR, self.pair.fdesc)]
        rm[C.addressof(self.fdesc)] = (self.fkey, self.fbytes)
        return [(L.OP_CONV2D, self.fdesc)]

    def bwd_recs(self):
        if self.wdesc is None:
            return []
        side = L.OP_SIDE if self.plan.overlap_wgrad else 0     # weight gradients run beside the data-gradient chain
        if self.x1 is self.plan.x_in:
            side = 0        # the net's first conv has no data gradient: nothing is left on the main stream to run beside,
                            # and the side stream is still busy with the two weight gradients enqueued before this one
        recs = [] if self.wg_fused else [(L.OP_WGRAD | side, self.wdesc)]
        if self.kind != "conv":
            P = self.plan
            self._csum = L.CsumArgs(_ptr(self.out.grad), self.out.npix, self.out.Cs, P.slab_ptr(self.csum_off), self.csum_blocks)
            recs.append((L.OP_CHANNEL_SUM | side, self._csum))
        rm = self.plan.rec_meta
        if not self.wg_fused:
            rm[C.addressof(self.wdesc)] = (self.wkey, self.wbytes)
        if self.pair is not None and self.pair.active and self.pair.bdesc is not None:
            if self is self.pair.c2:                    # both data gradients in one launch, behind the second conv's wgrad
                rm[C.addressof(self.pair.bdesc)] = (self.pair.key, self.pair.bbytes)
                if self.pair.dcp is not None:           # (MSAU_PAIR_DCOUPLE: that launch WRITES the gradient this conv's wgrad reads)
                    return [(L.OP_CONV_PAIR, self.pair.bdesc)] + recs
                recs.append((L.OP_CONV_PAIR, self.pair.bdesc))
            return recs
        if self.dgrad_in_pair is not None:              # a coupling conv whose data gradients are the prologue of the pair's backward launch
            return recs
        if self.proj is not None and self.proj.active:  # f, g, h of an attention block: one launch, behind the last of the three
            return recs + (self.proj.recs() if self is self.proj.f else [])
        if self.dgrad2 is not None:
            rm[C.addressof(self.dgrad2)] = self.d2meta
            return recs + [(L.OP_DGRAD2_1X1, self.dgrad2)]
        for si, dd in enumerate(self.ddesc):
            if dd is not None:
                rm[C.addressof(dd)] = self.dmeta[si]
                recs.append((L.OP_CONV2D, dd))
        return recs

    def bwd_wgrad(self, s):
        """weight / bias gradient: reads out.grad and the saved inputs, writes only this op's slabs"""
        if self.wdesc is None or self.wg_fused:
            return
        P = self.plan
        L.call("msau_conv2d_wgrad", s, P.dtype, C.byref(self.wdesc), key=self.wkey)
        if self.kind != "conv":
            L.call("msau_channel_sum", s, P.dtype, _ptr(self.out.grad), self.out.npix, self.out.Cs,
                   P.slab_ptr(self.csum_off), self.csum_blocks)

    def bwd_dgrad(self, s):
        if self.wdesc is None:
            return
        P = self.plan
        if self.pair is not None and self.pair.active and self.pair.bdesc is not None:
            if self is self.pair.c2:
                L.call("msau_conv_pair", s, P.dtype, C.byref(self.pair.bdesc), key=self.pair.key)
            return
        if self.proj is not None and self.proj.active:
            if self is self.proj.f:
                self.proj.launch(s)
            return
        if self.dgrad2 is not None:
            L.call("msau_dgrad2_1x1", s, P.dtype, C.byref(self.dgrad2), key=self.d2meta[0])
            return
        if self.dgrad_in_pair is not None:
            return
        for si, dd in enumerate(self.ddesc):
            if dd is not None:
                L.call("msau_conv2d", s, P.dtype, C.byref(dd), key=self.dmeta[si][0])

    def bwd(self, s):
        if self.pair is not None and self.pair.active and self.pair.dcp is not None and self is self.pair.c2:
            self.bwd_dgrad(s)                           # (MSAU_PAIR_DCOUPLE: the pair launch writes the gradient the wgrad reads)
            self.bwd_wgrad(s)
            return
        self.bwd_wgrad(s)
        self.bwd_dgrad(s)


class PairOp:
    """The two convs of a res_depth-2 residual block (model/model.py:37-50) as ONE launch per sweep (msau_conv_pair,
    csrc/conv_pair.hip): forward r1 = ReLU(conv1(ReLU(x0))), out = ReLU(conv2(r1) + x0); backward both data gradients.
    The ConvOps keep their packed images, weight-gradient launches and bookkeeping; only their forward / data-gradient
    launches are replaced.  Not an entry of plan.ops: the ConvOps emit the fused records."""

    def __init__(self, plan: "Plan", c1: ConvOp, c2: ConvOp):
        self.plan, self.c1, self.c2 = plan, c1, c2
        self.active = False
        self.fdesc = self.bdesc = None
        self.dcp: Optional[ConvOp] = None            # the coupling conv whose two data gradients this pair's backward launch computes
        c1.pair = c2.pair = self
        plan.pairs.append(self)

    def bind(self):
        """after both ConvOps are bound: build the fused descriptors if an instance takes the shape"""
        P, c1, c2 = self.plan, self.c1, self.c2
        x0, r1, out = c1.x1, c1.out, c2.out
        if os.environ.get("MSAU_FUSE_PAIR", "1") == "0" or not P.cfg.get("fuse_pair", True) or P.act_flag:
            return
        ok = (c1.kind == c2.kind == "conv" and c1.k == c2.k == 3 and c1.dil == c2.dil == 1 and c1.x2 is None and c2.x2 is None
              and c1.relu_in and c1.relu_out and c1.fwd_add is None and not c2.relu_in and c2.relu_out and c2.fwd_add is x0
              and c2.x1 is r1 and x0.C == x0.Cs == r1.C == out.C and x0.Cs in (8, 16, 32) and not c1.head and not c2.head
              and r1.n_contrib == (1 if P.training else 0))
        if not ok:
            return
        f = L.ConvPairDesc()
        f.B, f.H, f.W, f.C = P.B, x0.H, x0.W, x0.Cs
        f.flags1, f.flags2 = L.PAIR_RELU_IN | L.PAIR_RELU_MID, L.CONV_ADD | L.CONV_RELU_OUT
        f.x, f.w1, f.b1, f.mid = _ptr(x0.data), P.pack_ptr(c1.w_off), P.pack_ptr(c1.b_off), _ptr(r1.data)
        f.w2, f.b2, f.add, f.y = P.pack_ptr(c2.w_off), P.pack_ptr(c2.b_off), _ptr(x0.data), _ptr(out.data)
        if not L.load().msau_conv_pair_applicable(P.dtype, C.byref(f)):
            return
        pool = getattr(c2, "pool", None)
        if pool is not None and x0.Cs < int(os.environ.get("MSAU_PAIR_POOL_MINC", "16")):
            # measured: at 8 channels the pooled-output variant of the pair launch (97 VGPRs, 4 waves per SIMD instead of 6)
            # costs 12 us for a 10.8 us pool launch; at 16 channels 3.3 us for 8.6 us
            pool.fused_into = None
            pool = None
        if pool is not None:                 # the pooled output rides on the fused launch (c2's own descriptor is not launched)
            f.flags2 |= L.CONV_POOL
            f.pool_y, f.pool_idx = _ptr(pool.y.data), _ptr(pool.idx)
            if L.load().msau_conv_pair_applicable(P.dtype, C.byref(f)):
                pool.fused_into = self
            else:
                f.flags2 &= ~L.CONV_POOL
                pool.fused_into = None
        self.fdesc = f
        T = "f32" if P.dtype == L.F32 else "bf16"
        esz = 4 if P.dtype == L.F32 else 2
        self.key = f"conv_pair_kernel<{T},C{x0.Cs}>"
        n = P.B * x0.H * x0.W * x0.Cs
        self.fbytes = 3 * n * esz                       # x0 read once (it is also the residual operand), r1 and out written once
        if f.flags2 & L.CONV_POOL:
            self.fbytes += (n // 4) * (esz + 1)
        if P.training and c1.ddesc[0] is not None and c2.ddesc[0] is not None and c1.ddesc[1] is None and c2.ddesc[1] is None \
                and c1.d_off[0] is not None and c2.d_off[0] is not None:
            d1, d2 = c1.ddesc[0], c2.ddesc[0]
            if d2.flags == L.CONV_MASK_B and not (d1.flags & ~(L.CONV_MASK_A | L.CONV_ADD | L.CONV_ACCUM | L.CONV_MASK_B)):
                b = L.ConvPairDesc()
                b.B, b.H, b.W, b.C = P.B, x0.H, x0.W, x0.Cs
                b.flags1, b.flags2 = L.PAIR_MASK_MID, d1.flags
                b.x, b.w1, b.b1, b.mask_mid, b.mid = _ptr(out.grad), P.pack_ptr(c2.d_off[0]), None, d2.mask_b, _ptr(r1.grad)
                b.w2, b.b2, b.add, b.mask_a, b.mask_b, b.y = P.pack_ptr(c1.d_off[0]), None, d1.add, d1.mask_a, d1.mask_b, d1.y
                use_bits = os.environ.get("MSAU_PAIR_BITS", "1") != "0"
                if use_bits:
                    b.bits_mid = b.bits_a = 1       # placeholders (the planes are allocated below): the row-streaming instances
                                                    # take the backward flag set only WITH planes, so the probe must carry them
                if L.load().msau_conv_pair_applicable(P.dtype, C.byref(b)):     # MASK_A + ADD of the input tensor only
                    self.bdesc = b
                else:
                    b.bits_mid = b.bits_a = None
                self.bbytes = 5 * n * esz               # g (also the ADD operand), r1 mask, x0 mask read once; g_r1, dx written once
                if self.bdesc is not None and use_bits:
                    # the two ReLU masks as bit planes: written by the forward launch, read by the backward launch; the
                    # layout (and so the size) belongs to the instance that takes the shape (msau_conv_pair_bits_bytes)
                    nbits = int(L.load().msau_conv_pair_bits_bytes(P.dtype, C.byref(b)))
                    if nbits != int(L.load().msau_conv_pair_bits_bytes(P.dtype, C.byref(f))):
                        # the two launches would go to different instances (a forward flag only the tile kernels have)
                        f.flags1 |= L.PAIR_TILES
                        b.flags1 |= L.PAIR_TILES
                        nbits = int(L.load().msau_conv_pair_bits_bytes(P.dtype, C.byref(b)))
                    self.bits_mid = torch.zeros((nbits,), dtype=torch.uint8, device=P.device)
                    self.bits_a = torch.zeros_like(self.bits_mid)
                    for dsc in (f, b):
                        dsc.bits_mid, dsc.bits_a = _ptr(self.bits_mid), _ptr(self.bits_a)
                    self.fbytes += 2 * nbits
                    self.bbytes = 3 * n * esz + 2 * nbits
                # the block's input is an LRN output nothing else reads: its backward rides on this launch (conv_rows.hip)
                lrn = getattr(x0, "lrn_producer", None)
                if self.bdesc is not None and lrn is not None and x0.n_contrib == 1 and not (d1.flags & L.CONV_ACCUM) \
                        and lrn.a.grad is not None and lrn.a.n_contrib == 1 and not lrn.a.relu_out and lrn.a.C == lrn.a.Cs == x0.Cs \
                        and os.environ.get("MSAU_FUSE_LRN_BWD", "1") != "0":
                    b.flags1 |= L.PAIR_LRN_BWD
                    b.lrn_a, b.lrn_da = _ptr(lrn.a.data), _ptr(lrn.a.grad)
                    b.lrn_alpha_over_n, b.lrn_beta, b.lrn_k = 1e-4 / lrn.a.C, 0.75, 1.0
                    if L.load().msau_conv_pair_instance(P.dtype, C.byref(b)) == 2:
                        lrn.bwd_fused_into = self
                        self.bbytes += n * esz              # a read, da written instead of dx
                    else:
                        b.flags1 &= ~L.PAIR_LRN_BWD
                        b.lrn_a = b.lrn_da = None
                # the first conv's weight gradient rides on this launch too: its g operand is the row the walk has just produced
                u = c1.uentry
                if self.bdesc is not None and u is not None and c1.wdesc is not None and c1.relu_in and r1.n_contrib == 1 \
                        and u.kext == 80 and u.slab_elems == 640 and os.environ.get("MSAU_PAIR_WGRAD", "1") != "0":
                    b.flags1 |= L.PAIR_WGRAD1
                    b.wg1_x, b.wg1_slabs = _ptr(x0.data), 1                 # (placeholder: the slab arena does not exist yet)
                    ns = int(L.load().msau_conv_pair_wgrad_slabs(P.dtype, C.byref(b)))
                    if ns > 0:
                        b.wg1_nslabs = ns
                        self.wg_slab_off = P.alloc_slab(ns * 640)
                        u.slab_off, u.nslabs = self.wg_slab_off, ns
                        u.b_src_off, u.b_nslabs = self.wg_slab_off + 72, ns
                        c1.wg_fused = True
                        self.bbytes += ns * 640 * 4                      # x0 read instead of the intermediate gradient written; the slabs
                    else:
                        b.flags1 &= ~L.PAIR_WGRAD1
                        b.wg1_x = b.wg1_slabs = None
        # the coupling conv that reads this block's output (model/model.py:143-148,246-252) rides on the forward launch where an
        # instance has it (row-streaming 8 / 16 channels): one launch and one read of the block's output less per coupled level
        cp = c2.cpl
        self.cpl = None
        if cp is not None and not (f.flags2 & L.CONV_POOL) and cp.x2 is out and getattr(cp, "lrn", None) is None and not cp.head \
                and os.environ.get("MSAU_PAIR_COUPLE", "1") != "0" and P.cfg.get("fuse_couple", True):
            f.flags1 |= L.PAIR_COUPLE
            f.cpl_prev, f.cpl_w, f.cpl_b, f.cpl_y = _ptr(cp.x1.data), P.pack_ptr(cp.w_off), P.pack_ptr(cp.b_off), _ptr(cp.out.data)
            cpool = getattr(cp, "pool", None)                        # the pool behind the coupling conv (encoder levels): it moves along
            if cpool is not None:
                f.cpl_pool_y, f.cpl_pool_idx = _ptr(cpool.y.data), _ptr(cpool.idx)
                if not L.load().msau_conv_pair_applicable(P.dtype, C.byref(f)):
                    f.cpl_pool_y = f.cpl_pool_idx = None             # ... or stays where it was (the coupling launch's epilogue / a launch of its own)
                    cpool = None
            if L.load().msau_conv_pair_applicable(P.dtype, C.byref(f)):
                self.cpl = cp
                cp.cpl_fused_into = self
                if cpool is not None:
                    cpool.fused_into = self
                elif getattr(cp, "pool", None) is not None and cp.pool.fused_into is cp:
                    cp.pool.fused_into = None                        # (the coupling launch is gone: the pool runs on its own)
                self.fbytes += 2 * n * esz + ((n // 4) * (esz + 1) if f.cpl_pool_y else 0)     # prev read, z written (+ pooled z, positions)
            else:
                f.flags1 &= ~L.PAIR_COUPLE
                f.cpl_prev = f.cpl_w = f.cpl_b = f.cpl_y = f.cpl_pool_y = f.cpl_pool_idx = None
        # ... and its two-output data gradient becomes the PROLOGUE of the backward launch where an instance has that (the 32-channel
        # tile pair): the launch reads d(z) instead of d(y), computes d(y) for its tile and writes d(y) and d(prev) of its own pixels
        b = self.bdesc
        plain = False                                  # d(prev) a plain write, d(y) masked by y and nothing else: what the prologue implements
        if b is not None and cp is not None and cp.x2 is out and getattr(cp, "dcp_off", None) is not None and out.n_contrib == 1:
            d1, d2 = cp.ddesc
            if cp.dd_off is not None:
                plain = d1 is not None and d1.flags == L.CONV_DOUT and d1.flags2 == L.CONV_MASK_B and d1.mask_b2 == _ptr(out.data)
            else:
                plain = d1 is not None and d2 is not None and d1.flags == 0 and d2.flags == L.CONV_MASK_B and d2.mask_b == _ptr(out.data)
        if plain and os.environ.get("MSAU_PAIR_DCOUPLE", "1") != "0" and P.cfg.get("fuse_couple", True):
            b.flags1 |= L.PAIR_DCOUPLE
            b.dcp_dz, b.dcp_w, b.dcp_mask, b.dcp_dprev = _ptr(cp.out.grad), P.pack_ptr(cp.dcp_off), _ptr(out.data), _ptr(cp.x1.grad)
            if L.load().msau_conv_pair_applicable(P.dtype, C.byref(b)) and L.load().msau_conv_pair_instance(P.dtype, C.byref(b)) == 1:
                self.dcp = cp
                cp.dgrad_in_pair = self
                self.bbytes += 3 * n * esz                             # y (the mask) read, d(y) and d(prev) written; d(z) read in place of d(y)
            else:
                b.flags1 &= ~L.PAIR_DCOUPLE
                b.dcp_dz = b.dcp_w = b.dcp_mask = b.dcp_dprev = None
        # label by the instance that takes the launches (the backward descriptor, once it has its planes, decides for both)
        probe = self.bdesc if self.bdesc is not None else f
        if L.load().msau_conv_pair_instance(P.dtype, C.byref(probe)) == 2:
            self.key = f"rowpair_kernel<{T},C{x0.Cs}>"
        self.active = True

    def late_bind(self):
        if self.bdesc is not None and self.c1.wg_fused:
            self.bdesc.wg1_slabs = self.plan.slab_ptr(self.wg_slab_off)

    def note(self):
        """replace the two convs' launch accounting by the fused launches' (bench.py roofline)"""
        P, c1, c2 = self.plan, self.c1, self.c2
        if not self.active:
            return
        for c in (c1, c2):
            P.unnote_launch(c.fkey, c.fbytes, c.flops)
        cflops = 0.0
        if self.cpl is not None:
            P.unnote_launch(self.cpl.fkey, self.cpl.fbytes, self.cpl.flops)
            cflops = self.cpl.flops
        P.note_launch(self.key, self.fbytes, c1.flops + c2.flops + cflops)
        if self.bdesc is not None:
            for c in (c1, c2):
                P.unnote_launch(c.dmeta[0][0], c.dmeta[0][1], c.flops)
            P.note_launch(self.key, self.bbytes, c1.flops + c2.flops + (2.0 * self.dcp.flops if self.dcp is not None else 0.0))


class BoxOp(Op):
    """BoxConv2d(c, F, max, max) of the model_box.py variant (model/model_box.py:30-33; csrc/boxconv.hip): c -> c*F channels,
    normalised box integrals read from a fp32 integral image.  PARITY UNPINNED (third-party op, see oracle/box_oracle.py).
    The forward integral image is kept for the backward (box-parameter gradient); the integral image of the output
    gradient is a scratch buffer shared by all box ops of the plan."""

    def __init__(self, plan, name, x: Act, y: Act, pname: str, relu_in: bool):
        self.plan, self.name, self.x, self.y, self.pname, self.relu_in = plan, name, x, y, pname, relu_in
        self.F = plan.cfg["num_box_per_channels"]
        assert y.C == x.C * self.F and x.C == x.Cs
        self.slot = x.register() if x.needs_grad else None
        self.bwd_add: Optional[Act] = None           # d(x) += g(block output): the block's residual add (first box op only)
        self.stage = plan._cur_stage
        plan.ops.append(self)
        plan.box_ops.append(self)

    def reads(self):
        return [self.x]

    def writes(self):
        return [self.y]

    def finalize(self):
        P, x = self.plan, self.x
        n = P.B * x.C * (x.H + 1) * (x.W + 1)
        self.ii = torch.zeros((n,), dtype=torch.float32, device=P.device) if P.training else None
        P.box_scratch = max(P.box_scratch, n * (self.F if P.training else 1))
        P.box_ws_ii = max(P.box_ws_ii, int(L.load().msau_box_integral_ws_floats(P.B, x.H, x.W, x.C * (self.F if P.training else 1))))
        self.params = torch.zeros((2, 4, x.C * self.F), dtype=torch.float32, device=P.device)       # [fwd | reflected]
        self.ws = None
        if P.training and self.y.grad is not None:
            self.ws = torch.zeros((int(L.load().msau_box_pgrad_ws_floats(P.B, x.H, x.W, x.C, self.F)),), dtype=torch.float32, device=P.device)

    def note(self):
        P, x, y = self.plan, self.x, self.y
        esz = 4 if P.dtype == L.F32 else 2
        T = "f32" if P.dtype == L.F32 else "bf16"
        nii = P.B * x.C * (x.H + 1) * (x.W + 1) * 4
        self.fkey, self.bkey = f"box_fwd<{T},C{x.C}>", f"box_bwd<{T},C{x.C}>"
        # forward: x read, II written + scanned in place (3 passes) + gathered once, y written
        self.fbytes = x.npix * x.Cs * esz + 4 * nii + y.npix * y.Cs * esz
        P.note_launch(self.fkey, self.fbytes, 0.0)
        self.bbytes = 0
        if self.ws is not None:
            # backward: g(y) read twice (parameter gradient, integral), II read, II of g(y) (F x larger) written / scanned / gathered, g(x) written
            self.bbytes = 2 * y.npix * y.Cs * esz + nii + 4 * self.F * nii + x.npix * x.Cs * esz
            P.note_launch(self.bkey, self.bbytes, 0.0)

    def _args(self):
        P, x, y = self.plan, self.x, self.y
        a = L.BoxArgs()
        a.in_, a.ii = _ptr(x.data), _ptr(self.ii) if self.ii is not None else _ptr(P.box_ii_g)
        a.params_fwd, a.params_refl = self.params[0].data_ptr(), self.params[1].data_ptr()
        a.out, a.ws_ii = _ptr(y.data), _ptr(P.box_ws)
        a.B, a.H, a.W, a.C, a.F, a.Cs_in, a.Cs_out = P.B, x.H, x.W, x.C, self.F, x.Cs, y.Cs
        a.relu_in = int(self.relu_in)
        mb = float(P.cfg["max_box_sizes"])
        a.max_h = a.max_w = mb
        offs = [P.poff[f"{self.pname}.{nm}"] for nm in ("x_min", "x_max", "y_min", "y_max")]
        a.off_hmin, a.off_hmax, a.off_wmin, a.off_wmax = offs
        return a

    def fwd_recs(self):
        self._fa = self._args()
        self.plan.rec_meta[C.addressof(self._fa)] = (self.fkey, self.fbytes)
        return [(L.OP_BOX_FWD, self._fa)]

    def bwd_recs(self):
        P, x, y = self.plan, self.x, self.y
        if self.ws is None:
            return []
        a = self._args()
        a.gout, a.ii_g, a.ws = _ptr(y.grad), _ptr(P.box_ii_g), _ptr(self.ws)
        a.gin = None
        if x.grad is not None and self.slot is not None:
            accum, maskb = x.slot_flags(self.slot)
            a.gin, a.accumulate = _ptr(x.grad), int(accum)
            a.mask_a = _ptr(x.data) if self.relu_in else None
            a.add = _ptr(self.bwd_add.grad) if self.bwd_add is not None else None
            a.mask_b = _ptr(x.data) if maskb else None
        self._ba = a
        P._box_bwd_args.append(a)
        P.rec_meta[C.addressof(a)] = (self.bkey, self.bbytes)
        return [(L.OP_BOX_BWD, a)]

    def fwd(self, s):
        L.call("msau_box_fwd", s, self.plan.dtype, C.byref(self._fa), key=self.fkey)

    def bwd(self, s):
        if self.ws is not None:
            L.call("msau_box_bwd", s, self.plan.dtype, C.byref(self._ba), key=self.bkey)

    def load_params(self, s, flat_params: torch.Tensor):
        a = self._fa
        L.call("msau_box_params", s, flat_params.data_ptr(), a.off_hmin, a.off_hmax, a.off_wmin, a.off_wmax, a.C, a.F, a.max_h, a.max_w,
               self.params[0].data_ptr(), self.params[1].data_ptr())


class LrnOp(Op):
    """LocalResponseNorm(size=C) after the dilated conv: layers.py:145,161-162."""

    def __init__(self, plan, name, a: Act, y: Act):
        self.plan, self.name, self.a, self.y = plan, name, a, y
        self.slot = a.register() if a.needs_grad else None
        self.stage = plan._cur_stage
        self.fused_into = None             # the ConvOp whose epilogue writes y (MSAU_CONV_LRN), when its instance can
        self.bwd_fused_into = None         # the PairOp whose data-gradient launch also runs this backward (MSAU_PAIR_LRN_BWD)
        y.lrn_producer = self
        prod = plan.ops[-1] if plan.ops else None
        if isinstance(prod, ConvOp) and prod.out is a and os.environ.get("MSAU_FUSE_LRN", "1") != "0":
            prod.lrn = self
        plan.ops.append(self)

    def reads(self):
        return [self.a]

    def writes(self):
        return [self.y]

    def note(self):
        """algorithmic bytes: forward reads a, writes y; backward reads a and dy, writes da"""
        P, a = self.plan, self.a
        esz = 4 if P.dtype == L.F32 else 2
        T = "f32" if P.dtype == L.F32 else "bf16"
        n = a.npix * a.Cs * esz
        self.fkey, self.bkey = f"lrn_fwd<{T},C{a.Cs}>", f"lrn_bwd<{T},C{a.Cs}>"
        self.fbytes, self.bbytes = 2 * n, 3 * n
        if self.fused_into is None:
            P.note_launch(self.fkey, self.fbytes, 0.0)
        if P.training and self.y.grad is not None and a.grad is not None and self.bwd_fused_into is None:
            P.note_launch(self.bkey, self.bbytes, 0.0)

    def fwd_recs(self):
        a, y = self.a, self.y
        if self.fused_into is not None:
            return []
        self._fa = L.LrnArgs(_ptr(a.data), None, _ptr(y.data), a.npix, a.C, a.Cs, a.C, 1e-4, 0.75, 1.0)
        self.plan.rec_meta[C.addressof(self._fa)] = (self.fkey, self.fbytes)
        return [(L.OP_LRN_FWD, self._fa)]

    def bwd_recs(self):
        a, y = self.a, self.y
        if y.grad is None or a.grad is None or self.bwd_fused_into is not None:
            return []
        self._ba = L.LrnArgs(_ptr(a.data), _ptr(y.grad), _ptr(a.grad), a.npix, a.C, a.Cs, a.C, 1e-4, 0.75, 1.0)
        self.plan.rec_meta[C.addressof(self._ba)] = (self.bkey, self.bbytes)
        # (round 1 tagged this launch MSAU_OP_JOIN in deterministic mode: its results varied from run to run beside a
        # side-stream kernel.  Root cause found in round 2 -- packed-fp32 instructions, msau_amd/build.py -- so the join,
        # and its 4 % cost, are gone.)
        return [(L.OP_LRN_BWD, self._ba)]

    def fwd(self, s):
        a, y = self.a, self.y
        if self.fused_into is not None:
            return
        L.call("msau_lrn_fwd", s, self.plan.dtype, _ptr(a.data), _ptr(y.data), a.npix, a.C, a.Cs, a.C, 1e-4, 0.75, 1.0, key=self.fkey)

    def bwd(self, s):
        a, y = self.a, self.y
        if y.grad is None or a.grad is None or self.bwd_fused_into is not None:
            return
        assert a.n_contrib == 1 and not a.relu_out
        L.call("msau_lrn_bwd", s, self.plan.dtype, _ptr(a.data), _ptr(y.grad), _ptr(a.grad), a.npix, a.C, a.Cs, a.C,
               1e-4, 0.75, 1.0, key=self.bkey)


class PoolOp(Op):
    """zero SAME pad + MaxPool2d(2,2): model/model.py:158-160."""

    def __init__(self, plan, name, x: Act, y: Act):
        self.plan, self.name, self.x, self.y = plan, name, x, y
        assert y.H == (x.H + 1) // 2 and y.W == (x.W + 1) // 2 and y.Cs == x.Cs
        self.idx = torch.zeros((plan.B, y.H, y.W, y.Cs), dtype=torch.uint8, device=plan.device) if plan.training else None
        self.slot = x.register() if x.needs_grad else None
        self.stage = plan._cur_stage
        self.fused_into = None             # the ConvOp / PairOp whose epilogue writes y and idx (MSAU_CONV_POOL)
        prod = plan.ops[-1] if plan.ops else None
        if isinstance(prod, ConvOp) and prod.out is x and os.environ.get("MSAU_FUSE_POOL", "1") != "0":
            prod.pool = self
        plan.ops.append(self)

    def reads(self):
        return [self.x]

    def writes(self):
        return [self.y]

    def note(self):
        """forward: x read, y (+ 1-byte argmax) written; backward: dy + argmax read, dx written (+ mask / accumulate reads)"""
        P, x, y = self.plan, self.x, self.y
        esz = 4 if P.dtype == L.F32 else 2
        T = "f32" if P.dtype == L.F32 else "bf16"
        nx, ny = x.npix * x.Cs, y.npix * y.Cs
        self.fkey, self.bkey = f"pool_fwd<{T},C{x.Cs}>", f"pool_bwd<{T},C{x.Cs}>"
        self.fbytes = (nx + ny) * esz + (ny if P.training else 0)
        if self.fused_into is None:
            P.note_launch(self.fkey, self.fbytes, 0.0)
        self.bbytes = 0
        if P.training and y.grad is not None and x.grad is not None:
            accum, maskb = x.slot_flags(self.slot)
            self.bbytes = (ny + nx * (1 + int(accum) + int(maskb))) * esz + ny
            P.note_launch(self.bkey, self.bbytes, 0.0)

    def fwd_recs(self):
        x, y = self.x, self.y
        if self.fused_into is not None:
            return []
        self._fa = L.PoolArgs(_ptr(x.data), _ptr(y.data), _ptr(self.idx), None, self.plan.B, x.H, x.W, x.Cs, 0)
        self.plan.rec_meta[C.addressof(self._fa)] = (self.fkey, self.fbytes)
        return [(L.OP_POOL_FWD, self._fa)]

    def bwd_recs(self):
        x, y = self.x, self.y
        if y.grad is None or x.grad is None:
            return []
        accum, maskb = x.slot_flags(self.slot)
        self._ba = L.PoolArgs(_ptr(y.grad), _ptr(x.grad), _ptr(self.idx), _ptr(x.data) if maskb else None,
                              self.plan.B, x.H, x.W, x.Cs, int(accum) | (2 if maskb and self.plan.act_flag else 0))
        self.plan.rec_meta[C.addressof(self._ba)] = (self.bkey, self.bbytes)
        return [(L.OP_POOL_BWD, self._ba)]

    def fwd(self, s):
        x, y = self.x, self.y
        if self.fused_into is not None:
            return
        L.call("msau_maxpool2x2_fwd", s, self.plan.dtype, _ptr(x.data), _ptr(y.data), _ptr(self.idx), self.plan.B, x.H, x.W, x.Cs,
               key=self.fkey)

    def bwd(self, s):
        x, y = self.x, self.y
        if y.grad is None or x.grad is None:
            return
        accum, maskb = x.slot_flags(self.slot)
        L.call("msau_maxpool2x2_bwd", s, self.plan.dtype, _ptr(y.grad), _ptr(self.idx), _ptr(x.grad),
               _ptr(x.data) if maskb else None, self.plan.B, x.H, x.W, x.Cs, int(accum) | (2 if maskb and self.plan.act_flag else 0),
               key=self.bkey)


class AttnCoreOp(Op):
    """y = x + h . softmax_rows(g^T f): model/layers/attention.py:156-162 (f, g, h are ConvOps)."""

    def __init__(self, plan, name, f: Act, g: Act, h: Act, x: Act, y: Act):
        self.plan, self.name, self.f, self.g, self.h, self.x, self.y = plan, name, f, g, h, x, y
        self.N = x.H * x.W
        self.stats = torch.zeros((plan.B, self.N, 2), dtype=torch.float32, device=plan.device)
        self.ws = torch.zeros((plan.B, self.N), dtype=torch.float32, device=plan.device)
        for t in (f, g, h):
            t.register()
        # the residual path (dx += dy) is folded into the h-projection's data gradient (ConvOp.bwd_add)
        self.stage = plan._cur_stage
        plan.ops.append(self)

    def reads(self):
        return [self.f, self.g, self.h, self.x]

    def writes(self):
        return [self.y]

    def _args(self, bwd):
        return L.AttnArgs(_ptr(self.f.data), _ptr(self.g.data), _ptr(self.h.data),
                          _ptr(self.y.grad) if bwd else _ptr(self.x.data), _ptr(self.y.data), _ptr(self.stats),
                          _ptr(self.f.grad), _ptr(self.g.grad), _ptr(self.h.grad), _ptr(self.ws),
                          self.plan.B, self.N, self.f.Cs, self.h.Cs)

    def note(self):
        """forward: f, g, h, x read, y written (two sweeps: stats, output); backward: f, g, h, dy read, df, dg, dh written.
        flops: the N x N score matrix (2*N*N*d) is formed twice per sweep pair, h . beta costs 2*N*N*C."""
        P = self.plan
        esz = 4 if P.dtype == L.F32 else 2
        T = "f32" if P.dtype == L.F32 else "bf16"
        npx = P.B * self.N
        small, big = npx * self.f.Cs * esz, npx * self.h.Cs * esz
        self.fkey, self.bkey = f"selfattn_fwd<{T}>", f"selfattn_bwd<{T}>"
        self.fbytes = 2 * small + 3 * big
        ffl = 2.0 * P.B * self.N * self.N * (self.f.C + self.h.C)
        P.note_launch(self.fkey, self.fbytes, ffl)
        self.bbytes = 4 * small + 4 * big
        if P.training and self.y.grad is not None:
            P.note_launch(self.bkey, self.bbytes, 2.0 * ffl)

    def fwd_recs(self):
        self._fa = self._args(False)
        self.plan.rec_meta[C.addressof(self._fa)] = (self.fkey, self.fbytes)
        return [(L.OP_ATTN_FWD, self._fa)]

    def bwd_recs(self):
        if self.y.grad is None:
            return []
        self._ba = self._args(True)
        self.plan.rec_meta[C.addressof(self._ba)] = (self.bkey, self.bbytes)
        # (round 4: the core's backward as a side branch too -- released at the start of its stage's backward, the level-3 data gradients
        #  waiting for an event right behind it -- made the step 65 us SLOWER: in the backward the side queue is as loaded as the main one)
        return [(L.OP_ATTN_BWD, self._ba)]

    def fwd(self, s):
        L.call("msau_selfattn_fwd", s, self.plan.dtype, _ptr(self.f.data), _ptr(self.g.data), _ptr(self.h.data),
               _ptr(self.x.data), _ptr(self.y.data), _ptr(self.stats), self.plan.B, self.N, self.f.Cs, self.h.Cs, key=self.fkey)

    def bwd(self, s):
        if self.y.grad is None:
            return
        L.call("msau_selfattn_bwd", s, self.plan.dtype, _ptr(self.f.data), _ptr(self.g.data), _ptr(self.h.data),
               _ptr(self.y.grad), _ptr(self.stats), _ptr(self.f.grad), _ptr(self.g.grad), _ptr(self.h.grad),
               _ptr(self.ws), self.plan.B, self.N, self.f.Cs, self.h.Cs, key=self.bkey)


class ProjBwd:
    """The data gradients of an attention block's f, g, h projections (attention.py:152-154: three 1x1 convs reading the same
    tensor) as ONE launch (msau_attn_proj_bwd, csrc/attention_mfma.hip) instead of three accumulating msau_conv2d launches.
    Like PairOp not an entry of plan.ops: the ConvOps keep their packed images, weight-gradient launches and bookkeeping; the
    LAST of the three in backward order (f) emits the fused record, the other two emit no data gradient."""

    def __init__(self, plan: "Plan", f: ConvOp, g: ConvOp, h: ConvOp):
        self.plan, self.f, self.g, self.h = plan, f, g, h
        self.active = False
        self.args = None
        for c in (f, g, h):
            c.proj = self
        plan.projs.append(self)

    def bind(self):
        """after ConvOp.bind (the three descriptors say what the launches would have done), before late_bind (launch accounting)"""
        P, f, g, h = self.plan, self.f, self.g, self.h
        x = f.x1
        df, dg, dh = f.ddesc[0], g.ddesc[0], h.ddesc[0]
        if P.dtype != L.BF16 or P.act_flag or os.environ.get("MSAU_ATTN_PROJ_FUSE", "1") == "0" or None in (df, dg, dh):
            return
        if not (x.C == x.Cs == 64 and f.out.Cs == g.out.Cs == 8 and h.out.Cs == 64 and f.k == g.k == h.k == 1
                and g.x1 is x and h.x1 is x and f.x2 is None and g.x2 is None and h.x2 is None
                and f.dd_off is None and not (f.relu_in or g.relu_in or h.relu_in)):
            return
        # what the three launches would have carried, in backward order h, g, f: only the first may write, only the last may mask
        A, M, D = L.CONV_ACCUM, L.CONV_MASK_B, L.CONV_ADD
        if (dh.flags & ~(A | D)) or dg.flags != A or (df.flags & ~M) != A:
            return
        for c, kch in ((f, 32), (g, 32), (h, 64)):          # the image layout the kernel reads its weight fragments from
            gd = c._geom(c.out.Cs, 0, x.Cs, 1, 1, 1)
            if (gd.nchunks, gd.rows, gd.kchunk) != (1, 64, kch):
                return
        a = L.AttnProjBwdArgs()
        a.df, a.dg, a.dh = _ptr(f.out.grad), _ptr(g.out.grad), _ptr(h.out.grad)
        a.wf_pack, a.wg_pack, a.wh_pack = (P.pack_ptr(c.d_off[0]) for c in (f, g, h))
        a.add = dh.add if dh.flags & D else None
        a.mask_b = df.mask_b if df.flags & M else None
        a.dx, a.npix, a.C, a.accumulate = _ptr(x.grad), x.npix, x.Cs, int(bool(dh.flags & A))
        self.args, self.active = a, True

    def late_bind(self):
        if not self.active:
            return
        P, x = self.plan, self.f.x1
        esz = 2
        nops = 1 + (1 if self.args.add else 0) + self.args.accumulate + (1 if self.args.mask_b else 0)
        self.key = "attn_proj_bwd<bf16,C64>"
        self.bytes = x.npix * (self.f.out.Cs + self.g.out.Cs + self.h.out.Cs + nops * x.Cs) * esz
        self.flops = 2.0 * x.npix * x.C * (self.f.out.C + self.g.out.C + self.h.out.C)
        P.note_launch(self.key, self.bytes, self.flops)

    def recs(self):
        self.plan.rec_meta[C.addressof(self.args)] = (self.key, self.bytes)
        return [(L.OP_ATTN_PROJ_BWD, self.args)]

    def launch(self, s):
        L.call("msau_attn_proj_bwd", s, self.plan.dtype, C.byref(self.args), key=self.key)


class Plan:
    def __init__(self, cfg: dict, B: int, H: int, W: int, dtype: int, device, poff: Dict[str, int],
                 pshape: Dict[str, Tuple[int, ...]], training: bool = True, builder=None):
        self.cfg, self.B, self.H, self.W, self.dtype, self.device = cfg, B, H, W, dtype, device
        self.tdtype = torch.float32 if dtype == L.F32 else torch.bfloat16
        self.poff, self.pshape, self.training = poff, pshape, training
        # forward-only plans keep no activation beyond its last reader: buffers are handed out by liveness
        self.reuse = (not training) and bool(cfg.get("reuse_activations", True))
        # activation_name="elu" (model/model.py:412-416): the convs' output activation and its derivative on the generic kernels
        # (MSAU_CONV_ELU); the residual block's leading activation stays ReLU as in the reference (model.py:35,39)
        assert cfg.get("activation", "relu") in ("relu", "elu"), cfg.get("activation")
        self.act_flag = L.CONV_ELU if cfg.get("activation", "relu") == "elu" else 0
        self.head_probs = self.head_argmax = None
        self.head_fused = False
        self.acts: List[Act] = []
        self.ops: List[Op] = []
        self.pairs: List[PairOp] = []
        self.projs: List[ProjBwd] = []
        self.box_ops: List["BoxOp"] = []
        self.box_scratch = 0
        self.box_ws_ii = 0
        self.box_ws = None
        self.box_ii_g = None
        self._box_bwd_args: list = []
        self._pack_bytes = 0
        self._slab_elems = 0
        self._pack_entries: List[L.PackEntry] = []
        self._unpack_entries: List[L.UnpackEntry] = []
        self._unpack_stage: List[int] = []
        self._pack_max = 1
        self._unpack_max = 1
        self.launch_meta: Dict[str, Tuple[int, float, float]] = {}
        self.rec_meta: Dict[int, Tuple[str, float]] = {}       # address of a launch record's args -> (kernel key, bytes)
        self._cur_stage = 0
        # The weight gradients form no dependency chain (each reads a finished out.grad and writes its own slabs),
        # so they run on a side stream beside the data-gradient chain.  Measured 2026-10-03 with the lean kernels:
        # 5.76 -> 5.20 ms/step (with the first, generic kernels it was 4 % slower: both chains were issue-bound).
        self.overlap_wgrad = bool(cfg.get("overlap_wgrad", os.environ.get("MSAU_OVERLAP_WGRAD", "1") != "0")) and \
            str(device).startswith("cuda")
        self._side = None
        # bit-reproducible training steps: costs 4 % (the LRN backward launches wait for the weight-gradient stream)
        self.deterministic = bool(cfg.get("deterministic", os.environ.get("MSAU_DETERMINISTIC", "0") == "1"))
        self.overlap_max_pix = int(cfg.get("overlap_max_pix", 1 << 62))     # only layers this small go to the side stream
        self.x_in = Act(self, "input", H, W, cfg["channels"], needs_grad=bool(cfg.get("input_grad", False)))
        self.logits: Optional[Act] = None
        self.aux: Optional[Act] = None
        if builder is None:
            self._build_net()
        else:
            builder(self)               # custom graph (op-level tests): must set self.logits (and maybe self.aux)
        self._finish()

    # ---- arenas -----------------------------------------------------------------------------
    def alloc_pack(self, nbytes: int) -> int:
        off = self._pack_bytes
        self._pack_bytes += _ru(int(nbytes), 256)
        return off

    def alloc_slab(self, nelems: int) -> int:
        off = self._slab_elems
        self._slab_elems += _ru(int(nelems), 64)
        return off

    def add_pack_entry(self, e, nelems):
        self._pack_entries.append(e)
        self._pack_max = max(self._pack_max, int(nelems))

    def add_unpack_entry(self, e, nelems, stage=0):
        self._unpack_entries.append(e)
        self._unpack_stage.append(stage)
        self._unpack_max = max(self._unpack_max, int(nelems))

    def note_launch(self, key: str, nbytes: float, flops: float):
        """algorithmic bytes / flops of one launch, accumulated per kernel symbol for one train step"""
        c, b, f = self.launch_meta.get(key, (0, 0.0, 0.0))
        self.launch_meta[key] = (c + 1, b + nbytes, f + flops)

    def unnote_launch(self, key: str, nbytes: float, flops: float):
        c, b, f = self.launch_meta[key]
        if c <= 1:
            del self.launch_meta[key]
        else:
            self.launch_meta[key] = (c - 1, b - nbytes, f - flops)

    def pack_ptr(self, off: int) -> int:
        return self.pack_arena.data_ptr() + off

    def slab_ptr(self, off: int) -> int:
        return self.slab_arena.data_ptr() + 4 * off

    # ---- graph construction -------------------------------------------------------------------
    def _res_block(self, x0: Act, prefix: str, tag: str) -> Act:
        """MultiConvResidualBlock: model/model.py:37-50."""
        R, k = self.cfg["res_depth"], self.cfg["filter_size"]
        r_in, first = x0, None
        for i in range(R):
            o = Act(self, f"{tag}.res{i}", x0.H, x0.W, x0.C, relu_out=True)
            op = ConvOp(self, f"{tag}.res{i}", r_in, None, f"{prefix}.conv_res_list.{i}.custom_conv.weight",
                        f"{prefix}.conv_res_list.{i}.custom_conv.bias", o, k, relu_in=(i == 0), relu_out=True,
                        fwd_add=(x0 if i == R - 1 else None))
            first = first or op
            r_in = o
        first.bwd_add = r_in                    # d(x0) += g(block output): the in-place residual add
        if R == 2:
            PairOp(self, first, op)             # one launch per sweep when an instance takes the shape (bind() decides)
        return r_in

    def _box_block(self, x0: Act, prefix: str, tag: str) -> Act:
        """MultiBoxConvBlock: model/model_box.py:51-59 -- ReLU, then num_box_convs x [BoxConv2d(c -> F*c) -> conv1x1(F*c -> c)]
        (ReLU after every 1x1 but the last), residual add, ReLU."""
        n, Fn = self.cfg["num_box_convs"], self.cfg["num_box_per_channels"]
        r_in, first = x0, None
        for i in range(n):
            bx = Act(self, f"{tag}.box{i}", x0.H, x0.W, x0.C * Fn)
            bop = BoxOp(self, f"{tag}.box{i}", r_in, bx, f"{prefix}.conv_list.{2 * i}", relu_in=(i == 0))
            first = first or bop
            o = Act(self, f"{tag}.bres{i}", x0.H, x0.W, x0.C, relu_out=True)
            ConvOp(self, f"{tag}.bres{i}", bx, None, f"{prefix}.conv_list.{2 * i + 1}.custom_conv.weight",
                   f"{prefix}.conv_list.{2 * i + 1}.custom_conv.bias", o, 1, relu_out=True, fwd_add=(x0 if i == n - 1 else None))
            r_in = o
        first.bwd_add = r_in
        return r_in

    def _build_net(self):
        cfg = self.cfg
        box = cfg.get("variant") == "box"
        block = (lambda x0, pre, tag: self._box_block(x0, pre.replace(".conv_res_list.", ".conv_box_list."), tag)) if box else self._res_block
        S, Fr, k, nb = cfg["scale_space_num"], cfg["featRoot"], cfg["filter_size"], cfg.get("num_blocks", 3)
        ncls = cfg["n_class"]
        assert cfg["pool_size"] == 2
        inp = self.x_in
        prev_dw = prev_up = None
        self.stage_logits: List[Act] = []
        for b in range(nb):
            self._cur_stage = b
            coupled, last = b > 0, b == nb - 1
            pd = f"msau_net.blocks.{b}.downsamplingblock"
            pu = f"msau_net.blocks.{b}.upsamplingblock"
            dw: Dict[int, Act] = {}
            x_in = inp
            x2 = None
            for l in range(S):                                            # model.py:136-162
                c = Fr * 2 ** l
                t = f"s{b}.d{l}"
                a = Act(self, t + ".a", x_in.H, x_in.W, c)
                ConvOp(self, t + ".dil", x_in, None, f"{pd}.conv1s.{l}.conv.weight", f"{pd}.conv1s.{l}.conv.bias", a, k,
                       dil=2 ** l)
                x0 = Act(self, t + ".lrn", a.H, a.W, c)
                LrnOp(self, t + ".lrn", a, x0)
                x1 = block(x0, f"{pd}.conv_res_list.{l}", t)
                if coupled:                                               # model.py:143-148
                    x2 = Act(self, t + ".cpl", a.H, a.W, c, relu_out=True)
                    ConvOp(self, t + ".cpl", prev_dw[l], x1, f"{pd}.conv1_1s.{l}.custom_conv.weight",
                           f"{pd}.conv1_1s.{l}.custom_conv.bias", x2, 1, relu_out=True)
                else:
                    x2 = x1
                if l == S - 1 and not last:                               # model.py:149-150 (dead in the last stage)
                    pa = f"{pd}.layer_attentions.attention_block"
                    fa = Act(self, t + ".f", a.H, a.W, c // 8)
                    ga = Act(self, t + ".g", a.H, a.W, c // 8)
                    ha = Act(self, t + ".h", a.H, a.W, c)
                    fop = ConvOp(self, t + ".f", x2, None, f"{pa}.f.conv.weight", f"{pa}.f.conv.bias", fa, 1)
                    gop = ConvOp(self, t + ".g", x2, None, f"{pa}.g.conv.weight", f"{pa}.g.conv.bias", ga, 1)
                    hop = ConvOp(self, t + ".h", x2, None, f"{pa}.h.conv.weight", f"{pa}.h.conv.bias", ha, 1)
                    if self.training:
                        ProjBwd(self, fop, gop, hop)
                    y = Act(self, t + ".attn", a.H, a.W, c)
                    AttnCoreOp(self, t + ".attn", fa, ga, ha, x2, y)
                    hop.bwd_add = y
                    dw[l] = y
                else:
                    dw[l] = x2
                if l < S - 1:                                             # model.py:158-160
                    pooled = Act(self, t + ".pool", (x2.H + 1) // 2, (x2.W + 1) // 2, c, relu_out=False)
                    PoolOp(self, t + ".pool", x2, pooled)
                    x_in = pooled
            cur = x2                                                      # pre-attention tensor: model.py:162-164
            up: Dict[int, Act] = {}
            for l in range(S - 2, -1, -1):                                # model.py:226-254
                c = Fr * 2 ** l
                t = f"s{b}.u{l}"
                skip = dw[l]
                d = Act(self, t + ".deconv", skip.H, skip.W, c)
                ConvOp(self, t + ".deconv", cur, None, f"{pu}.deconvs.{l}.conv.weight", f"{pu}.deconvs.{l}.conv.bias", d, k,
                       kind="deconv")
                xm = Act(self, t + ".merge", skip.H, skip.W, c)
                ConvOp(self, t + ".merge", skip, d, f"{pu}.conv1s.{l}.custom_conv.weight",
                       f"{pu}.conv1s.{l}.custom_conv.bias", xm, k)
                x1 = block(xm, f"{pu}.conv_res_list.{l}", t)
                if coupled:
                    x2 = Act(self, t + ".cpl", skip.H, skip.W, c, relu_out=True)
                    ConvOp(self, t + ".cpl", prev_up[l], x1, f"{pu}.conv1_1s.{l}.custom_conv.weight",
                           f"{pu}.conv1_1s.{l}.custom_conv.bias", x2, 1, relu_out=True)
                else:
                    x2 = x1
                up[l] = x2
                cur = x2
            logits = Act(self, f"s{b}.logits", self.H, self.W, ncls)
            ConvOp(self, f"s{b}.end", cur, None, f"msau_net.end_convs.{b}.custom_conv.weight",
                   f"msau_net.end_convs.{b}.custom_conv.bias", logits, 4,          # model.py:390, 375-376
                   head=(not self.training and b == nb - 1 and ncls <= 255))
            self.stage_logits.append(logits)
            inp, prev_dw, prev_up = logits, dw, up
        self.logits = self.stage_logits[-1]
        self.aux = self.stage_logits[-2] if nb >= 2 else None              # model.py:392-393

    def _finish(self):
        # external gradient (loss) contributions come last in forward order -> first in backward
        self.ext_slot = {}
        if self.training:
            for t in (self.logits, self.aux):
                if t is not None:
                    self.ext_slot[t.name] = t.register()
        # ---- allocate activations (forward-only), gradients, pack images, descriptors
        if self.reuse:
            self._assign_buffers()
        if not self.training and self.logits.C <= 255:
            lg = self.logits
            self.head_probs = torch.zeros((self.B, lg.H, lg.W, lg.C), dtype=torch.float32, device=self.device)
            self.head_argmax = torch.zeros((self.B, lg.H, lg.W), dtype=torch.uint8, device=self.device)
        for a in self.acts:
            a.alloc_grad()
        for op in self.ops:
            op.finalize()
        if self.box_ops:
            self.box_ii_g = torch.zeros((max(self.box_scratch, 1),), dtype=torch.float32, device=self.device)
            self.box_ws = torch.zeros((max(self.box_ws_ii, 1),), dtype=torch.float32, device=self.device)
        self.pack_arena = torch.zeros(max(self._pack_bytes, 256), dtype=torch.uint8, device=self.device)
        for op in self.ops:
            if isinstance(op, ConvOp):
                op.bind()
        for pr in self.pairs:
            pr.bind()                    # (before the slab arena exists: a pair that takes over a weight gradient sizes its slabs)
        for pj in self.projs:
            pj.bind()
        self.slab_arena = torch.zeros(max(self._slab_elems, 64), dtype=torch.float32, device=self.device)
        for op in self.ops:
            if isinstance(op, ConvOp):
                op.late_bind()
        for pj in self.projs:
            pj.late_bind()
        for pr in self.pairs:
            pr.late_bind()
            pr.note()
        for op in self.ops:
            if not isinstance(op, ConvOp):
                op.note()
        self._note_boundary()
        # The bottleneck attention of a stage (f, g, h projections + core: model/layers/attention.py:152-162) feeds ONLY the next stage's
        # level-3 coupling (model/model.py:149-150: the decoder consumes the pre-attention tensor), so in the forward sweep it is a side
        # branch: its launches go to the side stream -- idle during the forward -- beside the stage's decoder, and the consumer of its
        # output joins.  Training plans only (forward-only plans recycle activation buffers by sequential liveness).
        side_ops, join_ops = set(), set()
        if self.training and self.overlap_wgrad and os.environ.get("MSAU_ATTN_SIDE", "1") != "0":
            for op in self.ops:
                if isinstance(op, AttnCoreOp):
                    branch = [c for c in self.ops if isinstance(c, ConvOp) and c.out in (op.f, op.g, op.h)]
                    # every reader of the branch's output, of any op type; the FIRST one must be a conv that emits a launch record of
                    # its own (it carries the join) -- anything else keeps the branch on the main stream
                    users = [c for c in self.ops if c is not op and any(t is op.y for t in c.reads())]
                    later = self.ops[self.ops.index(op) + 1:]
                    first = min(users, key=self.ops.index) if users else None
                    if len(branch) == 3 and users and all(u in later for u in users) and \
                            isinstance(first, ConvOp) and first.fwd_recs() and \
                            not any(t in (op.f, op.g, op.h) for o2 in later for t in o2.reads()):
                        side_ops.update(id(c) for c in branch)
                        side_ops.add(id(op))
                        join_ops.add(id(first))
        frecs = []
        for op in self.ops:
            for kind, args in op.fwd_recs():
                if id(op) in side_ops:
                    kind |= L.OP_SIDE
                elif id(op) in join_ops:
                    kind |= L.OP_JOIN
                    join_ops.discard(id(op))                 # (a fused pair emits one record; only the first record of an op joins)
                frecs.append((kind, args))
        assert not join_ops, "a side-stream attention branch lost its join: its first reader emitted no launch record"
        self._fwd_side = bool(side_ops)

        self._fwd_seq = self._make_seq(frecs)
        self.pack_table = self._upload(self._pack_entries, L.PackEntry) if self._pack_entries else None
        self.unpack_table = self._upload(self._unpack_entries, L.UnpackEntry) if self._unpack_entries else None
        self._bwd_seq, self._bwd_segs, self._reduce_args = None, [], []
        if self.training:
            # backward = stages in reverse; after a stage's last op its slabs are reduced into the flat gradient
            # (on the side stream, behind that stage's weight gradients) while the next stage's backward runs
            recs = []
            stages = sorted({op.stage for op in self.ops}, reverse=True)
            esz = C.sizeof(L.UnpackEntry)
            def reduce_rec(group):
                idx = [i for i, st in enumerate(self._unpack_stage) if st == group]
                if not idx:
                    return
                assert idx == list(range(idx[0], idx[-1] + 1)), "unpack entries of a group must be contiguous"
                ra = L.ReduceArgs(self.slab_arena.data_ptr(), None, self.unpack_table.data_ptr() + idx[0] * esz,
                                  len(idx), self._unpack_max)
                self._reduce_args.append(ra)
                recs.append((L.OP_WGRAD_REDUCE | (L.OP_SIDE if self.overlap_wgrad else 0), ra))

            for b in stages:
                start = len(recs)
                ops_b = [op for op in reversed(self.ops) if op.stage == b]
                cut = next((i for i, op in enumerate(ops_b) if isinstance(op, ConvOp) and op.reduce_group() != b), len(ops_b))
                assert all(not isinstance(op, ConvOp) or op.reduce_group() != b for op in ops_b[cut:]), \
                    "separately reduced convs must close their stage's backward"
                recs += [r for op in ops_b[:cut] for r in op.bwd_recs()]
                reduce_rec(b)
                if cut < len(ops_b):
                    recs += [r for op in ops_b[cut:] for r in op.bwd_recs()]
                    reduce_rec(-1)
                self._bwd_segs.append((b, start, len(recs) - start))
            self._bwd_seq = self._make_seq(recs)
        HW = self.H * self.W
        lg = self.logits
        self.out_logits = torch.zeros((self.B, lg.C, lg.H, lg.W), dtype=torch.float32, device=self.device)
        self.out_aux = torch.zeros((self.B, self.aux.C, self.aux.H, self.aux.W), dtype=torch.float32,
                                   device=self.device) if self.aux is not None else None
        if self.training:
            # label counts: a sample spread over K workgroups (msau_label_counts_split), the K integers added up by the CE launch
            self.counts_k = max(1, min(16, 256 // self.B)) if self.B <= 1024 and os.environ.get("MSAU_LABEL_SPLIT", "1") != "0" else 1
            self.counts = torch.zeros((self.B * self.counts_k,), dtype=torch.int32, device=self.device)
            self.ce_ws = torch.zeros((int(L.load().msau_ce_ws_floats(self.B * HW)),), dtype=torch.float32, device=self.device)
            self.ce_multi_ws = torch.zeros((int(L.load().msau_ce_multi_ws_floats(self.B * HW)),), dtype=torch.float32, device=self.device)
            self.loss_buf = torch.zeros((1,), dtype=torch.float32, device=self.device)

    def _assign_buffers(self):
        """Forward-only: give every activation a buffer that is free from its producer to its last reader.  Buffers are
        pooled by exact byte size (the sizes repeat per level), a producer never gets a buffer one of its own inputs
        still occupies, and the tensors read after the sweep (all stage logits: aux + final) are never released."""
        last_read: Dict[int, int] = {}
        for i, op in enumerate(self.ops):
            for t in op.reads():
                last_read[id(t)] = i
        keep = {id(t) for t in getattr(self, "stage_logits", [])} | {id(t) for t in (self.logits, self.aux) if t is not None}
        free: Dict[int, List[torch.Tensor]] = {}
        self.buffers: List[torch.Tensor] = []

        def take(a: Act):
            n = self.B * a.H * a.W * a.Cs
            pool = free.setdefault(n, [])
            if pool:
                buf = pool.pop()
            else:
                buf = torch.zeros((n,), dtype=self.tdtype, device=self.device)
                self.buffers.append(buf)
            a.data = buf.view(self.B, a.H, a.W, a.Cs)
            a._buf = buf

        def release(a: Act):
            if id(a) not in keep and a.data is not None:
                free[a._buf.numel()].append(a._buf)

        take(self.x_in)
        for i, op in enumerate(self.ops):
            for t in op.writes():
                if t.data is None:
                    take(t)
            for t in set(op.reads()) | set(op.writes()):
                if last_read.get(id(t), -1) <= i and (t in op.reads() or id(t) not in last_read):
                    release(t)
        for a in self.acts:                       # anything no op touches (custom builders)
            if a.data is None:
                take(a)

    def _note_boundary(self):
        """launches outside the op list: parameter packing, NCHW fp32 -> NHWC conversion of the API input, masked CE (+ label
        counting), slab reduction, clip + Adam -- so that `launch_meta` sums to the whole step"""
        esz = 4 if self.dtype == L.F32 else 2
        a = self.x_in
        self.in_bytes = self.B * a.C * a.H * a.W * 4 + a.npix * a.Cs * esz
        c = self._nchw_first_conv()
        if c is None:
            self.note_launch("msau_nchw_to_nhwc", self.in_bytes, 0.0)
        else:
            # the first conv reads the API tensor itself (MSAU_CONV_NCHW, csrc/conv_first.hip): the conversion launch and the conv's
            # read of the converted copy disappear; the copy is still written for the backward's weight gradient (training plans)
            self.unnote_launch(c.fkey, c.fbytes, c.flops)
            T = "f32" if self.dtype == L.F32 else "bf16"
            c.fkey = f"first_conv_nchw<{T},CIN{a.Cs},CO{c.out.Cs}>"
            c.fbytes = self.B * a.C * a.H * a.W * 4 + (a.npix * a.Cs * esz if self.training else 0) + c.out.npix * c.out.Cs * esz
            self.note_launch(c.fkey, c.fbytes, c.flops)
        nparam = sum(int(math.prod(shp)) for shp in self.pshape.values())
        self.note_launch("msau_pack_params", nparam * 4 + self._pack_bytes, 0.0)
        if self.training:
            HW = self.H * self.W
            lg = self.logits
            nl = 2 if self.aux is not None else 1
            self.note_launch("msau_label_counts", self.B * HW * 8, 0.0)
            self.ce_bytes = self.B * HW * (8 + nl * 2 * lg.Cs * esz)
            self.note_launch("msau_masked_ce_multi" if lg.Cs <= 16 else "msau_masked_ce", self.ce_bytes, 0.0)
            self.reduce_bytes = self._slab_elems * 4 + nparam * 4
            self.note_launch("msau_wgrad_reduce", self.reduce_bytes, 0.0)
            self.note_launch("msau_clip_adam_step", nparam * 4 * 8, 0.0)       # g twice, p, m, v read; p, m, v written

    def set_probe_keys(self, keys, side: bool = False) -> List[Tuple[str, float]]:
        """Mark every launch of ONE stream (main, or with side=True the weight-gradient side stream) whose kernel key is in
        `keys` (None: unmark all) with MSAU_OP_PROBE in the forward and backward sequences.  Returns [(key, algorithmic
        bytes)] of the marked launches in the order `read_probe` reports their durations: sequence order, which is also the
        enqueue order as long as only one stream is probed at a time (side launches are released in batches; a probed
        weight gradient is launched on its own, not inside a grouped grid)."""
        order = []
        for seq in (self._fwd_seq, self._bwd_seq):
            if seq is None:
                continue
            arr, cnt, _ = seq
            for i in range(cnt):
                meta = self.rec_meta.get(arr[i].args)
                if keys is not None and meta is not None and meta[0] in keys and bool(arr[i].kind & L.OP_SIDE) == side:
                    arr[i].kind |= L.OP_PROBE
                    order.append(meta)
                else:
                    arr[i].kind &= ~L.OP_PROBE
        return order

    def set_probe(self, key: Optional[str]) -> int:
        """Mark (or with None: unmark) every conv launch whose kernel symbol is `key` with MSAU_OP_PROBE in the forward
        and backward sequences; the next sweeps then time those launches in place (read with `read_probe`).
        Returns the number of marked launches per step."""
        want = set()
        if key is not None:
            for op in self.ops:
                if isinstance(op, ConvOp):
                    if op.fkey == key:
                        want.add(C.addressof(op.fdesc))
                    for si, dd in enumerate(op.ddesc):
                        if dd is not None and op.dmeta[si][0] == key:
                            want.add(C.addressof(dd))
            for pr in self.pairs:
                if pr.active and pr.key == key:
                    want.add(C.addressof(pr.fdesc))
                    if pr.bdesc is not None:
                        want.add(C.addressof(pr.bdesc))
        n = 0
        for seq in (self._fwd_seq, self._bwd_seq):
            if seq is None:
                continue
            arr, cnt, _ = seq
            for i in range(cnt):
                if arr[i].args in want:
                    arr[i].kind |= L.OP_PROBE
                    n += 1
                else:
                    arr[i].kind &= ~L.OP_PROBE
        return n

    @staticmethod
    def read_probe(cap: int = 1 << 16):
        """durations (us) of the probed launches since the last read, in launch order"""
        buf = (C.c_float * cap)()
        n = C.c_int(0)
        L.call("msau_probe_read", buf, cap, C.byref(n))
        return list(buf[:n.value])

    def _make_seq(self, recs):
        """(kind, args struct) records -> (msau_op array, n); the structs are kept alive by the ops / this list"""
        arr = (L.Op * max(len(recs), 1))()
        for i, (kind, args) in enumerate(recs):
            arr[i].kind, arr[i].dtype, arr[i].args = kind, self.dtype, C.addressof(args)
        return arr, len(recs), recs

    def _run_seq(self, seq, s):
        L.call("msau_run_ops", s, seq[0], seq[1])

    def _upload(self, entries, ctype):
        n = len(entries)
        arr = (ctype * n)(*entries)
        raw = bytes(memoryview(arr))
        host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        return host.to(self.device)

    # ---- execution ----------------------------------------------------------------------------
    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    def pack(self, flat_params: torch.Tensor):
        if self.pack_table is None:
            return
        L.call("msau_pack_params", self._stream(), flat_params.data_ptr(), self.pack_arena.data_ptr(),
               self.pack_table.data_ptr(), len(self._pack_entries), self._pack_max)
        for bop in self.box_ops:                     # stored box parameters -> valid boxes in pixels (+ reflected)
            bop.load_params(self._stream(), flat_params)

    def load_input(self, x_nchw: torch.Tensor):
        assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous() and tuple(x_nchw.shape) == \
            (self.B, self.cfg["channels"], self.H, self.W), (x_nchw.shape, x_nchw.dtype)
        a = self.x_in
        L.call("msau_nchw_to_nhwc", self._stream(), self.dtype, x_nchw.data_ptr(), a.data.data_ptr(), self.B, a.C, a.Cs, self.H, self.W)

    def _nchw_first_conv(self):
        """the net's first conv if msau_conv2d takes it with MSAU_CONV_NCHW (the API's fp32 NCHW tensor as its input), else None"""
        if not hasattr(self, "_nchw_conv"):
            c = next((op for op in self.ops if isinstance(op, ConvOp) and op.x1 is self.x_in and op.x2 is None), None)
            ok = c is not None and os.environ.get("MSAU_FIRST_NCHW", "1") != "0" and not (c.pair is not None and c.pair.active) \
                and not (c.fdesc.flags & ~L.CONV_RELU_OUT)
            if ok:
                info = (L.i32 * 8)()
                L.call("msau_conv2d_launch_info", self.dtype, C.byref(c.fdesc), info)
                ok = bool(info[7] & 64)
            self._nchw_conv = c if ok else None
            self._nchw_on = False
        return self._nchw_conv

    def _feed_nchw(self, x_nchw: Optional[torch.Tensor]) -> bool:
        """Switch the net's first conv between the plan's NHWC input buffer and the API's fp32 NCHW tensor `x_nchw` (MSAU_CONV_NCHW:
        the conv converts while it loads and writes the NHWC copy the backward's weight gradient reads; same bits as
        msau_nchw_to_nhwc + the dense launch).  Returns True when the conv takes the tensor: no conversion launch then.  Call it
        after _feed_ids / _feed_owner (they reset the descriptor's input).  The descriptor is launched by value."""
        c = self._nchw_first_conv()
        if c is None:
            return False
        if x_nchw is not None:
            assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous() and tuple(x_nchw.shape) == \
                (self.B, self.cfg["channels"], self.H, self.W), (x_nchw.shape, x_nchw.dtype)
            if x_nchw.data_ptr() % 16:
                x_nchw = None                                   # (a view at an odd offset: the plain conversion takes any alignment)
        if x_nchw is not None:
            c.fdesc.flags |= L.CONV_NCHW
            c.fdesc.x1 = x_nchw.data_ptr()
            c.fdesc.y2 = _ptr(self.x_in.data) if self.training else None
            c.fdesc.head_classes = self.x_in.C
            self._nchw_on = True
            return True
        if self._nchw_on:
            c.fdesc.flags &= ~L.CONV_NCHW
            c.fdesc.x1 = _ptr(self.x_in.data)
            c.fdesc.y2 = None
            c.fdesc.head_classes = 0
            self._nchw_on = False
        return False

    def _feed_ids(self, ids: Optional[torch.Tensor]) -> bool:
        """Switch the net's first conv (and its weight gradient) between the dense input tensor and an id mask
        (MSAU_CONV_IDS: the one-hot tile is synthesised in LDS, nothing is painted or read; same bits as the dense launch).
        Returns True when the id mask feeds the conv directly; False = paint the dense one-hot input (`load_ids`).  The
        descriptors are launched by value, so flipping them between sweeps is safe."""
        self._feed_nchw(None)
        if not hasattr(self, "_ids_conv"):
            c = next((op for op in self.ops if isinstance(op, ConvOp) and op.x1 is self.x_in and op.x2 is None), None)
            ok = c is not None and os.environ.get("MSAU_IDS_DIRECT", "1") != "0" and not (c.pair is not None and c.pair.active)
            if ok:
                probe = L.ConvDesc.from_buffer_copy(c.fdesc)
                probe.flags |= L.CONV_IDS
                info = (L.i32 * 8)()
                L.call("msau_conv2d_launch_info", self.dtype, C.byref(probe), info)
                ok = bool(info[7] & 16)
            if ok and self.training and c.wdesc is not None:        # the id-mask weight gradient is the bf16 64 -> 8 instance
                ok = self.dtype == L.BF16 and c.wgeom.lean and c.wdesc.C1 == 64 and c.wdesc.Cout == 8 and c.wgeom.nchunks == 1
            self._ids_conv = c if ok else None
            self._ids_keep = None
        c = self._ids_conv
        if c is None:
            return False
        if ids is not None:
            c.fdesc.flags |= L.CONV_IDS
            c.fdesc.x1 = ids.data_ptr()
            if c.wdesc is not None:
                c.wdesc.flags |= L.CONV_IDS
                c.wdesc.x1 = ids.data_ptr()
            self._ids_keep = ids                                    # read again by the backward's weight gradient
        else:
            c.fdesc.flags &= ~L.CONV_IDS
            c.fdesc.x1 = _ptr(self.x_in.data)
            if c.wdesc is not None:
                c.wdesc.flags &= ~L.CONV_IDS
                c.wdesc.x1 = _ptr(self.x_in.data)
            self._ids_keep = None
        return True

    def _feed_owner(self, src, flat_params: Optional[torch.Tensor] = None) -> bool:
        """Switch the net's first conv (and its weight gradient) between the painted input tensor and BOX LISTS
        (MSAU_CONV_OWNER, csrc/ownerconv.hip): `src` = (owner int32 [B,H,W], boxes int32 [n,6] device tensor or None, n, feats
        fp32 [n_vec, C] device tensor) or None.  Returns False when no instance takes the conv (paint the tensor instead)."""
        self._feed_nchw(None)
        if not hasattr(self, "_owner_conv"):
            c = next((op for op in self.ops if isinstance(op, ConvOp) and op.x1 is self.x_in and op.x2 is None), None)
            ok = c is not None and os.environ.get("MSAU_OWNER_CONV", "1") != "0" and not (c.pair is not None and c.pair.active) \
                and not (c.fdesc.flags & ~L.CONV_RELU_OUT)
            if ok:
                info = (L.i32 * 8)()
                L.call("msau_conv2d_launch_info", self.dtype, C.byref(c.fdesc), info)
                ok = bool(info[7] & 32)
            if ok and self.training and c.wdesc is not None:
                ok = c.wdesc.flags == 0 and c.uentry is not None and c.kind == "conv"
            self._owner_conv = c if ok else None
            self._owner_keep = None
            self._owner_slabs = None
        c = self._owner_conv
        if c is None:
            return False
        if src is None:
            if self._owner_keep is not None:
                c.fdesc.flags &= ~L.CONV_OWNER
                c.fdesc.x1 = _ptr(self.x_in.data)
                if c.wdesc is not None:
                    c.wdesc.flags &= ~L.CONV_OWNER
                    c.wdesc.x1 = _ptr(self.x_in.data)
                    self._set_unpack_slabs(c, c.wdesc.nslabs)
                self._owner_keep = None
            return True
        owner, boxes, n_boxes, feats = src
        assert owner.dtype == torch.int32 and owner.is_contiguous() and tuple(owner.shape) == (self.B, self.H, self.W)
        assert feats.dtype == torch.float32 and feats.is_contiguous() and feats.dim() == 2 and int(feats.shape[1]) == self.x_in.C, feats.shape
        self._feed_ids(None)
        n_vec = int(feats.shape[0])
        ctx = L.OwnerCtx()
        blocks = max(1, min(256, c.out.npix // 1024))
        wt = torch.empty((_ru(self.x_in.C, 32) * 72,), dtype=torch.float32, device=self.device)      # msau_owner_ctx.wt_floats
        table = torch.empty((max(n_vec, 1) * 72,), dtype=torch.float32, device=self.device)
        sums = torch.empty((max(n_boxes, 1) * 72,), dtype=torch.float32, device=self.device)
        csum = torch.empty((blocks * 8,), dtype=torch.float32, device=self.device)
        ctx.owner, ctx.boxes, ctx.feats = owner.data_ptr(), (boxes.data_ptr() if n_boxes else None), feats.data_ptr()
        ctx.w = flat_params.data_ptr() + 4 * self.poff[c.wname]
        ctx.wt, ctx.table, ctx.sums, ctx.csum = wt.data_ptr(), table.data_ptr(), sums.data_ptr(), csum.data_ptr()
        ctx.n_boxes, ctx.n_vec, ctx.C, ctx.csum_blocks, ctx.wt_floats = n_boxes, n_vec, self.x_in.C, blocks, wt.numel()
        c.fdesc.flags |= L.CONV_OWNER
        c.fdesc.x1 = C.addressof(ctx)
        if c.wdesc is not None:
            c.wdesc.flags |= L.CONV_OWNER
            c.wdesc.x1 = C.addressof(ctx)
            self._set_unpack_slabs(c, int(L.load().msau_owner_slabs(C.byref(c.wdesc))))     # the slabs the box-list weight gradient writes
        self._owner_keep = (ctx, owner, boxes, feats, wt, table, sums, csum, flat_params)  # read again by the backward
        return True

    def _set_unpack_slabs(self, c: "ConvOp", n: int):
        """patch the slab count of a conv's entry in the device-side reduction table"""
        if self._owner_slabs == n or self.unpack_table is None:
            return
        idx = next(i for i, e in enumerate(self._unpack_entries) if e is c.uentry)
        off = idx * C.sizeof(L.UnpackEntry) + L.UnpackEntry.nslabs.offset
        self.unpack_table[off:off + 4].copy_(torch.tensor([n], dtype=torch.int32).view(torch.uint8))
        self._owner_slabs = n

    def load_ids(self, ids: torch.Tensor):
        """Paint the one-hot input from a character-id mask int32 [B,H,W] (to_categorical, generic_util.py:97-98):
        H*W*4 bytes cross PCIe instead of the dense H*W*C float grid."""
        assert ids.dtype == torch.int32 and ids.is_contiguous() and tuple(ids.shape) == (self.B, self.H, self.W), (ids.shape, ids.dtype)
        a = self.x_in
        L.call("msau_onehot_ids", self._stream(), self.dtype, ids.data_ptr(), a.data.data_ptr(), a.npix, a.C, a.Cs)

    def predict(self, flat_params: torch.Tensor, x_nchw: Optional[torch.Tensor] = None, ids: Optional[torch.Tensor] = None):
        """Forward-only sweep ending in the inference head -> (probs fp32 [B,H,W,n_class], argmax uint8 [B,H,W]),
        the NHWC layout `KVModel._extract_value` consumes (kv_model.py:305-313).  The buffers are the plan's own."""
        assert not self.training and self.head_probs is not None
        s = self._stream()
        self.pack(flat_params)
        if ids is not None:
            assert ids.dtype == torch.int32 and ids.is_contiguous() and tuple(ids.shape) == (self.B, self.H, self.W), (ids.shape, ids.dtype)
            if not self._feed_ids(ids):
                self.load_ids(ids)
        else:
            self._feed_ids(None)
            if not self._feed_nchw(x_nchw):
                self.load_input(x_nchw)
        self._run_seq(self._fwd_seq, s)
        if not self.head_fused:
            lg = self.logits
            L.call("msau_softmax_argmax_nhwc", s, self.dtype, lg.data.data_ptr(), self.head_probs.data_ptr(),
                   self.head_argmax.data_ptr(), lg.npix, lg.C, lg.Cs)
        return self.head_probs, self.head_argmax

    def input_buffer(self, k: int) -> torch.Tensor:
        """Input buffer k (0 = the plan's own, 1 = a second one allocated on first use): a producer on another stream paints
        the NEXT batch into the buffer this step does not read (TrainEngine.prefetch_boxes)."""
        if not hasattr(self, "_x_bufs"):
            self._x_bufs = [self.x_in.data, None]
        if self._x_bufs[k] is None:
            self._x_bufs[k] = torch.zeros_like(self.x_in.data)
        return self._x_bufs[k]

    def use_input(self, k: int):
        """Point the launches that read the net's input (the first conv and its weight gradient) at input buffer k.  The
        descriptors are launched by value, so flipping them between sweeps is safe."""
        buf = self.input_buffer(k)
        if buf is self.x_in.data:
            return
        self._feed_ids(None)
        old = self.x_in.data.data_ptr()
        for op in self.ops:
            if isinstance(op, ConvOp):
                for d in (op.fdesc, op.wdesc):
                    if d is not None:
                        for f in ("x1", "x2"):
                            if getattr(d, f) == old:
                                setattr(d, f, buf.data_ptr())
        self.x_in.data = buf

    @property
    def input_nhwc(self) -> torch.Tensor:
        """The net's input buffer: [B][H][W][Cs] in the storage dtype, channels beyond `channels` zero.  A producer on the
        device (the chargrid painters, msau_amd/data/raster.py) writes here and calls forward(..., nhwc_ready=True)."""
        return self.x_in.data

    def forward(self, flat_params: torch.Tensor, x_nchw: Optional[torch.Tensor], export: bool = True,
                ids: Optional[torch.Tensor] = None, nhwc_ready: bool = False, owner=None, single_stream: bool = False):
        """`ids` (int32 [B,H,W] character ids, -1 = empty) instead of `x_nchw`: the one-hot grid is painted on the device.
        `nhwc_ready`: the input buffer (`input_nhwc`) already holds the grid -- no boundary conversion at all."""
        s = self._stream()
        self.pack(flat_params)
        if owner is not None:
            # box lists instead of a painted input (`owner` as Plan._feed_owner takes it); the caller checked that an instance exists
            assert x_nchw is None and ids is None and not nhwc_ready
            ok = self._feed_owner(owner, flat_params)
            assert ok, "no MSAU_CONV_OWNER instance for this plan's first conv"
        elif getattr(self, "_owner_keep", None) is not None:
            self._feed_owner(None)
        if owner is not None:
            pass
        elif nhwc_ready:
            assert x_nchw is None and ids is None
            self._feed_ids(None)
        elif ids is not None:
            assert ids.dtype == torch.int32 and ids.is_contiguous() and tuple(ids.shape) == (self.B, self.H, self.W), (ids.shape, ids.dtype)
            if not self._feed_ids(ids):
                self.load_ids(ids)
        else:
            self._feed_ids(None)
            if not self._feed_nchw(x_nchw):
                self.load_input(x_nchw)
        if L._profiler is None:
            # a sweep that is being captured into a HIP graph stays on ONE stream whoever the caller is (a captured fork onto the
            # side stream is the configuration of profiles/r04_graph_destroy.md; creating the side stream synchronises the device,
            # which is illegal during capture)
            single_stream = single_stream or (str(self.device).startswith("cuda") and torch.cuda.is_current_stream_capturing())
            if self._fwd_side and not single_stream:
                if self._side is None:
                    self._side = L.concurrent_stream(self.device)
                L.call("msau_run_ops_overlap", s, self._side.cuda_stream, self._fwd_seq[0], self._fwd_seq[1], 1)
            else:
                self._run_seq(self._fwd_seq, s)      # one C call enqueues the whole forward sweep
        else:
            for op in self.ops:
                op.fwd(s)
        if export:
            self.export_logits()
        return self.out_logits, self.out_aux

    def export_logits(self):
        s = self._stream()
        for act, dst in ((self.logits, self.out_logits), (self.aux, self.out_aux)):
            if act is not None:
                L.call("msau_nhwc_to_nchw", s, self.dtype, act.data.data_ptr(), dst.data_ptr(), self.B, act.C, act.Cs, act.H, act.W)

    def set_external_grads(self, g_logits: Optional[torch.Tensor], g_aux: Optional[torch.Tensor]):
        """Gradients w.r.t. the NCHW fp32 outputs (the autograd path)."""
        s = self._stream()
        for act, g in ((self.logits, g_logits), (self.aux, g_aux)):
            if act is None:
                continue
            accum, maskb = act.slot_flags(self.ext_slot[act.name])
            assert not accum
            if g is None:
                act.grad.zero_()
            else:
                g = g.contiguous().float()
                L.call("msau_nchw_grad_to_nhwc", s, self.dtype, g.data_ptr(), act.grad.data_ptr(), self.B, act.C, act.Cs,
                       act.H, act.W, 0)
                if maskb:       # output produced through a ReLU and consumed by nothing else (op-level tests only)
                    act.grad.mul_(act.data > 0)

    def loss_grads(self, labels: torch.Tensor) -> torch.Tensor:
        """Masked CE (model/model.py:446-459, batch rule SURVEY 8e) fused with its gradient: writes d(logits), d(aux)
        in place and returns the loss as a 1-element device tensor.  One launch for final + aux."""
        assert labels.dtype == torch.int64 and labels.is_contiguous() and tuple(labels.shape) == (self.B, self.H, self.W)
        s = self._stream()
        HW = self.H * self.W
        lg, ax = self.logits, self.aux
        if lg.Cs <= 16 and self.B * HW < (1 << 31) and (ax is None or (ax.C, ax.Cs) == (lg.C, lg.Cs)) \
                and os.environ.get("MSAU_CE_MULTI", "1") != "0":
            K = self.counts_k
            if K > 1:
                L.call("msau_label_counts_split", s, labels.data_ptr(), self.counts.data_ptr(), self.B, HW, K, key="msau_label_counts")
            else:
                L.call("msau_label_counts", s, labels.data_ptr(), self.counts.data_ptr(), self.B, HW)
            L.call("msau_masked_ce_multi", s, self.dtype, lg.data.data_ptr(), _ptr(ax.data) if ax is not None else None,
                   labels.data_ptr(), self.counts.data_ptr(), lg.grad.data_ptr(), _ptr(ax.grad) if ax is not None else None,
                   self.loss_buf.data_ptr(), self.ce_multi_ws.data_ptr(), self.B, HW, lg.C, lg.Cs, 1.0 / self.B, K)
            return self.loss_buf
        L.call("msau_label_counts", s, labels.data_ptr(), self.counts.data_ptr(), self.B, HW)
        self.loss_buf.zero_()
        for act in (self.logits, self.aux):
            if act is None:
                continue
            L.call("msau_masked_ce", s, self.dtype, act.data.data_ptr(), labels.data_ptr(), self.counts.data_ptr(),
                   act.grad.data_ptr(), self.loss_buf.data_ptr(), self.ce_ws.data_ptr(), self.B, HW, act.C, act.Cs,
                   1.0 / self.B)
        return self.loss_buf

    def set_native_dp(self, comm: int, comm_stream, buckets, flat_grads: torch.Tensor):
        """Data parallelism through the C ABI (msau_allreduce_bucket, csrc/comm.hip): rebuild the backward sequence with one
        MSAU_OP_ALLREDUCE record behind each stage's slab reduction (bucket order = backward order: last stage first, the
        end-conv tail last -- msau_amd/dp.py::stage_buckets) -- the whole sweep INCLUDING the gradient exchange is then one
        call of msau_run_ops_dp.  `comm`: handle from msau_comm_init; `buckets`: [(lo, hi)] element ranges of `flat_grads`."""
        assert self.training and self._bwd_seq is not None and self.overlap_wgrad
        nb = len(self._bwd_segs)
        assert len(buckets) == nb + 1, (len(buckets), nb)
        arr, n, keep = self._bwd_seq
        recs = [(arr[i].kind, arr[i].args) for i in range(n)]
        self._dp_args = []
        out = []
        for i, (b, start, cnt) in enumerate(self._bwd_segs):
            out += recs[start:start + cnt]
            lo, hi = buckets[i + 1]                      # buckets[0] is the end-conv tail, final only after the last stage
            self._dp_args.append(L.AllreduceArgs(comm, flat_grads.data_ptr() + 4 * lo, hi - lo))
            out.append((L.OP_ALLREDUCE | L.OP_COMM, C.addressof(self._dp_args[-1])))
        lo, hi = buckets[0]
        self._dp_args.append(L.AllreduceArgs(comm, flat_grads.data_ptr() + 4 * lo, hi - lo))
        out.append((L.OP_ALLREDUCE | L.OP_COMM, C.addressof(self._dp_args[-1])))
        seq = (L.Op * len(out))()
        for i, (kind, args) in enumerate(out):
            seq[i].kind, seq[i].dtype, seq[i].args = kind, self.dtype, args
        self._bwd_seq_dp = (seq, len(out), (keep, self._dp_args))
        self._dp_stream = comm_stream
        self._dp_flat = flat_grads.data_ptr()

    def backward(self, flat_grads: torch.Tensor, on_stage_done=None, native_dp: bool = False, single_stream: bool = False):
        """Run the backward sweep (external gradients must already be in place) and write the flat fp32
        parameter gradient.  `on_stage_done(stage, side_stream)` is called after each stage's launches are
        enqueued (its slab reduction is the last thing on `side_stream`): the data-parallel engine starts that
        stage's all-reduce bucket there."""
        s = self._stream()
        if L._profiler is not None:                      # per-launch timing path (bench roofline pass)
            for ba in self._box_bwd_args:
                ba.flat_grads = flat_grads.data_ptr()
            for op in reversed(self.ops):
                op.bwd(s)
            if self.unpack_table is not None:
                L.call("msau_wgrad_reduce", s, self.slab_arena.data_ptr(), flat_grads.data_ptr(), self.unpack_table.data_ptr(),
                       len(self._unpack_entries), self._unpack_max)
            return
        for ra in self._reduce_args:
            ra.flat_grads = flat_grads.data_ptr()
        for ba in self._box_bwd_args:
            ba.flat_grads = flat_grads.data_ptr()
        arr, n, _ = self._bwd_seq
        single_stream = single_stream or (str(self.device).startswith("cuda") and torch.cuda.is_current_stream_capturing())
        if not self.overlap_wgrad or single_stream:         # (single_stream: a sweep that is being captured into a HIP graph)
            L.call("msau_run_ops", s, arr, n)
            if on_stage_done is not None:
                for b, _, _ in self._bwd_segs:
                    on_stage_done(b, torch.cuda.current_stream())
            return
        if self._side is None:
            self._side = L.concurrent_stream(self.device)      # a stream that demonstrably overlaps with this one
        side = self._side.cuda_stream
        if native_dp:
            seq, m, _ = self._bwd_seq_dp
            assert flat_grads.data_ptr() == self._dp_flat, "set_native_dp was given another gradient buffer"
            L.call("msau_run_ops_dp", s, side, self._dp_stream.cuda_stream, seq, m, 1)
            return
        if on_stage_done is None:
            L.call("msau_run_ops_overlap", s, side, arr, n, 1)
            return
        for i, (b, start, cnt) in enumerate(self._bwd_segs):
            last = i == len(self._bwd_segs) - 1
            seg = C.cast(C.byref(arr, start * C.sizeof(L.Op)), C.POINTER(L.Op))
            L.call("msau_run_ops_overlap", s, side, seg, cnt, 1 if last else 0)
            on_stage_done(b, self._side)

    def input_grad_nchw(self) -> torch.Tensor:
        """d(loss)/d(input) as NCHW fp32 (only with cfg['input_grad'], used by the op-level tests)."""
        a = self.x_in
        out = torch.empty((self.B, a.C, a.H, a.W), dtype=torch.float32, device=self.device)
        L.call("msau_nhwc_to_nchw", self._stream(), self.dtype, a.grad.data_ptr(), out.data_ptr(), self.B, a.C, a.Cs, a.H, a.W)
        return out

    def activation_bytes(self) -> int:
        n = 0
        for a in self.acts:
            n += a.data.numel() * a.data.element_size()
            if a.grad is not None:
                n += a.grad.numel() * a.grad.element_size()
        return n
