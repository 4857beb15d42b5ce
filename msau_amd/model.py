"""Drop-in counterpart of the reference's model API (model/model.py:399-459 `MSAUWrapper`).

Same constructor, `forward -> (pred, logits, aux_logits)`, `loss`, `save`, `load_weights`, the same
`state_dict` keys / shapes (fp32, OIHW / IOHW), but the network itself runs as a static plan of
hand-written HIP kernels (msau_amd/plan.py, msau_amd/csrc/*).  There is no PyTorch-op fallback: on a
machine without the HIP library or without a GPU the compute entry points raise.

Two ways to train:
  * reference style -- `model(x)`, `model.loss(...)`, `loss.backward()`, `clip_grad_norm_`, `Adam.step()`
    (train_chargrid_funsd_msau.py:46-59): the whole net is ONE autograd node;
  * `TrainEngine.step(x, labels)` -- the same kernels plus fused masked-CE, RCCL gradient
    all-reduce and a fused clip+Adam over the flat parameter buffer, optionally replayed as a HIP graph.
"""
from __future__ import annotations

import os
import weakref

import math

import numpy as np
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from .plan import Plan

DTYPES = {"fp32": L.F32, "float32": L.F32, "bf16": L.BF16, "bfloat16": L.BF16}


def param_shapes(cfg: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict key -> shape in the reference's registration order
    (model/model.py:98-127 encoder, :197-222 decoder, :356-376 stages and end convs)."""
    S, R, Fr = cfg["scale_space_num"], cfg["res_depth"], cfg["featRoot"]
    k, nb = cfg["filter_size"], cfg.get("num_blocks", 3)
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(prefix, co, ci, kk):
        out[prefix + ".weight"] = (co, ci, kk, kk)
        out[prefix + ".bias"] = (co,)

    if cfg.get("variant") == "box":
        # model/model_box.py: the residual 3x3 blocks are MultiBoxConvBlocks (:9-59): num_box_convs x [BoxConv2d with four
        # [c, F] parameters x_min / x_max / y_min / y_max, then a 1x1 conv F*c -> c]; everything else as the plain net
        n, Fn = cfg["num_box_convs"], cfg["num_box_per_channels"]

        def block(prefix, c):
            for i in range(n):
                for nm in ("x_min", "x_max", "y_min", "y_max"):
                    out[f"{prefix}.conv_list.{2 * i}.{nm}"] = (c, Fn)
                conv(f"{prefix}.conv_list.{2 * i + 1}.custom_conv", c, Fn * c, 1)

        for b in range(nb):
            cin = cfg["channels"] if b == 0 else cfg["n_class"]
            pd = f"msau_net.blocks.{b}.downsamplingblock"
            pu = f"msau_net.blocks.{b}.upsamplingblock"
            for l in range(S):
                block(f"{pd}.conv_box_list.{l}", Fr * 2 ** l)
            last = cin
            for l in range(S):
                conv(f"{pd}.conv1s.{l}.conv", Fr * 2 ** l, last, k)
                last = Fr * 2 ** l
            if b > 0:
                for l in range(S):
                    conv(f"{pd}.conv1_1s.{l}.custom_conv", Fr * 2 ** l, 2 * Fr * 2 ** l, 1)
            Cb = Fr * 2 ** (S - 1)
            for nm, co in (("f", Cb // 8), ("g", Cb // 8), ("h", Cb)):
                conv(f"{pd}.layer_attentions.attention_block.{nm}.conv", co, Cb, 1)
            for l in range(S - 1):
                conv(f"{pu}.conv1s.{l}.custom_conv", Fr * 2 ** l, 2 * Fr * 2 ** l, k)
            if b > 0:
                for l in range(S - 1):
                    conv(f"{pu}.conv1_1s.{l}.custom_conv", Fr * 2 ** l, 2 * Fr * 2 ** l, 1)
            for l in range(S - 1):
                c = Fr * 2 ** l
                out[f"{pu}.deconvs.{l}.conv.weight"] = (2 * c, c, k, k)
                out[f"{pu}.deconvs.{l}.conv.bias"] = (c,)
            for l in range(S - 1):
                block(f"{pu}.conv_box_list.{l}", Fr * 2 ** l)
        for b in range(nb):
            conv(f"msau_net.end_convs.{b}.custom_conv", cfg["n_class"], Fr, 4)
        return out

    for b in range(nb):
        cin = cfg["channels"] if b == 0 else cfg["n_class"]
        pd = f"msau_net.blocks.{b}.downsamplingblock"
        pu = f"msau_net.blocks.{b}.upsamplingblock"
        for l in range(S):
            for r in range(R):
                conv(f"{pd}.conv_res_list.{l}.conv_res_list.{r}.custom_conv", Fr * 2 ** l, Fr * 2 ** l, k)
        last = cin
        for l in range(S):
            conv(f"{pd}.conv1s.{l}.conv", Fr * 2 ** l, last, k)
            last = Fr * 2 ** l
        if b > 0:
            for l in range(S):
                conv(f"{pd}.conv1_1s.{l}.custom_conv", Fr * 2 ** l, 2 * Fr * 2 ** l, 1)
        Cb = Fr * 2 ** (S - 1)
        for nm, co in (("f", Cb // 8), ("g", Cb // 8), ("h", Cb)):
            conv(f"{pd}.layer_attentions.attention_block.{nm}.conv", co, Cb, 1)
        for l in range(S - 1):
            for r in range(R):
                conv(f"{pu}.conv_res_list.{l}.conv_res_list.{r}.custom_conv", Fr * 2 ** l, Fr * 2 ** l, k)
        for l in range(S - 1):
            conv(f"{pu}.conv1s.{l}.custom_conv", Fr * 2 ** l, 2 * Fr * 2 ** l, k)
        if b > 0:
            for l in range(S - 1):
                conv(f"{pu}.conv1_1s.{l}.custom_conv", Fr * 2 ** l, 2 * Fr * 2 ** l, 1)
        for l in range(S - 1):
            c = Fr * 2 ** l
            out[f"{pu}.deconvs.{l}.conv.weight"] = (2 * c, c, k, k)        # ConvTranspose2d: [in, out, k, k]
            out[f"{pu}.deconvs.{l}.conv.bias"] = (c,)
    for b in range(nb):
        conv(f"msau_net.end_convs.{b}.custom_conv", cfg["n_class"], Fr, 4)
    return out


def _init_param(key: str, shape, gen: Optional[torch.Generator], fan_in: int = 0) -> torch.Tensor:
    """Reference initialisation statistics: W ~ N(0, sqrt(2/(kh*kw*K2+K3))), b ~ N(0.1, 1e-5)
    (layers.py:33-36,59-60,111-114,130-131,216,227-228); attention 1x1 convs keep torch's Conv2d
    default (attention.py:19-21)."""
    if ".attention_block." in key:
        # kaiming_uniform(a=sqrt(5)) weights and U(-1/sqrt(fan_in), +) biases share the same bound
        bound = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=gen) * 2 - 1) * bound
    if len(shape) == 1:
        return 0.1 + 1e-5 * torch.randn(shape, generator=gen)
    std = math.sqrt(2.0 / (shape[2] * shape[3] * shape[1] + shape[0]))
    return std * torch.randn(shape, generator=gen)


class _Node(nn.Module):
    """Name-only container used to reproduce the reference's module tree (and thereby its state_dict keys)."""


class _MSAUFunction(torch.autograd.Function):
    """The whole network as one autograd node: forward / backward are the plan's kernel sequences."""

    @staticmethod
    def forward(ctx, wrapper, x, *params):
        plan = wrapper._plan_for(x, training=True)
        # The saved activations are the plan's own buffers: a second grad-mode forward of the same shape overwrites them.
        # backward() checks that it still belongs to the latest forward instead of returning silently wrong gradients.
        plan.generation += 1
        logits, aux = plan.forward(wrapper._flat, x)
        ctx.wrapper, ctx.plan, ctx.generation = wrapper, plan, plan.generation
        outs = (logits.clone(), aux.clone() if aux is not None else None)
        ctx.has_aux = aux is not None
        return outs if aux is not None else (outs[0],)

    @staticmethod
    def backward(ctx, *gouts):
        w, plan = ctx.wrapper, ctx.plan
        if plan.generation != ctx.generation:
            raise RuntimeError("MSAUWrapper: backward() of a forward whose saved activations were overwritten by a later "
                               f"forward of the same input shape {(plan.B, plan.H, plan.W)} (gradient accumulation over "
                               "micro-batches: call backward() before the next forward, or use TrainEngine)")
        g_logits = gouts[0]
        g_aux = gouts[1] if ctx.has_aux else None
        plan.set_external_grads(g_logits, g_aux)
        flat_g = torch.zeros_like(w._flat)               # fresh buffer: no aliasing between backward calls
        plan.backward(flat_g)
        grads = []
        for key, p in w._named:
            off, n = w._poff[key], p.numel()
            grads.append(flat_g[off:off + n].view(p.shape) if key not in w._dead else None)
        return (None, None, *grads)


class _MaskedCEFunction(torch.autograd.Function):
    """MSAUWrapper.loss (model/model.py:446-459) on NCHW fp32 logits, batch rule of SURVEY 8(e)."""

    @staticmethod
    def forward(ctx, logits, aux, label):
        B, C, H, W = logits.shape
        dev = logits.device
        Cs = -(-C // 8) * 8
        s = torch.cuda.current_stream().cuda_stream
        label = label.reshape(B, H, W).contiguous().long()
        counts = torch.zeros((B,), dtype=torch.int32, device=dev)
        L.call("msau_label_counts", s, label.data_ptr(), counts.data_ptr(), B, H * W)
        loss = torch.zeros((1,), dtype=torch.float32, device=dev)
        ws = torch.zeros((int(L.load().msau_ce_ws_floats(B * H * W)),), dtype=torch.float32, device=dev)
        grads = []
        for t in (logits, aux):
            if t is None:
                grads.append(None)
                continue
            t = t.contiguous().float()
            nhwc = torch.empty((B, H, W, Cs), dtype=torch.float32, device=dev)
            dn = torch.empty_like(nhwc)
            L.call("msau_nchw_to_nhwc", s, L.F32, t.data_ptr(), nhwc.data_ptr(), B, C, Cs, H, W)
            L.call("msau_masked_ce", s, L.F32, nhwc.data_ptr(), label.data_ptr(), counts.data_ptr(), dn.data_ptr(),
                   loss.data_ptr(), ws.data_ptr(), B, H * W, C, Cs, 1.0 / B)
            g = torch.empty((B, C, H, W), dtype=torch.float32, device=dev)
            L.call("msau_nhwc_to_nchw", s, L.F32, dn.data_ptr(), g.data_ptr(), B, C, Cs, H, W)
            grads.append(g)
        ctx.save_for_backward(*[g for g in grads if g is not None])
        ctx.has_aux = aux is not None
        return loss.reshape(())

    @staticmethod
    def backward(ctx, go):
        saved = ctx.saved_tensors
        g_logits = saved[0] * go
        g_aux = saved[1] * go if ctx.has_aux else None
        return g_logits, g_aux, None


class MSAUWrapper(nn.Module):
    """API-compatible with model/model.py:399-459.  Extra model_kwargs: `num_blocks` (stages,
    reference hard-codes 3: model.py:355) and `dtype` ("fp32" | "bf16" activation/weight storage)."""

    def __init__(self, channels=1, n_class=2, model_kwargs={}):
        super().__init__()
        self.n_class = n_class
        self.channels = channels
        kw = dict(model_kwargs)
        self.scale_space_num = kw.get("scale_space_num", 6)
        self.res_depth = kw.get("res_depth", 3)
        self.featRoot = kw.get("featRoot", 8)
        self.filter_size = kw.get("filter_size", 3)
        self.pool_size = kw.get("pool_size", 2)
        self.activation_name = kw.get("activation_name", "relu")
        if self.activation_name not in ("relu", "elu"):
            # (the reference's constructor leaves `self.activation` unset for any other name and fails with AttributeError two lines on)
            raise ValueError("activation_name must be 'relu' or 'elu' (model/model.py:412-416)")
        self.model = kw.get("model", "msau")
        self.num_scales = kw.get("num_scales", 3)
        self.final_act = kw.get("final_act", "softmax")
        if self.final_act not in ("softmax", "identity"):
            # the reference's default "sigmoid" raises TypeError in its own constructor (model.py:429)
            raise ValueError("final_act must be 'softmax' or 'identity'")
        self.num_blocks = kw.get("num_blocks", 3)
        self.dtype_name = kw.get("dtype", "fp32")
        self._dtype = DTYPES[self.dtype_name]
        if self.pool_size != 2 or self.filter_size % 2 != 1:
            raise NotImplementedError("pool_size must be 2 and filter_size odd")
        widest = self.featRoot * 2 ** (self.scale_space_num - 1)
        if widest > 256:
            # 256 = the reference's constructor defaults (6 scales from 8 channels, dilation 32: model/model.py:406-408); the
            # attention kernels are instantiated up to (32, 256) and the LRN fast path up to 256 channels
            raise NotImplementedError(f"featRoot * 2^(scale_space_num-1) = {widest} channels: the HIP kernels support up to 256")
        self.cfg = dict(channels=channels, n_class=n_class, scale_space_num=self.scale_space_num,
                        res_depth=self.res_depth, featRoot=self.featRoot, filter_size=self.filter_size,
                        pool_size=self.pool_size, num_blocks=self.num_blocks, activation=self.activation_name)
        self.cfg.update(self._variant_cfg(kw))
        for opt in ("reuse_activations", "overlap_wgrad", "overlap_max_pix", "deterministic"):      # execution options of the plan
            if opt in kw:
                self.cfg[opt] = kw[opt]

        shapes = param_shapes(self.cfg)
        self._poff: Dict[str, int] = {}
        self._pshape: Dict[str, Tuple[int, ...]] = dict(shapes)
        off = 0
        for k, shp in shapes.items():
            self._poff[k] = off
            off += _ru4(int(math.prod(shp)))
        self._total = off
        self._flat = torch.zeros(self._total, dtype=torch.float32)
        gen = torch.Generator().manual_seed(int(kw.get("seed", torch.initial_seed() % (2 ** 31))))
        self._named = []
        self.msau_net = _Node()
        boxes: Dict[tuple, tuple] = {}
        for k, shp in shapes.items():
            n = int(math.prod(shp))
            leaf = k.rsplit(".", 1)[1]
            if leaf in ("x_min", "x_max", "y_min", "y_max"):
                # BoxConv2d boxes in units of the max box size: a random centre in the middle half, a random half extent of
                # 1/28 .. 1/4 (the third-party package's own initialiser is unavailable: unpinned, oracle/box_oracle.py)
                axis = (k.rsplit(".", 1)[0], leaf[0])
                if axis not in boxes:
                    centre = (torch.rand(shp, generator=gen) - 0.5) * 0.5
                    half = 1.0 / 28 + torch.rand(shp, generator=gen) * (0.25 - 1.0 / 28)
                    boxes[axis] = (centre - half, centre + half)
                val = boxes[axis][0 if leaf.endswith("min") else 1]
            else:
                wshp = shapes[k[:-4] + "weight"] if k.endswith(".bias") else shp
                val = _init_param(k, shp, gen, wshp[1] * wshp[2] * wshp[3])
            self._flat[self._poff[k]:self._poff[k] + n] = val.reshape(-1)
            p = nn.Parameter(self._flat[self._poff[k]:self._poff[k] + n].view(shp))
            self._named.append((k, p))
            node = self
            parts = k.split(".")
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            node.register_parameter(parts[-1], p)
        # parameters that never receive a gradient: the last stage's attention (model.py:149-150)
        self._dead = {k for k in shapes
                      if f"blocks.{self.num_blocks - 1}.downsamplingblock.layer_attentions" in k}
        self.predictor = nn.Softmax(dim=1) if self.final_act == "softmax" else nn.Sequential()
        self.criterion = nn.CrossEntropyLoss()
        self._plans: "OrderedDict[tuple, Plan]" = OrderedDict()
        self.max_cached_plans = 4               # LRU bounds: number of plans and bytes of their activation buffers
        self.max_plan_bytes = 64 << 30

    def _variant_cfg(self, kw: dict) -> dict:
        """extra plan / parameter configuration of a network variant (BMSAUWrapper: the box-convolution blocks)"""
        return {}

    # ---- flat parameter storage ---------------------------------------------------------------
    def _rebind(self):
        for k, p in self._named:
            n = p.numel()
            p.data = self._flat[self._poff[k]:self._poff[k] + n].view(self._pshape[k])
            p.grad = None
        self._plans.clear()

    def _apply(self, fn, recurse=True):
        new_flat = fn(self._flat)
        if new_flat.dtype != torch.float32:
            raise TypeError("MSAUWrapper keeps fp32 master parameters; choose bf16 storage with model_kwargs['dtype']")
        self._flat = new_flat
        self._rebind()
        return self

    @property
    def flat_parameters(self) -> torch.Tensor:
        return self._flat

    # ---- plans ----------------------------------------------------------------------------------
    def _plan_for(self, x: torch.Tensor, training: bool) -> Plan:
        if not x.is_cuda:
            raise RuntimeError("MSAUWrapper runs on an MI355X through libmsau_hip.so; input must be a CUDA/HIP tensor "
                               "(there is no CPU fallback)")
        B, C, H, W = x.shape
        if C != self.channels:
            raise ValueError(f"expected {self.channels} input channels, got {C}")
        return self._plan_for_shape(B, H, W, x.device, training)

    def _plan_for_shape(self, B: int, H: int, W: int, device, training: bool) -> Plan:
        if self._flat.device != device:
            raise RuntimeError(f"model is on {self._flat.device}, input on {device}")
        key = (B, H, W, training)
        plan = self._plans.get(key)
        if plan is None:
            plan = Plan(self.cfg, B, H, W, self._dtype, device, self._poff, self._pshape, training=training)
            plan.generation = 0                 # bumped by every grad-mode forward (see _MSAUFunction)
            self._plans[key] = plan
            while len(self._plans) > 1 and (len(self._plans) > self.max_cached_plans or
                                            sum(p.activation_bytes() for p in self._plans.values()) > self.max_plan_bytes):
                self._plans.popitem(last=False)     # captured graphs live on the plan object and go with it
        else:
            self._plans.move_to_end(key)
        return plan

    # ---- reference API ----------------------------------------------------------------------------
    def forward(self, inp):
        x = inp.contiguous().float()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for _, p in self._named)
        if need_grad:
            outs = _MSAUFunction.apply(self, x, *[p for _, p in self._named])
            logits = outs[0]
            aux = outs[1] if len(outs) > 1 else None
        else:
            plan = self._plan_for(x, training=False)
            lg, ax = plan.forward(self._flat, x)
            logits, aux = lg.clone(), (ax.clone() if ax is not None else None)
        if self.final_act == "softmax":
            with torch.no_grad():
                B, C, H, W = logits.shape
                pred = torch.empty_like(logits)
                L.call("msau_softmax_channels_nchw", torch.cuda.current_stream().cuda_stream, logits.data_ptr(),
                       pred.data_ptr(), B, C, H * W)
        else:
            pred = logits
        return pred, logits, aux

    @torch.no_grad()
    def predict_nhwc(self, inp: Optional[torch.Tensor] = None, ids: Optional[torch.Tensor] = None, graph: bool = False):
        """Forward-only path for `KVModel.predict` (inference/kv_model.py:305-313): one of
          inp float [B,C,H,W]   -- the dense grid the reference builds with to_categorical, or
          ids int   [B,H,W]     -- the character-id mask itself; the one-hot grid is painted on the device
        -> (pred fp32 [B,H,W,n_class] = softmax over classes, already in the NHWC order `_extract_value` wants,
            argmax uint8 [B,H,W] = np.argmax(pred, -1)).
        No activations are kept (buffers are reused by liveness) and softmax + argmax run in the last conv's epilogue.
        The returned tensors are the plan's buffers: copy them before the next call if they must survive it.
        graph=True replays the sweep as a HIP graph (captured per shape on first use, on a dedicated stream): at
        batch 1 the ~120 launches are host-bound and the replay is what sets the latency."""
        if (inp is None) == (ids is None):
            raise ValueError("give exactly one of inp / ids")
        if self.final_act != "softmax":
            raise ValueError("predict_nhwc is the softmax head; final_act is %r" % (self.final_act,))
        if self.n_class > 255:
            raise ValueError("the argmax map is uint8: at most 255 classes")
        if ids is not None:
            if ids.dim() != 3:
                raise ValueError("ids must be [B,H,W]")
            ids = ids.to(device=self._flat.device, dtype=torch.int32).contiguous()
            B, H, W = ids.shape
            ref = ids
        else:
            inp = inp.contiguous().float()
            B, C, H, W = inp.shape
            ref = inp
        if not ref.is_cuda:
            raise RuntimeError("MSAUWrapper runs on an MI355X through libmsau_hip.so; input must be a CUDA/HIP tensor "
                               "(there is no CPU fallback)")
        plan = self._plan_for(inp, False) if ids is None else self._plan_for_shape(B, H, W, ref.device, False)
        if not graph:
            return plan.predict(self._flat, x_nchw=inp, ids=ids)
        kind = "ids" if ids is not None else "dense"
        if getattr(self, "_pstream", None) is None:
            self._pstream = torch.cuda.Stream(device=ref.device)      # never replay into the NULL stream (see TrainEngine)
        cur, gs = torch.cuda.current_stream(), self._pstream
        gs.wait_stream(cur)
        with torch.cuda.stream(gs):
            cache = plan.__dict__.setdefault("_pgraphs", {})
            if kind not in cache:
                static = ref.clone()
                kw = dict(ids=static) if ids is not None else dict(x_nchw=static)
                plan.predict(self._flat, **kw)                        # warm-up outside capture
                torch.cuda.synchronize()
                g = _keep_graph(torch.cuda.CUDAGraph())
                with _capture_section():
                    with torch.cuda.graph(g, stream=gs):
                        plan.predict(self._flat, **kw)
                cache[kind] = (g, static)
            g, static = cache[kind]
            static.copy_(ref, non_blocking=True)
            g.replay()
        cur.wait_stream(gs)
        return plan.head_probs, plan.head_argmax

    def save(self, path):
        torch.save(self.state_dict(), path)

    def load_weights(self, path):
        self.load_state_dict(torch.load(path, map_location=self._flat.device))

    def loss(self, out_grid, out_grid_aux, label_mask):
        """Masked CE of final + aux logits over pixels with label != 0 (model/model.py:446-459).
        Accepts label_mask [B,H,W] (the reference: B = 1): per-sample masked mean, then mean over B."""
        return _MaskedCEFunction.apply(out_grid, out_grid_aux, label_mask)


# A stream capture is in "global" mode: a device synchronisation (or a graph's destruction) from ANY Python context while it runs --
# e.g. the finaliser of another engine, whenever the garbage collector gets to it -- makes hipStreamEndCapture abort the process
# (seen once in four full GPU test runs, 2026-10-04).  Graphs whose engine dies during a capture wait in _graveyard until the next
# safe point (TrainEngine.step / _drop_graphs outside a capture).
_capturing = 0
_graveyard: list = []
# FINDING (MI355X / ROCm 7.2 / torch 2.10; rounds 3-4).  Destroying a captured graph (torch.cuda.CUDAGraph's destructor ->
# hipGraphExecDestroy / hipGraphDestroy) and capturing another one crashed the process -- abort inside the destruction or inside
# hipStreamEndCapture, a segmentation fault, once a hang -- in half of the stand-alone runs of tests/test_train_gpu.py.  Round 4 narrowed
# it down (tools/repro/, profiles/r04_graph_destroy.md): a stand-alone HIP program and a torch-only script that capture TWO-stream
# sweeps (fork / join by events from a reused pool, eager sweeps in between), destroy and re-capture pass 240 / 120 cycles; this code
# base with graph destruction crashes 4 of 8 runs when the captured backward forks its weight gradients onto the side stream and
# 0 of 8 when the same sweep is captured on ONE stream.  So a captured sweep is single-stream (Plan.backward(single_stream=True): graph
# mode is the slower, opt-in mode anyway, section 2 of DESIGN.md), graphs are ordinary objects again, and nothing is leaked: the
# round-3 remedy (one immortal reference per captured graph) is gone.
def _keep_graph(g):
    return g


class _capture_section:
    def __enter__(self):
        global _capturing
        _capturing += 1

    def __exit__(self, *exc):
        global _capturing
        _capturing -= 1
        return False


def _bury_graphs():
    """destroy the graphs of dead engines: not while a replay may still be running (ROCm 7.2: the process segfaults), not during a capture"""
    if _graveyard and not _capturing:
        torch.cuda.synchronize()
        _graveyard.clear()          # (the graphs, their private pools and the static input copies go)


def _ru4(n: int) -> int:
    return -(-n // 4) * 4          # keep every parameter 16-byte aligned inside the flat buffer


class TrainEngine:
    """Fused training step on one GPU (one process per GPU under data parallelism).

    forward -> masked CE (+grad) -> backward -> [RCCL all-reduce of the flat gradient] ->
    global-norm clip + Adam, all on the flat fp32 parameter buffer of `model`
    (train_chargrid_funsd_msau.py:46-59 with lr 1e-4, clip 1.0)."""

    def __init__(self, model: MSAUWrapper, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_norm: float = 1.0, process_group=None, use_graph: bool = False):
        self.model, self.lr, self.betas, self.eps, self.max_norm = model, lr, betas, eps, max_norm
        flat = model._flat
        if not flat.is_cuda:
            raise RuntimeError("TrainEngine needs the model on a GPU (model.cuda())")
        self.flat_grad = torch.zeros_like(flat)
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.state = torch.zeros(8, dtype=torch.float32, device=flat.device)
        self.adam_ws = torch.zeros(int(L.load().msau_adam_ws_floats(flat.numel())), dtype=torch.float32, device=flat.device)
        from .dp import GradSync, stage_buckets
        self.pg = process_group
        self.sync = GradSync(self.flat_grad, stage_buckets(model._poff, model._total, model.num_blocks), process_group)
        self.world = self.sync.world
        self.use_graph = use_graph
        self._comm = None
        self._init_native_comm(process_group)
        # Graph replays run on a dedicated non-default stream.  Replaying on the legacy NULL stream after the
        # host had synchronised produced corrupted steps on ROCm 7.2 / gfx950 (nodes of consecutive launches
        # overlapping; found 2026-10-03 with tools/loss_trace.py) -- never launch these graphs into stream 0.
        self._gstream = torch.cuda.Stream(device=flat.device)
        # Captured graphs are filed on the plan under this token.  It is never re-used (id(self) is, once an engine is
        # freed: a second engine for the same model -- an lr sweep, a resume -- could then replay the dead engine's graphs,
        # whose kernel arguments point at ITS freed moments and carry ITS hyper-parameters), and it changes whenever a
        # by-value kernel argument of the captured optimiser step changes (_invalidate_graphs).
        TrainEngine._tokens += 1
        self._token = TrainEngine._tokens
        weakref.finalize(self, TrainEngine._drop_graphs, weakref.ref(model), self._token)

    _tokens = 0

    @staticmethod
    def _drop_graphs(model_ref, token):
        model = model_ref()
        if model is not None:
            doomed = [plan._tgraphs.pop(token) for plan in model._plans.values() if token in plan.__dict__.get("_tgraphs", {})]
            if doomed:
                _graveyard.extend(doomed)            # (this may be a finaliser running in the middle of somebody's stream capture)
                del doomed
                _bury_graphs()

    def _invalidate_graphs(self):
        """lr / betas / eps / max_norm are by-value arguments of msau_clip_adam_step, frozen into a captured optimiser graph:
        after changing them (load_state_dict, set_lr) the graphs of this engine are dropped and re-captured on the next step"""
        TrainEngine._drop_graphs(weakref.ref(self.model), self._token)
        TrainEngine._tokens += 1
        self._token = TrainEngine._tokens

    def set_hyper(self, lr=None, betas=None, eps=None, max_norm=None):
        """change optimiser hyper-parameters between steps (the lr schedule of model/training/trainer.py:124)"""
        if lr is not None: self.lr = float(lr)
        if betas is not None: self.betas = tuple(betas)
        if eps is not None: self.eps = float(eps)
        if max_norm is not None: self.max_norm = float(max_norm)
        self._invalidate_graphs()

    def _init_native_comm(self, group):
        """The gradient exchange through the C ABI (msau_allreduce_bucket over RCCL, csrc/comm.hip) when the process group is an
        RCCL one: rank 0 draws the communicator id, the group broadcasts it, every rank joins.  Then the all-reduce of each
        stage's bucket is a record of the native backward sequence (Plan.set_native_dp) -- no torch.distributed call, no
        Python between the launches.  gloo groups (CPU tests, several ranks on one card) and MSAU_DP_NATIVE=0 keep
        msau_amd/dp.py::GradSync."""
        import torch.distributed as dist
        # Opt-in (MSAU_DP_NATIVE=1) since round 4: the path has only ever run at world size 1 on the one-GPU boxes this tree is
        # measured on (tests/test_dp_gpu.py: bit-equal to the torch.distributed path there, where an all-reduce is the identity);
        # until a run with >= 2 RCCL ranks has shown the same, a multi-GPU job takes the GradSync path that the 2-rank tests cover.
        if not self.sync.active or self.use_graph or os.environ.get("MSAU_DP_NATIVE", "0") != "1":
            return
        if not (dist.is_available() and dist.is_initialized()) or dist.get_backend(group) != "nccl":
            return
        if not L.load().msau_comm_available():
            return
        import ctypes as C
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ident = torch.zeros(128, dtype=torch.uint8, device=self.model._flat.device)
        if rank == 0:
            buf = (C.c_ubyte * 128)()
            L.call("msau_comm_unique_id", buf, 128)
            ident.copy_(torch.tensor(list(buf), dtype=torch.uint8))
        dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(ident.cpu().tolist())
        comm = C.c_void_p()
        L.call("msau_comm_init", C.byref(comm), world, rank, raw, 128)
        self._comm = comm.value
        self._comm_stream = L.concurrent_stream(self.model._flat.device, index=1)
        weakref.finalize(self, L.load().msau_comm_destroy, self._comm)

    # -- pieces (each is a fixed launch sequence on the current stream) --
    def _fwd_bwd(self, plan: Plan, x, labels, ids=None, nhwc_ready=False, owner=None):
        plan.forward(self.model._flat, x, export=False, ids=ids, nhwc_ready=nhwc_ready, owner=owner, single_stream=self.use_graph)
        loss = plan.loss_grads(labels)
        # MSAU_DP_BUCKETS=1: ONE all-reduce of the whole flat gradient after the backward instead of a bucket per stage
        # issued while the earlier stages' backward still runs (fewer launches and joins, no overlap)
        self._ar_native = False
        # the native sequence needs one bucket per backward segment + the end-conv tail (stage_buckets drops EMPTY buckets: then
        # the counts differ and the exchange stays with GradSync), and the per-launch profiling path does not run sequences at all
        native = self._comm is not None and self.sync.active and plan.overlap_wgrad and L._profiler is None \
            and len(self.sync.buckets) == len(plan._bwd_segs) + 1
        if native:
            if getattr(plan, "_dp_flat", None) != self.flat_grad.data_ptr() or getattr(plan, "_dp_comm", None) != self._comm:
                plan.set_native_dp(self._comm, self._comm_stream, self.sync.buckets, self.flat_grad)
                plan._dp_comm = self._comm
            plan.backward(self.flat_grad, native_dp=True)      # stage buckets exchanged inside the native sequence, joined at its end
            self._ar_started = False
            self._ar_native = True
        elif self.sync.active and not self.use_graph and os.environ.get("MSAU_DP_BUCKETS", "stage") != "1":
            # bucket i of GradSync = [end convs, last stage, ..., stage 0]; a stage's bucket is reduced over RCCL as
            # soon as that stage's slab reduction is enqueued, while the earlier stages' backward still runs
            nb = self.model.num_blocks
            plan.backward(self.flat_grad, on_stage_done=lambda b, side: self.sync.start(nb - b, after=side))
            self._ar_started = True
        else:
            plan.backward(self.flat_grad, single_stream=self.use_graph)      # (a captured sweep stays on one stream: see _keep_graph)
            self._ar_started = False
        return loss

    def _optim(self):
        b1, b2 = self.betas
        n = self.model._flat.numel()
        L.call("msau_clip_adam_step", torch.cuda.current_stream().cuda_stream, self.model._flat.data_ptr(),
               self.flat_grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.state.data_ptr(),
               self.adam_ws.data_ptr(), n, self.lr, b1, b2, self.eps, self.max_norm, 1.0 / self.world)

    def _allreduce(self):
        if getattr(self, "_ar_native", False):
            return                                       # already in the backward sequence (msau_run_ops_dp)
        if self.sync.active:
            if getattr(self, "_ar_started", False):
                self.sync.start(0)                       # the end-conv tail: final once every stage is done
            elif os.environ.get("MSAU_DP_BUCKETS", "stage") == "1":
                self.sync.start_whole()
            else:
                self.sync.start_all()
            self.sync.finish()

    def step(self, x: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """One optimisation step.  Returns the (local) loss as a 1-element device tensor (no host sync)."""
        x = x.contiguous().float()
        labels = labels.reshape(x.shape[0], x.shape[2], x.shape[3]).contiguous().long()
        plan = self.model._plan_for(x, training=True)
        if not self.use_graph:
            loss = self._fwd_bwd(plan, x, labels)
            self._allreduce()
            self._optim()
            return loss
        # The captured graphs hold the plan's buffer addresses: they are stored ON the plan (as predict_nhwc does), so
        # that an evicted / rebuilt plan can never be replayed through a stale graph.
        _bury_graphs()
        graphs = plan.__dict__.setdefault("_tgraphs", {})
        key = self._token
        cur = torch.cuda.current_stream()
        gs = self._gstream
        gs.wait_stream(cur)
        with torch.cuda.stream(gs):
            if key not in graphs:
                sx, sl = x.clone(), labels.clone()
                # warm up outside capture (hipFuncSetAttribute calls, lazy allocations)
                self._fwd_bwd(plan, sx, sl)
                torch.cuda.synchronize()
                g1 = _keep_graph(torch.cuda.CUDAGraph())
                with _capture_section():
                    with torch.cuda.graph(g1, stream=gs):
                        loss = self._fwd_bwd(plan, sx, sl)
                g2 = _keep_graph(torch.cuda.CUDAGraph())
                with _capture_section():
                    with torch.cuda.graph(g2, stream=gs):
                        self._optim()
                graphs[key] = (g1, g2, loss, sx, sl)
            g1, g2, loss, sx, sl = graphs[key]
            sx.copy_(x, non_blocking=True)
            sl.copy_(labels, non_blocking=True)
            g1.replay()
            self._allreduce()
            g2.replay()
        cur.wait_stream(gs)
        return loss

    def step_ids(self, ids: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """One optimisation step fed with the character-id mask int32 [B,H,W] (-1 or any id outside [0, channels) = empty
        pixel) instead of the dense one-hot float tensor: what `to_categorical` / the chargrid painter would have produced
        is painted straight into the NHWC input on the device -- 4 B per pixel cross the boundary instead
        of 4*C, and the 352 MB NCHW -> NHWC conversion of cfg 2 disappears.  Same kernels after that: bit-identical to
        `step(one_hot(ids), labels)`.  Eager only."""
        if self.use_graph:
            raise RuntimeError("step_ids is an eager path (use_graph=False)")
        ids = ids.to(dtype=torch.int32).contiguous()
        B, H, W = ids.shape
        labels = labels.reshape(B, H, W).contiguous().long()
        plan = self.model._plan_for_shape(B, H, W, ids.device, True)
        loss = self._fwd_bwd(plan, None, labels, ids=ids)
        self._allreduce()
        self._optim()
        return loss

    def input_nhwc(self, B: int, H: int, W: int) -> torch.Tensor:
        """The training plan's own input buffer for this shape, [B][H][W][Cs] in the storage dtype: the zero-copy target of
        a device-side producer (msau_amd.data.raster: `rasterize(..., out=...)`, `rasterize_dense(..., out=...)`)."""
        return self.model._plan_for_shape(B, H, W, self.model._flat.device, True).input_nhwc

    def step_nhwc(self, grid: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """One optimisation step on a chargrid that is ALREADY on the device in the kernels' layout ([B][H][W][Cs], storage
        dtype) -- what the device painters produce (data_generator_funsd_bert.py:64-93,240).  When `grid` is
        the plan's own buffer (`input_nhwc(B, H, W)`) nothing is copied or converted; otherwise one device copy.  `step(x, l)`
        on the same grid as fp32 NCHW spends a quarter of the 768-channel step (cfg 4) converting 4.2 GB: identical result."""
        if self.use_graph:
            raise RuntimeError("step_nhwc is an eager path (use_graph=False)")
        B, H, W, Cs = grid.shape
        plan = self.model._plan_for_shape(B, H, W, grid.device, True)
        buf = plan.input_nhwc
        if tuple(grid.shape) != tuple(buf.shape) or grid.dtype != buf.dtype:
            raise ValueError(f"step_nhwc wants {tuple(buf.shape)} {buf.dtype} (channels padded to a multiple of 8), got {tuple(grid.shape)} {grid.dtype}")
        if grid.data_ptr() != buf.data_ptr():
            buf.copy_(grid)
        labels = labels.reshape(B, H, W).contiguous().long()
        loss = self._fwd_bwd(plan, None, labels, nhwc_ready=True)
        self._allreduce()
        self._optim()
        return loss

    def step_boxes(self, grid_boxes, label_boxes, B: int, H: int, W: int, feats=None) -> torch.Tensor:
        """One optimisation step from BOX LISTS (int32 [n][6] = sample, y0, y1, x0, x1, value; msau_amd/data/raster.py): the
        one-hot grid (feats None: value = character id) or the dense embedding grid (feats fp32 [n_vectors][channels]: value
        = row of feats) and the label mask are painted on the device, the grid straight into the plan's input buffer.
        Only the lists (KBs) and the feature table cross PCIe.  Arguments may be numpy arrays or device tensors."""
        from .data import raster
        plan = self.model._plan_for_shape(B, H, W, self.model._flat.device, True)
        if feats is not None and not self.use_graph and plan._feed_owner(None):
            # the embedding grid is piecewise constant: the first conv and its weight gradient work from the per-pixel box index
            # and the feature table (MSAU_CONV_OWNER, csrc/ownerconv.hip) -- the 1536-bytes-per-pixel tensor is never painted
            dev = self.model._flat.device
            ft = feats if isinstance(feats, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(feats, dtype=np.float32)).to(dev)
            owner, fb, nf, labels = raster.owner_maps(grid_boxes, label_boxes, B, H, W, dev)
            loss = self._fwd_bwd(plan, None, labels, owner=(owner, fb, nf, ft))
            self._allreduce()
            self._optim()
            return loss
        buf = self.input_nhwc(B, H, W)
        if feats is None:
            _, labels = raster.rasterize(grid_boxes, label_boxes, B, H, W, self.model.channels, self.model.dtype_name, buf.device, out=buf)
        else:
            _, labels = raster.rasterize_dense(grid_boxes, label_boxes, feats, B, H, W, self.model.dtype_name, buf.device, out=buf)
        return self.step_nhwc(buf, labels)

    def prefetch_boxes(self, grid_boxes, label_boxes, B: int, H: int, W: int, feats=None):
        """Paint the NEXT batch (arguments as `step_boxes`) while the current step runs: the grid goes into the input buffer the
        current step does not read (the plan keeps two), on the plan's side stream -- idle during the forward sweep, which is
        when the painter's store (2.1 GB per batch at 768 channels) is absorbed.  `step_prefetched()` then trains on it.  The
        data-loader counterpart of the reference's generator thread (data_generator_funsd_bert.py:216-240)."""
        from .data import raster
        if self.use_graph:
            raise RuntimeError("prefetch_boxes is an eager path (use_graph=False)")
        plan = self.model._plan_for_shape(B, H, W, self.model._flat.device, True)
        q = self.__dict__.setdefault("_prefetched", [])
        if len(q) >= 2:
            # the plan keeps TWO input buffers: a third batch would be painted over one that is still queued, unread
            raise RuntimeError("prefetch_boxes: two batches are already queued (the plan has two input buffers); call "
                               "step_prefetched() before painting another one")
        k = (q[-1][1] + 1) % 2 if q else (getattr(self, "_pf_last", 1) + 1) % 2
        buf = plan.input_buffer(k)
        cur = torch.cuda.current_stream()
        if plan._side is None:
            plan._side = L.concurrent_stream(plan.device)
        side = plan._side
        side.wait_stream(cur)                    # the buffer's last readers (two steps back) are behind everything enqueued so far
        with torch.cuda.stream(side):
            if feats is None:
                _, labels = raster.rasterize(grid_boxes, label_boxes, B, H, W, self.model.channels, self.model.dtype_name, buf.device, out=buf)
            else:
                _, labels = raster.rasterize_dense(grid_boxes, label_boxes, feats, B, H, W, self.model.dtype_name, buf.device, out=buf)
            ev = torch.cuda.Event()
            ev.record(side)
        labels.record_stream(cur)
        q.append((plan, k, labels, ev))

    def step_prefetched(self) -> torch.Tensor:
        """One optimisation step on the oldest batch `prefetch_boxes` painted."""
        plan, k, labels, ev = self._prefetched.pop(0)
        self._pf_last = k
        torch.cuda.current_stream().wait_event(ev)
        plan.use_input(k)
        loss = self._fwd_bwd(plan, None, labels, nhwc_ready=True)
        self._allreduce()
        self._optim()
        return loss

    @property
    def grad_norm(self) -> torch.Tensor:
        return self.state[1]

    # -- optimiser state (resume): the counterpart of torch.optim.Adam.state_dict() for the flat buffers --
    def state_dict(self) -> dict:
        """{"engine": 1, "step", "exp_avg", "exp_avg_sq" (flat fp32, the model's parameter order), hyper-parameters}"""
        return {"engine": 1, "step": int(round(float(self.state[0]))), "exp_avg": self.m.detach().clone(),
                "exp_avg_sq": self.v.detach().clone(), "lr": self.lr, "betas": tuple(self.betas), "eps": self.eps,
                "max_norm": self.max_norm, "numel": int(self.m.numel())}

    def load_state_dict(self, sd: dict):
        """Accepts `TrainEngine.state_dict()` or a `torch.optim.Adam.state_dict()` over the model's parameters in
        registration order (what the reference's `save_checkpoint` stores: utils/io_utils.py:83-105); parameters Adam
        never stepped (the dead last-stage attention) have no entry and keep zero moments."""
        if sd.get("engine") == 1:
            if int(sd["numel"]) != self.m.numel():
                raise ValueError(f"optimizer state is for {sd['numel']} parameters, the model has {self.m.numel()}")
            self.m.copy_(sd["exp_avg"])
            self.v.copy_(sd["exp_avg_sq"])
            self.state.zero_()
            self.state[0] = float(sd["step"])
            self.lr, self.betas, self.eps = float(sd["lr"]), tuple(sd["betas"]), float(sd["eps"])
            self.max_norm = float(sd.get("max_norm", self.max_norm))
            self._invalidate_graphs()
            return
        groups, state = sd["param_groups"], sd["state"]
        order = [pid for g in groups for pid in g["params"]]
        named = [(k, p) for k, p in self.model._named if p.requires_grad]
        if len(order) != len(named):
            raise ValueError(f"optimizer state covers {len(order)} parameters, the model has {len(named)}")
        self.m.zero_()
        self.v.zero_()
        step = 0
        for pid, (key, p) in zip(order, named):
            st = state.get(pid)
            if st is None:
                continue
            off, n = self.model._poff[key], p.numel()
            if st["exp_avg"].numel() != n:
                raise ValueError(f"optimizer state of {key}: {st['exp_avg'].numel()} elements, parameter has {n}")
            self.m[off:off + n] = st["exp_avg"].reshape(-1).to(self.m)
            self.v[off:off + n] = st["exp_avg_sq"].reshape(-1).to(self.v)
            step = max(step, int(round(float(st["step"]))))
        self.state.zero_()
        self.state[0] = float(step)
        g0 = groups[0]
        self.lr, self.betas, self.eps = float(g0["lr"]), tuple(g0["betas"]), float(g0["eps"])
        self._invalidate_graphs()
