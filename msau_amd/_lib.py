"""ctypes binding of libmsau_hip.so (C ABI declared in include/msau_hip.h).

There is no CPU fallback: if the library is missing or a symbol is absent this module raises, and
every op that would have used it fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSAU_HIP_LIB", os.path.join(HERE, "libmsau_hip.so"))

F32, BF16 = 0, 1

CONV_RELU_IN, CONV_RELU_OUT, CONV_ADD, CONV_ACCUM, CONV_MASK_A, CONV_MASK_B, CONV_HEAD, CONV_DOUT = 1, 2, 4, 8, 16, 32, 64, 128
CONV_LRN, CONV_POOL, CONV_IDS, CONV_OWNER, CONV_NCHW, CONV_WGRAD, CONV_ELU = 256, 512, 1024, 2048, 4096, 8192, 32768

i32, i64, vp, f32 = C.c_int32, C.c_int64, C.c_void_p, C.c_float


class ConvDesc(C.Structure):
    _fields_ = [(n, i32) for n in ("B", "Hin", "Win", "Hout", "Wout", "C1", "C2", "Cout", "KH", "KW", "dil",
                                   "pad_t", "pad_l", "stride", "ups", "flags")] + \
               [(n, vp) for n in ("x1", "x2", "wpack", "bias", "add", "mask_a", "mask_b", "y", "head_probs", "head_argmax")] + \
               [("head_classes", i32), ("flags2", i32), ("y2", vp), ("mask_b2", vp)] + \
               [("lrn_alpha_over_n", f32), ("lrn_beta", f32), ("lrn_k", f32), ("reserved0", i32), ("pool_y", vp), ("pool_idx", vp)] + \
               [("wg_x1", vp), ("wg_slabs", vp), ("wg_nslabs", i32), ("reserved1", i32)]


class ConvPairDesc(C.Structure):
    _fields_ = [(n, i32) for n in ("B", "H", "W", "C", "flags1", "flags2")] + \
               [(n, vp) for n in ("x", "w1", "b1", "mask_mid", "mid", "w2", "b2", "add", "mask_a", "mask_b", "y", "pool_y", "pool_idx", "bits_mid", "bits_a",
                                  "lrn_a", "lrn_da")] + \
               [("lrn_alpha_over_n", f32), ("lrn_beta", f32), ("lrn_k", f32), ("reserved0", i32), ("wg1_x", vp), ("wg1_slabs", vp), ("wg1_nslabs", i32), ("reserved1", i32)] + \
               [(n, vp) for n in ("cpl_prev", "cpl_w", "cpl_b", "cpl_y", "cpl_pool_y", "cpl_pool_idx", "dcp_dz", "dcp_w", "dcp_mask", "dcp_dprev")]


PAIR_RELU_IN, PAIR_RELU_MID, PAIR_MASK_MID, PAIR_TILES, PAIR_LRN_BWD, PAIR_WGRAD1, PAIR_COUPLE, PAIR_DCOUPLE = 1, 2, 4, 8, 16, 32, 64, 128


class BoxArgs(C.Structure):
    _fields_ = [(n, vp) for n in ("in_", "ii", "params_fwd", "params_refl", "out", "gout", "gin", "ii_g", "ws", "flat_grads", "ws_ii",
                                  "mask_a", "add", "mask_b")] + \
               [(n, i64) for n in ("off_hmin", "off_hmax", "off_wmin", "off_wmax")] + \
               [(n, i32) for n in ("B", "H", "W", "C", "F", "Cs_in", "Cs_out", "relu_in", "accumulate")] + \
               [("max_h", f32), ("max_w", f32)]


class ConvPackGeom(C.Structure):
    _fields_ = [("cch", i32), ("nchunks", i32), ("kchunk", i32), ("rows", i32), ("bytes", i64)]


class WgradDesc(C.Structure):
    _fields_ = [(n, i32) for n in ("B", "Hin", "Win", "Hout", "Wout", "C1", "C2", "Cout", "KH", "KW", "dil",
                                   "pad_t", "pad_l", "stride", "flags")] + \
               [(n, vp) for n in ("x1", "x2", "g", "slabs")] + [("nslabs", i32)]


class WgradGeom(C.Structure):
    _fields_ = [("cch", i32), ("nchunks", i32), ("kext", i32), ("max_slabs", i32), ("slab_bytes", i64),
                ("lean", i32), ("reserved", i32)]


class PackEntry(C.Structure):
    _fields_ = [("src_off", i64), ("dst_off", i64)] + \
               [(n, i32) for n in ("kind", "dim0", "dim1", "KH", "KW", "row_is_dim0", "flip", "row_off", "rows_real",
                                   "rows_pad", "k1_real", "k1_store", "k2_real", "k2_store", "cch", "nchunks",
                                   "kchunk", "dtype")]


class UnpackEntry(C.Structure):
    _fields_ = [("slab_off", i64), ("w_off", i64), ("b_off", i64), ("b_src_off", i64), ("b_slab_stride", i64)] + \
               [(n, i32) for n in ("b_elem_stride", "b_nslabs", "b_count", "nslabs", "slab_elems", "kext", "dim0", "dim1", "KH",
                                   "KW", "row_is_dim0", "rows_real", "k1_real", "k1_store", "k2_real", "k2_store",
                                   "cch", "nchunks", "accumulate")]


class Op(C.Structure):
    _fields_ = [("kind", i32), ("dtype", i32), ("args", vp)]


class LrnArgs(C.Structure):
    _fields_ = [("a", vp), ("dy", vp), ("out", vp), ("npix", i64), ("C", i32), ("Cs", i32), ("n", i32),
                ("alpha", f32), ("beta", f32), ("k", f32)]


class PoolArgs(C.Structure):
    _fields_ = [("x_or_dy", vp), ("y_or_dx", vp), ("idx", vp), ("mask", vp)] + \
               [(n, i32) for n in ("B", "H", "W", "Cs", "accumulate")]


class AttnArgs(C.Structure):
    _fields_ = [(n, vp) for n in ("f", "g", "h", "x_or_dy", "y", "stats", "df", "dg", "dh", "ws")] + \
               [(n, i32) for n in ("B", "N", "Ds", "Cs")]


class CsumArgs(C.Structure):
    _fields_ = [("g", vp), ("npix", i64), ("Cs", i32), ("partials", vp), ("nblk", i32)]


class ReduceArgs(C.Structure):
    _fields_ = [("slab_arena", vp), ("flat_grads", vp), ("table_dev", vp), ("n_entries", i32), ("max_elems", i32)]


class AllreduceArgs(C.Structure):
    _fields_ = [("comm", vp), ("buf", vp), ("count", i64)]


class OwnerCtx(C.Structure):
    _fields_ = [(n, vp) for n in ("owner", "boxes", "feats", "w", "wt", "table", "sums", "csum")] + \
               [(n, i32) for n in ("n_boxes", "n_vec", "C", "csum_blocks", "wt_floats", "reserved0")]


class AttnProjBwdArgs(C.Structure):
    _fields_ = [(n, vp) for n in ("df", "dg", "dh", "wf_pack", "wg_pack", "wh_pack", "add", "mask_b", "dx")] + \
               [("npix", i64), ("C", i32), ("accumulate", i32)]


class Dgrad2Args(C.Structure):
    _fields_ = [(n, vp) for n in ("g", "w1_pack", "w2_pack", "dx1", "dx2", "mask1", "mask2")] + \
               [("npix", i64), ("C", i32), ("accumulate1", i32), ("accumulate2", i32), ("reserved", i32)]


OP_SIDE = 0x100
OP_PROBE = 0x200
OP_COMM = 0x400
OP_JOIN = 0x800
OP_ALLREDUCE = 14
OP_WGRAD_REDUCE = 10
OP_CONV_PAIR = 11
OP_BOX_FWD, OP_BOX_BWD = 12, 13
OP_ATTN_PROJ_BWD = 15
OP_DGRAD2_1X1 = 16
OP_CONV2D, OP_WGRAD, OP_LRN_FWD, OP_LRN_BWD, OP_POOL_FWD, OP_POOL_BWD, OP_ATTN_FWD, OP_ATTN_BWD, OP_CHANNEL_SUM = range(1, 10)

_SIGNATURES = {
    "msau_last_error": (C.c_char_p, []),
    "msau_version": (C.c_int, []),
    "msau_source_hash": (C.c_char_p, []),
    "msau_sizeof": (C.c_int, [C.c_int]),
    "msau_lds_pixel_stride": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "msau_lds_wrow_stride": (C.c_int, [C.c_int, C.c_int]),
    "msau_conv_pack_geometry": (C.c_int, [C.c_int] * 9 + [C.POINTER(ConvPackGeom)]),
    "msau_conv2d": (C.c_int, [vp, C.c_int, C.POINTER(ConvDesc)]),
    "msau_conv2d_rider_slabs": (C.c_int, [C.c_int, C.POINTER(ConvDesc)]),
    "msau_attn_proj_bwd": (C.c_int, [vp, C.c_int, C.POINTER(AttnProjBwdArgs)]),
    "msau_dgrad2_1x1": (C.c_int, [vp, C.c_int, C.POINTER(Dgrad2Args)]),
    "msau_conv2d_launch_info": (C.c_int, [C.c_int, C.POINTER(ConvDesc), C.POINTER(i32)]),
    "msau_conv_pair_applicable": (C.c_int, [C.c_int, C.POINTER(ConvPairDesc)]),
    "msau_conv_pair": (C.c_int, [vp, C.c_int, C.POINTER(ConvPairDesc)]),
    "msau_conv_pair_bits_bytes": (C.c_int64, [C.c_int, C.POINTER(ConvPairDesc)]),
    "msau_conv_pair_instance": (C.c_int, [C.c_int, C.POINTER(ConvPairDesc)]),
    "msau_conv_pair_wgrad_slabs": (C.c_int, [C.c_int, C.POINTER(ConvPairDesc)]),
    "msau_owner_slabs": (C.c_int, [C.POINTER(WgradDesc)]),
    "msau_comm_available": (C.c_int, []),
    "msau_comm_unique_id": (C.c_int, [vp, C.c_int]),
    "msau_comm_init": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, vp, C.c_int]),
    "msau_comm_destroy": (C.c_int, [vp]),
    "msau_allreduce_bucket": (C.c_int, [vp, vp, vp, C.c_int64]),
    "msau_run_ops_dp": (C.c_int, [vp, vp, vp, C.POINTER(Op), C.c_int, C.c_int]),
    "msau_reload_env": (None, []),
    "msau_wgrad_geometry": (C.c_int, [C.c_int, C.POINTER(WgradDesc), C.POINTER(WgradGeom)]),
    "msau_conv2d_wgrad": (C.c_int, [vp, C.c_int, C.POINTER(WgradDesc)]),
    "msau_conv2d_wgrad_groupable": (C.c_int, [C.c_int, C.POINTER(WgradDesc), C.POINTER(WgradDesc)]),
    "msau_conv2d_wgrad_group": (C.c_int, [vp, C.c_int, C.POINTER(C.POINTER(WgradDesc)), C.c_int]),
    "msau_pack_params": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int]),
    "msau_wgrad_reduce": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int]),
    "msau_channel_sum": (C.c_int, [vp, C.c_int, vp, i64, C.c_int, vp, C.c_int]),
    "msau_nchw_to_nhwc": (C.c_int, [vp, C.c_int, vp, vp] + [C.c_int] * 5),
    "msau_nhwc_to_nchw": (C.c_int, [vp, C.c_int, vp, vp] + [C.c_int] * 5),
    "msau_nchw_grad_to_nhwc": (C.c_int, [vp, C.c_int, vp, vp] + [C.c_int] * 6),
    "msau_lrn_fwd": (C.c_int, [vp, C.c_int, vp, vp, i64, C.c_int, C.c_int, C.c_int, f32, f32, f32]),
    "msau_lrn_bwd": (C.c_int, [vp, C.c_int, vp, vp, vp, i64, C.c_int, C.c_int, C.c_int, f32, f32, f32]),
    "msau_maxpool2x2_fwd": (C.c_int, [vp, C.c_int, vp, vp, vp] + [C.c_int] * 4),
    "msau_maxpool2x2_bwd": (C.c_int, [vp, C.c_int, vp, vp, vp, vp] + [C.c_int] * 5),
    "msau_selfattn_fwd": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp] + [C.c_int] * 4),
    "msau_selfattn_bwd": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp] + [C.c_int] * 4),
    "msau_label_counts": (C.c_int, [vp, vp, vp, C.c_int, i64]),
    "msau_label_counts_split": (C.c_int, [vp, vp, vp, C.c_int, i64, C.c_int]),
    "msau_ce_ws_floats": (i64, [i64]),
    "msau_ce_multi_ws_floats": (i64, [i64]),
    "msau_masked_ce_multi": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, i64, C.c_int, C.c_int, f32, C.c_int]),
    "msau_masked_ce": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, i64, C.c_int, C.c_int, f32]),
    "msau_softmax_ce": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, i64, C.c_int, C.c_int, f32]),
    "msau_softmax_ce_weighted": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, i64, C.c_int, C.c_int]),
    "msau_adam_ws_floats": (i64, [i64]),
    "msau_clip_adam_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32]),
    "msau_raster_owner": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]),
    "msau_raster_onehot": (C.c_int, [vp, C.c_int, vp, vp, vp] + [C.c_int] * 5),
    "msau_raster_labels": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int]),
    "msau_raster_dense": (C.c_int, [vp, C.c_int, vp, vp, vp, vp] + [C.c_int] * 5),
    "msau_box_integral_ws_floats": (i64, [C.c_int] * 4),
    "msau_box_integral": (C.c_int, [vp, C.c_int, vp, vp, vp] + [C.c_int] * 6),
    "msau_box_params": (C.c_int, [vp, vp, i64, i64, i64, i64, C.c_int, C.c_int, f32, f32, vp, vp]),
    "msau_box_filter": (C.c_int, [vp, C.c_int, vp, vp, vp] + [C.c_int] * 8 + [vp, vp, vp]),
    "msau_box_fwd": (C.c_int, [vp, C.c_int, C.POINTER(BoxArgs)]),
    "msau_box_bwd": (C.c_int, [vp, C.c_int, C.POINTER(BoxArgs)]),
    "msau_box_pgrad_ws_floats": (i64, [C.c_int] * 5),
    "msau_box_param_grad": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, i64, i64, i64, i64] + [C.c_int] * 6 + [f32, f32]),
    "msau_run_ops": (C.c_int, [vp, C.POINTER(Op), C.c_int]),
    "msau_run_ops_overlap": (C.c_int, [vp, vp, C.POINTER(Op), C.c_int, C.c_int]),
    "msau_spin": (C.c_int, [vp, C.c_int]),
    "msau_fill_zero": (C.c_int, [vp, vp, i64]),
    "msau_fork_visibility_check": (C.c_int, [vp, vp, C.c_int, i64, C.c_int, C.POINTER(i64)]),
    "msau_stream_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "msau_stream_destroy": (C.c_int, [vp]),
    "msau_softmax_channels_nchw": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, i64]),
    "msau_probe_read": (C.c_int, [vp, C.c_int, vp]),
    "msau_probe_overhead": (C.c_int, [vp, C.c_int, vp]),
    "msau_softmax_argmax_nhwc": (C.c_int, [vp, C.c_int, vp, vp, vp, i64, C.c_int, C.c_int]),
    "msau_onehot_ids": (C.c_int, [vp, C.c_int, vp, vp, i64, C.c_int, C.c_int]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

# ctypes mirrors in the order of msau_sizeof(which): load() refuses a library whose structs have another size
ABI_STRUCTS = (ConvDesc, WgradDesc, PackEntry, UnpackEntry, Op, LrnArgs, PoolArgs, AttnArgs, CsumArgs, ReduceArgs,
               ConvPackGeom, WgradGeom, ConvPairDesc, BoxArgs, AllreduceArgs, OwnerCtx, AttnProjBwdArgs, Dgrad2Args)


class MsauHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load the library (once).  Raises if it is missing -- build it with `python -m msau_amd.build`."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MsauHipError(f"{LIB_PATH} not found: the MSAU HIP kernels are not built "
                           f"(run `python -m msau_amd.build`); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the ABI and this table disagree
        fn.restype = res
        fn.argtypes = args
    for which, st in enumerate(ABI_STRUCTS):
        if lib.msau_sizeof(which) != C.sizeof(st):
            raise MsauHipError(f"{LIB_PATH}: sizeof({st.__name__}) is {lib.msau_sizeof(which)} in the library, "
                               f"{C.sizeof(st)} in msau_amd/_lib.py -- rebuild (python -m msau_amd.build)")
    # the library must be the build of the kernel sources that sit beside it (the .so ships with the tree, git-ignored: nothing
    # else ties the two together).  MSAU_HIP_LIB (an explicitly chosen library) and a source-less install skip the comparison.
    if "MSAU_HIP_LIB" not in os.environ and os.path.isdir(os.path.join(HERE, "csrc")):
        from .build import source_hash
        have, want = lib.msau_source_hash().decode(), source_hash()
        if have != want:
            raise MsauHipError(f"{LIB_PATH} was built from kernel sources {have}, the tree holds {want}: rebuild "
                               f"(python -m msau_amd.build)")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().msau_last_error().decode("utf-8", "replace")
        raise MsauHipError(f"{what or 'msau call'} failed (status {rc}): {msg}")


class Profiler:
    """Per-call HIP-event timing of C-ABI launches on the current stream (bench / roofline only)."""

    def __init__(self):
        self.records = []          # (key, start_event, end_event)
        self.key = None

    def summary(self):
        import collections
        out = collections.OrderedDict()
        for key, e0, e1 in self.records:
            c, t = out.get(key, (0, 0.0))
            out[key] = (c + 1, t + e0.elapsed_time(e1))
        return out


_profiler = None


def set_profiler(p):
    global _profiler
    _profiler = p


def call(name: str, *args, key=None):
    """Call an int-status entry point and raise on failure."""
    if _profiler is None:
        check(getattr(load(), name)(*args), name)
        return
    import torch
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    check(getattr(load(), name)(*args), name)
    e1.record()
    _profiler.records.append((key or name, e0, e1))


# ---- a side stream that really runs beside the current one ------------------------------------------------------
_side_streams: dict = {}


def concurrent_stream(device=None, index: int = 0):
    """A HIP stream whose kernels overlap with the current stream's.  HIP multiplexes streams onto a few hardware
    queues; two streams that land on the same queue serialise, and which queue a new stream gets depends on how many
    streams the process (torch, RCCL, ...) created before.  Measured 2026-10-03 on MI355X / ROCm 7.2: with an RCCL
    communicator initialised first, the weight-gradient side stream shared the main stream's queue in about half of the
    runs and the step went from 5.2 ms to 6.5 ms (= no overlap at all).  So: create candidates and keep the first one on
    which a tiny kernel finishes while the current stream is still busy with a 3 ms spin.  Cached per (device, stream,
    index); different `index` values give different streams (weight gradients: 0, gradient all-reduce: 1)."""
    import time
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    main = torch.cuda.current_stream(dev)
    key = (dev.index, main.cuda_stream, index)
    if key in _side_streams:
        return _side_streams[key]
    probe = torch.zeros(1024, dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    chosen, tried = None, []
    for _ in range(8):                                       # (an explicit lowest / highest queue priority for this stream: 0 %, round 3)
        cand = torch.cuda.Stream(device=dev)
        tried.append(cand)                                   # keep rejected candidates alive: their queue slots stay taken
        call("msau_spin", main.cuda_stream, 3000)
        call("msau_fill_zero", cand.cuda_stream, probe.data_ptr(), 4096)
        ev = torch.cuda.Event()
        ev.record(cand)
        t0 = time.perf_counter()
        overlapped = False
        while time.perf_counter() - t0 < 0.002:              # well inside the 3 ms spin
            if ev.query():
                overlapped = True
                break
        torch.cuda.synchronize(dev)
        if overlapped:
            chosen = cand
            break
    if chosen is None:
        chosen = tried[-1]                                   # no concurrency to be had: correct, just not overlapped
    _side_streams[key] = chosen
    _side_streams[("rejected",) + key] = tried
    return chosen
