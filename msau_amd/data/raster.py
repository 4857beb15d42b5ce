"""Device-side chargrid rasteriser: documents -> compact box lists (a few KB) -> one-hot NHWC
grid + label mask painted by HIP kernels (msau_amd/csrc/raster.hip), bit-identical to the CPU painter
`funsd.get_box_mask_box_label_word` (which is pinned to the reference)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from .. import _lib as L
from .funsd import chargrid_geometry, line_boxes, word_char_boxes


def document_boxes(doc: dict, sample: int = 0) -> Tuple[np.ndarray, np.ndarray, int, int]:
    """-> (char boxes [n,6], label boxes [m,6], H, W) for one preprocessed document.
    char value = index of the character in the charset (argmax of its one-hot row), -1 if unknown
    (an all-zero row: painting it clears whatever was underneath, as the reference does)."""
    words, lines = doc["cells_word"], doc["cells"]
    geo = chargrid_geometry(words)
    chars = []
    for wi, j, y0, y1, x0, x1 in word_char_boxes(words, geo):
        row = doc["charset_feature"][wi][j]
        v = int(np.argmax(row)) if row.any() else -1
        chars.append((sample, y0, y1, x0, x1, v))
    labs = []
    for li, c in enumerate(lines):
        x = int((c.x - geo["min_x"]) / geo["min_w"])
        y = int((c.y - geo["min_y"]) / geo["min_h"])
        w = max(int(c.w / geo["min_w"]), 1)
        h = max(int(c.h / geo["min_h"]), 1)
        labs.append((sample, y, y + h, x, x + w, int(doc["labels"][li]) + 1))
    return (np.asarray(chars, np.int32).reshape(-1, 6), np.asarray(labs, np.int32).reshape(-1, 6), geo["H"], geo["W"])


def _dev_boxes(boxes, dev):
    """box list -> (device int32 [n][6] tensor or None, n); numpy arrays are uploaded, device tensors taken as they are"""
    if isinstance(boxes, torch.Tensor):
        assert boxes.dtype == torch.int32 and boxes.is_contiguous() and boxes.device == dev
        return (boxes if boxes.numel() else None), int(boxes.shape[0])
    arr = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 6)
    return (torch.from_numpy(arr).to(dev) if len(arr) else None), len(arr)


def rasterize(char_boxes, label_boxes, B: int, H: int, W: int, C: int,
              dtype: str = "bf16", device="cuda", out: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (grid [B,H,W,Cs] one-hot in `dtype` storage, labels int64 [B,H,W]) on `device`.  `out`: paint the grid into this
    buffer (a plan's own input buffer: TrainEngine.input_nhwc) instead of a new tensor.  Box lists may be numpy arrays
    (uploaded here) or int32 device tensors (nothing crosses PCIe, no host synchronisation)."""
    dt = L.BF16 if dtype in ("bf16", "bfloat16") else L.F32
    Cs = -(-C // 8) * 8
    dev = torch.device(device)
    if dev.index is None and dev.type == "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    s = torch.cuda.current_stream(dev).cuda_stream
    owner = torch.empty((B, H, W), dtype=torch.int32, device=dev)
    tdt = torch.bfloat16 if dt == L.BF16 else torch.float32
    grid = out if out is not None else torch.empty((B, H, W, Cs), dtype=tdt, device=dev)
    assert tuple(grid.shape) == (B, H, W, Cs) and grid.dtype == tdt and grid.is_contiguous(), (grid.shape, grid.dtype)
    labels = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    keep = []
    for boxes, kind in ((char_boxes, "grid"), (label_boxes, "labels")):
        bt, n = _dev_boxes(boxes, dev)
        keep.append(bt)
        L.call("msau_raster_owner", s, bt.data_ptr() if n else None, n, owner.data_ptr(), B, H, W)
        if kind == "grid":
            L.call("msau_raster_onehot", s, dt, bt.data_ptr() if n else None, owner.data_ptr(), grid.data_ptr(), B, H, W, C, Cs)
        else:
            L.call("msau_raster_labels", s, bt.data_ptr() if n else None, owner.data_ptr(), labels.data_ptr(), B, H, W)
    # uploaded box lists must outlive the launches that read them; torch's caching allocator frees stream-ordered, and the
    # launches above are on torch's current stream -- record the tensors on it instead of synchronising the host
    for t in keep + [owner]:
        if t is not None:
            t.record_stream(torch.cuda.current_stream(dev))
    return grid, labels


def owner_maps(feat_boxes, label_boxes, B: int, H: int, W: int, device="cuda"):
    """-> (owner int32 [B,H,W]: index of the feature box that owns each pixel, -1 = none; the feature boxes as a device
    tensor [n,6] (or None) and n; labels int64 [B,H,W]).  What the dense painters start from -- and all that the net's first
    conv needs when it is fed with box lists (MSAU_CONV_OWNER, csrc/ownerconv.hip): the grid itself is never painted."""
    dev = torch.device(device)
    if dev.index is None and dev.type == "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    s = torch.cuda.current_stream(dev).cuda_stream
    owner = torch.empty((B, H, W), dtype=torch.int32, device=dev)
    lown = torch.empty((B, H, W), dtype=torch.int32, device=dev)
    labels = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    fb, nf = _dev_boxes(feat_boxes, dev)
    lb, nl = _dev_boxes(label_boxes, dev)
    L.call("msau_raster_owner", s, fb.data_ptr() if nf else None, nf, owner.data_ptr(), B, H, W)
    L.call("msau_raster_owner", s, lb.data_ptr() if nl else None, nl, lown.data_ptr(), B, H, W)
    L.call("msau_raster_labels", s, lb.data_ptr() if nl else None, lown.data_ptr(), labels.data_ptr(), B, H, W)
    for t in (fb, lb, lown):
        if t is not None:
            t.record_stream(torch.cuda.current_stream(dev))
    return owner, fb, nf, labels


def document_line_boxes(doc: dict, sample: int = 0, feat_base: int = 0):
    """-> (feature boxes [n,6] with value = feat_base + line index, label boxes [n,6], H, W) for the dense (BERT) painter
    `funsd.get_box_mask_box_label`: both the feature vectors and the labels cover the text-LINE boxes"""
    H, W, boxes = line_boxes(doc["cells"])
    fb = [(sample, y0, y1, x0, x1, feat_base + li) for li, y0, y1, x0, x1 in boxes]
    lb = [(sample, y0, y1, x0, x1, int(doc["labels"][li]) + 1) for li, y0, y1, x0, x1 in boxes]
    return np.asarray(fb, np.int32).reshape(-1, 6), np.asarray(lb, np.int32).reshape(-1, 6), H, W


def rasterize_dense(feat_boxes, label_boxes, feats, B: int, H: int, W: int,
                    dtype: str = "bf16", device="cuda", out: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (grid [B,H,W,Cs] with feats[value] painted over each box in `dtype` storage, labels int64 [B,H,W]); only the
    box list and the [n_lines, C] feature table cross PCIe (at 768 channels the dense fp32 grid is 264 MB per tile).  `out`,
    device-tensor arguments: as `rasterize`."""
    dt = L.BF16 if dtype in ("bf16", "bfloat16") else L.F32
    dev = torch.device(device)
    if dev.index is None and dev.type == "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    if isinstance(feats, torch.Tensor):
        assert feats.dtype == torch.float32 and feats.is_contiguous() and feats.device == dev
        ft = feats
    else:
        ft = torch.from_numpy(np.ascontiguousarray(feats, dtype=np.float32)).to(dev)
    C = int(ft.shape[1])
    Cs = -(-C // 8) * 8
    s = torch.cuda.current_stream(dev).cuda_stream
    owner = torch.empty((B, H, W), dtype=torch.int32, device=dev)
    tdt = torch.bfloat16 if dt == L.BF16 else torch.float32
    grid = out if out is not None else torch.empty((B, H, W, Cs), dtype=tdt, device=dev)
    assert tuple(grid.shape) == (B, H, W, Cs) and grid.dtype == tdt and grid.is_contiguous(), (grid.shape, grid.dtype)
    labels = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    keep = [ft]
    for boxes, kind in ((feat_boxes, "grid"), (label_boxes, "labels")):
        bt, n = _dev_boxes(boxes, dev)
        keep.append(bt)
        L.call("msau_raster_owner", s, bt.data_ptr() if n else None, n, owner.data_ptr(), B, H, W)
        if kind == "grid":
            L.call("msau_raster_dense", s, dt, bt.data_ptr() if n else None, owner.data_ptr(), ft.data_ptr(), grid.data_ptr(), B, H, W, C, Cs)
        else:
            L.call("msau_raster_labels", s, bt.data_ptr() if n else None, owner.data_ptr(), labels.data_ptr(), B, H, W)
    for t in keep + [owner]:
        if t is not None:
            t.record_stream(torch.cuda.current_stream(dev))
    return grid, labels
