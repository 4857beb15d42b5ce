"""Data-parallel gradient synchronisation (one process per GPU, RCCL over xGMI; gloo on CPU for tests).

The reference has no distributed code at all; this is the new piece the north star asks
for.  Tiles are independent, so the only exchange step is ONE sum all-reduce of the flat fp32
gradient buffer (636 167 floats = 2.5 MB for cfg 2) per step; the 1/world factor is folded into the
fused clip+Adam kernel (`grad_scale`).  The message is latency-bound (tens of microseconds over
xGMI), so the buffer is reduced in at most `n_buckets` contiguous pieces -- stage by stage, in the
order backward finishes them -- on a side stream, and joined before the optimiser.

Loss rule that makes sharding invisible: every rank computes
    loss_r = mean over its B_local samples of [masked-mean CE(final) + masked-mean CE(aux)]
so that  mean_r(grad loss_r)  ==  grad of the same expression over the global batch.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def stage_buckets(param_offsets: "dict[str, int]", total: int, num_blocks: int) -> List[Tuple[int, int]]:
    """[lo, hi) element ranges of the flat buffer, one per stage (last stage first = backward order).
    The end convs live after all blocks in registration order and join the first bucket issued."""
    starts = []
    for b in range(num_blocks):
        offs = [o for k, o in param_offsets.items() if k.startswith(f"msau_net.blocks.{b}.")]
        starts.append(min(offs))
    ends = starts[1:] + [min(o for k, o in param_offsets.items() if k.startswith("msau_net.end_convs."))]
    buckets = [(starts[b], ends[b]) for b in range(num_blocks)]
    buckets.append((ends[-1], total))                       # end convs
    # backward order: last stage first; merge the end-conv tail into the last stage's bucket if adjacent
    order = [buckets[-1]] + buckets[num_blocks - 1::-1]
    return [b for b in order if b[1] > b[0]]


class GradSync:
    """Sum all-reduce of a flat gradient buffer in contiguous buckets, asynchronously."""

    def __init__(self, flat_grad: torch.Tensor, buckets: Optional[Sequence[Tuple[int, int]]] = None,
                 group=None):
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # MSAU_FORCE_DIST=1 keeps the exchange step in the sequence at world size 1 (an identity all-reduce): a
        # rehearsal of the RCCL path -- side stream, per-stage buckets, joins -- on a one-GPU box
        self.active = self.world > 1 or (os.environ.get("MSAU_FORCE_DIST") == "1" and dist.is_available() and dist.is_initialized())
        n = flat_grad.numel()
        self.buckets = list(buckets) if buckets else [(0, n)]
        covered = sorted(self.buckets)
        assert covered[0][0] == 0 and covered[-1][1] == n and all(a[1] == b[0] for a, b in zip(covered, covered[1:])), \
            "buckets must tile the flat buffer"
        self._pending = []
        self._side = None
        if flat_grad.is_cuda:
            from . import _lib
            self._side = _lib.concurrent_stream(flat_grad.device, index=1)      # overlaps with the compute stream

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def start(self, i: int, after=None):
        """Issue bucket i.  Its gradients must be final on stream `after` (default: the current stream)."""
        if not self.active:
            return
        lo, hi = self.buckets[i]
        view = self.flat[lo:hi]
        if self._side is not None:
            self._side.wait_stream(after if after is not None else torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                self._pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def start_whole(self):
        """one all-reduce of the whole flat buffer (all gradients final on the current stream)"""
        if not self.active:
            return
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                self._pending.append(dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._pending.append(dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def start_all(self):
        for i in range(len(self.buckets)):
            self.start(i)

    def finish(self):
        """Join every outstanding bucket; afterwards the buffer holds the SUM over ranks."""
        for w in self._pending:
            w.wait()
        self._pending.clear()
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
