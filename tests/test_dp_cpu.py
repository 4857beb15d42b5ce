"""CPU, world_size 2, gloo: the data-parallel path.  Kernels cannot run here, so the per-rank
gradients come from the CPU oracle; what is under test is the product's sharding contract --
flat-buffer layout (MSAUWrapper._poff), stage buckets, GradSync's bucketed async all-reduce and the
1/world scale -- against the single-process result on the same global batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from msau_amd.dp import GradSync, stage_buckets
from msau_amd.model import MSAUWrapper
from oracle import msau_oracle as O

CFG = dict(channels=13, n_class=5, scale_space_num=3, res_depth=2, featRoot=4, filter_size=3, pool_size=2, num_blocks=3)


def _flat_grads(model, sd, x, label):
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits, aux = O.msau_forward(leaves, x, CFG)
    O.msau_loss(logits, aux, label).backward()
    flat = torch.zeros(model._total)
    for k, off in model._poff.items():
        g = leaves[k].grad
        if g is not None:
            flat[off:off + g.numel()] = g.reshape(-1)
    return flat


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kw = dict(scale_space_num=3, res_depth=2, featRoot=4, final_act="softmax", seed=5)
        model = MSAUWrapper(13, 5, kw)
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        x, label = O.synthetic_batch(4, 13, 24, 20, 5, seed=77)
        label[1] = 0                                       # a sample without labels contributes 0
        lo, hi = rank * 2, rank * 2 + 2
        flat = _flat_grads(model, sd, x[lo:hi], label[lo:hi])
        sync = GradSync(flat, stage_buckets(model._poff, model._total, model.num_blocks))
        assert sync.world == world and abs(sync.grad_scale - 1.0 / world) < 1e-12
        sync.start_all()
        sync.finish()
        flat *= sync.grad_scale
        if rank == 0:
            full = _flat_grads(model, sd, x, label)
            out.put((float((flat - full).abs().max()), float(full.abs().max()), len(sync.buckets)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gradient_equals_single_process_batch():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    err, scale, nb = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert nb == 4                                          # end convs + 3 stages
    assert err <= 2e-6 * scale, (err, scale)                # fp32, different summation order only


def test_stage_buckets_tile_the_flat_buffer_in_backward_order():
    m = MSAUWrapper(64, 5, dict(scale_space_num=4, res_depth=2, featRoot=8, final_act="softmax"))
    b = stage_buckets(m._poff, m._total, 3)
    assert sorted(b)[0][0] == 0 and sorted(b)[-1][1] == m._total
    assert sum(hi - lo for lo, hi in b) == m._total
    # issued last-stage first (the order backward finishes them)
    assert b[1][0] > b[2][0] > b[3][0]
    sizes = [hi - lo for lo, hi in b]
    assert sizes[1] == sizes[2]                             # stages 1 and 2 have identical parameter sets
