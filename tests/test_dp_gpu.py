"""GPU, 2 ranks sharing cuda:0 over gloo (RCCL refuses two ranks on one device; the 8-GPU RCCL run is the
driver's): the whole data-parallel TrainEngine -- HIP forward/backward per rank, bucketed all-reduce of the
flat gradient, 1/world folded into clip+Adam -- must reproduce the single-process step on the global batch."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

KW = dict(scale_space_num=3, res_depth=2, featRoot=8, final_act="softmax", dtype="fp32", seed=21)


def _data():
    from oracle import msau_oracle as O
    x, label = O.synthetic_batch(4, 13, 40, 36, 5, seed=5)
    label[2] = 0                                   # a sample without labels
    return x, label


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from msau_amd import MSAUWrapper, TrainEngine
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
      try:
        torch.cuda.set_device(0)
        m = MSAUWrapper(13, 5, KW).cuda()
        eng = TrainEngine(m)
        assert eng.world == world
        x, label = _data()
        lo = rank * 2
        for _ in range(2):
            loss = eng.step(x[lo:lo + 2].cuda(), label[lo:lo + 2].cuda())
        torch.cuda.synchronize()
        q.put((rank, float(loss), m.flat_parameters.cpu().numpy()))
        dist.barrier()
      except Exception as e:                                    # surface the worker's error in the parent
        import traceback
        q.put((rank, None, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def test_two_rank_engine_equals_single_process_global_batch():
    from msau_amd import MSAUWrapper, TrainEngine
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for g in got:
        assert g[1] is not None, g[2]
    got = [(r, l, torch.from_numpy(p)) for r, l, p in got]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m = MSAUWrapper(13, 5, KW).cuda()
    eng = TrainEngine(m)
    x, label = _data()
    for _ in range(2):
        loss = eng.step(x.cuda(), label.cuda())
    ref = m.flat_parameters.cpu()
    assert torch.equal(got[0][2], got[1][2])                               # replicas stay bit-identical
    # two Adam steps of lr 1e-4 move a weight by <= 2e-4: the sharded run must agree to summation-order noise (Adam's
    # g / sqrt(v) turns a relative gradient difference into the same relative step difference: 5 % of a step is the bar)
    assert float((got[0][2] - ref).abs().max()) < 1e-5
    assert abs(0.5 * (got[0][1] + got[1][1]) - float(loss)) < 1e-5 * abs(float(loss))   # mean of local losses


def _rccl_worker(port, q):
    """one rank, an RCCL process group: the exchange goes through the C ABI (msau_allreduce_bucket inside msau_run_ops_dp)"""
    import torch.distributed as dist
    from msau_amd import MSAUWrapper, TrainEngine
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MSAU_FORCE_DIST="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        try:
            x, label = _data()
            out = {}
            for native in ("1", "0"):
                os.environ["MSAU_DP_NATIVE"] = native
                m = MSAUWrapper(13, 5, KW).cuda()
                eng = TrainEngine(m)
                assert (eng._comm is not None) == (native == "1") and eng.sync.active
                for _ in range(3):
                    loss = eng.step(x.cuda(), label.cuda())
                torch.cuda.synchronize()
                plan = m._plan_for(x.cuda(), True)
                # native: the backward sequence carries one all-reduce record per stage bucket + the end-conv tail
                n_ar = sum(1 for i in range(plan._bwd_seq_dp[1]) if (plan._bwd_seq_dp[0][i].kind & 0xff) == 14) if native == "1" else 0
                out[native] = (float(loss), m.flat_parameters.cpu().numpy(), n_ar)
            q.put(("ok", out))
        except Exception:
            import traceback
            q.put(("err", traceback.format_exc()))
            raise
    finally:
        dist.destroy_process_group()


def test_gradient_exchange_through_the_c_abi_equals_the_torch_distributed_path():
    """msau_comm_* + msau_allreduce_bucket (csrc/comm.hip, RCCL) inside the native backward sequence against the
    torch.distributed path (msau_amd/dp.py) and against no exchange at all, at world size 1 (an identity all-reduce that still
    runs the whole machinery: communicator from a broadcast id, comm stream, forks behind each stage's slab reduction, join
    before the optimiser).  More ranks need more GPUs: the driver's run."""
    from msau_amd import MSAUWrapper, TrainEngine
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    tag, out = q.get(timeout=300)
    p.join(timeout=60)
    assert tag == "ok", out
    assert p.exitcode == 0
    assert out["1"][2] == KW.get("num_blocks", 3) + 1
    m = MSAUWrapper(13, 5, KW).cuda()
    eng = TrainEngine(m)
    x, label = _data()
    for _ in range(3):
        loss = eng.step(x.cuda(), label.cuda())
    ref = m.flat_parameters.cpu().numpy()
    import numpy as np
    assert np.array_equal(out["1"][1], out["0"][1]) and np.array_equal(out["1"][1], ref)
    assert out["1"][0] == out["0"][0] == float(loss)


def test_bench_py_scale_command_path_with_two_ranks():
    """The exact command path of the driver's SCALE run, which no 8-GPU node has exercised yet: `bench.py --gpus 2` relaunches
    itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ...` as a CHILD process
    (before its own first GPU call), every rank builds the engine with a process group, the timed region is bracketed by
    barrier + synchronize, the MAX over ranks is taken and rank 0 prints ONE JSON line.  Here both ranks share cuda:0
    (MSAU_BENCH_DEVICE=0) over gloo (RCCL refuses two ranks on one device); a smaller image keeps it to seconds."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MSAU_BENCH_DEVICE="0", MSAU_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MSAU_FORCE_DIST", "MSAU_DP_NATIVE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4",
           "--height", "168", "--width", "128", "--no-secondary", "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                           # ONE JSON line on stdout, whatever the libraries print
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp2"
    assert out["unit"] == "tiles/s" and out["value"] > 0 and out["higher_is_better"] is True
    assert abs(out["value"] - 8 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-2 * out["value"]     # whole-job tiles over the MAX-over-ranks time
    loss = out["config"]["loss"]
    assert loss == loss and 0.0 < loss < 50.0, loss                         # finite
