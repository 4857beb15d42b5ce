#!/usr/bin/env python3
"""Headline benchmark: chargrid tiles/sec, MSAU train step (fwd + loss + bwd + clip + Adam, + RCCL
gradient all-reduce when N > 1) on BASELINE.json's configs[1]: FUNSD-like one-hot chargrid,
3-stage MSAU, 336x256x64, bf16 storage, batch 16 per GPU, synthetic data, random-init weights.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel symbol: algorithmic bytes per launch / HIP-event launch time
  cpu_baseline -- the CPU oracle (port of the reference path) timed on this host's cores, N=1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16, help="tiles per GPU")
    ap.add_argument("--height", type=int, default=336)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--stages", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as HIP graphs instead of launching eagerly (measured slower on ROCm 7.2: "
                         "6.08 vs 5.73 ms/step; the host keeps up with ~450 launches of ~10 us)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurements (cfg 4: 768-channel 2-stage net; cfg 2 in fp32 storage; cfg 2 fed with "
                         "character-id masks instead of the dense one-hot tensor)")
    ap.add_argument("--secondary-steps", type=int, default=30)
    ap.add_argument("--cpu-steps", type=int, default=200)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="time budget of the CPU baseline sample")
    ap.add_argument("--dump-kernels", default="", help="write the per-kernel HIP-event table to this file")
    return ap.parse_args()


def synthetic(B, C, H, W, n_class, seed, device):
    """one-hot chargrid, 30 % occupancy, labels = occupied * U{1..n_class-1}."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    occ = torch.rand((B, H, W), generator=g) < 0.3
    ch = torch.randint(0, C, (B, H, W), generator=g)
    x = torch.zeros((B, C, H, W))
    x.scatter_(1, ch[:, None], occ[:, None].float())
    label = torch.randint(1, n_class, (B, H, W), generator=g) * occ.long()
    return x.to(device), label.to(device)


def usable_cores() -> int:
    """cores this process may actually use: min(affinity, cgroup cpu quota), never more than 64"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    env = os.environ.get("MSAU_CPU_THREADS")
    if env:
        n = int(env)
    return max(1, min(n, 64))


def kernel_source_hash() -> str:
    """sha256 over the HIP sources + the ABI header: ties a PMC traffic file -- and the built library (msau_source_hash()) -- to
    the kernels it was measured on"""
    from msau_amd.build import source_hash
    return source_hash()


def load_traffic(key):
    """HBM bytes per launch of `key` from the PMC file of this round -- only if it was measured on exactly these kernel
    sources (tools/make_traffic.py writes the hash); otherwise (None, reason)"""
    path = os.path.join(ROOT, "profiles", "r05_traffic.json")
    try:
        tj = json.load(open(path))
    except Exception as e:                                   # noqa: BLE001
        return None, f"no traffic file ({e.__class__.__name__})"
    if tj.get("kernel_source_hash") != kernel_source_hash():
        return None, (f"profiles/r05_traffic.json was measured on kernel sources {tj.get('kernel_source_hash')}, this tree is "
                      f"{kernel_source_hash()}: re-run tools/make_traffic.py")
    ent = tj.get("kernels", {}).get(key)
    if ent is None:
        return None, f"profiles/r05_traffic.json has no entry for {key}"
    return ent, None


def load_pmc(key):
    """SQ-counter numbers of the kernel family `key` from tools/pmc_step.py's file, if measured on these kernel sources"""
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc.json")))
    except Exception as e:                                   # noqa: BLE001
        return None, f"no PMC file ({e.__class__.__name__})"
    if pj.get("kernel_source_hash") != kernel_source_hash():
        return None, f"profiles/r05_pmc.json was measured on kernel sources {pj.get('kernel_source_hash')}, this tree is {kernel_source_hash()}: re-run tools/pmc_step.py"
    fam = pj.get("families", {})
    ent = fam.get(key) or next((v for k, v in fam.items() if key.startswith(k)), None)
    return (ent, None) if ent else (None, f"profiles/r05_pmc.json has no family for {key}")


def profile_pass(L, eng, plan, x, label, nprof):
    """per-launch HIP-event timing of `nprof` eager steps with nothing overlapping -> rows (ms, key, count, meta)"""
    import torch
    prof = L.Profiler()
    for _ in range(nprof):
        # park the device behind a sleeping wave so the whole step queues up: the events then bracket
        # device execution only (no host launch gap inside a bracket)
        L.call("msau_spin", torch.cuda.current_stream().cuda_stream, 40000)
        L.set_profiler(prof)
        eng.step(x, label)
        L.set_profiler(None)
        torch.cuda.synchronize()
    rows = [(ms, key, cnt, plan.launch_meta.get(key)) for key, (cnt, ms) in prof.summary().items()]
    rows.sort(reverse=True)
    return rows


def probe_in_place(eng, plan, x, label, keys, steps, side=False):
    """durations of the launches of one stream (main; side=True: the weight-gradient stream) whose kernel key is in `keys`,
    measured inside the normal native sequence (the other stream's kernels running beside them):
    {key: (launches per step, avg us, algorithmic bytes per launch)}"""
    import torch
    order = plan.set_probe_keys(set(keys), side=side)
    plan.read_probe()
    acc = {}
    for _ in range(max(steps, 1)):
        eng.step(x, label)
        us = plan.read_probe()
        assert len(us) == len(order), (len(us), len(order))
        for (key, nbytes), t in zip(order, us):
            a = acc.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += t
            a[2] += nbytes
    plan.set_probe_keys(None)
    torch.cuda.synchronize()
    n = max(steps, 1)
    return {k: (c // n, t / c, b / c) for k, (c, t, b) in acc.items()}


def cpu_baseline(args, n_class):
    """The CPU restatement of the reference step (oracle/, validated against the reference's own
    outputs) on this host: fp32, batch-1 loop as in train_chargrid_funsd_msau.py:45-59."""
    from oracle import msau_oracle as O
    cfg = dict(O.DEFAULT_CFG, channels=args.channels, num_blocks=args.stages, n_class=n_class)
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd = O.init_params(cfg, seed=0)
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v = {k: torch.zeros_like(t) for k, t in sd.items()}
    x, label = O.synthetic_batch(1, args.channels, args.height, args.width, n_class, seed=99)
    step = 0
    for _ in range(2):
        step += 1
        O.train_step(sd, m, v, step, x, label, cfg)
    n = 0
    t0 = time.perf_counter()
    while n < max(1, args.cpu_steps) and (n == 0 or time.perf_counter() - t0 < args.cpu_seconds):
        step += 1
        n += 1
        O.train_step(sd, m, v, step, x, label, cfg)
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "tiles/s", "cores": cores, "kind": "port",
           "sample": f"{n} fp32 train steps (fwd+loss+bwd+clip+Adam) of batch 1 at {args.height}x{args.width}x{args.channels}, "
                     f"{args.stages}-stage, PyTorch-CPU restatement in oracle/msau_oracle.py, {dt:.1f} s"}
    # the same port with the GPU run's batch ("B=16 batched (fairer)"): 1 warm-up + >= 2 timed steps, bounded
    try:
        xb, lb = O.synthetic_batch(args.batch, args.channels, args.height, args.width, n_class, seed=98)
        O.train_step(sd, m, v, step + 1, xb, lb, cfg)
        nb, tb = 0, time.perf_counter()
        while nb < 2 or (nb < 4 and time.perf_counter() - tb < args.cpu_seconds / 2):
            nb += 1
            O.train_step(sd, m, v, step + 1 + nb, xb, lb, cfg)
        dtb = time.perf_counter() - tb
        out["batched"] = {"batch": args.batch, "value": nb * args.batch / dtb, "unit": "tiles/s", "steps": nb, "seconds": round(dtb, 1)}
    except Exception as e:                                   # noqa: BLE001  (e.g. host memory): the batch-1 figure stands
        out["batched"] = {"error": str(e)[:120]}
    return out


def secondary_measurements(args, L, MSAUWrapper, TrainEngine, dev, n_class):
    """Clearly separate from the headline: same step, other configurations, each its own model.  Every entry: tiles/s over
    `--secondary-steps` steps (inputs resident, no host sync inside), and the dominant kernel of that configuration with
    its serialised HIP-event time against the HBM roofline."""
    import torch
    out = []

    def run(name, channels, stages, dtype, batch, feed="dense", box=False, hw=None):
        H, W = hw or (args.height, args.width)
        kw = dict(scale_space_num=4, res_depth=2, featRoot=8, filter_size=3, pool_size=2, final_act="softmax",
                  num_blocks=stages, dtype=dtype, seed=0)
        if box:
            from msau_amd import BMSAUWrapper
            model = BMSAUWrapper(channels, n_class, kw).to(dev)        # reference defaults: 3 box convs, 3 boxes per channel, max 28
        else:
            model = MSAUWrapper(channels, n_class, kw).to(dev)
        eng = TrainEngine(model, lr=1e-4)
        x, label = synthetic(batch, channels, H, W, n_class, 4321, dev)
        if name.startswith("cfg4"):
            # BERT-like dense input: N(0,1) vectors at occupied pixels, zeros elsewhere
            g = torch.Generator(device="cpu").manual_seed(7)
            occ = (x.sum(1, keepdim=True) > 0).float()
            x = (torch.randn((batch, channels, H, W), generator=g).to(dev)) * occ
        ids = None
        if feed == "ids":
            occ = x.sum(1) > 0
            ids = torch.where(occ, x.argmax(1), torch.full_like(x.argmax(1), -1)).to(torch.int32).contiguous()
        step = (lambda: eng.step_ids(ids, label)) if feed == "ids" else (lambda: eng.step(x, label))
        if feed in ("device-painted", "device-painted, prefetched", "box-lists"):
            os.environ["MSAU_OWNER_CONV"] = "1" if feed == "box-lists" else "0"     # (read when the plan first meets a box-list step)
            # the BERT chargrid as its loader builds it (data_generator_funsd_bert.py:64-93): one feature vector per text line
            # painted over the line's box.  Box lists and the feature table live on the device; every step paints the grid
            # into the plan's own NHWC buffer (msau_raster_dense) and the label mask (msau_raster_labels) -- no fp32 NCHW tensor,
            # no conversion.  ~30 % of the pixels covered, as the dense entry's synthetic occupancy.
            import numpy as np
            rng = np.random.default_rng(7)
            fb, lb = [], []
            for b in range(batch):
                for r in range(0, H - 6, 8):                       # a line of text every 8 rows, 3 rows high, several fields per line
                    xx = int(rng.integers(0, 8))
                    while xx < W - 8:
                        w = int(rng.integers(24, 96))
                        fb.append((b, r, r + 3, xx, min(W, xx + w), len(fb)))
                        lb.append((b, r, r + 3, xx, min(W, xx + w), int(rng.integers(1, n_class))))
                        xx += w + int(rng.integers(4, 24))
            fbt = torch.from_numpy(np.asarray(fb, np.int32)).to(dev)
            lbt = torch.from_numpy(np.asarray(lb, np.int32)).to(dev)
            feats = torch.randn((len(fb), channels), device=dev)
            step = lambda: eng.step_boxes(fbt, lbt, batch, H, W, feats=feats)
            if feed == "device-painted, prefetched":
                # the loader's pipeline: the NEXT batch is painted (into the second input buffer, on the side stream) while this
                # one trains; every timed step still contains one paint and one optimisation step
                eng.prefetch_boxes(fbt, lbt, batch, H, W, feats=feats)

                def step():
                    eng.prefetch_boxes(fbt, lbt, batch, H, W, feats=feats)
                    return eng.step_prefetched()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.secondary_steps):
            loss = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ent = {"name": name, "value": round(batch * args.secondary_steps / dt, 1), "unit": "tiles/s",
               "ms_per_step": round(1e3 * dt / args.secondary_steps, 3), "dtype": dtype, "batch": batch, "feed": feed,
               "workload": f"{H}x{W}x{channels}, {stages}-stage" + (", box-convolution blocks (parity unpinned)" if box else ""),
               "loss": round(float(loss), 5)}
        if feed == "dense" and not args.no_roofline:
            plan = model._plan_for(x, training=True)
            rows = profile_pass(L, eng, plan, x, label, 3)
            ms, key, cnt, meta = next(r for r in rows if r[3] and not r[1].startswith("msau_"))
            us = 1e3 * ms / cnt
            gbs = meta[1] / meta[0] / (us * 1e-6) / 1e9
            ent["dominant_kernel"] = {"kernel": key, "launches_per_step": meta[0], "avg_launch_us_serialised": round(us, 2),
                                      "alg_MB_per_launch": round(meta[1] / meta[0] / 1e6, 2), "GB/s": round(gbs, 1),
                                      "frac": round(gbs / HBM_PEAK_GBS, 4), "share_of_step": round(ms / sum(r[0] for r in rows), 3)}
        out.append(ent)
        os.environ.pop("MSAU_OWNER_CONV", None)
        del eng, model, x, label
        torch.cuda.empty_cache()

    run("cfg4: BERT-embedding chargrid 336x256x768, 2-stage (BASELINE configs[3])", 768, 2, "bf16", args.batch)
    if hasattr(TrainEngine, "step_boxes"):
        run("cfg4 fed from box lists: the embedding grid is painted on the device into the plan's NHWC input every step (no fp32 NCHW tensor, no boundary conversion; SURVEY 8f N1)",
            768, 2, "bf16", args.batch, feed="device-painted")
    if hasattr(TrainEngine, "step_boxes"):
        run("cfg4 fed from box lists, the grid NEVER painted: the first conv gathers per-tap partial products per feature row, its weight gradient sums the output gradient per box (MSAU_CONV_OWNER, csrc/ownerconv.hip; SURVEY 8f N1)",
            768, 2, "bf16", args.batch, feed="box-lists")
    run("cfg2 in fp32 storage (the parity mode of the same kernels)", args.channels, args.stages, "fp32", args.batch)
    run("cfg5: model_box variant 512x384x64, 3-stage (BASELINE configs[4]; BoxConv2d is third-party: self-consistent only)", 64, 3,
        "bf16", args.batch, box=True, hw=(512, 384))
    # the reference's own operating point (train_chargrid_funsd_msau.py:45-59): batch 1, a different H x W per document
    try:
        import argparse as _ap
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import funsd_loop
        r = funsd_loop.run(_ap.Namespace(docs=64, epochs=2, channels=args.channels, dtype="bf16"), False)
        out.append({"name": "funsd-loop: batch 1, 64 documents of distinct FUNSD-like sizes (60-170 x 40-130), plans cached; the reference's "
                            "training regime, launch-bound (tools/funsd_loop.py; DESIGN section 6)",
                    "value": r["docs_per_s"], "unit": "docs/s", "ms_per_step": r["ms_per_doc"], "dtype": "bf16", "batch": 1, "feed": "dense",
                    "host_ms_per_step": r["host_ms_per_doc"], "first_epoch_ms_per_doc": r["first_epoch_ms_per_doc"],
                    "launches_per_step_median_doc": r["launches_per_step"], "workload": f"B=1, HxWx{args.channels}, {args.stages}-stage",
                    "loss": r["loss"]})
        torch.cuda.empty_cache()
    except Exception as e:                                   # noqa: BLE001  (a secondary line never takes the headline down)
        out.append({"name": "funsd-loop", "error": f"{e.__class__.__name__}: {e}"})
    if hasattr(TrainEngine, "step_ids"):
        run("cfg2 fed with character-id masks (the first conv and its weight gradient read the id mask and synthesise the one-hot tile in LDS: no dense input tensor at all; SURVEY 8f N1) -- NOT the headline input", args.channels,
            args.stages, "bf16", args.batch, feed="ids")
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # MSAU_BENCH_DEVICE: all ranks on one card (2-rank gloo rehearsal of the multi-GPU control flow on a one-GPU box)
    local_rank = int(os.environ.get("MSAU_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if args.gpus > 1 and world == 1:
        # convenience: relaunch under torch.distributed.run as a CHILD process (never exec after HIP init)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(29500 + os.getpid() % 1000), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the MSAU HIP path has no CPU fallback)")
    # stdout carries ONE JSON line.  Libraries write there too (RCCL prints a version banner when a communicator is created):
    # everything but the final line goes to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1 or os.environ.get("MSAU_FORCE_DIST") == "1":       # the latter: RCCL rehearsal on one GPU (msau_amd/dp.py)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("MSAU_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    from msau_amd import _lib as L
    from msau_amd.model import MSAUWrapper, TrainEngine

    n_class = 5
    kw = dict(scale_space_num=4, res_depth=2, featRoot=8, filter_size=3, pool_size=2, final_act="softmax",
              num_blocks=args.stages, dtype=args.dtype, seed=0)
    model = MSAUWrapper(args.channels, n_class, kw).to(dev)
    eng = TrainEngine(model, lr=1e-4, use_graph=args.graph)
    x, label = synthetic(args.batch, args.channels, args.height, args.width, n_class, 1234 + rank, dev)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(max(args.warmup, 1 if args.graph else 0)):
        eng.step(x, label)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = eng.step(x, label)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    tiles = world * args.batch * args.steps
    value = tiles / dt
    loss_val = float(loss)

    roof = None
    if rank == 0 and not args.no_roofline:
        # the measuring passes below run on rank 0 only: take the gradient exchange out of the step (a collective that
        # only one rank enters would hang); the other ranks wait at the barrier before the process group is torn down
        eng.sync.active = False
        eng.use_graph = False
        plan = model._plan_for(x, training=True)
        nprof = min(args.steps, 5)
        rows = profile_pass(L, eng, plan, x, label, nprof)
        total_ms = sum(r[0] for r in rows)
        if args.dump_kernels:
            with open(args.dump_kernels, "w") as f:
                f.write(f"# HIP-event per-launch timing, {nprof} eager steps, nothing overlapping, batch {args.batch}, {args.dtype}\n")
                f.write("# kernel, launches, total_ms, avg_us, share, alg_GB/s, alg_TFLOP/s\n")
                for ms, key, cnt, meta in rows:
                    gbs = tfs = float("nan")
                    if meta:
                        gbs = meta[1] * nprof / (ms * 1e-3) / 1e9
                        tfs = meta[2] * nprof / (ms * 1e-3) / 1e12
                    f.write(f"{key}, {cnt}, {ms:.3f}, {1e3 * ms / cnt:.1f}, {ms / total_ms:.3f}, {gbs:.0f}, {tfs:.1f}\n")
        # dominant kernel = the largest serialised share of the step among the launches of the op list, EITHER stream
        # (weight gradients included; msau_* boundary / loss / optimiser passes are not layer kernels)
        ms, key, cnt, meta = next(r for r in rows if r[3] and not r[1].startswith("msau_"))
        n_launch, alg_bytes, alg_flops = meta
        serial_us = 1e3 * ms / cnt
        # timed IN PLACE: the same native launch sequence as the timed region (weight gradients running beside it on the
        # side stream), a HIP-event pair around each launch on its own stream.  An EMPTY event pair on this stack measures 4.6-4.8 us
        # (msau_probe_overhead, below: `event_pair_us`); `avg_launch_us` is the raw reading, `avg_launch_us_net` = raw - empty pair is
        # what `achieved` / `frac` are computed from -- the figure that agrees with the average duration rocprofv3 --kernel-trace
        # reports for the same kernel in the same command (profiles/rNN_final_kernel_stats.csv; r04: 26.0 net vs 24.8 us).
        # Together with it the memory-bound norm ops north_star names: LRN forward / backward and the max pool, per channel width.
        norm_keys = [k for k in plan.launch_meta if k.startswith(("lrn_", "pool_"))]
        probed = probe_in_place(eng, plan, x, label, [key] + norm_keys, min(args.steps, 30))
        if key not in probed:                                # the dominant symbol runs on the side stream (a weight gradient)
            probed.update(probe_in_place(eng, plan, x, label, [key], min(args.steps, 30), side=True))
        n_probe, insitu_us, bytes_probe = probed[key]
        import ctypes
        ov = ctypes.c_float(0.0)
        L.call("msau_spin", torch.cuda.current_stream().cuda_stream, 2000)
        L.call("msau_probe_overhead", torch.cuda.current_stream().cuda_stream, 256, ctypes.byref(ov))
        # The subtraction is only trusted where it was validated against rocprofv3 (a ~26 us launch): for launches of at least four
        # empty pairs.  Shorter launches (the 6-10 us norm ops) keep the RAW reading as their headline figure -- subtracting there
        # would more than double the reported bandwidth on the strength of a 4.6 us constant.  The raw-basis numbers
        # (`achieved_raw`, `frac_raw`: the accounting of rounds 1-3) are always reported beside the net ones.
        net_us = lambda us: us - ov.value if us >= 4.0 * ov.value else us
        insitu_net = net_us(insitu_us)
        achieved = bytes_probe / (insitu_net * 1e-6) / 1e9
        achieved_raw = bytes_probe / (insitu_us * 1e-6) / 1e9
        tr, why = load_traffic(key)
        pm, pm_why = load_pmc(key)
        serial = {r[1]: 1e3 * r[0] / r[2] for r in rows}
        norm_ops = {}
        for k in sorted(norm_keys):
            if k in probed:
                c, us, nb = probed[k]
                norm_ops[k] = {"launches_per_step": c, "avg_launch_us": round(us, 2), "avg_launch_us_net": round(net_us(us), 2),
                               "alg_MB_per_launch": round(nb / 1e6, 2),
                               "GB/s": round(nb / (net_us(us) * 1e-6) / 1e9, 1), "frac": round(nb / (net_us(us) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                               "GB/s_raw": round(nb / (us * 1e-6) / 1e9, 1),
                               "avg_launch_us_serialised": round(serial.get(k, float("nan")), 2)}
                tk, _ = load_traffic(k)
                if tk:                                               # PMC bytes of the stand-alone pass (tools/make_traffic.py)
                    norm_ops[k]["traffic"] = tk["hbm_bytes_per_launch"]
                    norm_ops[k]["traffic_over_algorithmic"] = round(tk["hbm_bytes_per_launch"] / tk["alg_bytes_of_measured_launch"], 3)
        step_bytes = sum(m[1] for m in plan.launch_meta.values())
        roof = {"bound": "hbm", "kernel": key, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "achieved_raw": round(achieved_raw, 1), "frac_raw": round(achieved_raw / HBM_PEAK_GBS, 4),
                "basis": ("event-net: in-place HIP-event reading minus an empty event pair" if insitu_net != insitu_us else
                          "raw in-place HIP-event reading (launch shorter than four empty event pairs: nothing subtracted)"),
                "traffic": tr["hbm_bytes_per_launch"] if tr else None,
                "traffic_note": (f"PMC (FETCH_SIZE x2 + WRITE_SIZE, separate passes) of {tr['launch']}: {tr['hbm_bytes_per_launch']} HBM bytes for "
                                 f"{tr['alg_bytes_of_measured_launch']} algorithmic (x{tr['hbm_bytes_per_launch'] / tr['alg_bytes_of_measured_launch']:.2f}); "
                                 f"mangled name {tr.get('mangled', '?')}; tools/make_traffic.py at {tr.get('git_sha', '?')}") if tr else why,
                "launches_per_step": n_launch, "avg_launch_us": round(insitu_us, 2),
                "event_pair_us": round(ov.value, 2), "avg_launch_us_net": round(insitu_net, 2),
                "avg_launch_us_serialised": round(serial_us, 2), "timed_launches": n_probe * min(args.steps, 30),
                "alg_bytes_per_launch": round(bytes_probe),
                "share_of_step": round(ms / total_ms, 3),
                "mfma_frac": round((alg_flops / n_launch) / (insitu_net * 1e-6) / 1e12 / MFMA_BF16_PEAK_TF, 4),
                # from counters (tools/pmc_step.py, inside the step): SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32 SIMDs per shader
                # engine) and vector instructions issued per matrix instruction
                "mfma_busy": (pm or {}).get("mfma_busy"), "valu_per_mfma": (pm or {}).get("valu_per_mfma"),
                "issue_frac": (pm or {}).get("issue_frac"), "pmc_note": pm_why or "profiles/r05_pmc.json (tools/pmc_step.py)",
                "norm_ops": norm_ops,
                "launches_per_step_total": int(sum(m[0] for m in plan.launch_meta.values())),
                "whole_step": {"alg_GB": round(step_bytes / 1e9, 3),
                               "alg_TFLOP": round(sum(m[2] for m in plan.launch_meta.values()) / 1e12, 4),
                               "hbm_frac": round(step_bytes * value / (world * args.batch) / 1e9 / HBM_PEAK_GBS, 4)}}
        if (args.height, args.width, args.channels, args.stages) == (336, 256, 64, 3):
            # SURVEY.md 8(d) compulsory-traffic model of the whole step (tools/roofline.py re-derives it)
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
            import roofline as RF
            ach = RF.achieved(RF.CONFIGS["cfg2"], value / world, 2 if args.dtype == "bf16" else 4)
            roof["whole_step"]["compulsory_hbm_frac"] = round(ach["hbm_frac"], 4)
            roof["whole_step"]["compulsory_mfma_frac"] = round(ach["mfma_frac"], 4)

    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary:
        del eng, model
        torch.cuda.empty_cache()
        secondary = secondary_measurements(args, L, MSAUWrapper, TrainEngine, dev, n_class)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, n_class)

    if rank == 0:
        headline = (args.height, args.width, args.channels, args.stages) == (336, 256, 64, 3)
        shape = f"{args.height}x{args.width}x{args.channels} {args.stages}-stage MSAU"
        cfg = "configs[1]: one-hot" if headline else ("configs[3]: BERT-embedding" if args.channels == 768 else "non-headline")
        out = {"metric": f"chargrid tiles/sec (train fwd+bwd), {shape}", "value": round(value, 2),
               "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"{cfg} chargrid {args.height}x{args.width}x{args.channels}, "
                                      f"{args.stages}-stage MSAU (featRoot 8, 4 scales, res_depth 2), "
                                      f"batch {args.batch}/GPU, fwd+masked-CE+bwd+clip+Adam"
                                      + (", RCCL all-reduce" if world > 1 else ""),
                          "global_batch": world * args.batch, "parallelism": f"dp{world}",
                          "graph": bool(args.graph), "loss": round(loss_val, 5)},
               "roofline": roof, "cpu_baseline": cpu, "secondary": secondary}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if world > 1 or os.environ.get("MSAU_FORCE_DIST") == "1":
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
