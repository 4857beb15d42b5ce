"""Timeline view of a rocprofv3 kernel trace (rocpd sqlite): for the steady-state steps, per HIP queue busy time and
idle gaps, and the per-step critical list of the main queue (kernel, duration, gap before it).
    python tools/timeline.py gpurun_out/x/prof/run_results.db [--step-kernel clip_adam] [--list]"""
import argparse
import collections
import re
import sqlite3
import subprocess


_DEM = {}


def short(name):
    if name not in _DEM:
        d = name
        if name.startswith("_Z"):
            d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        d = d.replace("(anonymous namespace)::", "").replace("__hip_bfloat16", "bf16").replace("__bf16", "bf16")
        d = re.sub(r"^void ", "", d)
        d = re.sub(r"\(.*$", "", d)
        d = d.replace("false", "0").replace("true", "1")
        _DEM[name] = d[:90]
    return _DEM[name]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--step-kernel", default="adam_kernel(")
    ap.add_argument("--nsteps", type=int, default=30)
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--skip", type=int, default=10)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute("select name, queue_id, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if a.step_kernel in r[0]]
    marks = marks[a.skip:a.skip + a.nsteps + 1]
    nsteps = len(marks) - 1
    lo, hi = marks[0], marks[-1]
    sel = rows[lo + 1:hi + 1]
    t0, t1 = rows[lo][3], rows[hi][3]
    print(f"{nsteps} steps, {1e-6 * (t1 - t0) / nsteps:.3f} ms/step, {len(sel) / nsteps:.1f} launches/step")
    byq = collections.defaultdict(list)
    for r in sel:
        byq[r[1]].append(r)
    for q, rs in sorted(byq.items()):
        busy = sum(r[3] - r[2] for r in rs)
        gaps = [rs[i + 1][2] - rs[i][3] for i in range(len(rs) - 1)]
        pos = [g for g in gaps if g > 0]
        print(f"queue {q}: {len(rs) / nsteps:.1f} launches/step, busy {1e-6 * busy / nsteps:.3f} ms/step, "
              f"gaps {1e-6 * sum(pos) / nsteps:.3f} ms/step (median {sorted(pos)[len(pos) // 2] / 1e3:.2f} us)")
        agg = collections.defaultdict(lambda: [0, 0, 0])
        prev_end = None
        for r in rs:
            k = agg[short(r[0])]
            k[0] += 1; k[1] += r[3] - r[2]
            if prev_end is not None:
                k[2] += max(0, r[2] - prev_end)
            prev_end = r[3]
        print(f"  {'kernel':90s} {'n/step':>7s} {'avg us':>8s} {'us/step':>8s} {'gap-before us/step':>10s}")
        for name, (n, t, g) in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
            print(f"  {name:90s} {n / nsteps:7.1f} {t / n / 1e3:8.2f} {t / nsteps / 1e3:8.1f} {g / nsteps / 1e3:10.1f}")
    if a.list:
        s0 = rows[marks[-2]][3]
        step = rows[marks[-2] + 1:marks[-1] + 1]
        ends = {}
        for r in step:
            gap = (r[2] - ends[r[1]]) / 1e3 if r[1] in ends else 0.0
            ends[r[1]] = r[3]
            print(f"q{r[1]} t={1e-3 * (r[2] - s0):9.1f} dur={1e-3 * (r[3] - r[2]):7.2f} gap={gap:7.2f}  {short(r[0])}")


if __name__ == "__main__":
    main()
