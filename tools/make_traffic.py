#!/usr/bin/env python3
"""HBM traffic per launch of the dominant kernels from PMC counters -> profiles/r02_traffic.json (read by bench.py).

Collected exactly as MI355X_MICROARCH.md prescribes: `rocprofv3 --kernel-trace --pmc <one counter>` in SEPARATE passes
(never combined with another trace domain), FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of wide coalesced
reads at 64 B), WRITE_SIZE as is; both are reported in KB.  The file carries the hash of the kernel sources it was measured
on: bench.py ignores it (traffic = null, with the reason) when the tree's kernels differ.

    python tools/make_traffic.py            # on the GPU box; this script itself never touches the GPU
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                    # kernel_source_hash (no GPU use at import)

OUT = os.path.join(ROOT, "gpurun_out", "traffic")
B, H, W = 16, 336, 256


def passes(tag, prog):
    res = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(OUT, f"{tag}_{ctr}")
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--"] + prog,
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"))
        rows = []
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            rows += list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        per = collections.OrderedDict()
        for r in rows:
            if r["Counter_Name"] == ctr:
                per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
        res[ctr] = per
    return res


def entry(res, name, launch, alg_bytes):
    f = res["FETCH_SIZE"][name]
    w = res["WRITE_SIZE"][name]
    fetch_kb, write_kb = sum(f) / len(f), sum(w) / len(w)
    return {"launch": launch, "mangled": name, "FETCH_SIZE_KB": round(fetch_kb, 1), "WRITE_SIZE_KB": round(write_kb, 1),
            "hbm_bytes_per_launch": int(round(2 * fetch_kb * 1024 + write_kb * 1024)), "alg_bytes_of_measured_launch": int(alg_bytes),
            "dispatches_averaged": len(f)}


def main():
    os.makedirs(OUT, exist_ok=True)
    sha = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or "unknown"
    n8 = B * H * W * 8 * 2                                       # one 8-channel bf16 tensor at level 0
    kernels = {}
    # ---- the fused residual pair at level 0: forward launch and data-gradient launch
    res = passes("pair", ["python3", os.path.join(ROOT, "tools", "pair_bench.py"), "3", "8"])
    names = [n for n in res["FETCH_SIZE"] if "conv_pair_kernel" in n]
    assert len(names) == 2, names
    fwd = entry(res, names[0], "conv_pair forward (x0 -> r1, out, two mask bit planes), B=16 336x256 C=8", 3 * n8 + 2 * (n8 // 16))
    bwd = entry(res, names[1], "conv_pair data gradient (g + two mask bit planes -> g_r1, g_x0), B=16 336x256 C=8", 3 * n8 + 2 * (n8 // 16))
    both = {"launch": "mean of the forward and the data-gradient launch (12 each per step), B=16 336x256 C=8",
            "mangled": names[0] + " | " + names[1],
            "hbm_bytes_per_launch": (fwd["hbm_bytes_per_launch"] + bwd["hbm_bytes_per_launch"]) // 2,
            "alg_bytes_of_measured_launch": (fwd["alg_bytes_of_measured_launch"] + bwd["alg_bytes_of_measured_launch"]) // 2,
            "forward": fwd, "backward": bwd, "git_sha": sha}
    kernels["conv_pair_kernel<bf16,C8>"] = both
    # ---- the plain 8 -> 8 3x3 launch of conv_lean (the round-1 dominant symbol), forward
    res = passes("lean", ["python3", os.path.join(ROOT, "tools", "kbench.py"), "--only", "L0 8->8", "--iters", "3"])
    names = [n for n in res["FETCH_SIZE"] if "conv_lean_kernel" in n]
    e = entry(res, names[0], "conv_lean plain forward 8 -> 8 3x3 (bias only), B=16 336x256", 2 * n8)
    e["git_sha"] = sha
    kernels["conv_lean_kernel<bf16,CIN8,CT1,K3>"] = e
    out = {"kernel_source_hash": bench.kernel_source_hash(), "git_sha": sha,
           "how": "tools/make_traffic.py: rocprofv3 --kernel-trace --pmc, one counter per pass; FETCH_SIZE x2 (gfx950), WRITE_SIZE as is; KB = 1024 B",
           "kernels": kernels}
    # profiles/ is the tracked copy; gpurun only brings gpurun_out/ back from the GPU box
    for path in (os.path.join(ROOT, "profiles", "r02_traffic.json"), os.path.join(OUT, "r02_traffic.json")):
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
