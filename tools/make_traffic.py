#!/usr/bin/env python3
"""HBM traffic per launch of the dominant kernels from PMC counters -> profiles/r05_traffic.json (read by bench.py).

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
import re
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
        res.setdefault("_rows", {})[ctr] = [r for r in rows if r["Counter_Name"] == ctr]
    return res


def entry(res, name, launch, alg_bytes):
    f = res["FETCH_SIZE"][name]
    w = res["WRITE_SIZE"][name]
    fetch_kb, write_kb = sum(f) / len(f), sum(w) / len(w)
    return {"launch": launch, "mangled": name, "FETCH_SIZE_KB": round(fetch_kb, 1), "WRITE_SIZE_KB": round(write_kb, 1),
            "hbm_bytes_per_launch": int(round(2 * fetch_kb * 1024 + write_kb * 1024)), "alg_bytes_of_measured_launch": int(alg_bytes),
            "dispatches_averaged": len(f)}


def avg_entry(res, pick, launch, alg_bytes):
    """`pick`: kernel names (one launch kind may be several dispatches); their counters are summed per launch"""
    fetch_kb = sum(sum(res["FETCH_SIZE"][n]) / len(res["FETCH_SIZE"][n]) for n in pick)
    write_kb = sum(sum(res["WRITE_SIZE"][n]) / len(res["WRITE_SIZE"][n]) for n in pick)
    return {"launch": launch, "mangled": " | ".join(pick), "FETCH_SIZE_KB": round(fetch_kb, 1), "WRITE_SIZE_KB": round(write_kb, 1),
            "hbm_bytes_per_launch": int(round(2 * fetch_kb * 1024 + write_kb * 1024)), "alg_bytes_of_measured_launch": int(alg_bytes),
            "dispatches_averaged": len(res["FETCH_SIZE"][pick[0]])}


def main():
    os.makedirs(OUT, exist_ok=True)
    # (the GPU box has no .git: the caller passes the commit in, `MSAU_GIT_SHA=$(git rev-parse --short HEAD)` inside the gpurun command)
    sha = os.environ.get("MSAU_GIT_SHA") or subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or "unknown"
    n8 = B * H * W * 8 * 2                                       # one 8-channel bf16 tensor at level 0
    kernels = {}
    tools = os.path.join(ROOT, "tools")
    # ---- the residual pair at level 0 / 1 (row-streaming kernels): forward launch and data-gradient launch
    for c, tag in ((8, "C8"), (16, "C16")):
        res = passes(f"pair{c}", ["python3", os.path.join(tools, "pair_bench.py"), "3", str(c)])
        def targs(n):
            """template arguments of a rowpair kernel name, mangled or demangled, as a tuple of 0/1"""
            m = re.search(r"rowpair_c\d+_kernelI((?:Lb[01]E)+)E", n)
            if m:
                return tuple(int(x) for x in re.findall(r"Lb([01])E", m.group(1)))
            m = re.search(r"rowpair_c\d+_kernel<([^>]*)>", n)
            return tuple(1 if t.strip() in ("true", "1") else 0 for t in m.group(1).split(",")) if m else ()
        names = [n for n in res["FETCH_SIZE"] if f"rowpair_c{c}_kernel" in n]
        fw = [n for n in names if targs(n)[0] == 0]
        bw = [n for n in names if targs(n)[0] == 1 and not any(targs(n)[2:] if c == 8 else ())]     # the plain data-gradient launch
        assert len(fw) == 1 and len(bw) == 1, (fw, bw, names)
        nt = n8 if c == 8 else n8 // 2                          # 16 channels at half the resolution: half the bytes
        planes = 2 * B * (H if c == 8 else H // 2) * -(-(W if c == 8 else W // 2) // (30 if c == 8 else 14)) * 32
        fwd = avg_entry(res, fw, f"residual pair forward (x0 -> r1, out, two ballot planes), B=16 C={c}", 3 * nt + planes)
        bwd = avg_entry(res, bw, f"residual pair data gradient (g + two ballot planes -> g_r1, g_x0), B=16 C={c}", 3 * nt + planes)
        ent = {"forward": fwd, "backward": bwd, "git_sha": sha}
        parts = [(fwd, 6), (bwd, 6)]
        if c == 8:
            # the launches the step actually runs at 8 channels: the first conv's weight gradient rides on every data-gradient launch
            # (x0 read, the intermediate gradient not written, one slab per workgroup), the LRN backward on the encoder block's (a read,
            # da written instead of dx)
            slabs = 768 * 8 * 80 * 4
            wg1 = [n for n in names if targs(n) == (1, 1, 0, 1)]
            both = [n for n in names if targs(n) == (1, 1, 1, 1)]
            if len(wg1) == 1 and len(both) == 1:
                ent["backward_wgrad1"] = avg_entry(res, wg1, "data gradient + first conv's weight gradient (g, x0, planes -> g_x0, slabs), B=16 C=8", 3 * nt + planes + slabs)
                ent["backward_lrn_wgrad1"] = avg_entry(res, both, "data gradient + LRN backward + first conv's weight gradient (g, x0, a, planes -> da, slabs), B=16 C=8", 4 * nt + planes + slabs)
                parts = [(fwd, 6), (ent["backward_wgrad1"], 3), (ent["backward_lrn_wgrad1"], 3)]
        tot = sum(k for _, k in parts)
        ent.update({"launch": f"mean over the step's {tot} launches ({', '.join(str(k) + ' x ' + e['launch'].split(' (')[0] for e, k in parts)}), B=16 C={c}",
                    "mangled": " | ".join(e["mangled"] for e, _ in parts),
                    "hbm_bytes_per_launch": sum(e["hbm_bytes_per_launch"] * k for e, k in parts) // tot,
                    "alg_bytes_of_measured_launch": sum(e["alg_bytes_of_measured_launch"] * k for e, k in parts) // tot})
        kernels[f"rowpair_kernel<bf16,{tag}>"] = ent
    # ---- 8 -> 8 3x3 at level 0: forward (row-streaming), data gradient (tile kernel), weight gradient
    res = passes("l0", ["python3", os.path.join(tools, "kbench.py"), "--only", "L0 8->8", "--iters", "3"])
    names = list(res["FETCH_SIZE"])
    rc = [n for n in names if "rowconv8_kernel" in n]
    wg = [n for n in names if "rowwgrad8_kernel" in n] or [n for n in names if "wgrad_lean_kernel" in n]
    if rc:
        e = avg_entry(res, rc[:1], "rowconv8 plain forward 8 -> 8 3x3 (bias only), B=16 336x256", 2 * n8)
        e["git_sha"] = sha
        kernels["rowconv_kernel<bf16,CIN8,CO8,K3>"] = e
    if wg:
        e = avg_entry(res, wg[:1], "weight gradient 8 -> 8 3x3 (x, g read once; slabs written), B=16 336x256", 2 * n8)
        e["git_sha"] = sha
        kernels["rowwgrad_kernel<bf16,C8,CO8,K3>" if "rowwgrad8_kernel" in wg[0] else "wgrad_lean_kernel<bf16,C8,CO8,K3>"] = e
    # ---- the normalisation / pooling / boundary passes (tools/norm_bench.py: one launch of each per iteration, fixed order)
    sys.path.insert(0, tools)
    res = passes("norm", ["python3", os.path.join(tools, "norm_bench.py"), "3"])
    fams = {"lrn_fwd": ("lrn_fast_kernel", False), "lrn_bwd": ("lrn_fast_kernel", True), "pool_fwd": ("pool_fwd_kernel", None),
            "pool_bwd": ("pool_bwd_kernel", None), "msau_nchw_to_nhwc": ("nchw_to_nhwc", None)}
    # per-dispatch rows in launch order: norm_bench launches (3 warm-up + 3 timed) x each op, op after op
    order = [("lrn_fwd<bf16,C8>", 2 * n8), ("lrn_bwd<bf16,C8>", 3 * n8), ("pool_fwd<bf16,C8>", n8 + n8 // 4 + n8 // 8), ("pool_bwd<bf16,C8>", n8 + n8 // 4 + n8 // 8),
             ("lrn_fwd<bf16,C16>", n8), ("lrn_bwd<bf16,C16>", 3 * n8 // 2), ("pool_fwd<bf16,C16>", n8 // 2 + n8 // 8 + n8 // 16),
             ("pool_bwd<bf16,C16>", n8 // 2 + n8 // 8 + n8 // 16), ("msau_nchw_to_nhwc", B * 64 * H * W * 6)]
    seq = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        flat = [(int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])) for r in res["_rows"][ctr]]
        flat.sort()
        flat = [f for f in flat if any(k in f[1] for k in ("lrn_fast_kernel", "pool_fwd_kernel", "pool_bwd_kernel", "nchw_to_nhwc"))]
        assert len(flat) == 6 * len(order), (len(flat), len(order))
        for i, (key, _) in enumerate(order):
            vals = flat[6 * i + 3:6 * i + 6]                     # the three timed launches of op i
            seq.setdefault(key, {})[ctr] = (sum(v[2] for v in vals) / len(vals), vals[0][1])
    for key, alg in order:
        f_kb, name = seq[key]["FETCH_SIZE"]
        w_kb, _ = seq[key]["WRITE_SIZE"]
        kernels[key] = {"launch": key + " stand-alone pass at its cfg-2 shape, B=16", "mangled": name, "FETCH_SIZE_KB": round(f_kb, 1),
                        "WRITE_SIZE_KB": round(w_kb, 1), "hbm_bytes_per_launch": int(round(2 * f_kb * 1024 + w_kb * 1024)),
                        "alg_bytes_of_measured_launch": int(alg), "dispatches_averaged": 3, "git_sha": sha}
    out = {"kernel_source_hash": bench.kernel_source_hash(), "git_sha": sha,
           "how": "tools/make_traffic.py: rocprofv3 --kernel-trace --pmc, one counter per pass; FETCH_SIZE x2 (gfx950), WRITE_SIZE as is; KB = 1024 B",
           "kernels": kernels}
    # profiles/ is the tracked copy; gpurun only brings gpurun_out/ back from the GPU box
    for path in (os.path.join(ROOT, "profiles", "r05_traffic.json"), os.path.join(OUT, "r05_traffic.json")):
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
