#!/usr/bin/env python3
"""SQ counters of every kernel of the training step, IN the step (round-4 verdict item 5: MFMA utilisation on the conv GEMMs from
counters, not from flops / time) -> profiles/r05_pmc_step.txt (table) + profiles/r05_pmc.json (read by bench.py).

`rocprofv3 --kernel-trace --pmc ...` passes of `bench.py --steps 3` (counters only: never with another trace domain), per kernel
symbol the mean per dispatch and

    mfma_busy     = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32)      SQ_BUSY_CYCLES is summed over the 32 shader engines and a
                    shader engine has 32 of the chip's 1024 SIMDs: the denominator is (busy cycles of the dispatch) x SIMDs
    valu_per_mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA                         vector instructions issued per matrix instruction
    issue / wait  = SQ_ACTIVE_INST_ANY, SQ_WAIT_ANY, SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES

    python tools/pmc_step.py            # on the GPU box; this script itself never touches the GPU
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

OUT = os.path.join(ROOT, "gpurun_out", "pmc_step")
PASSES = ["SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU",
          "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"]
# launch-kind key of the plan (what bench.py's roofline object names) -> substring of the kernel symbol
FAMILIES = {"rowpair_kernel<bf16,C8>": "rowpair_c8_kernel", "rowpair_kernel<bf16,C16>": "rowpair_c16_kernel",
            "conv_pair_kernel<bf16,C32>": "conv_pair_kernel", "rowconv_kernel": "rowconv8_kernel", "rowwgrad_kernel": "rowwgrad8_kernel",
            "conv_lean_kernel": "conv_lean_kernel", "wgrad_lean_kernel": "wgrad_lean_kernel", "conv_chunked_kernel": "conv_chunked_kernel",
            "selfattn": "attn_"}


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(.*)E+v", name)
    if m:                                                       # mangled template instance: kernel<digits / b0 b1 ...>
        args = re.findall(r"L[ib](n?\d+)E|(DF16b|f)", m.group(2))
        return m.group(1) + "<" + ",".join((a[0].replace("n", "-") or {"DF16b": "bf16", "f": "f32"}[a[1]]) for a in args) + ">"
    return name.split("(")[0][:90]


def main():
    os.makedirs(OUT, exist_ok=True)
    sha = os.environ.get("MSAU_GIT_SHA") or subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or "unknown"
    prog = ["python3", os.path.join(ROOT, "bench.py"), "--no-secondary", "--no-cpu-baseline", "--no-roofline", "--steps", "3", "--warmup", "2"]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for i, ctrs in enumerate(PASSES):
        d = os.path.join(OUT, f"pass{i}")
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + ctrs.split() + ["--output-format", "csv", "-d", d, "--"] + prog,
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            os.remove(f)

    def derived(c):
        g = lambda k: c.get(k, 0.0)
        out = {"dispatches": int(c.get("_n", 0)), "waves": round(g("SQ_WAVES")), "insts_valu": round(g("SQ_INSTS_VALU")), "insts_mfma": round(g("SQ_INSTS_MFMA")),
               "insts_lds": round(g("SQ_INSTS_LDS")), "insts_salu": round(g("SQ_INSTS_SALU"))}
        if g("SQ_BUSY_CYCLES") > 0:
            out["mfma_busy"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("SQ_BUSY_CYCLES") * 32), 4)
        if g("SQ_INSTS_MFMA") > 0:
            out["valu_per_mfma"] = round(g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA"), 2)
        if g("SQ_WAVE_CYCLES") > 0:
            out["issue_frac"] = round(g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), 3)
            out["wait_any_frac"] = round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 3)
            out["wait_inst_frac"] = round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 3)
        if g("SQ_LDS_IDX_ACTIVE") > 0:
            out["lds_conflict_frac"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 3)
        return out

    per = {}
    for name, cs in agg.items():
        c = {k: sum(v) / len(v) for k, v in cs.items()}
        c["_n"] = max(len(v) for v in cs.values())
        per[name] = (c, derived(c))
    # families: dispatch-weighted totals of the raw counters, then the same derived numbers
    fam = {}
    for key, sub in FAMILIES.items():
        tot = collections.defaultdict(float)
        names = [n for n in agg if sub in n]
        for n in names:
            for k, v in agg[n].items():
                tot[k] += sum(v)
        ndisp = sum(max(len(v) for v in agg[n].values()) for n in names)
        if ndisp:
            c = {k: v / ndisp for k, v in tot.items()}
            c["_n"] = ndisp
            fam[key] = dict(derived(c), symbols=len(names))
    lines = [f"# SQ counters per kernel symbol, mean per dispatch, inside bench.py's training step (cfg 2, B=16, bf16; 5 steps x 2 passes)",
             f"# tools/pmc_step.py at {sha}, kernel sources {bench.kernel_source_hash()};  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32)",
             f"# {'kernel':74s} {'disp':>5s} {'waves':>7s} {'VALU':>9s} {'MFMA':>8s} {'V/M':>6s} {'mfma_busy':>9s} {'issue':>6s} {'wait':>6s} {'ldsconf':>7s}"]
    for name, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][0].get("SQ_BUSY_CYCLES", 0) * kv[1][0]["_n"]):
        if "at::native" in name or "spin_kernel" in name:
            continue
        lines.append(f"  {short(name)[:74]:74s} {d['dispatches']:5d} {d['waves']:7d} {d['insts_valu']:9d} {d['insts_mfma']:8d} {d.get('valu_per_mfma', 0):6.1f} "
                     f"{d.get('mfma_busy', 0):9.4f} {d.get('issue_frac', 0):6.3f} {d.get('wait_any_frac', 0):6.3f} {d.get('lds_conflict_frac', 0):7.3f}")
    lines.append("# families (all instances of a kernel template):")
    for key, d in fam.items():
        lines.append(f"  {key:40s} symbols {d['symbols']:3d} disp {d['dispatches']:5d}  V/M {d.get('valu_per_mfma', 0):6.1f}  mfma_busy {d.get('mfma_busy', 0):.4f}  "
                     f"issue {d.get('issue_frac', 0):.3f}  wait {d.get('wait_any_frac', 0):.3f}")
    txt = "\n".join(lines) + "\n"
    out = {"kernel_source_hash": bench.kernel_source_hash(), "git_sha": sha,
           "how": "tools/pmc_step.py: rocprofv3 --kernel-trace --pmc (SQ counters, two passes) of bench.py --steps 3; per dispatch means; "
                  "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32)",
           "families": fam}
    for base in (os.path.join(ROOT, "profiles"), OUT):
        with open(os.path.join(base, "r05_pmc_step.txt"), "w") as f:
            f.write(txt)
        with open(os.path.join(base, "r05_pmc.json"), "w") as f:
            json.dump(out, f, indent=1)
    print(txt)


if __name__ == "__main__":
    main()
