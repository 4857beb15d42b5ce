"""Build libmsau_hip.so (gfx950) in-tree with hipcc.  `python -m msau_amd.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmsau_hip.so")
SOURCES = ["pack.hip", "conv.hip", "conv_lean.hip", "conv_first.hip", "conv_pair.hip", "conv_rows.hip", "conv_wgrad.hip", "wgrad_lean.hip", "elementwise.hip", "attention.hip", "attention_mfma.hip", "raster.hip", "boxconv.hip", "sequence.hip", "comm.hip", "ownerconv.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"] + os.environ.get("MSAU_EXTRA_HIPCC_FLAGS", "").split() + [
         "-ffp-contract=fast"]


# Per-source device flags.  elementwise.hip is compiled WITHOUT packed-fp32 instructions (v_pk_add_f32 / v_pk_mul_f32 /
# v_pk_fma_f32): with them, the level-0 LRN backward (lrn_fast_kernel<bf16,1,BWD>: chains of v_pk_add_f32 with op_sel, i.e.
# operands taken from the other half of a register pair) occasionally returned a wrong adjoint window sum for a 16-lane
# group when a weight-gradient kernel of the side stream shared the device -- same inputs, different bits, in 25-40 % of
# fresh processes.  Found 2026-10-04 with tools/det_check.py (interleaved A/B on one box: 9 of 24 processes deviated with
# packed ops, 0 of 24 without); neither the IEEE division, nor the library sqrt sequences, nor the transcendental unit
# (a VALU-only Newton variant deviated just the same) were the cause.  DESIGN.md section 2.  The host pass of hipcc prints
# "'-packed-fp32-ops' is not a recognized feature" for it: harmless, the flag is for the gfx950 pass.
EXTRA_FLAGS = {"elementwise.hip": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"],
               # MFMA results straight into VGPRs (gfx950 has one unified file): without it the row kernel's accumulators live in
               # AGPRs and every epilogue starts with four v_accvgpr_read
               # ... and, since round 3 put the LRN backward into the residual pair's epilogue (MSAU_PAIR_LRN_BWD: the same adjoint-window
               # chains as lrn_fast_kernel, beside the same side-stream kernels), WITHOUT packed-fp32 instructions like elementwise.hip:
               # the compiler had produced 31 op_sel'd v_pk_*_f32 in exactly those two instances (tests/test_host_cpu.py reads the ISA)
               # the box-list-fed first conv (cfg 4's bf16 train path, beside the same side-stream kernels): its fp32 partial-sum loops
               # compiled to 247 packed-fp32 instructions, 91 of them with op_sel_hi -- the pattern the rule above is about
               "ownerconv.hip": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"],
               # attention on the matrix cores: MFMA results in VGPRs as well (the statistics kernel read every score back with
               # v_accvgpr_read: a quarter of its vector instructions)
               "attention_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "conv_rows.hip": ["-std=c++20"] + (["-mllvm", "-amdgpu-mfma-vgpr-form"] if os.environ.get("MSAU_ROWS_VGPR_FORM", "1") != "0" else [])
                                + (["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"] if os.environ.get("MSAU_ROWS_PACKED_FP32", "0") != "1" else [])
                                + (["-DMSAU_ROWCONV_PF=" + os.environ["MSAU_ROWCONV_PF"]] if os.environ.get("MSAU_ROWCONV_PF") else [])}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # this file is a dependency too: it holds the compile flags
    headers = [os.path.join(CSRC, "msau_common.h"), os.path.join(HERE, "..", "include", "msau_hip.h"), os.path.abspath(__file__)]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        err = "\n".join(ln for ln in r.stderr.splitlines() if "is not a recognized feature for this target" not in ln)
        if err.strip():
            print(err, file=sys.stderr, flush=True)
        if r.returncode:
            raise subprocess.CalledProcessError(r.returncode, cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl", "-lpthread"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
