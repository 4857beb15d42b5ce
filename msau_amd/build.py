"""Build libmsau_hip.so (gfx950) in-tree with hipcc.  `python -m msau_amd.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmsau_hip.so")
SOURCES = ["pack.hip", "conv.hip", "conv_lean.hip", "conv_first.hip", "conv_pair.hip", "conv_rows.hip", "conv_wgrad.hip", "wgrad_lean.hip", "elementwise.hip", "attention.hip", "attention_mfma.hip", "pointwise.hip", "raster.hip", "boxconv.hip", "sequence.hip", "comm.hip", "ownerconv.hip"]
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
               # the generic conv kernel: its ELU epilogue (expm1f, the (y + 1) derivative factor; round 5) compiled to 210 op_sel'd
               # packed-fp32 instructions -- the same rule; the generic kernel takes a handful of level-3 launches of the bf16 step
               "conv.hip": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"],
               # the box variant's kernels (cfg 5) run in a bf16 train path beside the same side-stream weight gradients: same rule
               "boxconv.hip": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"],
               # attention on the matrix cores: MFMA results in VGPRs as well (the statistics kernel read every score back with
               # v_accvgpr_read: a quarter of its vector instructions)
               "attention_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "pointwise.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"],
               "conv_rows.hip": ["-std=c++20"] + (["-mllvm", "-amdgpu-mfma-vgpr-form"] if os.environ.get("MSAU_ROWS_VGPR_FORM", "1") != "0" else [])
                                + (["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"] if os.environ.get("MSAU_ROWS_PACKED_FP32", "0") != "1" else [])
                                + (["-DMSAU_ROWCONV_PF=" + os.environ["MSAU_ROWCONV_PF"]] if os.environ.get("MSAU_ROWCONV_PF") else [])}


def source_hash() -> str:
    """sha256 over the HIP sources, their headers and the ABI header (16 hex digits).  build() stamps it into the library
    (msau_source_hash()), msau_amd/_lib.py::load() compares it with the sources it finds beside the library, bench.py ties its
    PMC files to it: nothing but this ties a shipped libmsau_hip.so to a tree (file times do not survive a snapshot)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))
                    + [os.path.join(HERE, "..", "include", "msau_hip.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


STAMP_MARK = b"MSAU_SRC_HASH="


def stamped_hash(lib: str = LIB):
    """the hash a built library carries, read from its bytes (no dlopen); None for a library without a stamp"""
    try:
        blob = open(lib, "rb").read()
    except OSError:
        return None
    i = blob.find(STAMP_MARK)
    return blob[i + len(STAMP_MARK):i + len(STAMP_MARK) + 16].decode("ascii", "replace") if i >= 0 else None


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # this file is a dependency too: it holds the compile flags
    headers = [os.path.join(CSRC, "msau_common.h"), os.path.join(HERE, "..", "include", "msau_hip.h"), os.path.abspath(__file__)]
    want = source_hash()
    if os.path.exists(LIB) and stamped_hash() != want:
        force = True                     # a library of other sources (or of no known ones), whatever the file times say
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
        # the stamp: a host-only object generated at link time, never a file of the tree
        stamp_o = os.path.join(CSRC, "stamp.o")
        code = ('extern "C" __attribute__((visibility("default"))) const char* msau_source_hash(void) '
                '{ static const char s[] = "%s%s"; return s + %d; }\n' % (STAMP_MARK.decode(), want, len(STAMP_MARK)))
        if verbose:
            print(f"g++ -c <stamp {want}> -o {stamp_o}", flush=True)
        subprocess.run(["g++", "-O1", "-fPIC", "-x", "c++", "-c", "-", "-o", stamp_o], input=code, text=True, check=True)
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, stamp_o, "-ldl", "-lpthread"])
        assert stamped_hash() == want, (stamped_hash(), want)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
