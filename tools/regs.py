#!/usr/bin/env python3
"""Register / LDS footprint of every kernel in a built object: tools/regs.py msau_amd/csrc/conv_rows.o [name-filter]
(reads the gfx950 code object's metadata note with llvm-readelf; no GPU needed)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(obj):
    with tempfile.TemporaryDirectory() as tmp:
        import shutil
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)                          # (the extracted images are written beside the object)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dev = [f for f in os.listdir(tmp) if "gfx950" in f]
        assert dev, os.listdir(tmp)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(tmp, dev[0])], check=True, capture_output=True, text=True).stdout
    out, cur = [], {}
    for ln in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):
            out.append(cur)
            cur = {}
        if k in ("name", "vgpr_count", "sgpr_count", "agpr_count", "group_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count",
                 "private_segment_fixed_size", "max_flat_workgroup_size"):
            cur[k] = v
    if cur.get("name"):
        out.append(cur)
    return out


def demangle(n):
    try:
        return subprocess.run([f"{LLVM}/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in kernels(sys.argv[1]):
        name = demangle(k.get("name", "?"))
        if flt and flt not in name:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        print(f"vgpr {k.get('vgpr_count', '?'):>4} agpr {k.get('agpr_count', '?'):>4} sgpr {k.get('sgpr_count', '?'):>4} "
              f"spill {k.get('vgpr_spill_count', '?'):>3} lds {k.get('group_segment_fixed_size', '?'):>6} scratch {k.get('private_segment_fixed_size', '?'):>4}  {name[:150]}")
