#!/usr/bin/env python3
"""Compare two rocprofv3 kernel_stats.csv files per kernel symbol: tools/cmp_stats.py A.csv B.csv STEPS [min_us]
(us per step = TotalDurationNs / STEPS; rows whose per-step time moved by >= min_us, default 2)."""
import csv
import re
import sys


def load(f, steps):
    out = {}
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        n = re.sub(r"\(.*\)$", "", n)[:100]
        c, t = out.get(n, (0, 0.0))
        out[n] = (c + int(r["Calls"]) / steps, t + float(r["TotalDurationNs"]) / steps / 1e3)
    return out


a, b = load(sys.argv[1], float(sys.argv[3])), load(sys.argv[2], float(sys.argv[3]))
thr = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
rows = []
for k in sorted(set(a) | set(b)):
    ca, ta = a.get(k, (0, 0.0))
    cb, tb = b.get(k, (0, 0.0))
    if "spin" in k:
        continue
    if abs(tb - ta) >= thr:
        rows.append((tb - ta, k, ca, ta, cb, tb))
for d, k, ca, ta, cb, tb in sorted(rows):
    print(f"{d:+8.1f} us/step  {ca:5.1f} x {ta / max(ca, 1e-9):6.1f} -> {cb:5.1f} x {tb / max(cb, 1e-9):6.1f}   {k}")
print(f"total {sum(v[1] for k, v in a.items() if 'spin' not in k):.0f} -> {sum(v[1] for k, v in b.items() if 'spin' not in k):.0f} us/step of kernel time; "
      f"launches {sum(v[0] for k, v in a.items() if 'spin' not in k):.0f} -> {sum(v[0] for k, v in b.items() if 'spin' not in k):.0f}")
