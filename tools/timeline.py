#!/usr/bin/env python3
"""Per-step timeline from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py: for one steady-state step
(between two Adam launches) the busy time of every hardware queue, the main queue's time by kernel family, and the idle
gaps between its dependent launches (the launch-latency floor of DESIGN.md section 5).

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
    python tools/timeline.py out/*/*kernel_trace.csv [step index]
"""
import collections
import csv
import sys

FAMILIES = (("conv_lean", "conv_lean"), ("conv_kernel", "conv_generic"), ("wgrad", "wgrad"), ("unpack", "slab reduce"), ("attn", "attention"),
            ("lrn", "lrn"), ("pool", "pool"), ("nchw", "layout"), ("nhwc", "layout"), ("masked_ce", "loss"), ("label_count", "loss"),
            ("ordered_sum", "loss"), ("adam", "adam"), ("sqsum", "adam"), ("pack_kernel", "pack"), ("channel_sum", "channel_sum"))


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return name.split("(")[0][-32:]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"] and "prep" not in r["Kernel_Name"]]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) // 2
    seg = rows[adam[k] + 1: adam[k + 1] + 1]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    print(f"step {k}: {(t1 - t0) / 1e3:.0f} us, {len(seg)} kernels")
    queues = collections.defaultdict(list)
    for r in seg:
        queues[r["Queue_Id"]].append(r)
    for q, rs in sorted(queues.items(), key=lambda kv: -len(kv[1])):
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
        print(f"  queue {q}: {len(rs)} kernels, busy {busy / 1e3:.0f} us, from {(int(rs[0]['Start_Timestamp']) - t0) / 1e3:.0f} to "
              f"{(int(rs[-1]['End_Timestamp']) - t0) / 1e3:.0f} us")
    main_q = max(queues.values(), key=len)
    fam = collections.defaultdict(float)
    for r in main_q:
        fam[family(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gaps = [(int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3 for x, y in zip(main_q, main_q[1:])]
    print("  main queue by family (us):", {k_: round(v) for k_, v in sorted(fam.items(), key=lambda kv: -kv[1])})
    print(f"  main queue idle between launches: {sum(gaps):.0f} us over {len(gaps)} boundaries; largest {max(gaps):.0f} us")


if __name__ == "__main__":
    main()
