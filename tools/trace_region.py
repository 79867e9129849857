#!/usr/bin/env python3
"""Kernel-by-kernel view of bench.py's timed region from a rocprofv3 kernel trace: duration of every kernel of the
last N steps and the idle gap in front of it.  Usage: trace_region.py <kernel_trace.csv> [steps]"""
import csv
import sys


def main(path, steps):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ks = [(r["Kernel_Name"][:30], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    idx = [i for i, k in enumerate(ks) if "masked_lsm" in k[0]]
    start = idx[-steps]
    while start > 0 and not any(s in ks[start - 1][0] for s in ("masked_lsm", "beam_step", "read_probe")):
        start -= 1
    prev_end, t0 = None, ks[start][1]
    busy = 0
    for i in range(start, min(len(ks), idx[-1] + 4)):
        n, s, e = ks[i]
        gap = (s - prev_end) / 1000 if prev_end else 0.0
        busy += e - s
        print(f"{(s - t0) / 1000:9.1f}  {n:30s} dur {(e - s) / 1000:8.1f} us  gap {gap:7.1f}")
        prev_end = e
    print(f"span {(prev_end - t0) / 1000:.1f} us, kernels {busy / 1000:.1f} us, {steps} steps: {(prev_end - t0) / 1000 / steps:.1f} us per step on the GPU")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20)
