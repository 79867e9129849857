#!/usr/bin/env python3
"""Per-depth kernel durations and gaps of a bench.py run from a rocprofv3 kernel trace (CSV)."""
import collections
import csv
import sys


def main(path, last_steps=100):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seq = [(r["Kernel_Name"].split("(")[0][:34], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    idx = [i for i, s in enumerate(seq) if "masked_lsm" in s[0]]
    seq = seq[idx[-last_steps]:idx[-1] + 2]
    dur, gaps = collections.defaultdict(list), collections.defaultdict(list)
    depth, prev = None, None
    for n, s, e in seq:
        if "beam_init" in n:
            depth = 0
        key = n
        if depth is not None and ("masked_lsm" in n or "beam_step" in n):
            key = f"{n} d{depth}"
            if "beam_step" in n:
                depth += 1
        dur[key].append((e - s) / 1e3)
        if prev is not None:
            gaps[prev[0][:12] + "->" + n[:12]].append((s - prev[1]) / 1e3)
        prev = (n, e)
    total = (seq[-1][2] - seq[0][1]) / 1e3
    print(f"window {total:.1f} us over {last_steps} steps = {total / last_steps:.2f} us/step")
    for k in sorted(dur):
        v = dur[k]
        print(f"  {k:44s} n={len(v):4d} avg={sum(v) / len(v):8.2f} per-step={sum(v) / last_steps:7.2f}")
    for k, v in gaps.items():
        print(f"  gap {k:30s} n={len(v):4d} avg={sum(v) / len(v):6.2f} per-step={sum(v) / last_steps:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 100)
