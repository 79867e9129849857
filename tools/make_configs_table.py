#!/usr/bin/env python3
"""Markdown table of the bench lines that tools/bench_configs.sh wrote (one JSON line per configuration)."""
import json
import sys

LABELS = ["C1: I&S (3-level id2sid index), 1024 rows, fp32", "C3: Games, fixed-hint processor, 32,768 rows, bf16",
          "C4: codebook 32, 65,536 rows, bf16", "C4: codebook 512, 65,536 rows, bf16",
          "evaluation-style: Arts, 50 beams x 160 prompts = 8,000 rows, bf16",
          "headline shape with fp32 logits (8192 rows)"]


def main(path):
    print("| configuration | steps/s | µs per step | stream kernel µs | GB/s | % of 8 TB/s | µs per step outside the stream kernel |")
    print("|---|---|---|---|---|---|---|")
    for label, line in zip(LABELS, open(path)):
        d = json.loads(line)
        r = d["roofline"]
        step, k = d["ms_per_step"] * 1e3, r["kernel_ms"] * 1e3
        print(f"| {label} | {d['value']:.1f} | {step:.1f} | {k:.1f} | {r['achieved']:.0f} | {r['frac'] * 100:.1f} | {step - k:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
