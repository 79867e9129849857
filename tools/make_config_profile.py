#!/usr/bin/env python3
"""profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc.json from what tools/collect_config_profiles.sh left
under gpurun_out/<tag>/ (a rocprofv3 kernel trace with --stats and one counter pass each for FETCH_SIZE and
WRITE_SIZE over the same bench.py configuration).

    python tools/make_config_profile.py <tag> <the bench.py arguments of the run...>
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter(path, name):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "avg_kb": sum(v) / len(v), "min_kb": min(v), "max_kb": max(v)}
            for k, v in per.items() if "at::native" not in k and "rocprim" not in k}


def main():
    tag, bench_args = sys.argv[1], sys.argv[2:]
    src = os.path.join(ROOT, "gpurun_out", tag)
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    with open(os.path.join(src, "line.json")) as f:
        line = json.loads([ln for ln in f.read().splitlines() if ln.startswith("{")][-1])
    fetch = counter(glob.glob(os.path.join(src, "pmc_fetch", "**", "*counter_collection.csv"), recursive=True)[0],
                    "FETCH_SIZE")
    write = counter(glob.glob(os.path.join(src, "pmc_write", "**", "*counter_collection.csv"), recursive=True)[0],
                    "WRITE_SIZE")
    rows = list(csv.DictReader(open(stats)))
    dominant = max((r for r in rows if "at::native" not in r["Name"] and "rocprim" not in r["Name"]),
                   key=lambda r: float(r["TotalDurationNs"]))
    name = dominant["Name"]
    key = [k for k in fetch if k == name or k.startswith(name.split("(")[0])][0]
    alg = line["roofline"]["algorithmic_bytes_per_launch"]
    rd = fetch[key]["avg_kb"] * 1024 * 2
    wr = write[key]["avg_kb"] * 1024
    sys.path.insert(0, ROOT)
    import bench

    key = None
    try:  # the key bench.py looks a recorded figure up by
        import argparse

        ap = argparse.ArgumentParser()
        ap.add_argument("--trie", default="Instruments")
        ap.add_argument("--processor", default="count")
        ap.add_argument("--batch", type=int, default=512)
        ap.add_argument("--beams", type=int, default=16)
        ap.add_argument("--dtype", default="bf16")
        ap.add_argument("--prompt-len", type=int, default=61)
        ap.add_argument("--repetition-penalty", type=float, default=1.0)
        known, _ = ap.parse_known_args(bench_args)
        key = bench.workload_key(known)
    except SystemExit:
        pass
    out = {
        "bench_args": bench_args,
        "workload_key": key,
        "workload": line["config"]["workload"],
        "value_steps_per_s": line["value"],
        "ms_per_step": line["ms_per_step"],
        "dominant_kernel": name,
        "rocprofv3_avg_us": float(dominant["AverageNs"]) / 1e3,
        "rocprofv3_calls": int(dominant["Calls"]),
        "hip_event_avg_us": line["roofline"]["kernel_ms"] * 1e3,
        "algorithmic_bytes_per_launch": alg,
        "frac_of_peak_rocprofv3": alg / (float(dominant["AverageNs"]) * 1e-9) / 8e12,
        "hbm_bytes_per_launch": {"read_corrected": rd, "write": wr, "total": rd + wr, "ratio_to_algorithmic": (rd + wr) / alg,
                                 "note": "FETCH_SIZE x 1024 x 2 (gfx950 counts half the bytes of a 16-B-per-lane "
                                         "coalesced stream) + WRITE_SIZE x 1024, separate --pmc passes, averages over the "
                                         "launches of the dominant kernel"},
        "FETCH_SIZE": fetch, "WRITE_SIZE": write,
        "kernel_sources_sha256": bench.kernel_sources_sha256(),
    }
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(f"{tag}: {line['value']:.1f} steps/s; {name[:60]} {out['rocprofv3_avg_us']:.1f} us (rocprofv3, "
          f"{out['rocprofv3_calls']} launches) / {out['hip_event_avg_us']:.1f} us (events) = "
          f"{out['frac_of_peak_rocprofv3'] * 100:.1f} % of peak; traffic {out['hbm_bytes_per_launch']['ratio_to_algorithmic']:.4f} x algorithmic")


if __name__ == "__main__":
    main()
