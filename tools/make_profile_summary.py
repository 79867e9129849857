#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of a benchmark run into the committed summaries under profiles/.

    python tools/make_profile_summary.py r1 gpurun_out/r1_bench.json gpurun_out/r1_prof/bench_kernel_stats.csv \
        gpurun_out/r1_pmc_fetch/pmc_counter_collection.csv gpurun_out/r1_pmc_write/pmc_counter_collection.csv

Inputs: the JSON line of `python bench.py`, the kernel stats of
`rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-processor-surface --no-settle`, and the counter
files of two separate passes `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` over
`python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-processor-surface --no-kernel-timing --no-settle`.
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = ("void masked_lsm", "beam_step", "plan_", "beam_init", "void sid_rew", "group_adv", "state_", "fill_rows",
        "hit_ranks", "group_norm")


def main():
    tag, bench_json, stats_csv, fetch_csv, write_csv = sys.argv[1:6]
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    with open(bench_json) as f:
        b = json.loads(f.read().strip().splitlines()[-1])
    with open(os.path.join(out_dir, f"{tag}_bench_line.json"), "w") as f:
        json.dump(b, f, indent=1)
    shutil.copy(stats_csv, os.path.join(out_dir, f"{tag}_bench_kernel_stats.csv"))

    pmc = {}
    for name, path in (("FETCH_SIZE", fetch_csv), ("WRITE_SIZE", write_csv)):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == name:
                per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        pmc[name] = {k: {"launches": len(v), "avg_kb": sum(v) / len(v), "min_kb": min(v), "max_kb": max(v)}
                     for k, v in per.items() if k.startswith(OURS)}
    key = [k for k in pmc["FETCH_SIZE"] if "masked_lsm" in k][0]
    fetch = pmc["FETCH_SIZE"][key]["avg_kb"] * 1024 * 2
    write = pmc["WRITE_SIZE"][key]["avg_kb"] * 1024
    alg = b["roofline"]["algorithmic_bytes_per_launch"]
    sys.path.insert(0, ROOT)
    import bench

    pmc["kernel_sources_sha256"] = bench.kernel_sources_sha256()
    pmc["masked_lsm_kernel_hbm_bytes_per_launch"] = {
        "read_corrected": fetch, "write": write, "total": fetch + write, "algorithmic": alg,
        "note": "FETCH_SIZE x 1024 x 2 (gfx950 reports exactly half the bytes of a 16-B-per-lane coalesced "
                "stream) + WRITE_SIZE x 1024; separate --pmc passes of: python3 bench.py --steps 10 --warmup 5 "
                "--no-cpu-baseline --no-processor-surface --no-kernel-timing --no-settle"}
    with open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w") as f:
        json.dump(pmc, f, indent=1)

    rows = list(csv.DictReader(open(stats_csv)))
    table = ["| kernel | calls | avg µs | total ms | % |", "|---|---|---|---|---|"]
    for r in rows[:16]:
        table.append(f"| `{r['Name'][:72]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | "
                     f"{float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['Percentage']):.2f} |")
    lsm = [r for r in rows if "masked_lsm" in r["Name"]][0]
    rf, cb, ps = b["roofline"], b.get("cpu_baseline"), b.get("processor_surface")
    lines = [
        f"# {tag} profile summary (1 x MI355X, ROCm 7.2)", "",
        "Command: `python bench.py` (defaults: --gpus 1 --steps 100 --warmup 20).",
        f"Workload: {b['config']['workload']}", "",
        f"* value: **{b['value']:.1f} steps/s**, {b['ms_per_step'] * 1e3:.1f} µs per step "
        f"({b['value'] / 3200 * 100:.1f} % of the 3200 steps/s a perfect 8 TB/s stream of the logits alone would give)",
        f"* dominant kernel `{rf['kernel']}`: {rf['kernel_ms'] * 1e3:.1f} µs per launch by HIP events on the dispatch "
        f"({rf['launches_timed']} launches) = **{rf['achieved']:.0f} GB/s = {rf['frac'] * 100:.1f} % of the 8 TB/s "
        f"HBM peak**; rocprofv3 kernel trace of the same command: {float(lsm['AverageNs']) / 1e3:.1f} µs average over "
        f"{lsm['Calls']} launches",
        f"* HBM traffic of that kernel from PMC (`{tag}_pmc_summary.json`): read {fetch / 1e9:.4f} GB (FETCH_SIZE, x2 "
        f"correction) + write {write / 1e6:.2f} MB per launch vs {alg / 1e9:.4f} GB algorithmic -> ratio "
        f"{(fetch + write) / alg:.4f}",
    ]
    if rf.get("read_probe"):
        rp = rf["read_probe"]
        lines.append(f"* read probe (the same rows read by a kernel that only loads them, same access pattern, "
                     f"{rp['launches']} launches): {rp['ms'] * 1e3:.1f} µs = {rp['GBps']:.0f} GB/s = "
                     f"{rp['GBps'] / 80:.1f} % of peak; the streaming kernel reaches "
                     f"{rf['frac_of_read_probe'] * 100:.1f} % of that")
    if rf.get("settle"):
        st = rf["settle"]
        lines.append(f"* before the warm-up the run waited {st['waited_s']} s for the GPU's read rate to settle "
                     f"(probe {st['probe_GBps_first']:.0f} -> {st['probe_GBps_last']:.0f} GB/s, target {st['target_GBps']:.0f})")
    if cb:
        lines.append(f"* cpu_baseline ({cb['kind']}, {cb['cores']} threads): {cb['value']:.3f} steps/s ({cb['sample']})"
                     + (f"; on 8 threads {cb['at_8_cores']['value']:.3f} steps/s ({cb['at_8_cores']['sample']})"
                        if cb.get("at_8_cores") else ""))
    if ps:
        for k, v in ps.items():
            lines.append(f"* processor surface {k} (dense [R, V] output): {v['ms_per_call'] * 1e3:.0f} µs per call"
                         + (f" = {v['achieved_GBps']:.0f} GB/s = {v['frac_of_peak'] * 100:.1f} % of peak on "
                            f"{v['algorithmic_bytes'] / 1e9:.1f} GB" if v.get("algorithmic_bytes") else ""))
    lines += ["", "`rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-processor-surface --no-settle` "
              f"(`{tag}_bench_kernel_stats.csv`; the `at::native` rows are the synthetic-logit generation before the "
              "timed region):", ""] + table + [""]
    with open(os.path.join(out_dir, f"{tag}_summary.md"), "w") as f:
        f.write("\n".join(lines))
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main()
