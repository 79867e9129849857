#!/usr/bin/env python3
"""Host time against GPU time of one bench step for a given shape.

    python tools/host_step_cost.py --batch 4 --beams 16
    python tools/host_step_cost.py --batch 8 --beams 50 --trie Arts --repetition-penalty 1.1 --prompt-len 253

host us per step: the time the Python loop needs to issue a step (library calls, ctypes, torch views), measured over
whole rollouts issued into an empty queue; gpu us per step: the same steps from first kernel start to last kernel end
(events around the batch, GPU never waiting for the host: the batch is issued ahead behind a long-running kernel);
wall us per step: what a free-running loop achieves.  With --profile the cProfile table of the issue loop."""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--beams", type=int, default=16)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--trie", default="Instruments")
ap.add_argument("--processor", default="count")
ap.add_argument("--prompt-len", type=int, default=61)
ap.add_argument("--repetition-penalty", type=float, default=1.0)
ap.add_argument("--no-fused-step", action="store_true")
ap.add_argument("--rollouts", type=int, default=40)
ap.add_argument("--profile", action="store_true")
args = ap.parse_args()
for k, v in dict(no_keep_top=False, cand_width=0, no_plan_ahead=False, lazy_turnover=False).items():
    setattr(args, k, v)
dev = torch.device("cuda", 0)
wl = bench.build_workload(args, dev)
roll = bench.Rollout(args, wl, dev)
L = wl["levels"]
for _ in range(6 * L):
    roll.step()
torch.cuda.synchronize()
n = args.rollouts * L

# host: issue time of n steps into an empty queue
t0 = time.perf_counter()
for _ in range(n):
    roll.step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_wall = time.perf_counter() - t0

# gpu: the same steps queued behind a kernel that runs longer than their issue takes
blocker = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
for _ in range(max(1, int(t_issue / 0.3e-3) + 2)):
    blocker.add_(1)  # ~0.3 ms each
e0.record()
for _ in range(n):
    roll.step()
e1.record()
torch.cuda.synchronize()
t_gpu = e0.elapsed_time(e1) * 1e-3
R = args.batch * args.beams
print(f"rows {R} ({args.batch} x {args.beams}), trie {args.trie}, penalty {args.repetition_penalty}, "
      f"{'two calls' if args.no_fused_step else 'one call'} per step: host {t_issue / n * 1e6:.1f} us per step, "
      f"gpu {t_gpu / n * 1e6:.1f} us per step, free-running loop {t_wall / n * 1e6:.1f} us per step "
      f"({n / t_wall:.0f} steps/s) over {n} steps")
wl["proc"].check_errors()
if args.profile:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        roll.step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
