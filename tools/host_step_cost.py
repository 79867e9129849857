"""Host time of one bench step (Python + launches), measured while the GPU queue is never full: how long the GPU
waits for its first kernel at the start of a timed region, and what cProfile says the time goes to."""
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
args = ap.parse_args([])
defaults = dict(batch=512, beams=16, dtype="bf16", trie="Instruments", processor="count", prompt_len=61,
                repetition_penalty=1.0, no_keep_top=False, cand_width=0, no_plan_ahead=False)
for k, v in defaults.items():
    setattr(args, k, v)
dev = torch.device("cuda", 0)
wl = bench.build_workload(args, dev)
roll = bench.Rollout(args, wl, dev)
for _ in range(30):
    roll.step()
torch.cuda.synchronize()
ts = []
for _ in range(25):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    roll.step()
    ts.append((time.perf_counter() - t0) * 1e6)
print("host us per step (GPU idle at call):", [round(t) for t in ts])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    roll.step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
