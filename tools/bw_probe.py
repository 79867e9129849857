#!/usr/bin/env python3
"""Reference points for HBM traffic mixes on this GPU (torch kernels): fill, copy, read-reduce."""
import torch

dev = torch.device("cuda", 0)
n = 8192 * 152640
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


for name, fn, nbytes in (
    ("fill_(-inf) fp32", lambda: x.fill_(float("-inf")), n * 4),
    ("zero_ fp32", lambda: x.zero_(), n * 4),
    ("copy_ fp32", lambda: y.copy_(x), n * 8),
    ("sum fp32", lambda: x.sum(), n * 4),
):
    ms = timeit(fn)
    print(f"{name:20s} {ms * 1e3:9.1f} us  {nbytes / ms / 1e6:7.0f} GB/s")
