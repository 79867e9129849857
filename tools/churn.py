"""Diagnostics: put the GPU through either a memory churn (many large allocations touched and released) or a
sustained streaming load on little memory, to see which of the two leaves later processes with slower HBM reads."""
import sys
import time

import torch

mode = sys.argv[1]
dev = torch.device("cuda", 0)
if mode == "memory":
    held = []
    for i in range(120):
        t = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
        t.fill_(i)
        held.append(t)
    torch.cuda.synchronize()
    del held[::2]
    torch.cuda.empty_cache()
    more = [torch.empty(1 << 30, dtype=torch.uint8, device=dev).fill_(1) for _ in range(60)]
    torch.cuda.synchronize()
    print("memory churn done", torch.cuda.memory_allocated() >> 30, "GiB held at exit")
elif mode == "load":
    x = torch.empty(4 << 30, dtype=torch.uint8, device=dev)
    t0 = time.time()
    n = 0
    while time.time() - t0 < float(sys.argv[2]):
        for _ in range(50):
            x.add_(1)
        torch.cuda.synchronize()
        n += 50
    print("load done", n, "passes")
