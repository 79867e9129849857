#!/usr/bin/env python3
"""How long the LM head that produces the [R, V] logits takes on this GPU (library GEMM through torch),
next to the constrained step that consumes them.  Input to the decision whether a fused LM-head epilogue
(logits never written to HBM) is worth a hand-written GEMM."""
import torch

R, V = 8192, 152640
dev = torch.device("cuda", 0)
for H in (1536, 2048, 3584):
    h = torch.randn(R, H, device=dev, dtype=torch.bfloat16)
    w = torch.randn(V, H, device=dev, dtype=torch.bfloat16) * 0.02
    out = torch.empty(R, V, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(h, w.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        torch.matmul(h, w.t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    flops = 2.0 * R * V * H
    print(f"hidden {H}: LM head [{R} x {H}] x [{H} x {V}] bf16 = {ms:.3f} ms = {flops / ms / 1e9:.0f} TFLOP/s; "
          f"logits written {R * V * 2 / 1e9:.2f} GB")
    del h, w, out
