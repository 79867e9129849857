"""Diagnostics: read throughput of one 2.5 GB buffer (grx_read_probe, 10 launches) once a second, to watch how
long a GPU takes to get back to its steady read rate after another process released a lot of memory."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genrec_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.lib()
R, V = 8192, 152640
x = torch.empty((R, V), dtype=torch.bfloat16, device=dev)
x.normal_()
out = torch.empty(R * 4, dtype=torch.int32, device=dev)
stream = _lib.current_stream_ptr(dev)
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 40
t0 = time.time()
line = []
while time.time() - t0 < seconds:
    lib.grx_timing_enable(1)
    for _ in range(10):
        _lib.check(lib.grx_read_probe(x.data_ptr(), V * 2, R, V * 2, out.data_ptr(), stream))
    torch.cuda.synchronize()
    tot, n = C.c_double(), C.c_int32()
    _lib.check(lib.grx_timing_read(C.byref(tot), C.byref(n)))
    lib.grx_timing_enable(0)
    line.append(f"{time.time() - t0:.0f}s:{tot.value / n.value * 1e3:.1f}")
    time.sleep(1.0)
print(" ".join(line))
