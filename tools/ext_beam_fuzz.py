#!/usr/bin/env python3
"""More of tests/test_gpu_fuzz.py's random beam searches, weighted towards groups of 17 .. 64 beams with rows of up
to 256 candidates (the 1024-thread instantiation of the beam step): 30 further blocks of 20 searches, every live
slot of every step against the CPU restatement of the transformers loop.  Run on a GPU box from the repo root."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as fuzz  # noqa: E402

_plain_case = fuzz._random_beam_case


def _case(rng):
    c = _plain_case(rng)
    if rng.random() < 0.6:
        c["K"] = int(rng.choice([17, 20, 31, 32, 33, 40, 48, 50, 63, 64]))
        c["cmax"] = int(rng.integers(1, min(c["V"] - 2, 256) + 1))
    return c


fuzz._random_beam_case = _case
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for block in range(first, first + 30):
    fuzz.test_random_beam_searches_against_the_oracle(block)
    print("block", block, "ok", flush=True)
