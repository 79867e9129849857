"""Whole generations as a HIP graph.

At the shapes the reference's launchers use -- 64 rows per device in RL (trl_trainer.sh:107-118: 16
generations x 4 prompts), 8 x 50 = 400 rows in evaluation (evaluate.sh:31-36) -- a decode step is a few
tens of microseconds of GPU work, about what the host needs to issue it (library calls through ctypes,
torch views).  Recording the launches of whole generations once and replaying them takes the host out
of the loop.

What makes a sequence of library calls replayable: every kernel argument is baked in at capture, among
them the slot of the processor's three-slot state ring that a count-based call owns (its call number
modulo 3).  A recorded sequence therefore has to make a multiple of three count-based calls -- e.g.
three generations of any depth -- so that a replay starts in the ring phase the capture started in and
leaves it there; `capture` checks this.  Everything else a generation needs on the host side (the
counter reset, plans made ahead) is balanced within a generation: `proc.count = 0` before the state's
`reset(..., planner=proc)`, no plan made ahead by a generation's last step.

The recorded work reads and writes fixed device buffers (prompts, logits, beam state, candidates,
rewards): refill the inputs in place between replays, read the outputs after one.
"""

from __future__ import annotations

from typing import Callable, Iterable

import torch


class CapturedSteps:
    """A recorded sequence of decode work; ``replay()`` enqueues it on the current stream."""

    def __init__(self, graph: "torch.cuda.CUDAGraph", calls: int):
        self.graph = graph
        self.processor_calls = calls

    def replay(self) -> None:
        self.graph.replay()


def _calls(proc) -> int:
    total = 0
    for index in list(proc._states):
        total += proc._read_state(index)[3]
    return total


def capture(fn: Callable[[], None], processors: Iterable, warmup: int = 1) -> CapturedSteps:
    """Record ``fn()`` -- which issues decode work through genrec_amd (beam state resets, processor
    calls, beam steps, scoring) on the current device -- as a HIP graph.

    ``processors``: every genrec_amd processor that ``fn`` drives.  ``fn`` runs ``warmup`` times first
    (buffers allocated, state objects created: nothing may allocate outside torch's allocator during
    capture) and once more under capture.  It must not synchronise or read anything back (no
    ``check_errors()``, no ``proc.count`` reads, ``sync_errors`` off) and must return the host-side
    objects to where they were: whole generations, begun with ``proc.count = 0`` and a state reset."""
    processors = list(processors)
    for _ in range(max(1, warmup)):
        fn()
    torch.cuda.synchronize()
    before = [_calls(p) for p in processors]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn()
    torch.cuda.synchronize()
    made = [_calls(p) - b for p, b in zip(processors, before)]
    for p, n in zip(processors, made):
        if getattr(p, "_mode", 0) == 0 and n % 3:
            raise ValueError(
                f"the recorded work makes {n} count-based processor calls; a replayable sequence makes a multiple of "
                "three (the state ring has three slots): record three generations instead of one")
    return CapturedSteps(graph, sum(made))
