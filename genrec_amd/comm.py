"""The library's own RCCL communicator (``grx_comm_*`` of the C ABI).

The reference gathers the per-rank rewards with ``accelerate.utils.gather``
(fixed_hint_grpo_trainer.py:382, token_prefix_grpo_trainer.py:215-274), i.e. an NCCL all-gather
issued by torch on its own communication stream.  ``RcclComm.allgather`` enqueues the same RCCL
collective on the CURRENT (compute) stream: the reward kernel before it and the advantage kernel
after it are ordered by that stream alone.  ``advantage.use_native_comm(comm)`` routes the exchange
step through it.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib

ID_BYTES = 128


def unique_id() -> bytes:
    """A fresh communicator id (rank 0 draws it, every rank needs the same bytes)."""
    buf = (C.c_char * ID_BYTES)()
    _lib.check(_lib.lib().grx_comm_unique_id(buf))
    return bytes(buf)


class RcclComm:
    """One rank's end of an RCCL communicator bound to ``device``.  Creating it is collective.

    The all-gather runs on the caller's current stream through this communicator, beside whatever
    ``torch.distributed`` does with its own NCCL communicator on its own stream: collectives of the two must
    not be in flight on a rank at the same time -- every rank has to issue them in one common order, so
    synchronise the stream (or wait for the torch collective's work handle) before switching from one to
    the other.  ``close()`` waits for the device before the communicator is destroyed."""

    def __init__(self, n_ranks: int, rank: int, comm_id: bytes, device):
        if len(comm_id) != ID_BYTES:
            raise ValueError(f"communicator ids are {ID_BYTES} bytes")
        self.device = torch.device(device)
        self.n_ranks, self.rank = int(n_ranks), int(rank)
        out = C.c_void_p()
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(index):
            _lib.check(_lib.lib().grx_comm_create(comm_id, self.n_ranks, self.rank, C.byref(out)))
        self.ptr = out.value
        self._index = index

    @classmethod
    def from_process_group(cls, device, group=None) -> "RcclComm":
        """Build the communicator next to an initialised ``torch.distributed`` group, which only
        carries the id from rank 0 to the others."""
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(world, rank, box[0], device)

    def allgather(self, t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """All-gather along dim 0 on the current stream; ``out`` is [n_ranks * rows, ...]."""
        _lib.require_cuda(t, "allgather input")
        t = t.contiguous()
        shape = (self.n_ranks * t.shape[0],) + tuple(t.shape[1:])
        if out is None:
            out = torch.empty(shape, dtype=t.dtype, device=t.device)
        assert out.shape == shape and out.dtype == t.dtype and out.is_contiguous() and out.device == t.device
        with torch.cuda.device(self._index):
            _lib.check(_lib.lib().grx_comm_allgather(self.ptr, t.data_ptr(), out.data_ptr(), t.numel() * t.element_size(),
                                                     _lib.current_stream_ptr(t.device)))
        return out

    def close(self) -> None:
        if self.ptr:
            _lib.lib().grx_comm_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
