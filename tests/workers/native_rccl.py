"""Worker of tests/test_gpu_exchange_ranks.py: the library's own RCCL communicator (grx_comm_*) with
the one rank a one-GPU box offers -- id, communicator, all-gather on the compute stream -- and the
exchange step routed through it."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from genrec_amd import advantage
    from genrec_amd.comm import ID_BYTES, RcclComm, unique_id

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    a, b = unique_id(), unique_id()
    assert len(a) == ID_BYTES and a != b
    comm = RcclComm(1, 0, a, dev)
    torch.manual_seed(1)
    for shape, dtype in (((640, 2), torch.float32), ((7,), torch.int64), ((3, 5, 2), torch.bfloat16)):
        t = (torch.randn(shape, device=dev) * 10).to(dtype)
        out = comm.allgather(t)
        assert out.data_ptr() != t.data_ptr() and torch.equal(out, t)
    side = torch.cuda.Stream(device=dev)  # the collective follows the current stream
    with torch.cuda.stream(side):
        x = torch.full((1 << 20,), 3.0, device=dev)
        y = comm.allgather(x * 2)
    side.synchronize()
    assert float(y.sum()) == 6.0 * (1 << 20)
    # the exchange step through it
    advantage.use_native_comm(comm)
    K, G = 16, 24
    rpf = torch.randn(G * K, 2, device=dev)
    w = torch.tensor([1.0, 0.5], device=dev)
    for scale in ("group", "batch", "none"):
        want = advantage.group_advantages(rpf, w, K, scale)
        res = advantage.rollout_advantages(rpf, w, K, scale, process_index=0)
        g = res.gathered_rewards_per_func()
        assert g.data_ptr() != rpf.data_ptr() and torch.equal(g, rpf)
        assert torch.equal(res.local_advantages, want[1])
        for p, q in zip(res.wait_global(), want):
            assert torch.equal(p, q)
    advantage.use_native_comm(None)
    comm.close()
    torch.cuda.synchronize()
    print("native rccl ok", flush=True)


if __name__ == "__main__":
    main()
