"""Worker of tests/test_gpu_exchange_ranks.py: the exchange step through RCCL itself (backend "nccl") on
the one GPU of the test box.  A process group of one rank gathers nothing, so
GENREC_AMD_SINGLE_RANK_COLLECTIVES=1 makes the code issue its collectives anyway: the asynchronous
``all_gather_into_tensor`` on device tensors, the stream-ordered wait, the reduction of the cascade's
consensus flag -- the calls an 8-GPU job makes, with one participant."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    os.environ["GENREC_AMD_SINGLE_RANK_COLLECTIVES"] = "1"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    from genrec_amd import advantage
    from genrec_amd.advantage import gather_rows, group_advantages, rollout_advantages
    from genrec_amd.cascade import _stage_sync

    K, G = 16, 40
    torch.manual_seed(3)
    rpf = torch.randn(G * K, 2, device=dev)
    w = torch.tensor([1.0, 0.5], device=dev)
    g = gather_rows(rpf)
    assert g.data_ptr() != rpf.data_ptr() and torch.equal(g, rpf)  # went through the collective
    for scale in ("group", "batch", "none"):
        want = group_advantages(rpf, w, K, scale)
        res = rollout_advantages(rpf, w, K, scale)
        assert res._pending[0] is not None or scale == "batch"  # an asynchronous collective is in flight
        assert torch.equal(res.local_advantages, want[1])
        got = res.wait_global()
        assert torch.equal(res.gathered_rewards_per_func(), rpf)
        for a, b in zip(got, want):
            assert torch.equal(a, b), scale
    # several exchanges in flight, collected later (what bench.py does across rollouts)
    pend = [rollout_advantages(rpf + i, w, K, "group") for i in range(4)]
    for i, res in enumerate(pend):
        assert torch.equal(res.gathered_rewards_per_func(), rpf + i)
    # the cascade's packed read with the consensus reduced on the device
    # (world size 1 short-circuits inside _stage_sync; call the collective directly)
    flag = torch.tensor([1], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    assert _stage_sync(torch.tensor([3, 5], dtype=torch.int32, device=dev), 7, dev) == (3, 5, True)
    torch.cuda.synchronize()
    print("rccl single rank ok", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
