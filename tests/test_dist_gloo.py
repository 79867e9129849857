"""CPU, two processes over gloo: the rollout's one exchange step -- all-gather of the
per-rank rewards, group statistics on the gathered rows, local slice
(reference: fixed_hint_grpo_trainer.py:382-408) -- and the group-wise sharding of prompts."""

import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.scoring import ref_group_advantage

        from genrec_amd.advantage import gather_rows

        K, groups_per_rank, F = 4, 3, 2
        torch.manual_seed(123)
        full = torch.randn(world * groups_per_rank * K, F)  # what a single process would hold
        n_local = groups_per_rank * K
        local = full[rank * n_local : (rank + 1) * n_local].clone()  # whole groups per rank
        gathered = gather_rows(local)
        assert torch.equal(gathered, full), "all-gather must rebuild the global row order"
        w = torch.tensor([1.0, 0.5])
        rewards, adv, mean, std = ref_group_advantage(gathered, w, K, "group")
        mine = adv[rank * n_local : (rank + 1) * n_local]
        # groups never straddle ranks, so local statistics equal the global ones
        _, adv_local, _, _ = ref_group_advantage(local, w, K, "group")
        assert torch.equal(mine, adv_local)
        # "batch" scaling needs the gather: the std spans all ranks
        _, adv_b, _, std_b = ref_group_advantage(gathered, w, K, "batch")
        _, _, _, std_b_local = ref_group_advantage(local, w, K, "batch")
        assert not torch.equal(std_b[:1], std_b_local[:1])
        one = gather_rows(torch.tensor([float(rank)]))
        assert one.tolist() == [float(r) for r in range(world)]
        torch.save(mine, os.path.join(out_dir, f"adv{rank}.pt"))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_reward_allgather_and_local_slices(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(tmp_path / f"adv{r}.pt") for r in range(world)]
    assert all(p.shape == (12,) for p in parts)


def test_gather_rows_is_identity_without_a_process_group():
    from genrec_amd.advantage import gather_rows

    t = torch.arange(6.0).view(3, 2)
    assert gather_rows(t) is t
