"""Worker of tests/test_gpu_exchange_ranks.py: one rank of a two-rank job sharing the test box's GPU
(gloo; RCCL refuses two ranks on one device).  Every rank scores its shard of whole prompt groups
with the reward kernel, runs the exchange step (``rollout_advantages``) and compares what it got with
a single-process run over the concatenated rows of all ranks: its own advantages, the gathered
reward matrix and the global statistics, for every ``scale_rewards`` mode."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from genrec_amd.advantage import group_advantages, rollout_advantages
    from genrec_amd.rewards.ranking_reward import SidRewardScorer

    dev = torch.device("cuda", 0)
    K, groups_per_rank, L = 16, 24, 4
    n_local = K * groups_per_rank
    g = torch.Generator().manual_seed(77)  # same stream on every rank: everyone can rebuild all rows
    table = torch.full((600,), -1, dtype=torch.int32)
    table[100:600] = torch.arange(500, dtype=torch.int32) % 37
    gt = torch.randint(100, 600, (world * groups_per_rank, L), generator=g).repeat_interleave(K, dim=0)
    comp = torch.randint(100, 600, (world * n_local, L), generator=g)
    hit = torch.rand(world * n_local, generator=g) < 0.2
    comp[hit] = gt[hit]  # some exact rows, so NDCG penalties and rule rewards are not all zero
    comp[K : 2 * K] = gt[K : 2 * K]  # a zero-variance group on rank 0
    weights = torch.tensor([1.0, 0.7], device=dev)
    scorer = SidRewardScorer(table.numpy(), K, dev)

    def score(rows):
        out = torch.empty((rows.stop - rows.start, 2), dtype=torch.float32, device=dev)
        scorer(comp[rows].to(dev), gt[rows].to(dev), want=("rule", "ndcg"), out=out)
        return out

    mine = slice(rank * n_local, (rank + 1) * n_local)
    local_rpf = score(mine)
    full_rpf = score(slice(0, world * n_local))  # what a single process holding every row computes
    assert float(full_rpf[:, 0].sum()) > 0 and float(full_rpf[:, 1].abs().sum()) > 0
    for scale in ("group", "batch", "none"):
        want_rewards, want_adv, want_mean, want_std = group_advantages(full_rpf, weights, K, scale)
        res = rollout_advantages(local_rpf, weights, K, scale)
        assert res.local_advantages.shape == (n_local,)
        assert torch.equal(res.local_advantages, want_adv[mine]), (rank, scale)
        assert torch.equal(res.local_rewards, want_rewards[mine])
        assert torch.equal(res.gathered_rewards_per_func(), full_rpf), (rank, scale)
        rewards, adv, mean, std = res.wait_global()
        assert torch.equal(rewards, want_rewards) and torch.equal(adv, want_adv)
        assert torch.equal(mean, want_mean) and torch.equal(std, want_std)
        local, all_adv, r2, m2, s2 = res  # the 5-tuple the call used to return
        assert torch.equal(local, want_adv[mine]) and torch.equal(all_adv, want_adv)
        if scale == "batch":  # the std spans both ranks: a rank alone would get another value
            alone = group_advantages(local_rpf, weights, K, scale)[3]
            assert not torch.equal(alone[:1], std[:1])
    torch.cuda.synchronize()
    dist.barrier()
    print(f"exchange rank {rank} ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
