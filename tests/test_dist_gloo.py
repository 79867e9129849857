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
        # the product's exchange step (async gather, local-first advantages, slice of the global result in
        # "batch" mode) with the kernel swapped for the oracle, which runs on CPU tensors
        import genrec_amd.advantage as adv_mod

        def cpu_group_advantages(rpf, weights, K_, scale, eps=1e-4):
            r, a, m, s_ = ref_group_advantage(rpf, weights, K_, scale)
            return r, a, m, s_.expand_as(r) if s_.dim() == 0 else s_

        adv_mod.group_advantages = cpu_group_advantages
        for scale in ("group", "batch", "none"):
            want = cpu_group_advantages(full, w, K, scale)
            res = adv_mod.rollout_advantages(local, w, K, scale)
            sl = slice(rank * n_local, (rank + 1) * n_local)
            assert torch.equal(res.local_advantages, want[1][sl]), scale
            assert torch.equal(res.local_rewards, want[0][sl]) and torch.equal(res.local_std, want[3][sl])
            assert torch.equal(res.gathered_rewards_per_func(), full)
            for got_t, want_t in zip(res.wait_global(), want):
                assert torch.equal(got_t, want_t), scale
            assert torch.equal(res[0], want[1][sl]) and torch.equal(tuple(res)[1], want[1])
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


class _CpuRuleScorer:
    """Stand-in for the reward kernel so the cascade controller can run on CPU tensors here."""

    def __init__(self, table):
        self.table = table

    def __call__(self, completion_ids, gt_ids, hint_ids=None, want=("rule",), out=None):
        full = completion_ids if hint_ids is None else torch.cat([hint_ids, completion_ids], 1)
        hits = []
        for row, gt in zip(full.tolist(), gt_ids.tolist()):
            codes = [int(self.table[t]) for t in row if self.table[t] >= 0]
            hits.append(1.0 if codes == [int(self.table[t]) for t in gt] else 0.0)
        return {"rule": torch.tensor(hits)}


def _cascade_worker(rank: int, world: int, port: int, out_dir: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from helpers import cascade_case_tensors, load_json_gz
        from oracle import cascade as ocascade

        from genrec_amd.cascade import run_dynamic_hint_cascade

        cases = [c for c in load_json_gz("cascade.json.gz")["cases"] if c["max_depth"] == 3 and c["L"] == 4]
        pairs = [(cases[0], cases[5]), (cases[3], cases[1]), (cases[6], cases[6])]
        for mine, theirs in (p if rank == 0 else p[::-1] for p in pairs):
            def flags_of(case):  # per stage: does this rank still have unresolved groups afterwards
                table = {(g, d): comps for g, d, comps in case["table"]}
                out = ocascade.run_cascade(case["gts"], case["overrides"], case["K"], case["max_depth"],
                                           lambda gs, ds: [table[(g, d)] for g, d in zip(gs, ds)],
                                           [True] * (case["max_depth"] + 1))
                return [st["remaining_group_count"] > 0 for st in out["stage_stats"]]

            table = {(g, d): comps for g, d, comps in mine["table"]}
            want = ocascade.run_cascade(mine["gts"], mine["overrides"], mine["K"], mine["max_depth"],
                                        lambda gs, ds: [table[(g, d)] for g, d in zip(gs, ds)], flags_of(theirs))
            tok, prompts, gt, caps, ttable = cascade_case_tensors(mine, torch.device("cpu"))
            calls = []

            def generate(stage_prompts, depths, groups):
                calls.append({"groups": groups.tolist(), "depths": depths.tolist()})
                return torch.cat([ttable[(g, d)] for g, d in zip(groups.tolist(), depths.tolist())], 0)

            got = run_dynamic_hint_cascade(generate, prompts, gt, mine["K"], mine["max_depth"],
                                           _CpuRuleScorer(tok.code_table()), tok.pad_token_id, depth_cap=caps)
            assert got.hint_depth.tolist() == want["selected_group_hint_depths"]
            assert got.rule_hit_any.tolist() == want["selected_group_rule_hits"]
            assert got.stage_stats == want["stage_stats"], (rank, got.stage_stats, want["stage_stats"])
            assert calls == [{"groups": c["groups"], "depths": c["depths"]} for c in want["calls"]]
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_cascade_keeps_running_while_the_other_rank_has_unresolved_groups():
    """Two ranks with different scenarios: each runs as many stages as the slower one needs (dummy
    generations once it is done itself), and its decisions equal the oracle's."""
    mp.spawn(_cascade_worker, args=(2, _free_port(), ""), nprocs=2, join=True)


def test_gather_rows_is_identity_without_a_process_group():
    from genrec_amd.advantage import gather_rows

    t = torch.arange(6.0).view(3, 2)
    assert gather_rows(t) is t
