"""The rollout's exchange step with world_size 2 on the GPU: every rank's advantages, the gathered
rewards and the global statistics equal a single-process run over the concatenated rows, for
scale_rewards "group", "batch" and "none" (reference: fixed_hint_grpo_trainer.py:382-408).  Two ranks
share the one GPU of the test box, so the process group is gloo; the kernels and the slicing logic are
the ones an RCCL job runs."""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_get_the_single_process_values():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "tests", "workers", "exchange_ranks.py")]
    run = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=200)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "exchange rank 0 ok" in run.stdout and "exchange rank 1 ok" in run.stdout


@pytest.mark.gpu
def test_the_rccl_code_path_runs_with_one_rank():
    """backend "nccl" (RCCL) on the box's one GPU: the asynchronous device all-gather of the exchange
    step and the device-side reduction, issued with a single participant."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29619", os.path.join(ROOT, "tests", "workers", "rccl_single_rank.py")]
    run = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=200)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "rccl single rank ok" in run.stdout


@pytest.mark.gpu
def test_the_librarys_own_rccl_communicator_with_one_rank():
    """grx_comm_unique_id / grx_comm_create / grx_comm_allgather (RCCL bound at run time, the collective
    enqueued on the compute stream) and the exchange step routed through them."""
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "workers", "native_rccl.py")], cwd=ROOT,
                         capture_output=True, text=True, timeout=200)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "native rccl ok" in run.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("scale", ["group", "batch", "none"])
def test_exchange_through_a_stream_ordered_communicator_uses_the_rows_of_all_ranks(scale):
    """A communicator whose all-gather is ordered by the stream hands back no work handle (the library's
    own RCCL communicator): the exchange step must still normalise "batch" over the rows of all ranks.
    The second rank is played by a stand-in whose all-gather puts another rank's rewards in front."""
    import torch

    from genrec_amd import advantage

    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    K, G = 4, 6
    torch.manual_seed(3)
    other = torch.randn(G * K, 2, device=dev)
    mine = torch.randn(G * K, 2, device=dev) * 3 + 1
    w = torch.tensor([1.0, 0.5], device=dev)

    class TwoRanks:
        n_ranks, rank = 2, 1

        def allgather(self, t):
            return torch.cat([other, t.contiguous()], dim=0)

    want = advantage.group_advantages(torch.cat([other, mine]), w, K, scale)
    advantage.use_native_comm(TwoRanks())
    try:
        res = advantage.rollout_advantages(mine, w, K, scale)
        local, all_adv, rewards, mean, std = res
    finally:
        advantage.use_native_comm(None)
    sl = slice(G * K, 2 * G * K)
    assert torch.equal(all_adv, want[1]) and torch.equal(rewards, want[0])
    assert torch.equal(mean, want[2]) and torch.equal(std, want[3])
    if scale == "batch":
        assert torch.equal(local, want[1][sl])
        alone = advantage.group_advantages(mine, w, K, scale)[1]
        assert not torch.equal(local, alone), "the batch std must come from both ranks' rows"
    else:
        assert float((local - want[1][sl]).abs().max()) <= 1e-6
    assert torch.equal(res.gathered_rewards_per_func(), torch.cat([other, mine]))
