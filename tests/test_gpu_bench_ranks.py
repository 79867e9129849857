"""bench.py's multi-rank flow (barriers, per-rank shards, reward all-gather, max-over-ranks timing,
rank-0 JSON line) rehearsed with two ranks sharing the one GPU of the test box.  RCCL refuses two ranks
on one device, so the rehearsal swaps the backend for gloo (GENREC_BENCH_BACKEND); everything else is
the code path of `--gpus N`."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_rank_bench_line():
    env = dict(os.environ, GENREC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10",
           "--warmup", "5", "--batch", "32"]
    run = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["warmup"] == 5 and d["scaling"] == "weak"
    assert d["config"]["parallelism"].startswith("dp2")
    # whole-job value: both ranks' steps over the slowest rank's time
    assert abs(d["value"] - 2 * 10 / (d["ms_per_step"] * 10 / 1e3)) <= 1e-6 * d["value"]
    assert d["roofline"]["launches_timed"] >= 2 and d["roofline"]["achieved"] > 0
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only


@pytest.mark.gpu
def test_bench_through_rccl_with_one_rank():
    """`bench.py` with the process group created on backend "nccl" (RCCL) and every collective of the
    multi-rank flow issued -- barriers, the asynchronous reward all-gather per rollout, the max-reduction of
    the elapsed time -- with the one rank a one-GPU box can give it (GENREC_BENCH_FORCE_DIST=1)."""
    env = dict(os.environ, GENREC_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "5", "--batch", "64",
           "--no-cpu-baseline", "--no-processor-surface"]
    for native in ("0", "1"):  # the gather through torch.distributed, then through the library's own communicator
        run = subprocess.run(cmd, env=dict(env, GENREC_BENCH_NATIVE_RCCL=native), cwd=ROOT, capture_output=True,
                             text=True, timeout=280)
        assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
        lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["launches_timed"] >= 2


@pytest.mark.gpu
def test_plain_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (the reference's
    launchers do the same: trl_trainer.sh:359-364, evaluate.sh:140-152) and says in the line who answered."""
    env = dict(os.environ, GENREC_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "32", "--steps", "10",
           "--warmup", "5"]
    run = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and len(d["devices"]) == 2
    assert d["roofline"]["step_frac"] > 0


@pytest.mark.gpu
def test_rank_count_that_contradicts_the_flag_is_refused():
    """A launcher that started one rank cannot produce a line labelled two GPUs."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "32", "--steps", "5",
           "--warmup", "5", "--no-cpu-baseline", "--no-processor-surface"]
    run = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert run.returncode != 0
    assert not [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert "WORLD_SIZE=1" in run.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [["--batch", "4", "--beams", "16"],
                                   ["--batch", "8", "--beams", "50", "--trie", "Arts", "--repetition-penalty", "1.1",
                                    "--prompt-len", "253"]])
def test_bench_modes_at_the_launchers_shapes(shape):
    """`bench.py` at the shapes the reference's launchers run (trl_trainer.sh:107-118: 64 rows per device;
    evaluate.sh:31-36: 8 x 50 rows with penalty 1.1), eagerly, as a replayed HIP graph and with two library calls per
    step: one JSON line each, the mode recorded in it, and the graph no slower than the eager loop."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "30", "--no-cpu-baseline",
            "--no-processor-surface", "--no-settle"] + shape
    got = {}
    for mode, flags in (("eager", []), ("graph", ["--graph"]), ("two calls", ["--no-fused-step"])):
        run = subprocess.run(base + flags, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
        assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
        lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        got[mode] = json.loads(lines[0])
        assert got[mode]["value"] > 0 and got[mode]["roofline"]["hip_graph"] == (mode == "graph")
    assert got["graph"]["value"] >= 0.95 * got["eager"]["value"]
