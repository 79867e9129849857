#!/usr/bin/env python3
"""Benchmark of the trie-constrained decode step on MI355X.

    python bench.py --gpus 1 --steps 100 --warmup 20
    python bench.py --gpus N ...      (starts the N ranks itself, one per GPU, and relays rank 0's line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" is one pass of the decoding hot path over batch x beam = 8192 rows of
synthetic logits resident in HBM: resolve the allowed-token node of every row,
stream the [8192, 152640] logits once (masked log-softmax emitting the compact
candidates) and run the beam step (top-2K, parents, finished merge, reorder
index).  Steps run as real generations: 5 consecutive steps (4 SID levels + EOS)
advance one beam search over the Instruments 4-256 trie, then the rollout is
scored (rule + NDCG rewards, all-gather of the rewards across ranks, group
advantages) and the next generation starts.  Every rank runs its own shard of
prompt groups; the only collective is the reward all-gather.

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def build_workload(args, device):
    import numpy as np
    import torch

    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows, write_index_file
    from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn

    tmp = tempfile.mkdtemp(prefix="genrec_bench_")
    if args.trie.startswith("cb"):  # codebook sweep: synthetic 4-level index with the given codebook size
        from genrec_amd.synth import synthetic_index, write_index

        path = write_index(os.path.join(tmp, f"{args.trie}.index.json"),
                           synthetic_index(9922, 4, int(args.trie[2:]), seed=0))
    else:
        path = write_index_file(args.trie, tmp)
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    fn = create_prefix_allowed_tokens_fn(trie, suffix)
    V = tok.vocab_size_padded
    cls = FixedHintConstrainedLogitsProcessor if args.processor == "fixed_hint" else ConstrainedLogitsProcessor
    extras = {"repetition_penalty": args.repetition_penalty} if args.repetition_penalty != 1.0 else {}
    proc = cls(fn, args.beams, pidx, suffix, tok.eos_token_id, **extras)
    levels = len(trie.level_fanout()) - len(suffix)  # SID levels + EOS
    rank = int(os.environ.get("RANK", "0"))
    prompts = sample_decode_rows(trie, args.batch, 1, 0, prompt_len=args.prompt_len, seed=1000 + rank)
    prompts = torch.from_numpy(prompts).to(device)
    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}[args.dtype]
    R = args.batch * args.beams
    gen = torch.Generator(device=device).manual_seed(4242 + rank)
    logits = []
    for _ in range(levels):
        x = torch.empty((R, V), dtype=dtype, device=device)
        # N(0, 4^2) in chunks to bound the fp32 temporary
        for r0 in range(0, R, 1024):
            r1 = min(R, r0 + 1024)
            x[r0:r1] = (torch.randn((r1 - r0, V), generator=gen, device=device) * 4.0).to(dtype)
        logits.append(x)
    # ground truth of the rollout: one random item per prompt, as SID token ids
    seqs = np.asarray(sorted(trie), dtype=np.int64)
    rng = np.random.default_rng(7 + rank)
    gt = seqs[rng.integers(0, len(seqs), size=args.batch)][:, len(suffix) : -1]
    gt = torch.from_numpy(np.repeat(gt, args.beams, axis=0)).to(device)
    return dict(trie=trie, tok=tok, proc=proc, V=V, levels=levels, prompts=prompts, logits=logits, gt=gt,
                suffix=suffix, pidx=pidx, fn=fn)


class Rollout:
    """Generation loop state: beam search over `levels` steps, then scoring."""

    def __init__(self, args, wl, device):
        import torch

        from genrec_amd.rewards.ranking_reward import SidRewardScorer

        self.args, self.wl, self.device = args, wl, device
        self.R = args.batch * args.beams
        from genrec_amd.logit_processor import Candidates

        # a group keeps 2 x beams candidates per step, so rows hand over at most that many (their best)
        self.keep = 0 if getattr(args, "no_keep_top", False) else wl["proc"].keep_top_for(2 * args.beams)
        width = wl["proc"].candidate_width(2 * args.beams if self.keep else 0)
        self.cand = Candidates(self.R, max(width, getattr(args, "cand_width", 0)), device)
        self.scorer = SidRewardScorer(wl["tok"].code_table(wl["V"]), args.beams, device)
        self.weights = torch.tensor([1.0, 1.0], dtype=torch.float32, device=device)
        self.rpf = torch.empty((self.R, 2), dtype=torch.float32, device=device)
        self.state = None
        self.depth = 0
        self.exchange = None
        self.plan_ahead = not getattr(args, "no_plan_ahead", False)
        self.lazy_turnover = bool(getattr(args, "lazy_turnover", False))
        self.fused = not getattr(args, "no_fused_step", False)
        self.prompt_rows = wl["prompts"].repeat_interleave(args.beams, dim=0).contiguous()

    def new_generation(self):
        from genrec_amd.beam import BeamSearchState

        wl, a = self.wl, self.args
        if hasattr(wl["proc"], "reset"):
            wl["proc"].count = 0  # no error read-back inside the timed loop
        planner = wl["proc"] if self.plan_ahead else None
        if self.state is None:
            self.state = BeamSearchState(self.prompt_rows, a.beams, self.prompt_rows.shape[1] + wl["levels"] + 1,
                                         wl["tok"].eos_token_id, wl["tok"].pad_token_id, 1.0, False,
                                         shared_prompt=True, planner=planner)
        else:
            self.state.reset(self.prompt_rows, planner)  # full re-initialisation from the prompts
        self.depth = 0

    def step(self):
        wl = self.wl
        if self.state is None:
            self.new_generation()
        elif self.depth == wl["levels"] and self.lazy_turnover:
            self.score()
            self.new_generation()
        # the beam step plans the next processor call, except after the last level of a generation
        planner = wl["proc"] if (self.plan_ahead and self.depth + 1 < wl["levels"]) else None
        if self.fused:
            # one library call per step: a group's beam step runs in the streaming kernel as soon as its rows are done
            self.state.fused_step(wl["proc"], wl["logits"][self.depth], self.cand, keep_top=self.keep, planner=planner)
        else:
            flat = self.state.flat_running_sequences()
            wl["proc"].candidates_into(flat, wl["logits"][self.depth], self.cand, keep_top=self.keep,
                                       row_bias=self.state.row_bias() if self.keep else None)
            self.state.step(self.cand, wl["V"], planner)
        self.depth += 1
        if self.depth == wl["levels"] and not self.lazy_turnover:
            # the generation is complete: its last step scores it and re-initialises the beam state from the
            # prompts for the next one.  One scoring and one initialisation per rollout either way; done here,
            # the host issues them while it is far ahead of the GPU, instead of as the first thing after a
            # synchronisation, with the GPU waiting for the launches
            self.score()
            self.new_generation()

    def score(self):
        from genrec_amd.advantage import rollout_score

        st = self.state
        gen = st.sequences[:, :, st.prompt_len : st.cur_len].reshape(self.R, -1)
        # the advantages of the local groups are ready in stream order; the all-gather behind the logged
        # global statistics runs beside the next generation and is collected one rollout later
        if self.exchange is not None:
            self.gathered = self.exchange.gathered_rewards_per_func()
        self.exchange = rollout_score(self.scorer, gen, self.wl["gt"], self.weights, "group", want=("rule", "ndcg"),
                                      out=self.rpf)
        self.adv = self.exchange.local_advantages

    def finish(self):
        """Collect the exchange of the last rollout (inside the timed region)."""
        if self.exchange is not None:
            self.gathered = self.exchange.gathered_rewards_per_func()
            self.exchange = None


def time_cpu_baseline(args, wl, cores=None):
    """The oracle (CPU restatement of the reference processor + HF beam step) on the workload's own
    row count: one generation (`levels` steps over all batch x beams rows), timed step by step, with
    `cores` torch threads (default: the box's CPU share for one GPU, 16 cores)."""
    import torch

    from oracle.beam import RefBeamSearch
    from oracle.processors import DictTrie, RefConstrainedLogitsProcessor, RefFixedHintConstrainedLogitsProcessor

    groups = args.batch if args.cpu_groups <= 0 else min(args.cpu_groups, args.batch)
    K = args.beams
    trie = DictTrie(sorted(wl["trie"]))
    cls = RefFixedHintConstrainedLogitsProcessor if args.processor == "fixed_hint" else RefConstrainedLogitsProcessor
    proc = cls(trie.get, K, wl["pidx"], wl["suffix"], wl["tok"].eos_token_id)
    prompts = wl["prompts"][:groups].cpu().repeat_interleave(K, dim=0)
    V, levels = wl["V"], wl["levels"]
    # the box's CPU share for one GPU is 16 cores
    cores = min(cores or 16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    t_total, n_steps = 0.0, 0
    while n_steps < levels or (t_total < args.cpu_seconds and groups < args.batch):
        state = RefBeamSearch(prompts, K, prompts.shape[1] + levels + 1, wl["tok"].eos_token_id,
                              wl["tok"].pad_token_id, 1.0, False)
        if hasattr(proc, "reset"):
            proc.reset()
        for d in range(levels):
            x = wl["logits"][d][: groups * K].float().cpu()  # fp32 on the host, as HF hands them over (untimed)
            t0 = time.perf_counter()
            flat = state.flat_running_sequences()
            lp = torch.nn.functional.log_softmax(x, dim=-1)  # HF normalises first (utils.py:3388)
            lp = proc(flat, lp)
            state.step(lp)
            t_total += time.perf_counter() - t0
            n_steps += 1
            del x, lp
    rows = groups * K
    out = {
        "value": n_steps / t_total * (rows / (args.batch * K)),
        "unit": "steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n_steps} steps of {rows} rows (fp32 logits, V={V}) in {t_total:.1f}s"
                  + ("" if rows == args.batch * K else f"; scaled to steps of {args.batch * K} rows"),
    }
    if rows != args.batch * K:
        out["value_extrapolated_from_rows"] = rows
    return out


def time_processor_surface(args, wl, device, calls=10):
    """The unmodified-HF-loop form of the step: processor(input_ids, scores) -> dense [R, V] result
    (fill kernel + stream kernel), timed by the library's dispatch events.  Algorithmic bytes are
    input once + output once.  Not part of `value`."""
    import ctypes as C

    import torch

    from genrec_amd import _lib
    from genrec_amd.synth import sample_decode_rows

    lib = _lib.lib()
    proc = wl["proc"]
    R, V = args.batch * args.beams, wl["V"]
    ids = torch.from_numpy(sample_decode_rows(wl["trie"], args.batch, args.beams, 1, prompt_len=args.prompt_len,
                                              seed=5)).to(device)
    out = {}
    # third mode: what generate() hands over when nothing runs ahead of the processor -- rows that already
    # are log-softmax values; the processor is told so and only writes (fill + gather of the allowed values)
    make = lambda **kw: type(proc)(wl["fn"], args.beams, wl["pidx"], wl["suffix"], wl["tok"].eos_token_id, **kw)
    vouched = make(assume_normalized=True)
    # fourth and fifth mode: the processor keeps ONE result tensor for every call (safe under generate(),
    # which reads the result once per step): a call moves the few finite positions instead of writing
    # rows x vocab of -inf, so its traffic is the read of the scores -- or, with normalised rows, next to none
    kept = make(reuse_output=True)
    kept_vouched = make(reuse_output=True, assume_normalized=True)
    procs = [proc, vouched, kept, kept_vouched]
    for name, x, p in (("bf16->bf16", wl["logits"][1], proc), ("fp32->fp32", None, proc),
                       ("fp32->fp32, rows already normalised", None, vouched),
                       ("fp32->fp32, one kept result tensor", None, kept),
                       ("fp32->fp32, rows already normalised, one kept result tensor", None, kept_vouched)):
        if x is None:
            x = wl["logits"][1].float()
            if p.assume_normalized:
                x = torch.log_softmax(x, dim=-1)
        res = None if p.reuse_output else torch.empty_like(x)
        for i in range(calls + 2):
            if i == 2:
                torch.cuda.synchronize()
                lib.grx_timing_enable(1)
            if hasattr(p, "reset"):
                p.count = 1
            p.call_into(ids, x, res)
        torch.cuda.synchronize()
        tot, n = C.c_double(), C.c_int32()
        _lib.check(lib.grx_timing_read(C.byref(tot), C.byref(n)))
        lib.grx_timing_enable(0)
        ms = tot.value / max(n.value, 1)
        # bytes a call has to move: the scores once unless vouched for, the result once unless kept
        nbytes = ((0 if p.assume_normalized else 1) + (0 if p.reuse_output else 1)) * R * V * x.element_size()
        out[name] = {"ms_per_call": ms, "algorithmic_bytes": nbytes,
                     "achieved_GBps": nbytes / ms / 1e6 if nbytes else None,
                     "frac_of_peak": nbytes / ms / 1e6 / HBM_PEAK_GBS if nbytes else None}
        p._kept.clear()
        del res, x
    for p in procs:
        p.check_errors()
    return out


def time_read_probe(wl, device, launches=20):
    """Read-only pass over one of the logit buffers with the streaming kernel's access pattern
    (grx_read_probe): the throughput this GPU gives that pattern, timed by the same dispatch events."""
    import ctypes as C

    import torch

    from genrec_amd import _lib

    lib = _lib.lib()
    x = wl["logits"][min(1, len(wl["logits"]) - 1)]
    R, row_bytes = x.shape[0], x.shape[1] * x.element_size()
    out = torch.empty(R * 4, dtype=torch.int32, device=device)
    stream = _lib.current_stream_ptr(device)
    for i in range(launches + 2):
        if i == 2:
            torch.cuda.synchronize()
            lib.grx_timing_enable(1)
        _lib.check(lib.grx_read_probe(x.data_ptr(), x.stride(0) * x.element_size(), R, row_bytes, out.data_ptr(), stream))
    torch.cuda.synchronize()
    tot, n = C.c_double(), C.c_int32()
    _lib.check(lib.grx_timing_read(C.byref(tot), C.byref(n)))
    lib.grx_timing_enable(0)
    ms = tot.value / max(n.value, 1)
    return {"ms": ms, "GBps": R * row_bytes / ms / 1e6, "launches": n.value}


SETTLED_GBPS = 6900.0  # read-probe rate of an MI355X that nothing else is using (measured: 6.93-7.05 TB/s)


def settle(wl, device, patience_s=10.0, cap_s=45.0):
    """Wait, before the timed region, until the GPU's read rate is its own again.

    After a process that held a lot of HBM exits (a test suite, a previous benchmark) the driver goes on
    working on the released memory for seconds, and whatever runs meanwhile reads 3-4 % slower
    (`tools/churn.py memory` followed by `tools/settle_probe.py` shows the step back up after ~6 s).
    The read probe is sampled twice a second until four samples in a row reach the rate of an idle part and have
    stopped climbing, or it has not improved by 0.5 % for `patience_s`, or `cap_s` have passed.  What was seen goes into the result line.
    """
    x = wl["logits"][min(1, len(wl["logits"]) - 1)]
    if x.numel() * x.element_size() < (1 << 30):
        return None  # a launch this small never reaches the idle rate: nothing to compare against
    t0 = time.perf_counter()
    first = best = last = time_read_probe(wl, device, launches=6)["GBps"]
    best_at = 0.0
    seen = [first]

    def settled():
        # four samples in a row at the idle rate (the background work comes in bursts) that are no longer
        # climbing: their mean within 0.3 % of the four before them
        if len(seen) < 8 or min(seen[-4:]) < SETTLED_GBPS:
            return False
        return sum(seen[-4:]) <= 1.003 * sum(seen[-8:-4])

    while not settled():
        now = time.perf_counter() - t0
        if (last < SETTLED_GBPS and now - best_at > patience_s) or now > cap_s:
            break
        time.sleep(0.5)
        last = time_read_probe(wl, device, launches=6)["GBps"]
        seen.append(last)
        if last > best * 1.005:
            best, best_at = last, time.perf_counter() - t0
    return {"waited_s": round(time.perf_counter() - t0, 2), "probe_GBps_first": round(first, 1),
            "probe_GBps_last": round(last, 1), "target_GBps": SETTLED_GBPS}


def baseline_metric() -> str:
    """The metric's name exactly as BASELINE.json words it (its second half, the share of the HBM
    roofline, is `roofline.frac` of the same line)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json"), encoding="utf-8") as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "constrained decode steps/sec (batch\u00d7beam=8192, V\u2248152k); % HBM roofline"


KERNEL_SOURCES = ("genrec_amd/csrc/masked_logsoftmax.hip", "genrec_amd/csrc/lse_math.h", "genrec_amd/csrc/common.h",
                  "genrec_amd/csrc/resolve.h", "genrec_amd/csrc/beam_body.h", "include/genrec_hip.h")


def kernel_sources_sha256() -> str:
    """Fingerprint of the sources the streaming kernel is built from: counter figures recorded for one
    version of the kernel are not reported for another."""
    import hashlib

    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def workload_key(args) -> str:
    """What identifies a workload for the recorded counter figures: everything that changes which kernel
    runs or how many bytes it moves."""
    return (f"{args.trie}|{args.processor}|batch={args.batch}|beams={args.beams}|{args.dtype}|"
            f"prompt={args.prompt_len}|penalty={args.repetition_penalty}")


HEADLINE_WORKLOAD_KEY = "Instruments|count|batch=512|beams=16|bf16|prompt=61|penalty=1.0"


def recorded_traffic(args, alg_bytes):
    """HBM bytes per launch of the dominant kernel from hardware counters.

    Counters cannot be read from inside this process: they come from separate
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes over this same
    command, whose per-launch averages are kept under profiles/ -- the headline
    workload's in r*_pmc_summary.json, the other configurations' in r*_<config>_pmc.json
    (FETCH_SIZE is in KiB and, on gfx950, counts half the bytes of a wide
    coalesced stream, hence the factor 2 applied there).  A recorded figure
    only applies to the workload and to the kernel sources (their SHA-256 is kept
    with it) it was collected on; anything else reports null.
    """
    import glob

    sha = kernel_sources_sha256()
    key = workload_key(args)
    # the headline workload's own collection first
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if files and key == HEADLINE_WORKLOAD_KEY:
        with open(files[-1]) as f:
            summary = json.load(f)
        rec = summary["masked_lsm_kernel_hbm_bytes_per_launch"]
        # counters taken on other kernel sources say nothing about this build
        if summary.get("kernel_sources_sha256") == sha and int(rec["algorithmic"]) == int(alg_bytes):
            return float(rec["total"]), os.path.relpath(files[-1], ROOT)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        with open(path) as f:
            rec = json.load(f)
        if rec.get("workload_key") == key and rec.get("kernel_sources_sha256") == sha and \
                int(rec.get("algorithmic_bytes_per_launch", -1)) == int(alg_bytes):
            return float(rec["hbm_bytes_per_launch"]["total"]), os.path.relpath(path, ROOT)
    return None, None


def launch_ranks(n: int) -> int:
    """Start `n` ranks of this script, one per GPU, the way the reference's launchers start theirs
    (trl_trainer.sh:359-364 `accelerate launch --num_processes N`; evaluate.sh:140-152 one process per GPU),
    relay rank 0's JSON line and return the exit status: non-zero when the ranks fail or print no line.

    Runs before anything in this process has initialised the GPU; the ranks are a child process.
    """
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    child = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    lines = 0
    for ln in child.stdout:
        if ln.startswith("{"):
            lines += 1
        sys.stdout.write(ln)
        sys.stdout.flush()
    rc = child.wait()
    if rc != 0:
        print(f"bench.py: the {n}-rank run exited with status {rc}", file=sys.stderr)
        return rc if 0 < rc < 256 else 1
    if lines != 1:
        print(f"bench.py: the {n}-rank run printed {lines} result lines instead of one", file=sys.stderr)
        return 1
    return 0


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=512, help="prompt groups per GPU")
    ap.add_argument("--beams", type=int, default=16)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp16"])
    ap.add_argument("--trie", default="Instruments",
                    help="Industrial_and_Scientific | Instruments | Games | Arts | cb<codebook size> (synthetic)")
    ap.add_argument("--processor", default="count", choices=["count", "fixed_hint"])
    ap.add_argument("--prompt-len", type=int, default=61)
    ap.add_argument("--repetition-penalty", type=float, default=1.0,
                    help="fold transformers' repetition penalty over the whole row into the step (evaluate.py:218 uses 1.1)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-groups", type=int, default=0,
                    help="prompt groups of the CPU baseline sample (0: all of --batch, i.e. the real row count)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-processor-surface", action="store_true", help="skip the dense [R, V] mode measurement")
    ap.add_argument("--no-plan-ahead", action="store_true",
                    help="let every processor call run its own row planning kernel (as under the plain HF loop)")
    ap.add_argument("--no-keep-top", action="store_true",
                    help="rows emit every allowed token instead of their 2 x beams best (A/B of the row truncation)")
    ap.add_argument("--graph", action="store_true",
                    help="record three generations (3 x depth steps with their scoring and re-initialisation) as one HIP "
                         "graph and replay it: the host leaves the loop (small launches are host-bound); whole replays "
                         "only, the remainder of --steps runs eagerly; no per-kernel events (roofline.kernel_ms null)")
    ap.add_argument("--no-fused-step", action="store_true",
                    help="processor call and beam step as two library calls (two kernels) instead of one (A/B)")
    ap.add_argument("--lazy-turnover", action="store_true",
                    help="score a finished generation and re-initialise the beam state at the start of the next "
                         "generation's first step instead of at the end of its own last step (A/B)")
    ap.add_argument("--cand-width", type=int, default=0, help="candidate entries per row, at least (tuning)")
    ap.add_argument("--ramp-steps", type=int, default=100,
                    help="untimed steps that precede the timed region at least (the --warmup steps count towards them)")
    ap.add_argument("--no-settle", action="store_true",
                    help="start timing at once instead of first waiting for the GPU's read rate to recover from "
                         "whatever released memory before this process")
    ap.add_argument("--no-kernel-timing", action="store_true", help="leave the per-launch HIP events out")
    ap.add_argument("--time-every", type=int, default=4,
                    help="attach HIP events to every n-th launch of the streaming kernel in the timed region (an "
                         "event pair costs ~5 us of stream time; keep n coprime with the number of decode depths so "
                         "every depth is sampled equally)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  Nothing here has touched the
        # GPU (torch is not even imported yet), and the ranks are started as a child, never by replacing
        # this process.
        raise SystemExit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; "
                         f"a line labelled {args.gpus} GPUs must come from {args.gpus} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no HIP device is visible")
    # GENREC_BENCH_BACKEND=gloo rehearses the multi-rank flow with several ranks on one GPU
    backend = os.environ.get("GENREC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # GENREC_BENCH_FORCE_DIST=1: create the process group and issue every collective even with one rank
    # (rehearses the RCCL code path of `--gpus N` on a one-GPU box)
    dist_on = world > 1 or os.environ.get("GENREC_BENCH_FORCE_DIST", "0") not in ("", "0")
    if dist_on and world == 1:
        os.environ["GENREC_AMD_SINGLE_RANK_COLLECTIVES"] = "1"
        os.environ.setdefault("MASTER_PORT", "29623")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    # who is really here: every rank reports the device it runs on, and the line carries the count
    seen = [(rank, torch.cuda.current_device(), torch.cuda.get_device_name(device))]
    if dist_on:
        seen = [None] * world
        dist.all_gather_object(seen, (rank, torch.cuda.current_device(), torch.cuda.get_device_name(device)))
    if len({r for r, _, _ in seen}) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {len(seen)} ranks answered")

    from genrec_amd import _lib

    if os.environ.get("GENREC_BENCH_LIB"):  # kernel tuning: time a variant build instead of the shipped library
        _lib.LIB_PATH = os.path.abspath(os.environ["GENREC_BENCH_LIB"])
    wl = build_workload(args, device)
    roll = Rollout(args, wl, device)

    def barrier():
        if dist_on:
            dist.barrier()

    if dist_on and backend == "nccl" and os.environ.get("GENREC_BENCH_NATIVE_RCCL", "0") not in ("", "0"):
        # the exchange step through the library's own RCCL communicator (on the compute stream) instead of
        # torch.distributed, which then only carries the communicator id
        from genrec_amd import advantage
        from genrec_amd.comm import RcclComm

        advantage.use_native_comm(RcclComm.from_process_group(device))
    if dist_on:
        # set the communicator's all-gather path up outside the timed region even when the warm-up is
        # shorter than one rollout (the first collective of a process group connects its transports)
        from genrec_amd.advantage import gather_rows

        gather_rows(torch.zeros((args.batch * args.beams, 2), dtype=torch.float32, device=device))
    # Untimed set-up, whatever --warmup says: one whole rollout and its scoring (every kernel launched
    # once, every buffer allocated), plus as many steps as it takes for the warm-up to end on a rollout
    # boundary -- the timed region then starts with a fresh generation without cutting one short.
    for _ in range(wl["levels"] + 1 + (-(args.warmup + 1)) % wl["levels"]):
        roll.step()
    torch.cuda.synchronize()
    # ... then, if this GPU is still busy with memory that other processes released, wait for it; the
    # warm-up steps come after the wait, so the timed region follows work, not sleep
    settled = None if args.no_settle else settle(wl, device)
    # ... and bring GPU and host up to speed: the wait above sleeps between its samples, and a region that
    # follows a few milliseconds of work runs ~2 % slower than one that follows 40 ms of it (clocks on both
    # sides of the bus).  At least `--ramp-steps` steps run here, in whole rollouts, ahead of the W warm-up steps.
    ramp = -(-max(0, args.ramp_steps - args.warmup) // wl["levels"]) * wl["levels"]
    for _ in range(ramp + args.warmup):
        roll.step()
    assert roll.depth == (wl["levels"] if roll.lazy_turnover else 0)  # the timed region starts with a fresh generation
    torch.cuda.synchronize()
    barrier()
    lib = _lib.lib()
    captured, per_replay = None, 3 * wl["levels"]
    if args.graph:
        from genrec_amd.graph import capture

        def three_generations():
            for _ in range(per_replay):
                roll.step()

        captured = capture(three_generations, [wl["proc"]])
        for _ in range(3):
            captured.replay()
        torch.cuda.synchronize()
        barrier()
    lib.grx_timing_enable(0 if (args.no_kernel_timing or args.graph) else max(1, args.time_every))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if captured is not None:
        for _ in range(args.steps // per_replay):
            captured.replay()
        for _ in range(args.steps % per_replay):
            roll.step()
    else:
        for _ in range(args.steps):
            roll.step()
    roll.finish()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    import ctypes as C

    tot, n = C.c_double(), C.c_int32()
    _lib.check(lib.grx_timing_read(C.byref(tot), C.byref(n)))
    lib.grx_timing_enable(0)
    wl["proc"].check_errors()

    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        R = args.batch * args.beams
        esize = 4 if args.dtype == "fp32" else 2
        kernel_ms = tot.value / n.value if n.value else None
        alg_bytes = R * wl["V"] * esize
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms else None
        traffic, traffic_source = recorded_traffic(args, alg_bytes)
        probe = time_read_probe(wl, device)
        out = {
            "metric": baseline_metric(),
            "value": world * args.steps / elapsed,
            "unit": "steps/s",
            "n_gpus": world,
            "ranks_seen": len(seen),
            "devices": [f"rank {r}: cuda:{d} {name}" for r, d, name in sorted(seen)],
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{'C2: ' if (args.trie, args.processor, args.batch) == ('Instruments', 'count', 512) else ''}"
                            f"{'C1 (3-level id2sid index): ' if args.trie == 'Industrial_and_Scientific' else ''}"
                            f"{args.trie} SID trie, {args.processor} processor, beam={args.beams}, "
                            f"batch={args.batch} per GPU (R={R} rows), V={wl['V']}, {args.dtype} logits N(0,16); "
                            f"fused step = row planning + masked log-softmax (compact candidates) + beam step; every "
                            f"{wl['levels']} steps one rollout is scored (rule+NDCG rewards, reward all-gather, "
                            f"group advantages)",
                "rows_per_step_per_gpu": R,
                "vocab": wl["V"],
                "parallelism": f"dp{world} over prompt groups",
            },
            "roofline": {
                "bound": "hbm",
                # launches of up to 512 rows carry their groups' beam steps (the library's own choice)
                "kernel": ("masked_lsm_kernel (compact output)"
                           if (args.no_fused_step or args.beams > 16 or args.repetition_penalty != 1.0
                               or R > int(os.environ.get("GRX_FUSE_MAX_ROWS", "512")))
                           else "masked_lsm_beam_kernel (compact candidates; a prompt group's beam step runs in the "
                                "workgroup that emits the group's last row)"),
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if achieved else None,
                # the same bytes over the whole step (everything `value` pays for), slowest rank
                "step_frac": alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": kernel_ms,
                "launches_timed": n.value,
                "timed_every": max(1, args.time_every),
                # the same rows read by a kernel that only loads them (same access pattern): what the part offers
                "read_probe": probe,
                "settle": settled,
                "untimed_steps_before_region": ramp + args.warmup,
                "frac_of_read_probe": achieved / probe["GBps"] if (achieved and probe["GBps"]) else None,
                "hip_graph": bool(args.graph),
            },
        }
        if world == 1 and not args.no_processor_surface:
            out["processor_surface"] = time_processor_surface(args, wl, device)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = time_cpu_baseline(args, wl)
            if out["cpu_baseline"]["cores"] > 8:  # the same sample on 8 threads, beside the whole CPU share
                at8 = time_cpu_baseline(args, wl, cores=8)
                out["cpu_baseline"]["at_8_cores"] = {"value": at8["value"], "sample": at8["sample"]}
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
