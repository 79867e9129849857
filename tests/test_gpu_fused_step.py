"""GPU: the decode step as one call (``grx_constrained_beam_step`` / ``BeamSearchState.fused_step``).

The workgroup of the streaming kernel that emits the last candidate row of a prompt group runs that
group's beam step inside the same launch.  Whatever the route -- one kernel, or the two kernels back to
back where the library cannot fuse -- every buffer of the beam state, the per-step results, the
candidate rows and the processor's counter must equal, bit for bit, what ``candidates_into`` followed
by ``step`` leave behind (the loop body of transformers' ``_beam_search``, generation/utils.py:3384-3508,
with logit_processor.py:29-49 / fixed_hint_logit_processor.py:79-96 inside it)."""

import tempfile

import numpy as np
import pytest
import torch

from genrec_amd.beam import BeamSearchState
from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor
from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows, synthetic_index, write_index
from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn
from helpers import write_index_file
from test_gpu_processors import _dev

pytestmark = pytest.mark.gpu


def _setup(trie_name, K, cls, with_prefix_ids=True, synth=(600, 3, 24, 3)):
    tmp = tempfile.mkdtemp()
    if trie_name.startswith("synth"):
        path = write_index(f"{tmp}/s.index.json", synthetic_index(synth[0], synth[1], synth[2], seed=synth[3]))
    else:
        path = write_index_file(trie_name, tmp)
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    fn = create_prefix_allowed_tokens_fn(trie, suffix)
    make = lambda: cls(fn, K, pidx, suffix if with_prefix_ids else None, tok.eos_token_id)
    levels = len(trie.level_fanout()) - len(suffix)
    return tok, trie, make, levels


def _snapshot(state, out, cand, proc_count):
    c = cand.count.clone()
    m = torch.arange(cand.cmax, device=c.device)[None, :] < c[:, None]
    big = torch.iinfo(torch.int32).max
    order = torch.where(m, cand.tokens, big).argsort(dim=1, stable=True)  # cut rows come in no particular order
    snap = {
        "run_seq": state.run_seq[state.cur].clone(), "fin_seq": state.fin_seq[state.cur].clone(),
        "run_score": state.run_score[state.cur].clone(), "fin_score": state.fin_score[state.cur].clone(),
        "run_bidx": state.run_bidx[state.cur].clone(), "fin_bidx": state.fin_bidx[state.cur].clone(),
        "fin_done": state.fin_done[state.cur].clone(), "unsat": state.unsat.clone(),
        "topk_scores": out.topk_scores.clone(), "topk_parent": out.topk_parent.clone(),
        "topk_token": out.topk_token.clone(), "topk_hit": out.topk_hit.clone(), "next_parent": out.next_parent.clone(),
        "cand_count": c, "cand_tok": torch.where(m, cand.tokens, 0).gather(1, order),
        "cand_lp": torch.where(m, cand.logp.view(torch.int32), 0).gather(1, order), "cand_lse": cand.logsumexp.clone(),
    }
    return snap, proc_count


def _run(make, tok, prompts, K, levels, logits, fused, generations=1, keep=True, plan=True):
    dev = prompts.device
    proc = make()
    rows = prompts.repeat_interleave(K, dim=0).contiguous()
    snaps = []
    state = None
    for _ in range(generations):
        proc.count = 0
        planner = proc if plan else None
        if state is None:
            state = BeamSearchState(rows, K, rows.shape[1] + levels + 1, tok.eos_token_id, tok.pad_token_id, 1.0, False,
                                    shared_prompt=True, planner=planner)
            keep_top = proc.keep_top_for(state.n_keep) if keep else 0
            cand = Candidates(state.B * K, proc.candidate_width(state.n_keep if keep else 0), dev)
        else:
            state.reset(rows, planner)
        for d in range(levels):
            x = logits[d]
            nxt = planner if d + 1 < levels else None
            if fused:
                out = state.fused_step(proc, x, cand, keep_top=keep_top, planner=nxt)
            else:
                proc.candidates_into(state.flat_running_sequences(), x, cand, keep_top=keep_top,
                                     row_bias=state.row_bias() if keep_top else None)
                out = state.step(cand, x.shape[-1], nxt)
            torch.cuda.synchronize()
            snaps.append(_snapshot(state, out, cand, None))
        proc.check_errors()
        snaps.append(({}, proc.count))
    return snaps


def _assert_same(a, b):
    assert len(a) == len(b)
    for step, ((sa, ca), (sb, cb)) in enumerate(zip(a, b)):
        assert ca == cb, step
        for k in sa:
            assert torch.equal(sa[k], sb[k]), (step, k)


@pytest.mark.parametrize("cls", [ConstrainedLogitsProcessor, FixedHintConstrainedLogitsProcessor])
@pytest.mark.parametrize("B,dtype", [(3, torch.float32), (12, torch.bfloat16), (40, torch.bfloat16)])
def test_fused_step_equals_the_two_calls(cls, B, dtype):
    """3 groups = 48 rows and 12 groups = 192 rows are split rows (several workgroups per row, the last one
    emits and counts the row in); 40 groups = 640 rows take one workgroup per row.  Three generations on one
    state and one workspace: the arrival counters must be back at zero after every call."""
    dev = _dev()
    K = 16
    tok, trie, make, levels = _setup("Instruments", K, cls)
    prompts = torch.from_numpy(sample_decode_rows(trie, B, 1, 0, prompt_len=21, seed=B)).to(dev)
    g = torch.Generator(device=dev).manual_seed(B)
    logits = [(torch.randn((B * K, tok.vocab_size_padded), generator=g, device=dev) * 4).to(dtype) for _ in range(levels)]
    want = _run(make, tok, prompts, K, levels, logits, fused=False, generations=3)
    got = _run(make, tok, prompts, K, levels, logits, fused=True, generations=3)
    _assert_same(got, want)


def test_fused_step_without_plans_made_ahead_and_without_prefix_ids():
    """No planner: every call prepares itself.  With prefix_ids the match kernel does the counter's
    bookkeeping and the step still runs as one launch; without them (evaluate.py:283-288) the streaming
    kernel would have to do it, and the call falls back on the two kernels.  Same results either way."""
    dev = _dev()
    K, B = 16, 20
    for with_prefix in (True, False):
        tok, trie, make, levels = _setup("synth", K, ConstrainedLogitsProcessor, with_prefix_ids=with_prefix)
        prompts = torch.from_numpy(sample_decode_rows(trie, B, 1, 0, prompt_len=9, seed=4)).to(dev)
        g = torch.Generator(device=dev).manual_seed(1)
        logits = [torch.randn((B * K, tok.vocab_size_padded), generator=g, device=dev) * 3 for _ in range(levels)]
        want = _run(make, tok, prompts, K, levels, logits, fused=False, generations=2, plan=False)
        got = _run(make, tok, prompts, K, levels, logits, fused=True, generations=2, plan=False)
        _assert_same(got, want)


def test_fused_step_falls_back_for_wide_searches_and_uncut_rows():
    """50 beams are beyond the instantiation that the streaming kernel carries: the call runs the two
    kernels.  And 16 beams with rows handed over whole (no keep_top)."""
    dev = _dev()
    for K, B, keep in ((50, 4, True), (16, 24, False)):
        tok, trie, make, levels = _setup("Arts", K, ConstrainedLogitsProcessor)
        prompts = torch.from_numpy(sample_decode_rows(trie, B, 1, 0, prompt_len=13, seed=2)).to(dev)
        g = torch.Generator(device=dev).manual_seed(2)
        logits = [(torch.randn((B * K, tok.vocab_size_padded), generator=g, device=dev) * 4).to(torch.bfloat16)
                  for _ in range(levels)]
        want = _run(make, tok, prompts, K, levels, logits, fused=False, keep=keep)
        got = _run(make, tok, prompts, K, levels, logits, fused=True, keep=keep)
        _assert_same(got, want)


def test_fused_step_at_the_headline_shape():
    """512 groups x 16 beams, V = 152,640, bf16: one generation, every buffer compared after every step."""
    dev = _dev()
    K, B = 16, 512
    tok, trie, make, levels = _setup("Instruments", K, ConstrainedLogitsProcessor)
    prompts = torch.from_numpy(sample_decode_rows(trie, B, 1, 0, prompt_len=61, seed=9)).to(dev)
    g = torch.Generator(device=dev).manual_seed(9)
    bufs = []
    for _ in range(2):
        x = torch.empty((B * K, tok.vocab_size_padded), dtype=torch.bfloat16, device=dev)
        for r0 in range(0, B * K, 1024):
            x[r0 : r0 + 1024] = (torch.randn((1024, x.shape[1]), generator=g, device=dev) * 4).to(torch.bfloat16)
        bufs.append(x)
    logits = [bufs[d & 1] for d in range(levels)]
    want = _run(make, tok, prompts, K, levels, logits, fused=False, generations=2)
    got = _run(make, tok, prompts, K, levels, logits, fused=True, generations=2)
    _assert_same(got, want)


def test_fused_step_refuses_a_beam_step_on_other_candidates():
    dev = _dev()
    K, B = 16, 4
    tok, trie, make, levels = _setup("synth", K, ConstrainedLogitsProcessor)
    prompts = torch.from_numpy(sample_decode_rows(trie, B, 1, 0, prompt_len=9, seed=4)).to(dev)
    rows = prompts.repeat_interleave(K, dim=0).contiguous()
    proc = make()
    state = BeamSearchState(rows, K, rows.shape[1] + levels + 1, tok.eos_token_id, tok.pad_token_id)
    cand = Candidates(B * K, trie.max_fanout, dev)
    other = Candidates(B * K, trie.max_fanout, dev)
    a, keep, _ = state._step_args(other, tok.vocab_size_padded, None, False)
    x = torch.randn((B * K, tok.vocab_size_padded), device=dev)
    from genrec_amd import _lib

    with pytest.raises(_lib.GenrecHipError, match="candidate buffer"):
        proc.candidates_into(state.flat_running_sequences(), x, cand, beam_step=a)
    assert proc.count == 0  # a refused call does not count


@pytest.mark.parametrize("seed", range(10))
def test_fused_step_on_random_searches(seed):
    """Random beam counts (1 .. 16), group counts, trie shapes, dtypes, both processors, with and without
    prefix_ids, plans made ahead or not, rows cut or whole: the one call against the two calls."""
    rng = np.random.default_rng(1000 + seed)
    dev = _dev()
    K = int(rng.integers(1, 17))
    B = int(rng.integers(1, 40))
    levels_n = int(rng.integers(2, 5))
    cb = int(rng.choice([4, 12, 40, 90]))
    cls = ConstrainedLogitsProcessor if rng.random() < 0.6 else FixedHintConstrainedLogitsProcessor
    with_prefix = bool(rng.random() < 0.6) or cls is FixedHintConstrainedLogitsProcessor
    tok, trie, make, levels = _setup("synth", K, cls, with_prefix_ids=with_prefix,
                                     synth=(int(rng.integers(30, 900)), levels_n, cb, seed))
    dtype = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 3))]
    prompts = torch.from_numpy(sample_decode_rows(trie, B, 1, 0, prompt_len=int(rng.integers(1, 40)), seed=seed)).to(dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    logits = [(torch.randn((B * K, tok.vocab_size_padded), generator=g, device=dev) * 3).to(dtype) for _ in range(levels)]
    plan, keep = bool(rng.random() < 0.7), bool(rng.random() < 0.7)
    want = _run(make, tok, prompts, K, levels, logits, fused=False, generations=2, keep=keep, plan=plan)
    got = _run(make, tok, prompts, K, levels, logits, fused=True, generations=2, keep=keep, plan=plan)
    _assert_same(got, want)


def test_one_launch_route_with_thousands_of_rows():
    """The library only fuses launches of up to 512 rows by itself (larger ones are faster as two kernels);
    GRX_FUSE_MAX_ROWS lifts the limit.  In a process of its own (the limit is read once), the headline shape
    and a 640-row shape through the one-launch route, against the two calls."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import torch; import test_gpu_fused_step as t; "
            "t.test_fused_step_at_the_headline_shape(); "
            "t.test_fused_step_equals_the_two_calls(t.ConstrainedLogitsProcessor, 40, torch.bfloat16); print('OK')"
            % (root, os.path.join(root, "tests")))
    env = dict(os.environ, GRX_FUSE_MAX_ROWS="1000000")
    run = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "OK" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]
