"""GPU parity of the constrained processors (through the C ABI) against the
golden vectors captured from the reference and against the CPU oracle."""

import numpy as np
import pytest
import torch

from helpers import load_json_gz, load_npz, write_index_file

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-5   # log-probs from fp32 logits
BF16_TOL = 2e-3   # log-probs from bf16 logits


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda", 0)


def _small_setup():
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    data, meta = load_npz("processors.npz")
    trie = FlatTrie(data["trie_sequences"].tolist())
    fn = create_prefix_allowed_tokens_fn(trie, meta["prefix"])
    return data, meta, trie, fn


def _assert_dense_close(got: torch.Tensor, want: np.ndarray, tol: float):
    got = got.float().cpu().numpy()
    finite_w = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), finite_w), "allowed-token sets differ"
    assert np.all(got[~finite_w] == -np.inf)
    assert np.max(np.abs(got[finite_w] - want[finite_w])) <= tol


def _make(case, fn, meta, **kw):
    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor

    K, prefix, eos = meta["K"], meta["prefix"], meta["eos"]
    if case == "count_rl" or case == "count_midloop_reset":
        return ConstrainedLogitsProcessor(fn, K, 3, prefix, eos, **kw)
    if case == "count_eval":
        return ConstrainedLogitsProcessor(fn, K, 3, None, eos, **kw)
    if case == "count_seeded2":
        return ConstrainedLogitsProcessor(fn, K, 3, prefix, eos, initial_count=2, **kw)
    if case == "fixed_hint":
        return FixedHintConstrainedLogitsProcessor(fn, K, 3, prefix, eos, **kw)
    raise KeyError(case)


@pytest.mark.parametrize("case", ["count_rl", "count_eval", "count_seeded2", "count_midloop_reset", "fixed_hint"])
def test_processor_matches_reference_golden(case):
    data, meta, trie, fn = _small_setup()
    dev = _dev()
    proc = _make(case, fn, meta)
    info = meta["cases"][case]
    for i in range(info["steps"]):
        ids = torch.from_numpy(data[f"{case}.ids.{i}"]).to(dev)
        scores = torch.from_numpy(data[f"{case}.scores.{i}"]).to(dev)
        before = scores.clone()
        out = proc(ids, scores)
        assert out.shape == scores.shape and out.dtype == scores.dtype and out.data_ptr() != scores.data_ptr()
        assert torch.equal(scores, before), "inputs must not be modified"
        _assert_dense_close(out, data[f"{case}.out.{i}"], FP32_TOL)
        if info["counts_after"][i] >= 0:
            assert proc.count == info["counts_after"][i]
    proc.check_errors()


@pytest.mark.parametrize("case", ["count_rl", "fixed_hint"])
def test_candidates_match_dense_golden(case):
    """The fused (compact) entry point returns exactly the finite columns of the dense result."""
    data, meta, trie, fn = _small_setup()
    dev = _dev()
    proc = _make(case, fn, meta)
    for i in range(meta["cases"][case]["steps"]):
        ids = torch.from_numpy(data[f"{case}.ids.{i}"]).to(dev)
        scores = torch.from_numpy(data[f"{case}.scores.{i}"]).to(dev)
        cand = proc.candidates(ids, scores)
        want = data[f"{case}.out.{i}"]
        count, tok, logp = cand.count.cpu().numpy(), cand.tokens.cpu().numpy(), cand.logp.cpu().numpy()
        lse = torch.logsumexp(scores.double(), dim=-1).cpu().numpy()
        assert np.max(np.abs(cand.logsumexp.cpu().numpy() - lse)) <= FP32_TOL
        for r in range(want.shape[0]):
            cols = np.nonzero(np.isfinite(want[r]))[0]
            assert count[r] == len(cols)
            assert np.array_equal(tok[r, : count[r]], cols)  # bit-exact token ids, ascending
            assert np.max(np.abs(logp[r, : count[r]] - want[r, cols])) <= FP32_TOL


def test_reference_error_conventions():
    data, meta, trie, fn = _small_setup()
    dev = _dev()
    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor

    K, prefix, eos, V = meta["K"], meta["prefix"], meta["eos"], meta["V"]
    bad = torch.from_numpy(data["bad_rows"]).to(dev)
    scores = torch.randn(bad.shape[0], V, device=dev)
    want = {e["processor"]: e for e in meta["errors"]}

    proc = ConstrainedLogitsProcessor(fn, K, 3, None, eos, sync_errors=True)
    proc.count = 1
    with pytest.raises(AssertionError, match="No valid tokens for prefix_ids"):
        proc(bad, scores)
    assert want["count"]["type"] == "AssertionError"

    fixed = FixedHintConstrainedLogitsProcessor(fn, K, 3, prefix, eos, sync_errors=True)
    with pytest.raises(ValueError, match="Failed to find a valid trie prefix match"):
        fixed(bad, scores)
    assert want["fixed"]["type"] == "ValueError"
    nop = torch.from_numpy(data["noprefix_rows"]).to(dev)
    with pytest.raises(ValueError, match="Failed to find a valid trie prefix match"):
        fixed(nop, scores)

    # the default: the processor surface raises inside the call, like the reference; the fused form defers
    dflt = ConstrainedLogitsProcessor(fn, K, 3, None, eos)
    dflt.count = 1
    with pytest.raises(AssertionError, match="No valid tokens for prefix_ids"):
        dflt(bad, scores)
    dflt.count = 1
    dflt.candidates(bad, scores)
    with pytest.raises(AssertionError, match="No valid tokens for prefix_ids"):
        dflt.reset()
    with pytest.raises(ValueError, match="Failed to find a valid trie prefix match"):
        FixedHintConstrainedLogitsProcessor(fn, K, 3, prefix, eos)(bad, scores)

    # lazy mode: the failure surfaces at reset() / count
    lazy = ConstrainedLogitsProcessor(fn, K, 3, None, eos, sync_errors=False)
    lazy.count = 1
    lazy(bad, scores)
    with pytest.raises(AssertionError):
        lazy.reset()
    lazy.reset()  # the error was consumed
    assert lazy.count == 0

    with pytest.raises(AssertionError, match="input_ids must be a 2D tensor"):
        proc(bad[0], scores)
    with pytest.raises(RuntimeError):
        proc(bad[:5], scores[:5])  # rows not a multiple of num_beams
    with pytest.raises(Exception, match="GPU"):
        proc(bad.cpu(), scores.cpu())  # no CPU path


def test_count_attribute_protocol():
    """reset()/count as MIMIGenRec.generate and evaluate.py drive them
    (reference: MIMIGenRec.py:26-33, evaluate.py:358-361)."""
    data, meta, trie, fn = _small_setup()
    dev = _dev()
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor

    proc = ConstrainedLogitsProcessor(fn, meta["K"], 3, None, meta["eos"])
    assert hasattr(proc, "count") and proc.count == 0
    ids = torch.from_numpy(data["count_eval.ids.0"]).to(dev)
    scores = torch.from_numpy(data["count_eval.scores.0"]).to(dev)
    proc(ids, scores)
    assert proc.count == 1
    proc.count = 0
    assert proc.count == 0
    proc(ids, scores)
    proc.reset()
    assert proc.count == 0


def test_full_vocab_rows_fp32_and_bf16(tmp_path):
    """Real Instruments trie at V=152,640: allowed ids bit-exact, log-probs within tolerance."""
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer
    from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn

    dev = _dev()
    data, meta = load_npz("full_vocab_rows.npz")
    path = write_index_file("Instruments", tmp_path)
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    tries = load_json_gz("tries.json.gz")["Instruments"]
    assert (trie.n_nodes, trie.max_fanout, tok.vocab_size_padded) == (tries["n_nodes"], tries["max_fanout"], meta["V"])
    fn = create_prefix_allowed_tokens_fn(trie, suffix)
    K, V = meta["K"], meta["V"]
    p32 = ConstrainedLogitsProcessor(fn, K, pidx, suffix, tok.eos_token_id)
    p16 = ConstrainedLogitsProcessor(fn, K, pidx, suffix, tok.eos_token_id)

    def check(tk, lp, n, want_tok, want_lp, exact, tol):
        """Token ids bit-exact.  Log-probs: within `tol` of the exactly evaluated
        log-softmax, and within `tol` + the reference's own fp32 summation error of the
        captured reference output (PyTorch-CPU sums 152k terms sequentially per SIMD lane
        and lands 1-2e-5 away from the float64 value at this width)."""
        tk, lp = tk.cpu().numpy(), lp.cpu().numpy()
        for r in range(tk.shape[0]):
            cols = want_tok[r, : n[r]]
            assert np.array_equal(tk[r, : n[r]], cols)
            ex = exact[r, cols]
            ref_err = float(np.max(np.abs(want_lp[r, : n[r]] - ex)))
            assert np.max(np.abs(lp[r, : n[r]] - ex)) <= tol
            assert np.max(np.abs(lp[r, : n[r]] - want_lp[r, : n[r]])) <= tol + ref_err

    for step in range(meta["steps"]):
        ids = torch.from_numpy(data[f"ids.{step}"]).to(dev)
        gen = torch.Generator().manual_seed(meta["seed"] + step)
        logits = torch.randn(ids.shape[0], V, generator=gen) * meta["scale"]
        n = data[f"count.{step}"]
        c32 = p32.candidates(ids, logits.to(dev), cmax=trie.max_fanout)
        count, tk, lp = c32.count, c32.tokens, c32.logp
        assert np.array_equal(count.cpu().numpy(), n)
        exact = torch.log_softmax(logits.double(), dim=-1).numpy()
        check(tk, lp, n, data[f"tok.{step}"], data[f"logp.{step}"], exact, FP32_TOL)

        l16 = logits.to(torch.bfloat16)
        cb16 = p16.candidates(ids, l16.to(dev), cmax=trie.max_fanout)
        c16, t16, lp16 = cb16.count, cb16.tokens, cb16.logp
        assert np.array_equal(c16.cpu().numpy(), n)
        exact16 = torch.log_softmax(l16.double(), dim=-1).numpy()
        check(t16, lp16, n, data[f"tok.{step}"], data[f"logp_bf16.{step}"], exact16, BF16_TOL)
        # for bf16 logits the fp32 tolerance holds as well: the inputs are exactly representable
        check(t16, lp16, n, data[f"tok.{step}"], data[f"logp_bf16.{step}"], exact16, FP32_TOL)

        # dense surface on bf16 input keeps the dtype
        dense = ConstrainedLogitsProcessor(fn, K, pidx, None, tok.eos_token_id)
        dense.count = step
        out = dense(ids, l16.to(dev))
        assert out.dtype == torch.bfloat16
        assert np.array_equal(torch.isfinite(out).sum(dim=1).cpu().numpy(), n)
    p32.check_errors()
    p16.check_errors()


@pytest.mark.parametrize("V", [1000, 1001, 1023, 4099])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_ragged_vocab_widths_against_oracle(V, dtype):
    """Rows whose width breaks 16-byte alignment take the element-wise path; results match the oracle."""
    from oracle.processors import DictTrie, RefConstrainedLogitsProcessor

    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    data, meta, _, _ = _small_setup()
    dev = _dev()
    seqs = data["trie_sequences"].tolist()
    trie = FlatTrie(seqs)
    fn = create_prefix_allowed_tokens_fn(trie, meta["prefix"])
    o_trie = DictTrie(seqs)
    K = meta["K"]
    ids = torch.from_numpy(data["count_rl.ids.1"])
    torch.manual_seed(V)
    scores = (torch.randn(ids.shape[0], V) * 5).to(dtype)
    ref = RefConstrainedLogitsProcessor(o_trie.get, K, 3, meta["prefix"], meta["eos"])
    ref.count = 1
    want = ref(ids, scores.float())
    proc = ConstrainedLogitsProcessor(fn, K, 3, meta["prefix"], meta["eos"])
    proc.count = 1
    got = proc(ids.to(dev), scores.to(dev))
    assert got.dtype == dtype
    tol = FP32_TOL if dtype == torch.float32 else (BF16_TOL if dtype == torch.float16 else None)
    got32 = got.float().cpu()
    assert torch.equal(torch.isfinite(got32), torch.isfinite(want))
    if tol is None:  # bf16 output: one bf16 rounding of the fp32 value
        assert torch.equal(got32[torch.isfinite(want)], want[torch.isfinite(want)].to(torch.bfloat16).float()) or (
            (got32 - want)[torch.isfinite(want)].abs().max() <= 0.07
        )
    else:
        lim = tol if dtype == torch.float32 else 1e-2  # fp16 output rounding
        assert (got32 - want)[torch.isfinite(want)].abs().max() <= lim
    proc.count = 1  # the dense call above advanced the counter
    cb = proc.candidates(ids.to(dev), scores.to(dev))
    c, t, lp = cb.count, cb.tokens, cb.logp
    for r in range(ids.shape[0]):
        cols = torch.nonzero(torch.isfinite(want[r])).flatten()
        assert c[r].item() == cols.numel()
        assert torch.equal(t[r, : cols.numel()].cpu().long(), cols)
        assert (lp[r, : cols.numel()].cpu() - want[r, cols]).abs().max() <= FP32_TOL


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_caller_owned_output_padded_rows_and_fp32_results(dtype):
    """`call_into`: padded output rows (row-wise fill path) and fp32 log-probs from bf16 logits equal the
    values of the plain call / the candidates."""
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    data, meta, _, _ = _small_setup()
    dev = _dev()
    trie = FlatTrie(data["trie_sequences"].tolist())
    fn = create_prefix_allowed_tokens_fn(trie, meta["prefix"])
    K, V = meta["K"], 4096
    ids = torch.from_numpy(data["count_rl.ids.1"]).to(dev)
    R = ids.shape[0]
    torch.manual_seed(3)
    scores = (torch.randn(R, V) * 5).to(dtype).to(dev)
    proc = ConstrainedLogitsProcessor(fn, K, 3, meta["prefix"], meta["eos"], sync_errors=True)
    proc.count = 1
    plain = proc(ids, scores)
    for pad in (0, 8, 5):  # 5 breaks the 16-byte row alignment: element-wise path
        backing = torch.full((R, V + pad), 7.0, dtype=dtype, device=dev)
        proc.count = 1
        got = proc.call_into(ids, scores, backing[:, :V])
        assert got.data_ptr() == backing.data_ptr()
        if pad == 5:  # other summation order: same values up to rounding
            assert torch.equal(torch.isfinite(got), torch.isfinite(plain))
            fin = torch.isfinite(plain)
            assert float((got.float()[fin] - plain.float()[fin]).abs().max()) <= (FP32_TOL if dtype == torch.float32 else 0.07)
        else:
            assert torch.equal(got, plain)
        if pad:
            assert bool((backing[:, V:] == 7.0).all()), "padding columns must stay untouched"
    out32 = torch.empty((R, V), dtype=torch.float32, device=dev)
    proc.count = 1
    proc.call_into(ids, scores, out32)
    proc.count = 1
    cand = proc.candidates(ids, scores)
    finite = torch.isfinite(out32)
    assert torch.equal(finite.sum(dim=1).int(), cand.count)
    valid = torch.arange(cand.cmax, device=dev)[None, :] < cand.count[:, None]
    assert torch.equal(out32[finite], cand.logp[valid])  # same fp32 values, row-major token order
    assert torch.equal(finite, torch.isfinite(plain))
    with pytest.raises(ValueError):
        proc.call_into(ids, scores, torch.empty((R, V), dtype=torch.float16, device=dev))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_few_rows_are_split_across_workgroups_with_the_same_results(dtype):
    """Launches of few rows stream each row with several workgroups and merge partial (max, sum)
    pairs; the same rows inside a large launch take the one-workgroup path.  Tokens are identical,
    log-probs agree to rounding and with float64, and back-to-back calls of different sizes share
    one scratch buffer."""
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows
    from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn
    from helpers import write_index_file
    import tempfile

    dev = _dev()
    path = write_index_file("Instruments", tempfile.mkdtemp())
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    fn = create_prefix_allowed_tokens_fn(trie, suffix)
    V, K = tok.vocab_size_padded, 16
    proc = ConstrainedLogitsProcessor(fn, K, pidx, None, tok.eos_token_id, sync_errors=True)
    big_ids = torch.from_numpy(sample_decode_rows(trie, 40, K, 2, seed=9)).to(dev)      # 640 rows: not split
    torch.manual_seed(11)
    big = (torch.randn(640, V, device=dev) * 4).to(dtype)
    big[3, 17] = 60.0  # a dominant logit far from the row's allowed tokens
    proc.count = 2
    want = proc.candidates(big_ids, big)
    exact = torch.logsumexp(big.double(), dim=-1)
    assert float((want.logsumexp.double() - exact).abs().max()) <= FP32_TOL
    for R in (512, 256, 16, 64, 400, 16):  # descending and ascending: the shared scratch is left clean every time
        proc.count = 2
        got = proc.candidates(big_ids[:R], big[:R])
        assert torch.equal(got.count, want.count[:R])
        valid = torch.arange(got.cmax, device=dev)[None, :] < got.count[:, None]
        assert torch.equal(got.tokens[valid], want.tokens[:R][valid])
        assert float((got.logsumexp.double() - exact[:R]).abs().max()) <= FP32_TOL
        assert float((got.logp[valid] - want.logp[:R][valid]).abs().max()) <= 4e-6
        proc.count = 2
        dense = proc(big_ids[:R], big[:R])
        fin = torch.isfinite(dense)
        assert torch.equal(fin.sum(dim=1).int(), want.count[:R])
        ref = want.logp[:R][valid]
        err = (dense[fin].float() - ref).abs()
        if dtype == torch.float32:
            assert float(err.max()) <= 4e-6
        else:  # one bf16 rounding of the fp32 value: half an ulp = 2^-9 relative
            assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-6).all())
    # split rows whose width leaves an element tail behind the last whole vector (handled by the last part)
    Vr = V - 3
    ragged = big[:16, :Vr]
    ragged[5, Vr - 1] = 70.0  # the row's maximum sits in the tail
    exact_r = torch.logsumexp(ragged.double(), dim=-1)
    proc.count = 2
    got = proc.candidates(big_ids[:16], ragged)
    assert float((got.logsumexp.double() - exact_r).abs().max()) <= FP32_TOL


def test_fused_repetition_penalty_matches_the_transformers_chain():
    """repetition_penalty= folds transformers' RepetitionPenaltyLogitsProcessor (applied to
    log_softmax(logits), as generate() does) into the streaming pass.  Allowed ids are bit-exact; values
    are within 1e-5 of a float64 evaluation of the chain and within 1e-5 (+ the chain's own fp32 error)
    of the recorded fp32 chain; bf16 logits within 2e-3."""
    import tempfile

    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer
    from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn
    from helpers import write_index_file

    dev = _dev()
    data, meta = load_npz("penalty.npz")
    path = write_index_file("Instruments", tempfile.mkdtemp())
    tok = SidTokenizer.from_index(path)
    trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    fn = create_prefix_allowed_tokens_fn(trie, suffix)
    K, V, eos = meta["K"], meta["V"], meta["eos"]
    assert sum(c["penalised_allowed"] for c in meta["cases"]) >= 40
    for ci, case in enumerate(meta["cases"]):
        ids = torch.from_numpy(data[f"{ci}.ids"])
        logits = torch.randn(ids.shape[0], V, generator=torch.Generator().manual_seed(case["seed"])) * meta["scale"]
        if case["peaked"]:
            logits[torch.arange(ids.shape[0]), ids[:, 7]] += 40.0
        kw = dict(repetition_penalty=case["penalty"], prompt_ignore_length=case["skip"], sync_errors=True)
        if case["processor"] == "count":
            proc = ConstrainedLogitsProcessor(fn, K, pidx, None, eos, **kw)
        else:
            proc = FixedHintConstrainedLogitsProcessor(fn, K, pidx, suffix, eos, **kw)
        n = torch.from_numpy(data[f"{ci}.count"])
        valid = torch.arange(data[f"{ci}.tok"].shape[1])[None, :] < n[:, None]
        want_tok = torch.from_numpy(data[f"{ci}.tok"])[valid]
        want = torch.from_numpy(data[f"{ci}.val"])[valid]
        exact = torch.from_numpy(data[f"{ci}.val64"])[valid]
        for dtype, tol in ((torch.float32, FP32_TOL), (torch.bfloat16, BF16_TOL)):
            x = logits.to(dtype).to(dev)
            if hasattr(proc, "reset"):
                proc.count = case["step"]
            cand = proc.candidates(ids.to(dev), x)
            assert torch.equal(cand.count.cpu(), n)
            cvalid = torch.arange(cand.cmax)[None, :] < n[:, None]
            assert torch.equal(cand.tokens.cpu()[cvalid], want_tok)
            got = cand.logp.cpu()[cvalid]
            if dtype == torch.float32:
                assert float((got.double() - exact).abs().max()) <= tol
                assert bool(((got - want).abs() <= tol + (want.double() - exact).abs().float()).all())
            else:
                lp64 = torch.log_softmax(logits.to(dtype).double(), dim=-1)
                from oracle.processors import ref_repetition_penalty

                e16 = torch.log_softmax(ref_repetition_penalty(ids, lp64, case["penalty"], case["skip"]), dim=-1)
                rows = torch.arange(ids.shape[0])[:, None].expand_as(cvalid)[cvalid]
                assert float((got.double() - e16[rows, want_tok.long()]).abs().max()) <= FP32_TOL
                assert float((got - want).abs().max()) <= 0.2  # bf16 logits: coarse agreement with the fp32 chain
            if hasattr(proc, "reset"):
                proc.count = case["step"]
            dense = proc(ids.to(dev), x).float().cpu()
            fin = torch.isfinite(dense)
            assert torch.equal(fin.sum(dim=1), n.long())
            if dtype == torch.float32:
                # dense and compact forms carry the same values (up to how each instantiation fuses a multiply-add)
                assert float((dense[fin] - got).abs().max()) <= 4e-6
    with pytest.raises(ValueError):
        ConstrainedLogitsProcessor(fn, K, pidx, None, eos, repetition_penalty=0.0)


def test_empty_and_degenerate_batches_follow_the_reference():
    """No rows, rows shorter than the window, rows without tokens, one-beam groups: same results,
    same counter movement and same exceptions as the oracle."""
    from oracle.processors import (DictTrie, RefConstrainedLogitsProcessor,
                                   RefFixedHintConstrainedLogitsProcessor)

    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    data, meta, _, _ = _small_setup()
    dev = _dev()
    seqs = data["trie_sequences"].tolist()
    trie = FlatTrie(seqs)
    fn = create_prefix_allowed_tokens_fn(trie, meta["prefix"])
    o_get = DictTrie(seqs).get
    prefix, eos, V = meta["prefix"], meta["eos"], 2048

    def pair(kind, K):
        if kind == "count":
            return (ConstrainedLogitsProcessor(fn, K, 3, prefix, eos, sync_errors=True),
                    RefConstrainedLogitsProcessor(o_get, K, 3, prefix, eos))
        return (FixedHintConstrainedLogitsProcessor(fn, K, 3, prefix, eos, sync_errors=True),
                RefFixedHintConstrainedLogitsProcessor(o_get, K, 3, prefix, eos))

    def same(got, want):
        assert got.shape == want.shape
        assert torch.equal(torch.isfinite(got.cpu()), torch.isfinite(want))
        fin = torch.isfinite(want)
        if fin.any():
            assert float((got.cpu()[fin] - want[fin]).abs().max()) <= FP32_TOL

    torch.manual_seed(1)
    # no rows at all: nothing to emit, the shared counter still advances (logit_processor.py:48)
    ours, ref = pair("count", 4)
    out = ours(torch.zeros((0, 7), dtype=torch.int64, device=dev), torch.zeros((0, V), device=dev))
    assert out.shape == (0, V)
    ref(torch.zeros((0, 7), dtype=torch.int64), torch.zeros((0, V)))
    assert ours.count == ref.count == 1
    assert ours.candidates(torch.zeros((0, 7), dtype=torch.int64, device=dev), torch.zeros((0, V), device=dev)).count.numel() == 0
    # one-beam groups, rows that are exactly the prefix / shorter than the window / empty
    for kind in ("count", "fixed"):
        for ids in (torch.tensor([prefix, prefix]), torch.tensor([prefix[1:]]), torch.zeros((2, 0), dtype=torch.int64)):
            ours, ref = pair(kind, 1)
            scores = torch.randn(ids.shape[0], V) * 3
            try:
                want = ref(ids, scores)
            except (AssertionError, ValueError, RuntimeError) as e:
                with pytest.raises(type(e)):
                    ours(ids.to(dev), scores.to(dev))
                continue
            same(ours(ids.to(dev), scores.to(dev)), want)
    # a row count that does not split into groups fails like the reference's view() does
    ours, ref = pair("count", 4)
    with pytest.raises(RuntimeError):
        ref(torch.tensor([prefix] * 3), torch.zeros(3, V))
    with pytest.raises(RuntimeError):
        ours(torch.tensor([prefix] * 3).to(dev), torch.zeros(3, V, device=dev))
    # the trie's largest token must be a column of the scores
    ours, _ = pair("count", 1)
    with pytest.raises(Exception):
        ours(torch.tensor([prefix]).to(dev), torch.zeros(1, 8, device=dev))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("V", [2041, 2044, 2047])
def test_aligned_rows_with_a_ragged_width(V, dtype):
    """Row starts are 16-byte aligned (views into wider buffers) but the width is not a multiple of
    the vector width: the vector stream covers the whole vectors, an element-wise tail the rest, for
    the dense fill as well."""
    from oracle.processors import DictTrie, RefConstrainedLogitsProcessor

    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    data, meta, _, _ = _small_setup()
    dev = _dev()
    seqs = data["trie_sequences"].tolist()
    assert max(t for s_ in seqs for t in s_) < V
    fn = create_prefix_allowed_tokens_fn(FlatTrie(seqs), meta["prefix"])
    K = meta["K"]
    ids = torch.from_numpy(data["count_rl.ids.1"])
    R = ids.shape[0]
    torch.manual_seed(V)
    wide = (torch.randn(R, 2048) * 5).to(dtype)
    wide[:, -1] = 50.0  # a dominant value in the last (tail) column where it exists
    ref = RefConstrainedLogitsProcessor(DictTrie(seqs).get, K, 3, meta["prefix"], meta["eos"])
    ref.count = 1
    want = ref(ids, wide[:, :V].float())
    proc = ConstrainedLogitsProcessor(fn, K, 3, meta["prefix"], meta["eos"], sync_errors=True)
    scores = wide.to(dev)[:, :V]
    assert scores.stride(0) == 2048
    backing = torch.full((R, 2048), 3.0, dtype=torch.float32, device=dev)
    proc.count = 1
    got = proc.call_into(ids.to(dev), scores, backing[:, :V]).cpu()
    fin = torch.isfinite(want)
    assert torch.equal(torch.isfinite(got), fin)
    assert bool((backing[:, V:] == 3.0).all())
    assert float((got[fin] - want[fin]).abs().max()) <= FP32_TOL * 2
    proc.count = 1
    cand = proc.candidates(ids.to(dev), scores)
    valid = torch.arange(cand.cmax, device=dev)[None, :] < cand.count[:, None]
    assert torch.equal(cand.count.cpu().long(), fin.sum(dim=1))
    assert float((cand.logp[valid].cpu() - want[fin]).abs().max()) <= FP32_TOL * 2


def test_strided_inputs_and_unnormalised_extremes():
    """Non-contiguous views, very peaked rows and rows containing -inf."""
    from oracle.processors import DictTrie, RefFixedHintConstrainedLogitsProcessor

    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    data, meta, _, _ = _small_setup()
    dev = _dev()
    seqs = data["trie_sequences"].tolist()
    fn = create_prefix_allowed_tokens_fn(FlatTrie(seqs), meta["prefix"])
    o_trie = DictTrie(seqs)
    K, V = meta["K"], meta["V"]
    ids = torch.from_numpy(data["fixed_hint.ids.1"])
    R = ids.shape[0]
    torch.manual_seed(3)
    big = torch.randn(R, V + 64) * 30.0
    big[0, :10] = float("-inf")
    big[1, 5] = 80.0
    big[2] = -1e4
    scores = big[:, 32 : 32 + V]  # row stride V + 64, 16-byte aligned start
    wide_ids = torch.zeros(R, ids.shape[1] + 5, dtype=torch.int64)
    wide_ids[:, : ids.shape[1]] = ids
    ref = RefFixedHintConstrainedLogitsProcessor(o_trie.get, K, 3, meta["prefix"], meta["eos"])
    want = ref(ids, scores.contiguous())
    proc = FixedHintConstrainedLogitsProcessor(fn, K, 3, meta["prefix"], meta["eos"], sync_errors=True)
    got = proc(wide_ids.to(dev)[:, : ids.shape[1]], big.to(dev)[:, 32 : 32 + V]).cpu()
    assert torch.equal(torch.isfinite(got), torch.isfinite(want))
    fin = torch.isfinite(want)
    assert (got - want)[fin].abs().max() <= 2e-5 * max(1.0, float(want[fin].abs().max()) / 10)


def test_a_plan_made_ahead_and_then_dropped_leaves_no_trace():
    """A beam kernel plans the next count-based call for prompt rows (which end in prefix_ids, so the
    plan records a counter reset); the caller then processes other rows instead.  The call that
    runs must not see the dropped plan's reset: results and counter as if it had never been made."""
    from genrec_amd.beam import BeamSearchState

    data, meta, trie, fn = _small_setup()
    dev = _dev()
    K, eos = meta["K"], meta["eos"]
    case = "count_rl"
    prompts = torch.from_numpy(data[f"{case}.ids.0"]).to(dev)
    ids2 = torch.from_numpy(data[f"{case}.ids.2"]).to(dev)
    scores2 = torch.from_numpy(data[f"{case}.scores.2"]).to(dev)
    assert meta["cases"][case]["counts_after"][1] == 2  # the golden step 2 ran with count == 2

    for drops in (1, 2):  # one dropped plan, and a dropped plan replaced by another dropped plan
        proc = _make(case, fn, meta, sync_errors=True)
        for _ in range(drops):
            BeamSearchState(prompts, K, prompts.shape[1] + 8, eos, 0, 1.0, False, planner=proc)
            assert proc._planned is not None
        proc.count = 2
        out = proc(ids2, scores2)
        _assert_dense_close(out, data[f"{case}.out.2"], FP32_TOL)
        assert proc.count == 3
        proc.check_errors()


def test_subclassing_the_way_the_hint_analysis_script_does():
    """analyze_rl_beam_hint.py:553-560 derives a seeded processor from ConstrainedLogitsProcessor by
    assigning ``count`` in __init__ and overriding reset(); the same subclass of our class must behave
    like the built-in ``initial_count=`` (golden case count_seeded2)."""
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor

    class Seeded(ConstrainedLogitsProcessor):
        def __init__(self, *args, initial_count=0, **kwargs):
            super().__init__(*args, **kwargs)
            self.initial_count = initial_count
            self.count = initial_count

        def reset(self):
            self.count = self.initial_count

    data, meta, trie, fn = _small_setup()
    dev = _dev()
    case = "count_seeded2"
    proc = Seeded(prefix_allowed_tokens_fn=fn, num_beams=meta["K"], prefix_index=3, prefix_ids=meta["prefix"],
                  eos_token_id=meta["eos"], initial_count=2)
    info = meta["cases"][case]
    for rep in range(2):
        for i in range(info["steps"]):
            ids = torch.from_numpy(data[f"{case}.ids.{i}"]).to(dev)
            scores = torch.from_numpy(data[f"{case}.scores.{i}"]).to(dev)
            _assert_dense_close(proc(ids, scores), data[f"{case}.out.{i}"], FP32_TOL)
            if info["counts_after"][i] >= 0:
                assert proc.count == info["counts_after"][i]
        proc.reset()
        assert proc.count == 2
    proc.check_errors()


def test_step_entry_points_reject_bad_operands_before_launching():
    """With a real trie and state in hand: operands that do not fit what the kernels assume are
    refused on the host (status + message), never launched."""
    import ctypes as C

    from genrec_amd import _lib

    data, meta, trie, fn = _small_setup()
    dev = _dev()
    lib = _lib.lib()
    proc = _make("count_rl", fn, meta)
    V, K = meta["V"], meta["K"]
    R, T = 2 * K, 6
    ids = torch.zeros((R, T), dtype=torch.int64, device=dev)
    x = torch.zeros((R, V), dtype=torch.float32, device=dev)
    handle, st = trie.device_handle(dev), proc._state(0).ptr
    ws = torch.zeros(int(lib.grx_workspace_bytes(R)) + 16, dtype=torch.uint8, device=dev)
    stride = int(lib.grx_candidate_stride(trie.max_fanout))
    cand = torch.zeros((R, stride, 2), dtype=torch.int32, device=dev)

    def rows(**kw):
        r = _lib.GrxRows()
        r.input_ids, r.ids_stride, r.R, r.T = ids.data_ptr(), T, R, T
        r.scores, r.scores_stride, r.V, r.dtype = x.data_ptr(), V, V, _lib.GRX_F32
        for k, v in kw.items():
            setattr(r, k, v)
        return r

    def call(r, con=None, cand_ptr=None, cand_stride=stride, cmax=trie.max_fanout, ws_ptr=None, ws_bytes=None):
        con = con or proc._constraint()
        return lib.grx_constrained_candidates(handle, st, C.byref(r), C.byref(con),
                                              cand.data_ptr() if cand_ptr is None else cand_ptr, cand_stride, cmax,
                                              ws.data_ptr() if ws_ptr is None else ws_ptr,
                                              ws.numel() if ws_bytes is None else ws_bytes, None)

    def msg():
        return lib.grx_last_error().decode()

    calls_before = proc._read_state(0)[3]
    assert call(rows(V=trie.max_token)) == -1 and "token id" in msg()            # trie tokens outside the rows
    assert call(rows(scores_stride=V - 1)) == -1 and "stride" in msg()
    assert call(rows(ids_stride=T - 1)) == -1 and "stride" in msg()
    assert call(rows(dtype=9)) == -1 and "dtype" in msg()
    assert call(rows(scores=None)) == -1
    assert call(rows(), ws_bytes=64) == -4 and "workspace" in msg()
    assert call(rows(), ws_ptr=ws.data_ptr() + 4) == -4
    assert call(rows(), cand_ptr=cand.data_ptr() + 8) == -1 and "128-byte" in msg()
    assert call(rows(), cand_stride=stride + 8) == -1 and "multiple of 16" in msg()
    assert call(rows(), cmax=0) == -1
    con = proc._constraint()
    con.eos_token_id = V
    assert call(rows(), con=con) == -1 and "eos_token_id" in msg()
    con = proc._constraint()
    con.n_prefix = 17
    assert call(rows(), con=con) == -3
    con = proc._constraint()
    con.mode = 5
    assert call(rows(), con=con) == -1
    con = proc._constraint()
    con.repetition_penalty = -2.0
    assert call(rows(), con=con) == -1 and "positive" in msg()
    out = torch.empty_like(x)
    assert lib.grx_constrained_logsoftmax(handle, st, C.byref(rows()), C.byref(proc._constraint()), x.data_ptr(), V, 0,
                                          ws.data_ptr(), ws.numel(), None) == -1 and "alias" in msg()
    assert lib.grx_constrained_logsoftmax(handle, st, C.byref(rows()), C.byref(proc._constraint()), out.data_ptr(), V - 1,
                                          0, ws.data_ptr(), ws.numel(), None) == -1
    assert lib.grx_constrained_logsoftmax(handle, st, C.byref(rows()), C.byref(proc._constraint()), out.data_ptr(), V,
                                          _lib.GRX_BF16, ws.data_ptr(), ws.numel(), None) == -3
    torch.cuda.synchronize()
    assert not bool(cand.any()), "a refused call must not have written anything"
    assert proc._read_state(0)[3] == calls_before, "a refused call must not count as a processor call"


@pytest.mark.parametrize("case", ["count_rl", "count_eval", "fixed_hint"])
def test_rows_that_are_already_normalised_need_not_be_streamed(case):
    """``assume_normalized=True``: generate() hands a processor log_softmax(logits) when nothing runs
    ahead of it, so the reference's own log_softmax is a near-identity (< 1e-6).  The kernel then skips
    the read pass: allowed positions carry the input value bit for bit, everything else is -inf, and the
    result is within the usual 1e-5 of the reference output recorded for the raw scores."""
    data, meta, trie, fn = _small_setup()
    dev = _dev()
    proc = _make(case, fn, meta, assume_normalized=True)
    info = meta["cases"][case]
    for i in range(info["steps"]):
        ids = torch.from_numpy(data[f"{case}.ids.{i}"]).to(dev)
        raw = torch.from_numpy(data[f"{case}.scores.{i}"]).to(dev)
        logp = torch.log_softmax(raw, dim=-1)
        out = proc(ids, logp)
        want = data[f"{case}.out.{i}"]
        _assert_dense_close(out, want, FP32_TOL)
        fin = torch.isfinite(out)
        assert torch.equal(out[fin], logp[fin])  # passed through untouched
        if info["counts_after"][i] >= 0:
            assert proc.count == info["counts_after"][i]
            proc.count = proc.count - 1
        cand = proc.candidates(ids, logp)
        valid = torch.arange(cand.cmax, device=dev)[None, :] < cand.count[:, None]
        assert torch.equal(cand.count.long(), fin.sum(dim=1))
        assert torch.equal(cand.logp[valid], logp[fin])
        assert bool((cand.logsumexp == 0).all())
    proc.check_errors()
    with pytest.raises(ValueError, match="assume_normalized"):
        _make(case, fn, meta, assume_normalized=True, repetition_penalty=1.1)


@pytest.mark.parametrize("V", [1001, 1023, 4099])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_normalised_rows_on_the_element_wise_dense_path(V, dtype):
    """``assume_normalized=True`` where rows or output are not 16-byte aligned: the kernel fills the
    dense row with -inf itself, nothing is streamed, and the fill of every wave must be ordered
    before any wave scatters an allowed value (the allowed ids sit in the last columns, which are
    filled last).  Finite set == oracle on the same input; finite entries == the input bit for bit;
    through ``__call__`` and through ``call_into`` with a padded output view."""
    from oracle.processors import DictTrie, RefConstrainedLogitsProcessor

    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    data, meta, _, _ = _small_setup()
    dev = _dev()
    seqs = data["trie_sequences"].tolist()
    fn = create_prefix_allowed_tokens_fn(FlatTrie(seqs), meta["prefix"])
    o_trie = DictTrie(seqs)
    K = meta["K"]
    ids = torch.from_numpy(data["count_rl.ids.1"]).repeat(32, 1)  # many workgroups in flight at once
    R = ids.shape[0]
    torch.manual_seed(V + 7)
    raw = torch.randn(R, V) * 5
    logp = torch.log_softmax(raw, dim=-1).to(dtype)
    ref = RefConstrainedLogitsProcessor(o_trie.get, K, 3, meta["prefix"], meta["eos"])
    ref.count = 1
    want = ref(ids, logp.float())  # the reference normalises once more: a near-identity
    fin_w = torch.isfinite(want)
    tol = FP32_TOL if dtype == torch.float32 else 0.07  # 16-bit rows are not normalised to fp32 accuracy
    proc = ConstrainedLogitsProcessor(fn, K, 3, meta["prefix"], meta["eos"], assume_normalized=True)
    ids_d, logp_d = ids.to(dev), logp.to(dev)
    for rep in range(3):
        proc.count = 1
        got = proc(ids_d, logp_d)
        assert got.dtype == dtype
        for pad in (5, 3):
            backing = torch.full((R, V + pad), 7.0, dtype=dtype, device=dev)
            proc.count = 1
            into = proc.call_into(ids_d, logp_d, backing[:, :V])
            assert into.data_ptr() == backing.data_ptr()
            assert bool((backing[:, V:] == 7.0).all()), "padding columns must stay untouched"
            assert torch.equal(into, got)
        g = got.cpu()
        assert torch.equal(torch.isfinite(g), fin_w), "allowed-token sets differ"
        assert bool((g[~fin_w] == -float("inf")).all())
        assert torch.equal(g[fin_w], logp[fin_w])  # passed through untouched
        assert float((g.float()[fin_w] - want[fin_w]).abs().max()) <= tol
    if dtype != torch.float32:
        out32 = torch.empty((R, V), dtype=torch.float32, device=dev)
        proc.count = 1
        proc.call_into(ids_d, logp_d, out32)
        o = out32.cpu()
        assert torch.equal(torch.isfinite(o), fin_w)
        assert torch.equal(o[fin_w], logp.float()[fin_w])
    proc.check_errors()
