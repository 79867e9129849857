"""Randomised parity of both processors against the CPU oracle on small random tries: ragged item
lengths, prefix lengths 0..4, EOS present or None, rows that leave the trie, rows shorter than the
window, decoy prefix occurrences, arbitrary counter values.  Results, counter movement and exception
types must agree case by case."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda", 0)


def _random_case(rng):
    V = int(rng.integers(64, 700))
    n_prefix = int(rng.integers(0, 5))
    prefix = rng.integers(3, 20, size=n_prefix).tolist()
    eos = int(rng.integers(0, 3)) if rng.random() < 0.8 else None
    levels = int(rng.integers(1, 5))
    items = set()
    for _ in range(int(rng.integers(1, 60))):
        depth = int(rng.integers(1, levels + 1))
        items.add(tuple(int(rng.integers(20, V)) for _ in range(depth)))
    tail = [eos] if eos is not None else [int(rng.integers(20, V))]
    seqs = [prefix + list(it) + tail for it in sorted(items)]
    K = int(rng.integers(1, 5))
    groups = int(rng.integers(1, 4))
    R = K * groups
    step = int(rng.integers(0, levels + 1))
    T = int(rng.integers(max(1, n_prefix + step), n_prefix + step + 12))
    rows = rng.integers(20, V, size=(R, T))
    for r in range(R):
        kind = rng.random()
        it = list(sorted(items)[int(rng.integers(0, len(items)))])
        gen = it[:step]
        body = prefix + gen
        if kind < 0.75 and len(body) <= T:
            rows[r, T - len(body):] = body           # a valid path (possibly shorter than `step`)
            if rng.random() < 0.3 and n_prefix and T - len(body) - n_prefix - 1 >= 0:
                at = int(rng.integers(0, T - len(body) - n_prefix))
                rows[r, at : at + n_prefix] = prefix  # a decoy occurrence earlier in the row
        elif kind < 0.85 and eos is not None:
            rows[r, -1] = eos                         # finished row
        # else: leave random tokens (leaves the trie / no prefix at all)
    count = int(rng.integers(0, levels + 2)) if rng.random() < 0.3 else step
    return dict(V=V, prefix=prefix, eos=eos, seqs=seqs, K=K, rows=rows.astype(np.int64), count=count,
                use_prefix_ids=bool(rng.random() < 0.6), dtype=[torch.float32, torch.bfloat16][int(rng.integers(0, 2))])


@pytest.mark.parametrize("block", range(6))
def test_random_tries_rows_and_counters(block):
    from oracle.processors import (DictTrie, RefConstrainedLogitsProcessor,
                                   RefFixedHintConstrainedLogitsProcessor)

    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    dev = _dev()
    rng = np.random.default_rng(1000 + block)
    outcomes = {"ok": 0, "AssertionError": 0, "ValueError": 0}
    for _ in range(25):
        c = _random_case(rng)
        trie = FlatTrie(c["seqs"])
        fn = create_prefix_allowed_tokens_fn(trie, c["prefix"])
        get = DictTrie(c["seqs"]).get
        ids = torch.from_numpy(c["rows"])
        scores = (torch.randn(ids.shape[0], c["V"], generator=torch.Generator().manual_seed(int(rng.integers(1 << 30))))
                  * 4).to(c["dtype"])
        pidx, K, eos = len(c["prefix"]), c["K"], c["eos"]
        pids = c["prefix"] if c["use_prefix_ids"] else None
        for kind in ("count", "fixed"):
            if kind == "count":
                ours = ConstrainedLogitsProcessor(fn, K, pidx, pids, eos, sync_errors=True)
                ref = RefConstrainedLogitsProcessor(get, K, pidx, pids, eos)
                ours.count = ref.count = c["count"]
            else:
                ours = FixedHintConstrainedLogitsProcessor(fn, K, pidx, c["prefix"], eos, sync_errors=True)
                ref = RefFixedHintConstrainedLogitsProcessor(get, K, pidx, c["prefix"], eos)
            try:
                want = ref(ids, scores.float())
            except (AssertionError, ValueError) as e:
                with pytest.raises(type(e)):
                    ours(ids.to(dev), scores.to(dev))
                outcomes[type(e).__name__] += 1
                continue
            outcomes["ok"] += 1
            got = ours(ids.to(dev), scores.to(dev)).float().cpu()
            fin = torch.isfinite(want)
            assert torch.equal(torch.isfinite(got), fin), (kind, c["prefix"], c["count"])
            tol = TOL if c["dtype"] == torch.float32 else 0.07
            assert float((got[fin] - want[fin]).abs().max()) <= tol
            if kind == "count":
                assert ours.count == ref.count
                ours.count = c["count"]
            cand = ours.candidates(ids.to(dev), scores.to(dev))
            assert torch.equal(cand.count.cpu().long(), fin.sum(dim=1))
            valid = torch.arange(cand.cmax)[None, :] < cand.count.cpu()[:, None]
            cols = torch.nonzero(fin)[:, 1]
            assert torch.equal(cand.tokens.cpu()[valid].long(), cols)
            assert float((cand.logp.cpu()[valid] - want[fin]).abs().max()) <= TOL
    assert outcomes["ok"] >= 10 and outcomes["AssertionError"] + outcomes["ValueError"] >= 5, outcomes
