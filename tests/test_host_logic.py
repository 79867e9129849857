"""CPU: host-side logic of the package (CSR trie, index builder, synthetic workloads,
the C-ABI library's symbols) -- no kernel is launched here."""

import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from helpers import INDEX_NAMES, load_json_gz, load_npz, write_index_file
from oracle import processors as oproc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from genrec_amd import _lib

    header = open(os.path.join(ROOT, "include", "genrec_hip.h")).read()
    declared = set(re.findall(r"GRX_API\s+[\w\s\*]+?\b(grx_\w+)\s*\(", header))
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.grx_abi_version() == 15
    assert lib.grx_candidate_stride(162) == 176 and lib.grx_candidate_stride(1) == 16
    assert lib.grx_workspace_bytes(8192) >= 0
    # argument validation happens before any device work
    assert lib.grx_trie_create(None, None, 1, 0, C.byref(C.c_void_p())) != 0
    assert b"null" in lib.grx_last_error()


def test_structs_match_the_header_layout():
    from genrec_amd import _lib

    assert C.sizeof(_lib.GrxRows) == 56
    assert C.sizeof(_lib.GrxConstraint) == 16 + 8 * _lib.GRX_MAX_PREFIX + 16 + 8 + 8
    assert _lib.GrxBeamStepArgs.cand.offset == 48 and _lib.GrxBeamStepArgs.cand_stride.offset == 56
    assert C.sizeof(_lib.GrxBeamStepArgs) == 64 + 8 * 21 + 8
    assert C.sizeof(_lib.GrxBeamInitArgs) == 16 + 8 + 8 + 8 + 7 * 16 + 8 + 8 + 8
    assert C.sizeof(_lib.GrxPlanRequest) == 40


def test_flat_trie_equals_reference_tries(tmp_path):
    from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer
    from genrec_amd.trie import FlatTrie, as_flat_trie, build_trie_from_index, create_prefix_allowed_tokens_fn
    import hashlib

    gold = load_json_gz("tries.json.gz")
    for name in INDEX_NAMES:
        g = gold[name]
        path = write_index_file(name, tmp_path)
        tok = SidTokenizer.from_index(path)
        trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(trie.child_off, dtype="<i4").tobytes())
        h.update(np.ascontiguousarray(trie.child_tok, dtype="<i4").tobytes())
        assert h.hexdigest() == g["csr_sha256"]
        assert (len(trie), trie.n_nodes, trie.n_edges, trie.max_fanout) == (
            g["n_unique_items"], g["n_nodes"], g["n_edges"], g["max_fanout"])
        assert [list(x) for x in trie.level_fanout()] == [list(x) for x in g["level_fanout"]]
        for q, a in zip(g["queries"], g["answers"]):
            assert trie.get(q) == a
        fn = create_prefix_allowed_tokens_fn(trie, suffix)
        assert as_flat_trie(fn) is trie
    # sid_levels truncation and prefix from the chat template (evaluate.py:38-60)
    path = write_index_file("Instruments", tmp_path)
    tok = SidTokenizer.from_index(path)
    t3, suffix, pidx = build_trie_from_index(path, tok, prefix=None, sid_levels=3)
    o3, o_suffix, o_pidx = oproc.ref_build_trie_from_index(path, tok, prefix=None, sid_levels=3)
    assert (suffix, pidx, len(t3)) == (o_suffix, o_pidx, len(o3)) and sorted(t3) == sorted(o3)


def test_trie_sources_and_edge_cases():
    from genrec_amd.trie import FlatTrie, as_flat_trie

    seqs = [[5, 6, 7], [5, 6, 8], [5, 9], [5, 6, 7]]
    ref = oproc.DictTrie(seqs)
    flat = FlatTrie(seqs)
    assert len(flat) == 4 and sorted(flat) == sorted(ref)
    for q in ([], [5], [5, 6], [5, 6, 7], [5, 6, 7, 1], [4], [5, 9], [2 ** 40]):
        assert flat.get(q) == sorted(ref.get(q))
    # from a reference-style object, from a closure over one, and from an opaque function
    assert as_flat_trie(ref).child_tok.tolist() == flat.child_tok.tolist()
    assert as_flat_trie(lambda ids: ref.get(ids)).child_off.tolist() == flat.child_off.tolist()

    class Opaque:
        def __call__(self, ids):
            return ref.get(ids)

    assert as_flat_trie(Opaque()).child_tok.tolist() == flat.child_tok.tolist()
    empty = FlatTrie([])
    assert empty.n_nodes == 1 and empty.get([]) == [] and empty.max_fanout == 0 and empty.max_token == -1
    with pytest.raises(TypeError):
        as_flat_trie(42)
    with pytest.raises(ValueError):
        FlatTrie.from_allowed_tokens_fn(lambda ids: [1], max_depth=8)  # never terminates


def test_synthetic_workload_and_tokenizer(tmp_path):
    from genrec_amd.synth import (ASSISTANT_PREFIX, ASSISTANT_PREFIX_IDS, SidTokenizer, padded_vocab,
                                  sample_decode_rows, synthetic_index, write_index)
    from genrec_amd.trie import build_trie_from_index

    assert padded_vocab(151665 + 930) == 152640 and padded_vocab(151665 + 100) == 151936
    for cb, want_v in ((32, 151936), (64, 151936), (128, 152192), (512, 153728)):
        path = write_index(str(tmp_path / f"cb{cb}.json"), synthetic_index(9922, 4, cb, seed=0))
        tok = SidTokenizer.from_index(path)
        assert tok.vocab_size_padded == want_v
        trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
        assert suffix == ASSISTANT_PREFIX_IDS and trie.max_fanout <= cb
    rows = sample_decode_rows(trie, 3, 4, 2, prompt_len=7, seed=1, hint_depths=[0, 1, 2], distractors=True)
    assert rows.shape == (12, 7 + 2 + 3 + 2)
    for r in rows.tolist():
        allowed = oproc.fixed_hint_allowed_row(r, suffix, trie.get, tok.eos_token_id)
        assert len(allowed) > 0
    table = tok.code_table(tok.vocab_size_padded)
    assert table[tok.encode("<a_17>")[0]] == 17 and table[tok.eos_token_id] == -1
    assert tok.decode(tok.encode("<a_1><b_2>")) == "<a_1><b_2>"


def test_processors_refuse_cpu_tensors_and_missing_library(monkeypatch):
    from genrec_amd import _lib
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.trie import FlatTrie, create_prefix_allowed_tokens_fn

    fn = create_prefix_allowed_tokens_fn(FlatTrie([[1, 2, 3, 4]]), [1, 2, 3])
    proc = ConstrainedLogitsProcessor(fn, 1, 3, [1, 2, 3], 0)
    assert proc.count == 0 and proc.prefix_index == 3 and proc._num_beams == 1
    with pytest.raises(_lib.GenrecHipError, match="no CPU path"):
        proc(torch.tensor([[1, 2, 3]]), torch.zeros(1, 8))
    with pytest.raises(AssertionError, match="2D"):
        proc(torch.tensor([1, 2, 3]), torch.zeros(1, 8))
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgenrec_hip.so")
    with pytest.raises(_lib.GenrecHipError, match="not built"):
        _lib.lib()


def test_builders_keep_the_reference_surface(tmp_path):
    from transformers import LogitsProcessorList
    from transformers.generation import LogitsProcessor

    from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
    from genrec_amd.logit_processor import ConstrainedLogitsProcessor
    from genrec_amd.rewards import build_reward_setup
    from genrec_amd.synth import SidTokenizer, synthetic_index, write_index
    from genrec_amd.util import (_extract_number, build_constrained_logits_processor,
                                 build_fixed_hint_constrained_logits_processor)

    path = write_index(str(tmp_path / "i.json"), synthetic_index(50, 3, 8, seed=1))
    tok = SidTokenizer.from_index(path)
    for build, cls in ((build_constrained_logits_processor, ConstrainedLogitsProcessor),
                       (build_fixed_hint_constrained_logits_processor, FixedHintConstrainedLogitsProcessor)):
        lst = build(path, tok, num_beams=16)
        assert isinstance(lst, LogitsProcessorList) and len(lst) == 1
        p = lst[0]
        assert isinstance(p, cls) and isinstance(p, LogitsProcessor)
        assert (p._num_beams, p.prefix_index, p.prefix_ids, p.eos_token_id) == (16, 3, [151644, 77091, 198], 151645)
        assert callable(p._prefix_allowed_tokens_fn) and p._prefix_allowed_tokens_fn([151644, 77091, 198])
    assert hasattr(lst[0], "count") is False and not hasattr(lst[0], "reset")  # the fixed-hint processor is stateless
    assert _extract_number("<a_12><b_7>x_3") == [12, 7, 3]
    names = {m: [f.__name__ for f in build_reward_setup(m, 4)[0]] for m in
             ("rule_only", "ranking_only", "prefix_rule_only", "prefix_only", "prefix_ranking", "ranking")}
    assert names == {
        "rule_only": ["rule_reward"],
        "ranking_only": ["ndcg_rule_reward", "rule_reward"],
        "prefix_rule_only": ["prefix_rule_reward", "rule_reward"],
        "prefix_only": ["prefix_rule_reward", "ndcg_rule_reward", "rule_reward"],
        "prefix_ranking": ["prefix_rule_reward", "rule_reward", "ndcg_rule_reward"],
        "ranking": ["rule_reward", "ndcg_rule_reward"],
    }
    assert build_reward_setup("ranking_only", 4)[1] == [1.0, 0.0]
    with pytest.raises(ValueError):
        build_reward_setup("nope", 4)


def test_metrics_from_histogram_matches_oracle():
    from oracle.scoring import ref_compute_metrics

    from genrec_amd.metrics import metrics_from_histogram

    g = load_json_gz("scoring.json.gz")
    data = g["metrics_data"]
    n_beam = len(data[0]["predict"])
    hist = [0] * (n_beam + 1)
    for s in data:
        rank = next((i for i, p in enumerate(s["predict"]) if p == s["output"]), n_beam)
        hist[rank] += 1
    got = metrics_from_histogram(hist, len(data), n_beam)
    want = ref_compute_metrics(data)
    assert set(got) == set(want) and all(abs(got[k] - want[k]) < 1e-12 for k in want)


def test_cascade_hinted_prompts_are_left_padded_prompt_plus_hint():
    import torch

    from genrec_amd.cascade import hinted_prompts

    prompts = torch.tensor([[5, 6, 7], [8, 9, 10]])
    gt = torch.tensor([[101, 102, 103, 104], [201, 202, 203, 204]])
    rows = hinted_prompts(prompts, gt, torch.tensor([0, 2], dtype=torch.int32), 3 + 3, 0)
    assert rows.tolist() == [[0, 0, 0, 5, 6, 7], [0, 8, 9, 10, 201, 202]]
    rows = hinted_prompts(prompts, gt, torch.tensor([3, 1], dtype=torch.int32), 3 + 3, 0)
    assert rows.tolist() == [[5, 6, 7, 101, 102, 103], [0, 0, 8, 9, 10, 201]]
    assert hinted_prompts(prompts, gt, torch.tensor([0, 0], dtype=torch.int32), 3, 0).tolist() == prompts.tolist()


def test_trie_cache_round_trip_and_staleness(tmp_path):
    import json as _json

    from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, synthetic_index, write_index
    from genrec_amd.trie import FlatTrie, build_trie_from_index

    path = write_index(str(tmp_path / "a.index.json"), synthetic_index(300, 4, 16, seed=1))
    tok = SidTokenizer.from_index(path)
    built, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX, cache=True)
    files = [f for f in os.listdir(tmp_path) if f.endswith(".grxtrie")]
    assert len(files) == 1
    cached, suffix2, pidx2 = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX, cache=True)
    assert (suffix2, pidx2, len(cached)) == (suffix, pidx, len(built))
    assert np.array_equal(cached.child_off, built.child_off) and np.array_equal(cached.child_tok, built.child_tok)
    assert sorted(cached) == sorted(built)
    # a different level cut, a changed index, or a changed tokenizer never reuses the file
    t3, _, _ = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX, sid_levels=3, cache=True)
    assert t3.n_nodes != built.n_nodes and len([f for f in os.listdir(tmp_path) if f.endswith(".grxtrie")]) == 2
    index = _json.load(open(path))
    index["0"] = ["<a_15>", "<b_15>", "<c_15>", "<d_15>"]
    write_index(path, index)
    rebuilt, _, _ = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX, cache=True)
    fresh, _, _ = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
    assert np.array_equal(rebuilt.child_tok, fresh.child_tok) and np.array_equal(rebuilt.child_off, fresh.child_off)
    # a damaged cache file is rebuilt, and load() refuses foreign files
    victim = str(tmp_path / [f for f in os.listdir(tmp_path) if f.endswith(".grxtrie")][0])
    with open(victim, "r+b") as f:
        f.truncate(80)
    with pytest.raises(ValueError):
        FlatTrie.load(victim)
    with pytest.raises(KeyError):
        good = [str(tmp_path / f) for f in os.listdir(tmp_path) if f.endswith(".grxtrie") and str(tmp_path / f) != victim][0]
        FlatTrie.load(good, key=b"x" * 32)


def test_decode_beams_reproduces_batch_decode():
    from genrec_amd.results import decode_beams, token_text_table
    from genrec_amd.synth import SidTokenizer

    tok = SidTokenizer(f"<{l}_{c}>" for l in "abcd" for c in range(5))
    table = token_text_table(tok, len(tok) + 7)
    ids = np.array([[[tok.encode("<a_1>")[0], tok.encode("<b_2>")[0], tok.eos_token_id, tok.pad_token_id],
                     [tok.encode("<a_0>")[0], tok.encode("<b_4>")[0], tok.encode("<c_3>")[0], tok.eos_token_id]]])
    assert decode_beams(ids, table) == [["<a_1><b_2>", "<a_0><b_4><c_3>"]]
    assert decode_beams(ids, table)[0] == [tok.decode(r) for r in ids[0]]


def test_saved_metric_files_have_the_layout_pandas_gives_the_reference(tmp_path):
    """metrics.json / metrics.tsv as evaluate.py:182-202 writes them (pandas DataFrame.to_csv there)."""
    import json as _json

    import pandas as pd

    from genrec_amd.metrics import save_metrics

    metrics = {"HR@1": 0.123456, "HR@5": 0.5, "HR@10": 1.0, "NDCG@1": 0.123456, "NDCG@5": 0.33335, "NDCG@10": 2 / 3}
    valid = [1, 5, 10]
    for sub in ("runs/exp1/eval", "flat"):
        path = tmp_path / sub / "metrics.json"
        cwd = os.getcwd()
        os.chdir(tmp_path)
        try:
            save_metrics(metrics, os.path.join(sub, "metrics.json"), valid)
        finally:
            os.chdir(cwd)
        rounded = {k: round(float(v), 4) for k, v in metrics.items()}
        assert _json.load(open(path)) == rounded
        assert open(path).read() == _json.dumps(rounded, indent=2, ensure_ascii=False)
        cols = [f"HR@{k}" for k in valid] + [f"NDCG@{k}" for k in valid]
        row_name = sub.split("/", 1)[1] if "/" in sub else sub
        df = pd.DataFrame([{c: rounded.get(c) for c in cols}], index=[row_name])
        df.index.name = "path"
        want = df.to_csv(sep="\t", encoding="utf-8")
        assert open(tmp_path / sub / "metrics.tsv", newline="").read() == want


def test_entry_points_reject_bad_arguments_on_the_host():
    """Every shape / pointer check runs before the first device call, so it can be exercised without a
    GPU: a caller that passes garbage gets GRX_E_INVALID (or _UNSUPPORTED / _WORKSPACE) and a message,
    never a kernel launched on bad operands."""
    import numpy as np

    from genrec_amd import _lib

    lib = _lib.lib()
    INVALID, UNSUPPORTED = -1, -3

    def msg():
        return lib.grx_last_error().decode()

    def trie(off, tok):
        off = np.asarray(off, dtype=np.int32)
        tok = np.asarray(tok, dtype=np.int32)
        out = C.c_void_p()
        rc = lib.grx_trie_create(off.ctypes.data, tok.ctypes.data if len(tok) else None, len(off) - 1, len(tok),
                                 C.byref(out))
        assert out.value is None or rc == 0
        return rc

    # malformed child tables (a well-formed one would go on to allocate device memory)
    assert trie([0, 2, 2, 2], [7, 5]) == INVALID and "strictly increasing" in msg()
    assert trie([0, 2, 2, 2], [-1, 5]) == INVALID and "negative token" in msg()
    assert trie([0, 1, 3, 2], [4, 5]) == INVALID                       # offsets not monotone / not ending at n_edges
    assert trie([1, 2, 2], [4]) == INVALID and "start at 0" in msg()
    assert trie([0, 0, 1], [4]) == INVALID and "backwards" in msg()      # node 1 pointing at itself
    assert lib.grx_trie_create(None, None, 3, 1, C.byref(C.c_void_p())) == INVALID  # n_nodes != n_edges + 1

    # beam step: shapes and aliasing
    a = _lib.GrxBeamStepArgs()
    a.B, a.K, a.V, a.n_keep, a.cmax = 1, 4, 100, 8, 4
    a.cur_len, a.prompt_len, a.max_length, a.copy_from = 3, 3, 8, 3
    a.eos_token_id, a.early_stopping, a.length_penalty, a.cand_stride = 1, 0, 1.0, 16
    assert lib.grx_beam_step(C.byref(a), None) == INVALID and "null buffer" in msg()
    a.K = 300
    assert lib.grx_beam_step(C.byref(a), None) == UNSUPPORTED and "at most" in msg()
    a.K, a.n_keep = 4, 2
    assert lib.grx_beam_step(C.byref(a), None) == INVALID and "n_keep" in msg()
    a.n_keep, a.cur_len = 8, 8
    assert lib.grx_beam_step(C.byref(a), None) == INVALID and "max_length" in msg()
    a.cur_len, a.cand_stride = 3, 4
    assert lib.grx_beam_step(C.byref(a), None) == INVALID and "cand_stride" in msg()
    a.cand_stride, a.cand_shared = 16, 7
    assert lib.grx_beam_step(C.byref(a), None) == INVALID and "cand_shared" in msg()
    a.cand_shared, a.V = 0, 1 << 25
    assert lib.grx_beam_step(C.byref(a), None) == UNSUPPORTED and "2^24" in msg()

    # scoring kernels
    assert lib.grx_group_advantage(None, None, 10, 1, 4, 1, 1e-4, None, None, None, None, None) == INVALID
    assert "do not split" in msg()
    assert lib.grx_group_advantage(None, None, 8, 1, 4, 5, 1e-4, None, None, None, None, None) == INVALID
    assert lib.grx_group_advantage(None, None, 8, 1, 4, 1, 1e-4, None, None, None, None, None) == INVALID
    assert "null input" in msg()
    assert lib.grx_sid_rewards(None, None, 4, None, None, 4, 8, 300, 1, None, None, None, None, 1, None) == UNSUPPORTED
    assert lib.grx_hit_ranks(None, None, -1, 4, 4, None, None, None) == INVALID
    assert lib.grx_group_normalize_tokenwise(None, None, 9, 4, 4, 1e-4, None, None) == INVALID
    # empty work is fine and touches nothing
    assert lib.grx_group_advantage(None, None, 0, 1, 4, 1, 1e-4, None, None, None, None, None) == 0
    assert lib.grx_hit_ranks(None, None, 0, 4, 4, None, None, None) == 0
    a.B, a.V = 0, 100
    assert lib.grx_beam_step(C.byref(a), None) == 0


def test_bench_launcher_reports_failed_ranks():
    """`bench.py --gpus 2` without a launcher starts its ranks as a child process; when they fail (here:
    no GPU) the launcher prints no result line and exits non-zero instead of labelling one rank's run."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "32", "--steps", "5"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=120)
    assert run.returncode != 0
    assert not [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert "2-rank run exited" in run.stderr
