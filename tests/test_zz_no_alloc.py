"""GPU: the entry point that allocates nothing (``grx_constrained_candidates_noalloc``) against
``grx_constrained_candidates``, both called through the C ABI on the same inputs: candidate rows
must be identical bit for bit.  Two calls in a row share the library's built-in scratch array, the
second on a smaller batch (few rows are streamed by several workgroups each, whose arrival counters
live in that array)."""

import ctypes as C

import pytest
import torch

from helpers import load_npz

pytestmark = pytest.mark.gpu


def _setup():
    from genrec_amd.trie import FlatTrie

    data, meta = load_npz("processors.npz")
    return data, meta, FlatTrie(data["trie_sequences"].tolist())


def _constraint(mode, meta, with_prefix=True):
    from genrec_amd import _lib

    con = _lib.GrxConstraint()
    con.mode = mode
    con.prefix_index = 3
    con.n_prefix = len(meta["prefix"]) if with_prefix else -1
    for i, t in enumerate(meta["prefix"] if with_prefix else []):
        con.prefix_ids[i] = int(t)
    con.eos_token_id = int(meta["eos"])
    con.repetition_penalty = 1.0
    return con


def _rows(ids, scores):
    from genrec_amd import _lib

    rows = _lib.GrxRows()
    rows.input_ids = ids.data_ptr()
    rows.ids_stride = ids.stride(0)
    rows.R, rows.T = ids.shape
    rows.scores = scores.data_ptr()
    rows.scores_stride = scores.stride(0)
    rows.V = scores.shape[1]
    rows.dtype = _lib.torch_dtype_code(scores.dtype)
    return rows


def _state(count):
    from genrec_amd import _lib

    st = C.c_void_p()
    _lib.check(_lib.lib().grx_state_create(C.byref(st)))
    _lib.check(_lib.lib().grx_state_set_count(st, count, None))
    return st


def _both(trie, mode, meta, ids, scores, count, cmax):
    """Candidate buffers of the workspace entry point and of the one that allocates nothing."""
    from genrec_amd import _lib

    lib = _lib.lib()
    dev = scores.device
    stride = int(lib.grx_candidate_stride(cmax))
    R = ids.shape[0]
    stream = _lib.current_stream_ptr(dev)
    handle = trie.device_handle(dev)
    con = _constraint(mode, meta)
    rows = _rows(ids, scores)
    out = []
    for noalloc in (False, True):
        cand = torch.full((R, stride, 2), -7, dtype=torch.int32, device=dev)
        st = _state(count)
        if noalloc:
            before = torch.cuda.memory_allocated(dev)
            _lib.check(lib.grx_constrained_candidates_noalloc(
                handle, st, C.byref(rows), C.byref(con), cand.data_ptr(), stride, cmax, stream))
            assert torch.cuda.memory_allocated(dev) == before
        else:
            ws = torch.zeros(int(lib.grx_workspace_bytes(R)), dtype=torch.uint8, device=dev)
            _lib.check(lib.grx_constrained_candidates(
                handle, st, C.byref(rows), C.byref(con), cand.data_ptr(), stride, cmax, ws.data_ptr(), ws.numel(),
                stream))
        torch.cuda.synchronize(dev)
        state = (C.c_int32 * 4)()
        _lib.check(lib.grx_state_read(st, C.byref(state), stream))
        out.append((cand, list(state)))
        _lib.check(lib.grx_state_destroy(st))
    return out


@pytest.mark.parametrize("mode_name", ["count", "fixed_hint"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_no_alloc_entry_point_equals_the_workspace_entry_point(mode_name, dtype):
    from genrec_amd import _lib

    data, meta, trie = _setup()
    dev = torch.device("cuda", 0)
    mode = _lib.GRX_MODE_COUNT if mode_name == "count" else _lib.GRX_MODE_FIXED_HINT
    case = "count_rl" if mode_name == "count" else "fixed_hint"
    V = 8192  # 16-byte aligned rows long enough for few rows to be split across workgroups
    cmax = max(trie.max_fanout, 1)
    torch.manual_seed(11)
    ids1 = torch.from_numpy(data[f"{case}.ids.1"])
    base = ids1.shape[0]
    # first call: many rows (one workgroup per row); second call, right behind it: fewer rows (split rows)
    for reps, step in ((40, 1), (1, 2), (3, 1)):
        ids = torch.from_numpy(data[f"{case}.ids.{step}"]).repeat(reps, 1).to(dev)
        scores = (torch.randn(ids.shape[0], V) * 4).to(dtype).to(dev)
        (want, st_w), (got, st_g) = _both(trie, mode, meta, ids, scores, step, cmax)
        assert st_w == st_g and st_w[1] == _lib.GRX_ROWERR_NONE
        assert torch.equal(got, want), f"R={base * reps}: candidate rows differ"
        assert int(want[:, 0, 0].min()) >= 1


def test_no_alloc_entry_point_rejects_what_it_cannot_hold():
    from genrec_amd import _lib

    data, meta, trie = _setup()
    dev = torch.device("cuda", 0)
    lib = _lib.lib()
    limit = int(lib.grx_noalloc_max_rows())
    assert limit == 65536
    con = _constraint(_lib.GRX_MODE_FIXED_HINT, meta)
    ids = torch.zeros((4, 8), dtype=torch.int64, device=dev)
    scores = torch.zeros((4, 2048), device=dev)
    rows = _rows(ids, scores)
    cand = torch.zeros((4, 16 * 8, 2), dtype=torch.int32, device=dev)
    handle = trie.device_handle(dev)
    rows.R = limit + 1  # refused on the count alone, before anything is touched
    rc = lib.grx_constrained_candidates_noalloc(handle, None, C.byref(rows), C.byref(con), cand.data_ptr(), 128, 100, None)
    assert rc == -3 and b"at most 65536" in lib.grx_last_error()
    rows.R = 4
    rows.planned = 1
    rc = lib.grx_constrained_candidates_noalloc(handle, None, C.byref(rows), C.byref(con), cand.data_ptr(), 128, 100, None)
    assert rc == -3 and b"planned" in lib.grx_last_error()
    torch.cuda.synchronize(dev)
    assert not bool(cand.any())
