"""ctypes binding of libgenrec_hip.so (declared in include/genrec_hip.h).

There is no fallback: if the shared library is missing or a call fails, an
exception is raised.  Build it with ``make -C genrec_amd/csrc`` (or
``python -c "import __graft_entry__ as g; g.build()"`` from the repo root).
"""

from __future__ import annotations

import ctypes as C
import os
import threading

LIB_NAME = "libgenrec_hip.so"
# GENREC_AMD_LIB: load another build of the library (kernel tuning: variant builds timed side by side)
LIB_PATH = os.environ.get("GENREC_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

GRX_OK = 0
GRX_F32, GRX_BF16, GRX_F16 = 0, 1, 2
GRX_MODE_COUNT, GRX_MODE_FIXED_HINT = 0, 1
GRX_ROWERR_NONE, GRX_ROWERR_EMPTY, GRX_ROWERR_NOMATCH, GRX_ROWERR_OVERFLOW = 0, 1, 2, 3
GRX_MAX_PREFIX = 16

# Every symbol include/genrec_hip.h declares.
EXPORTED_SYMBOLS = (
    "grx_abi_version",
    "grx_last_error",
    "grx_trie_create",
    "grx_trie_destroy",
    "grx_trie_max_fanout",
    "grx_trie_max_token",
    "grx_state_create",
    "grx_state_destroy",
    "grx_state_set_count",
    "grx_state_clear_error",
    "grx_state_read",
    "grx_workspace_bytes",
    "grx_constrained_logsoftmax",
    "grx_reuse_bytes",
    "grx_constrained_logsoftmax_reuse",
    "grx_constrained_candidates",
    "grx_constrained_beam_step",
    "grx_lm_head_scratch_bytes",
    "grx_lm_head_candidates",
    "grx_candidate_stride",
    "grx_noalloc_max_rows",
    "grx_constrained_candidates_noalloc",
    "grx_chunk_parts",
    "grx_constrained_partial",
    "grx_constrained_finish",
    "grx_constrained_finish_penalised",
    "grx_timing_enable",
    "grx_timing_read",
    "grx_read_probe",
    "grx_beam_step",
    "grx_beam_init",
    "grx_sid_rewards",
    "grx_sid_rewards_tokens",
    "grx_rollout_score_tokens",
    "grx_hit_ranks",
    "grx_group_advantage",
    "grx_group_normalize_tokenwise",
    "grx_token_prefix_advantages",
    "grx_comm_unique_id",
    "grx_comm_create",
    "grx_comm_destroy",
    "grx_comm_allgather",
)


class GrxRows(C.Structure):
    _fields_ = [
        ("input_ids", C.c_void_p),
        ("ids_stride", C.c_int64),
        ("R", C.c_int32),
        ("T", C.c_int32),
        ("scores", C.c_void_p),
        ("scores_stride", C.c_int64),
        ("V", C.c_int32),
        ("dtype", C.c_int32),
        ("planned", C.c_int32),
        ("reserved", C.c_int32),
    ]


class GrxConstraint(C.Structure):
    _fields_ = [
        ("mode", C.c_int32),
        ("prefix_index", C.c_int32),
        ("n_prefix", C.c_int32),
        ("eos_token_id", C.c_int32),
        ("prefix_ids", C.c_int64 * GRX_MAX_PREFIX),
        ("repetition_penalty", C.c_float),
        ("penalty_skip", C.c_int32),
        ("assume_normalized", C.c_int32),
        ("keep_top", C.c_int32),
        ("row_bias", C.c_void_p),
        ("shared_prefix_len", C.c_int32),
        ("shared_group", C.c_int32),
    ]


class GrxLmHead(C.Structure):
    _fields_ = [
        ("hidden", C.c_void_p),
        ("hidden_stride", C.c_int64),
        ("weight", C.c_void_p),
        ("weight_stride", C.c_int64),
        ("H", C.c_int32),
        ("reserved", C.c_int32),
    ]


class GrxChunk(C.Structure):
    _fields_ = [
        ("v_begin", C.c_int32),
        ("v_total", C.c_int32),
        ("slot", C.c_int32),
        ("n_slots", C.c_int32),
        ("parts", C.c_int32),
        ("first", C.c_int32),
        ("pen_logits", C.c_void_p),
        ("pen_stride", C.c_int64),
    ]


class GrxTokenPrefixArgs(C.Structure):
    _fields_ = [
        ("completion_ids", C.c_void_p),
        ("ids_stride", C.c_int64),
        ("completion_mask", C.c_void_p),
        ("gt_ids", C.c_void_p),
        ("gt_stride", C.c_int64),
        ("gt_len", C.c_void_p),
        ("ndcg", C.c_void_p),
        ("rank_weights", C.c_void_p),
        ("R", C.c_int32),
        ("Tc", C.c_int32),
        ("K", C.c_int32),
        ("eos_token_id", C.c_int32),
        ("normalize", C.c_int32),
        ("total_token_normalize", C.c_int32),
        ("has_ndcg", C.c_int32),
        ("error_token_penalty", C.c_int32),
        ("rank_offset", C.c_int32),
        ("prefix_weight", C.c_float),
        ("ndcg_weight", C.c_float),
        ("eps", C.c_float),
        ("token_rewards", C.c_void_p),
        ("seq_rewards", C.c_void_p),
        ("error_mask", C.c_void_p),
        ("scratch", C.c_void_p),
        ("token_advantages", C.c_void_p),
    ]


class GrxPlanRequest(C.Structure):
    _fields_ = [
        ("trie", C.c_void_p),
        ("state", C.c_void_p),
        ("con", C.c_void_p),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t),
    ]


class GrxBeamStepArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int32),
        ("K", C.c_int32),
        ("V", C.c_int32),
        ("n_keep", C.c_int32),
        ("cmax", C.c_int32),
        ("cur_len", C.c_int32),
        ("prompt_len", C.c_int32),
        ("max_length", C.c_int32),
        ("copy_from", C.c_int32),
        ("eos_token_id", C.c_int32),
        ("early_stopping", C.c_int32),
        ("length_penalty", C.c_float),
        ("cand", C.c_void_p),
        ("cand_stride", C.c_int64),
        ("run_seq_in", C.c_void_p),
        ("run_seq_out", C.c_void_p),
        ("run_score_in", C.c_void_p),
        ("run_score_out", C.c_void_p),
        ("run_bidx_in", C.c_void_p),
        ("run_bidx_out", C.c_void_p),
        ("fin_seq_in", C.c_void_p),
        ("fin_seq_out", C.c_void_p),
        ("fin_score_in", C.c_void_p),
        ("fin_score_out", C.c_void_p),
        ("fin_bidx_in", C.c_void_p),
        ("fin_bidx_out", C.c_void_p),
        ("fin_done_in", C.c_void_p),
        ("fin_done_out", C.c_void_p),
        ("unsat", C.c_void_p),
        ("topk_score", C.c_void_p),
        ("topk_parent", C.c_void_p),
        ("topk_token", C.c_void_p),
        ("topk_hit", C.c_void_p),
        ("next_parent", C.c_void_p),
        ("plan_next", C.c_void_p),
        ("cand_shared", C.c_int32),
        ("reserved", C.c_int32),
    ]


class GrxBeamInitArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int32),
        ("K", C.c_int32),
        ("prompt_len", C.c_int32),
        ("max_length", C.c_int32),
        ("fill_token", C.c_int64),
        ("input_ids", C.c_void_p),
        ("ids_stride", C.c_int64),
        ("run_seq", C.c_void_p * 2),
        ("fin_seq", C.c_void_p * 2),
        ("run_score", C.c_void_p * 2),
        ("fin_score", C.c_void_p * 2),
        ("run_bidx", C.c_void_p * 2),
        ("fin_bidx", C.c_void_p * 2),
        ("fin_done", C.c_void_p * 2),
        ("unsat", C.c_void_p),
        ("plan_next", C.c_void_p),
        ("plan_one_per_group", C.c_int32),
        ("reserved", C.c_int32),
    ]


_lock = threading.Lock()
_lib = None


class GenrecHipError(RuntimeError):
    """A libgenrec_hip.so call returned a failure status."""


def _declare(lib: C.CDLL) -> None:
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.grx_abi_version.restype = C.c_int
    lib.grx_abi_version.argtypes = []
    lib.grx_last_error.restype = C.c_char_p
    lib.grx_last_error.argtypes = []
    lib.grx_trie_create.restype = C.c_int
    lib.grx_trie_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    lib.grx_trie_destroy.restype = C.c_int
    lib.grx_trie_destroy.argtypes = [vp]
    lib.grx_trie_max_fanout.restype = i32
    lib.grx_trie_max_fanout.argtypes = [vp]
    lib.grx_trie_max_token.restype = i32
    lib.grx_trie_max_token.argtypes = [vp]
    lib.grx_state_create.restype = C.c_int
    lib.grx_state_create.argtypes = [C.POINTER(vp)]
    lib.grx_state_destroy.restype = C.c_int
    lib.grx_state_destroy.argtypes = [vp]
    lib.grx_state_set_count.restype = C.c_int
    lib.grx_state_set_count.argtypes = [vp, i32, vp]
    lib.grx_state_clear_error.restype = C.c_int
    lib.grx_state_clear_error.argtypes = [vp, vp]
    lib.grx_state_read.restype = C.c_int
    lib.grx_state_read.argtypes = [vp, C.POINTER(i32 * 4), vp]
    lib.grx_workspace_bytes.restype = C.c_size_t
    lib.grx_workspace_bytes.argtypes = [i32]
    lib.grx_constrained_logsoftmax.restype = C.c_int
    lib.grx_constrained_logsoftmax.argtypes = [
        vp, vp, C.POINTER(GrxRows), C.POINTER(GrxConstraint), vp, i64, i32, vp, C.c_size_t, vp,
    ]
    lib.grx_reuse_bytes.restype = C.c_size_t
    lib.grx_reuse_bytes.argtypes = [i32]
    lib.grx_constrained_logsoftmax_reuse.restype = C.c_int
    lib.grx_constrained_logsoftmax_reuse.argtypes = [
        vp, vp, C.POINTER(GrxRows), C.POINTER(GrxConstraint), vp, i64, i32, vp, i32, vp, C.c_size_t, vp,
    ]
    lib.grx_constrained_candidates.restype = C.c_int
    lib.grx_constrained_candidates.argtypes = [
        vp, vp, C.POINTER(GrxRows), C.POINTER(GrxConstraint), vp, i64, i32, vp, C.c_size_t, vp,
    ]
    lib.grx_constrained_beam_step.restype = C.c_int
    lib.grx_constrained_beam_step.argtypes = [
        vp, vp, C.POINTER(GrxRows), C.POINTER(GrxConstraint), vp, i64, i32, vp, C.c_size_t, C.POINTER(GrxBeamStepArgs), vp,
    ]
    lib.grx_lm_head_scratch_bytes.restype = C.c_size_t
    lib.grx_lm_head_scratch_bytes.argtypes = [vp, i32, i32]
    lib.grx_lm_head_candidates.restype = C.c_int
    lib.grx_lm_head_candidates.argtypes = [
        vp, vp, C.POINTER(GrxRows), C.POINTER(GrxConstraint), C.POINTER(GrxLmHead), vp, C.c_size_t, vp, i64, i32, vp,
        C.c_size_t, vp,
    ]
    lib.grx_noalloc_max_rows.restype = i32
    lib.grx_noalloc_max_rows.argtypes = []
    lib.grx_constrained_candidates_noalloc.restype = C.c_int
    lib.grx_constrained_candidates_noalloc.argtypes = [
        vp, vp, C.POINTER(GrxRows), C.POINTER(GrxConstraint), vp, i64, i32, vp,
    ]
    lib.grx_chunk_parts.restype = i32
    lib.grx_chunk_parts.argtypes = [i32, i32, i32]
    lib.grx_constrained_partial.restype = C.c_int
    lib.grx_constrained_partial.argtypes = [
        vp, vp, C.POINTER(GrxRows), C.POINTER(GrxConstraint), C.POINTER(GrxChunk), vp, vp, i64, i32, vp, C.c_size_t, vp,
    ]
    lib.grx_constrained_finish.restype = C.c_int
    lib.grx_constrained_finish.argtypes = [i32, i32, vp, vp, i64, i32, vp, C.c_size_t, vp]
    lib.grx_constrained_finish_penalised.restype = C.c_int
    lib.grx_constrained_finish_penalised.argtypes = [
        C.POINTER(GrxRows), C.POINTER(GrxConstraint), i32, i32, vp, vp, i64, vp, i64, i32, vp, C.c_size_t, vp,
    ]
    lib.grx_candidate_stride.restype = i64
    lib.grx_candidate_stride.argtypes = [i32]
    lib.grx_timing_enable.restype = C.c_int
    lib.grx_timing_enable.argtypes = [C.c_int]
    lib.grx_timing_read.restype = C.c_int
    lib.grx_timing_read.argtypes = [C.POINTER(C.c_double), C.POINTER(i32)]
    lib.grx_read_probe.restype = C.c_int
    lib.grx_read_probe.argtypes = [vp, i64, i32, i64, vp, vp]
    lib.grx_beam_step.restype = C.c_int
    lib.grx_beam_step.argtypes = [C.POINTER(GrxBeamStepArgs), vp]
    lib.grx_beam_init.restype = C.c_int
    lib.grx_beam_init.argtypes = [C.POINTER(GrxBeamInitArgs), vp]
    lib.grx_sid_rewards.restype = C.c_int
    lib.grx_sid_rewards.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp]
    lib.grx_sid_rewards_tokens.restype = C.c_int
    lib.grx_sid_rewards_tokens.argtypes = [vp, i64, i32, vp, i64, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp]
    lib.grx_rollout_score_tokens.restype = C.c_int
    lib.grx_rollout_score_tokens.argtypes = [vp, i64, i32, vp, i64, i32, vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32,
                                             f32, vp, vp, vp, vp, vp]
    lib.grx_hit_ranks.restype = C.c_int
    lib.grx_hit_ranks.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    lib.grx_group_advantage.restype = C.c_int
    lib.grx_group_advantage.argtypes = [vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp]
    lib.grx_comm_unique_id.restype = C.c_int
    lib.grx_comm_unique_id.argtypes = [vp]
    lib.grx_comm_create.restype = C.c_int
    lib.grx_comm_create.argtypes = [vp, i32, i32, C.POINTER(vp)]
    lib.grx_comm_destroy.restype = C.c_int
    lib.grx_comm_destroy.argtypes = [vp]
    lib.grx_comm_allgather.restype = C.c_int
    lib.grx_comm_allgather.argtypes = [vp, vp, vp, C.c_size_t, vp]
    lib.grx_group_normalize_tokenwise.restype = C.c_int
    lib.grx_group_normalize_tokenwise.argtypes = [vp, vp, i32, i32, i32, f32, vp, vp]
    lib.grx_token_prefix_advantages.restype = C.c_int
    lib.grx_token_prefix_advantages.argtypes = [C.POINTER(GrxTokenPrefixArgs), vp]


def lib() -> C.CDLL:
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise GenrecHipError(
                        f"{LIB_PATH} not found: the HIP extension is not built "
                        "(run `make -C genrec_amd/csrc`); genrec_amd has no CPU fallback"
                    )
                handle = C.CDLL(LIB_PATH)
                _declare(handle)
                _lib = handle
    return _lib


def check(status: int) -> None:
    if status != GRX_OK:
        msg = lib().grx_last_error()
        raise GenrecHipError(f"libgenrec_hip status {status}: {msg.decode() if msg else '?'}")


def torch_dtype_code(dtype) -> int:
    import torch

    if dtype == torch.float32:
        return GRX_F32
    if dtype == torch.bfloat16:
        return GRX_BF16
    if dtype == torch.float16:
        return GRX_F16
    raise TypeError(f"unsupported logits dtype {dtype}; expected float32, bfloat16 or float16")


def require_cuda(t, what: str) -> None:
    if not t.is_cuda:
        raise GenrecHipError(
            f"{what} must live on the GPU (got device {t.device}); genrec_amd runs on MI355X only and has no CPU path"
        )


def current_stream_ptr(device) -> int:
    """Raw handle of torch's current stream on ``device`` (a torch.device or a device index)."""
    import torch

    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:  # no Stream object per call: this sits on every step's path
        index = device if isinstance(device, int) else device.index
        return int(raw(torch.cuda.current_device() if index is None else index))
    return int(torch.cuda.current_stream(device).cuda_stream)


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def on_device(index: int):
    """Context in which ``index`` is the current device; costs nothing when it already is."""
    import torch

    return _NO_SWITCH if torch.cuda.current_device() == index else torch.cuda.device(index)
