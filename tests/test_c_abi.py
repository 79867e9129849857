"""The C ABI from plain C: include/genrec_hip.h compiles as C99 and a gcc-built client drives the
library without Python or torch in the loop."""

import ctypes as C
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "abi_client.c")
PREBUILT = os.path.join(ROOT, "tests", "c_abi", "abi_client")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _facts(stdout: str) -> dict:
    lines = stdout.strip().splitlines()
    assert lines and lines[-1] == "OK", stdout
    return dict(line.split("=", 1) for line in lines[:-1])


def _compile(out_path: str) -> None:
    libdir = os.path.join(ROOT, "genrec_amd")
    subprocess.run(
        ["gcc", "-std=c99", "-pedantic-errors", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
         SRC, "-o", out_path, "-L", libdir, "-lgenrec_hip", "-L", os.path.join(ROCM, "lib"), "-lamdhip64", "-lm",
         f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{os.path.join(ROCM, 'lib')}"],
        check=True, capture_output=True, text=True,
    )


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_header_is_c99_and_the_python_mirror_has_the_same_layout(tmp_path):
    from genrec_amd import _lib

    exe = str(tmp_path / "abi_client")
    _compile(exe)
    run = subprocess.run([exe, "host"], check=True, capture_output=True, text=True, timeout=60)
    facts = _facts(run.stdout)
    assert int(facts["abi_version"]) == _lib.lib().grx_abi_version()
    # what the C compiler lays out is what the ctypes mirror lays out
    assert int(facts["sizeof_grx_rows"]) == C.sizeof(_lib.GrxRows)
    assert int(facts["sizeof_grx_constraint"]) == C.sizeof(_lib.GrxConstraint)
    assert int(facts["sizeof_grx_plan_request"]) == C.sizeof(_lib.GrxPlanRequest)
    assert int(facts["sizeof_grx_beam_step_args"]) == C.sizeof(_lib.GrxBeamStepArgs)
    assert int(facts["sizeof_grx_beam_init_args"]) == C.sizeof(_lib.GrxBeamInitArgs)
    assert int(facts["sizeof_grx_chunk"]) == C.sizeof(_lib.GrxChunk)
    assert int(facts["sizeof_grx_lm_head"]) == C.sizeof(_lib.GrxLmHead)
    assert int(facts["noalloc_max_rows"]) == 65536
    assert int(facts["candidate_stride_242"]) == 256
    assert "null trie" in facts["null_trie_message"]


@pytest.mark.gpu
def test_plain_c_client_decodes_one_step_on_the_gpu(tmp_path):
    """Trie upload, count-based compact candidates and a beam step driven from C; the client checks
    the results against its own float64 log-softmax."""
    exe = PREBUILT
    newest_input = max(os.path.getmtime(p) for p in (SRC, os.path.join(ROOT, "include", "genrec_hip.h")))
    if not os.path.exists(exe) or os.path.getmtime(exe) < newest_input:  # normally built by __graft_entry__.build()
        exe = str(tmp_path / "abi_client")
        _compile(exe)
    run = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    facts = _facts(run.stdout)
    assert float(facts["max_abs_logp_error"]) <= 1e-5  # fp32 tolerance of the log-probs
    assert int(facts["beam_step_groups_checked"]) == 3
    assert int(facts["one_call_step_checked"]) == 1  # grx_constrained_beam_step == the two calls, bit for bit
    assert int(facts["noalloc_rows_identical"]) == 12
    assert float(facts["sliced_max_abs_logp_difference"]) <= 1e-5
    # dense form kept from call to call == a fresh dense result, on both calls (4 tokens, then 1 per row)
    assert int(facts["dense_kept_call_0_finite"]) == 48 and int(facts["dense_kept_call_1_finite"]) == 12
