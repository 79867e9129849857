#!/usr/bin/env python3
"""Time grx_beam_step of build variants (build/variants/lib_*.so) on synthetic candidates.
Shape from the environment: AB_B groups, AB_K beams, AB_CMAX row width, AB_FANS comma-separated fan-outs."""
import ctypes as C
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genrec_amd import _lib  # noqa: E402
from genrec_amd.beam import BeamSearchState  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    B, K, cmax, V = int(os.environ.get('AB_B', 512)), int(os.environ.get('AB_K', 16)), int(os.environ.get('AB_CMAX', 162)), 152704
    fans = [int(x) for x in os.environ.get('AB_FANS', '162,43,2').split(',')]
    R = B * K
    prompts = torch.randint(5, 90, (R, 64), device=dev)
    names = sys.argv[1:] or sorted(os.path.basename(p)[4:-3] for p in glob.glob(os.path.join(ROOT, "build/variants/lib_*.so")))
    for fan in fans:
        cnt = torch.full((R,), fan, dtype=torch.int32, device=dev)
        tok = (torch.arange(cmax, device=dev, dtype=torch.int32)[None, :] * 3 + 151665).repeat(R, 1).contiguous()
        lp = -torch.rand((R, cmax), device=dev) * 20
        from genrec_amd.logit_processor import Candidates
        cand = Candidates.from_parts(cnt, tok, lp, dev)
        for name in names:
            h = C.CDLL(os.path.join(ROOT, "build/variants", f"lib_{name}.so"))
            _lib._declare(h)
            real = _lib._lib
            _lib._lib = h
            try:
                st = BeamSearchState(prompts, K, 64 + 6, 151645, 151643, 1.0, False, shared_prompt=True)
                # two real steps bring the state to depth 2; the timing loop then replays hand-built
                # argument blocks (ping-pong between the double buffers) without Python in between
                st.reset(prompts)
                for _ in range(2):
                    st.step(cand, V)
                ts = []
                stream = torch.cuda.current_stream().cuda_stream
                # rebuild the two arg blocks by hand (depth 2 -> 3 and back), same buffers as the state uses
                def block(i, o, cur_len):
                    a = _lib.GrxBeamStepArgs()
                    a.B, a.K, a.V, a.n_keep, a.cmax = st.B, st.K, V, st.n_keep, cmax
                    a.cur_len, a.prompt_len, a.max_length, a.copy_from = cur_len, st.prompt_len, st.max_length, st.copy_from
                    a.eos_token_id, a.early_stopping, a.length_penalty = 151645, 0, 1.0
                    a.cand, a.cand_stride = cand.buf.data_ptr(), cand.stride
                    for nm in ("run_seq", "run_score", "run_bidx", "fin_seq", "fin_score", "fin_bidx", "fin_done"):
                        setattr(a, nm + "_in", getattr(st, nm)[i].data_ptr())
                        setattr(a, nm + "_out", getattr(st, nm)[o].data_ptr())
                    a.unsat = st.unsat.data_ptr()
                    return a
                blocks = [block(0, 1, st.prompt_len + 2), block(1, 0, st.prompt_len + 3)]
                for rep in range(6):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for it in range(40):
                        h.grx_beam_step(C.byref(blocks[it & 1]), stream)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 40 * 1e3)
                if hasattr(h, "grx_debug_read_stamps"):
                    import numpy as np
                    buf = np.zeros(1024 * 16, dtype=np.uint64)
                    h.grx_debug_read_stamps(buf.ctypes.data_as(C.c_void_p))
                    st_ = buf.reshape(1024, 16)[: min(B, 1024)].astype(np.int64)
                    t0 = st_[:, 0].min()
                    rel = (st_ - t0) * 10  # ns
                    print("  stamps (ns from first WG start): median per stamp", [int(x) for x in np.median(rel, axis=0)[:15]])
                    print("  per-WG deltas median", [int(x) for x in np.median(np.diff(st_[:, :15], axis=1) * 10, axis=0)])
                    print("  WG start spread (ns)", int((st_[:, 0].max() - t0) * 10), " last end", int((st_[:, 14].max() - t0) * 10))
                print(f"fan={fan:4d} {name:8s} per-step {sorted(ts)[len(ts)//2]:7.1f} us (min {min(ts):.1f})")
            finally:
                _lib._lib = real


if __name__ == "__main__":
    main()
