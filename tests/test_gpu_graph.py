"""GPU: whole generations recorded as a HIP graph (``genrec_amd.graph.capture``) and replayed.

Three generations -- beam state reset with the first call planned ahead, depth x the one-call decode step,
results copied out -- are recorded once; replays over refilled input buffers must leave exactly what the same
calls issued eagerly leave: final sequences, beam parents, scores, and the processor's counter
(the loop of transformers' generate() that MIMIGenRec.py:79 / evaluate.py:363-371 enter)."""

import pytest
import torch

from genrec_amd.beam import BeamSearchState
from genrec_amd.fixed_hint_logit_processor import FixedHintConstrainedLogitsProcessor
from genrec_amd.graph import capture
from genrec_amd.logit_processor import Candidates, ConstrainedLogitsProcessor
from genrec_amd.synth import sample_decode_rows
from test_gpu_fused_step import _setup
from test_gpu_processors import _dev

pytestmark = pytest.mark.gpu


class _Loop:
    def __init__(self, cls, trie_name, K, B, dtype, penalty=1.0, generations=3):
        self.dev = _dev()
        self.K, self.B, self.G = K, B, generations
        self.tok, self.trie, make, self.levels = _setup(trie_name, K, cls)
        self.proc = make()
        if penalty != 1.0:
            self.proc.repetition_penalty = penalty
        self.proc.sync_errors = False
        V = self.tok.vocab_size_padded
        self.prompts = [torch.empty((B * K, 24), dtype=torch.int64, device=self.dev) for _ in range(generations)]
        self.logits = [torch.empty((B * K, V), dtype=dtype, device=self.dev) for _ in range(self.levels)]
        self.fill(0)
        T = self.prompts[0].shape[1]
        self.state = BeamSearchState(self.prompts[0], K, T + self.levels + 1, self.tok.eos_token_id, self.tok.pad_token_id,
                                     1.0, False, shared_prompt=True)
        self.keep = self.proc.keep_top_for(self.state.n_keep)
        self.cand = Candidates(B * K, self.proc.candidate_width(self.state.n_keep), self.dev)
        L = self.state.max_length
        self.out_seq = torch.zeros((generations, B, K, L), dtype=torch.int64, device=self.dev)
        self.out_par = torch.zeros((generations, B, K, L - T), dtype=torch.int32, device=self.dev)
        self.out_score = torch.zeros((generations, B, K), dtype=torch.float32, device=self.dev)

    def fill(self, seed):
        """New prompts and logits, written into the buffers the recorded work reads."""
        g = torch.Generator(device=self.dev).manual_seed(100 + seed)
        for i, p in enumerate(self.prompts):
            rows = torch.from_numpy(sample_decode_rows(self.trie, self.B, 1, 0, prompt_len=21, seed=7 * seed + i))
            p.copy_(rows.to(self.dev).repeat_interleave(self.K, dim=0))
        for x in self.logits:
            x.copy_((torch.randn(x.shape, generator=g, device=self.dev) * 4).to(x.dtype))

    def generations(self):
        for gi in range(self.G):
            self.proc.count = 0
            self.state.reset(self.prompts[gi], self.proc)
            for d in range(self.levels):
                self.state.fused_step(self.proc, self.logits[d], self.cand, keep_top=self.keep,
                                      planner=self.proc if d + 1 < self.levels else None)
            self.out_seq[gi].copy_(self.state.sequences)
            self.out_par[gi].copy_(self.state.beam_indices)
            self.out_score[gi].copy_(self.state.beam_scores)

    def results(self):
        torch.cuda.synchronize()
        self.proc.check_errors()
        count = self.proc.count if isinstance(self.proc, ConstrainedLogitsProcessor) else None  # the fixed-hint one is stateless
        return self.out_seq.clone(), self.out_par.clone(), self.out_score.clone(), count


@pytest.mark.parametrize("cls,trie_name,K,B,dtype,penalty", [
    (ConstrainedLogitsProcessor, "Instruments", 16, 4, torch.bfloat16, 1.0),    # the RL launcher's 64 rows per device
    (ConstrainedLogitsProcessor, "Arts", 50, 8, torch.bfloat16, 1.1),           # the evaluation launcher's 8 x 50 rows
    (FixedHintConstrainedLogitsProcessor, "Instruments", 16, 40, torch.float32, 1.0),
])
def test_replayed_generations_equal_eager_ones(cls, trie_name, K, B, dtype, penalty):
    loop = _Loop(cls, trie_name, K, B, dtype, penalty)
    captured = capture(loop.generations, [loop.proc])
    counts = cls is ConstrainedLogitsProcessor
    for seed in (1, 2):
        loop.fill(seed)
        loop.generations()  # eager
        want = loop.results()
        for t in (loop.out_seq, loop.out_par, loop.out_score):
            t.zero_()
        captured.replay()
        got = loop.results()
        assert got[3] == want[3] == (loop.levels if counts else None)  # the counter after a whole generation
        for g, w in zip(got[:3], want[:3]):
            assert torch.equal(g, w)
        live = want[2] > -1.0e8
        assert int(live.sum()) > 0
    captured.replay()  # back to back on unchanged inputs
    again = loop.results()
    for g, w in zip(again[:3], want[:3]):
        assert torch.equal(g, w)


def test_a_single_generation_of_a_depth_not_divisible_by_three_is_refused():
    """Five count-based calls leave the state ring in another phase than they found it: a replay would read
    the wrong slot.  capture() says so instead of recording it."""
    loop = _Loop(ConstrainedLogitsProcessor, "Instruments", 16, 4, torch.bfloat16, generations=1)
    assert loop.levels % 3 != 0
    with pytest.raises(ValueError, match="multiple of"):
        capture(loop.generations, [loop.proc])
    # the processor and the state are still usable
    loop.fill(3)
    loop.generations()
    assert loop.results()[3] == loop.levels
