import os, sys, tempfile, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from transformers import Qwen2Config, Qwen2ForCausalLM
from genrec_amd.generate import constrained_generate
from genrec_amd.logit_processor import ConstrainedLogitsProcessor
from genrec_amd.synth import ASSISTANT_PREFIX, SidTokenizer, sample_decode_rows, write_index_file
from genrec_amd.trie import build_trie_from_index, create_prefix_allowed_tokens_fn
dev = torch.device("cuda", 0)
B, K, T, hidden, layers = 512, 16, 61, 2048, 4
path = write_index_file("Instruments", tempfile.mkdtemp())
tok = SidTokenizer.from_index(path)
trie, suffix, pidx = build_trie_from_index(path, tok, prefix=ASSISTANT_PREFIX)
fn = create_prefix_allowed_tokens_fn(trie, suffix)
V = tok.vocab_size_padded
torch.manual_seed(0)
cfg = Qwen2Config(vocab_size=V, hidden_size=hidden, intermediate_size=hidden * 3, num_hidden_layers=layers,
                  num_attention_heads=hidden // 128, num_key_value_heads=max(1, hidden // 512),
                  max_position_embeddings=1024, eos_token_id=tok.eos_token_id, pad_token_id=tok.pad_token_id,
                  bos_token_id=None, tie_word_embeddings=False)
model = Qwen2ForCausalLM(cfg).to(torch.bfloat16).to(dev).eval()
prompts = torch.from_numpy(sample_decode_rows(trie, B, 1, 0, prompt_len=T, seed=1)).to(dev)
mask = torch.ones_like(prompts)
def run():
    proc = ConstrainedLogitsProcessor(fn, K, pidx, suffix, tok.eos_token_id)
    return constrained_generate(model, prompts, proc, K, 5, attention_mask=mask, eos_token_id=tok.eos_token_id,
                                pad_token_id=tok.pad_token_id, length_penalty=1.0, max_steps=5)
run(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as p:
    run(); torch.cuda.synchronize()
print(p.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70))
