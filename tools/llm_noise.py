import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
from cogstream_amd.llm import Qwen2Engine
from cogstream_amd.weights import LlmConfig, random_llm_state
dev = torch.device("cuda:0")
cfg = LlmConfig()
st = random_llm_state(cfg, 2, dev, torch.bfloat16)
eng = Qwen2Engine(st, cfg, dtype=torch.bfloat16, device=dev)
torch.manual_seed(3)
S = 1500
emb = (torch.randn(S + 1, cfg.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
ref = eng.forward(emb, None)["logits"]
cache = eng.new_cache(S + 8)
eng.forward(emb[:S], cache, want_logits=False)
got = eng.forward(emb[S:], cache)["logits"]
# fp32 truth on the same (bf16-valued) weights and inputs
st32 = {k: v.float() for k, v in st.items()}
eng32 = Qwen2Engine(st32, cfg, dtype=torch.float32, device=dev)
truth = eng32.forward(emb.float(), None)["logits"]
mx = float(truth.abs().max())
print("max |truth|", mx)
print("prefill bf16 vs fp32 truth: max abs diff", float((ref - truth).abs().max()), "rel", float((ref - truth).abs().max()) / mx)
print("decode  bf16 vs fp32 truth: max abs diff", float((got - truth).abs().max()), "rel", float((got - truth).abs().max()) / mx)
print("decode vs prefill (bf16):   max abs diff", float((got - ref).abs().max()), "rel", float((got - ref).abs().max()) / mx)
print("rms: prefill", float((ref - truth).pow(2).mean().sqrt()), "decode", float((got - truth).pow(2).mean().sqrt()))
