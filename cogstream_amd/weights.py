"""Weight containers and the one-off packing into the layouts the HIP kernels expect.

Names follow the reference checkpoint (model/model.safetensors.index.json; module tree
model/cogreasoner_chat.py:214-248,591-598): model.vision_encoder.*, model.mm_projector.readout.{0,2}.*,
model.embed_tokens, model.layers.N.*, model.norm, lm_head. Packing rules are documented in include/cogs.h:
  * q/k rows of every head interleaved as rotary pairs (d, d+hd/2) -> (2i, 2i+1);
  * q,k,v stacked into one [.., hidden] matrix; gate/up rows interleaved;
  * K dimensions zero-padded to the GEMM slab (64 bf16 / 32 fp32 elements).
torch is used here for tensor storage and the one-time reshuffles only."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import _lib as L
from .ops import k_slab, pad_cols


@dataclass
class VisionConfig:
    """model/config.json:35-42 (vision_encoder_config) + configuration_videollama3_encoder.py:22-49"""
    hidden_size: int = 1152
    intermediate_size: int = 4304
    num_hidden_layers: int = 27
    num_attention_heads: int = 16
    num_channels: int = 3
    patch_size: int = 14
    layer_norm_eps: float = 1e-6

    @property
    def patch_dim(self) -> int:
        return self.num_channels * self.patch_size * self.patch_size


@dataclass
class LlmConfig:
    """model/config.json:12-43"""
    hidden_size: int = 3584
    intermediate_size: int = 18944
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    vocab_size: int = 152064
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    image_token_index: int = 151665
    eos_token_id: int = 151645

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


def rope_perm(hd: int) -> torch.Tensor:
    """new row 2i <- old i, 2i+1 <- old i + hd/2"""
    return torch.arange(hd).view(2, hd // 2).t().reshape(-1)


def _perm_heads(w: torch.Tensor, heads: int, hd: int) -> torch.Tensor:
    p = rope_perm(hd).to(w.device)
    shp = w.shape
    return w.reshape(heads, hd, *shp[1:])[:, p].reshape(shp)


def _pad_rows(x: torch.Tensor, rows: int) -> torch.Tensor:
    if x.shape[0] == rows:
        return x.contiguous()
    out = x.new_zeros(rows, *x.shape[1:])
    out[: x.shape[0]] = x
    return out


def zero_sum_rows(wf: torch.Tensor) -> torch.Tensor:
    """Make every row of a bf16 matrix sum to (essentially) zero by moving a few of its entries ONE bf16 step.
    A centred row rounded to bf16 keeps a row sum s of about sqrt(K/12) steps; the folded-LayerNorm GEMM never evaluates
    the term rstd * mean * s (include/cogs.h, cogs_gemm_desc.ln_ab), so s is an error that grows with |mean / std| of an
    activation row. Greedy, exact in fp64: starting at the row's most populated binade (many candidates, small steps)
    and going down binade by binade, floor(|s| / step) entries of that binade move one step against the sign of s. What
    is left is smaller than one step of the row's smallest entries. A moved entry is still a correctly rounded bf16
    neighbour of its exact value (error < 1.5 steps instead of <= 0.5); about ten of 1152 entries move."""
    assert wf.dtype == torch.bfloat16 and wf.dim() == 2
    w = wf.double()
    mag = w.abs()
    tiny = 2.0 ** -126
    expo = torch.floor(torch.log2(mag.clamp_min(tiny)))
    # log2 of an exact power of two is exact in fp64, but guard the binade edges against a 1-ulp slip of log2
    expo = torch.where(torch.ldexp(torch.ones_like(mag), expo.int()) > mag, expo - 1, expo)
    expo = torch.where(torch.ldexp(torch.ones_like(mag), (expo + 1).int()) <= mag, expo + 1, expo)
    live = mag >= 2.0 ** -100                       # zeros (padding rows / columns) and denormal dust never move
    s = w.sum(dim=1)
    if not bool(live.any()):
        return wf
    e_hi = int(expo[live].max().item())
    e_lo = int(expo[live].min().item())
    # per row: the most populated binade
    levels = torch.arange(e_lo, e_hi + 1, device=w.device, dtype=torch.float64)
    counts = torch.stack([((expo == lv) & live).sum(dim=1) for lv in levels.tolist()], dim=1)       # [rows, levels]
    start = levels[counts.argmax(dim=1)]                                                              # [rows]
    start_max = float(start.max().item())
    for lv in reversed(levels.tolist()):
        if lv > start_max:
            continue
        step = 2.0 ** (lv - 7)                      # bf16: 8 significant bits
        want = torch.floor(s.abs() / step)
        if not bool((want > 0).any()):               # no row has a whole step of this binade left to shed
            continue
        cand = (expo == lv) & live & (start >= lv)[:, None]
        rank = torch.cumsum(cand.to(torch.int32), dim=1)
        take = cand & (rank.double() <= want[:, None])
        delta = -torch.sign(s)[:, None] * step * take.double()
        w = w + delta
        s = s + delta.sum(dim=1)
    out = w.to(torch.bfloat16)
    assert bool((out.double() == w).all()), "a moved weight is not a bf16 value"
    return out


def fold_layernorm(w: torch.Tensor, b: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """Linear(LayerNorm(x)) with the affine part folded into the linear layer (modeling_videollama3_encoder.py:382-391):
    LN(x) W^T + b = rstd (x - mean) (W diag(gamma))^T + (b + W beta). The rows of W' = W diag(gamma) are CENTRED (their
    mean over the input dimension subtracted; zero-padded input columns excluded): sum_k x_k W''[n][k] then equals
    sum_k (x_k - mean) W'[n][k] and the GEMM epilogue only evaluates rstd * acc + c (cogs_gemm_desc.ln_ab / col_c). In
    bf16 the ROUNDED rows are made zero-sum as well (zero_sum_rows), which is what the contract in include/cogs.h asks
    for. Returns (W'' in W's dtype, s = fp32 row sums of the stored W'' (~0; for tests), c = fp32 b + W beta)."""
    wd = w.double() * gamma.double()[None, :]
    wd = wd - wd.mean(dim=1, keepdim=True)
    wf = wd.to(w.dtype)
    if wf.dtype == torch.bfloat16:
        wf = zero_sum_rows(wf)
    s = wf.double().sum(dim=1).float().contiguous()
    c = (b.double() + w.double() @ beta.double()).float().contiguous()
    return wf.contiguous(), s, c


class PackedVit:
    def __init__(self, state: Dict[str, torch.Tensor], cfg: VisionConfig, dtype=torch.bfloat16, device="cuda",
                 fold_ln: Optional[bool] = None):
        """fold_ln (default: bf16 yes, fp32 no): fold layer_norm1 into the QKV projection and layer_norm2 into fc1"""
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.fold_ln = (dtype == torch.bfloat16 and cfg.hidden_size % 64 == 0) if fold_ln is None else bool(fold_ln)
        slab = k_slab(dtype)
        H, I, hd = cfg.hidden_size, cfg.intermediate_size, cfg.hidden_size // cfg.num_attention_heads
        self.inter_pad = (I + slab - 1) // slab * slab
        self.patch_pad = (cfg.patch_dim + slab - 1) // slab * slab
        cv = lambda t: t.to(device=self.device, dtype=dtype)
        self.keep: List[torch.Tensor] = []

        def hold(t):
            t = t.contiguous()
            self.keep.append(t)
            return t

        g = lambda k: cv(state[k])
        self.patch_w = hold(pad_cols(g("embeddings.patch_embedding.weight").reshape(H, -1), slab))
        self.patch_b = hold(g("embeddings.patch_embedding.bias"))
        self.post_g = hold(g("post_layernorm.weight"))
        self.post_b = hold(g("post_layernorm.bias"))
        self.layers = (L.VitLayer * cfg.num_hidden_layers)()
        for i in range(cfg.num_hidden_layers):
            p = f"encoder.layers.{i}."
            a = p + "self_attn."
            qw = _perm_heads(g(a + "q_proj.weight"), cfg.num_attention_heads, hd)
            kw = _perm_heads(g(a + "k_proj.weight"), cfg.num_attention_heads, hd)
            qb = _perm_heads(g(a + "q_proj.bias"), cfg.num_attention_heads, hd)
            kb = _perm_heads(g(a + "k_proj.bias"), cfg.num_attention_heads, hd)
            lay = self.layers[i]
            ln1_g, ln1_b = hold(g(p + "layer_norm1.weight")), hold(g(p + "layer_norm1.bias"))
            lay.ln1_g, lay.ln1_b = ln1_g.data_ptr(), ln1_b.data_ptr()
            qkv_w = torch.cat([qw, kw, g(a + "v_proj.weight")], 0)
            qkv_b = hold(torch.cat([qb, kb, g(a + "v_proj.bias")], 0))
            if self.fold_ln:
                qkv_w, _, c_ = fold_layernorm(qkv_w, qkv_b, ln1_g, ln1_b)
                lay.qkv_c = hold(c_).data_ptr()
            lay.qkv_w = hold(qkv_w).data_ptr()
            lay.qkv_b = qkv_b.data_ptr()
            lay.o_w = hold(g(a + "out_proj.weight")).data_ptr()
            lay.o_b = hold(g(a + "out_proj.bias")).data_ptr()
            ln2_g, ln2_b = hold(g(p + "layer_norm2.weight")), hold(g(p + "layer_norm2.bias"))
            lay.ln2_g, lay.ln2_b = ln2_g.data_ptr(), ln2_b.data_ptr()
            fc1_w = _pad_rows(g(p + "mlp.fc1.weight"), self.inter_pad)
            fc1_b = hold(_pad_rows(g(p + "mlp.fc1.bias"), self.inter_pad))
            if self.fold_ln:
                fc1_w, _, c_ = fold_layernorm(fc1_w, fc1_b, ln2_g, ln2_b)
                lay.fc1_c = hold(c_).data_ptr()
            lay.fc1_w = hold(fc1_w).data_ptr()
            lay.fc1_b = fc1_b.data_ptr()
            lay.fc2_w = hold(pad_cols(g(p + "mlp.fc2.weight"), slab)).data_ptr()
            lay.fc2_b = hold(g(p + "mlp.fc2.bias")).data_ptr()
        w = L.VitWeights()
        w.dtype = L.dtype_code(dtype)
        w.hidden, w.inter_pad, w.layers, w.heads = H, self.inter_pad, cfg.num_hidden_layers, cfg.num_attention_heads
        w.patch_dim, w.patch_pad, w.ln_eps = cfg.patch_dim, self.patch_pad, cfg.layer_norm_eps
        w.patch_w, w.patch_b = self.patch_w.data_ptr(), self.patch_b.data_ptr()
        w.post_ln_g, w.post_ln_b = self.post_g.data_ptr(), self.post_b.data_ptr()
        w.layer = C.cast(self.layers, C.POINTER(L.VitLayer))
        self.struct = w


class PackedProjector:
    def __init__(self, state: Dict[str, torch.Tensor], dtype=torch.bfloat16, device="cuda"):
        cv = lambda t: t.to(device=device, dtype=dtype).contiguous()
        self.w1, self.b1 = cv(state["readout.0.weight"]), cv(state["readout.0.bias"])
        self.w2, self.b2 = cv(state["readout.2.weight"]), cv(state["readout.2.bias"])
        w = L.ProjWeights()
        w.dtype = L.dtype_code(dtype)
        w.in_dim, w.out_dim = self.w1.shape[1], self.w1.shape[0]
        w.w1, w.b1, w.w2, w.b2 = self.w1.data_ptr(), self.b1.data_ptr(), self.w2.data_ptr(), self.b2.data_ptr()
        self.struct = w
        self.in_dim, self.out_dim = w.in_dim, w.out_dim


class PackedLlm:
    def __init__(self, state: Dict[str, torch.Tensor], cfg: LlmConfig, dtype=torch.bfloat16, device="cuda"):
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        hd, hq, hkv = cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads
        cv = lambda t: t.to(device=self.device, dtype=dtype)
        self.keep: List[torch.Tensor] = []

        def hold(t):
            t = t.contiguous()
            self.keep.append(t)
            return t

        g = lambda k: cv(state[k])
        self.embed = hold(g("embed_tokens.weight"))
        self.final_norm = hold(g("norm.weight"))
        self.lm_head = hold(g("lm_head.weight"))
        self.layers = (L.LlmLayer * cfg.num_hidden_layers)()
        for i in range(cfg.num_hidden_layers):
            p = f"layers.{i}."
            a = p + "self_attn."
            qw, qb = _perm_heads(g(a + "q_proj.weight"), hq, hd), _perm_heads(g(a + "q_proj.bias"), hq, hd)
            kw, kb = _perm_heads(g(a + "k_proj.weight"), hkv, hd), _perm_heads(g(a + "k_proj.bias"), hkv, hd)
            lay = self.layers[i]
            lay.in_ln = hold(g(p + "input_layernorm.weight")).data_ptr()
            lay.qkv_w = hold(torch.cat([qw, kw, g(a + "v_proj.weight")], 0)).data_ptr()
            lay.qkv_b = hold(torch.cat([qb, kb, g(a + "v_proj.bias")], 0)).data_ptr()
            lay.o_w = hold(g(a + "o_proj.weight")).data_ptr()
            lay.post_ln = hold(g(p + "post_attention_layernorm.weight")).data_ptr()
            gu = torch.stack([g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight")], dim=1)
            lay.gu_w = hold(gu.reshape(2 * cfg.intermediate_size, cfg.hidden_size)).data_ptr()
            lay.down_w = hold(g(p + "mlp.down_proj.weight")).data_ptr()
        w = L.LlmWeights()
        w.dtype = L.dtype_code(dtype)
        w.hidden, w.inter, w.layers = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
        w.heads, w.kv_heads, w.head_dim, w.vocab = hq, hkv, hd, cfg.vocab_size
        w.rms_eps, w.rope_theta = cfg.rms_norm_eps, cfg.rope_theta
        w.final_norm, w.lm_head = self.final_norm.data_ptr(), self.lm_head.data_ptr()
        w.layer = C.cast(self.layers, C.POINTER(L.LlmLayer))
        self.struct = w


# ---- random initialisation at arbitrary dimensions (no checkpoints are reachable: README.md:56-58) ----

def random_vit_state(cfg: VisionConfig, seed: int = 0, device="cpu", dtype=torch.float32, std: float = 0.02):
    g = torch.Generator(device=device).manual_seed(seed)
    r = lambda *s: (torch.randn(*s, generator=g, device=device, dtype=torch.float32) * std).to(dtype)
    H, I = cfg.hidden_size, cfg.intermediate_size
    st = {
        "embeddings.patch_embedding.weight": r(H, cfg.num_channels, cfg.patch_size, cfg.patch_size),
        "embeddings.patch_embedding.bias": r(H),
        "post_layernorm.weight": 1 + r(H), "post_layernorm.bias": r(H),
    }
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            st[p + f"self_attn.{n}.weight"] = r(H, H)
            st[p + f"self_attn.{n}.bias"] = r(H)
        for n in ("layer_norm1", "layer_norm2"):
            st[p + n + ".weight"] = 1 + r(H)
            st[p + n + ".bias"] = r(H)
        st[p + "mlp.fc1.weight"], st[p + "mlp.fc1.bias"] = r(I, H), r(I)
        st[p + "mlp.fc2.weight"], st[p + "mlp.fc2.bias"] = r(H, I), r(H)
    return st


def random_proj_state(in_dim: int, out_dim: int, seed: int = 1, device="cpu", dtype=torch.float32, std: float = 0.02):
    g = torch.Generator(device=device).manual_seed(seed)
    r = lambda *s: (torch.randn(*s, generator=g, device=device, dtype=torch.float32) * std).to(dtype)
    return {"readout.0.weight": r(out_dim, in_dim), "readout.0.bias": r(out_dim),
            "readout.2.weight": r(out_dim, out_dim), "readout.2.bias": r(out_dim)}


def random_llm_state(cfg: LlmConfig, seed: int = 2, device="cpu", dtype=torch.float32, std: float = 0.02):
    g = torch.Generator(device=device).manual_seed(seed)
    r = lambda *s: (torch.randn(*s, generator=g, device=device, dtype=torch.float32) * std).to(dtype)
    H, I, hd = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
    kvd = cfg.num_key_value_heads * hd
    st = {"embed_tokens.weight": r(cfg.vocab_size, H), "norm.weight": 1 + r(H), "lm_head.weight": r(cfg.vocab_size, H)}
    for i in range(cfg.num_hidden_layers):
        p = f"layers.{i}."
        st[p + "input_layernorm.weight"] = 1 + r(H)
        st[p + "post_attention_layernorm.weight"] = 1 + r(H)
        st[p + "self_attn.q_proj.weight"], st[p + "self_attn.q_proj.bias"] = r(H, H), r(H)
        st[p + "self_attn.k_proj.weight"], st[p + "self_attn.k_proj.bias"] = r(kvd, H), r(kvd)
        st[p + "self_attn.v_proj.weight"], st[p + "self_attn.v_proj.bias"] = r(kvd, H), r(kvd)
        st[p + "self_attn.o_proj.weight"] = r(H, H)
        st[p + "mlp.gate_proj.weight"], st[p + "mlp.up_proj.weight"] = r(I, H), r(I, H)
        st[p + "mlp.down_proj.weight"] = r(H, I)
    return st


# ---------------------------------------------------------------------------------------------------------
# LoRA adapters. evaluate/answer_generate.py:181-182 loads two peft adapters
# ("full_module" for generate, "language_module" for qa_selection; r = 8, alpha = 16 on q,k,v,o,gate,up,down of
# the 28 decoder layers, the second-stage one also on both projector linears: train/second_stage_training.py:
# 241-264, first_stage_training.py:447-465) and switches them with set_adapter (:71-73). Here every adapter gets
# its own MERGED copy of the targeted matrices (W + alpha/r * B.A, summed in fp32, rounded once): 288 GB of HBM
# hold several 15 GB weight sets, the kernels stay untouched and switching is a pointer-table swap.

LORA_LLM_TARGETS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                    "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")


def _lora_lookup(lora_state: Dict[str, torch.Tensor], module: str):
    """(A [r,in], B [out,r]) of `module` under any of the key spellings peft writes / the tests use"""
    for prefix in ("", "base_model.model.model.", "base_model.model.", "model."):
        for mid in ("", ".default"):
            a = lora_state.get(f"{prefix}{module}.lora_A{mid}.weight")
            b = lora_state.get(f"{prefix}{module}.lora_B{mid}.weight")
            if a is not None and b is not None:
                return a, b
    return None


def merge_lora(llm_state: Dict[str, torch.Tensor], proj_state: Optional[Dict[str, torch.Tensor]],
               lora_state: Dict[str, torch.Tensor], cfg: LlmConfig, lora_alpha: float = 16.0):
    """-> (llm_state', proj_state') with W' = W + (alpha / r) * B @ A for every targeted module; untouched
    tensors are shared with the inputs (no copy)."""
    def merged(w, ab):
        a, b = ab
        r = a.shape[0]
        d = (b.to(w.device, torch.float32) @ a.to(w.device, torch.float32)) * (lora_alpha / r)
        return (w.float() + d).to(w.dtype)

    out = dict(llm_state)
    for i in range(cfg.num_hidden_layers):
        for t in LORA_LLM_TARGETS:
            ab = _lora_lookup(lora_state, f"layers.{i}.{t}")
            if ab is not None:
                out[f"layers.{i}.{t}.weight"] = merged(llm_state[f"layers.{i}.{t}.weight"], ab)
    pout = None
    if proj_state is not None:
        pout = dict(proj_state)
        for t in ("readout.0", "readout.2"):
            ab = _lora_lookup(lora_state, f"mm_projector.{t}")
            if ab is not None:
                pout[t + ".weight"] = merged(proj_state[t + ".weight"], ab)
    return out, pout


def random_lora_state(cfg: LlmConfig, seed: int = 7, r: int = 8, proj_dims=None, device="cpu", dtype=torch.float32,
                      std: float = 0.05):
    """test adapter in peft's key layout (both A and B random, so the branch is not a no-op)"""
    g = torch.Generator(device=device).manual_seed(seed)
    rnd = lambda *s: (torch.randn(*s, generator=g, device=device, dtype=torch.float32) * std).to(dtype)
    H, I, hd = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
    kvd = cfg.num_key_value_heads * hd
    dims = {"self_attn.q_proj": (H, H), "self_attn.k_proj": (kvd, H), "self_attn.v_proj": (kvd, H),
            "self_attn.o_proj": (H, H), "mlp.gate_proj": (I, H), "mlp.up_proj": (I, H), "mlp.down_proj": (H, I)}
    st = {}
    for i in range(cfg.num_hidden_layers):
        for t, (o, n) in dims.items():
            k = f"base_model.model.model.layers.{i}.{t}"
            st[k + ".lora_A.weight"], st[k + ".lora_B.weight"] = rnd(r, n), rnd(o, r)
    if proj_dims is not None:
        din, dout = proj_dims
        for t, (o, n) in (("readout.0", (dout, din)), ("readout.2", (dout, dout))):
            k = f"base_model.model.model.mm_projector.{t}"
            st[k + ".lora_A.weight"], st[k + ".lora_B.weight"] = rnd(r, n), rnd(o, r)
    return st
