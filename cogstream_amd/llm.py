"""Host mirror of the Qwen2 seams the reference calls (transformers, pinned 4.46.3):
self.get_model()(inputs_embeds=..., attention_mask=...) (model/cogreasoner_chat.py:312-316,322) and
super().generate(inputs_embeds=...) (:802-807), computing through cogs_llm_forward."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L
from . import ops
from .runtime import get_handle
from .weights import LlmConfig, PackedLlm


class KVCache:
    def __init__(self, cfg: LlmConfig, max_len: int, dtype, device):
        kvd = cfg.num_key_value_heads * cfg.head_dim
        self.k = torch.empty(cfg.num_hidden_layers, max_len, kvd, device=device, dtype=dtype)
        self.v = torch.empty_like(self.k)
        self.struct = L.KV(self.k.data_ptr(), self.v.data_ptr(), max_len, 0)

    @property
    def len(self) -> int:
        return self.struct.len

    def reset(self, length: int = 0):
        self.struct.len = length


class PrefixKV:
    """KV rows of the previous prompt, kept so that the next generate() only prefills what differs (SURVEY.md §8f
    rank 3; the reference rebuilds the whole conversation every turn, evaluate/answer_generate.py:130-148).
    The reusable prefix is found by comparing the new input embeddings with the previous ones bit for bit on the
    device: KV row i depends only on rows 0..i and the position i, so every invalidation rule (resized frames, a
    history turn dropped by the selection stage, another question) is covered by that comparison. Only rows that
    were PREFILLED are reused -- rows written by decode steps went through the GEMV kernels."""

    def __init__(self, engine: "Qwen2Engine"):
        self.engine = engine
        self.cache: Optional[KVCache] = None
        self.rows: Optional[torch.Tensor] = None   # [n, H] embeddings whose KV rows 0..n-1 are in `cache`
        self.reused = 0    # rows taken from the cache, summed over calls
        self.seen = 0      # prompt rows, summed over calls

    def plan(self, embeds: torch.Tensor, need: int) -> int:
        """-> number of leading rows of `embeds` whose KV is already in self.cache (at most len(embeds) - 2)"""
        if self.cache is None or self.cache.k.shape[1] < need:
            old = self.cache
            self.cache = self.engine.new_cache(need + need // 2)   # sessions grow: leave room for the next turns
            if old is not None and self.rows is not None:
                n = self.rows.shape[0]
                self.cache.k[:, :n].copy_(old.k[:, :n])
                self.cache.v[:, :n].copy_(old.v[:, :n])
            else:
                self.rows = None
        p = 0
        if self.rows is not None and self.rows.dtype == embeds.dtype:
            # at least 2 rows are always prefilled: a 1-row forward is the decode GEMV, whose rounding differs
            n = min(self.rows.shape[0], embeds.shape[0] - 2)
            if n > 0:
                bits = torch.int16 if embeds.element_size() == 2 else torch.int32
                neq = (self.rows[:n].view(bits) != embeds[:n].view(bits)).any(dim=1)
                idx = torch.arange(n, device=embeds.device)
                p = int(torch.where(neq, idx, torch.full_like(idx, n)).min().item())
        self.seen += embeds.shape[0]
        self.reused += p
        return p

    def drop(self):
        self.rows = None


class Qwen2Engine:
    def __init__(self, state: Dict[str, torch.Tensor], cfg: LlmConfig, dtype=torch.bfloat16, device="cuda"):
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.handle = get_handle(self.device)
        self.packed = PackedLlm(state, cfg, dtype, self.device)
        self._activate()

    def _activate(self):
        self.handle.activate("llm", self, lambda: L.check(
            L.lib.cogs_llm_load(self.handle.h, C.byref(self.packed.struct)), "cogs_llm_load"))

    def new_cache(self, max_len: int) -> KVCache:
        return KVCache(self.cfg, max_len, self.dtype, self.device)

    def embed_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        return ops.gather_rows(self.packed.embed, None, ids.reshape(-1).to(self.device, torch.int64))

    def forward(self, embeds: torch.Tensor, cache: Optional[KVCache] = None, *, want_logits: bool = True,
                want_pooled: bool = False, want_hidden: bool = False):
        """returns dict(logits fp32 [vocab] | pooled fp32 [H] | hidden [S,H])"""
        self._activate()
        embeds = embeds.contiguous()
        S = embeds.shape[0]
        ctx = S + (cache.len if cache is not None else 0)
        nbytes = C.c_size_t()
        L.check(L.lib.cogs_llm_workspace_bytes(self.handle.h, S, ctx, C.byref(nbytes)), "cogs_llm_workspace_bytes")
        ws = self.handle.workspace("llm", nbytes.value)
        out = {}
        logits = torch.empty(self.cfg.vocab_size, device=self.device, dtype=torch.float32) if want_logits else None
        pooled = torch.empty(self.cfg.hidden_size, device=self.device, dtype=torch.float32) if want_pooled else None
        hidden = torch.empty(S, self.cfg.hidden_size, device=self.device, dtype=self.dtype) if want_hidden else None
        L.check(L.lib.cogs_llm_forward(self.handle.h, L.current_stream(), embeds.data_ptr(), S,
                                       C.byref(cache.struct) if cache is not None else None, L.ptr(logits),
                                       L.ptr(pooled), L.ptr(hidden), ws.data_ptr(), ws.numel()), "cogs_llm_forward")
        if want_logits:
            out["logits"] = logits
        if want_pooled:
            out["pooled"] = pooled
        if want_hidden:
            out["hidden"] = hidden
        return out

    def forward_segments(self, embeds: torch.Tensor, seg_lens: Sequence[int]) -> torch.Tensor:
        """several independent sequences stored back to back -> mean of the final hidden states of each, fp32
        [nseg, H] (cogs_llm_forward_segments): one var-len prefill instead of one forward per sequence"""
        self._activate()
        embeds = embeds.contiguous()
        S = embeds.shape[0]
        cu = [0]
        for n in seg_lens:
            cu.append(cu[-1] + int(n))
        assert cu[-1] == S and all(int(n) > 0 for n in seg_lens)
        nbytes = C.c_size_t()
        L.check(L.lib.cogs_llm_workspace_bytes(self.handle.h, S, S, C.byref(nbytes)), "cogs_llm_workspace_bytes")
        ws = self.handle.workspace("llm", nbytes.value)
        out = torch.empty(len(seg_lens), self.cfg.hidden_size, device=self.device, dtype=torch.float32)
        cua = (C.c_int32 * len(cu))(*cu)
        L.check(L.lib.cogs_llm_forward_segments(self.handle.h, L.current_stream(), embeds.data_ptr(), S, cua, len(seg_lens),
                                                out.data_ptr(), ws.data_ptr(), ws.numel()), "cogs_llm_forward_segments")
        return out

    def generate(self, embeds: torch.Tensor, *, max_new_tokens: int, eos_token_id: Sequence[int] = (),
                 do_sample: bool = False, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0,
                 repetition_penalty: float = 1.0, allowed_ids: Optional[Sequence[int]] = None,
                 prompt_ids: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None,
                 sampler: str = "device", seed: Optional[int] = None,
                 cache: Optional[KVCache] = None, ignore_eos: bool = False,
                 prefix: Optional[PrefixKV] = None, prefilled: Optional[dict] = None,
                 stage_times: Optional[dict] = None) -> List[int]:
        """GenerationMixin.generate with inputs_embeds: prefill, then one cogs_llm_forward per token.
        Returns the NEW token ids only. Logits processors run in HF order:
        repetition penalty -> custom (allowed-id mask) -> temperature -> top-k -> top-p.
        `prefix`: reuse the KV rows of the leading embeddings that are unchanged since the previous call.
        `prefilled`: the result of a forward(embeds, cache) the caller has already run into `cache` (bench.py times the
        prefill and the token loop separately this way); the prompt is then not prefilled again.
        Sampling (do_sample): cogs_sample on the device, tokens never visit the host. sampler="device": Philox draws
        keyed by `seed` (default: one draw of the CPU generator) and the step; sampler="host": the CPU generator
        (`generator`, default the global one) makes the [vocab] exponential draws of every step exactly as
        torch.multinomial does in the reference's CPU run, so the sampled ids are the reference's."""
        S = embeds.shape[0]
        if stage_times is not None:        # bench.py's pipeline split: the prompt pass and the token loop, drained on both sides
            import time
            torch.cuda.synchronize()
            t_start = time.perf_counter()
        if prefilled is not None:
            assert cache is not None and prefix is None and cache.len >= S and "logits" in prefilled
            pos_start = cache.len
            res = prefilled
        elif prefix is not None:
            assert cache is None
            embeds = embeds.contiguous()
            p = prefix.plan(embeds, S + max_new_tokens)
            cache = prefix.cache
            cache.reset(p)
            prefix.rows = None          # not valid while this call is rewriting the cache
            res = self.forward(embeds[p:], cache)
            prefix.rows = embeds        # rows 0..S-1 are prefilled; decode rows behind them are never reused
            pos_start = S
        else:
            if cache is None:
                cache = self.new_cache(S + max_new_tokens)
            pos_start = cache.len + S
            res = self.forward(embeds, cache)
        if stage_times is not None:
            torch.cuda.synchronize()
            t_loop = time.perf_counter()
            stage_times["answer_prefill"] = stage_times.get("answer_prefill", 0.0) + t_loop - t_start
        allowed = (torch.tensor(list(allowed_ids), dtype=torch.int32, device=self.device) if allowed_ids is not None else None)
        eos = set(int(e) for e in eos_token_id)
        # token ids stay on the device: the greedy id feeds the next embedding gather and the repetition-penalty
        # history without a host round trip; the host only looks at them every `check_every` steps (EOS test).
        n_prompt = int(prompt_ids.numel()) if prompt_ids is not None else 0
        seen = torch.empty(n_prompt + max_new_tokens, dtype=torch.int64, device=self.device)
        if n_prompt:
            seen[:n_prompt] = prompt_ids.reshape(-1).to(self.device)
        toks = seen[n_prompt:]        # the generated ids ARE the tail of the penalty history: argmax / the sampler write them once
        n_seen, produced = n_prompt, 0
        # host-drawn sampling (parity mode) consumes one [vocab] draw of the CPU generator per step: steps issued past
        # the EOS would advance that generator beyond where GenerationMixin stops, so the stop test runs every step
        check_every = 1 if (do_sample and sampler == "host") else 8
        if do_sample and sampler == "device" and seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), generator=generator).item())
        need_proc = repetition_penalty != 1.0 or allowed is not None     # the temperature is applied inside cogs_sample
        stop_at = None
        for step in range(max_new_tokens):
            logits = res["logits"]
            if need_proc:
                prev = seen[:n_seen] if (n_seen and repetition_penalty != 1.0) else None
                ops.logits_process(logits, prev, repetition_penalty, allowed, 1.0)
            if do_sample:
                tok_dev = ops.sample(logits, top_k or 0, 1.0 if top_p is None else top_p,
                                     draws=self._draws(generator) if sampler == "host" else None,
                                     seed=seed or 0, offset=step, temperature=temperature, out=toks[step:step + 1])
            else:
                tok_dev = ops.argmax(logits, out=toks[step:step + 1])      # straight into the generated-ids buffer
            n_seen += 1
            produced += 1
            last = step + 1 == max_new_tokens
            if eos and not ignore_eos and ((step + 1) % check_every == 0 or last):
                host = toks[:produced].tolist()
                hit = next((i for i, t in enumerate(host) if t in eos), None)
                if hit is not None:
                    stop_at = hit + 1
                    break
            if not last:
                res = self.forward(self.embed_tokens(tok_dev), cache)
        out = toks[:produced].tolist()
        if stage_times is not None:
            stage_times["decode"] = stage_times.get("decode", 0.0) + time.perf_counter() - t_loop
        if stop_at is not None:
            out = out[:stop_at]
            # forwards issued past the EOS only wrote KV rows that are dropped again here
            cache.reset(min(cache.len, pos_start + stop_at - 1))
        return out

    def _draws(self, generator) -> torch.Tensor:
        """parity mode: the [vocab] Exponential(1) draws torch.multinomial(probs, 1) makes on the CPU generator
        (GenerationMixin._sample, transformers 4.46.3; the reference's CPU path draws from the global generator)"""
        q = torch.empty(self.cfg.vocab_size, dtype=torch.float32).exponential_(1, generator=generator)
        return q.to(self.device, non_blocking=True)
