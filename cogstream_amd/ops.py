"""Operator-level wrappers: torch CUDA tensors in, HIP kernels through the C ABI, torch tensors out.
torch is used for device memory and streams only; no arithmetic happens here."""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import _lib as L
from ._lib import check, current_stream, dtype_code, ptr


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.CogsError("cogstream_amd ops need CUDA (ROCm) tensors; there is no CPU path")


def pad_cols(x: torch.Tensor, mult: int) -> torch.Tensor:
    """zero-pad the last dim of a 2-D tensor to a multiple of `mult` (host-side weight packing helper)"""
    k = x.shape[-1]
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return x.contiguous()
    out = x.new_zeros(*x.shape[:-1], kp)
    out[..., :k] = x
    return out


def k_slab(dtype) -> int:
    return 64 if dtype == torch.bfloat16 else 32


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, act: int = L.ACT_NONE,
         residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, out_f32: bool = False,
         rope_cos: Optional[torch.Tensor] = None, rope_sin: Optional[torch.Tensor] = None, rope_cols: int = 0,
         head_dim: int = 0, rope_lut: Optional[torch.Tensor] = None, rope_rowpos: Optional[torch.Tensor] = None,
         row_stats: Optional[torch.Tensor] = None, ln_ab: Optional[torch.Tensor] = None,
         col_c: Optional[torch.Tensor] = None, hm_cols: int = 0, lib=None) -> torch.Tensor:
    """out[M,N] = epilogue(a[M,K] @ w[N,K]^T)  (see cogs_gemm in include/cogs.h). hm_cols > 0: head-major output -- the
    result tensor is [N / hm_cols, hm_cols / head_dim, M, head_dim] (cogs_gemm_desc.hm_rows = M). lib: another build of
    the library (A/B tests only)"""
    _need_cuda(a, w, bias, residual, out)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.stride(1) == 1 and w.stride(1) == 1
    ncols = N // 2 if act == L.ACT_SWIGLU else N
    if out is None:
        if hm_cols > 0:
            out = torch.empty(N // hm_cols, hm_cols // head_dim, M, head_dim, device=a.device, dtype=a.dtype)
        else:
            out = torch.empty(M, ncols, device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    d = L.GemmDesc()
    d.dtype = dtype_code(a.dtype)
    d.A, d.lda = ptr(a), a.stride(0)
    d.W, d.ldw = ptr(w), w.stride(0)
    d.C, d.ldc = ptr(out), (N if hm_cols > 0 else out.stride(0))
    d.hm_rows, d.hm_cols = (M if hm_cols > 0 else 0), hm_cols
    d.bias = ptr(bias)
    d.residual, d.ldr = ptr(residual), (residual.stride(0) if residual is not None else 0)
    d.M, d.N, d.K = M, N, K
    d.act, d.out_f32 = act, int(out_f32)
    d.rope_cos, d.rope_sin, d.rope_cols, d.head_dim = ptr(rope_cos), ptr(rope_sin), rope_cols, head_dim
    d.rope_lut, d.rope_rowpos = ptr(rope_lut), ptr(rope_rowpos)
    d.rope_maxpos = int(rope_lut.shape[0]) if rope_lut is not None else 0
    _need_cuda(row_stats, ln_ab, col_c)
    d.row_stats, d.ln_ab, d.col_c = ptr(row_stats), ptr(ln_ab), ptr(col_c)
    fn = L.lib.cogs_gemm
    if lib is not None:
        fn = lib.cogs_gemm
        fn.restype, fn.argtypes = L.SIGNATURES["cogs_gemm"]
    check(fn(current_stream(), C.byref(d)), "cogs_gemm")
    return out


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, hq: int, hkv: int, head_dim: int,
              scale: Optional[float] = None, cu_seqlens: Optional[torch.Tensor] = None, max_seqlen: int = 0,
              row_lo: Optional[torch.Tensor] = None, row_hi: Optional[torch.Tensor] = None, bias: float = 0.0,
              causal: bool = False, q_pos0: int = 0, force_rowwise: bool = False, nsplit: int = 1,
              out: Optional[torch.Tensor] = None, q_prescaled: bool = False, head_major: bool = False, lib=None) -> torch.Tensor:
    """token-major attention: q [Lq, hq*hd] (may be a column view of a fused buffer), k/v [Lk, hkv*hd]. head_major: q, k, v
    are [heads, L, hd] instead (cogs_attn_desc.head_stride; the output stays token-major). lib: another build of the
    library (tests: A/B identity)"""
    _need_cuda(q, k, v)
    if head_major:       # [heads, L, hd] with contiguous (L, hd) blocks; the head stride may exceed L * hd (a view of a larger buffer)
        assert q.dim() == 3 and all(t.stride(2) == 1 and (t.shape[1] == 1 or t.stride(1) == head_dim) for t in (q, k, v))
        Lq, Lk = q.shape[1], k.shape[1]
    else:
        Lq, Lk = q.shape[0], k.shape[0]
    if out is None:
        out = torch.empty(Lq, hq * head_dim, device=q.device, dtype=q.dtype)
    d = L.AttnDesc()
    d.dtype = dtype_code(q.dtype)
    d.Q, d.K, d.V, d.O = ptr(q), ptr(k), ptr(v), ptr(out)
    if head_major:      # (strides of size-1 dimensions are not trusted: one head -> the dense stride)
        assert Lq == Lk and q.shape == k.shape == v.shape, "one head stride for q, k, v"
        hs = [t.stride(0) if t.shape[0] > 1 else Lq * head_dim for t in (q, k, v)]
        assert hs[0] == hs[1] == hs[2] >= Lq * head_dim, "one head stride for q, k, v"
        d.ldq = d.ldk = d.ldv = head_dim
        d.ldo = out.stride(0)
        d.head_stride = hs[0]
    else:
        d.ldq, d.ldk, d.ldv, d.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
        d.head_stride = 0
    d.cu_seqlens = ptr(cu_seqlens)
    d.nseg = (cu_seqlens.numel() - 1) if cu_seqlens is not None else 1
    d.max_seqlen = max_seqlen
    d.row_lo, d.row_hi, d.bias = ptr(row_lo), ptr(row_hi), bias
    d.q_len, d.kv_len = Lq, Lk
    d.hq, d.hkv, d.head_dim = hq, hkv, head_dim
    d.scale = scale if scale is not None else 1.0 / math.sqrt(head_dim)
    d.causal, d.q_pos0 = int(causal), q_pos0
    d.force_rowwise = int(force_rowwise)
    d.q_prescaled = int(q_prescaled)
    ws = None
    if nsplit > 1:
        ws = torch.empty(nsplit * Lq * hq * (head_dim + 2), device=q.device, dtype=torch.float32)
        d.nsplit, d.ws, d.ws_bytes = nsplit, ptr(ws), ws.numel() * 4
    else:
        d.nsplit = 1
    fn = L.lib.cogs_attention
    if lib is not None:
        fn = lib.cogs_attention
        fn.restype, fn.argtypes = L.SIGNATURES["cogs_attention"]
    check(fn(current_stream(), C.byref(d)), "cogs_attention")
    return out


def layernorm(x, gamma, beta, eps: float = 1e-6):
    _need_cuda(x, gamma, beta)
    y = torch.empty_like(x)
    check(L.lib.cogs_layernorm(current_stream(), dtype_code(x.dtype), ptr(x), ptr(y), ptr(gamma), ptr(beta),
                               x.shape[0], x.shape[1], eps), "cogs_layernorm")
    return y


def rmsnorm(x, gamma, eps: float = 1e-6):
    _need_cuda(x, gamma)
    y = torch.empty_like(x)
    check(L.lib.cogs_rmsnorm(current_stream(), dtype_code(x.dtype), ptr(x), ptr(y), ptr(gamma), x.shape[0],
                             x.shape[1], eps), "cogs_rmsnorm")
    return y


def ln_merge(x, gamma, beta, group: int, eps: float = 1e-6):
    _need_cuda(x, gamma, beta)
    y = torch.empty(x.shape[0] // group, x.shape[1], device=x.device, dtype=x.dtype)
    check(L.lib.cogs_ln_merge(current_stream(), dtype_code(x.dtype), ptr(x), ptr(y), ptr(gamma), ptr(beta),
                              y.shape[0], group, x.shape[1], eps), "cogs_ln_merge")
    return y


def pixdiff_mask(pix: torch.Tensor, t: int, P: int, thr: float = 0.1, min_tokens: int = 1,
                 minor: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pix: one video's rows [t*P*merge^2, 588] -> uint8 keep-mask [t*P]"""
    _need_cuda(pix, minor)
    assert pix.is_contiguous()
    E = pix.numel() // (t * P)
    mask = torch.empty(t * P, device=pix.device, dtype=torch.uint8)
    check(L.lib.cogs_pixdiff_mask(current_stream(), dtype_code(pix.dtype), ptr(pix), t, P, E, thr, min_tokens,
                                  ptr(minor), ptr(mask)), "cogs_pixdiff_mask")
    return mask


def frame_mean_to_slot0(feats: torch.Tensor, P: int, frames: torch.Tensor) -> None:
    _need_cuda(feats, frames)
    assert feats.is_contiguous() and frames.dtype == torch.int32
    check(L.lib.cogs_frame_mean_to_slot0(current_stream(), dtype_code(feats.dtype), ptr(feats), P, feats.shape[1],
                                         ptr(frames), frames.numel()), "cogs_frame_mean_to_slot0")


def gather_rows(table_a: torch.Tensor, table_b: Optional[torch.Tensor], idx: torch.Tensor,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_cuda(table_a, table_b, idx, out)
    assert idx.dtype == torch.int64 and table_a.is_contiguous()
    if out is None:
        out = torch.empty(idx.numel(), table_a.shape[1], device=table_a.device, dtype=table_a.dtype)
    assert out.is_contiguous() and out.shape == (idx.numel(), table_a.shape[1]) and out.dtype == table_a.dtype
    check(L.lib.cogs_gather_rows(current_stream(), dtype_code(table_a.dtype), ptr(table_a), ptr(table_b), ptr(idx),
                                 ptr(out), idx.numel(), table_a.shape[1]), "cogs_gather_rows")
    return out


def mean_rows(x: torch.Tensor) -> torch.Tensor:
    _need_cuda(x)
    out = torch.empty(x.shape[1], device=x.device, dtype=torch.float32)
    check(L.lib.cogs_mean_rows(current_stream(), dtype_code(x.dtype), ptr(x), x.stride(0), x.shape[0], x.shape[1],
                               ptr(out)), "cogs_mean_rows")
    return out


def cosine(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _need_cuda(a, b)
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and b.is_contiguous()
    out = torch.empty(b.shape[0], device=a.device, dtype=torch.float32)
    check(L.lib.cogs_cosine(current_stream(), ptr(a), ptr(b), b.shape[0], b.shape[1], ptr(out)), "cogs_cosine")
    return out


def pack_rows(x: torch.Tensor, out_dtype, cols_out: int) -> torch.Tensor:
    _need_cuda(x)
    out = torch.empty(x.shape[0], cols_out, device=x.device, dtype=out_dtype)
    check(L.lib.cogs_pack_rows(current_stream(), dtype_code(x.dtype), dtype_code(out_dtype), ptr(x), x.stride(0),
                               ptr(out), cols_out, x.shape[0], x.shape[1], cols_out), "cogs_pack_rows")
    return out


def kmeans_workspace(T: int, PD: int, K: int, device) -> torch.Tensor:
    n = C.c_size_t()
    check(L.lib.cogs_kmeans_workspace_bytes(T, PD, K, C.byref(n)), "cogs_kmeans_workspace_bytes")
    return torch.empty(n.value, device=device, dtype=torch.uint8)


def kmeans_sqdist(feats, centres, centre_rows, K: int, ws) -> torch.Tensor:
    T, PD = feats.shape
    dist2 = torch.empty(T, K, device=feats.device, dtype=torch.float32)
    check(L.lib.cogs_kmeans_sqdist(current_stream(), dtype_code(feats.dtype), ptr(feats), T, PD, ptr(centres),
                                   ptr(centre_rows), K, ptr(dist2), ptr(ws), ws.numel()), "cogs_kmeans_sqdist")
    return dist2


def kmeans_assign(dist2, ts, centre_ts, alpha: float):
    T, K = dist2.shape
    assign = torch.empty(T, device=dist2.device, dtype=torch.int64)
    counts = torch.empty(K, device=dist2.device, dtype=torch.int32)
    check(L.lib.cogs_kmeans_assign(current_stream(), ptr(dist2), ptr(ts), ptr(centre_ts), T, K, alpha, ptr(assign),
                                   ptr(counts)), "cogs_kmeans_assign")
    return assign, counts


def kmeans_update(feats, ts, assign, reseed_rows, centres, centre_ts, ws) -> torch.Tensor:
    T, PD = feats.shape
    K = centres.shape[0]
    shift = torch.empty(1, device=feats.device, dtype=torch.float32)
    check(L.lib.cogs_kmeans_update(current_stream(), dtype_code(feats.dtype), ptr(feats), ptr(ts), T, PD, K,
                                   ptr(assign), ptr(reseed_rows), ptr(centres), ptr(centre_ts), ptr(shift), ptr(ws),
                                   ws.numel()), "cogs_kmeans_update")
    return shift


def select_near_centroid(dist2: torch.Tensor, assign: torch.Tensor, n: int):
    """cogs_select_near_centroid -> (picks int64 [K, n] (-1 padded), counts int32 [K]) on the device"""
    _need_cuda(dist2, assign)
    T, K = dist2.shape
    assert assign.dtype == torch.int64 and assign.numel() == T and dist2.dtype == torch.float32 and dist2.is_contiguous()
    picks = torch.empty(K, n, device=dist2.device, dtype=torch.int64)
    counts = torch.empty(K, device=dist2.device, dtype=torch.int32)
    check(L.lib.cogs_select_near_centroid(current_stream(), ptr(dist2), ptr(assign), T, K, int(n), ptr(picks), ptr(counts)),
          "cogs_select_near_centroid")
    return picks, counts


def kmeans_pp_step(feats, row: int, first: bool, nearest2: torch.Tensor, probs_host: Optional[torch.Tensor], ws) -> None:
    """cogs_kmeans_pp_step: nearest2 (device fp32 [T]) <- min(nearest2, |x - x[row]|^2); with probs_host (pinned CPU fp32
    [T]) the result is copied there and the stream synchronised"""
    T, PD = feats.shape
    assert nearest2.is_cuda and nearest2.dtype == torch.float32 and nearest2.numel() == T
    if probs_host is not None:
        assert (not probs_host.is_cuda) and probs_host.is_pinned() and probs_host.dtype == torch.float32 and probs_host.numel() == T
    check(L.lib.cogs_kmeans_pp_step(current_stream(), dtype_code(feats.dtype), ptr(feats), T, PD, int(row), int(bool(first)),
                                    ptr(nearest2), ptr(probs_host), ptr(ws), ws.numel()), "cogs_kmeans_pp_step")


def kmeans_pp(feats, first_row: int, K: int, q_draws: torch.Tensor, ws):
    """cogs_kmeans_pp -> (centre rows int32 [K] on the device, zero-sum flag int32 [1] on the device). q_draws: device
    fp32 [K-1, T] Exponential(1) draws of the CPU generator"""
    T, PD = feats.shape
    assert q_draws.is_cuda and q_draws.dtype == torch.float32 and q_draws.shape == (K - 1, T) and q_draws.is_contiguous()
    buf = torch.empty(K + 1, device=feats.device, dtype=torch.int32)       # [K] centre rows + the flag, one allocation
    idx, flag = buf[:K], buf[K:]
    nearest2 = torch.empty(T, device=feats.device, dtype=torch.float32)
    check(L.lib.cogs_kmeans_pp(current_stream(), dtype_code(feats.dtype), ptr(feats), T, PD, int(K), int(first_row), ptr(q_draws),
                               ptr(idx), ptr(flag), ptr(nearest2), ptr(ws), ws.numel()), "cogs_kmeans_pp")
    return idx, flag


def kmeans_margins(T: int, PD: int, K: int, ws):
    """cogs_kmeans_margins -> (min relative margin of the final distances, (row, iteration) pairs below 1e-3) of the last
    kmeans_lloyd call on `ws`"""
    m, n = C.c_float(0), C.c_int32(0)
    check(L.lib.cogs_kmeans_margins(current_stream(), int(T), int(PD), int(K), ptr(ws), ws.numel(), C.byref(m), C.byref(n)),
          "cogs_kmeans_margins")
    return float(m.value), int(n.value)


def kmeans_lloyd(feats, ts, centres, centre_ts, assign, alpha: float, max_iter: int, tol: float, pool, ws):
    """cogs_kmeans_lloyd -> (iterations completed, reseeds used, pool exhausted). pool: python list of pre-drawn rows"""
    T, PD = feats.shape
    K = centres.shape[0]
    arr = (C.c_int32 * max(len(pool), 1))(*pool)
    it, used, ex = C.c_int(0), C.c_int(0), C.c_int(0)
    check(L.lib.cogs_kmeans_lloyd(current_stream(), dtype_code(feats.dtype), ptr(feats), ptr(ts), T, PD, K, float(alpha),
                                  int(max_iter), float(tol), C.cast(arr, C.c_void_p), len(pool), ptr(centres), ptr(centre_ts),
                                  ptr(assign), C.byref(it), C.byref(used), C.byref(ex), ptr(ws), ws.numel()),
          "cogs_kmeans_lloyd")
    return it.value, used.value, bool(ex.value)


def argmax(logits: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out: an int64 [1] device tensor (e.g. a slot of the generated-ids buffer) to write the index into"""
    _need_cuda(logits, out)
    if out is None:
        out = torch.empty(1, device=logits.device, dtype=torch.int64)
    assert out.dtype == torch.int64 and out.numel() == 1 and out.is_contiguous()
    ws = torch.empty(128, device=logits.device, dtype=torch.float32)
    check(L.lib.cogs_argmax(current_stream(), ptr(logits), logits.numel(), ptr(out), ptr(ws)), "cogs_argmax")
    return out


def logits_process(logits: torch.Tensor, prev: Optional[torch.Tensor], repetition_penalty: float = 1.0,
                   allowed: Optional[torch.Tensor] = None, temperature: float = 1.0) -> None:
    """in place on a fp32 [vocab] row"""
    _need_cuda(logits, prev, allowed)
    n_prev = prev.numel() if prev is not None else 0
    tmp = torch.empty(max(n_prev, 1), device=logits.device, dtype=torch.float32)
    check(L.lib.cogs_logits_process(current_stream(), ptr(logits), logits.numel(), ptr(prev), n_prev,
                                    repetition_penalty, ptr(allowed), allowed.numel() if allowed is not None else 0,
                                    temperature, ptr(tmp)), "cogs_logits_process")


def topk(logits: torch.Tensor, k: int):
    _need_cuda(logits)
    val = torch.empty(k, device=logits.device, dtype=torch.float32)
    idx = torch.empty(k, device=logits.device, dtype=torch.int32)
    ws = torch.empty(logits.numel(), device=logits.device, dtype=torch.float32)
    check(L.lib.cogs_topk(current_stream(), ptr(logits), logits.numel(), k, ptr(val), ptr(idx), ptr(ws)), "cogs_topk")
    return val, idx


def sample(logits: torch.Tensor, top_k: int = 0, top_p: float = 1.0, draws: Optional[torch.Tensor] = None,
           seed: int = 0, offset: int = 0, out: Optional[torch.Tensor] = None, want_kept: bool = False,
           temperature: float = 1.0):
    """cogs_sample on a processed fp32 [vocab] row -> int64 [1] token on the device (+ (ids, probs) of the surviving
    tokens if want_kept). draws: device float [vocab] of Exponential(1) draws (parity with torch.multinomial on the
    CPU generator) or None (Philox on the device, keyed by seed/offset)."""
    _need_cuda(logits, draws)
    n = logits.numel()
    if out is None:
        out = torch.empty(1, device=logits.device, dtype=torch.int64)
    ws = torch.empty(L.lib.cogs_sample_workspace_bytes(), device=logits.device, dtype=torch.uint8)
    kept_idx = kept_prob = n_kept = None
    cap = 0
    if want_kept:
        cap = n
        kept_idx = torch.empty(cap, device=logits.device, dtype=torch.int32)
        kept_prob = torch.empty(cap, device=logits.device, dtype=torch.float32)
        n_kept = torch.zeros(1, device=logits.device, dtype=torch.int32)
    check(L.lib.cogs_sample(current_stream(), ptr(logits), n, float(temperature), int(top_k or 0),
                            float(top_p if top_p is not None else 1.0),
                            ptr(draws), int(seed) & (2 ** 64 - 1), int(offset), ptr(out), ptr(kept_idx), ptr(kept_prob),
                            ptr(n_kept), cap, ptr(ws)), "cogs_sample")
    if want_kept:
        m = int(n_kept.item())
        return out, kept_idx[:m], kept_prob[:m]
    return out


def ln_finalize(row_stats: torch.Tensor, H: int, eps: float) -> torch.Tensor:
    """row_stats [rows, H/64, 2] fp32 (written by gemm(..., row_stats=)) -> ln_ab [rows, 2] = (rstd, -rstd * mean)"""
    _need_cuda(row_stats)
    rows = row_stats.shape[0]
    ab = torch.empty(rows, 2, device=row_stats.device, dtype=torch.float32)
    check(L.lib.cogs_ln_finalize(current_stream(), ptr(row_stats), rows, H, eps, ptr(ab)), "cogs_ln_finalize")
    return ab
