// Fused epilogue shared by the MFMA GEMM (gemm.hip) and the weight-streaming GEMV (gemv.hip).
// A caller hands over 4 CONSECUTIVE output columns n..n+3 of row m as fp32 accumulators.
//
//   bias      + b[n..n+3]
//   rope      columns < rope_cols: weight rows were packed so that (n, n+1) is the rotary pair
//             (d, d+hd/2) of rotate_half (model/modeling_videollama3_encoder.py:154-170; Qwen2
//             apply_rotary_pos_emb); cos/sin tables are [M, hd/2] fp32
//   act       gelu_pytorch_tanh (ViT fc1), erf GELU (projector), SwiGLU on (gate_i, up_i)-interleaved
//             rows (Qwen2 MLP) -> two outputs at column n/2
//   residual  + R[m][n..n+3]
//   store     T or fp32 (lm_head logits)
#pragma once
#include "common.h"
#include "kernels.h"

struct EpiArgs {
    char* C; long ldc;          // elements per row
    const void* bias;
    const char* R; long ldr;    // elements per row
    int act;
    int out_f32;
    const float* rope_cos;
    const float* rope_sin;
    int rope_pairs;
    int rope_cols;
    int head_dim;
};

template <typename T>
__device__ __forceinline__ void epilogue4(const EpiArgs& p, int m, int n, f32x4 v) {
    if (p.bias) v += ld4_f<T>(reinterpret_cast<const T*>(p.bias) + n);
    if (p.rope_cos && n < p.rope_cols) {
        const int pi = (n % p.head_dim) >> 1;
        const f32x2 c = *reinterpret_cast<const f32x2*>(p.rope_cos + (long)m * p.rope_pairs + pi);
        const f32x2 s = *reinterpret_cast<const f32x2*>(p.rope_sin + (long)m * p.rope_pairs + pi);
        f32x4 r;
        r[0] = v[0] * c[0] - v[1] * s[0];
        r[1] = v[1] * c[0] + v[0] * s[0];
        r[2] = v[2] * c[1] - v[3] * s[1];
        r[3] = v[3] * c[1] + v[2] * s[1];
        v = r;
    }
    if (p.act == COGS_ACT_GELU_TANH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(v[e]);
    } else if (p.act == COGS_ACT_GELU_ERF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
    } else if (p.act == COGS_ACT_SWIGLU) {
        const float o0 = silu_f(v[0]) * v[1];
        const float o1 = silu_f(v[2]) * v[3];
        T* cp = reinterpret_cast<T*>(p.C) + (long)m * p.ldc + (n >> 1);
        st_f<T>(cp, o0);
        st_f<T>(cp + 1, o1);
        return;
    }
    if (p.R) v += ld4_f<T>(reinterpret_cast<const T*>(p.R) + (long)m * p.ldr + n);
    if (p.out_f32)
        st4_f<float>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n, v);
    else
        st4_f<T>(reinterpret_cast<T*>(p.C) + (long)m * p.ldc + n, v);
}

inline int cogs_fill_epi(const CogsGemm& g, EpiArgs* e) {
    if (g.act == COGS_ACT_SWIGLU && (g.bias || g.residual || g.out_f32)) return COGS_E_INVALID;
    if (g.rope_cos && (g.head_dim <= 0 || g.head_dim % 4 != 0 || g.rope_cols % g.head_dim != 0)) return COGS_E_INVALID;
    e->C = (char*)g.C; e->ldc = g.ldc;
    e->bias = g.bias;
    e->R = (const char*)g.residual; e->ldr = g.ldr;
    e->act = g.act; e->out_f32 = g.out_f32;
    e->rope_cos = g.rope_cos; e->rope_sin = g.rope_sin;
    e->rope_pairs = g.head_dim / 2; e->rope_cols = g.rope_cols;
    e->head_dim = g.head_dim > 0 ? g.head_dim : 4;
    return COGS_OK;
}
