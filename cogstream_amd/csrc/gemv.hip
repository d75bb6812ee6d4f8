// Weight-streaming GEMV for single-token decode:  y[1,N] = epilogue( x[1,K] . W[N,K]^T ).
//
// Replaces the M = 1 case of the Qwen2 linears during generation (transformers GenerationMixin
// decode steps behind model/cogreasoner_chat.py:802-807). HBM-bound: every weight byte is read
// exactly once, straight into registers (no LDS round trip: nothing is shared between waves);
// each wave owns 4 (or 2) consecutive output rows (whole epilogue groups: bias / rope pair / swiglu
// pair / residual), lanes stride over K in 16-byte chunks, 8 independent 16-B loads in flight per lane
// per step; x is re-read from L2. fp32 accumulation, 64-lane butterfly per row.
#include "common.h"
#include "kernels.h"
#include "gemm_epilogue.h"
#include "debug.h"

// 16-byte chunks per row in flight per lane (R * U = 8 loads); swept with tools/micro/gemv_micro.cpp: (2 rows, U 4) and
// (4 rows, U 2) are the fastest or level on every decode shape (U 7 / 8 / 3 / 4: within +-4 %, each slower somewhere)
#ifndef GEMV_U_SMALL
#define GEMV_U_SMALL 4
#endif
#ifndef GEMV_U_BIG
#define GEMV_U_BIG 2
#endif

namespace {

struct GemvArgs {
    const char* x;
    const char* W; long ldw;  // bytes
    int N, K;
    const void* rms_gamma;    // != null: x is RMS-normalised on the fly (Qwen2RMSNorm fused into the GEMV)
    float rms_eps;
    char* kv_k; char* kv_v; int kv_col0, kv_dim;   // != null: columns >= kv_col0 are the new K / V cache row
    EpiArgs epi;
};

// Qwen2RMSNorm on 8 (4) elements: weight * (x * rstd).to(T), result rounded to T like the standalone kernel
template <typename T, int E>
__device__ __forceinline__ void rms_apply(float (&x)[E], const T* gamma, float rstd) {
    float g[E];
    if constexpr (sizeof(T) == 2) ld8_f<T>(gamma, reinterpret_cast<float(&)[8]>(g));
    else { f32x4 t0 = ld4_f<T>(gamma); for (int e = 0; e < 4; ++e) g[e] = t0[e]; }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        float n = x[e] * rstd;
        if (sizeof(T) == 2) n = bf2f(f2bf(n));
        n = g[e] * n;
        if (sizeof(T) == 2) n = bf2f(f2bf(n));
        x[e] = n;
    }
}

// the same on 8 values with gamma already in registers (bf16 path of the register-resident prologue below)
__device__ __forceinline__ void rms_apply_reg(float (&x)[8], const float (&g)[8], float rstd) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float n = x[e] * rstd;
        n = bf2f(f2bf(n));
        n = g[e] * n;
        x[e] = bf2f(f2bf(n));
    }
}

// acc + a.lo * b.lo + a.hi * b.hi on packed bf16 pairs (v_dot2c_f32_bf16)
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float acc) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a), __builtin_bit_cast(bf2_t, b), acc, false);
}

// epilogue of 2 consecutive output columns n, n+1 (one rotary / SwiGLU pair): same arithmetic as epilogue4
template <typename T>
__device__ __forceinline__ void epilogue2(const EpiArgs& p, int n, float v0, float v1) {
    const T* bias = reinterpret_cast<const T*>(p.bias);
    if (bias) {
        if constexpr (sizeof(T) == 2) { v0 += bf2f(bias[n]); v1 += bf2f(bias[n + 1]); }
        else { v0 += bias[n]; v1 += bias[n + 1]; }
    }
    if (p.rope_cos && n < p.rope_cols) {
        const int pi = (n % p.head_dim) >> 1;
        float c, sn;
        if (p.rope_sin) { c = p.rope_cos[pi]; sn = p.rope_sin[pi]; }
        else { c = p.rope_cos[2 * pi]; sn = p.rope_cos[2 * pi + 1]; }
        const float r0 = v0 * c - v1 * sn;
        const float r1 = v1 * c + v0 * sn;
        v0 = r0; v1 = r1;
    }
    if (n < p.q_cols) { v0 *= p.q_scale; v1 *= p.q_scale; }
    if (p.act == COGS_ACT_GELU_TANH) { v0 = gelu_tanh_f(v0); v1 = gelu_tanh_f(v1); }
    else if (p.act == COGS_ACT_GELU_ERF) { v0 = gelu_erf_f(v0); v1 = gelu_erf_f(v1); }
    else if (p.act == COGS_ACT_SWIGLU) {
        st_f<T>(reinterpret_cast<T*>(p.C) + (n >> 1), silu_f(v0) * v1);
        return;
    }
    if (p.R) {
        const T* r = reinterpret_cast<const T*>(p.R);
        if constexpr (sizeof(T) == 2) { v0 += bf2f(r[n]); v1 += bf2f(r[n + 1]); }
        else { v0 += r[n]; v1 += r[n + 1]; }
    }
    if (p.out_f32) { reinterpret_cast<float*>(p.C)[n] = v0; reinterpret_cast<float*>(p.C)[n + 1] = v1; }
    else { st_f<T>(reinterpret_cast<T*>(p.C) + n, v0); st_f<T>(reinterpret_cast<T*>(p.C) + n + 1, v1); }
}

// R output rows per wave (4: big N; 2: N so small that 4 rows per wave would leave CUs without enough bytes in
// flight -- the 3584-row o/down projections launched 224 workgroups on 256 CUs and ran at 2-5 TB/s), U chunks of
// 16 bytes per row in flight per lane (R*U = 8 independent loads, kept raw until used).
// DOT2 (round 5, bf16 only): the products run on v_dot2c_f32_bf16 -- two bf16 x bf16 products and an fp32 accumulate per instruction,
// straight from the packed registers -- instead of unpacking both operands (1 VALU each) for an fp32 fma per element: 8 VALU per
// 16-byte weight chunk instead of 16 + 8 (+ 8 for x). The small GEMVs are bound by exactly that: a wave of the qkv projection
// spent ~840 vector instructions on 14 KB of weights (56 normalised x values, unpacked and rounded per wave, times 2 rows).
// The normalised x is formed once per wave as packed bf16 pairs -- the same values as before, bit for bit; the sums differ from
// the fma chain's in the last fp32 bits only (products exact either way, two-term partial sums before the accumulate).
template <typename T, int R, int U, bool DOT2 = false>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs p) {
    constexpr int EPC = 16 / sizeof(T);  // elements per 16-byte chunk
    const int lane = threadIdx.x & 63;
    const int n = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (n >= p.N) return;
    const int nch = p.K / EPC;
    const T* x = reinterpret_cast<const T*>(p.x);
    const T* w[R];
#pragma unroll
    for (int r = 0; r < R; ++r) w[r] = reinterpret_cast<const T*>(p.W + (long)min(n + r, p.N - 1) * p.ldw);
    const T* gam = reinterpret_cast<const T*>(p.rms_gamma);
    auto to_f = [](const u32x4& raw, float (&o)[EPC]) {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { o[2 * i] = bf_lo(raw[i]); o[2 * i + 1] = bf_hi(raw[i]); }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = __uint_as_float(raw[i]);
        }
    };
    // Weights are read exactly once per token by exactly one CU: non-temporal loads keep them from displacing what IS
    // re-read (x, the KV cache, the next kernels' operands) in L2 / the Infinity Cache (A/B: build with -DCOGS_GEMV_NO_NT)
    auto ldw = [](const T* q) -> u32x4 {
#ifndef COGS_GEMV_NO_NT
        return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q));
#else
        return *reinterpret_cast<const u32x4*>(q);
#endif
    };
    // The first batch of weight chunks is requested BEFORE the fused RMSNorm statistics: the weights do not depend
    // on x, and the statistics (a dependent load -> reduce -> rsqrt chain, ~2 us) would otherwise sit in front of the
    // whole stream (the qkv GEMV took 14.8 us against 8.0 us for the o projection of nearly the same size).
    int ch = lane;
    u32x4 w_first[U][R];
    const bool have_first = ch + 64 * (U - 1) < nch;
    if (have_first) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) w_first[u][r] = ldw(w[r] + (ch + 64 * u) * EPC);
    }
    float rstd = 1.f;
    // Fused Qwen2RMSNorm with K = 3584 (every fused call of the model: qkv, gate/up, lm_head): x and gamma are 7 chunks
    // per lane, loaded ONCE into registers right behind the first weight batch and used for the statistics and for the
    // products (the generic path below reads x twice and gamma once more, from L2, inside the dependent chain: the
    // prologue cost 4.3 us of a 12.9 us qkv call). Same summation order as the generic path: bit-identical results.
    constexpr int XC = 7;
    if constexpr (sizeof(T) == 2) {
        if (gam && nch == XC * 64) {
            u32x4 xall[XC], gall[XC];
#pragma unroll
            for (int i = 0; i < XC; ++i) {
                xall[i] = *reinterpret_cast<const u32x4*>(x + (lane + 64 * i) * EPC);
                gall[i] = *reinterpret_cast<const u32x4*>(gam + (lane + 64 * i) * EPC);
            }
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < XC; ++i) {
                float xv[EPC];
                to_f(xall[i], xv);
#pragma unroll
                for (int e = 0; e < EPC; ++e) ss += xv[e] * xv[e];
            }
            ss = wave_sum(ss);
            rstd = rsqrtf(ss / (float)p.K + p.rms_eps);
            float accr[R];
#pragma unroll
            for (int r = 0; r < R; ++r) accr[r] = 0.f;
            if constexpr (DOT2) {
                // weight * (x * rstd).to(bf16), rounded to bf16: rms_apply_reg's values, kept as packed pairs
                // (sharing this between the four waves of a workgroup through LDS -- a quarter of the work per wave, one barrier --
                // was measured slower: 2.887 vs 2.877 ms per token)
#pragma unroll
                for (int i = 0; i < XC; ++i) {
                    u32x4 xn;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t t = pack_bf2(bf_lo(xall[i][e]) * rstd, bf_hi(xall[i][e]) * rstd);
                        xn[e] = pack_bf2(bf_lo(gall[i][e]) * bf_lo(t), bf_hi(gall[i][e]) * bf_hi(t));
                    }
                    xall[i] = xn;
                }
#pragma unroll
                for (int b0 = 0; b0 < XC; b0 += U) {
                    u32x4 wr[U][R];
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (b0 + u < XC) wr[u][r] = b0 == 0 ? w_first[u][r] : ldw(w[r] + (lane + 64 * (b0 + u)) * EPC);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (b0 + u >= XC) continue;
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int e = 0; e < 4; ++e) accr[r] = dot2_bf16(wr[u][r][e], xall[b0 + u][e], accr[r]);
                    }
                }
            } else {
#pragma unroll
            for (int b0 = 0; b0 < XC; b0 += U) {
                constexpr int dummy = 0; (void)dummy;
                u32x4 wr[U][R];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (b0 + u < XC) wr[u][r] = b0 == 0 ? w_first[u][r] : ldw(w[r] + (lane + 64 * (b0 + u)) * EPC);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (b0 + u >= XC) continue;
                    float xa[EPC], ga[EPC];
                    to_f(xall[b0 + u], xa);
                    to_f(gall[b0 + u], ga);
                    rms_apply_reg(reinterpret_cast<float(&)[8]>(xa), reinterpret_cast<const float(&)[8]>(ga), rstd);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float wa[EPC];
                        to_f(wr[u][r], wa);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) accr[r] += wa[e] * xa[e];
                    }
                }
            }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) accr[r] = wave_sum(accr[r]);
            if (lane == 0) {
                EpiArgs e = p.epi;
                if (p.kv_k && n >= p.kv_col0)
                    e.C = (n < p.kv_col0 + p.kv_dim) ? p.kv_k - (long)p.kv_col0 * sizeof(T)
                                                     : p.kv_v - (long)(p.kv_col0 + p.kv_dim) * sizeof(T);
                if constexpr (R == 4) epilogue4<T>(e, 0, n, f32x4{accr[0], accr[1], accr[2], accr[3]});
                else epilogue2<T>(e, n, accr[0], accr[1]);
            }
            return;
        }
    }
    if (gam) {
        // every wave covers the whole x with its 64 lanes: the statistics need no LDS and no barrier
        float ss = 0.f;
        for (int c = lane; c < nch; c += 64) {
            float xv[EPC];
            if constexpr (sizeof(T) == 2) ld8_f<T>(x + c * EPC, xv);
            else { f32x4 t0 = ld4_f<T>(x + c * EPC); for (int e = 0; e < 4; ++e) xv[e] = t0[e]; }
#pragma unroll
            for (int e = 0; e < EPC; ++e) ss += xv[e] * xv[e];
        }
        ss = wave_sum(ss);
        rstd = rsqrtf(ss / (float)p.K + p.rms_eps);
    }
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    if (have_first) {
        u32x4 xr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xr[u] = *reinterpret_cast<const u32x4*>(x + (ch + 64 * u) * EPC);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (DOT2) {
                if (!gam) {
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[r] = dot2_bf16(w_first[u][r][e], xr[u][e], acc[r]);
                    continue;
                }
            }
            float xa[EPC];
            to_f(xr[u], xa);
            if (gam) rms_apply<T, EPC>(xa, gam + (ch + 64 * u) * EPC, rstd);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float wa[EPC];
                to_f(w_first[u][r], wa);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[r] += wa[e] * xa[e];
            }
        }
        ch += 64 * U;
    }

    for (; ch + 64 * (U - 1) < nch; ch += 64 * U) {
        u32x4 wr[U][R], xr[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) wr[u][r] = ldw(w[r] + (ch + 64 * u) * EPC);
#pragma unroll
        for (int u = 0; u < U; ++u) xr[u] = *reinterpret_cast<const u32x4*>(x + (ch + 64 * u) * EPC);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (DOT2) {
                if (!gam) {
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[r] = dot2_bf16(wr[u][r][e], xr[u][e], acc[r]);
                    continue;
                }
            }
            float xa[EPC];
            to_f(xr[u], xa);
            if (gam) rms_apply<T, EPC>(xa, gam + (ch + 64 * u) * EPC, rstd);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float wa[EPC];
                to_f(wr[u][r], wa);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[r] += wa[e] * xa[e];
            }
        }
    }
    for (; ch < nch; ch += 64) {
        u32x4 wr[R];
#pragma unroll
        for (int r = 0; r < R; ++r) wr[r] = ldw(w[r] + ch * EPC);
        if constexpr (DOT2) {
            if (!gam) {
                const u32x4 xraw = *reinterpret_cast<const u32x4*>(x + ch * EPC);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[r] = dot2_bf16(wr[r][e], xraw[e], acc[r]);
                continue;
            }
        }
        float xa[EPC];
        to_f(*reinterpret_cast<const u32x4*>(x + ch * EPC), xa);    // (this spelling: hoisting the load above the block doubles hipcc's register count)
        if (gam) rms_apply<T, EPC>(xa, gam + ch * EPC, rstd);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float wa[EPC];
            to_f(wr[r], wa);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[r] += wa[e] * xa[e];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
    if (lane == 0) {
        EpiArgs e = p.epi;
        if (p.kv_k && n >= p.kv_col0)   // region boundaries are multiples of 4: a wave's columns lie in one region
            e.C = (n < p.kv_col0 + p.kv_dim) ? p.kv_k - (long)p.kv_col0 * sizeof(T)
                                             : p.kv_v - (long)(p.kv_col0 + p.kv_dim) * sizeof(T);
        if constexpr (R == 4) epilogue4<T>(e, 0, n, f32x4{acc[0], acc[1], acc[2], acc[3]});
        else epilogue2<T>(e, n, acc[0], acc[1]);
    }
}

}  // namespace

int cogs_k_gemv(hipStream_t st, const CogsGemm& g) {
    const int es = g.dtype == COGS_DT_BF16 ? 2 : 4;
    if (g.N % 4 || (g.K * es) % 16 || (g.ldw * es) % 16) return COGS_E_INVALID;
    GemvArgs p;
    const int rc = cogs_fill_epi(g, &p.epi);
    if (rc != COGS_OK) return rc;
    p.x = (const char*)g.A;
    p.W = (const char*)g.W; p.ldw = g.ldw * es;
    p.N = g.N; p.K = g.K;
    p.rms_gamma = g.rms_gamma; p.rms_eps = g.rms_eps;
    p.kv_k = (char*)g.kv_k; p.kv_v = (char*)g.kv_v; p.kv_col0 = g.kv_col0; p.kv_dim = g.kv_dim;
    if (g.kv_k && (g.kv_col0 % 4 || g.kv_dim % 4 || g.out_f32 || g.act == COGS_ACT_SWIGLU)) return COGS_E_INVALID;
    // 4 rows per wave when that still gives every CU several workgroups, else 2 (N % 4 == 0 keeps pairs whole)
    const bool small_n = g.N / 16 < (int)g_cogs_debug.gemv_small_n;
    const bool dot2 = g_cogs_debug.gemv_dot2 != 0;
    if (small_n) {
        const int grid = (g.N / 2 + 3) / 4;
        if (g.dtype == COGS_DT_BF16 && dot2) hipLaunchKernelGGL((gemv_kernel<bf16_t, 2, GEMV_U_SMALL, true>), dim3(grid), dim3(256), 0, st, p);
        else if (g.dtype == COGS_DT_BF16) hipLaunchKernelGGL((gemv_kernel<bf16_t, 2, GEMV_U_SMALL>), dim3(grid), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemv_kernel<float, 2, 4>), dim3(grid), dim3(256), 0, st, p);
    } else {
        const int grid = (g.N / 4 + 3) / 4;
        if (g.dtype == COGS_DT_BF16 && dot2) hipLaunchKernelGGL((gemv_kernel<bf16_t, 4, GEMV_U_BIG, true>), dim3(grid), dim3(256), 0, st, p);
        else if (g.dtype == COGS_DT_BF16) hipLaunchKernelGGL((gemv_kernel<bf16_t, 4, GEMV_U_BIG>), dim3(grid), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemv_kernel<float, 4, 2>), dim3(grid), dim3(256), 0, st, p);
    }
    return COGS_LAUNCH_CHECK();
}
