// Weight-streaming GEMV for single-token decode:  y[1,N] = epilogue( x[1,K] . W[N,K]^T ).
//
// Replaces the M = 1 case of the Qwen2 linears during generation (transformers GenerationMixin
// decode steps behind model/cogreasoner_chat.py:802-807). HBM-bound: every weight byte is read
// exactly once, straight into registers (no LDS round trip: nothing is shared between waves);
// each wave owns 4 consecutive output rows (one epilogue group: bias / rope pair / swiglu pair /
// residual), lanes stride over K in 16-byte chunks, 8 independent 16-B loads in flight per lane
// per step; x is re-read from L2. fp32 accumulation, 64-lane butterfly per row.
#include "common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

struct GemvArgs {
    const char* x;
    const char* W; long ldw;  // bytes
    int N, K;
    const void* rms_gamma;    // != null: x is RMS-normalised on the fly (Qwen2RMSNorm fused into the GEMV)
    float rms_eps;
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

template <typename T>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs p) {
    constexpr int EPC = 16 / sizeof(T);  // elements per 16-byte chunk
    const int lane = threadIdx.x & 63;
    const int n = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (n >= p.N) return;
    const int nch = p.K / EPC;
    const T* x = reinterpret_cast<const T*>(p.x);
    const T* w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = reinterpret_cast<const T*>(p.W + (long)min(n + r, p.N - 1) * p.ldw);
    const T* gam = reinterpret_cast<const T*>(p.rms_gamma);
    float rstd = 1.f;
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
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int ch = lane;
    for (; ch + 64 < nch; ch += 128) {
        float xa[EPC], xb[EPC], wa[4][EPC], wb[4][EPC];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (sizeof(T) == 2) { ld8_f<T>(w[r] + ch * EPC, wa[r]); ld8_f<T>(w[r] + (ch + 64) * EPC, wb[r]); }
            else { f32x4 t0 = ld4_f<T>(w[r] + ch * EPC), t1 = ld4_f<T>(w[r] + (ch + 64) * EPC);
                   for (int e = 0; e < 4; ++e) { wa[r][e] = t0[e]; wb[r][e] = t1[e]; } }
        }
        if constexpr (sizeof(T) == 2) { ld8_f<T>(x + ch * EPC, xa); ld8_f<T>(x + (ch + 64) * EPC, xb); }
        else { f32x4 t0 = ld4_f<T>(x + ch * EPC), t1 = ld4_f<T>(x + (ch + 64) * EPC);
               for (int e = 0; e < 4; ++e) { xa[e] = t0[e]; xb[e] = t1[e]; } }
        if (gam) { rms_apply<T, EPC>(xa, gam + ch * EPC, rstd); rms_apply<T, EPC>(xb, gam + (ch + 64) * EPC, rstd); }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int e = 0; e < EPC; ++e) { acc[r] += wa[r][e] * xa[e]; acc[r] += wb[r][e] * xb[e]; }
    }
    for (; ch < nch; ch += 64) {
        float xa[EPC], wa[EPC];
        if constexpr (sizeof(T) == 2) ld8_f<T>(x + ch * EPC, xa);
        else { f32x4 t0 = ld4_f<T>(x + ch * EPC); for (int e = 0; e < 4; ++e) xa[e] = t0[e]; }
        if (gam) rms_apply<T, EPC>(xa, gam + ch * EPC, rstd);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (sizeof(T) == 2) ld8_f<T>(w[r] + ch * EPC, wa);
            else { f32x4 t0 = ld4_f<T>(w[r] + ch * EPC); for (int e = 0; e < 4; ++e) wa[e] = t0[e]; }
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[r] += wa[e] * xa[e];
        }
    }
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = wave_sum(acc[r]);
    if (lane == 0) epilogue4<T>(p.epi, 0, n, v);
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
    const int grid = (g.N / 4 + 3) / 4;
    if (g.dtype == COGS_DT_BF16) hipLaunchKernelGGL(gemv_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(gemv_kernel<float>, dim3(grid), dim3(256), 0, st, p);
    return COGS_LAUNCH_CHECK();
}
