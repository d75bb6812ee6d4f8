// Row normalisations: LayerNorm (ViT), RMSNorm (Qwen2), and post-LayerNorm fused with the
// 2x2 token merge.
//
// Replaces: nn.LayerNorm(1152, eps=1e-6) x(2*27+1) (model/modeling_videollama3_encoder.py:382-384,
// :475), post_layernorm + un-permute + F.interpolate(bilinear, 1/2) (:482-501; with the
// merge-window-major row order every 4 consecutive rows are one 2x2 window, so the bilinear
// half-size resample is their mean), and transformers' Qwen2RMSNorm (fp32 variance,
// x*rsqrt(var+eps) cast back, then * weight).
//
// HBM-bound: one wave per row, the row lives in registers (16-byte loads, 8 elements per
// lane per chunk), statistics by 64-lane butterfly, one read + one write per element.
#include "common.h"
#include "kernels.h"

namespace {

template <typename T, int CPL, bool RMS>
__global__ __launch_bounds__(256) void norm_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                   const T* __restrict__ gamma, const T* __restrict__ beta,
                                                   int rows, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = H >> 3;
    const T* xr = x + (long)row * H;
    float v[CPL][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            ld8_f<T>(xr + ch * 8, v[c]);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += RMS ? v[c][e] * v[c][e] : v[c][e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
        }
    }
    sum = wave_sum(sum);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(sum / (float)H + eps);
    } else {
        mean = sum / (float)H;
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; sq += d * d; }
            }
        }
        sq = wave_sum(sq);
        rstd = rsqrtf(sq / (float)H + eps);
    }
    T* yr = y + (long)row * H;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            float gm[8], o[8];
            ld8_f<T>(gamma + ch * 8, gm);
            if (RMS) {
                // Qwen2RMSNorm: weight * (x*rstd).to(input_dtype)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float n = v[c][e] * rstd;
                    if (sizeof(T) == 2) n = bf2f(f2bf(n));
                    o[e] = gm[e] * n;
                }
            } else {
                float bt[8];
                ld8_f<T>(beta + ch * 8, bt);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (v[c][e] - mean) * rstd * gm[e] + bt[e];
            }
            st8_f<T>(yr + ch * 8, o);
        }
    }
}

template <typename T, int CPL>
__global__ __launch_bounds__(256) void ln_merge_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                       const T* __restrict__ gamma, const T* __restrict__ beta,
                                                       int out_rows, int group, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int orow = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (orow >= out_rows) return;
    const int nch = H >> 3;
    float acc[CPL][8];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[c][e] = 0.f;
    for (int gidx = 0; gidx < group; ++gidx) {
        const T* xr = x + ((long)orow * group + gidx) * H;
        float v[CPL][8];
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                ld8_f<T>(xr + ch * 8, v[c]);
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += v[c][e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
            }
        }
        sum = wave_sum(sum);
        const float mean = sum / (float)H;
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; sq += d * d; }
            }
        }
        sq = wave_sum(sq);
        const float rstd = rsqrtf(sq / (float)H + eps);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                float gm[8], bt[8];
                ld8_f<T>(gamma + ch * 8, gm);
                ld8_f<T>(beta + ch * 8, bt);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float n = (v[c][e] - mean) * rstd * gm[e] + bt[e];
                    if (sizeof(T) == 2) n = bf2f(f2bf(n));  // post_layernorm output is stored in T
                    acc[c][e] += n;
                }
            }
        }
    }
    const float invg = 1.0f / (float)group;
    T* yr = y + (long)orow * H;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = acc[c][e] * invg;
            st8_f<T>(yr + ch * 8, o);
        }
    }
}

template <typename T, bool RMS>
int launch_norm(hipStream_t st, const void* x, void* y, const void* g, const void* b, int rows, int H, float eps) {
    const int cpl = (H / 8 + 63) / 64;
    dim3 grid((rows + 3) / 4), blk(256);
#define COGS_NORM_CASE(C)                                                                             \
    hipLaunchKernelGGL((norm_kernel<T, C, RMS>), grid, blk, 0, st, (const T*)x, (T*)y, (const T*)g, \
                       (const T*)b, rows, H, eps)
    if (cpl <= 1) COGS_NORM_CASE(1);
    else if (cpl <= 2) COGS_NORM_CASE(2);
    else if (cpl <= 3) COGS_NORM_CASE(3);
    else if (cpl <= 4) COGS_NORM_CASE(4);
    else if (cpl <= 8) COGS_NORM_CASE(8);
    else return COGS_E_UNSUPPORTED;
#undef COGS_NORM_CASE
    return COGS_LAUNCH_CHECK();
}

template <typename T>
int launch_ln_merge(hipStream_t st, const void* x, void* y, const void* g, const void* b, int out_rows, int group,
                    int H, float eps) {
    const int cpl = (H / 8 + 63) / 64;
    dim3 grid((out_rows + 3) / 4), blk(256);
#define COGS_LNM_CASE(C)                                                                            \
    hipLaunchKernelGGL((ln_merge_kernel<T, C>), grid, blk, 0, st, (const T*)x, (T*)y, (const T*)g, \
                       (const T*)b, out_rows, group, H, eps)
    if (cpl <= 1) COGS_LNM_CASE(1);
    else if (cpl <= 2) COGS_LNM_CASE(2);
    else if (cpl <= 3) COGS_LNM_CASE(3);
    else if (cpl <= 4) COGS_LNM_CASE(4);
    else if (cpl <= 8) COGS_LNM_CASE(8);
    else return COGS_E_UNSUPPORTED;
#undef COGS_LNM_CASE
    return COGS_LAUNCH_CHECK();
}

// (a, b) = (rstd, -rstd * mean) of every row from the per-tile partial sums the producing GEMM's epilogue wrote
// (EPI_ROWSTAT): fixed-order fp64 combination, biased variance like nn.LayerNorm
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ part, int rows, int tiles, float inv_h,
                                                          float eps, float* __restrict__ ab) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const f32x2* p = reinterpret_cast<const f32x2*>(part) + (long)r * tiles;
    double s1 = 0.0, s2 = 0.0;
    for (int t = 0; t < tiles; ++t) { const f32x2 v = p[t]; s1 += (double)v[0]; s2 += (double)v[1]; }
    const double mean = s1 * inv_h;
    const double var = fmax(s2 * inv_h - mean * mean, 0.0);
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    *reinterpret_cast<f32x2*>(ab + 2 * (long)r) = f32x2{rstd, (float)(-(double)rstd * mean)};
}

}  // namespace

int cogs_k_layernorm(hipStream_t st, int dtype, const void* x, void* y, const void* gamma, const void* beta,
                     int rows, int H, float eps) {
    if (rows <= 0) return COGS_OK;
    if (H % 8) return COGS_E_INVALID;
    return dtype == COGS_DT_BF16 ? launch_norm<bf16_t, false>(st, x, y, gamma, beta, rows, H, eps)
                                 : launch_norm<float, false>(st, x, y, gamma, beta, rows, H, eps);
}

int cogs_k_rmsnorm(hipStream_t st, int dtype, const void* x, void* y, const void* gamma, int rows, int H, float eps) {
    if (rows <= 0) return COGS_OK;
    if (H % 8) return COGS_E_INVALID;
    return dtype == COGS_DT_BF16 ? launch_norm<bf16_t, true>(st, x, y, gamma, nullptr, rows, H, eps)
                                 : launch_norm<float, true>(st, x, y, gamma, nullptr, rows, H, eps);
}

int cogs_k_ln_merge(hipStream_t st, int dtype, const void* x, void* y, const void* gamma, const void* beta,
                    int out_rows, int group, int H, float eps) {
    if (out_rows <= 0) return COGS_OK;
    if (H % 8 || group <= 0) return COGS_E_INVALID;
    return dtype == COGS_DT_BF16 ? launch_ln_merge<bf16_t>(st, x, y, gamma, beta, out_rows, group, H, eps)
                                 : launch_ln_merge<float>(st, x, y, gamma, beta, out_rows, group, H, eps);
}

int cogs_k_ln_finalize(hipStream_t st, const float* stat_part, int rows, int tiles, int H, float eps, float* ab) {
    if (rows <= 0 || tiles <= 0 || H <= 0) return COGS_E_INVALID;
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, stat_part, rows, tiles, 1.0f / (float)H,
                       eps, ab);
    return COGS_LAUNCH_CHECK();
}
