// Question-aware visual-token compression kernels (all HBM-bound, one pass over their input).
//
//   pixdiff_mask           model/cogreasoner_chat.py:405-414  keep-mask of merged tokens whose mean
//                          |x_t - x_{t-1}| * 255 exceeds the threshold (frame 0 kept whole,
//                          at least min_tokens per frame), then the "minor frame" override
//                          (:416-422: only token 0 of an unimportant frame survives).
//   frame_mean_to_slot0    model/cogreasoner_chat.py:434-447  token 0 <- mean over the frame's tokens.
//   gather_rows            model/cogreasoner_chat.py:458-471,567-572  boolean gathers + embed_tokens +
//                          masked scatter, expressed as one row gather from two tables.
//   mean_rows / cosine     model/cogreasoner_chat.py:317-325  mean-pool over the sequence and cosine
//                          similarity of event summaries against the question.
//
// bf16 inputs reproduce torch's bf16 op-by-op rounding (difference, mean, *255 and the threshold
// are each rounded to bf16; accumulation is fp32) so that the keep-mask is bit-identical to the
// reference's on the same bf16 pixel_values; fp32 inputs follow the fp32 CPU path.
#include "common.h"
#include "kernels.h"

namespace {

template <typename T> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<bf16_t>(float v) { return bf2f(f2bf(v)); }
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }

// one wave per (frame f >= 1, token j); frame 0 is written as all ones
template <typename T>
__global__ __launch_bounds__(256) void pixdiff_kernel(const T* __restrict__ pix, int t, int P, int E, float thr,
                                                      uint8_t* __restrict__ mask) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (long)t * P) return;
    const int f = (int)(w / P);
    if (f == 0) {
        if (lane == 0) mask[w] = 1;
        return;
    }
    const T* cur = pix + w * E;
    const T* prv = cur - (long)P * E;
    float acc = 0.f;
    const int nch = E >> 3;
    // four chunk pairs (8 x 16-byte loads) in flight per lane, accumulated in the original order (the sum is the same)
    int ch = lane;
    for (; ch + 192 < nch; ch += 256) {
        float a[4][8], b[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ld8_f<T>(cur + (ch + 64 * u) * 8, a[u]);
            ld8_f<T>(prv + (ch + 64 * u) * 8, b[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += fabsf(rnd<T>(a[u][e] - b[u][e]));
    }
    for (; ch < nch; ch += 64) {
        float a[8], b[8];
        ld8_f<T>(cur + ch * 8, a);
        ld8_f<T>(prv + ch * 8, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += fabsf(rnd<T>(a[e] - b[e]));
    }
    for (int e = (nch << 3) + lane; e < E; e += 64) acc += fabsf(rnd<T>(ld_f<T>(cur + e) - ld_f<T>(prv + e)));
    acc = wave_sum(acc);
    if (lane == 0) {
        const float mean = rnd<T>(acc / (float)E);
        const float v = rnd<T>(mean * 255.0f);
        mask[w] = v > rnd<T>(thr) ? 1 : 0;
    }
}

// one block per frame: enforce min_tokens, then the minor-frame override
__global__ __launch_bounds__(256) void mask_fix_kernel(uint8_t* __restrict__ mask, int P, int min_tokens,
                                                       const uint8_t* __restrict__ minor) {
    __shared__ int cnt;
    const int f = blockIdx.x;
    uint8_t* m = mask + (long)f * P;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int c = 0;
    for (int j = threadIdx.x; j < P; j += blockDim.x) c += m[j];
    atomicAdd(&cnt, c);
    __syncthreads();
    if (cnt < min_tokens)
        for (int j = threadIdx.x; j < min_tokens && j < P; j += blockDim.x) m[j] = 1;
    if (minor && minor[f])
        for (int j = threadIdx.x; j < P; j += blockDim.x) m[j] = (j == 0) ? 1 : 0;
}

template <typename T>
__global__ __launch_bounds__(256) void frame_mean_kernel(T* __restrict__ feats, int P, int D,
                                                         const int* __restrict__ frames) {
    const int f = frames[blockIdx.x];
    T* base = feats + (long)f * P * D;
    for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < P; ++j) acc += ld4_f<T>(base + (long)j * D + d);
        acc *= (1.0f / (float)P);
        st4_f<T>(base + d, acc);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ ta, const T* __restrict__ tb,
                                                          const int64_t* __restrict__ idx, T* __restrict__ out,
                                                          int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int64_t i = idx[r];
    const T* src = i >= 0 ? ta + i * D : tb + (-i - 1) * D;
    T* dst = out + (long)r * D;
    constexpr int EPC = 16 / sizeof(T);
    const int nch = D / EPC;
    for (int ch = lane; ch < nch; ch += 64)
        *reinterpret_cast<u32x4*>(dst + ch * EPC) = *reinterpret_cast<const u32x4*>(src + ch * EPC);
}

// column means: block = 32 column chunks (of 4) x 8 row lanes
template <typename T>
__global__ __launch_bounds__(256) void mean_rows_kernel(const T* __restrict__ x, long ldx, int rows, int D,
                                                        float* __restrict__ out) {
    __shared__ f32x4 red[8][32];
    const int cc = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = (blockIdx.x * 32 + cc) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (col < D)
        for (int r = rl; r < rows; r += 8) acc += ld4_f<T>(x + (long)r * ldx + col);
    red[rl][cc] = acc;
    __syncthreads();
    if (rl == 0 && col < D) {
        f32x4 s = red[0][cc];
#pragma unroll
        for (int i = 1; i < 8; ++i) s += red[i][cc];
        s *= (1.0f / (float)rows);
        *reinterpret_cast<f32x4*>(out + col) = s;
    }
}

// out[i] = a.b_i / max(|a| |b_i|, 1e-8)
__global__ __launch_bounds__(64) void cosine_kernel(const float* __restrict__ a, const float* __restrict__ b, int D,
                                                    float* __restrict__ out) {
    const int lane = threadIdx.x, i = blockIdx.x;
    const float* bi = b + (long)i * D;
    float ab = 0.f, aa = 0.f, bb = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float x = a[d], y = bi[d];
        ab += x * y; aa += x * x; bb += y * y;
    }
    ab = wave_sum(ab); aa = wave_sum(aa); bb = wave_sum(bb);
    if (lane == 0) out[i] = ab / fmaxf(sqrtf(aa) * sqrtf(bb), 1e-8f);
}

}  // namespace

int cogs_k_pixdiff_mask(hipStream_t st, int dtype, const void* pix, int t, int P, int E, float thr, int min_tokens,
                        uint8_t* mask) {
    if (t <= 0 || P <= 0) return COGS_OK;
    if (E % 8) return COGS_E_INVALID;
    const long waves = (long)t * P;
    dim3 grid((unsigned)((waves + 3) / 4));
    if (dtype == COGS_DT_BF16)
        hipLaunchKernelGGL(pixdiff_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)pix, t, P, E, thr, mask);
    else
        hipLaunchKernelGGL(pixdiff_kernel<float>, grid, dim3(256), 0, st, (const float*)pix, t, P, E, thr, mask);
    (void)min_tokens;
    return COGS_LAUNCH_CHECK();
}

int cogs_k_mask_fix(hipStream_t st, uint8_t* mask, int t, int P, int min_tokens, const uint8_t* minor) {
    if (t <= 0) return COGS_OK;
    hipLaunchKernelGGL(mask_fix_kernel, dim3(t), dim3(256), 0, st, mask, P, min_tokens, minor);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_frame_mean_to_slot0(hipStream_t st, int dtype, void* feats, int P, int D, const int* frames, int n_frames) {
    if (n_frames <= 0) return COGS_OK;
    if (D % 4) return COGS_E_INVALID;
    if (dtype == COGS_DT_BF16)
        hipLaunchKernelGGL(frame_mean_kernel<bf16_t>, dim3(n_frames), dim3(256), 0, st, (bf16_t*)feats, P, D, frames);
    else
        hipLaunchKernelGGL(frame_mean_kernel<float>, dim3(n_frames), dim3(256), 0, st, (float*)feats, P, D, frames);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_gather_rows(hipStream_t st, int dtype, const void* ta, const void* tb, const int64_t* idx, void* out,
                       int rows, int D) {
    if (rows <= 0) return COGS_OK;
    if (D % 8) return COGS_E_INVALID;
    dim3 grid((rows + 3) / 4);
    if (dtype == COGS_DT_BF16)
        hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)ta, (const bf16_t*)tb, idx,
                           (bf16_t*)out, rows, D);
    else
        hipLaunchKernelGGL(gather_rows_kernel<float>, grid, dim3(256), 0, st, (const float*)ta, (const float*)tb, idx,
                           (float*)out, rows, D);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_mean_rows(hipStream_t st, int dtype, const void* x, long ldx, int rows, int D, float* out) {
    if (rows <= 0 || D % 4) return COGS_E_INVALID;
    dim3 grid((D / 4 + 31) / 32);
    if (dtype == COGS_DT_BF16)
        hipLaunchKernelGGL(mean_rows_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, ldx, rows, D, out);
    else
        hipLaunchKernelGGL(mean_rows_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ldx, rows, D, out);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_cosine(hipStream_t st, const float* a, const float* b, int n, int D, float* out) {
    if (n <= 0) return COGS_OK;
    hipLaunchKernelGGL(cosine_kernel, dim3(n), dim3(64), 0, st, a, b, D, out);
    return COGS_LAUNCH_CHECK();
}
