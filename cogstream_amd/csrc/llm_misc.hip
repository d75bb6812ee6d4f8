// Small Qwen2-side helpers: KV-cache append, greedy argmax, HF logits processors, top-k.
//
// Replaces (third-party, transformers==4.46.3 as pinned by environment.yml:404):
//   DynamicCache.update (KV append), GenerationMixin greedy argmax,
//   RepetitionPenaltyLogitsProcessor / TemperatureLogitsWarper / TopKLogitsWarper
//   (model/generation_config.json:2-12) and the allowed-token mask of
//   StructuredLogitsProcessor (model/qaselect_module_predict.py:86-103).
#include "common.h"
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void copy_cols_kernel(const char* __restrict__ src, long ld_src,
                                                        char* __restrict__ dst, long ld_dst, int rows, int chunks) {
    const long total = (long)rows * chunks;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / chunks;
        const int c = (int)(i % chunks);
        *reinterpret_cast<u32x4*>(dst + r * ld_dst + c * 16) = *reinterpret_cast<const u32x4*>(src + r * ld_src + c * 16);
    }
}

__global__ __launch_bounds__(256) void kv_append_kernel(const char* __restrict__ ks, const char* __restrict__ vs,
                                                        long ld_src, char* __restrict__ kd, char* __restrict__ vd,
                                                        long ld_dst, int rows, int chunks) {
    const long total = (long)rows * chunks * 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int which = (int)(i & 1);
        const long j = i >> 1;
        const long r = j / chunks;
        const int c = (int)(j % chunks);
        const char* s = (which ? vs : ks) + r * ld_src + c * 16;
        char* d = (which ? vd : kd) + r * ld_dst + c * 16;
        *reinterpret_cast<u32x4*>(d) = *reinterpret_cast<const u32x4*>(s);
    }
}

// first-maximum argmax, two stages
constexpr int AM_BLOCKS = 64;
__global__ __launch_bounds__(256) void argmax_part_kernel(const float* __restrict__ x, int n, float* __restrict__ ws) {
    __shared__ float sv[4];
    __shared__ int si[4];
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += AM_BLOCKS * 256) {
        const float v = x[i];
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { sv[wid] = bv; si[wid] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        ws[blockIdx.x] = bv;
        reinterpret_cast<int*>(ws)[AM_BLOCKS + blockIdx.x] = bi;
    }
}
__global__ __launch_bounds__(64) void argmax_final_kernel(const float* __restrict__ ws, int64_t* __restrict__ out) {
    const int lane = threadIdx.x;
    float bv = ws[lane];
    int bi = reinterpret_cast<const int*>(ws)[AM_BLOCKS + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) out[0] = bi == 0x7fffffff ? 0 : bi;
}

__global__ __launch_bounds__(1024) void logits_process_kernel(float* __restrict__ logits, int n,
                                                              const int64_t* __restrict__ prev, int n_prev, float pen,
                                                              const int32_t* __restrict__ allowed, int n_allowed,
                                                              float temperature, float* __restrict__ tmp) {
    const int tid = threadIdx.x;
    if (n_prev > 0 && pen != 1.0f) {
        for (int i = tid; i < n_prev; i += 1024) tmp[i] = logits[prev[i]];
        __syncthreads();
        for (int i = tid; i < n_prev; i += 1024) {
            const float s = tmp[i];
            logits[prev[i]] = s < 0.f ? s * pen : s / pen;
        }
        __syncthreads();
    }
    if (n_allowed > 0) {
        __shared__ float keep[256];
        for (int i = tid; i < n_allowed; i += 1024) keep[i] = logits[allowed[i]];
        __syncthreads();
        for (int i = tid; i < n; i += 1024) logits[i] = -INFINITY;
        __syncthreads();
        for (int i = tid; i < n_allowed; i += 1024) logits[allowed[i]] = keep[i];
        __syncthreads();
    }
    if (temperature != 1.0f)
        for (int i = tid; i < n; i += 1024) logits[i] = logits[i] / temperature;
}

__global__ __launch_bounds__(1024) void topk_kernel(const float* __restrict__ logits, int n, int k,
                                                    float* __restrict__ tv, int32_t* __restrict__ ti,
                                                    float* __restrict__ work) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < n; i += 1024) work[i] = logits[i];
    __syncthreads();
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < n; i += 1024) {
            const float v = work[i];
            if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { sv[wid] = bv; si[wid] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 16; ++w)
                if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
            tv[r] = bv;
            ti[r] = bi == 0x7fffffff ? -1 : bi;
            if (bi != 0x7fffffff) work[bi] = -INFINITY;
            // a row of all -inf keeps returning -1; the host treats that as "fewer than k candidates"
        }
        __syncthreads();
    }
}

}  // namespace

int cogs_k_copy_cols(hipStream_t st, int dtype, const void* src, long ld_src, void* dst, long ld_dst, int rows, int cols) {
    if (rows <= 0 || cols <= 0) return COGS_OK;
    const int es = dtype == COGS_DT_BF16 ? 2 : 4;
    if ((cols * es) % 16 || (ld_src * es) % 16 || (ld_dst * es) % 16) return COGS_E_INVALID;
    const int chunks = cols * es / 16;
    long g = ((long)rows * chunks + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)g), dim3(256), 0, st, (const char*)src, ld_src * es, (char*)dst,
                       ld_dst * es, rows, chunks);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_kv_append(hipStream_t st, int dtype, const void* k_src, const void* v_src, long ld_src, void* k_dst,
                     void* v_dst, long ld_dst, int rows, int cols) {
    if (rows <= 0 || cols <= 0) return COGS_OK;
    const int es = dtype == COGS_DT_BF16 ? 2 : 4;
    if ((cols * es) % 16 || (ld_src * es) % 16 || (ld_dst * es) % 16) return COGS_E_INVALID;
    const int chunks = cols * es / 16;
    long g = ((long)rows * chunks * 2 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(kv_append_kernel, dim3((unsigned)g), dim3(256), 0, st, (const char*)k_src, (const char*)v_src,
                       ld_src * es, (char*)k_dst, (char*)v_dst, ld_dst * es, rows, chunks);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_argmax(hipStream_t st, const float* logits, int n, int64_t* out, float* ws) {
    if (n <= 0) return COGS_E_INVALID;
    hipLaunchKernelGGL(argmax_part_kernel, dim3(AM_BLOCKS), dim3(256), 0, st, logits, n, ws);
    hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(64), 0, st, ws, out);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_logits_process(hipStream_t st, float* logits, int n, const int64_t* prev, int n_prev, float rep_penalty,
                          const int32_t* allowed, int n_allowed, float temperature, float* tmp) {
    if (n <= 0 || n_allowed > 256) return COGS_E_INVALID;
    hipLaunchKernelGGL(logits_process_kernel, dim3(1), dim3(1024), 0, st, logits, n, prev, n_prev, rep_penalty, allowed,
                       n_allowed, temperature, tmp);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_topk(hipStream_t st, const float* logits, int n, int top_k, float* topk_val, int32_t* topk_idx, float* ws) {
    if (n <= 0 || top_k <= 0 || top_k > 1024) return COGS_E_INVALID;
    hipLaunchKernelGGL(topk_kernel, dim3(1), dim3(1024), 0, st, logits, n, top_k, topk_val, topk_idx, ws);
    return COGS_LAUNCH_CHECK();
}
