// Time-aware k-means device steps (model/kmeans_with_time.py:4-137). The host keeps the
// reference's RNG draws (random.randint / torch.multinomial on the CPU generator) and the
// iteration control; each Lloyd iteration is three launches:
//
//   sqdist   dist2[t,k] = sum_j (x[t,j]-c[k,j])^2   (kmeans_with_time.py:48,73  torch.cdist)
//   assign   per-row min-max of feature/time distances, sqrt(nf^2 + alpha nt^2), argmin,
//            member lists                            (:76-104)
//   update   per-cluster means (or reseed row), centre shift norms   (:107-125)
//
// HBM-bound: features [T, P*D] (bf16 or fp32, 92/183 MB at T=256) are read exactly once by
// sqdist and once by update per iteration; all sums run in a fixed order (slice partials are
// combined in slice order in fp64) so assignments are reproducible run to run.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int SL = 512;       // columns per slice
constexpr int KMAX = 32;      // clusters per launch (K = ceil(T/15) <= 12 for 180 frames, 18 for 256)

// grid.x = slices; 4 waves; LDS holds the centre slice [K][SL] fp32
template <typename T>
__global__ __launch_bounds__(256) void sqdist_kernel(const T* __restrict__ x, int Tn, long PD,
                                                     const float* __restrict__ centres,
                                                     const int* __restrict__ centre_rows, int K,
                                                     float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* cs = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long j0 = (long)blockIdx.x * SL;
    for (int i = tid; i < K * (SL / 8); i += 256) {
        const int k = i / (SL / 8), c = i % (SL / 8);
        const long j = j0 + c * 8;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (j < PD) {
            if (centre_rows) ld8_f<T>(x + (long)centre_rows[k] * PD + j, v);
            else ld8_f<float>(centres + (long)k * PD + j, v);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[k * SL + c * 8 + e] = v[e];
    }
    __syncthreads();
    const long j = j0 + lane * 8;
    for (int t = wid; t < Tn; t += 4) {
        float xv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool in = j < PD;
        if (in) ld8_f<T>(x + (long)t * PD + j, xv);
        for (int k = 0; k < K; ++k) {
            float acc = 0.f;
            if (in) {
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs + k * SL + lane * 8);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(cs + k * SL + lane * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = xv[e] - c0[e], d1 = xv[4 + e] - c1[e];
                    acc += d0 * d0;
                    acc += d1 * d1;
                }
            }
            acc = wave_sum(acc);
            if (lane == 0) partial[((long)blockIdx.x * Tn + t) * K + k] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void sqdist_reduce_kernel(const float* __restrict__ partial, int nslices, int TK,
                                                            float* __restrict__ dist2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= TK) return;
    double s = 0.0;
    for (int sl = 0; sl < nslices; ++sl) s += (double)partial[(long)sl * TK + i];
    dist2[i] = (float)s;
}

// single block, one thread per row t
__global__ __launch_bounds__(1024) void assign_kernel(const float* __restrict__ dist2, const float* __restrict__ ts,
                                                      const float* __restrict__ cts, int Tn, int K, float alpha,
                                                      int64_t* __restrict__ assign, int* __restrict__ counts) {
    __shared__ int cnt[KMAX];
    const int t = threadIdx.x;
    if (t < KMAX) cnt[t] = 0;
    __syncthreads();
    if (t < Tn) {
        float fmin_ = INFINITY, fmax_ = -INFINITY, tmin_ = INFINITY, tmax_ = -INFINITY;
        for (int k = 0; k < K; ++k) {
            const float df = sqrtf(dist2[t * K + k]);
            const float dt = fabsf(ts[t] - cts[k]);
            fmin_ = fminf(fmin_, df); fmax_ = fmaxf(fmax_, df);
            tmin_ = fminf(tmin_, dt); tmax_ = fmaxf(tmax_, dt);
        }
        float best = INFINITY;
        int bk = 0;
        for (int k = 0; k < K; ++k) {
            const float df = sqrtf(dist2[t * K + k]);
            const float dt = fabsf(ts[t] - cts[k]);
            const float nf = fmax_ > fmin_ ? (df - fmin_) / (fmax_ - fmin_) : 0.f;
            const float nt = tmax_ > tmin_ ? (dt - tmin_) / (tmax_ - tmin_) : 0.f;
            const float fd = sqrtf(nf * nf + alpha * (nt * nt));
            if (fd < best) { best = fd; bk = k; }
        }
        assign[t] = bk;
        atomicAdd(&cnt[bk], 1);
    }
    __syncthreads();
    if (t < K) counts[t] = cnt[t];
}

// each thread owns 4 columns; clusters and members are walked in ascending order
template <typename T>
__global__ __launch_bounds__(256) void update_kernel(const T* __restrict__ x, int Tn, long PD, int K,
                                                     const int64_t* __restrict__ assign,
                                                     const int* __restrict__ reseed_rows,
                                                     float* __restrict__ centres, float* __restrict__ shift_partial) {
    __shared__ short members[1024];
    __shared__ int offs[KMAX + 1];
    __shared__ float red[4][KMAX];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) {
        int o = 0;
        for (int k = 0; k < K; ++k) {
            offs[k] = o;
            for (int t = 0; t < Tn; ++t)
                if ((int)assign[t] == k) members[o++] = (short)t;
        }
        offs[K] = o;
    }
    __syncthreads();
    const long col = ((long)blockIdx.x * 256 + tid) * 4;
    const bool in = col < PD;
    for (int k = 0; k < K; ++k) {
        float sh = 0.f;
        if (in) {
            const int b = offs[k], e = offs[k + 1];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (e > b) {
                for (int i = b; i < e; ++i) acc += ld4_f<T>(x + (long)members[i] * PD + col);
                acc /= (float)(e - b);
            } else {
                acc = ld4_f<T>(x + (long)reseed_rows[k] * PD + col);
            }
            float* cp = centres + (long)k * PD + col;
            const f32x4 old = *reinterpret_cast<const f32x4*>(cp);
            const f32x4 d = acc - old;
            sh = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
            *reinterpret_cast<f32x4*>(cp) = acc;
        }
        sh = wave_sum(sh);
        if (lane == 0) red[wid][k] = sh;
    }
    __syncthreads();
    if (tid < K) shift_partial[(long)blockIdx.x * K + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// single block: centre times, shift norms, total movement
__global__ __launch_bounds__(64) void update_final_kernel(const float* __restrict__ ts, int Tn, int K,
                                                          const int64_t* __restrict__ assign,
                                                          const int* __restrict__ reseed_rows,
                                                          float* __restrict__ cts,
                                                          const float* __restrict__ shift_partial, int nblk,
                                                          float* __restrict__ shift_out) {
    __shared__ float feat_norm[KMAX];
    __shared__ float dts[KMAX];
    const int k = threadIdx.x;
    if (k < K) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += (double)shift_partial[(long)b * K + k];
        feat_norm[k] = sqrtf((float)s);
        float tsum = 0.f;
        int n = 0;
        for (int t = 0; t < Tn; ++t)
            if ((int)assign[t] == k) { tsum += ts[t]; ++n; }
        const float nt = n > 0 ? tsum / (float)n : ts[reseed_rows[k]];
        const float d = nt - cts[k];
        dts[k] = d * d;
        cts[k] = nt;
    }
    __syncthreads();
    if (k == 0) {
        float f = 0.f, tt = 0.f;
        for (int i = 0; i < K; ++i) { f += feat_norm[i]; tt += dts[i]; }
        shift_out[0] = f + sqrtf(tt);
    }
}

}  // namespace

size_t cogs_k_kmeans_ws(int T, long PD, int K, int* nslices) {
    const int ns = (int)((PD + SL - 1) / SL);
    if (nslices) *nslices = ns;
    const size_t a = (size_t)ns * T * K * sizeof(float);
    const size_t b = (size_t)cogs_k_kmeans_update_blocks(PD) * K * sizeof(float);
    return a > b ? a : b;
}

int cogs_k_kmeans_update_blocks(long PD) { return (int)((PD / 4 + 255) / 256); }

int cogs_k_kmeans_sqdist(hipStream_t st, int dtype, const void* feats, int T, long PD, const float* centres,
                         const int* centre_rows, int K, float* partial, int nslices, float* dist2) {
    if (K <= 0 || K > KMAX || T <= 0 || PD % 8) return COGS_E_INVALID;
    if (nslices != (int)((PD + SL - 1) / SL)) return COGS_E_WORKSPACE;
    const size_t lds = (size_t)K * SL * sizeof(float);
    if (dtype == COGS_DT_BF16) {
        (void)hipFuncSetAttribute((const void*)sqdist_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, KMAX * SL * 4);
        hipLaunchKernelGGL(sqdist_kernel<bf16_t>, dim3(nslices), dim3(256), lds, st, (const bf16_t*)feats, T, PD,
                           centres, centre_rows, K, partial);
    } else {
        (void)hipFuncSetAttribute((const void*)sqdist_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, KMAX * SL * 4);
        hipLaunchKernelGGL(sqdist_kernel<float>, dim3(nslices), dim3(256), lds, st, (const float*)feats, T, PD, centres,
                           centre_rows, K, partial);
    }
    const int TK = T * K;
    hipLaunchKernelGGL(sqdist_reduce_kernel, dim3((TK + 255) / 256), dim3(256), 0, st, partial, nslices, TK, dist2);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_kmeans_assign(hipStream_t st, const float* dist2, const float* ts, const float* centre_ts, int T, int K,
                         float alpha, int64_t* assign, int* counts) {
    if (T <= 0 || T > 1024 || K <= 0 || K > KMAX) return COGS_E_INVALID;
    hipLaunchKernelGGL(assign_kernel, dim3(1), dim3(1024), 0, st, dist2, ts, centre_ts, T, K, alpha, assign, counts);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_kmeans_update(hipStream_t st, int dtype, const void* feats, const float* ts, int T, long PD, int K,
                         const int64_t* assign, const int* reseed_rows, float* centres, float* centre_ts,
                         float* shift_partial, int nblk, float* shift_out) {
    if (T <= 0 || T > 1024 || K <= 0 || K > KMAX || PD % 4) return COGS_E_INVALID;
    if (nblk != cogs_k_kmeans_update_blocks(PD)) return COGS_E_WORKSPACE;
    if (dtype == COGS_DT_BF16)
        hipLaunchKernelGGL(update_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, (const bf16_t*)feats, T, PD, K, assign,
                           reseed_rows, centres, shift_partial);
    else
        hipLaunchKernelGGL(update_kernel<float>, dim3(nblk), dim3(256), 0, st, (const float*)feats, T, PD, K, assign,
                           reseed_rows, centres, shift_partial);
    hipLaunchKernelGGL(update_final_kernel, dim3(1), dim3(64), 0, st, ts, T, K, assign, reseed_rows, centre_ts,
                       shift_partial, nblk, shift_out);
    return COGS_LAUNCH_CHECK();
}
