// Time-aware k-means device steps (model/kmeans_with_time.py:4-137). The host keeps the
// reference's RNG draws (random.randint / torch.multinomial on the CPU generator) and the
// iteration control; each Lloyd iteration is three launches:
//
//   sqdist   dist2[t,k] = sum_j (x[t,j]-c[k,j])^2   (kmeans_with_time.py:48,73  torch.cdist)
//   assign   per-row min-max of feature/time distances, sqrt(nf^2 + alpha nt^2), argmin   (:76-104)
//   update   per-cluster means (or reseed row), centre shift norms   (:107-125)
//
// HBM-bound: features [T, P*D] (bf16 or fp32, 92/183 MB at T=256) are read exactly once by
// sqdist and once by update per iteration; all sums run in a fixed order (slice partials are
// combined in slice order in fp64) so assignments are reproducible run to run.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int SL = 512;       // columns per slice
constexpr int KMAX = 32;      // clusters per sqdist launch / per shift-partial chunk (K itself is unbounded:
                              // K = ceil(T/15) is 18 for 256 frames, 40 for a 600-frame session)

// grid.x = slices; 4 waves; LDS holds the centre slice [K][SL] fp32
template <typename T>
__global__ __launch_bounds__(256) void sqdist_kernel(const T* __restrict__ x, int Tn, long PD,
                                                     const float* __restrict__ centres,
                                                     const int* __restrict__ centre_rows, int K, int k0, int Ktot,
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
            if (centre_rows) ld8_f<T>(x + (long)centre_rows[k0 + k] * PD + j, v);
            else ld8_f<float>(centres + (long)(k0 + k) * PD + j, v);
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
            if (lane == 0) partial[((long)blockIdx.x * Tn + t) * Ktot + k0 + k] = acc;
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

// one thread per row t, any T and K; counts must be zero on entry
__global__ __launch_bounds__(256) void assign_kernel(const float* __restrict__ dist2, const float* __restrict__ ts,
                                                     const float* __restrict__ cts, int Tn, int K, float alpha,
                                                     int64_t* __restrict__ assign, int* __restrict__ counts) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= Tn) return;
    const float* d2 = dist2 + (long)t * K;
    float fmin_ = INFINITY, fmax_ = -INFINITY, tmin_ = INFINITY, tmax_ = -INFINITY;
    for (int k = 0; k < K; ++k) {
        const float df = sqrtf(d2[k]);
        const float dt = fabsf(ts[t] - cts[k]);
        fmin_ = fminf(fmin_, df); fmax_ = fmaxf(fmax_, df);
        tmin_ = fminf(tmin_, dt); tmax_ = fmaxf(tmax_, dt);
    }
    float best = INFINITY;
    int bk = 0;
    for (int k = 0; k < K; ++k) {
        const float df = sqrtf(d2[k]);
        const float dt = fabsf(ts[t] - cts[k]);
        const float nf = fmax_ > fmin_ ? (df - fmin_) / (fmax_ - fmin_) : 0.f;
        const float nt = tmax_ > tmin_ ? (dt - tmin_) / (tmax_ - tmin_) : 0.f;
        const float fd = sqrtf(nf * nf + alpha * (nt * nt));
        if (fd < best) { best = fd; bk = k; }
    }
    assign[t] = bk;
    atomicAdd(&counts[bk], 1);
}

// member lists (rows of every cluster in ascending order) for the update step: offs [K+1], members [T].
// One workgroup; thread k walks the assignment vector for cluster k (K may exceed the workgroup: strided).
__global__ __launch_bounds__(256) void members_kernel(const int64_t* __restrict__ assign, int Tn, int K,
                                                      int* __restrict__ offs, int* __restrict__ members) {
    for (int k = threadIdx.x; k < K; k += 256) {
        int n = 0;
        for (int t = 0; t < Tn; ++t) n += ((int)assign[t] == k) ? 1 : 0;
        offs[k + 1] = n;        // sizes first; prefix below
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int o = 0;
        offs[0] = 0;
        for (int k = 0; k < K; ++k) { const int n = offs[k + 1]; offs[k + 1] = o + n; o += n; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) {
        int o = offs[k];
        for (int t = 0; t < Tn; ++t)
            if ((int)assign[t] == k) members[o++] = t;
    }
}

// each thread owns 4 columns; clusters and members are walked in ascending order
template <typename T>
__global__ __launch_bounds__(256) void update_kernel(const T* __restrict__ x, int Tn, long PD, int K,
                                                     const int* __restrict__ offs, const int* __restrict__ members,
                                                     const int* __restrict__ reseed_rows,
                                                     float* __restrict__ centres, float* __restrict__ shift_partial) {
    __shared__ float red[4][KMAX];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long col = ((long)blockIdx.x * 256 + tid) * 4;
    const bool in = col < PD;
    for (int kc = 0; kc < K; kc += KMAX) {
        const int kn = min(KMAX, K - kc);
        for (int kk = 0; kk < kn; ++kk) {
            const int k = kc + kk;
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
            if (lane == 0) red[wid][kk] = sh;
        }
        __syncthreads();
        if (tid < kn) shift_partial[(long)blockIdx.x * K + kc + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        __syncthreads();
    }
}

// one workgroup: centre times, shift norms (scratch [2K]: feature-shift norm, squared time shift), total movement
__global__ __launch_bounds__(256) void update_final_kernel(const float* __restrict__ ts, int K,
                                                           const int* __restrict__ offs, const int* __restrict__ members,
                                                           const int* __restrict__ reseed_rows,
                                                           float* __restrict__ cts,
                                                           const float* __restrict__ shift_partial, int nblk,
                                                           float* __restrict__ scratch, float* __restrict__ shift_out) {
    for (int k = threadIdx.x; k < K; k += 256) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += (double)shift_partial[(long)b * K + k];
        scratch[k] = sqrtf((float)s);
        float tsum = 0.f;
        const int b0 = offs[k], e0 = offs[k + 1];
        for (int i = b0; i < e0; ++i) tsum += ts[members[i]];
        const float nt = e0 > b0 ? tsum / (float)(e0 - b0) : ts[reseed_rows[k]];
        const float d = nt - cts[k];
        scratch[K + k] = d * d;
        cts[k] = nt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float f = 0.f, tt = 0.f;
        for (int i = 0; i < K; ++i) { f += scratch[i]; tt += scratch[K + i]; }
        shift_out[0] = f + sqrtf(tt);
    }
}

}  // namespace

// workspace: max(sqdist slice partials [ns][T][K], update shift partials [nblk][K]) floats, then the member lists
// (offs [K+1], members [T]) and the update scratch [2K]
static size_t ws_floats(int T, long PD, int K, int ns) {
    const size_t a = (size_t)ns * T * K;
    const size_t b = (size_t)cogs_k_kmeans_update_blocks(PD) * K;
    return a > b ? a : b;
}
size_t cogs_k_kmeans_ws(int T, long PD, int K, int* nslices) {
    const int ns = (int)((PD + SL - 1) / SL);
    if (nslices) *nslices = ns;
    return (ws_floats(T, PD, K, ns) + (size_t)(K + 1) + (size_t)T + 2 * (size_t)K) * 4;
}

int cogs_k_kmeans_update_blocks(long PD) { return (int)((PD / 4 + 255) / 256); }

int cogs_k_kmeans_sqdist(hipStream_t st, int dtype, const void* feats, int T, long PD, const float* centres,
                         const int* centre_rows, int K, float* partial, int nslices, float* dist2) {
    if (K <= 0 || T <= 0 || PD % 8) return COGS_E_INVALID;
    if (nslices != (int)((PD + SL - 1) / SL)) return COGS_E_WORKSPACE;
    static std::atomic<uint64_t> done_bf{0}, done_f{0};
    for (int k0 = 0; k0 < K; k0 += KMAX) {          // the LDS centre slice holds KMAX clusters at a time
        const int kb = K - k0 < KMAX ? K - k0 : KMAX;
        const size_t lds = (size_t)kb * SL * sizeof(float);
        if (dtype == COGS_DT_BF16) {
            cogs_ensure_dyn_lds((const void*)sqdist_kernel<bf16_t>, KMAX * SL * 4, done_bf);
            hipLaunchKernelGGL(sqdist_kernel<bf16_t>, dim3(nslices), dim3(256), lds, st, (const bf16_t*)feats, T, PD,
                               centres, centre_rows, kb, k0, K, partial);
        } else {
            cogs_ensure_dyn_lds((const void*)sqdist_kernel<float>, KMAX * SL * 4, done_f);
            hipLaunchKernelGGL(sqdist_kernel<float>, dim3(nslices), dim3(256), lds, st, (const float*)feats, T, PD, centres,
                               centre_rows, kb, k0, K, partial);
        }
    }
    const long TK = (long)T * K;
    if (TK > 0x7fffffff) return COGS_E_INVALID;
    hipLaunchKernelGGL(sqdist_reduce_kernel, dim3((unsigned)((TK + 255) / 256)), dim3(256), 0, st, partial, nslices, (int)TK, dist2);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_kmeans_assign(hipStream_t st, const float* dist2, const float* ts, const float* centre_ts, int T, int K,
                         float alpha, int64_t* assign, int* counts) {
    if (T <= 0 || K <= 0) return COGS_E_INVALID;
    if (hipMemsetAsync(counts, 0, (size_t)K * sizeof(int), st) != hipSuccess) return COGS_E_HIP;
    hipLaunchKernelGGL(assign_kernel, dim3((T + 255) / 256), dim3(256), 0, st, dist2, ts, centre_ts, T, K, alpha, assign, counts);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_kmeans_update(hipStream_t st, int dtype, const void* feats, const float* ts, int T, long PD, int K,
                         const int64_t* assign, const int* reseed_rows, float* centres, float* centre_ts,
                         float* ws, int nblk, float* shift_out) {
    if (T <= 0 || K <= 0 || PD % 4) return COGS_E_INVALID;
    if (nblk != cogs_k_kmeans_update_blocks(PD)) return COGS_E_WORKSPACE;
    int ns = 0;
    (void)cogs_k_kmeans_ws(T, PD, K, &ns);
    float* shift_partial = ws;
    int* offs = (int*)(ws + ws_floats(T, PD, K, ns));
    int* members = offs + K + 1;
    float* scratch = (float*)(members + T);
    hipLaunchKernelGGL(members_kernel, dim3(1), dim3(256), 0, st, assign, T, K, offs, members);
    if (dtype == COGS_DT_BF16)
        hipLaunchKernelGGL(update_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, (const bf16_t*)feats, T, PD, K, offs, members,
                           reseed_rows, centres, shift_partial);
    else
        hipLaunchKernelGGL(update_kernel<float>, dim3(nblk), dim3(256), 0, st, (const float*)feats, T, PD, K, offs, members,
                           reseed_rows, centres, shift_partial);
    hipLaunchKernelGGL(update_final_kernel, dim3(1), dim3(256), 0, st, ts, K, offs, members, reseed_rows, centre_ts,
                       shift_partial, nblk, scratch, shift_out);
    return COGS_LAUNCH_CHECK();
}
