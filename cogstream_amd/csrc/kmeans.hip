// Time-aware k-means device steps (model/kmeans_with_time.py:4-137). The host keeps the reference's RNG draws
// (random.randint / torch.multinomial on the CPU generator); everything that touches the features runs here.
//
//   sqdist        partial[slice][t][k] = sum over the slice's 512 columns of (x[t,j]-c[k,j])^2   (:48,:73  torch.cdist)
//   reduce+assign dist2 = sum of the slice partials (fp64), per-row min-max of feature/time distances,
//                 sqrt(nf^2 + alpha nt^2), argmin                                                   (:76-104)
//   members       member lists per cluster, empty clusters -> reseed rows from the host-drawn pool    (:107-120)
//   update        per-cluster means (or the reseed row), centre shift                                (:107-125)
//
// HBM-bound by design: the features [T, P*D] (bf16 or fp32, 92/183 MB at T = 256) are read exactly once by sqdist and
// once by update per Lloyd iteration, once per k-means++ step. Round 3 rebuilt the kernels around that bound (rocprofv3
// at [256, 179 200] bf16, K = 18: sqdist 349 -> see DESIGN.md; the first version reduced every (row, centre) pair
// across the wave on its own -- 12 cross-lane steps per pair -- and walked member lists one load at a time):
//   * sqdist keeps one accumulator per centre in registers and reduces all of them across the wave in ONE
//     transposing butterfly per row (v_permlane32_swap / v_permlane16_swap on register pairs, then DPP row sums):
//     2.5 instructions per centre instead of 12; the centre slice sits in LDS in a lane-major order that makes the
//     two ds_read_b128 per centre conflict free; rows are taken four at a time so four 16-byte loads per lane are in
//     flight.
//   * the slice partials are summed in fp64 by lanes that own a centre each (coalesced reads), in a fixed order, and
//     the assignment of the row is computed by the same workgroup (one launch instead of two plus a host round trip).
//   * update runs one workgroup per (column block, cluster) -- K times the parallelism -- with four member rows in
//     flight per thread.
// All sums run in a fixed order (reproducible run to run). Distances are DIRECT sums of (x - c)^2, fp32 within a
// 512-column slice, fp64 across slices: DESIGN.md section 2 (near-tie study) says why not |x|^2 + |c|^2 - 2 x.c.
#include "common.h"
#include "kernels.h"
#include "debug.h"
#include <stdlib.h>

namespace {

// control words of the in-library Lloyd loop (device ints, in the workspace). The host queues SEVERAL iterations between
// two reads; every kernel of an iteration returns at once when the loop has ended (converged / pool exhausted), so the
// speculatively queued ones change nothing.
enum { CTL_USED = 0,      // reseed-pool entries consumed so far
       CTL_ABORT = 1,     // an iteration wanted more reseeds than the pool holds: it was not committed, the loop stops
       CTL_EMPTY = 2,     // empty clusters of the last iteration
       CTL_SHIFT = 3,     // centre shift of the last committed iteration (float bits)
       CTL_DONE = 4,      // shift <= tol: converged
       CTL_ITERS = 5,     // iterations committed
       CTL_MARGIN = 6,    // bits(+inf) - bits(min over rows and iterations of the decision margin) (reduce_assign_kernel)
       CTL_NEAR = 7,      // rows (summed over iterations) whose margin is below NEAR_TIE
       CTL_WORDS = 8 };
constexpr float NEAR_TIE = 1e-3f;   // below this the reference's own sgemm rounding decides a row

constexpr int SL = 512;       // columns per slice
constexpr int KMAX = 32;      // centres per sqdist launch (K itself is unbounded: 18 for 256 frames, 40 for 600)
constexpr int UB = 2048;      // columns per update workgroup (256 threads x 8)

__device__ __forceinline__ float swap32_sum(float a, float b) {
    // lanes 0..31: a[i] + a[i+32]; lanes 32..63: b[i-32] + b[i]
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float swap16_sum(float a, float b) {
    // 16-lane rows 0, 2: a's row pair summed; rows 1, 3: b's row pair summed
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
// sum over each 16-lane row, result in every lane of the row: four DPP adds (quad swaps, then the two mirrors). The
// __shfl_xor form compiles to four DEPENDENT ds_bpermute round trips -- ~400 exposed cycles per call, more than the
// arithmetic of the unit it closes (rocprofv3: 102 -> see DESIGN.md us per K = 18 pass).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_mov<0x141>(v);     // row_half_mirror: the other quad of the 8-lane half
    v += dpp_mov<0x140>(v);     // row_mirror: the other half of the 16-lane row
    return v;
}

// grid = (slices, row groups); 4 waves; LDS holds the centre slice as [k][half][lane][4] fp32 (half h, lane l = columns
// 8l+4h..+3) for this launch's K <= KMAX centres, padded with zero centres to a multiple of 4 (never stored).
// centre_row >= 0: the single centre is feature row `centre_row` (k-means++ step), else centre_rows / centres.
template <typename T>
__global__ __launch_bounds__(256, 4) void sqdist_kernel(const T* __restrict__ x, int Tn, long PD,
                                                        const float* __restrict__ centres,
                                                        const int* __restrict__ centre_rows, int centre_row, int K, int k0,
                                                        int Ktot, float* __restrict__ partial, const int* __restrict__ ctl) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* cs = reinterpret_cast<float*>(smem_raw);
    if (ctl && (ctl[CTL_ABORT] | ctl[CTL_DONE])) return;   // speculatively queued iteration behind the last one
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long j0 = (long)blockIdx.x * SL;
    const int KB = (K + 3) & ~3;
    for (int i = tid; i < KB * 64; i += 256) {
        const int k = i >> 6, l = i & 63;
        const long j = j0 + l * 8;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (k < K && j < PD) {
            if (centre_row >= 0) ld8_f<T>(x + (long)centre_row * PD + j, v);
            else if (centre_rows) ld8_f<T>(x + (long)centre_rows[k0 + k] * PD + j, v);
            else ld8_f<float>(centres + (long)(k0 + k) * PD + j, v);
        }
        *reinterpret_cast<f32x4*>(cs + k * SL + l * 4) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(cs + k * SL + 256 + l * 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
    __syncthreads();
    const long j = j0 + lane * 8;
    const bool in = j < PD;
    // Two rows per wave and step (rows t, t + 4: a centre's two LDS reads serve both), the NEXT step's global loads issued
    // before this step's arithmetic, and the centres taken FOUR at a time in a real loop: four accumulators per row are
    // exactly one unit of the transposing butterfly below, so nothing outlives a trip and the kernel needs ~80 VGPRs
    // (four workgroups per CU). History: one wave reduction per (row, centre) pair -- 12 cross-lane steps each -- 349 us
    // per pass at [256, 179 200] bf16, K = 18; all centres unrolled with one accumulator each: 147 us, and every
    // attempt to share the LDS reads between rows in that form made hipcc issue all the centre reads first (96-160
    // registers, 150 spilled or one wave per SIMD: 257 us).
    // grid.y row groups (a slice alone is 350 workgroups at cfg3 -- too few to hide the load latency on 256 CUs):
    // this workgroup takes rows [r_lo, r_hi)
    const int rows_per = (Tn + (int)gridDim.y - 1) / (int)gridDim.y;
    const int r_lo = (int)blockIdx.y * rows_per, r_hi = min(Tn, r_lo + rows_per);
    const int row = lane >> 4;
    const int kperm = ((row & 1) << 1) + (row >> 1);       // 16-lane row -> centre of a unit: 0, 2, 1, 3
    const bool writer = (lane & 15) == 0;
    const float* cl = cs + lane * 4;
    float xn[2][8];
    auto load_rows = [&](int t) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int e = 0; e < 8; ++e) xn[r][e] = 0.f;
            if (in && t + 4 * r < r_hi) ld8_f<T>(x + (long)(t + 4 * r) * PD + j, xn[r]);
        }
    };
    load_rows(r_lo + wid);
    for (int t = r_lo + wid; t < r_hi; t += 8) {
        f32x2 xab[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xab[e] = f32x2{xn[0][e], xn[1][e]};
        if (t + 8 < r_hi) load_rows(t + 8);
        const bool live_b = t + 4 < r_hi;
        float* pa = partial + ((long)blockIdx.x * Tn + t) * Ktot + k0 + kperm;
        float* pb = pa + 4 * (long)Ktot;
#pragma unroll 1
        for (int m = 0; m < KB; m += 4) {
            float a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(cl + (m + q) * SL);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(cl + (m + q) * SL + 256);
                // the two rows ride in the two halves of a register pair: v_pk_add_f32 / v_pk_fma_f32 do both rows'
                // subtraction and fma in one instruction each. Every accumulator still sees its own products in the same
                // order (column e of the first half, then of the second), so the sums are bit-identical to the scalar form
                f32x2 s = {0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 d0 = xab[e] - f32x2{c0[e], c0[e]};
                    const f32x2 d1 = xab[4 + e] - f32x2{c1[e], c1[e]};
                    s = __builtin_elementwise_fma(d0, d0, s);
                    s = __builtin_elementwise_fma(d1, d1, s);
                }
                a[q] = s[0];
                b[q] = s[1];
            }
            // transposing butterfly of one unit: 4 values x 64 lanes -> every 16-lane row holds the wave total of ONE
            // centre: row 0 -> m, row 1 -> m + 2, row 2 -> m + 1, row 3 -> m + 3
            const float ua = row16_sum(swap16_sum(swap32_sum(a[0], a[1]), swap32_sum(a[2], a[3])));
            const float ub = row16_sum(swap16_sum(swap32_sum(b[0], b[1]), swap32_sum(b[2], b[3])));
            if (writer && m + kperm < K) {
                pa[m] = ua;
                if (live_b) pb[m] = ub;
            }
        }
    }
}

// fp64 sum over the slices of partial[slice][t][k] for row t, by the lanes of one workgroup: lane (g, k) owns centre k
// and slice group g; groups are combined in a fixed order. Result in d2s[k] (LDS, fp32 like torch.cdist's output).
__device__ __forceinline__ void reduce_row(const float* __restrict__ partial, int nslices, int Tn, int K, int t,
                                           float* d2s, double* scratch /* [groups][kp]: 256 doubles */) {
    const int tid = threadIdx.x;
    for (int kc = 0; kc < K; kc += 64) {
        const int Kc = min(64, K - kc);
        int kp = 1;
        while (kp < Kc) kp <<= 1;                 // lanes per slice group (power of two <= 64)
        const int G = 256 / kp;                   // slice groups in the workgroup
        const int k = tid & (kp - 1), g = tid / kp;
        double s = 0.0;
        if (k < Kc) {
            const float* pp = partial + (long)t * K + kc + k;
            const long stride = (long)Tn * K;
            int sl = g;
            for (; sl + 7 * G < nslices; sl += 8 * G) {       // eight loads in flight, added in slice order
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = pp[(long)(sl + u * G) * stride];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += (double)v[u];
            }
            for (; sl < nslices; sl += G) s += (double)pp[(long)sl * stride];
        }
        if (k < Kc) scratch[g * kp + k] = s;
        __syncthreads();
        if (tid < Kc) {
            double tot = 0.0;
            for (int gg = 0; gg < G; ++gg) tot += scratch[gg * kp + tid];
            d2s[kc + tid] = (float)tot;
        }
        __syncthreads();
    }
}

// one workgroup per row t: slice reduction + the row's assignment (kmeans_with_time.py:76-104). dist2_out nullable.
// The arithmetic is torch's, operation by operation in fp32 (no contraction): sqrt, |.|, per-row min/max over the
// clusters, (d - min) / (max - min) or 0, sqrt(nf*nf + alpha*(nt*nt)), first minimum wins.
__global__ __launch_bounds__(256) void reduce_assign_kernel(const float* __restrict__ partial, int nslices, int Tn, int K,
                                                            const float* __restrict__ ts, const float* __restrict__ cts,
                                                            float alpha, float* __restrict__ dist2_out,
                                                            int64_t* __restrict__ assign, int* __restrict__ ctl) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* scratch = reinterpret_cast<double*>(smem_raw);           // 256 doubles
    if (ctl && (ctl[CTL_ABORT] | ctl[CTL_DONE])) return;
    float* d2s = reinterpret_cast<float*>(scratch + 256);            // [K]
    const int t = blockIdx.x;
    reduce_row(partial, nslices, Tn, K, t, d2s, scratch);
    const int tid = threadIdx.x;
    if (dist2_out)
        for (int k = tid; k < K; k += 256) dist2_out[(long)t * K + k] = d2s[k];
    if (!assign) return;
    if (tid >= 64) return;                        // one wave finishes the row
    const int lane = tid;
    const float tt = ts[t];
    float fmin_ = INFINITY, fmax_ = -INFINITY, tmin_ = INFINITY, tmax_ = -INFINITY;
    for (int k = lane; k < K; k += 64) {
        const float df = sqrtf(d2s[k]);
        const float dt = fabsf(tt - cts[k]);
        fmin_ = fminf(fmin_, df); fmax_ = fmaxf(fmax_, df);
        tmin_ = fminf(tmin_, dt); tmax_ = fmaxf(tmax_, dt);
    }
    fmin_ = wave_min(fmin_); fmax_ = wave_max(fmax_);
    tmin_ = wave_min(tmin_); tmax_ = wave_max(tmax_);
    float best = INFINITY, second = INFINITY;
    int bk = 0x7fffffff, sk = 0x7fffffff;
    auto final_dist = [&](int k, float& df, float& nf) {
        df = sqrtf(d2s[k]);
        const float dt = fabsf(tt - cts[k]);
        nf = fmax_ > fmin_ ? __fdiv_rn(__fsub_rn(df, fmin_), __fsub_rn(fmax_, fmin_)) : 0.f;
        const float nt = tmax_ > tmin_ ? __fdiv_rn(__fsub_rn(dt, tmin_), __fsub_rn(tmax_, tmin_)) : 0.f;
        return sqrtf(__fadd_rn(__fmul_rn(nf, nf), __fmul_rn(alpha, __fmul_rn(nt, nt))));
    };
    for (int k = lane; k < K; k += 64) {
        float df, nf;
        const float fd = final_dist(k, df, nf);
        if (fd < best) { second = best; sk = bk; best = fd; bk = k; }     // ascending k per lane: the lane keeps its FIRST minimum
        else if (fd < second) { second = fd; sk = k; }
    }
    const float wbest = wave_min(best);
    // first minimum overall = smallest k among the lanes that hold the minimum value
    int cand = (best == wbest) ? bk : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    if (lane == 0) assign[t] = cand == 0x7fffffff ? 0 : cand;        // all-NaN row: cluster 0, like a serial first-minimum scan
    if (ctl && K > 1 && cand != 0x7fffffff) {
        // How close was the decision? Runner-up = the smallest final distance of any OTHER cluster. The margin is
        // expressed in the unit the near-tie study uses: the RELATIVE change of the feature
        // distances that would flip the order. A relative change e of d_k moves nf_k by d_k e / R (R = max - min of
        // the row) and the final distance sqrt(nf^2 + alpha nt^2) by (nf_k / fd_k) d_k e / R; with the winner moving up
        // and the runner-up down the gap closes at e* = gap / (s_1 + s_2); reported as 2 e*, which for a pure feature
        // decision (all time terms equal) is (d_2 - d_1) / mean(d_1, d_2), the study's margin. A decision the feature
        // distances cannot flip (s_1 = s_2 = 0) reports +inf.
        const float oval = bk == cand ? second : best;
        const int okk = bk == cand ? sk : bk;
        const float wsecond = wave_min(oval);
        int cand2 = (oval == wsecond) ? okk : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand2 = min(cand2, __shfl_xor(cand2, o, 64));
        if (lane == 0 && cand2 != 0x7fffffff) {
            float d1, n1, d2, n2;
            const float f1 = final_dist(cand, d1, n1), f2 = final_dist(cand2, d2, n2);
            const float R = fmax_ - fmin_;
            float m = INFINITY;
            if (R > 0.f) {
                const float s1 = (f1 > 0.f ? n1 / f1 : 1.f) * d1 / R, s2 = (f2 > 0.f ? n2 / f2 : 1.f) * d2 / R;
                if (s1 + s2 > 0.f) m = 2.f * (f2 - f1) / (s1 + s2);
            }
            if (!(m == m) || m < 0.f) m = 0.f;
            // non-negative floats order like their bits; kept as the MAXIMUM of (bits(+inf) - bits(m)) so that the
            // all-zero control block the loop starts from means "margin = +inf"
            atomicMax(reinterpret_cast<unsigned*>(ctl + CTL_MARGIN), 0x7f800000u - __float_as_uint(m));
            if (m < NEAR_TIE) atomicAdd(ctl + CTL_NEAR, 1);
        }
    }
}

// k-means++ draw (:53-60) on the device: probs = (sqrt(nearest2))^2 as the reference computes them from cdist's distances,
// normalised by their sum; torch.multinomial(probs, 1) on the CPU IS argmax(probs / q), q ~ Exponential(1) one draw per
// element from the CPU generator (ATen multinomial, n_sample == 1) -- the host hands those draws over. A zero sum (every
// row coincides with a chosen centre) is the reference's random.randint branch: flagged, the host then takes the step-wise
// path. One workgroup; the sum runs in fp64 in a fixed order (torch's own fp32 summation order is not reproduced: the sum
// is a common divisor of all ratios and only moves the argmax between ratios closer than a few ulps).
__global__ __launch_bounds__(256) void kpp_select_kernel(const float* __restrict__ nearest2, const float* __restrict__ q, int Tn,
                                                         int* __restrict__ idx_out, int* __restrict__ zero_flag, int first_row) {
    __shared__ double red[4];
    __shared__ float rv[4];
    __shared__ int ri[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (first_row >= 0 && tid == 0) idx_out[-1] = first_row;      // step 1 also records the caller's first centre
    double sum = 0.0;
    for (int t = tid; t < Tn; t += 256) {
        const float d = sqrtf(nearest2[t]);
        sum += (double)__fmul_rn(d, d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) red[wid] = sum;
    __syncthreads();
    const float total = (float)((red[0] + red[1]) + (red[2] + red[3]));
    if (total == 0.f) {
        if (tid == 0) { *zero_flag = 1; *idx_out = 0; }
        return;
    }
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int t = tid; t < Tn; t += 256) {                 // ascending t per thread: the first maximum is kept
        const float d = sqrtf(nearest2[t]);
        const float r = __fdiv_rn(__fdiv_rn(__fmul_rn(d, d), total), q[t]);
        if (r > best || (r != r && best == best)) { best = r; bi = t; }      // NaN counts as the maximum, like torch.argmax
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        const bool take = (ob > best) || (ob != ob && best == best) || ((ob == best || (ob != ob && best != best)) && oi < bi);
        if (take) { best = ob; bi = oi; }
    }
    if (lane == 0) { rv[wid] = best; ri[wid] = bi; }
    __syncthreads();
    if (tid == 0) {
        float b = rv[0]; int i = ri[0];
        for (int w = 1; w < 4; ++w) {
            const float ob = rv[w]; const int oi = ri[w];
            const bool take = (ob > b) || (ob != ob && b == b) || ((ob == b || (ob != ob && b != b)) && oi < i);
            if (take) { b = ob; i = oi; }
        }
        *idx_out = i == 0x7fffffff ? 0 : i;
    }
}

// k-means++ step (:46-60): nearest2[t] = min(nearest2[t], dist2 to the newest centre) (first: plain store)
__global__ __launch_bounds__(256) void reduce_min_kernel(const float* __restrict__ partial, int nslices, int Tn, int first,
                                                         float* __restrict__ nearest2) {
    // one wave per row: lanes over slices, fp64 butterfly
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= Tn) return;
    double s = 0.0;
    for (int sl = lane; sl < nslices; sl += 64) s += (double)partial[(long)sl * Tn + t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        const float d2 = (float)s;
        nearest2[t] = first ? d2 : fminf(nearest2[t], d2);
    }
}

// ctl words (device ints): [0] reseeds used so far, [1] abort (pool exhausted), [2] empty clusters of this iteration
// One workgroup. Member lists (rows of every cluster in ascending order): offs [K+1], members [T]. Empty cluster k gets
// reseed_rows[k] = pool[used + (empty clusters below k)] -- the reference draws one random row per empty cluster in
// ascending k (:116-120) -- unless `reseed_in` is given (host-chosen rows, the step-wise API). Tn <= MEMB_T rows of
// assignments are staged in LDS.
constexpr int MEMB_T = 16384;
__global__ __launch_bounds__(1024) void members_kernel(const int64_t* __restrict__ assign, int Tn, int K,
                                                       int* __restrict__ offs, int* __restrict__ members,
                                                       const int* __restrict__ pool, int pool_len,
                                                       const int* __restrict__ reseed_in, int* __restrict__ reseed_rows,
                                                       int* __restrict__ ctl) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int* a = reinterpret_cast<int*>(smem_raw);        // [Tn]
    int* cnt = a + Tn;                                // [K + 1]
    const int tid = threadIdx.x, nt = blockDim.x;
    if (ctl && (ctl[CTL_ABORT] | ctl[CTL_DONE])) return;
    for (int k = tid; k <= K; k += nt) cnt[k] = 0;
    __syncthreads();
    for (int t = tid; t < Tn; t += nt) {
        int c = (int)assign[t];
        c = c < 0 ? 0 : (c >= K ? K - 1 : c);       // a foreign assignment vector must not index outside the tables
        a[t] = c;
        atomicAdd(&cnt[c], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int o = 0, ne = 0;
        const int used = ctl ? ctl[CTL_USED] : 0;
        offs[0] = 0;
        for (int k = 0; k < K; ++k) {
            const int n = cnt[k];
            if (n == 0) {
                if (reseed_in) reseed_rows[k] = reseed_in[k];
                else if (pool && used + ne < pool_len) reseed_rows[k] = pool[used + ne];
                else reseed_rows[k] = 0;
                ++ne;
            }
            cnt[k] = o;            // becomes the cluster's write cursor base
            o += n;
            offs[k + 1] = o;
        }
        if (ctl) {
            ctl[CTL_EMPTY] = ne;
            if (!reseed_in && ne > 0) {
                if (!pool || used + ne > pool_len) ctl[CTL_ABORT] = 1;      // abort: this iteration must not commit
                else ctl[CTL_USED] = used + ne;
            }
        }
    }
    __syncthreads();
    // rank of row t inside its cluster = rows t' < t of the same cluster (LDS broadcast reads)
    for (int t = tid; t < Tn; t += nt) {
        const int c = a[t];
        int r = 0;
        for (int u = 0; u < t; ++u) r += (a[u] == c) ? 1 : 0;
        members[cnt[c] + r] = t;
    }
}

// fallback for Tn > MEMB_T (no LDS staging): thread k walks the assignment vector
__global__ __launch_bounds__(256) void members_slow_kernel(const int64_t* __restrict__ assign, int Tn, int K,
                                                           int* __restrict__ offs, int* __restrict__ members) {
    for (int k = threadIdx.x; k < K; k += 256) {
        int n = 0;
        for (int t = 0; t < Tn; ++t) n += ((int)assign[t] == k) ? 1 : 0;
        offs[k + 1] = n;
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

// grid (column blocks, K): the mean of cluster k over this block's 2048 columns (members walked in ascending order, four
// loads in flight, summed in order), or the reseed row when the cluster is empty; squared shift of the block
template <typename T>
__global__ __launch_bounds__(256) void update_kernel(const T* __restrict__ x, long PD, int K,
                                                     const int* __restrict__ offs, const int* __restrict__ members,
                                                     const int* __restrict__ reseed_rows, const int* __restrict__ ctl,
                                                     float* __restrict__ centres, float* __restrict__ shift_partial) {
    __shared__ float red[4];
    if (ctl && (ctl[CTL_ABORT] | ctl[CTL_DONE])) return;   // pool exhausted: nothing of this iteration is committed; or the loop has ended
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int k = blockIdx.y;
    const long col = ((long)blockIdx.x * 256 + tid) * 8;
    const bool in = col < PD;
    float sh = 0.f;
    if (in) {
        const int b = offs[k], e = offs[k + 1];
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (e > b) {
            int i = b;
            for (; i + 4 <= e; i += 4) {
                float v[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) ld8_f<T>(x + (long)members[i + u] * PD + col, v[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c] += v[u][c];
            }
            for (; i < e; ++i) {
                float v[8];
                ld8_f<T>(x + (long)members[i] * PD + col, v);
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] += v[c];
            }
            const float n = (float)(e - b);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = acc[c] / n;
        } else {
            ld8_f<T>(x + (long)reseed_rows[k] * PD + col, acc);
        }
        float* cp = centres + (long)k * PD + col;
        float old[8];
        ld8_f<float>(cp, old);
#pragma unroll
        for (int c = 0; c < 8; ++c) { const float d = acc[c] - old[c]; sh = fmaf(d, d, sh); }
        st8_f<float>(cp, acc);
    }
    sh = wave_sum(sh);
    if (lane == 0) red[wid] = sh;
    __syncthreads();
    if (tid == 0) shift_partial[(long)blockIdx.x * K + k] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup: centre times, shift norms (scratch [2K]: feature-shift norm, squared time shift), total movement.
// status (nullable, host-mapped): {shift, reseeds used, abort, empty clusters} for the Lloyd loop's one read per iteration
__global__ __launch_bounds__(256) void update_final_kernel(const float* __restrict__ ts, int K,
                                                           const int* __restrict__ offs, const int* __restrict__ members,
                                                           const int* __restrict__ reseed_rows, int* __restrict__ ctl,
                                                           float* __restrict__ cts,
                                                           const float* __restrict__ shift_partial, int nblk,
                                                           float* __restrict__ scratch, float* __restrict__ shift_out,
                                                           volatile int* __restrict__ status, int Tn_lds, float tol) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* mts = reinterpret_cast<float*>(smem_raw);      // [Tn] time stamps in member order (0 floats when Tn_lds == 0)
    if (ctl && ctl[CTL_DONE]) return;                     // a speculatively queued iteration behind the converged one
    const bool aborted = ctl && ctl[CTL_ABORT];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (!aborted) {
        if (Tn_lds > 0) {
            for (int i = threadIdx.x; i < Tn_lds; i += 256) mts[i] = ts[members[i]];     // one dependent pair per thread, in parallel
            __syncthreads();
        }
        for (int k = wid; k < K; k += 4) {                  // one wave per cluster
            double s = 0.0;
            for (int b = lane; b < nblk; b += 64) s += (double)shift_partial[(long)b * K + k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) {
                scratch[k] = sqrtf((float)s);
                float tsum = 0.f;
                const int b0 = offs[k], e0 = offs[k + 1];
                if (Tn_lds > 0) { for (int i = b0; i < e0; ++i) tsum += mts[i]; }
                else { for (int i = b0; i < e0; ++i) tsum += ts[members[i]]; }
                const float nt = e0 > b0 ? tsum / (float)(e0 - b0) : ts[reseed_rows[k]];
                const float d = nt - cts[k];
                scratch[K + k] = d * d;
                cts[k] = nt;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.f;
        if (!aborted) {
            float f = 0.f, tt = 0.f;
            for (int i = 0; i < K; ++i) { f += scratch[i]; tt += scratch[K + i]; }
            total = f + sqrtf(tt);
            shift_out[0] = total;
        }
        if (ctl && !aborted) {
            ctl[CTL_SHIFT] = __builtin_bit_cast(int, total);
            ctl[CTL_ITERS] += 1;
            if (total <= tol) ctl[CTL_DONE] = 1;
        }
        if (status) {      // host-mapped copy of the control words (one read per batch of iterations)
#pragma unroll
            for (int i = 0; i < CTL_WORDS; ++i) status[i] = ctl ? ctl[i] : 0;
            __threadfence_system();
        }
    }
}

// one thread per row t from dist2 (the step-wise API, cogs_kmeans_assign); counts must be zero on entry
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
        const float nf = fmax_ > fmin_ ? __fdiv_rn(__fsub_rn(df, fmin_), __fsub_rn(fmax_, fmin_)) : 0.f;
        const float nt = tmax_ > tmin_ ? __fdiv_rn(__fsub_rn(dt, tmin_), __fsub_rn(tmax_, tmin_)) : 0.f;
        const float fd = sqrtf(__fadd_rn(__fmul_rn(nf, nf), __fmul_rn(alpha, __fmul_rn(nt, nt))));
        if (fd < best) { best = fd; bk = k; }
    }
    assign[t] = bk;
    atomicAdd(&counts[bk], 1);
}

// select_additional_frames (model/cogreasoner_chat.py:50-64): per cluster, all members if there are at most n, else the n
// members nearest to the centroid (smallest dist2, ties to the lower row; torch.topk(largest=False) order = ascending
// distance). One wave per cluster; picks [K][n] int64 (-1 padded), counts [K] int32. n <= 8.
__global__ __launch_bounds__(64) void select_near_kernel(const float* __restrict__ dist2, const int64_t* __restrict__ assign,
                                                         int Tn, int K, int n, int64_t* __restrict__ picks,
                                                         int* __restrict__ counts) {
    const int k = blockIdx.x, lane = threadIdx.x;
    // the lane's own candidates: its n best (distance, row) pairs, ascending
    float bd[8];
    int bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { bd[j] = INFINITY; bi[j] = 0x7fffffff; }
    int cnt = 0;
    for (int t = lane; t < Tn; t += 64) {
        if ((int)assign[t] != k) continue;
        ++cnt;
        float d = dist2[(long)t * K + k];
        int id = t;
#pragma unroll
        for (int j = 0; j < 8; ++j) {              // insertion into the sorted list (rows arrive in ascending order per lane)
            if (j < n && (d < bd[j] || (d == bd[j] && id < bi[j]))) {
                const float td = bd[j]; const int ti = bi[j];
                bd[j] = d; bi[j] = id; d = td; id = ti;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) counts[k] = cnt < n ? cnt : n;
    if (cnt <= n) {
        // all members, ascending row: rank of each member = members before it
        if (lane == 0)
            for (int j = cnt; j < n; ++j) picks[(long)k * n + j] = -1;        // the unused slots (never written below)
        for (int base = 0, rank = 0; base < Tn; base += 64) {
            const int t = base + lane;
            const bool mine = t < Tn && (int)assign[t] == k;
            const unsigned long long m = __ballot(mine);
            if (mine) picks[(long)k * n + rank + __popcll(m & ((1ull << lane) - 1ull))] = t;
            rank += __popcll(m);
        }
        return;
    }
    // n rounds: the wave's best head-of-list wins and is popped from its lane
    for (int r = 0; r < n; ++r) {
        float d = bd[0];
        int id = bi[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float od = __shfl_xor(d, o, 64);
            const int oi = __shfl_xor(id, o, 64);
            if (od < d || (od == d && oi < id)) { d = od; id = oi; }
        }
        if (lane == 0) picks[(long)k * n + r] = id;
        if (bi[0] == id) {                          // pop (rows are unique: exactly one lane holds the winner)
#pragma unroll
            for (int j = 0; j < 7; ++j) { bd[j] = bd[j + 1]; bi[j] = bi[j + 1]; }
            bd[7] = INFINITY; bi[7] = 0x7fffffff;
        }
    }
}

}  // namespace

int cogs_k_select_near(hipStream_t st, const float* dist2, const int64_t* assign, int T, int K, int n, int64_t* picks,
                       int* counts) {
    if (T <= 0 || K <= 0 || n <= 0 || n > 8) return COGS_E_INVALID;
    hipLaunchKernelGGL(select_near_kernel, dim3(K), dim3(64), 0, st, dist2, assign, T, K, n, picks, counts);
    return COGS_LAUNCH_CHECK();
}

// ---- workspace: [A] max(sqdist slice partials [ns][T][K], update shift partials [nblk][K]) floats, then the member
// lists (offs [K+1], members [T]), the update scratch [2K], and the Lloyd loop's own state: dist2 [T][K] floats,
// reseed rows [K], ctl [CTL_WORDS], reseed pool [POOL_MAX] ints
constexpr int POOL_MAX = 4096;
static size_t ws_floats(int T, long PD, int K, int ns) {
    const size_t a = (size_t)ns * T * K;
    const size_t b = (size_t)cogs_k_kmeans_update_blocks(PD) * K;
    return a > b ? a : b;
}
size_t cogs_k_kmeans_ws(int T, long PD, int K, int* nslices) {
    const int ns = (int)((PD + SL - 1) / SL);
    if (nslices) *nslices = ns;
    return (ws_floats(T, PD, K, ns) + (size_t)(K + 1) + (size_t)T + 2 * (size_t)K + (size_t)T * K + (size_t)K + CTL_WORDS + POOL_MAX) * 4;
}
namespace {
struct KmWs {
    float* partial; int* offs; int* members; float* scratch; float* dist2; int* reseed; int* ctl; int* pool;
};
KmWs carve(float* ws, int T, long PD, int K) {
    int ns = 0;
    (void)cogs_k_kmeans_ws(T, PD, K, &ns);
    KmWs w;
    w.partial = ws;
    w.offs = (int*)(ws + ws_floats(T, PD, K, ns));
    w.members = w.offs + K + 1;
    w.scratch = (float*)(w.members + T);
    w.dist2 = w.scratch + 2 * K;
    w.reseed = (int*)(w.dist2 + (size_t)T * K);
    w.ctl = w.reseed + K;
    w.pool = w.ctl + CTL_WORDS;
    return w;
}

template <typename T>
void launch_sqdist(hipStream_t st, const void* feats, int Tn, long PD, const float* centres, const int* centre_rows,
                   int centre_row, int kb, int k0, int K, float* partial, int nslices, const int* ctl) {
    static std::atomic<uint64_t> done{0};
    cogs_ensure_dyn_lds((const void*)sqdist_kernel<T>, KMAX * SL * 4, done);
    const int kp = (kb + 3) & ~3;
    // row groups on grid.y: as many as keep the whole grid RESIDENT at once (LDS: 2 KiB per centre and workgroup, at most
    // 8 workgroups per CU) -- a grid a few workgroups over that runs a second, nearly empty round (measured at K = 18:
    // 700 workgroups 84 us, 1 050 -> 103 us, 1 400 -> 80 us) -- with at least 16 rows each
    const int per_cu = 80 / kp < 1 ? 1 : (80 / kp > 8 ? 8 : 80 / kp);
    int rg = 256 * per_cu / nslices;
    const int env_rg = (int)g_cogs_debug.km_row_groups;     // experiments
    if (env_rg > 0) rg = env_rg;
    rg = rg < 1 ? 1 : (rg > 8 ? 8 : rg);
    while (rg > 1 && (Tn + rg - 1) / rg < 16) --rg;
    hipLaunchKernelGGL((sqdist_kernel<T>), dim3(nslices, rg), dim3(256), (size_t)kp * SL * 4, st, (const T*)feats, Tn, PD,
                       centres, centre_rows, centre_row, kb, k0, K, partial, ctl);
}
int sqdist_partials(hipStream_t st, int dtype, const void* feats, int T, long PD, const float* centres,
                    const int* centre_rows, int centre_row, int K, float* partial, int nslices, const int* ctl = nullptr) {
    if (K <= 0 || T <= 0 || PD % 8) return COGS_E_INVALID;
    if (nslices != (int)((PD + SL - 1) / SL)) return COGS_E_WORKSPACE;
    for (int k0 = 0; k0 < K; k0 += KMAX) {          // the LDS centre slice holds KMAX clusters at a time
        const int kb = K - k0 < KMAX ? K - k0 : KMAX;
        if (dtype == COGS_DT_BF16) launch_sqdist<bf16_t>(st, feats, T, PD, centres, centre_rows, centre_row, kb, k0, K, partial, nslices, ctl);
        else launch_sqdist<float>(st, feats, T, PD, centres, centre_rows, centre_row, kb, k0, K, partial, nslices, ctl);
    }
    return COGS_LAUNCH_CHECK();
}
int launch_reduce_assign(hipStream_t st, const float* partial, int nslices, int T, int K, const float* ts, const float* cts,
                         float alpha, float* dist2, int64_t* assign, int* ctl = nullptr) {
    if (K > 4096) return COGS_E_UNSUPPORTED;
    hipLaunchKernelGGL(reduce_assign_kernel, dim3(T), dim3(256), (size_t)256 * 8 + (size_t)K * 4, st, partial, nslices, T, K, ts, cts, alpha, dist2, assign, ctl);
    return COGS_LAUNCH_CHECK();
}
int launch_members(hipStream_t st, const int64_t* assign, int T, int K, const KmWs& w, const int* pool, int pool_len,
                   const int* reseed_in, int* ctl) {
    if (T <= MEMB_T && (size_t)(T + K + 1) * 4 <= 64 * 1024) {
        const int nt = T >= 1024 ? 1024 : ((T + 63) / 64) * 64;
        hipLaunchKernelGGL(members_kernel, dim3(1), dim3(nt < 64 ? 64 : nt), (size_t)(T + K + 1) * 4, st, assign, T, K, w.offs,
                           w.members, pool, pool_len, reseed_in, w.reseed, ctl);
    } else {
        if (!reseed_in) return COGS_E_UNSUPPORTED;     // the Lloyd loop needs the LDS path (T <= 16 384 rows)
        hipLaunchKernelGGL(members_slow_kernel, dim3(1), dim3(256), 0, st, assign, T, K, w.offs, w.members);
        if (hipMemcpyAsync(w.reseed, reseed_in, (size_t)K * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return COGS_E_HIP;
    }
    return COGS_LAUNCH_CHECK();
}
int launch_update(hipStream_t st, int dtype, const void* feats, const float* ts, int T, long PD, int K, const KmWs& w,
                  int* ctl, float* centres, float* centre_ts, float* shift_out, volatile int* status, float tol = -1.f) {
    const int nblk = cogs_k_kmeans_update_blocks(PD);
    if (dtype == COGS_DT_BF16)
        hipLaunchKernelGGL(update_kernel<bf16_t>, dim3(nblk, K), dim3(256), 0, st, (const bf16_t*)feats, PD, K, w.offs, w.members,
                           w.reseed, ctl, centres, w.partial);
    else
        hipLaunchKernelGGL(update_kernel<float>, dim3(nblk, K), dim3(256), 0, st, (const float*)feats, PD, K, w.offs, w.members,
                           w.reseed, ctl, centres, w.partial);
    const int t_lds = T <= MEMB_T ? T : 0;       // member-ordered time stamps staged in LDS (64 KiB at most)
    hipLaunchKernelGGL(update_final_kernel, dim3(1), dim3(256), (size_t)t_lds * 4, st, ts, K, w.offs, w.members, w.reseed, ctl,
                       centre_ts, w.partial, nblk, w.scratch, shift_out, status, t_lds, tol);
    return COGS_LAUNCH_CHECK();
}
}  // namespace

int cogs_k_kmeans_update_blocks(long PD) { return (int)((PD + UB - 1) / UB); }

int cogs_k_kmeans_sqdist(hipStream_t st, int dtype, const void* feats, int T, long PD, const float* centres,
                         const int* centre_rows, int K, float* ws, int nslices, float* dist2) {
    const KmWs w = carve(ws, T, PD, K);
    const int rc = sqdist_partials(st, dtype, feats, T, PD, centres, centre_rows, -1, K, w.partial, nslices);
    if (rc != COGS_OK) return rc;
    return launch_reduce_assign(st, w.partial, nslices, T, K, nullptr, nullptr, 0.f, dist2, nullptr);
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
    if (T <= 0 || K <= 0 || PD % 8) return COGS_E_INVALID;
    if (nblk != cogs_k_kmeans_update_blocks(PD)) return COGS_E_WORKSPACE;
    const KmWs w = carve(ws, T, PD, K);
    const int rc = launch_members(st, assign, T, K, w, nullptr, 0, reseed_rows, nullptr);
    if (rc != COGS_OK) return rc;
    return launch_update(st, dtype, feats, ts, T, PD, K, w, nullptr, centres, centre_ts, shift_out, nullptr);
}

// one k-means++ step: distance^2 of every row to feature row `row`, folded into nearest2 (device [T]); the updated
// nearest2 is copied to probs_host (pinned host memory, nullable) and the stream is synchronised when it is given
int cogs_k_kmeans_pp_step(hipStream_t st, int dtype, const void* feats, int T, long PD, int row, int first,
                          float* nearest2, float* probs_host, float* ws, int nslices) {
    if (row < 0 || row >= T) return COGS_E_INVALID;
    const KmWs w = carve(ws, T, PD, 1);
    const int rc = sqdist_partials(st, dtype, feats, T, PD, nullptr, nullptr, row, 1, w.partial, nslices);
    if (rc != COGS_OK) return rc;
    hipLaunchKernelGGL(reduce_min_kernel, dim3((T + 3) / 4), dim3(256), 0, st, w.partial, nslices, T, first, nearest2);
    if (COGS_LAUNCH_CHECK() != COGS_OK) return COGS_E_HIP;
    if (probs_host) {
        if (hipMemcpyAsync(probs_host, nearest2, (size_t)T * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return COGS_E_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) return COGS_E_HIP;
    }
    return COGS_OK;
}

// The whole k-means++ seeding (kmeans_with_time.py:41-62) in one call, no host round trip per centre: step m measures
// every row against the newest centre (row idx[m-1], read from device memory), folds it into nearest2 and draws centre m
// as argmax(probs / q[m-1]) (kpp_select_kernel). q: device fp32 [K-1][T], the Exponential(1) draws the reference's K-1
// torch.multinomial calls would make on the CPU generator, made ahead by the host. idx: device int32 [K], idx[0] given.
// *zero_flag (device int, zeroed here) is set when a step found all probabilities zero -- the reference's
// random.randint branch; the indices from that step on are then meaningless and the caller redoes the seeding step by step.
int cogs_k_kmeans_pp(hipStream_t st, int dtype, const void* feats, int T, long PD, int K, int first_row, const float* q, int* idx,
                     int* zero_flag, float* nearest2, float* ws, int nslices) {
    if (T <= 0 || K <= 0 || PD % 8 || !q || !idx || !zero_flag || !nearest2 || first_row < 0 || first_row >= T) return COGS_E_INVALID;
    const KmWs w = carve(ws, T, PD, 1);
    if (hipMemsetAsync(zero_flag, 0, sizeof(int), st) != hipSuccess) return COGS_E_HIP;
    for (int m = 1; m < K; ++m) {
        // step 1 measures against the caller's first centre (a kernel argument: nothing is uploaded), later steps against
        // the row the previous step drew (read from device memory)
        const int rc = m == 1 ? sqdist_partials(st, dtype, feats, T, PD, nullptr, nullptr, first_row, 1, w.partial, nslices)
                              : sqdist_partials(st, dtype, feats, T, PD, nullptr, idx + (m - 1), -1, 1, w.partial, nslices);
        if (rc != COGS_OK) return rc;
        hipLaunchKernelGGL(reduce_min_kernel, dim3((T + 3) / 4), dim3(256), 0, st, w.partial, nslices, T, m == 1 ? 1 : 0, nearest2);
        hipLaunchKernelGGL(kpp_select_kernel, dim3(1), dim3(256), 0, st, nearest2, q + (size_t)(m - 1) * T, T, idx + m, zero_flag,
                           m == 1 ? first_row : -1);
    }
    if (K == 1 && hipMemcpyAsync(idx, &first_row, sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) return COGS_E_HIP;
    return COGS_LAUNCH_CHECK();
}

// The Lloyd loop (kmeans_with_time.py:71-131). The host queues up to LLOYD_BATCH iterations between two reads of the
// control words (update_final copies them into host-mapped memory): every kernel of an iteration returns at once when an
// earlier one has converged or run the reseed pool dry, so the iterations queued behind the last one change nothing.
// Empty clusters take their rows from `pool`, the host's pre-drawn random.randint values, in the reference's order; if an
// iteration needs more than the pool holds it is not committed and the call returns with *exhausted = 1 (the caller
// draws more and calls again with the remaining budget).
constexpr int LLOYD_BATCH = 4;
namespace {
// host-mapped: [0, 64) the control words as update_final leaves them, [64, 64 + 4 * POOL_MAX) the reseed pool -- the
// device reads the pool in place (only when a cluster runs empty), so a call uploads nothing and frees the caller's
// array the moment it returns
struct StatusBlock { volatile int* h = nullptr; int* d = nullptr; int dev = -1; };
constexpr size_t STATUS_BYTES = 64 + (size_t)POOL_MAX * 4;
// one host-mapped block per calling thread; replaced (and the old one freed) when the thread's device changes
bool status_block(StatusBlock& b) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (b.h && b.dev == dev) return true;
    if (b.h) { (void)hipHostFree((void*)b.h); b.h = nullptr; b.d = nullptr; }
    void* hp = nullptr;
    if (hipHostMalloc(&hp, STATUS_BYTES, hipHostMallocMapped) != hipSuccess) return false;
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { (void)hipHostFree(hp); return false; }
    b.h = (volatile int*)hp; b.d = (int*)dp; b.dev = dev;
    return true;
}
}  // namespace

int cogs_k_kmeans_lloyd(hipStream_t st, int dtype, const void* feats, const float* ts, int T, long PD, int K, float alpha,
                        int max_iter, float tol, const int* pool_host, int pool_len, float* centres, float* centre_ts,
                        int64_t* assign, int* iterations, int* reseeds_used, int* exhausted, float* ws, int nslices) {
    if (T <= 0 || K <= 0 || PD % 8 || max_iter < 0 || pool_len < 0 || pool_len > POOL_MAX) return COGS_E_INVALID;
    if (T > MEMB_T || (size_t)(T + K + 1) * 4 > 64 * 1024 || K > 4096) return COGS_E_UNSUPPORTED;   // T + K + 1 <= 16 384
    static thread_local StatusBlock sb;
    if (!status_block(sb)) return COGS_E_HIP;
    if (iterations) *iterations = 0;
    if (reseeds_used) *reseeds_used = 0;
    if (exhausted) *exhausted = 0;
    if (max_iter == 0) return COGS_OK;             // nothing to do
    const KmWs w = carve(ws, T, PD, K);
    if (hipMemsetAsync(w.ctl, 0, CTL_WORDS * sizeof(int), st) != hipSuccess) return COGS_E_HIP;     // margin word 0 = +inf
    int* pool_d = sb.d + 16;
    for (int i = 0; i < pool_len; ++i) sb.h[16 + i] = pool_host[i];        // plain host stores into the mapped block
    for (int i = 0; i < CTL_WORDS; ++i) sb.h[i] = 0;
    int queued = 0, it = 0, used = 0, ex = 0;
    int rc = COGS_OK;
    while (queued < max_iter) {
        const int batch = max_iter - queued < LLOYD_BATCH ? max_iter - queued : LLOYD_BATCH;
        for (int b = 0; b < batch && rc == COGS_OK; ++b) {
            rc = sqdist_partials(st, dtype, feats, T, PD, centres, nullptr, -1, K, w.partial, nslices, w.ctl);
            if (rc == COGS_OK) rc = launch_reduce_assign(st, w.partial, nslices, T, K, ts, centre_ts, alpha, nullptr, assign, w.ctl);
            if (rc == COGS_OK) rc = launch_members(st, assign, T, K, w, pool_d, pool_len, nullptr, w.ctl);
            if (rc == COGS_OK) rc = launch_update(st, dtype, feats, ts, T, PD, K, w, w.ctl, centres, centre_ts, (float*)(w.ctl + CTL_SHIFT),
                                                  (volatile int*)sb.d, tol);
        }
        queued += batch;
        if (hipStreamSynchronize(st) != hipSuccess) return COGS_E_HIP;
        if (rc != COGS_OK) return rc;
        it = sb.h[CTL_ITERS];
        used = sb.h[CTL_USED];
        if (sb.h[CTL_ABORT]) { ex = 1; break; }      // the aborted iteration was not committed
        if (sb.h[CTL_DONE]) break;
    }
    if (iterations) *iterations = it;
    if (reseeds_used) *reseeds_used = used;
    if (exhausted) *exhausted = ex;
    return COGS_OK;
}

// how close the assignments of the last cogs_k_kmeans_lloyd call on this workspace were: the smallest relative margin
// (second-best - best) / best of the final distance over all rows and iterations, and the number of (row, iteration)
// pairs below 1e-3 -- below that the reference's own cdist rounding decides the row
int cogs_k_kmeans_margins(hipStream_t st, int T, long PD, int K, float* ws, float* min_margin, int* rows_below) {
    const KmWs w = carve(ws, T, PD, K);
    int h[CTL_WORDS];
    if (hipMemcpyAsync(h, w.ctl, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess) return COGS_E_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return COGS_E_HIP;
    if (min_margin) *min_margin = __builtin_bit_cast(float, 0x7f800000 - h[CTL_MARGIN]);
    if (rows_below) *rows_below = h[CTL_NEAR];
    return COGS_OK;
}
