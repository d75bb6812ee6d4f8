// Sampled decoding on the device: TopKLogitsWarper -> TopPLogitsWarper -> softmax -> multinomial of one fp32
// logits row, no host round trip.
//
// Replaces (third-party, transformers==4.46.3 as pinned by environment.yml:404) GenerationMixin._sample's
//   TopKLogitsWarper (keep scores >= the k-th largest), TopPLogitsWarper (sort ascending, drop the tail whose
//   cumulative probability is <= 1 - top_p, always keep the best token), softmax, torch.multinomial(probs, 1)
// with the reference's shipped settings model/generation_config.json:2-12 (top_k 20, top_p 0.8). Repetition
// penalty, the allowed-id mask and the temperature run before this, in cogs_logits_process (llm_misc.hip).
//
// torch.multinomial(probs, 1) on the CPU -- the reference's parity path -- is argmax_i(probs_i / q_i) with
// q ~ Exponential(1) drawn for EVERY vocabulary entry from the generator. `draws` hands those [n] host draws
// over (parity mode: the sampled ids then equal the reference's for the same generator state); with draws ==
// NULL the kernel makes its own q_i = -log(u_i), u_i from Philox4x32-10 keyed by (seed, offset, i).
//
// Two paths: 0 < top_k <= 64 (the shipped default): 64 workgroups each reduce a slice of the row to its top_k
// candidates, one workgroup merges 64 x top_k candidates and samples (2 launches, ~10 us). Otherwise (no top-k,
// or top_k > 64): one workgroup works on the whole row with radix selects (exact for any k and any top_p).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int SA_BLOCKS = 64;    // slices of the row in the fast path
constexpr int SA_SLICE = 4096;   // max entries per slice (n <= 262144)
constexpr int SA_KMAX = 64;

__device__ __forceinline__ uint32_t fkey(float s) {   // order-preserving float -> uint
    const uint32_t b = __float_as_uint(s);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Philox4x32-10 (Salmon et al. 2011), counter = (i, offset_lo, offset_hi, 0), key = seed
__device__ __forceinline__ uint32_t philox_u32(uint64_t seed, uint64_t offset, uint32_t i) {
    uint32_t c0 = i, c1 = (uint32_t)offset, c2 = (uint32_t)(offset >> 32), c3 = 0;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}
__device__ __forceinline__ float draw_q(const float* __restrict__ draws, uint64_t seed, uint64_t offset, int i) {
    if (draws) return draws[i];
    const float u = ((float)(philox_u32(seed, offset, (uint32_t)i) >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
    return -logf(u);
}

struct Best {
    uint32_t key;
    int idx;
};
__device__ __forceinline__ bool better(uint32_t ka, int ia, uint32_t kb, int ib) {   // a before b (desc key, asc idx)
    return ka > kb || (ka == kb && ia < ib);
}
__device__ __forceinline__ Best wave_best(Best b) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t ok = __shfl_xor(b.key, o, 64);
        const int oi = __shfl_xor(b.idx, o, 64);
        if (better(ok, oi, b.key, b.idx)) { b.key = ok; b.idx = oi; }
    }
    return b;
}

// ---- fast path, stage A: per-slice top-k candidates (key desc, index asc) ----
__global__ __launch_bounds__(256) void sample_slice_topk_kernel(const float* __restrict__ logits, int n, int k, float temperature,
                                                                uint32_t* __restrict__ cand_key, int* __restrict__ cand_idx) {
    __shared__ uint32_t sl[SA_SLICE];
    __shared__ uint32_t wk[4];
    __shared__ int wi[4];
    const int chunk = (n + SA_BLOCKS - 1) / SA_BLOCKS;
    const int lo = blockIdx.x * chunk, cnt = max(0, min(chunk, n - lo));
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < cnt; i += 256) {
        const float v = logits[lo + i] / temperature;   // TemperatureLogitsWarper: scores / temperature (IEEE division)
        sl[i] = v == -INFINITY ? 0u : fkey(v);   // a masked score (-inf) is never a candidate
    }
    __syncthreads();
    for (int r = 0; r < k; ++r) {
        Best b{0u, 0x7fffffff};
        for (int i = tid; i < cnt; i += 256) {
            const uint32_t v = sl[i];
            if (better(v, i, b.key, b.idx)) { b.key = v; b.idx = i; }
        }
        b = wave_best(b);
        if (lane == 0) { wk[wid] = b.key; wi[wid] = b.idx; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (better(wk[w], wi[w], b.key, b.idx)) { b.key = wk[w]; b.idx = wi[w]; }
            const bool ok = b.key != 0u && b.idx < cnt;
            cand_key[blockIdx.x * SA_KMAX + r] = ok ? b.key : 0u;          // key 0 = "no candidate"
            cand_idx[blockIdx.x * SA_KMAX + r] = ok ? lo + b.idx : 0x7fffffff;
            if (ok) sl[b.idx] = 0u;
        }
        __syncthreads();
    }
}

#ifdef COGS_SAMPLE_STAMPS     // diagnostic build (tools/micro/sample_micro.cpp): s_memtime at the phase boundaries of the merge
__device__ unsigned long long g_sample_stamps[8];
#define SSTAMP(i_) do { if (threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_sample_stamps[i_] = t_; } } while (0)
#else
#define SSTAMP(i_) do {} while (0)
#endif
// ---- fast path, stage B: merge the candidates, top-p, sample ----
__global__ __launch_bounds__(1024) void sample_merge_kernel(const uint32_t* __restrict__ cand_key, const int* __restrict__ cand_idx,
                                                            int k, float lim, const float* __restrict__ draws, uint64_t seed,
                                                            uint64_t offset, int64_t* __restrict__ out_token,
                                                            int32_t* __restrict__ kept_idx, float* __restrict__ kept_prob,
                                                            int* __restrict__ n_kept, int kept_cap) {
    __shared__ __attribute__((aligned(16))) uint32_t ck[SA_BLOCKS * SA_KMAX];
    __shared__ __attribute__((aligned(16))) int ci[SA_BLOCKS * SA_KMAX];
    constexpr int KEPT_MAX = 1024;                  // top_k + ties of the k-th value (more ties than that are dropped)
    __shared__ uint32_t sk[KEPT_MAX];               // kept, sorted: rank < k first, then ties with the k-th value
    __shared__ int si[KEPT_MAX];
    __shared__ int m_sh, ties_sh;
    __shared__ uint32_t t0_sh;
    __shared__ int nsurv_sh;
    const int tid = threadIdx.x;
    SSTAMP(0);
    // prefilter: the slice whose k-th candidate is largest already holds k keys >= that value T0, so the global k-th
    // largest key is >= T0 and every candidate below T0 is out. What is left (k .. a few k) is ranked; ranking all
    // 64 * k candidates against each other was 100+ us of LDS scanning per token.
    if (tid < 64) {
        uint32_t kth = cand_key[tid * SA_KMAX + k - 1];                 // 0 when the slice has fewer than k candidates
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kth = max(kth, (uint32_t)__shfl_xor((int)kth, o, 64));
        // a second certified lower bound on the k-th largest key: the k-th largest of the 64 slice maxima (k slices
        // hold a candidate >= it). On flat rows (and on i.i.d. test data) every slice's k-th candidate is about the same,
        // T0 alone keeps 700+ of the 1 280 candidates and the rank scan below took 45 + 35 us; the two bounds complement
        // each other (peaked rows: the top-k sit in few slices, whose k-th candidates are high).
        uint32_t v0 = 0u;
        if (k <= SA_BLOCKS) {
            const uint32_t mine = cand_key[tid * SA_KMAX];
            int rank = 0;
            for (int i = 0; i < 64; ++i) {
                const uint32_t o = (uint32_t)__shfl((int)mine, i, 64);
                rank += (o > mine || (o == mine && i < tid)) ? 1 : 0;
            }
            v0 = rank == k - 1 ? mine : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v0 = max(v0, (uint32_t)__shfl_xor((int)v0, o, 64));
        }
        if (tid == 0) { t0_sh = max(kth, v0); nsurv_sh = 0; m_sh = 0; ties_sh = 0; }
    }
    __syncthreads();
    const uint32_t T0 = t0_sh;
    SSTAMP(1);
    for (int j = tid; j < SA_BLOCKS * k; j += 1024) {
        const int b = j / k, r = j % k;
        const uint32_t key = cand_key[b * SA_KMAX + r];
        if (key != 0u && key >= T0) {
            const int slot = atomicAdd(&nsurv_sh, 1);
            if (slot < SA_BLOCKS * SA_KMAX) { ck[slot] = key; ci[slot] = cand_idx[b * SA_KMAX + r]; }
        }
    }
    __syncthreads();
    const int nsurv = min(nsurv_sh, SA_BLOCKS * SA_KMAX);
    SSTAMP(2);
    const int nc = (nsurv + 3) & ~3;                                    // padded with "no candidate" entries
    if (tid < nc - nsurv) { ck[nsurv + tid] = 0u; ci[nsurv + tid] = 0x7fffffff; }
    __syncthreads();
    // rank of a candidate = number of candidates in front of it (key descending, index ascending). The scan reads four
    // candidates per LDS access and is unrolled: one candidate per access made the loop wait for the LDS latency on
    // every iteration (1280 dependent round trips: 355 us per token; now ~15 us)
    auto rank_of = [&](uint32_t kj, int ij) {
        int rank = 0;
#pragma unroll 4
        for (int t = 0; t < nc; t += 4) {
            const u32x4 k4 = *reinterpret_cast<const u32x4*>(&ck[t]);
            const u32x4 i4 = *reinterpret_cast<const u32x4*>(&ci[t]);
#pragma unroll
            for (int e = 0; e < 4; ++e) rank += (k4[e] != 0u && better(k4[e], (int)i4[e], kj, ij)) ? 1 : 0;
        }
        return rank;
    };
    for (int j = tid; j < nc; j += 1024) {
        const uint32_t kj = ck[j];
        const int ij = ci[j];
        if (kj == 0u) continue;
        const int rank = rank_of(kj, ij);
        if (rank < k) { sk[rank] = kj; si[rank] = ij; atomicAdd(&m_sh, 1); }
    }
    __syncthreads();
    const int m0 = m_sh;   // = min(k, number of candidates)
    SSTAMP(3);
    if (m0 == k) {         // HF keeps every score >= the k-th largest: ties of the threshold value survive too
        const uint32_t thr = sk[k - 1];
        for (int j = tid; j < nc; j += 1024) {
            if (ck[j] != thr) continue;
            const int rank = rank_of(ck[j], ci[j]);
            if (rank >= k) {
                const int s = atomicAdd(&ties_sh, 1);
                if (k + s < KEPT_MAX) { sk[k + s] = ck[j]; si[k + s] = ci[j]; }
            }
        }
    }
    __syncthreads();
    const int m = min(m0 + ties_sh, KEPT_MAX);
    SSTAMP(4);
    if (m == 0) { if (tid == 0) { out_token[0] = 0; if (n_kept) n_kept[0] = 0; } return; }
    // exp and the exponential draw of every kept candidate in parallel (a Philox draw is ~100 instructions: done one
    // after the other by thread 0 they were most of this kernel's 72 us); the sums below keep their sequential order,
    // so every result is bit for bit what the single-threaded tail produced
    __shared__ float ev[KEPT_MAX], qv[KEPT_MAX];
    const float mx = fkey_inv(sk[0]);
    for (int j = tid; j < m; j += 1024) {
        ev[j] = expf(fkey_inv(sk[j]) - mx);
        qv[j] = draw_q(draws, seed, offset, si[j]);
    }
    __syncthreads();
    if (tid != 0) return;
    SSTAMP(5);
    // probabilities of the kept scores (softmax over the row with everything else at -inf), fp32 like torch
    float sum = 0.f;
    for (int j = 0; j < m; ++j) sum += ev[j];
    int keep = m;   // entries 0..keep-1 survive top-p (they are in descending order)
    if (lim > 0.0f) {
        float cum = 0.f;
        for (int j = m - 1; j >= 1; --j) {
            cum += ev[j] / sum;
            if (cum <= lim) keep = j; else break;
        }
    }
    float sum2 = 0.f;
    for (int j = 0; j < keep; ++j) sum2 += ev[j];
    float best = -1.f;
    int best_i = si[0];
    for (int j = 0; j < keep; ++j) {
        const float p = ev[j] / sum2;
        const float sc = p / qv[j];
        if (sc > best || (sc == best && si[j] < best_i)) { best = sc; best_i = si[j]; }
        if (kept_idx && j < kept_cap) { kept_idx[j] = si[j]; kept_prob[j] = p; }
    }
    SSTAMP(6);
    out_token[0] = best_i;
    if (n_kept) n_kept[0] = keep;
}

// ---- general path: one workgroup, the whole row, radix selects ----
__device__ __forceinline__ double block_sum_d(double v, double* sh /*16*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += sh[w];
    return t;
}

__global__ __launch_bounds__(1024) void sample_full_kernel(const float* __restrict__ logits_raw, int n, int top_k, float lim_f, float temperature,
                                                           const float* __restrict__ draws, uint64_t seed, uint64_t offset,
                                                           int64_t* __restrict__ out_token, int32_t* __restrict__ kept_idx,
                                                           float* __restrict__ kept_prob, int* __restrict__ n_kept, int kept_cap) {
    __shared__ unsigned int hist[256];
    __shared__ double dsh[16];
    __shared__ double dh[16][16];
    __shared__ uint32_t sel_sh;
    __shared__ float best_v[16];
    __shared__ int best_i[16];
    __shared__ int cnt_sh;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

    // 1. top-k threshold key: the k-th largest key by 4 levels of 8-bit counting
    uint32_t tk = 0u;   // keep key >= tk
    if (top_k > 0 && top_k < n) {
        uint32_t prefix = 0u;
        int need = top_k;
        for (int level = 0; level < 4; ++level) {
            const int shift = 24 - 8 * level;
            if (tid < 256) hist[tid] = 0u;
            __syncthreads();
            for (int i = tid; i < n; i += 1024) {
                const uint32_t key = fkey((logits_raw[i] / temperature));
                if (level == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int d = 255;
                for (; d > 0; --d) {
                    if ((int)hist[d] >= need) break;
                    need -= (int)hist[d];
                }
                sel_sh = (uint32_t)d;
                cnt_sh = need;
            }
            __syncthreads();
            prefix |= sel_sh << shift;
            need = cnt_sh;
            __syncthreads();
        }
        tk = prefix;
    }
    // 2. max and sum of exp over the kept scores
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 1024) {
        const float s = (logits_raw[i] / temperature);
        if (fkey(s) >= tk) mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    __syncthreads();
    if (lane == 0) best_v[wid] = mx;
    __syncthreads();
    for (int w = 0; w < 16; ++w) mx = fmaxf(mx, best_v[w]);
    double acc = 0.0;
    for (int i = tid; i < n; i += 1024) {
        const float s = (logits_raw[i] / temperature);
        if (fkey(s) >= tk) acc += (double)expf(s - mx);
    }
    const float sum = (float)block_sum_d(acc, dsh);
    // 3. top-p threshold on the probability bit patterns: remove p < tp where the ascending cumulative mass of
    //    everything below tp is <= 1 - top_p (8 levels of 4-bit digits, fixed-order fp64 sums)
    uint32_t tp = 0u;
    if (lim_f > 0.0f && lim_f < 1.0f) {
        const double lim = (double)lim_f;
        double below = 0.0;
        uint32_t prefix = 0u;
        for (int level = 0; level < 8; ++level) {
            const int shift = 28 - 4 * level;
            double part[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) part[d] = 0.0;
            for (int i = tid; i < n; i += 1024) {
                const float s = (logits_raw[i] / temperature);
                if (fkey(s) < tk) continue;
                const float p = expf(s - mx) / sum;
                const uint32_t pb = __float_as_uint(p);
                if (level == 0 || (pb >> (shift + 4)) == (prefix >> (shift + 4))) {
                    const int dg = (pb >> shift) & 15u;
#pragma unroll
                    for (int d = 0; d < 16; ++d) part[d] += (dg == d) ? (double)p : 0.0;
                }
            }
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                double v = part[d];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0) dh[wid][d] = v;
            }
            __syncthreads();
            if (tid == 0) {
                int d = 0;
                double b = below;
                for (; d < 15; ++d) {
                    double t = 0.0;
                    for (int w = 0; w < 16; ++w) t += dh[w][d];
                    if (b + t > lim) break;     // the ascending cumulative sum crosses 1 - top_p inside digit d
                    b += t;
                }
                sel_sh = (uint32_t)d;
                dsh[0] = b;
            }
            __syncthreads();
            prefix |= sel_sh << shift;
            below = dsh[0];
            __syncthreads();
        }
        tp = prefix;
    } else if (lim_f >= 1.0f) {
        tp = 0x7f800000u;   // nothing but the best token
    }
    // 4. renormalise over the survivors and draw
    acc = 0.0;
    for (int i = tid; i < n; i += 1024) {
        const float s = (logits_raw[i] / temperature);
        if (fkey(s) < tk) continue;
        const float e = expf(s - mx);
        if (__float_as_uint(e / sum) >= tp || s == mx) acc += (double)e;
    }
    const float sum2 = (float)block_sum_d(acc, dsh);
    if (tid == 0) cnt_sh = 0;
    __syncthreads();
    float bv = -1.f;
    int bi = 0x7fffffff;
    for (int i = tid; i < n; i += 1024) {
        const float s = (logits_raw[i] / temperature);
        if (fkey(s) < tk || s == -INFINITY) continue;
        const float e = expf(s - mx);
        if (!(__float_as_uint(e / sum) >= tp || s == mx)) continue;
        const float p = e / sum2;
        const float sc = p / draw_q(draws, seed, offset, i);
        if (sc > bv || (sc == bv && i < bi)) { bv = sc; bi = i; }
        if (n_kept) {
            const int slot = atomicAdd(&cnt_sh, 1);
            if (kept_idx && slot < kept_cap) { kept_idx[slot] = i; kept_prob[slot] = p; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    __syncthreads();
    if (lane == 0) { best_v[wid] = bv; best_i[wid] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (best_v[w] > bv || (best_v[w] == bv && best_i[w] < bi)) { bv = best_v[w]; bi = best_i[w]; }
        out_token[0] = bi == 0x7fffffff ? 0 : bi;
        if (n_kept) n_kept[0] = cnt_sh;
    }
}

}  // namespace

size_t cogs_k_sample_ws() { return (size_t)SA_BLOCKS * SA_KMAX * 8; }

int cogs_k_sample(hipStream_t st, const float* logits, int n, float temperature, int top_k, double top_p, const float* draws, uint64_t seed,
                  uint64_t offset, int64_t* out_token, int32_t* kept_idx, float* kept_prob, int* n_kept, int kept_cap,
                  void* ws) {
    if (n <= 0 || !logits || !out_token || top_k < 0 || !(temperature > 0.f)) return COGS_E_INVALID;
    if (kept_idx && (!kept_prob || !n_kept || kept_cap <= 0)) return COGS_E_INVALID;
    // TopPLogitsWarper compares the fp32 cumulative sum with the python double (1 - top_p) converted to fp32
    const float lim = top_p >= 1.0 ? 0.0f : (float)(1.0 - top_p);
    if (top_k > 0 && top_k <= SA_KMAX && n <= SA_BLOCKS * SA_SLICE && n >= SA_BLOCKS * top_k) {
        if (!ws) return COGS_E_WORKSPACE;
        uint32_t* ck = (uint32_t*)ws;
        int* ci = (int*)(ck + SA_BLOCKS * SA_KMAX);
        hipLaunchKernelGGL(sample_slice_topk_kernel, dim3(SA_BLOCKS), dim3(256), 0, st, logits, n, top_k, temperature, ck, ci);
        hipLaunchKernelGGL(sample_merge_kernel, dim3(1), dim3(1024), 0, st, ck, ci, top_k, lim, draws, seed, offset,
                           out_token, kept_idx, kept_prob, n_kept, kept_cap);
    } else {
        hipLaunchKernelGGL(sample_full_kernel, dim3(1), dim3(1024), 0, st, logits, n, top_k, lim, temperature, draws, seed, offset,
                           out_token, kept_idx, kept_prob, n_kept, kept_cap);
    }
    return COGS_LAUNCH_CHECK();
}
