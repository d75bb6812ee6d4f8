// Per-frame (block-diagonal) ViT attention, head dim 72, bf16, pre-scaled Q: the production encoder path.
//
// Replaces flash_attn_varlen_func(q, k, v, cu_seqlens, ...) of model/modeling_videollama3_encoder.py:309-312 for
// the shapes the encoder really runs (16 heads of 72, 200..1024 patches per frame). The general kernel in attn.hip
// (16x16x32 tiles, two barriers per key tile, causal / GQA / bias / split-KV modes) stays for everything else.
//
// Structure (gfx950, wave64):
//   * workgroup = 4 waves = 128 query rows of one (frame, head); wave = 32 query rows; key tiles of 64;
//   * v_mfma_f32_32x32x16_bf16 for both products. S^T[key][q] = K.Q^T puts a query COLUMN on each lane (lanes l and
//     l+32 share it), so the row maximum is 31 v_max + ONE v_permlane32_swap, and the S^T accumulator registers
//     8s..8s+7 of a lane ARE the B-operand fragment of k-step s of O^T[d][q] += V^T[d][key].P^T[key][q]
//     (MI355X fragment maps: register j of lane half h = key 16s + 8(j>>2) + 4h + (j&3)); V^T fragments come from
//     the row-major V tile through ds_read_b64_tr_b16 at exactly those keys. P never leaves registers, nothing is
//     permuted in LDS. Head dim 72 is padded to 80 for QK^T (5 k-steps) and 96 for PV (3 d-blocks): 22 MFMAs of
//     32 cycles per tile and wave against 44 of 16 in the 16x16 kernel -- half the matrix-instruction issues;
//   * the pad column d = 72 carries the softmax bookkeeping through the matrix pipe: K[key][72] = 1 and
//     Q[q][72] = -m(q) make the QK^T product come out as S - m (m = the running reference of the row, a bf16 value),
//     so P = exp2(acc) needs no per-score VALU; V[key][72] = 1 makes row 72 of O^T the softmax denominator;
//   * deferred maximum: the reference m only moves when a tile's maximum exceeds it by more than 2^6 (or on the
//     first tile); then (wave-uniform rare path) the tile's scores are shifted in VALU and O is rescaled BEFORE the
//     tile's PV product. P <= 64 otherwise, harmless in fp32 accumulation;
//   * K/V tiles are register-staged (global loads issued one tile ahead, LDS write after the barrier) into a
//     DOUBLE buffer: one barrier per tile. K rows are 176 B (odd number of 16-B chunks: conflict-free
//     ds_read_b128 over the 32 keys a 32x32 A-fragment read touches), V rows 192 B (four consecutive rows fall in
//     four different 64-B bank quarters for the transposing read). 46 KiB per workgroup: 3 workgroups per CU.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct VitAttnArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O;
    long ldq, ldk, ldv, ldo;   // elements
    const int* cu;             // [nseg + 1]
    int nseg, heads, nqb;      // grid = nseg * heads * nqb workgroups (nqb = 128-row query blocks of the longest segment)
};

constexpr float RESCALE_THR = 6.0f;

#ifdef COGS_ATTN_STAMPS   // diagnostic build (tools/micro/attn_vit_micro.cpp): where does a tile's time go, wave 0 of workgroup 0
__device__ unsigned long long g_attn_stamps[8];
#define STAMP(acc_) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_ += now_ - st_prev; st_prev = now_; } while (0)
#else
#define STAMP(acc_) do {} while (0)
#endif

__device__ __forceinline__ float bf16_round(float x) { return bf2f(f2bf(x)); }

template <int HD>
__global__ __launch_bounds__(256, 2) void attn_vit_kernel(VitAttnArgs p) {
    constexpr int NT = 256, QB = 128;               // threads and query rows per workgroup
    static_assert(HD % 8 == 0 && HD % 16 == 8 && HD < 96, "pad column HD must open a fresh 16-byte chunk inside the last k-step");
    constexpr int KS = (HD + 8) / 16;          // QK^T k-steps of 16 (72 -> 5: columns 0..79, column HD = the shift slot)
    constexpr int DB = (HD + 8 + 31) / 32;     // PV d-blocks of 32 (72 -> 3: rows 0..95 of O^T, row HD = the denominator)
    constexpr int CH = HD / 8;                 // real 16-byte chunks per K/V row
    constexpr int KRS = 176, VRS = 192;        // LDS row strides (bytes)
    static_assert(2 * KS * 16 <= KRS && DB * 64 <= VRS, "row strides");
    constexpr int KT = 64 * KRS, VT = 64 * VRS, BUF = KT + VT;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int h = lane >> 5, r32 = lane & 31;
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (id % 8), each with its own L2. The q-blocks of one
    // (frame, head) all read the same K/V: keep them on ONE XCD and close in time -- id = (group * nqb + qb) * 8 + slot with
    // (frame, head) = group * 8 + slot. (The plain (qb, head, frame) grid put the 8 q-blocks of a (frame, head) on 8
    // different XCDs at the same moment: every L2 fetched the same K/V from the fabric, 8x the traffic.)
    int seg, head, qb;
    {
        const int nhf = p.nseg * p.heads, id = blockIdx.x;
        if (nhf % 8 == 0) {
            const int slot = id & 7, rest = id >> 3;
            qb = rest % p.nqb;
            const int hf = (rest / p.nqb) * 8 + slot;
            seg = hf / p.heads; head = hf % p.heads;
        } else {
            qb = id % p.nqb; head = (id / p.nqb) % p.heads; seg = id / (p.nqb * p.heads);
        }
    }
    const int qs = p.cu[seg], qe = p.cu[seg + 1];
    const int q0 = qs + qb * QB;
    if (q0 >= qe) return;
    const int nt = (qe - qs + 63) >> 6;

    // pad chunks of both buffers, written once (staging only writes chunks 0..CH-1): K chunk CH = [1, 0 x7] (the shift
    // slot), V chunk CH = [1, 0 x7] (the denominator row), the remaining chunks zero
    for (int id = tid; id < 2 * 64; id += NT) {
        char* b = smem + (id >> 6) * BUF;
        const int row = id & 63;
        const u32x4 one = u32x4{0x00003f80u, 0, 0, 0}, zero = u32x4{0, 0, 0, 0};
        *reinterpret_cast<u32x4*>(b + row * KRS + CH * 16) = one;
#pragma unroll
        for (int c = CH + 1; c < KRS / 16; ++c) *reinterpret_cast<u32x4*>(b + row * KRS + c * 16) = zero;
        *reinterpret_cast<u32x4*>(b + KT + row * VRS + CH * 16) = one;
#pragma unroll
        for (int c = CH + 1; c < VRS / 16; ++c) *reinterpret_cast<u32x4*>(b + KT + row * VRS + c * 16) = zero;
    }

    // Q fragments (B operand of S^T = K.Q^T): lane (q = r32, h) holds Q[q][16s + 8h + j]; columns >= HD are zero
    const int qrow = q0 + wid * 32 + r32;
    const bool qok = qrow < qe;
    const bool wave_active = q0 + wid * 32 < qe;   // wave-uniform: the ragged last block (924 = 7*128 + 28)
    u32x4 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 16 * s + 8 * h;
        qf[s] = u32x4{0, 0, 0, 0};
        if (qok && k < HD) qf[s] = *reinterpret_cast<const u32x4*>(p.Q + (long)qrow * p.ldq + head * HD + k);
    }

    // staging: thread -> chunk column c = tid % CH, rows tid / CH + RSTEP * i (i < PER): every slot's offsets are the
    // slot-0 offsets plus a compile-time multiple of the row stride, so nothing per-slot lives in registers
    constexpr int RSTEP = NT / CH;                           // 28 rows per slot (256 threads), threads >= RSTEP * CH idle
    constexpr int PER = (64 + RSTEP - 1) / RSTEP;            // 3
    const int st_row0 = tid < RSTEP * CH ? tid / CH : 1 << 20;
    const int st_c = tid % CH;
    const int g_off_k0 = st_row0 * (int)p.ldk + head * HD + st_c * 8;
    const int g_off_v0 = st_row0 * (int)p.ldv + head * HD + st_c * 8;
    const int l_off_k0 = st_row0 * KRS + st_c * 16;
    const int l_off_v0 = KT + st_row0 * VRS + st_c * 16;
    u32x4 kreg[PER], vreg[PER];
    // (measured and dropped, tools/micro + DESIGN.md section 8: the same pieces spread over the tile's compute phases,
    // LDS-DMA instead of register staging, 8-wave workgroups, dedicated loader waves with a 3-deep LDS ring)
    auto load_k = [&](int t) {
        const int kbase = qs + t * 64;
        const int valid = min(64, qe - kbase);           // rows past the frame are zero filled (never read out of range)
        const bf16_t* kb = p.K + (long)kbase * p.ldk;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            kreg[i] = u32x4{0, 0, 0, 0};
            if (st_row0 + RSTEP * i < valid) kreg[i] = *reinterpret_cast<const u32x4*>(kb + g_off_k0 + (long)(RSTEP * i) * p.ldk);
        }
    };
    auto load_v = [&](int t) {
        const int kbase = qs + t * 64;
        const int valid = min(64, qe - kbase);
        const bf16_t* vb = p.V + (long)kbase * p.ldv;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            vreg[i] = u32x4{0, 0, 0, 0};
            if (st_row0 + RSTEP * i < valid) vreg[i] = *reinterpret_cast<const u32x4*>(vb + g_off_v0 + (long)(RSTEP * i) * p.ldv);
        }
    };
    auto write_k = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (st_row0 + RSTEP * i < 64) *reinterpret_cast<u32x4*>(buf + l_off_k0 + RSTEP * i * KRS) = kreg[i];
    };
    auto write_v = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (st_row0 + RSTEP * i < 64) *reinterpret_cast<u32x4*>(buf + l_off_v0 + RSTEP * i * VRS) = vreg[i];
    };

    // per-lane LDS read offsets
    const int k_rd = r32 * KRS + h * 16;                                              // + kb*32*KRS + s*32
    const int v_rd = KT + (4 * h + ((lane & 15) >> 2)) * VRS + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
                                                                                      // + (32kb + 16s2)*VRS (+8*VRS) + b*64
    f32x16 oacc[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[b][r] = 0.f;
    float sh = 0.f;          // the row's reference (bf16-exact); -sh sits in Q's pad column
    bool first = true;

#ifdef COGS_ATTN_STAMPS
    unsigned long long st_bar = 0, st_stage = 0, st_qk = 0, st_sm = 0, st_pv = 0, st_prev = __builtin_amdgcn_s_memtime();
#endif
    load_k(0); load_v(0);
    write_k(smem); write_v(smem);
    if (nt > 1) { load_k(1); load_v(1); }
    // The Q loads must be seen as COMPLETE before the loop: otherwise hipcc's wait insertion puts an `s_waitcnt
    // vmcnt(0)` in front of the first QK^T MFMA of every tile (the loop header merges "Q may be in flight"), and that
    // wait also drains the K/V loads issued a moment earlier for tile t+2 -- the whole prefetch would be serialised.
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qf[s]));

    auto compute = [&](const char* buf, const int kbase, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const int valid = qe - kbase;                                  // MASKED: keys >= valid do not exist
        const bool two = !MASKED || valid > 32;                        // wave-uniform: second 32-key block present
        f32x16 sc[2];
        // S^T - m for the tile (the shift rides in Q's pad column); all K fragments of a key block are read first
        auto qk = [&]() {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (MASKED && kb == 1 && !two) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[1][r] = -INFINITY;
                    continue;
                }
                u32x4 kf[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) kf[s] = *reinterpret_cast<const u32x4*>(buf + k_rd + kb * 32 * KRS + s * 32);
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    f32x16 c0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) c0[r] = 0.f;
                    sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[s]), __builtin_bit_cast(bf16x8, qf[s]),
                                                                     s == 0 ? c0 : sc[kb], 0, 0, 0);
                }
            }
            if constexpr (MASKED) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (key >= valid) sc[kb][r] = -INFINITY;
                    }
            }
        };
        qk();
        STAMP(st_qk);
        // V^T fragments of d-block 0, issued before the softmax so that their LDS latency hides under it
        auto read_v = [&](int b, int kb, int s2) -> u32x4 {
            const char* va = buf + v_rd + (32 * kb + 16 * s2) * VRS + b * 64;
            const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(va));
            const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(va + 8 * VRS));
            const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
            return u32x4{l2[0], l2[1], h2[0], h2[1]};
        };
        u32x4 vf0[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) vf0[kb][s2] = (MASKED && kb == 1 && !two) ? u32x4{0, 0, 0, 0} : read_v(0, kb, s2);
        // row maximum relative to the reference: own 32 keys, then the partner half of the lane pair
        float d = sc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) d = fmaxf(d, sc[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) d = fmaxf(d, sc[1][r]);
        {
            const unsigned db = __builtin_bit_cast(unsigned, d);
            const auto sw = __builtin_amdgcn_permlane32_swap(db, db, false, false);
            d = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
        }
        if (first || __any(d > RESCALE_THR)) {
            // rare (wave-uniform): move the reference of the rows that need it, m_new = bf16(sh + dd); O (expressed
            // relative to sh) is multiplied by 2^-(m_new - sh) and the tile's scores are simply computed AGAIN with the
            // new shift in Q's pad column -- the common path carries no correction arithmetic at all
            float dd = first ? d : fmaxf(d, 0.f);
            if (!(dd > -INFINITY)) dd = 0.f;
            const float m_new = (first || d > RESCALE_THR) ? bf16_round(sh + dd) : sh;
            if (!first) {
                const float al = __builtin_amdgcn_exp2f(sh - m_new);
#pragma unroll
                for (int b = 0; b < DB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[b][r] *= al;
            }
            sh = m_new;
            const unsigned bits = __float_as_uint(-sh) >> 16;          // exact: sh is a bf16 value
            if (h) qf[KS - 1][0] = (qf[KS - 1][0] & 0xffff0000u) | bits;
            first = false;
            qk();
        }
        u32x4 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    pf[kb][s2][w] = pack_bf2(__builtin_amdgcn_exp2f(sc[kb][8 * s2 + 2 * w]),
                                             __builtin_amdgcn_exp2f(sc[kb][8 * s2 + 2 * w + 1]));
#ifdef COGS_ATTN_STAMPS
        asm volatile("" :: "v"(pf[1][1]));
#endif
        STAMP(st_sm);
        // O^T[d][q] += V^T[d][key] . P^T[key][q]
#pragma unroll
        for (int b = 0; b < DB; ++b) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (MASKED && kb == 1 && !two) continue;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const u32x4 w = b == 0 ? vf0[kb][s2] : read_v(b, kb, s2);
                    oacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, pf[kb][s2]),
                                                                     oacc[b], 0, 0, 0);
                }
            }
        }
    };

    // full tiles in one loop, the ragged last tile (924 = 14 * 64 + 28) after it: one loop body holding both variants
    // made hipcc merge the accumulators of the two paths with 24 register-pair copies per tile
    const int full_tiles = (qe - qs) >> 6;
    for (int t = 0; t < full_tiles; ++t) {
        STAMP(st_pv);
        __syncthreads();                        // tile t is in buffer t&1; everyone is done with buffer (t+1)&1
        STAMP(st_bar);
        if (t + 1 < nt) {
#ifndef ABL_NOWRITE
            write_k(smem + ((t + 1) & 1) * BUF); write_v(smem + ((t + 1) & 1) * BUF);
#endif
#ifndef ABL_NOLOAD
            if (t + 2 < nt) { load_k(t + 2); load_v(t + 2); }
#endif
        }
        STAMP(st_stage);
#ifndef ABL_NOCOMPUTE
        if (wave_active) compute(smem + (t & 1) * BUF, qs + t * 64, std::false_type{});   // else: only stage and synchronise
#endif
    }
#ifdef COGS_ATTN_STAMPS
    if (blockIdx.x == 1234 && tid == 0) {
        g_attn_stamps[0] = st_bar; g_attn_stamps[1] = st_stage; g_attn_stamps[2] = st_qk; g_attn_stamps[3] = st_sm;
        g_attn_stamps[4] = st_pv; g_attn_stamps[5] = full_tiles;
    }
#endif
    if (full_tiles < nt) {
        __syncthreads();
        if (wave_active) compute(smem + (full_tiles & 1) * BUF, qs + full_tiles * 64, std::true_type{});
    }
    if (!wave_active) return;

    // epilogue: the denominator is row HD of O^T = block HD/32, register with (r&3) + 8(r>>2) + 4h == HD % 32
    constexpr int LB = HD / 32, LR = HD % 32;              // 72 -> block 2, row 8 -> h = 0, register 4
    constexpr int LH = (LR >> 2) & 1, LREG = (LR & 3) + 4 * (LR >> 3);
    float l = oacc[LB][LREG];
    l = __shfl(l, r32 + 32 * LH, 64);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    // lane (q, h) holds d = 32b + 8g4 + 4h + 0..3 (g4 = 0..3). Pack to bf16 and exchange between the two halves so that
    // a lane owns 8 consecutive d: half 0 gets the even g4 groups, half 1 the odd ones -> 16-byte stores
    bf16_t* orow = p.O + (long)qrow * p.ldo + head * HD;
#pragma unroll
    for (int b = 0; b < DB; ++b) {
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            if (32 * b + 16 * gp >= HD) continue;
            unsigned e0 = pack_bf2(oacc[b][8 * gp + 0] * inv, oacc[b][8 * gp + 1] * inv);
            unsigned e1 = pack_bf2(oacc[b][8 * gp + 2] * inv, oacc[b][8 * gp + 3] * inv);
            unsigned o0 = pack_bf2(oacc[b][8 * gp + 4] * inv, oacc[b][8 * gp + 5] * inv);
            unsigned o1 = pack_bf2(oacc[b][8 * gp + 6] * inv, oacc[b][8 * gp + 7] * inv);
            const auto s0 = __builtin_amdgcn_permlane32_swap(e0, o0, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(e1, o1, false, false);
            // half 0: (s0[0], s1[0]) = own even group, (s0[1], s1[1]) = partner's even group (d + 4)
            // half 1: (s0[0], s1[0]) = partner's odd group (d + 0), (s0[1], s1[1]) = own odd group (d + 4)
            const int d0 = 32 * b + 16 * gp + 8 * h;
            if (qok && d0 < HD) *reinterpret_cast<u32x4*>(orow + d0) = u32x4{(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};
        }
    }
}


}  // namespace

// block-diagonal bf16 attention with pre-scaled Q, hd 72, hq == hkv
int cogs_k_attention_vit(hipStream_t st, const CogsAttn& a) {
    if (a.head_dim != 72 || !a.cu_seqlens || a.nseg <= 0 || a.hq != a.hkv) return COGS_E_UNSUPPORTED;
    if (a.ldq % 8 || a.ldk % 8 || a.ldv % 8 || a.ldo % 8) return COGS_E_INVALID;
    VitAttnArgs p;
    p.Q = (const bf16_t*)a.Q; p.K = (const bf16_t*)a.K; p.V = (const bf16_t*)a.V; p.O = (bf16_t*)a.O;
    p.ldq = a.ldq; p.ldk = a.ldk; p.ldv = a.ldv; p.ldo = a.ldo;
    p.cu = a.cu_seqlens;
    p.nseg = a.nseg; p.heads = a.hq; p.nqb = (a.max_seqlen + 127) / 128;
    if ((long)p.nseg * p.heads * p.nqb > 0x7fffffffL) return COGS_E_INVALID;
    dim3 grid(p.nseg * p.heads * p.nqb);
    hipLaunchKernelGGL(attn_vit_kernel<72>, grid, dim3(256), 0, st, p);
    return COGS_LAUNCH_CHECK();
}
