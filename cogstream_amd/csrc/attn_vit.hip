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
#include "debug.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct VitAttnArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O;
    long ldq, ldk, ldv, ldo;   // elements
    const int* cu;             // [nseg + 1]
    int nseg, heads, nqb;      // grid = nseg * heads * nqb workgroups (nqb = 128-row query blocks of the longest segment)
    int xcd_order;             // 1: grid (8 nqb, heads / 8, nseg), head = 8 y + (x & 7); 0: grid (nqb, heads, nseg)
    int early_prefetch;        // pipe kernel prologue: tiles 1 and 2 issued before (1) / behind (0) the first wait
    long head_stride;          // elements from head h to head h + 1 of Q / K / V: HD for the token-major layout (ld* = row stride of
                               // the fused qkv buffer), rows * HD for the head-major one (ld* = HD: a head's rows back to back, so a
                               // key tile is ONE contiguous run of 64 x 144 bytes and every 1 KiB LDS-DMA piece is whole lines)
    int uniform_len;           // > 0: every segment has this many rows and segment s starts at row s * uniform_len (one video:
                               // all frames alike) -- the bounds are then arithmetic on kernel arguments instead of two
                               // dependent scalar loads at the head of every workgroup (700-1 700 cycles of its 39 000)
};

constexpr float RESCALE_THR = 6.0f;

#if defined(COGS_ATTN_STAMPS) || defined(COGS_PIPE_STAMPS) || defined(COGS_PHASE_STAMPS)   // diagnostic builds (tools/micro/attn_vit_micro.cpp)
__device__ unsigned long long g_attn_stamps[8];
__device__ unsigned long long g_tail_stamps[8];
#endif
#ifdef COGS_LIFE_STAMPS   // tools/micro/attn_vit_micro.cpp: summed workgroup lifetimes (s_memtime), wave 0
__device__ unsigned long long g_av_life[8];
#define LIFE_NOW(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define LIFE_ADD(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_av_life[i], (unsigned long long)(v)); } while (0)
#else
#define LIFE_NOW(v) do {} while (0)
#define LIFE_ADD(i, v) do {} while (0)
#endif
// the workgroup whose stamps are kept is named by its place in the dispatch order (the grid is 3-D since round 6)
#define COGS_AV_LINEAR_ID ((int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x))
#ifdef COGS_ATTN_STAMPS   // where does a tile's time go, wave 0 of one workgroup
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
    // The grid is 3-D so that nothing is divided here (round 6: the three signed divisions of the linear form were ~150 vector
    // instructions at the head of every workgroup's critical path, in front of its first address). heads % 8 == 0 (p.xcd_order):
    // x = query block * 8 + slot, y = head / 8, z = segment, head = 8 y + slot -- the linear dispatch order is
    // ((segment, head / 8), query block, slot), i.e. the same (frame, head) -> XCD map as described above; else plain (qb, head, seg).
    int seg, head, qb;
    if (p.xcd_order) { qb = blockIdx.x >> 3; head = blockIdx.y * 8 + (blockIdx.x & 7); seg = blockIdx.z; }
    else { qb = blockIdx.x; head = blockIdx.y; seg = blockIdx.z; }
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
        if (qok && k < HD) qf[s] = *reinterpret_cast<const u32x4*>(p.Q + (long)qrow * p.ldq + head * p.head_stride + k);
    }

    // staging: thread -> chunk column c = tid % CH, rows tid / CH + RSTEP * i (i < PER): every slot's offsets are the
    // slot-0 offsets plus a compile-time multiple of the row stride, so nothing per-slot lives in registers
    constexpr int RSTEP = NT / CH;                           // 28 rows per slot (256 threads), threads >= RSTEP * CH idle
    constexpr int PER = (64 + RSTEP - 1) / RSTEP;            // 3
    const int st_row0 = tid < RSTEP * CH ? tid / CH : 1 << 20;
    const int st_c = tid % CH;
    const long g_off_k0 = (long)st_row0 * p.ldk + head * p.head_stride + st_c * 8;
    const long g_off_v0 = (long)st_row0 * p.ldv + head * p.head_stride + st_c * 8;
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
    if (COGS_AV_LINEAR_ID == 1234 && tid == 0) {
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


// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined variant with LDS-DMA staging: the same products, padding trick and deferred maximum, but
//   (1) a wave's instruction stream always holds independent matrix AND vector work:
//         iteration t:   P(t) = exp2(S(t))   beside   S(t+1) = K(t+1).Q^T
//                        O += V(t)^T.P(t)    beside   max over S(t+1)
//       An MFMA occupies the matrix pipe for 32 cycles after a 4-cycle issue; the wave issues independent VALU / LDS
//       work into that shadow. In the unpipelined kernel above a wave alternates between pure-MFMA and pure-VALU
//       stretches and the two waves that share a SIMD fall into step -- measured with staging removed: 1 360 cycles
//       per wave and tile = the plain SUM of its 704 MFMA cycles and its VALU stream, no overlap at all;
//   (2) K/V tiles go global -> LDS by global_load_lds_dwordx4 (no staging registers, no ds_write, no load -> write
//       dependency inside the loop) into a ring of FOUR tile slots, three tiles ahead. A tile is the plain image of
//       its 64 rows x 144 bytes: the 144-byte row stride is conflict-free for the K fragment reads (36 dwords: the 16
//       rows of a ds_read_b128 lane group start at 16 different multiples of 4 banks) and costs the transposing V
//       reads a 2-way conflict on half their lanes. The pad column has no room in that image: the lanes that would
//       read it (K: k-half 1 of the last k-step; V: the pieces d = 72..95 of the last d-block) are pointed at a
//       constant chunk [1, 0 x7] / zeros behind the tile instead.
// Rows past the segment end are read from its last row (finite data, their scores are masked), never from outside.
//
// Stamped life of a workgroup (cfg2 shape, 15 tiles, -DCOGS_PIPE_STAMPS, shader cycles): 7 500 from kernel entry to "Q
// and tile 0 landed" (kernel arguments, segment bounds, Q + first tiles: three dependent round trips on a loaded chip),
// 900 to S(0), 29 000 in the tile loop (1 930 per tile at two workgroups per CU, i.e. 965 per wave and tile on a SIMD
// against 704 cycles of MFMA and an issue-bound floor near 880), 1 900 for normalise + store. A fifth of its life is the
// entry, during which its SIMDs hold one wave only. Keeping the K/V stream running across the seam was built three times
// -- a persistent grid walking all items (also with a rotating item order, so that no workgroup sees only the ragged
// blocks), the same at 32-key granularity, and 2 or 4 consecutive query blocks of one (frame, head) per workgroup -- and
// each time the tile loop lost more (hoisted per-item scalars: 60-80 spilled SGPRs, spilled VGPRs whose reloads count
// on vmcnt; a static item order instead of the dispatcher's dynamic one) than the seam gave back: 0.42-0.44 ms against
// 0.40 ms per layer in the same run. Removed.
// Round 5, measured and removed: the same kernel with EIGHT waves per workgroup (256 query rows, one workgroup per CU: every
// K/V tile staged once for twice the rows, 2.25 instead of 4.5 LDS-DMA pieces per wave and tile, half as many workgroup
// entries per (frame, head)). Bit-identical outputs; attention of a cfg2 step 9.84 -> 11.03 ms, cfg3 4.26 -> 4.78 ms
// (in-process A/B, tools/encoder_ab.py): with one workgroup per CU nothing computes during a workgroup's entry and exit
// (13-15 % of its life, stamps below), and the two waves of a SIMD, now in one workgroup, are put back in step by every
// tile barrier. Two 4-wave workgroups per CU it stays.
// Round 6: R > 0 = the shape of the ragged end is compile-time. Every segment of the launch has the same length (p.uniform_len: one
// video), nt >= 4 key tiles of which the last has NLB (1 or 2) 32-key blocks, and R = 4 + ((nt - 4) & 3) in 4..7 is the number of
// tiles behind the four-tile main loop -- chosen so that nt - R is a multiple of 4 (the first of them sits in ring slot 0) and that
// every tile the main loop requests is a full one (the last tile is requested from the end): ring slots, block kinds, waits, the
// tile requests of the end AND the ragged-tile test of every request are constants instead of run-time tests next to the MFMAs.
// Eight instantiations cover every length; R = 0 keeps the run-time form (segments of different lengths, fewer than 4 tiles). Same
// arithmetic in the same order: bit-identical outputs (checksums of profiles/r6_attn_vit_ct_end_ab.txt, tests/test_gpu_ops.py).
template <int HD, int R = 0, int NLB = 0>
__global__ __launch_bounds__(256, 2) void attn_vit_pipe_kernel(VitAttnArgs p) {
    constexpr int NT = 256, QB = 128, NS = 4, NW = 4;
    static_assert(HD % 8 == 0 && HD % 16 == 8 && HD < 96, "pad column HD must open a fresh 16-byte chunk inside the last k-step");
    constexpr int KS = (HD + 8) / 16, DB = (HD + 8 + 31) / 32, CH = HD / 8;
    constexpr int RS = HD * 2;                        // LDS row stride = the row itself (144 B)
    constexpr int TILE = 64 * RS;                     // 9216
    constexpr int PIECES = TILE / 1024;               // 9 DMA pieces of 1 KiB per matrix and tile
    static_assert(TILE % 1024 == 0, "a tile is a whole number of 1 KiB pieces");
    constexpr int C_ONE = 2 * TILE, C_ZERO = 2 * TILE + 16, STAGE = 2 * TILE + 64;     // [K | V | one-chunk | zero-chunk | -]
#ifdef ABL_ONE_WG      // ablation (tools/attn_abl.sh): one workgroup per CU -- the LDS request no longer fits twice
    __shared__ __attribute__((aligned(16))) char smem[NS * STAGE + 16384];
#else
    __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r32 = lane & 31;
#ifdef COGS_PIPE_STAMPS     // diagnostic build (tools/micro/attn_vit_micro.cpp): life of one workgroup (wave 0) in shader cycles
    unsigned long long ps_t[10];
    int ps_n = 0;
#define PSTAMP() do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); ps_t[ps_n++] = t_; } while (0)
#else
#define PSTAMP() do {} while (0)
#endif
#ifdef COGS_PHASE_STAMPS      // tools/experiments/attn_vit_phases.sh: where a main-loop tile's time goes, wave 0 of one workgroup
#define PHT(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PHT(v) do {} while (0)
#endif
    PSTAMP();      // 0: kernel entry
    unsigned long long life0 = 0, life1 = 0;
    LIFE_NOW(life0);
    int seg, head, qb;                                           // XCD-aware 3-D grid, see attn_vit_kernel
    if (p.xcd_order) { qb = blockIdx.x >> 3; head = blockIdx.y * 8 + (blockIdx.x & 7); seg = blockIdx.z; }
    else { qb = blockIdx.x; head = blockIdx.y; seg = blockIdx.z; }
    const int qs = (R > 0 || p.uniform_len > 0) ? seg * p.uniform_len : p.cu[seg];
    const int qe = (R > 0 || p.uniform_len > 0) ? qs + p.uniform_len : p.cu[seg + 1];
    const int q0 = qs + qb * QB;
#ifdef COGS_PIPE_STAMPS2
    asm volatile("" :: "s"(qs), "s"(qe));
    PSTAMP();      // segment bounds here
#endif
    if (q0 >= qe) return;
    const int nt = (qe - qs + 63) >> 6;
    const int full_tiles = (qe - qs) >> 6;

    if (tid < NS) {                                              // constant chunks, written once
        *reinterpret_cast<u32x4*>(smem + tid * STAGE + C_ONE) = u32x4{0x00003f80u, 0, 0, 0};
        *reinterpret_cast<u32x4*>(smem + tid * STAGE + C_ZERO) = u32x4{0, 0, 0, 0};
    }

    const int qrow = q0 + wid * 32 + r32;
    const bool qok = qrow < qe;
    const bool wave_active = q0 + wid * 32 < qe;
    u32x4 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 16 * s + 8 * h;
        qf[s] = u32x4{0, 0, 0, 0};
        if (qok && k < HD) qf[s] = *reinterpret_cast<const u32x4*>(p.Q + (long)qrow * p.ldq + head * p.head_stride + k);
    }

    // DMA pieces: piece j (0..8) of a matrix covers chunks 64j..64j+63 of the tile image (chunk c = row c / 9, 16-byte
    // column c % 9). Wave w issues pieces w, w + NW, ... < 8 of K and of V (two each with 4 waves, one with 8); piece 8 of
    // K goes to wave 0, of V to wave 1 (-DAV_WAVE_NINTH, the form up to round 5) -- round 6: the ninth pieces go out in HALVES, one
    // half per wave (K lanes 0-31: wave 0, K lanes 32-63: wave 1, V: waves 2 and 3) under a half EXEC mask: every wave issues the
    // same 2 NPI + 1 requests per tile, selected by scalar moves instead of two wave-dependent branches behind the tile's barrier,
    // and the counted waits are exact for all four waves.
    constexpr int NPI = 8 / NW;                                  // regular pieces per matrix and wave
#ifdef AV_WAVE_NINTH
    const int n_own = 2 * NPI + (wid < 2 ? 1 : 0);               // this wave's DMA instructions per tile
#else
    const int n_own = 2 * NPI + 1;
    const unsigned long long ninth_exec = (wid & 1) ? 0xffffffff00000000ull : 0x00000000ffffffffull;
    const bool ninth_v = wid >= 2;
#endif
    int pc_row[NPI + 1], pc_off[NPI + 1];                        // per-lane row and byte offset of the wave's pieces
#pragma unroll
    for (int i = 0; i < NPI + 1; ++i) {
        const int c = (i < NPI ? wid + NW * i : 8) * 64 + lane;
        pc_row[i] = c / CH;
        pc_off[i] = (pc_row[i] * (int)p.ldk + (c % CH) * 8) * 2;   // bytes; ldk == ldv (checked by the launcher). Head-major
    }                                                              // (ldk = HD): = 16 c, the piece is one contiguous KiB
    // The DMA is issued from inline assembly: hipcc's wait insertion treats a `global_load_lds` it can see as a pending
    // LDS store and puts `s_waitcnt vmcnt(0)` in front of the first ds_read_b64_tr_b16 of every tile (the transposing
    // read carries no memory operand to disambiguate), which would drain the three-tile prefetch every iteration. Hidden
    // vector-memory operations only make the compiler's own vmcnt waits (Q loads, O stores) more conservative.
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto uniform_ptr = [](const bf16_t* q) -> const bf16_t* {      // wave-uniform by construction; tell the compiler
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const bf16_t*)(((unsigned long long)hi << 32) | lo);
    };
    // (M0 is a reserved register for hipcc: it never keeps a value there across statements and sets it right in front
    // of its own uses, so writing it here needs no clobber -- naming it in the clobber list only draws a warning)
    auto dma16 = [&](const bf16_t* base, int off_bytes, unsigned lds) {
#ifdef AV_ABL_NORESCALE      // (in this ablation build hipcc loses track of the address being wave-uniform)
        lds = __builtin_amdgcn_readfirstlane(lds);
#endif
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(lds), "v"(off_bytes), "s"(base) : "memory");
    };
    // tiles are issued strictly in order 0, 1, 2, ...: the source of the next tile is a running pointer (one 64-bit scalar
    // add per matrix and tile; the round-4 form rebuilt base + row * ld from scratch: six s_mul and four v_readfirstlane
    // per tile head in every wave)
    const bf16_t* k_next = p.K + (long)qs * p.ldk + head * p.head_stride;
    const bf16_t* v_next = p.V + (long)qs * p.ldv + head * p.head_stride;
    const long tile_step = 64 * p.ldk;                           // elements; ldk == ldv
    auto issue_tile_g = [&](int t, int slot, auto full_tag) {     // slot = t & (NS - 1), a constant in the unrolled loop
        constexpr bool KNOWN_FULL = decltype(full_tag)::value;   // the caller knows the tile has all 64 rows: no test (R > 0 forms)
        const int valid = KNOWN_FULL ? 64 : qe - (qs + t * 64);  // >= 1
        const bf16_t* kb = uniform_ptr(k_next);
        const bf16_t* vb = uniform_ptr(v_next);
        k_next += tile_step; v_next += tile_step;
        const unsigned st = __builtin_amdgcn_readfirstlane(smem_lds + slot * STAGE);
        int off[NPI + 1];
#pragma unroll
        for (int i = 0; i < NPI + 1; ++i) off[i] = pc_off[i];
        if (valid < 64) {                                        // ragged last tile: rows past the end repeat the last row
#pragma unroll
            for (int i = 0; i < NPI + 1; ++i) off[i] = pc_off[i] - (pc_row[i] - min(pc_row[i], valid - 1)) * (int)p.ldk * 2;
        }
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            dma16(kb, off[i], st + (wid + NW * i) * 1024);
            dma16(vb, off[i], st + TILE + (wid + NW * i) * 1024);
        }
#ifdef AV_ABL_NO_NINTH       // timing experiment (tools/attn_abl.sh): what do the ninth pieces cost? (wrong results)
#elif defined(AV_WAVE_NINTH)
        if (wid == 0) dma16(kb, off[NPI], st + 8 * 1024);
        if (wid == 1) dma16(vb, off[NPI], st + TILE + 8 * 1024);
#else
        // EXEC is all ones here (wave-uniform control flow throughout the kernel) and is put back to all ones
        const unsigned long long nx = ninth_exec;                // (copies: an asm operand alone does not capture in a generic lambda)
        const bf16_t* const nb = ninth_v ? vb : kb;
#ifdef AV_ABL_NORESCALE
        const unsigned nl = __builtin_amdgcn_readfirstlane(st + (ninth_v ? TILE : 0) + 8 * 1024);
#else
        const unsigned nl = st + (ninth_v ? TILE : 0) + 8 * 1024;
#endif
        const int no = off[NPI];
        asm volatile("s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                     :: "s"(nx), "s"(nl), "v"(no), "s"(nb) : "memory");
#endif
    };
    auto issue_tile = [&](int t, int slot) { issue_tile_g(t, slot, std::false_type{}); };      // ragged or not: tested
    using FirstTilesFull = std::integral_constant<bool, (R > 0)>;                               // R > 0: nt >= 4
    // this wave's pieces of every tile but the `newer` most recently issued ones have landed (n_own = 2 NPI or 2 NPI + 1)
#ifdef AV_OLD_BRANCHES      // A/B builds (tools/attn_abl.sh): the round-5 control flow of the tile loop
    auto wait_tiles = [&](int newer) {
        if (newer <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (newer == 1) { if (n_own == 2 * NPI + 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NPI + 1) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NPI) : "memory"); }
        else { if (n_own == 2 * NPI + 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * NPI + 2) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * NPI) : "memory"); }
    };
#else
    // Round 6: ONE immediate per case for all four waves -- the count of the waves that issue 2 NPI pieces per tile. The two waves
    // with an extra piece then also wait for the oldest piece of the next-but-one tile (issued a whole tile earlier: landed), and
    // the tile loop loses a wave-dependent branch in front of its barrier (a scalar branch next to an MFMA block costs percent:
    // profiles/r6_gemm_idle_branch_ab.txt, r6_prefill_attn_branches_ab.txt).
    (void)n_own;
#ifdef AV_WAVE_NINTH
    constexpr int PER_TILE = 2 * NPI;
#else
    constexpr int PER_TILE = 2 * NPI + 1;      // every wave alike: exact
#endif
    auto wait_tiles = [&](int newer) {
        if (newer <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER_TILE) : "memory");
    };
#endif

#ifndef AV_LATE_ISSUE
    // Round 5: tile 0 goes out HERE, as soon as its addresses exist -- in front of the ~800 cycles of per-lane read offsets,
    // accumulator zeroing and fragment-address setup below, which then run under the memory round trip instead of in front of it
    // (inline-asm DMA is a scheduling boundary for hipcc: where the statement stands is where the instruction goes).
    issue_tile_g(0, 0, FirstTilesFull{});
#endif
    // per-lane LDS read offsets inside a stage. K: lane (key r32, k-half h) reads 16 bytes at column 32 ks + 16 h; in
    // the last k-step half 1 is the pad chunk -> constant [1, 0 x7]
    const int k_rd = r32 * RS + h * 16;
    int k_last[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) k_last[kb] = h ? C_ONE : k_rd + kb * 32 * RS + (KS - 1) * 32;
    // V: lane supplies the 8-byte piece (key 4h + ((lane & 15) >> 2) [+8], d = 32 b + 16 ((lane >> 4) & 1) + 4 (lane & 3) .. +3)
    const int v_d = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int v_rd = TILE + (4 * h + ((lane & 15) >> 2)) * RS + v_d * 2;
    // last d-block: d = 32 (DB - 1) + v_d; real data below HD, [1, 0, 0, 0] at HD, zeros above
    constexpr int DL = 32 * (DB - 1);
    const bool v_last_real = DL + v_d < HD;
    const int v_last_const = DL + v_d == HD ? C_ONE : C_ZERO;

    f32x16 oacc[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[b][r] = 0.f;
    float sh = 0.f;

    // The pipeline runs at 32-key BLOCK granularity (two blocks per tile): block j = tile j >> 1, key half j & 1.
    //     sub-step j:   P(j) = exp2(S(j))      beside   S(j+1) = K(j+1).Q^T      (5 chained MFMAs)
    //                   O += V(j)^T.P(j)       beside   max over S(j+1)          (6 MFMAs)
    // and the LDS fragment reads run one stage ahead of their MFMAs: the V fragments of block j are requested at the
    // start of sub-step j (consumed in its second phase), the K fragments of block j+2 at the start of its second phase
    // (consumed in sub-step j+1). Half-tile blocks keep the live set small enough for that (S 2 x 16, P 8, K fragments
    // 20, V fragments 24 registers): with whole tiles in flight the compiler had no registers left and read every V
    // fragment right in front of its MFMA -- twelve exposed LDS latencies per tile.
    const int len = qe - qs;
    const int nblk = (len + 31) >> 5, nfull = len >> 5;
    auto read_k = [&](const char* st, int kb, u32x4 (&kf)[KS]) {
#pragma unroll
        for (int ks = 0; ks < KS - 1; ++ks) kf[ks] = *reinterpret_cast<const u32x4*>(st + k_rd + kb * 32 * RS + ks * 32);
        kf[KS - 1] = *reinterpret_cast<const u32x4*>(st + k_last[kb]);
    };
    auto qk_blk = [&](const u32x4 (&kf)[KS], f32x16& sx) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f32x16 c0;
#pragma unroll
            for (int r = 0; r < 16; ++r) c0[r] = 0.f;
            sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[ks]), __builtin_bit_cast(bf16x8, qf[ks]),
                                                         ks == 0 ? c0 : sx, 0, 0, 0);
        }
    };
    auto mask_blk = [&](f32x16& sx, const int valid) {            // keys >= valid of the block do not exist
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (key >= valid) sx[r] = -INFINITY;
        }
    };
    // largest score of this lane's 16 keys of a block, and the same over the lane pair (l, l + 32) that shares a query row
    auto own_max = [&](const f32x16& sx) -> float {
        float d = sx[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) d = fmaxf(d, sx[r]);
        return d;
    };
    auto pair_max = [&](float d) -> float {
        const unsigned db = __builtin_bit_cast(unsigned, d);
        const auto sw = __builtin_amdgcn_permlane32_swap(db, db, false, false);
        return fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
    };
    auto blk_max = [&](const f32x16& sx) -> float { return pair_max(own_max(sx)); };
    auto set_shift = [&](float m_new) {
        sh = m_new;
        const unsigned bits = __float_as_uint(-sh) >> 16;          // exact: sh is a bf16 value
        if (h) qf[KS - 1][0] = (qf[KS - 1][0] & 0xffff0000u) | bits;
    };
    auto read_v = [&](const char* st, int b, int kb, int s2) -> u32x4 {
        const int koff = (32 * kb + 16 * s2) * RS;
        int a0 = v_rd + koff + b * 64, a1 = a0 + 8 * RS;
        if (b == DB - 1) {
            a0 = v_last_real ? a0 : v_last_const;
            a1 = v_last_real ? a1 : v_last_const;
        }
        const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(st + a0));
        const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(st + a1));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        return u32x4{l2[0], l2[1], h2[0], h2[1]};
    };
    auto stage = [&](int t) -> const char* { return smem + (t & (NS - 1)) * STAGE; };

    // ---- prologue: tiles 0..2 on their way, S(block 0) with its shift, the K fragments of block 1
    // (the Q loads and the first three tiles share one memory round trip: the DMA goes out behind the Q loads, and the
    // compiler's own wait for Q -- a vmcnt(0), it cannot see the DMA -- covers the tiles as well. Stamped entry of a
    // workgroup with Q waited for first: 700-1 700 cycles to the segment bounds, 800 of address setup, 1 300-1 900 for
    // Q, 3 700-4 600 for issuing three tiles and landing the first)
#ifdef COGS_PIPE_STAMPS2
    PSTAMP();      // Q loads issued
#endif
#ifdef AV_LATE_ISSUE
    issue_tile_g(0, 0, FirstTilesFull{});
#endif
    // round 4: only Q and tile 0 go out before the first wait (10 vector-memory instructions per wave instead of 18: the
    // CU's vector-memory path takes one 1 KiB piece per ~60-115 cycles, so the 8 pieces of tiles 1 and 2 used to stand
    // between every wave and its first MFMA); tiles 1 and 2 follow behind the barrier, in order, so the counted waits of
    // the tile loop see the same queue as before. (p.early_prefetch: the old order, for A/B runs)
    const bool EARLY = p.early_prefetch != 0;
    if (EARLY) {
        if (R > 0 || nt > 1) issue_tile_g(1, 1, FirstTilesFull{});
        if (R > 0 || nt > 2) issue_tile_g(2, 2, FirstTilesFull{});
    }
#ifdef COGS_PIPE_STAMPS2
    PSTAMP();      // tiles issued
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qf[s]));      // Q loads complete before the loop (see attn_vit_kernel)
    __builtin_amdgcn_s_waitcnt(0xc07f);                               // lgkmcnt(0): the constant chunks
    PSTAMP();      // 1: Q and tile 0 landed
    __builtin_amdgcn_s_barrier();
    if (!EARLY) {
        if (R > 0 || nt > 1) issue_tile_g(1, 1, FirstTilesFull{});
        if (R > 0 || nt > 2) issue_tile_g(2, 2, FirstTilesFull{});
    }
    f32x16 sa, sb;
    u32x4 kf[KS];                                                     // K fragments of the NEXT block to be multiplied
    if (wave_active) {
        read_k(stage(0), 0, kf);
        qk_blk(kf, sa);
        if (len < 32) mask_blk(sa, len);
        float d = blk_max(sa);
        if (!(d > -INFINITY)) d = 0.f;
        set_shift(bf16_round(d));
        qk_blk(kf, sa);
        if (len < 32) mask_blk(sa, len);
        read_k(stage(0), 1, kf);
    }

    // sub-step j: consumes sc = S(j) - m, produces sn = S(j+1) - m from the fragments in kf, leaves the fragments of
    // block j+2 in kf. KIND 1: block j+1 is full; 3: block j+1 is the ragged last block; 0: j is the last block
    // SLOT >= 0: the ring slot of tile j >> 1 is known at compile time (the four-tile unrolled main loop), so every LDS
    // address below is a lane-constant register plus an immediate; -1: computed from j
    auto substep = [&](f32x16& sc, f32x16& sn, const int j, auto kind_tag, auto slot_tag) {
        constexpr int KIND = decltype(kind_tag)::value;
        constexpr int SLOT = decltype(slot_tag)::value;
        auto stage_of = [&](int jb) -> const char* {              // jb in {j, j+1, j+2}: at most one tile ahead of j's
            if constexpr (SLOT >= 0) return smem + ((SLOT + ((jb >> 1) - (j >> 1))) & (NS - 1)) * STAGE;
            else return stage(jb >> 1);
        };
        const char* st_c = stage_of(j);
        u32x4 vf[DB][2];
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) vf[b][s2] = read_v(st_c, b, j & 1, s2);
        __builtin_amdgcn_sched_barrier(0);        // the reads stay in front of phase 1 (hipcc sinks them to their uses otherwise)
        // phase 1
        if constexpr (KIND != 0) qk_blk(kf, sn);
        u32x4 pf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int w = 0; w < 4; ++w)
                pf[s2][w] = pack_bf2(__builtin_amdgcn_exp2f(sc[8 * s2 + 2 * w]), __builtin_amdgcn_exp2f(sc[8 * s2 + 2 * w + 1]));
        // phase 2 (the slot of block j+2 holds landed data whenever that block exists; a stale image otherwise, unused)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KIND != 0) read_k(stage_of(j + 2), (j + 2) & 1, kf);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int b = 0; b < DB; ++b)
                oacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[b][s2]), __builtin_bit_cast(bf16x8, pf[s2]),
                                                                 oacc[b], 0, 0, 0);
        if constexpr (KIND != 0) {
            const int valid_next = len - 32 * (j + 1);
            if constexpr (KIND == 3) mask_blk(sn, valid_next);
            // the test is wave-uniform (ANY lane above the threshold), so the common path does not combine the two lanes of a
            // row: 7 v_max3 + 1 v_max on the lane's own 16 scores; the exchange with the partner lane (v_mov, wait states,
            // v_permlane32_swap, two more v_max) only runs on the rare path. Round 5, with -fno-honor-nans for this file (no
            // canonicalising v_max in front of every chain): 17 -> 8 vector instructions per 32-key block on the port this
            // kernel is bound by (per tile and wave: 22 MFMA x 8 issue cycles + 32 v_exp x 8 + 16 v_cvt_pk + the maxima)
            const float d_own = own_max(sn);
#ifdef AV_ABL_NORESCALE      // timing experiment (tools/attn_abl.sh): what does the rare path's wave vote + branch cost?
            if (false) {
#else
            if (__any(d_own > RESCALE_THR)) {
#endif
                // rare (wave-uniform): move the reference of the rows that need it; O, now complete up to block j, is
                // multiplied by 2^-(m_new - sh) and S(j+1) is simply computed again with the new shift
                const float d = pair_max(d_own);
                const float m_new = d > RESCALE_THR ? bf16_round(sh + d) : sh;
                const float al = __builtin_amdgcn_exp2f(sh - m_new);
#pragma unroll
                for (int b = 0; b < DB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[b][r] *= al;
                set_shift(m_new);
                u32x4 kt[KS];
                read_k(stage_of(j + 1), (j + 1) & 1, kt);
                qk_blk(kt, sn);
                if constexpr (KIND == 3) mask_blk(sn, valid_next);
            }
        }
    };
#ifdef COGS_PHASE_STAMPS
    unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // slot_tag: ring slot of tile t when known at compile time, else -1; steady_tag: tiles t + 2 and t + 3 are known to exist (the
    // unrolled main loop), so neither the wait nor the issue is behind a run-time test
    auto tile_head = [&](const int t, auto slot_tag, auto steady_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
#ifdef AV_OLD_BRANCHES
        constexpr bool STEADY = false;
#else
        constexpr bool STEADY = decltype(steady_tag)::value;
#endif
        // outstanding, oldest first: tile t+1 (if any), tile t+2 (if any); tile t+1 must have landed
        unsigned long long h0 = 0, h1 = 0, h2 = 0, h3 = 0;
        PHT(h0);
        if constexpr (STEADY) wait_tiles(1); else wait_tiles(t + 2 < nt ? 1 : 0);
        PHT(h1);
        __builtin_amdgcn_s_barrier();     // K(t+1), V(t) visible to all; slot of tile t-1 no longer read by anyone
        PHT(h2);
#ifndef ABL_NOLOAD
        if constexpr (STEADY && R > 0) issue_tile_g(t + 3, (SLOT + 3) & (NS - 1), std::true_type{});      // a full tile by the choice of R
        else if (STEADY || t + 3 < nt) issue_tile(t + 3, SLOT >= 0 ? (SLOT + 3) & (NS - 1) : (t + 3) & (NS - 1));
#endif
        PHT(h3);
#ifdef COGS_PHASE_STAMPS
        ph_acc[0] += h1 - h0; ph_acc[1] += h2 - h1; ph_acc[2] += h3 - h2; ph_acc[3] += 1;
#endif
    };

    // the same with everything decided at compile time: WAIT1 = tile t + 2 exists (it may stay in flight), ISSUE = tile t + 3 exists
    // (ISSUE 2: ... and is known to be full, i.e. not the segment's last tile)
    auto tile_head_ct = [&](const int t, auto slot_tag, auto wait1_tag, auto issue_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int ISSUE = decltype(issue_tag)::value;
        if constexpr (decltype(wait1_tag)::value) wait_tiles(1); else wait_tiles(0);
        __builtin_amdgcn_s_barrier();
#ifndef ABL_NOLOAD
        if constexpr (ISSUE == 2) issue_tile_g(t + 3, (SLOT + 3) & (NS - 1), std::true_type{});
        else if constexpr (ISSUE == 1) issue_tile(t + 3, (SLOT + 3) & (NS - 1));
#endif
    };

    PSTAMP();      // 2: S(0) ready
#ifndef AV_OLD_BRANCHES
    // A wave whose 32 query rows all lie past the segment end (the ragged last query block) only stages and keeps the barriers: it
    // walks the tile heads here and leaves, so the waves that compute carry no `wave_active` test between a tile's barrier and its MFMAs
    if (!wave_active) {
        for (int ti = 0; ti < nt; ++ti) tile_head(ti, std::integral_constant<int, -1>{}, std::false_type{});
        return;
    }
#endif
    int t = 0;
    using Full = std::integral_constant<int, 1>;
#ifndef AV_NO_UNROLL4
    // four tiles per trip: t is a multiple of 4 here, so tile t + i sits in ring slot i (compile-time LDS addresses)
    // (R > 0: the trips end R tiles before the end; else: while blocks up to 2 t + 8 are full and t + 6 < nt, the tile the trip's
    // last head requests)
    for (; R > 0 ? t + R < nt : 2 * (t + 3) + 2 < nfull && t + 6 < nt; t += 4) {
#ifdef ABL_NOCOMPUTE
#define COGS_AV_ACTIVE false
#elif defined(AV_OLD_BRANCHES)
#define COGS_AV_ACTIVE wave_active
#else
#define COGS_AV_ACTIVE true
#endif
#ifdef COGS_PHASE_STAMPS
#define COGS_PH_ADD(a_, b_, c_) ph_acc[4] += (b_) - (a_); ph_acc[5] += (c_) - (b_);
#else
#define COGS_PH_ADD(a_, b_, c_)
#endif
#define COGS_AV_TILE(I)                                                                                      \
        tile_head(t + I, std::integral_constant<int, I>{}, std::true_type{});                                                  \
        if (COGS_AV_ACTIVE) {                                                                                \
            unsigned long long c0_ = 0, c1_ = 0, c2_ = 0;                                                    \
            PHT(c0_);                                                                                        \
            substep(sa, sb, 2 * (t + I), Full{}, std::integral_constant<int, I>{});                          \
            PHT(c1_);                                                                                        \
            substep(sb, sa, 2 * (t + I) + 1, Full{}, std::integral_constant<int, I>{});                      \
            PHT(c2_);                                                                                        \
            COGS_PH_ADD(c0_, c1_, c2_)                                                                       \
        }
        COGS_AV_TILE(0) COGS_AV_TILE(1) COGS_AV_TILE(2) COGS_AV_TILE(3)
#undef COGS_AV_TILE
    }
#endif
    if constexpr (R > 0) {
        // the last R tiles, t = nt - R (a multiple of 4): tile t + I in ring slot I & 3; NB blocks in all, block E from the end is the
        // last one (KIND 0), the one in front of it computes the last block's scores under its mask (KIND 3; a full last block
        // passes the mask unchanged), every other one is an ordinary sub-step
        constexpr int NB = 2 * R - (2 - NLB);
#define COGS_AV_END_BLK(I, KB, SC, SN)                                                                                          \
        if constexpr (NB - 1 - (2 * (I) + (KB)) >= 0) {                                                                          \
            constexpr int E = NB - 1 - (2 * (I) + (KB));                                                                         \
            substep(SC, SN, 2 * (t + (I)) + (KB), std::integral_constant<int, E == 0 ? 0 : E == 1 ? 3 : 1>{}, std::integral_constant<int, (I) & 3>{}); \
        }
#define COGS_AV_END(I)                                                                                                          \
        if constexpr ((I) < R) {                                                                                                 \
            tile_head_ct(t + (I), std::integral_constant<int, (I) & 3>{}, std::integral_constant<bool, ((I) + 2 < R)>{},         \
                         std::integral_constant<int, ((I) + 3 < R - 1) ? 2 : ((I) + 3 < R) ? 1 : 0>{});                          \
            COGS_AV_END_BLK(I, 0, sa, sb)                                                                                        \
            COGS_AV_END_BLK(I, 1, sb, sa)                                                                                        \
        }
        COGS_AV_END(0) COGS_AV_END(1) COGS_AV_END(2) COGS_AV_END(3) COGS_AV_END(4) COGS_AV_END(5) COGS_AV_END(6)
#undef COGS_AV_END
#undef COGS_AV_END_BLK
    }
    if constexpr (R == 0)
    for (; 2 * t + 2 < nfull; ++t) {              // blocks 2t+1 and 2t+2 are full
        tile_head(t, std::integral_constant<int, -1>{}, std::false_type{});
#ifndef ABL_NOCOMPUTE
        if (wave_active) {
            substep(sa, sb, 2 * t, Full{}, std::integral_constant<int, -1>{});
            substep(sb, sa, 2 * t + 1, Full{}, std::integral_constant<int, -1>{});
        }
#endif
    }
#ifdef COGS_PHASE_STAMPS
    if (COGS_AV_LINEAR_ID == 3000 && tid == 0) { for (int i = 0; i < 8; ++i) g_attn_stamps[i] = ph_acc[i]; }
#endif
    PSTAMP();      // 3: main loop done
#ifdef COGS_PHASE_STAMPS
    unsigned long long tl_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int tl_n = 0;
#endif
    if constexpr (R == 0)
    for (; t < nt; ++t) {                         // the ragged end: block kinds decided at run time (wave-uniform)
        unsigned long long q0_ = 0, q1_ = 0;
        PHT(q0_);
        tile_head(t, std::integral_constant<int, -1>{}, std::false_type{});
        PHT(q1_);
#ifdef COGS_PHASE_STAMPS
        tl_acc[0] += q1_ - q0_;
#endif
        if (!wave_active) continue;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int j = 2 * t + kb;
            if (j >= nblk) break;
            f32x16& sc = kb == 0 ? sa : sb;
            f32x16& sn = kb == 0 ? sb : sa;
            unsigned long long u0_ = 0, u1_ = 0;
            PHT(u0_);
            if (j + 1 < nfull) substep(sc, sn, j, Full{}, std::integral_constant<int, -1>{});
            else if (j + 1 < nblk) substep(sc, sn, j, std::integral_constant<int, 3>{}, std::integral_constant<int, -1>{});
            else substep(sc, sn, j, std::integral_constant<int, 0>{}, std::integral_constant<int, -1>{});
            PHT(u1_);
#ifdef COGS_PHASE_STAMPS
            if (tl_n < 6) tl_acc[1 + tl_n] = u1_ - u0_;
            ++tl_n;
#endif
        }
    }
#ifdef COGS_PHASE_STAMPS
    if (COGS_AV_LINEAR_ID == 3000 && tid == 0) { for (int i = 0; i < 8; ++i) g_tail_stamps[i] = tl_acc[i]; }
#endif
    PSTAMP();      // 4: tail done
    if (!wave_active) return;

    constexpr int LB = HD / 32, LR = HD % 32;
    constexpr int LH = (LR >> 2) & 1, LREG = (LR & 3) + 4 * (LR >> 3);
    float l = oacc[LB][LREG];
    l = __shfl(l, r32 + 32 * LH, 64);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
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
            const int d0 = 32 * b + 16 * gp + 8 * h;
            if (qok && d0 < HD) *reinterpret_cast<u32x4*>(orow + d0) = u32x4{(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};
        }
    }
    LIFE_NOW(life1);
    LIFE_ADD(0, life1 - life0); LIFE_ADD(1, 1);
#ifdef COGS_PIPE_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PSTAMP();      // 5: O stored
    if (COGS_AV_LINEAR_ID == 3000 && tid == 0) { for (int i = 0; i < 8; ++i) g_attn_stamps[i] = ps_t[i]; }
#endif
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
    p.head_stride = a.head_stride > 0 ? a.head_stride : 72;
    if (a.head_stride > 0 && (a.ldq != 72 || a.ldk != 72 || a.ldv != 72)) return COGS_E_INVALID;
    p.uniform_len = (a.uniform_seqlen > 0 && (long)a.uniform_seqlen * a.nseg == a.q_len && a.uniform_seqlen == a.max_seqlen) ? a.uniform_seqlen : 0;
    if (p.nseg > 65535 || p.heads > 65535 || (long)p.nqb * 8 > 0x7fffffffL) return COGS_E_INVALID;
    p.xcd_order = p.heads % 8 == 0;
    const dim3 grid = p.xcd_order ? dim3(p.nqb * 8, p.heads / 8, p.nseg) : dim3(p.nqb, p.heads, p.nseg);
    const int variant = (int)g_cogs_debug.attn_vit;     // 1: unpipelined (A/B runs)
    const int env_early = (int)g_cogs_debug.attn_vit_early;
    p.early_prefetch = env_early;
    if (variant == 1 || a.ldk != a.ldv) { g_cogs_debug.attn_last_kernel = 2; hipLaunchKernelGGL(attn_vit_kernel<72>, grid, dim3(256), 0, st, p); }
    else {
        g_cogs_debug.attn_last_kernel = a.head_stride > 0 ? 8 : 3;
        // one video (all segments alike, 4 tiles or more): the instantiation whose ragged end has this length's shape
        const int nt = (p.uniform_len + 63) >> 6;
        const int r = (g_cogs_debug.attn_vit_len != 0 && p.uniform_len > 0 && nt >= 4) ? 4 + ((nt - 4) & 3) : 0;
        const int nlb = p.uniform_len - 64 * (nt - 1) > 32 ? 2 : 1;
#define COGS_AV_LAUNCH(R_, B_) hipLaunchKernelGGL((attn_vit_pipe_kernel<72, R_, B_>), grid, dim3(256), 0, st, p)
        g_cogs_debug.attn_vit_last_end = r * 10 + (r ? nlb : 0);
        switch (r * 10 + (r ? nlb : 0)) {
            case 41: COGS_AV_LAUNCH(4, 1); break;   case 42: COGS_AV_LAUNCH(4, 2); break;
            case 51: COGS_AV_LAUNCH(5, 1); break;   case 52: COGS_AV_LAUNCH(5, 2); break;
            case 61: COGS_AV_LAUNCH(6, 1); break;   case 62: COGS_AV_LAUNCH(6, 2); break;
            case 71: COGS_AV_LAUNCH(7, 1); break;   case 72: COGS_AV_LAUNCH(7, 2); break;
            default: COGS_AV_LAUNCH(0, 0);
        }
#undef COGS_AV_LAUNCH
    }
    return COGS_LAUNCH_CHECK();
}
