// Fused attention for the ViT encoder (per-frame block-diagonal, hd 72) and the Qwen2
// prefill (causal GQA, hd 128).
//
// Replaces: flash_attn_varlen_func (model/modeling_videollama3_encoder.py:309-312), the
// eager softmax(QK^T/sqrt(d) + mask) path (:257-271, "global + same-frame bias" mode,
// parity only) and transformers' Qwen2 attention (GQA repeat_kv, causal mask, fp32 softmax).
//
// bf16 kernel (gfx950):
//   * one workgroup = 4 waves = 128 query rows of one (segment, head); K/V tiles of 64 keys
//     are register-staged (global loads issued before the compute of the previous tile,
//     LDS write after it) into one LDS buffer;
//   * scores are computed TRANSPOSED, S^T[key][q] = K.Q^T (K rows are the MFMA A operand),
//     with the key rows of each 32-key half permuted so that the S^T accumulator of a lane
//     is exactly the B-operand fragment of the following O^T[d][q] = V^T.P^T product: P never
//     leaves registers; the softmax row statistics are per-lane scalars (q = lane&15) and
//     need two cross-lane exchanges per tile;
//   * V stays row-major [key][d] in LDS and is consumed column-major through the hardware
//     transpose read ds_read_b64_tr_b16; K rows are 256 B with a chunk XOR so the
//     ds_read_b128 fragment reads are bank-conflict free;
//   * head dim 72 is zero-padded to 96 for QK^T (3 k-steps) and 80 for PV (5 d-tiles).
// fp32 / generic kernel: one wave per (query row, head), two passes; parity mode only.
#include "common.h"
#include "kernels.h"
#include "debug.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// PRE (deferred-maximum) softmax: a tile whose maximum exceeds the running reference by more than this many log2 units moves
// the reference BEFORE its exponentials instead of scaling (O, l) behind them (exp2 of the difference would overflow towards
// inf, and inf * 2^-dd = NaN); up to 2^64 a row sum of 64 keys fits fp32 and P is exact in bf16
constexpr float PRE_FAR = 64.f;
#ifdef COGS_NO_PRE_FAR          // A/B builds only (tools/build_alt.sh attn -DCOGS_NO_PRE_FAR): what the far test costs
#define COGS_PRE_FAR_ON false
#else
#define COGS_PRE_FAR_ON true
#endif

struct AttnArgs {
    const void* Q; const void* K; const void* V; void* O;
    long ldq, ldk, ldv, ldo;        // elements
    const int* cu;                  // [nseg+1] or null
    const int* row_lo;              // bias mode: same-segment key range per query row
    const int* row_hi;
    int q_len, kv_len;
    int hq, hkv;
    float scale_log2;               // softmax scale * log2(e)
    float bias_log2;                // additive same-segment bias * log2(e)
    int q_pos0;                     // causal: key j visible to query i iff j <= i + q_pos0
    int causal;
    int heavy_first;                // causal, one sequence: query tiles in descending order
    int prio_mode;                  // prompt kernel (COGS_ATTN_PRIO): 1 = s_setprio 1 around the MFMA blocks, 2 (default) = around the softmax: the two
                                    // waves of a SIMD, which belong to different workgroups, fall out of step, one's MFMAs beside the other's VALU
    int nsplit;                     // >1: keys split over blocks, partials go to part_o/part_ml
    int gqa_pack;                   // decode: the q-heads of one kv head are the 16 query columns
    int q_prescaled;                // Q already carries scale*log2(e): use the PRE kernels
    float* part_o;                  // [nsplit][q_len][hq][HD] unnormalised
    float* part_ml;                 // [nsplit][q_len][hq][2]  (running max (log2 domain), sum)
};

// max over the four lanes {li, li+16, li+32, li+48} (one query column), result in all four, without an LDS round
// trip: v_permlane16_swap(x, x) leaves [r0 r0 r2 r2] / [r1 r1 r3 r3] (odd rows of the first <-> even rows of the
// second operand), v_permlane32_swap(y, y) leaves [lo lo] / [hi hi]. ds_bpermute (what __shfl_xor compiles to) put
// two dependent ~100-cycle LDS latencies into every tile's softmax.
__device__ __forceinline__ float colgroup_max(float x) {
    const unsigned xb = __builtin_bit_cast(unsigned, x);
    const auto a = __builtin_amdgcn_permlane16_swap(xb, xb, false, false);
    const float y = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
    const unsigned yb = __builtin_bit_cast(unsigned, y);
    const auto b = __builtin_amdgcn_permlane32_swap(yb, yb, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}

__device__ __forceinline__ int k_swz(int row) { return (row & 3) | (((row >> 3) & 3) << 2); }

typedef __attribute__((ext_vector_type(16))) float f32x16;

// NQ = 16-row query sub-tiles per wave: 2 -> 4 waves x 32 rows (256 threads, 2 waves per SIMD at 2 workgroups per CU);
// 1 -> 8 waves x 16 rows (512 threads): half the accumulator / score registers per wave, so twice the waves per
// SIMD overlap each other's MFMA, VALU and LDS phases, at the price of reading every K/V fragment for 16 rows only.
// PRE = Q arrives pre-multiplied by scale*log2(e) (the QKV GEMM epilogue does it before its single rounding) and the
// softmax runs "deferred max": the score accumulator is initialised with -m_ref (the running maximum BEFORE this
// tile), so P' = exp2(acc) needs no per-score fma; when a tile raises the maximum by d > 0 the (O, l) pair is
// multiplied by 2^-d after its P'V product -- the same sums as the classic update, with 32 VALU ops less per tile
// and wave (the ViT shape is VALU-bound: 225 VALU instructions against 44 MFMAs per tile).
template <int HD, int NQ, bool PRE>
__global__ __launch_bounds__(128 * (4 / NQ), NQ == 1 ? 4 : ((PRE && HD == 72) ? 3 : 2)) void attn_fwd_bf16_kernel(AttnArgs p) {
    constexpr int NT = 128 * (4 / NQ);        // threads per workgroup (always 128 query rows)
    constexpr int KS = (HD + 31) / 32;        // QK^T k-steps
    constexpr int DT = (HD + 15) / 16;        // PV d-tiles
    constexpr int CH = HD / 8;                // 16-byte chunks per K/V row
    constexpr int NCH = 64 * CH;
    constexpr int PER = (NCH + NT - 1) / NT;
    constexpr int VS = (HD > 80) ? 288 : 160; // V row stride, == 32*odd bytes
    constexpr int VG = 8 * VS + 128;          // stride of a group of 8 V rows
    constexpr int K_LDS = 64 * 256;
    __shared__ __attribute__((aligned(16))) char smem[K_LDS + 8 * VG];
    char* const Ks = smem;
    char* const Vs = smem + K_LDS;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int seg = blockIdx.z;
    const int gsz = p.hq / p.hkv;
    const int kvh = p.gqa_pack ? blockIdx.y : blockIdx.y / gsz;
    const int split = blockIdx.x % p.nsplit;

    int qs = 0, qe = p.q_len, ks = 0, ke = p.kv_len;
    if (p.cu) { qs = p.cu[seg]; qe = p.cu[seg + 1]; ks = qs; ke = qe; }
    // causal prompts: the late (long) query tiles of a head are issued first, the short ones fill the tail of the launch
    const int qt = p.heavy_first ? (int)(gridDim.x / p.nsplit) - 1 - (int)(blockIdx.x / p.nsplit) : (int)(blockIdx.x / p.nsplit);
    const int q0 = qs + qt * 128;
    if (q0 >= qe) return;

    const bf16_t* Qp = reinterpret_cast<const bf16_t*>(p.Q);
    const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.K);
    const bf16_t* Vp = reinterpret_cast<const bf16_t*>(p.V);

    // zero the pad columns once (staging never writes them)
    if (HD % 32 != 0) {
        for (int id = tid; id < 64 * (16 - CH); id += NT) {
            const int row = id / (16 - CH), c = CH + id % (16 - CH);
            *reinterpret_cast<u32x4*>(Ks + row * 256 + ((c ^ k_swz(row)) << 4)) = u32x4{0, 0, 0, 0};
        }
    }
    // hd 72: the 8 pad columns of V hold [1, 0, ...]: row d = HD of O^T then accumulates sum_key P[key][q],
    // i.e. the softmax denominator comes out of the PV MFMAs for free (and is rescaled with O)
    constexpr bool SUM_IN_V = (HD % 16 != 0);
    if (SUM_IN_V) {
        for (int row = tid; row < 64; row += NT)
            *reinterpret_cast<u32x4*>(Vs + (row >> 3) * VG + (row & 7) * VS + CH * 16) = u32x4{0x00003f80u, 0, 0, 0};
    }

    // Q fragments (B operand of S^T = K.Q^T): lane (q = li, g) holds Q[q][32s + 8g + j]
    bf16x8 qf[NQ][KS];
    int qrow[NQ], qhead[NQ];
    bool qok[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        if (p.gqa_pack) {
            // one query token; column li of sub-tile 0 of wave 0 is q-head kvh*gsz + li
            qrow[qi] = q0;
            qhead[qi] = kvh * gsz + li;
            qok[qi] = (wid == 0 && qi == 0 && li < gsz);
        } else {
            qrow[qi] = q0 + wid * (16 * NQ) + qi * 16 + li;
            qhead[qi] = blockIdx.y;
            qok[qi] = qrow[qi] < qe;
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 32 * s + 8 * g;
            u32x4 v = {0, 0, 0, 0};
            if (qok[qi] && k < HD)
                v = *reinterpret_cast<const u32x4*>(Qp + (long)qrow[qi] * p.ldq + qhead[qi] * HD + k);
            qf[qi][s] = __builtin_bit_cast(bf16x8, v);
        }
    }

    int kend = ke;
    if (p.causal) kend = min(ke, ks + (q0 - qs) + (p.gqa_pack ? 0 : 127) + p.q_pos0 + 1);
    const int nt_all = (kend - ks + 63) / 64;
    const int t_begin = (int)((long)nt_all * split / p.nsplit);
    const int nt = (int)((long)nt_all * (split + 1) / p.nsplit);

    f32x4 oacc[DT][NQ];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) oacc[d][qi] = f32x4{0, 0, 0, 0};
    float m_run[NQ], l_run[NQ];
    float m_ref[NQ];          // PRE: finite stand-in of the running maximum (0 until the first tile set it)
    bool first_tile = true;   // PRE: the first processed tile subtracts its own maximum explicitly
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) { m_run[qi] = -INFINITY; l_run[qi] = 0.f; m_ref[qi] = 0.f; }

    int blo[NQ], bhi[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) { blo[qi] = 0; bhi[qi] = 0; }
    if (p.row_lo) {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi)
            if (qok[qi]) { blo[qi] = p.row_lo[qrow[qi]]; bhi[qi] = p.row_hi[qrow[qi]]; }
    }

    // register staging of the next K/V tile; per-thread chunk coordinates are loop invariant
    u32x4 kreg[PER], vreg[PER];
    int k_goff[PER], v_goff[PER];   // element offsets inside one tile (row < 64): 32 bits are plenty
    int st_row[PER], k_loff[PER], v_loff[PER];
    bool st_ok[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int id = tid + i * NT;
        st_ok[i] = id < NCH;
        const int row = st_ok[i] ? id / CH : 0, c = st_ok[i] ? id % CH : 0;
        st_row[i] = row;
        k_goff[i] = row * (int)p.ldk + kvh * HD + c * 8;
        v_goff[i] = row * (int)p.ldv + kvh * HD + c * 8;
        k_loff[i] = row * 256 + ((c ^ k_swz(row)) << 4);
        v_loff[i] = (row >> 3) * VG + (row & 7) * VS + c * 16;
    }
    auto load_tile = [&](int kt) {
        const int kbase = ks + kt * 64;
        const bf16_t* kb = Kp + (long)kbase * p.ldk;
        const bf16_t* vb = Vp + (long)kbase * p.ldv;
        if (kbase + 64 <= ke && p.gqa_pack) {
            // single-token decode: every K/V byte is read once per token by one workgroup -- non-temporal, so that the
            // cache rows do not displace what the step re-reads (the same change took 5-10 % off the weight-streaming GEMVs)
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                if (i + 1 < PER || NCH % NT == 0 || st_ok[i]) {
                    kreg[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + k_goff[i]));
                    vreg[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + v_goff[i]));
                }
            }
        } else if (kbase + 64 <= ke) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                if (i + 1 < PER || NCH % NT == 0 || st_ok[i]) {
                    kreg[i] = *reinterpret_cast<const u32x4*>(kb + k_goff[i]);
                    vreg[i] = *reinterpret_cast<const u32x4*>(vb + v_goff[i]);
                }
            }
        } else {   // partial tile: rows past the key range are zero filled
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                kreg[i] = u32x4{0, 0, 0, 0};
                vreg[i] = u32x4{0, 0, 0, 0};
                if (st_ok[i] && kbase + st_row[i] < ke) {
                    kreg[i] = *reinterpret_cast<const u32x4*>(kb + k_goff[i]);
                    vreg[i] = *reinterpret_cast<const u32x4*>(vb + v_goff[i]);
                }
            }
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (i + 1 < PER || NCH % NT == 0 || st_ok[i]) {
                *reinterpret_cast<u32x4*>(Ks + k_loff[i]) = kreg[i];
                *reinterpret_cast<u32x4*>(Vs + v_loff[i]) = vreg[i];
            }
        }
    };

    // per-lane LDS addresses
    // K fragment (A operand): row r = li -> key 32u + 8(r>>2) + 4t + (r&3), chunk 4s + g
    const int krow0 = 8 * (li >> 2) + (li & 3);
    // V transposed read: lane i of a 16-lane group supplies row k0 + (i>>2), cols d0 + 4(i&3)
    const int vrow_off = g * VG + (li >> 2) * VS + (li & 3) * 8;

    const bool wave_active = p.gqa_pack ? (wid == 0) : (q0 + wid * (16 * NQ) < qe);   // wave-uniform
    if (t_begin < nt) load_tile(t_begin);
    // one K/V tile; MASKED is a compile-time tag: interior tiles run a compare-free softmax body, only the
    // tiles that touch the key-range end, the causal diagonal or the bias mode evaluate masks. (A run-time
    // uniform branch gets if-converted by hipcc into one block that executes both bodies.)
    auto process_tile = [&](const int kt, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        __syncthreads();
        write_tile();
        __syncthreads();
        if (kt + 1 < nt) load_tile(kt + 1);
        // a wave whose 32 query rows all lie past the segment end (the ragged last q-block: 924 = 7*128 + 28)
        // only stages and synchronises; its SIMD time goes to the co-resident workgroup
        if (!wave_active) return;

        const int kbase = ks + kt * 64;
        // ragged last key tile: when the second 32-key half is entirely past the range its MFMAs are skipped
        // (wave-uniform; P of that half would be exactly zero)
        const bool second_half = !MASKED || kbase + 32 < kend;
        f32x4 sacc[4][NQ];
#pragma unroll
        for (int ut = 0; ut < 4; ++ut) {
            if (MASKED && ut >= 2 && !second_half) {
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) sacc[ut][qi] = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            const int krow = 32 * (ut >> 1) + 4 * (ut & 1) + krow0;
            const int ksw = k_swz(krow);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + krow * 256 + (((4 * s + g) ^ ksw) << 4));
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f};                               // C = 0 is an inline constant
                    if constexpr (PRE) { const float nm = -m_ref[qi]; c0 = f32x4{nm, nm, nm, nm}; }
                    sacc[ut][qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, kf), qf[qi][s], s == 0 ? c0 : sacc[ut][qi], 0, 0, 0);
                }
            }
        }

        // online softmax; lane holds keys kbase + 32u + 8g + 4t + reg of query column li.
        // Masking (key range, causal diagonal, same-segment bias) is only evaluated on tiles that need it:
        // the test is block-uniform, so interior tiles run a compare-free body (max, fma, v_exp, add).
        bf16x8 pf[2][NQ];
        float post_alpha[NQ];   // PRE: factor applied to (O, l) after this tile's PV product (1 = none)
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            float sv[4][4];
            post_alpha[qi] = 1.f;
            if constexpr (PRE) {
                // acc = S - m_ref (S already in log2 units). d = how far this tile's maximum exceeds m_ref.
                float d = -INFINITY;
                if constexpr (MASKED) {
                    const int qloc = qrow[qi] - qs;
#pragma unroll
                    for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = kbase + 32 * (ut >> 1) + 8 * g + 4 * (ut & 1) + r;
                            bool valid = key < ke;
                            if (p.causal) valid = valid && (key - ks) <= qloc + p.q_pos0;
                            const float sc = valid ? sacc[ut][qi][r] : -INFINITY;
                            sv[ut][r] = sc;
                            d = fmaxf(d, sc);
                        }
                } else {
#pragma unroll
                    // (an inline-asm v_max3_f32 would save the canonicalising v_max hipcc puts in front of fmaxf on MFMA
                    // results, but asm operands get no MFMA->VALU hazard nops: measured wrong maxima. Left to hipcc.)
                    for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { sv[ut][r] = sacc[ut][qi][r]; d = fmaxf(d, sacc[ut][qi][r]); }
                }
                d = colgroup_max(d);
                const bool saw_key = d != -INFINITY;
                if (first_tile) {
                    // m_ref was 0: make this tile's own maximum the reference (a row with no visible key keeps 0)
                    const float d0 = (d == -INFINITY) ? 0.f : d;
#pragma unroll
                    for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sv[ut][r] -= d0;
                    asm volatile("" ::: "memory");   // keep this a real (wave-uniform) branch, not a select chain
                    m_ref[qi] = d0;
                    d = 0.f;
                }
                if (COGS_PRE_FAR_ON && __any(d > PRE_FAR)) {
                    // far above the reference (rare, wave-uniform): exp2(score - OLD reference) would overflow towards inf, and the
                    // factor applied afterwards would turn that into inf * 0 = NaN. Move the reference of those rows NOW: (O, l) take
                    // 2^-dd, the scores are shifted, and the test behind the exponentials finds nothing left to do for them
                    const float dd = d > PRE_FAR ? d : 0.f;
                    const float al = __builtin_amdgcn_exp2f(-dd);
#pragma unroll
                    for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sv[ut][r] -= dd;
                    l_run[qi] *= al;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) oacc[dt][qi] *= al;
                    m_ref[qi] += dd;
                    d -= dd;
                }
                float psum = 0.f;
#pragma unroll
                for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(sv[ut][r]);   // exp2(-inf) = 0 for masked keys
                        sv[ut][r] = pv;
                        if (!SUM_IN_V) psum += pv;
                    }
                l_run[qi] += psum;
                if (__any(d > 0.f)) {        // this tile raised the maximum of some row of the wave
                    const float dd = fmaxf(d, 0.f);
                    post_alpha[qi] = __builtin_amdgcn_exp2f(-dd);
                    m_ref[qi] += dd;
                }
                if (saw_key || m_run[qi] != -INFINITY) m_run[qi] = m_ref[qi];   // what the split-KV combine reads
            } else {
            float mx = -INFINITY;
            if constexpr (MASKED) {
                const int qloc = qrow[qi] - qs;
#pragma unroll
                for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kbase + 32 * (ut >> 1) + 8 * g + 4 * (ut & 1) + r;
                        bool valid = key < ke;
                        if (p.causal) valid = valid && (key - ks) <= qloc + p.q_pos0;
                        float sc = sacc[ut][qi][r] * p.scale_log2;
                        if (p.row_lo && key >= blo[qi] && key < bhi[qi]) sc += p.bias_log2;
                        sc = valid ? sc : -INFINITY;
                        sv[ut][r] = sc;
                        mx = fmaxf(mx, sc);
                    }
            } else {
#pragma unroll
                for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[ut][qi][r]);
                mx *= p.scale_log2;   // scale > 0
            }
            mx = colgroup_max(mx);
            const float m_new = fmaxf(m_run[qi], mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            float psum = 0.f;
            if constexpr (MASKED) {
#pragma unroll
                for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(sv[ut][r] - m_use);
                        sv[ut][r] = pv;
                        if (!SUM_IN_V) psum += pv;
                    }
            } else {
#pragma unroll
                for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[ut][qi][r], p.scale_log2, -m_use));
                        sv[ut][r] = pv;
                        if (!SUM_IN_V) psum += pv;
                    }
            }
            // the running maximum rarely moves after the first tiles: skip the O rescale when no row of
            // this wave changed it (wave-uniform vote; alpha is exactly 1 in that case)
            if (__any(m_new != m_run[qi])) {
                const float alpha = __builtin_amdgcn_exp2f(m_run[qi] - m_use);
                l_run[qi] = l_run[qi] * alpha + psum;
#pragma unroll
                for (int d = 0; d < DT; ++d) oacc[d][qi] *= alpha;
            } else {
                l_run[qi] += psum;
            }
            m_run[qi] = m_new;
            }   // !PRE
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                u32x4 w;
                w[0] = pack_bf2(sv[2 * u][0], sv[2 * u][1]);
                w[1] = pack_bf2(sv[2 * u][2], sv[2 * u][3]);
                w[2] = pack_bf2(sv[2 * u + 1][0], sv[2 * u + 1][1]);
                w[3] = pack_bf2(sv[2 * u + 1][2], sv[2 * u + 1][3]);
                pf[u][qi] = __builtin_bit_cast(bf16x8, w);
            }
        }

        // O^T[d][q] += V^T[d][key] . P^T[key][q]
#pragma unroll
        for (int d = 0; d < DT; ++d) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (MASKED && u == 1 && !second_half) continue;
                const char* va = Vs + u * 4 * VG + vrow_off + d * 32;
                const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) i16x4*)(va));
                const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) i16x4*)(va + 4 * VS));
                u32x4 w;
                u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                w[0] = l2[0]; w[1] = l2[1]; w[2] = h2[0]; w[3] = h2[1];
                const bf16x8 vf = __builtin_bit_cast(bf16x8, w);
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi)
                    oacc[d][qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[u][qi], oacc[d][qi], 0, 0, 0);
            }
        }
        if constexpr (PRE) {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                if (__any(post_alpha[qi] != 1.f)) {
                    l_run[qi] *= post_alpha[qi];
#pragma unroll
                    for (int d = 0; d < DT; ++d) oacc[d][qi] *= post_alpha[qi];
                }
            }
            first_tile = false;
        }
    };

    // leading tiles [t_begin, t_mid) need no mask: fully inside the key range and left of the causal diagonal
    int t_mid = t_begin;
    if (!p.row_lo) {
        int full = (ke - ks) / 64;
        if (p.causal) {
            const int lim = (q0 - qs) + p.q_pos0 - 63;   // tile kt is unmasked iff 64*kt <= lim
            full = min(full, lim >= 0 ? lim / 64 + 1 : 0);
        }
        t_mid = max(t_begin, min(full, nt));
    }
    for (int kt = t_begin; kt < t_mid; ++kt) process_tile(kt, std::false_type{});
    for (int kt = t_mid; kt < nt; ++kt) process_tile(kt, std::true_type{});

    bf16_t* Op = reinterpret_cast<bf16_t*>(p.O);
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        float l = l_run[qi];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (SUM_IN_V) {
            // row d = HD of O^T lives in d-tile HD/16, lane group (HD%16)/4, register 0
            l = __shfl(oacc[HD / 16][qi][0], li + 16 * ((HD % 16) / 4), 64);
        }
        if (!qok[qi]) continue;
        if (p.nsplit > 1) {
            const long slot = ((long)split * p.q_len + qrow[qi]) * p.hq + qhead[qi];
            if (g == 0) { p.part_ml[slot * 2] = m_run[qi]; p.part_ml[slot * 2 + 1] = l; }
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int dd = 16 * d + 4 * g;
                if (dd < HD) *reinterpret_cast<f32x4*>(p.part_o + slot * HD + dd) = oacc[d][qi];
            }
        } else {
            const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int dd = 16 * d + 4 * g;
                if (dd < HD) {
                    f32x4 v = oacc[d][qi] * inv;
                    st4_f<bf16_t>(Op + (long)qrow[qi] * p.ldo + qhead[qi] * HD + dd, v);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Qwen2 prompt attention (hd 128, causal, GQA, pre-scaled Q, one or several sequences): the kernel above with the K/V
// staging taken out of the registers. Same products, same deferred-max softmax, same masks -- what changes is how a
// tile gets into LDS and how often the workgroup synchronises:
//   * K/V tiles go global -> LDS by global_load_lds_dwordx4 (eight 1 KiB pieces per wave and tile, no staging
//     registers, no ds_write, no load -> write dependency) into a DOUBLE buffer, one tile ahead, ONE barrier per tile
//     (the kernel above: one buffer, two barriers, 8 global loads + 8 ds_write_b128 per thread and tile);
//   * LDS-DMA writes a piece as the plain image of its lanes (lane L -> bytes 16 L .. 16 L + 15), so the bank swizzles
//     live on the SOURCE address: K keeps the chunk XOR of the kernel above (lane L of a piece fetches chunk
//     (L & 15) ^ k_swz(row)); V rows are 256 contiguous bytes with their own chunk XOR (v_swz: the eight (row & 3,
//     row bit 3) classes a 32-lane half of a transposing read touches land in eight different 32-byte bank groups);
//   * rows past the end of the key range are fetched from its last row (finite data: their scores are masked, their
//     probabilities exactly zero) instead of being zero filled.
// The DMA is issued from inline assembly for the reason given in attn_vit.hip (hipcc's wait insertion would drain the
// prefetch in front of every transposing read); cogstream_amd/build.py checks that nothing else in the kernel touches M0.
__device__ __forceinline__ int v_swz(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; }

// DEEP (round 5): the fragment reads of the interior tiles are ordered by hand. hipcc issues every K / V fragment
// read right before the MFMAs that consume it (ds_read, s_waitcnt, 2-4 MFMAs, ds_read, ...: prefetch distance 0-1), so a wave
// stands for an LDS round trip (~100+ cycles) every 32-64 cycles of MFMA and the two waves of a SIMD can only half hide
// each other -- the ~50 % MFMA-busy both this kernel and its ping-pong form show. sched_group_barrier pins an order with
// the reads 6 (K) / 8 (V) instructions ahead of their MFMAs; hipcc still derives the counted s_waitcnt lgkmcnt(N).
// DEEP also defers the running maximum (PREFILL_THR below). What bounds the kernel, measured on it
// (profiles/r5_prefill_attn_anatomy.txt): with every MFMA removed it still runs 1.20 of its 1.89 ms, and the MFMAs alone
// would take 0.75 ms -- the two ADD: SQ_VALU_MFMA_BUSY_CYCLES + the VALU's active cycles are ~95 % of the resident time,
// bank conflicts 0, LDS array 21 % busy, 2.4 % of wave time waiting for LDS.
constexpr float PREFILL_THR = 6.f;
// PRIO: the wave-priority scheme (debug switch attn_prio) as a compile-time choice -- 2 = raised in the softmax phase (shipped), -1 =
// read p.prio_mode at run time (A/B runs: four s_cmp / s_cbranch pairs per tile, next to the MFMA blocks)
template <int DEEP, int PRIO = -1>
__global__ __launch_bounds__(256, 2) void attn_prefill_dma_kernel(AttnArgs p) {
    constexpr int HD = 128, NQ = 2, NT = 256, KS = 4, DT = 8;
    constexpr int K_LDS = 64 * 256, V_LDS = 64 * 256, BUF = K_LDS + V_LDS;     // 32 KiB per tile
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int seg = blockIdx.z;
    const int gsz = p.hq / p.hkv;
    const int kvh = blockIdx.y / gsz;

    int qs = 0, qe = p.q_len, ks = 0, ke = p.kv_len;
    if (p.cu) { qs = p.cu[seg]; qe = p.cu[seg + 1]; ks = qs; ke = qe; }
    const int qt = p.heavy_first ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int q0 = qs + qt * 128;
    if (q0 >= qe) return;

    const bf16_t* Qp = reinterpret_cast<const bf16_t*>(p.Q);
    const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.K);
    const bf16_t* Vp = reinterpret_cast<const bf16_t*>(p.V);

    bf16x8 qf[NQ][KS];
    int qrow[NQ];
    bool qok[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        qrow[qi] = q0 + wid * 32 + qi * 16 + li;
        qok[qi] = qrow[qi] < qe;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 v = {0, 0, 0, 0};
            if (qok[qi]) v = *reinterpret_cast<const u32x4*>(Qp + (long)qrow[qi] * p.ldq + blockIdx.y * HD + 32 * s + 8 * g);
            qf[qi][s] = __builtin_bit_cast(bf16x8, v);
        }
    }
    const int kend = min(ke, ks + (q0 - qs) + 127 + p.q_pos0 + 1);      // causal
    const int nt = (kend - ks + 63) / 64;

    f32x4 oacc[DT][NQ];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) oacc[d][qi] = f32x4{0, 0, 0, 0};
    float l_run[NQ], m_ref[NQ];
    bool first_tile_rt = true;      // round-4 form (DEEP = 0) only
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) { l_run[qi] = 0.f; m_ref[qi] = 0.f; }

    // ---- staging by LDS-DMA: wave w issues pieces w, w + 4, w + 8, w + 12 (4 tile rows each) of K and of V
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    int st_row[4], k_off[4], v_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * (wid + 4 * i) + (lane >> 4);
        st_row[i] = row;
        k_off[i] = (row * (int)p.ldk + kvh * HD + ((lane & 15) ^ k_swz(row)) * 8) * 2;     // bytes from the tile's first row
        v_off[i] = (row * (int)p.ldv + kvh * HD + ((lane & 15) ^ v_swz(row)) * 8) * 2;
    }
    auto uniform_ptr = [](const bf16_t* q) -> const bf16_t* {
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const bf16_t*)(((unsigned long long)hi << 32) | lo);
    };
    auto dma16 = [&](const bf16_t* base, int off_bytes, unsigned lds) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(lds), "v"(off_bytes), "s"(base) : "memory");
    };
    // all 8 pieces of a tile go out together at the top of an iteration. (Round 5 also tried the V pieces behind the QK^T
    // block, away from the moment all 8 waves of the CU issue at once: 1.76 -> 1.86 ms -- a piece issued between
    // fragment reads and MFMAs costs more than one issued beside other pieces. The pieces' issue, not the wait for them
    // nor the barrier, is what the staging costs: profiles/r5_prefill_attn_anatomy.txt)
    auto issue_tile = [&](int kt) {
        const int kbase = ks + kt * 64;
        const int valid = ke - kbase;                               // >= 1
        const bf16_t* kb = uniform_ptr(Kp + (long)kbase * p.ldk);
        const bf16_t* vb = uniform_ptr(Vp + (long)kbase * p.ldv);
        const unsigned st = __builtin_amdgcn_readfirstlane(smem_lds + (kt & 1) * BUF);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ko = k_off[i], vo = v_off[i];
            if (valid < 64) {                                        // rows past the key range repeat its last row
                const int back = st_row[i] - min(st_row[i], valid - 1);
                ko -= back * (int)p.ldk * 2;
                vo -= back * (int)p.ldv * 2;
            }
            dma16(kb, ko, st + (wid + 4 * i) * 1024);
            dma16(vb, vo, st + K_LDS + (wid + 4 * i) * 1024);
        }
    };

    // per-lane LDS read addressing
    const int krow0 = 8 * (li >> 2) + (li & 3);
    // V transposed read of (u, d): 16-lane group g reads rows 32u + 8g + (li >> 2) [+4], bytes 32 d + 8 (li & 3) .. +7
    const int vr = 8 * g + (li >> 2);                 // + 32u (+4): v_swz is the same for all of them (bits 0, 1, 3 of vr)
    const int v_sw = v_swz(vr);
    const int v_base = vr * 256 + ((li & 3) & 1) * 8;   // + 32u*256 (+4*256) + (((2d + ((li&3)>>1)) ^ v_sw) << 4)
    const int v_ch = (li & 3) >> 1;

    const bool wave_active = q0 + wid * 32 < qe;

    // first_tag: the wave's first tile (kt == 0: it subtracts its own maximum explicitly) -- a compile-time copy, so that the
    // other tiles carry no `first tile?` test (DEEP only; the round-4 form keeps the run-time flag)
    auto process_tile = [&](const int kt, auto masked_tag, auto first_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr bool FIRST_CT = decltype(first_tag)::value;
        const bool first_tile = DEEP ? FIRST_CT : first_tile_rt;
        const char* Ks = smem + (kt & 1) * BUF;
        const char* Vs = Ks + K_LDS;
        if (!wave_active) return;
        const int kbase = ks + kt * 64;
        const bool second_half = !MASKED || kbase + 32 < kend;
        f32x4 sacc[4][NQ];
        { const int pm = PRIO >= 0 ? PRIO : p.prio_mode; if (pm == 1) __builtin_amdgcn_s_setprio(1); else if (pm == 2) __builtin_amdgcn_s_setprio(0); }
#pragma unroll
        for (int ut = 0; ut < 4; ++ut) {
            if (MASKED && ut >= 2 && !second_half) {
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) sacc[ut][qi] = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            const int krow = 32 * (ut >> 1) + 4 * (ut & 1) + krow0;
            const int ksw = k_swz(krow);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + krow * 256 + (((4 * s + g) ^ ksw) << 4));
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    const float nm = -m_ref[qi];
                    const f32x4 c0 = f32x4{nm, nm, nm, nm};
#ifdef PF_ABL_NOQK
                    if (s == 0) sacc[ut][qi] = c0; else sacc[ut][qi][0] += __builtin_bit_cast(float, kf[0] & 0x3f000000u);
#else
                    sacc[ut][qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, kf), qf[qi][s], s == 0 ? c0 : sacc[ut][qi], 0, 0, 0);
#endif
                }
            }
        }
        #if !defined(PF_ABL_NOQK) && !defined(PF_ABL_NOPV)
        if constexpr (DEEP && !MASKED) {          // 16 reads, 32 MFMAs: 6 reads, then a read per pair of MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        }
#endif
        { const int pm = PRIO >= 0 ? PRIO : p.prio_mode; if (pm == 1) __builtin_amdgcn_s_setprio(0); else if (pm == 2) __builtin_amdgcn_s_setprio(1); }
        bf16x8 pf[2][NQ];
        float post_alpha[NQ];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            float sv[4][4];
            post_alpha[qi] = 1.f;
            float d = -INFINITY;
            if constexpr (MASKED) {
                const int qloc = qrow[qi] - qs;
#pragma unroll
                for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kbase + 32 * (ut >> 1) + 8 * g + 4 * (ut & 1) + r;
                        const bool valid = key < ke && (key - ks) <= qloc + p.q_pos0;
                        const float sc = valid ? sacc[ut][qi][r] : -INFINITY;
                        sv[ut][r] = sc;
                        d = fmaxf(d, sc);
                    }
            } else {
#pragma unroll
                for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sv[ut][r] = sacc[ut][qi][r]; d = fmaxf(d, sacc[ut][qi][r]); }
            }
#ifdef PF_ABL_NOMAX
            d = 0.f;
#else
            if (!DEEP || first_tile) d = colgroup_max(d);       // DEEP: d stays this lane's own maximum (4 of the row's 16 keys per block)
#endif
            if (first_tile) {
                const float d0 = (d == -INFINITY) ? 0.f : d;
#pragma unroll
                for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sv[ut][r] -= d0;
                asm volatile("" ::: "memory");
                m_ref[qi] = d0;
                d = 0.f;
            }
            // DEEP (the shipped form): the reference only moves when a score exceeds it by 2^PREFILL_THR (P <= 2^6 until then: exact in
            // fp32 sums, the same relative precision in bf16); the test is wave-uniform on the lanes' OWN maxima, so the two lane
            // exchanges of colgroup_max and the rescale of O leave the common path (they ran in ~25 % of the tiles). Round 6: the move
            // happens HERE, in front of the exponentials, for every threshold crossing -- one test per tile as before, and a key far
            // above the reference can no longer make exp2 overflow before a deferred factor is applied (inf * 2^-dd = NaN).
            // Round-4 form (DEEP = 0, A/B only): the deferred factor behind the PV product stays; only a crossing of more than
            // 2^PRE_FAR takes this path.
            constexpr float MOVE_THR = DEEP ? PREFILL_THR : PRE_FAR;
            if (COGS_PRE_FAR_ON && __any(d > MOVE_THR)) {
                const float dr = DEEP ? colgroup_max(d) : d;
                const float dd = dr > MOVE_THR ? dr : 0.f;
                const float al = __builtin_amdgcn_exp2f(-dd);
#pragma unroll
                for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sv[ut][r] -= dd;
                l_run[qi] *= al;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) oacc[dt][qi] *= al;
                m_ref[qi] += dd;
                d -= dd;
            }
            float psum = 0.f;
#pragma unroll
            for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#ifdef PF_ABL_NOEXP
                    const float pv = sv[ut][r];
#else
                    const float pv = __builtin_amdgcn_exp2f(sv[ut][r]);
#endif
                    sv[ut][r] = pv;
                    psum += pv;
                }
            l_run[qi] += psum;
            if constexpr (DEEP) {
                // (moved in front of the exponentials, above)
            } else if (__any(d > 0.f)) {
                const float dd = fmaxf(d, 0.f);
                post_alpha[qi] = __builtin_amdgcn_exp2f(-dd);
                m_ref[qi] += dd;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                u32x4 w;
                w[0] = pack_bf2(sv[2 * u][0], sv[2 * u][1]);
                w[1] = pack_bf2(sv[2 * u][2], sv[2 * u][3]);
                w[2] = pack_bf2(sv[2 * u + 1][0], sv[2 * u + 1][1]);
                w[3] = pack_bf2(sv[2 * u + 1][2], sv[2 * u + 1][3]);
                pf[u][qi] = __builtin_bit_cast(bf16x8, w);
            }
        }
        { const int pm = PRIO >= 0 ? PRIO : p.prio_mode; if (pm == 1) __builtin_amdgcn_s_setprio(1); else if (pm == 2) __builtin_amdgcn_s_setprio(0); }
#pragma unroll
        for (int d = 0; d < DT; ++d) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (MASKED && u == 1 && !second_half) continue;
                const char* va = Vs + u * 32 * 256 + v_base + ((((2 * d + v_ch) ^ v_sw)) << 4);
                const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(va));
                const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(va + 4 * 256));
                u32x4 w;
                u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                w[0] = l2[0]; w[1] = l2[1]; w[2] = h2[0]; w[3] = h2[1];
                const bf16x8 vf = __builtin_bit_cast(bf16x8, w);
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi)
#ifdef PF_ABL_NOPV
                    oacc[d][qi][0] += __builtin_bit_cast(float, (__builtin_bit_cast(u32x4, vf)[0] ^ __builtin_bit_cast(u32x4, pf[u][qi])[0]) & 0x3f000000u);
#else
                    oacc[d][qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[u][qi], oacc[d][qi], 0, 0, 0);
#endif
            }
        }
        #if !defined(PF_ABL_NOQK) && !defined(PF_ABL_NOPV)
        if constexpr (DEEP && !MASKED) {          // 32 transposed reads, 32 MFMAs: 8 reads lead (beside the exponentials)
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
#endif
        if constexpr (!DEEP) {        // the deferred factor exists in the round-4 form only
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            if (__any(post_alpha[qi] != 1.f)) {
                l_run[qi] *= post_alpha[qi];
#pragma unroll
                for (int d = 0; d < DT; ++d) oacc[d][qi] *= post_alpha[qi];
            }
        }
        }
        first_tile_rt = false;
    };


    // (Round 5 also built this loop with interior tiles as two 32-key halves pipelined INSIDE the tile -- S(h1) beside softmax(h0),
    // PV(h0) beside softmax(h1), pinned by sched_group_barrier -- and, as attn_prefill_sp_kernel, with S(t+1) beside softmax(t)
    // across tiles. Both correct at full size; 1.696 and 2.15 ms against 1.680 / 1.64: the matrix pipe and the VALU do not
    // overlap in this kernel whatever the order. Measurements and per-block stamps: profiles/r5_prefill_attn_anatomy.txt.)

    // leading tiles [0, t_mid) need no mask: fully inside the key range and left of the causal diagonal
    int t_mid;
    {
        int full = (ke - ks) / 64;
        const int lim = (q0 - qs) + p.q_pos0 - 63;
        full = min(full, lim >= 0 ? lim / 64 + 1 : 0);
        t_mid = max(0, min(full, nt));
    }
    issue_tile(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qf[qi][s]));     // Q loads complete before the loop
    __builtin_amdgcn_s_barrier();
    auto tile_tail = [&](int kt) {
        // tile kt + 1 (issued at the top of this iteration) has landed for this wave; after the barrier for every wave,
        // and nobody reads tile kt's buffer any more -- it is the one tile kt + 2 goes into
#ifndef PF_ABL_NOSYNC
#ifndef PF_ABL_NOWAIT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifndef PF_ABL_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
#endif
    };
    int kt = 0;
    if (nt > 0) {                                   // tile 0, the copy that sets the reference
        if (1 < nt) issue_tile(1);
        if (t_mid > 0) process_tile(0, std::false_type{}, std::true_type{});
        else process_tile(0, std::true_type{}, std::true_type{});
        tile_tail(0);
        kt = 1;
    }
    for (; kt < t_mid; ++kt) {
#ifndef PF_ABL_NOSYNC
        if (kt + 1 < nt) issue_tile(kt + 1);
#endif
        process_tile(kt, std::false_type{}, std::false_type{});
        tile_tail(kt);
    }
    for (; kt < nt; ++kt) {
        if (kt + 1 < nt) issue_tile(kt + 1);
        process_tile(kt, std::true_type{}, std::false_type{});
        tile_tail(kt);
    }

    bf16_t* Op = reinterpret_cast<bf16_t*>(p.O);
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        float l = l_run[qi];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (!qok[qi]) continue;
        const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            f32x4 v = oacc[d][qi] * inv;
            st4_f<bf16_t>(Op + (long)qrow[qi] * p.ldo + blockIdx.y * HD + 16 * d + 4 * g, v);
        }
    }
}

// combine the key-split partials: one 256-thread block per (query row, head). The split weights are computed
// once (thread s owns split s), then the [nsplit, HD] partial rows are summed with independent loads:
// 256/HDP split-groups run in parallel and meet in LDS.
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ part_o,
                                                           const float* __restrict__ part_ml, int nsplit, int q_len,
                                                           int hq, int HD, bf16_t* __restrict__ O, long ldo) {
    __shared__ float w_sh[128];
    __shared__ float red[256];
    __shared__ float l_sh[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int q = blockIdx.x, h = blockIdx.y;
    // weights: w_s = 2^(m_s - M), L = sum l_s w_s   (nsplit <= 128)
    float m = -INFINITY, l = 0.f;
    if (tid < nsplit) {
        const long slot = ((long)tid * q_len + q) * hq + h;
        m = part_ml[slot * 2];
        l = part_ml[slot * 2 + 1];
    }
    float M = wave_max(m);
    if (lane == 0) l_sh[wid] = M;
    __syncthreads();
    M = fmaxf(fmaxf(l_sh[0], l_sh[1]), fmaxf(l_sh[2], l_sh[3]));
    const float w = (m == -INFINITY) ? 0.f : exp2f(m - M);
    if (tid < 128) w_sh[tid] = w;
    float lw = wave_sum(l * w);
    __syncthreads();
    if (lane == 0) l_sh[wid] = lw;
    __syncthreads();
    const float L = l_sh[0] + l_sh[1] + l_sh[2] + l_sh[3];
    // o[d] = sum_s w_s part_o[s][d]: thread (d, group) strides the splits by the number of groups
    const int HDP = HD <= 64 ? 64 : (HD <= 128 ? 128 : 256);
    const int groups = 256 / HDP;
    const int d = tid % HDP, grp = tid / HDP;
    float acc = 0.f;
    if (d < HD) {
#pragma unroll 4
        for (int s = grp; s < nsplit; s += groups)
            acc += part_o[(((long)s * q_len + q) * hq + h) * HD + d] * w_sh[s];
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < HDP && tid < HD) {
        float o = 0.f;
        for (int g2 = 0; g2 < groups; ++g2) o += red[g2 * HDP + tid];
        O[(long)q * ldo + h * HD + tid] = f2bf(L > 0.f ? o / L : 0.f);
    }
}

// The same combine with four times the blocks and four times the loads in flight (round 5; head_dim % 32 == 0): one
// 256-thread block per (query row, head, 32-column slice); thread (d = tid & 31, group = tid >> 5) strides the splits by 8,
// so the 61 splits of a 15k-token context are 8 INDEPENDENT loads per thread (one round trip to the partials, which sit in
// other XCDs' L2s or beyond) where the kernel above makes 31 dependent-in-batches-of-4 ones: 6.1 -> ~3.5 us per layer of
// a decode step, 28 layers per token.
__global__ __launch_bounds__(256) void attn_combine32_kernel(const float* __restrict__ part_o,
                                                             const float* __restrict__ part_ml, int nsplit, int q_len,
                                                             int hq, int HD, bf16_t* __restrict__ O, long ldo) {
    __shared__ float w_sh[128];
    __shared__ float red[8][32];
    __shared__ float l_sh[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int q = blockIdx.x, h = blockIdx.y, d = blockIdx.z * 32 + (tid & 31), grp = tid >> 5;
    float m = -INFINITY, l = 0.f;
    if (tid < nsplit) {
        const long slot = ((long)tid * q_len + q) * hq + h;
        m = part_ml[slot * 2];
        l = part_ml[slot * 2 + 1];
    }
    // this thread's partial rows, requested before the weights are known (independent of them)
    float po[16];
    const float* pb = part_o + ((long)q * hq + h) * HD + d;
    const long sstride = (long)q_len * hq * HD;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int sidx = grp + 8 * i;
        po[i] = sidx < nsplit ? pb[sidx * sstride] : 0.f;
    }
    float M = wave_max(m);
    if (lane == 0) l_sh[wid] = M;
    __syncthreads();
    M = fmaxf(fmaxf(l_sh[0], l_sh[1]), fmaxf(l_sh[2], l_sh[3]));
    const float w = (m == -INFINITY) ? 0.f : exp2f(m - M);
    if (tid < 128) w_sh[tid] = w;
    float lw = wave_sum(l * w);
    __syncthreads();
    if (lane == 0) l_sh[wid] = lw;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int sidx = grp + 8 * i;
        if (sidx < nsplit) acc += po[i] * w_sh[sidx];
    }
    red[grp][tid & 31] = acc;
    __syncthreads();
    const float L = l_sh[0] + l_sh[1] + l_sh[2] + l_sh[3];
    if (tid < 32) {
        float o = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 8; ++g2) o += red[g2][tid];
        O[(long)q * ldo + h * HD + d] = f2bf(L > 0.f ? o / L : 0.f);
    }
}

// Generic reference-precision kernel: one wave per (query row, head); fp32 math.
template <typename T>
__global__ __launch_bounds__(64) void attn_rowwise_kernel(AttnArgs p, int HD, int nseg) {
    const int lane = threadIdx.x;
    const int q = blockIdx.x, head = blockIdx.y;
    const int kvh = head / (p.hq / p.hkv);
    int qs = 0, ks = 0, ke = p.kv_len;
    if (p.cu) {
        // binary search the segment of row q
        int lo = 0, hi = nseg;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p.cu[mid] <= q) lo = mid; else hi = mid; }
        qs = p.cu[lo]; ks = qs; ke = p.cu[lo + 1];
    }
    const T* Qp = reinterpret_cast<const T*>(p.Q) + (long)q * p.ldq + head * HD;
    const T* Kp = reinterpret_cast<const T*>(p.K) + kvh * HD;
    const T* Vp = reinterpret_cast<const T*>(p.V) + kvh * HD;
    __shared__ float qsh[256];
    for (int d = lane; d < HD; d += 64) qsh[d] = ld_f<T>(Qp + d);
    __syncthreads();
    int kend = ke;
    if (p.causal) kend = min(ke, ks + (q - qs) + p.q_pos0 + 1);
    int blo = 0, bhi = 0;
    if (p.row_lo) { blo = p.row_lo[q]; bhi = p.row_hi[q]; }
    const float LOG2E = 1.4426950408889634f;
    const float scale = p.scale_log2 / LOG2E, bias = p.bias_log2 / LOG2E;
    // pass 1: row max and sum
    float m = -INFINITY, l = 0.f;
    for (int j = ks + lane; j < kend; j += 64) {
        float s = 0.f;
        for (int d = 0; d < HD; ++d) s += qsh[d] * ld_f<T>(Kp + (long)j * p.ldk + d);
        s *= scale;
        if (p.row_lo && j >= blo && j < bhi) s += bias;
        const float mn = fmaxf(m, s);
        l = l * expf(m - mn) + expf(s - mn);
        m = mn;
    }
    const float mall = wave_max(m);
    l = (m == -INFINITY) ? 0.f : l * expf(m - mall);
    const float lall = wave_sum(l);
    const float inv = lall > 0.f ? 1.f / lall : 0.f;
    // pass 2: O[d] = sum_j p_j V[j][d]; lanes own d, d+64, ...
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = ks; j0 < kend; j0 += 64) {
        const int j = j0 + lane;
        float pj = 0.f;
        if (j < kend) {
            float s = 0.f;
            for (int d = 0; d < HD; ++d) s += qsh[d] * ld_f<T>(Kp + (long)j * p.ldk + d);
            s *= scale;
            if (p.row_lo && j >= blo && j < bhi) s += bias;
            pj = expf(s - mall) * inv;
        }
        const int cnt = min(64, kend - j0);
        for (int jj = 0; jj < cnt; ++jj) {
            const float pb = __shfl(pj, jj, 64);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = lane + 64 * e;
                if (d < HD) o[e] += pb * ld_f<T>(Vp + (long)(j0 + jj) * p.ldv + d);
            }
        }
    }
    T* Op = reinterpret_cast<T*>(p.O) + (long)q * p.ldo + head * HD;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int d = lane + 64 * e;
        if (d < HD) st_f<T>(Op + d, o[e]);
    }
}

}  // namespace

int cogs_k_attention(hipStream_t st, const CogsAttn& a) {
    if (a.q_len <= 0 || a.hq <= 0 || a.hkv <= 0 || a.hq % a.hkv != 0) return COGS_E_INVALID;
    if (a.head_dim > 256 || a.head_dim % 8 != 0) return COGS_E_UNSUPPORTED;
    AttnArgs p;
    p.Q = a.Q; p.K = a.K; p.V = a.V; p.O = a.O;
    p.ldq = a.ldq; p.ldk = a.ldk; p.ldv = a.ldv; p.ldo = a.ldo;
    p.cu = a.cu_seqlens; p.row_lo = a.row_lo; p.row_hi = a.row_hi;
    p.q_len = a.q_len; p.kv_len = a.kv_len; p.hq = a.hq; p.hkv = a.hkv;
    const float LOG2E = 1.4426950408889634f;
    p.scale_log2 = a.q_prescaled ? 1.0f : a.scale * LOG2E;   // prescaled Q: scores are already in log2 units
    p.bias_log2 = a.bias * LOG2E;
    p.q_pos0 = a.q_pos0; p.causal = a.causal;
    const int nseg = a.cu_seqlens ? a.nseg : 1;
    p.nsplit = 1; p.gqa_pack = 0; p.q_prescaled = 0; p.part_o = nullptr; p.part_ml = nullptr;
    const bool env_light_first = g_cogs_debug.attn_light_first == 1;   // A/B runs only
    p.heavy_first = (a.causal && !a.cu_seqlens && a.q_len > 128 && !env_light_first) ? 1 : 0;
    const int env_prio = (int)g_cogs_debug.attn_prio;   // in-run A/B at 15 395 tokens: 0 / 1 / 2 = 1.90-1.91 / 1.88-1.89 / 1.83 ms per layer
    p.prio_mode = env_prio;
    // the encoder's production shape (per-frame segments, hd 72, pre-scaled Q, no masks) has its own kernel
    const bool env_old_vit = g_cogs_debug.attn_vit == 0;   // A/B runs only
    if (a.dtype == COGS_DT_BF16 && !a.force_rowwise && a.head_dim == 72 && a.q_prescaled && a.cu_seqlens && !a.row_lo &&
        !a.causal && a.nsplit <= 1 && a.hq == a.hkv && a.ldo % 8 == 0 && !env_old_vit)
        return cogs_k_attention_vit(st, a);
    if (a.head_stride > 0) return COGS_E_UNSUPPORTED;      // head-major Q / K / V: the encoder's production kernel only
    if (a.dtype == COGS_DT_BF16 && !a.force_rowwise && (a.head_dim == 72 || a.head_dim == 128)) {
        if (a.ldq % 8 || a.ldk % 8 || a.ldv % 8 || a.ldo % 4) return COGS_E_INVALID;
        const int max_len = a.cu_seqlens ? a.max_seqlen : a.q_len;
        int qtiles = (max_len + 127) / 128;
        int gy = a.hq;
        if (a.nsplit > 1) {
            if (a.cu_seqlens || a.row_lo || a.nsplit > 128) return COGS_E_UNSUPPORTED;
            const size_t need = (size_t)a.nsplit * a.q_len * a.hq * (a.head_dim + 2) * sizeof(float);
            if (!a.ws || a.ws_bytes < need) return COGS_E_WORKSPACE;
            p.nsplit = a.nsplit;
            p.part_o = (float*)a.ws;
            p.part_ml = p.part_o + (size_t)a.nsplit * a.q_len * a.hq * a.head_dim;
        }
        if (a.q_len == 1 && a.hq / a.hkv <= 16 && !a.cu_seqlens) { p.gqa_pack = 1; gy = a.hkv; }
        dim3 grid(qtiles * p.nsplit, gy, nseg);
        const int env_nq = (int)g_cogs_debug.attn_nq;
        // 8 waves x 16 rows: measured faster for hd 128 (no spills, 4 waves per SIMD: causal prefill 2.58 -> 2.29 ms
        // at 15k tokens), slower for hd 72 (0.49 -> 0.52 ms: its fragment reads make the LDS the busiest unit)
        // (with pre-scaled Q the 4-wave hd 128 kernel needs 243 VGPRs and no longer spills: 2.34 -> 2.26 ms, so the
        // light variant is only the default for the classic softmax)
        const bool light = env_nq ? env_nq == 1 : (a.head_dim == 128 && !a.q_prescaled);
        const bool pre = a.q_prescaled && !a.row_lo;
        if (a.q_prescaled && a.row_lo) return COGS_E_UNSUPPORTED;   // the bias mode is parity-only and unscaled
        p.q_prescaled = pre;
#define COGS_ATTN_LAUNCH(HD_, NQ_, NT_)                                                                          \
    do {                                                                                                         \
        if (pre) hipLaunchKernelGGL((attn_fwd_bf16_kernel<HD_, NQ_, true>), grid, dim3(NT_), 0, st, p);          \
        else hipLaunchKernelGGL((attn_fwd_bf16_kernel<HD_, NQ_, false>), grid, dim3(NT_), 0, st, p);             \
    } while (0)
        // the generated tokens' attention (one query row, key-split) has its own kernel: every wave owns whole tiles
        const bool env_old_dec = g_cogs_debug.attn_decode == 0;   // A/B runs only
        const bool env_no_dma = g_cogs_debug.attn_prefill_dma == 0;   // A/B runs only
        // (every causal pre-scaled hd-128 call, whatever its length: a prompt continued behind a cached prefix with fewer than 128
        // new rows must run the same per-row arithmetic as the full prompt -- PrefixKV's transparency, tests/test_gpu_models.py --
        // and since round 5 this kernel's running maximum is no longer the general kernel's)
        if (a.head_dim == 128 && pre && a.causal && p.nsplit == 1 && !p.gqa_pack && !a.row_lo && !env_no_dma && a.ldo % 8 == 0) {
            // (two more forms were built, tested at full size and measured slower -- a ping-pong form, 1.89 vs 1.81-1.86 ms per layer
            // at 15 395 tokens, and 64 query rows per wave, 2.94 vs 1.97 ms: tools/experiments/attn_prefill_variants.hip)
            g_cogs_debug.attn_last_kernel = 5;
            if (g_cogs_debug.attn_prefill_deep && p.prio_mode == 2) hipLaunchKernelGGL((attn_prefill_dma_kernel<1, 2>), grid, dim3(256), 0, st, p);
            else if (g_cogs_debug.attn_prefill_deep) hipLaunchKernelGGL((attn_prefill_dma_kernel<1, -1>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((attn_prefill_dma_kernel<0, -1>), grid, dim3(256), 0, st, p);
        } else if (a.head_dim == 128 && a.q_len == 1 && p.nsplit > 1 && p.gqa_pack && pre && !env_old_dec) {
            const int rc = cogs_k_attention_decode(st, a, p.part_o, p.part_ml);
            if (rc != COGS_OK) return rc;
            g_cogs_debug.attn_last_kernel = 4;
        } else if (a.head_dim == 72) {
            g_cogs_debug.attn_last_kernel = 1;
            if (light && !p.gqa_pack) COGS_ATTN_LAUNCH(72, 1, 512);
            else COGS_ATTN_LAUNCH(72, 2, 256);
        } else {
            g_cogs_debug.attn_last_kernel = 1;
            if (light && !p.gqa_pack) COGS_ATTN_LAUNCH(128, 1, 512);
            else COGS_ATTN_LAUNCH(128, 2, 256);
        }
#undef COGS_ATTN_LAUNCH
        if (p.nsplit > 1) {
            if (a.head_dim % 32 == 0 && g_cogs_debug.attn_combine32)
                hipLaunchKernelGGL(attn_combine32_kernel, dim3(a.q_len, a.hq, a.head_dim / 32), dim3(256), 0, st, p.part_o, p.part_ml,
                                   p.nsplit, a.q_len, a.hq, a.head_dim, (bf16_t*)a.O, a.ldo);
            else
                hipLaunchKernelGGL(attn_combine_kernel, dim3(a.q_len, a.hq), dim3(256), 0, st, p.part_o, p.part_ml, p.nsplit,
                                   a.q_len, a.hq, a.head_dim, (bf16_t*)a.O, a.ldo);
        }
        return COGS_LAUNCH_CHECK();
    }
    dim3 grid(a.q_len, a.hq);
    g_cogs_debug.attn_last_kernel = 7;
    if (a.dtype == COGS_DT_BF16)
        hipLaunchKernelGGL(attn_rowwise_kernel<bf16_t>, grid, dim3(64), 0, st, p, a.head_dim, nseg);
    else
        hipLaunchKernelGGL(attn_rowwise_kernel<float>, grid, dim3(64), 0, st, p, a.head_dim, nseg);
    return COGS_LAUNCH_CHECK();
}
