// Single-token (decode) attention, key-split: bf16, head dim 128, pre-scaled Q, GQA with up to 16 query heads per
// key/value head packed into the 16 query columns of one MFMA tile.
//
// Replaces, for the generated tokens, the attention inside Qwen2Model.forward (transformers 4.46.3, reached from
// model/cogreasoner_chat.py:802 through GenerationMixin): one new query row against the whole KV cache. HBM-bound:
// the step streams 57 344 B x context per layer once (31.5 MB at 15.4k tokens).
//
// The general kernel (attn.hip, gqa_pack mode) walks the 64-key tiles of a split one after the other with ONE wave
// computing and all four staging: per tile load -> barrier -> LDS write -> barrier -> compute, four dependent rounds of
// ~1.5 us per 256-key split and 32 KiB in flight per CU (13.5 us per layer, 2.3 TB/s). Here every WAVE owns whole tiles
// (tile t of a split goes to wave t & 3): it requests its K and V tile at once (32 non-temporal 16-byte loads per
// lane: 128 KiB in flight per CU), takes K straight into MFMA A fragments (no LDS), passes V through a wave-private
// LDS region for the transposing fragment reads (no workgroup barrier), runs S^T = K.Q^T -> softmax -> O^T = V^T.P^T on
// its own, and the four waves' (m, l, O) are merged through LDS into the split's partial -- the same partial format as
// before, so attn.hip's combine kernel is reused unchanged.
//
// v_mfma_f32_16x16x32_bf16, S^T[key][q]: lane (li = lane & 15, g = lane >> 4) of the accumulator of key block u holds
// keys 16u + 4g + r (r = register) of query column li; those registers of blocks 2c, 2c+1 ARE the B operand of k-step c
// of O^T (k-slot 8g + e <-> key 32c + 16(e >> 2) + 4g + (e & 3)), and the matching V^T A fragment is two
// ds_read_b64_tr_b16 (keys 32c + 4g + 0..3 and + 16) of the row-major V tile. V rows are 288 B apart in LDS: the 8 rows a
// transposing read touches per cycle start in 8 different 32-byte bank groups.
#include "common.h"
#include "kernels.h"

namespace {

struct DecodeAttnArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V;
    long ldk, ldv;                 // elements
    int kend;                      // keys 0 .. kend-1 are visible
    int hq, hkv, nsplit, gsz;
    float* part_o;                 // [nsplit][hq][128] unnormalised, relative to part_ml[..][0]
    float* part_ml;                // [nsplit][hq][2] (running max in log2 units, sum)
};

// max / sum over the four lanes {li, li+16, li+32, li+48} (one query column), result in all four (see attn.hip)
__device__ __forceinline__ float col_max4(float x) {
    const unsigned xb = __builtin_bit_cast(unsigned, x);
    const auto a = __builtin_amdgcn_permlane16_swap(xb, xb, false, false);
    const float y = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
    const unsigned yb = __builtin_bit_cast(unsigned, y);
    const auto b = __builtin_amdgcn_permlane32_swap(yb, yb, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}
__device__ __forceinline__ float col_sum4(float x) {
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}

__global__ __launch_bounds__(256, 1) void attn_decode_kernel(DecodeAttnArgs p) {
    constexpr int HD = 128, VRS = 288, VT = 64 * VRS;
    __shared__ __attribute__((aligned(16))) char vs[4 * VT];          // one V tile per wave
    __shared__ float cm[4][16], cl[4][16];
    __shared__ float co[4][16][HD + 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int split = blockIdx.x, kvh = blockIdx.y;
    const int gsz = p.gsz;                                       // hq / hkv, divided on the host

    // (unsigned 32-bit: tiles x splits is far below 2^32, checked by the launcher; the signed 64-bit form was two ~100-instruction
    // divisions at the head of a 10 us kernel)
    const unsigned nt_all = (unsigned)(p.kend + 63) >> 6;
    const int t_begin = (int)(nt_all * (unsigned)split / (unsigned)p.nsplit);
    const int t_end = (int)(nt_all * (unsigned)(split + 1) / (unsigned)p.nsplit);

    // Q fragments (B operand of S^T = K.Q^T): lane (q = li, g) holds Q[head kvh*gsz + li][32s + 8g .. +7]
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        u32x4 v = {0, 0, 0, 0};
        if (li < gsz) v = *reinterpret_cast<const u32x4*>(p.Q + (long)(kvh * gsz + li) * HD + 32 * s + 8 * g);
        qf[s] = __builtin_bit_cast(bf16x8, v);
    }

    f32x4 oacc[8];
#pragma unroll
    for (int db = 0; db < 8; ++db) oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;      // running maximum of the column (same in its four lanes), this lane's share of the sum

    char* const vw = vs + wid * VT;
    const int v_wr = (lane >> 4) * VRS + (lane & 15) * 16;                         // + 4i rows: row (lane>>4) + 4i, chunk lane&15
    const int v_rd = (4 * g + (li >> 2)) * VRS + (li & 3) * 8;                     // + (32c [+16]) rows + 32 db bytes

    for (int kt = t_begin + wid; kt < t_end; kt += 4) {
        const int kbase = kt * 64;
        const int nvalid = min(64, p.kend - kbase);                                // >= 1
        const bf16_t* kb = p.K + (long)kbase * p.ldk + kvh * HD;
        const bf16_t* vb = p.V + (long)kbase * p.ldv + kvh * HD;
        // every byte is read once per token by this wave: non-temporal. Rows past the end repeat the last row
        // (finite data; their scores are masked)
        u32x4 kf[4][4], vreg[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long ro = (long)min(16 * u + li, nvalid - 1) * p.ldk;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                kf[u][s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + ro + 32 * s + 8 * g));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long ro = (long)min((lane >> 4) + 4 * i, nvalid - 1) * p.ldv;
            vreg[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + ro + (lane & 15) * 8));
        }
        // S^T, four key blocks of 16
        f32x4 sacc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sacc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s)
                sacc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[u][s]), qf[s], sacc[u], 0, 0, 0);
        }
        // V tile -> this wave's LDS region (the previous tile's fragment reads of this wave are complete: in-order LDS)
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<u32x4*>(vw + v_wr + 4 * i * VRS) = vreg[i];
        // softmax of the tile (scores are in log2 units: Q is pre-scaled)
        float tmax = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * u + 4 * g + r >= nvalid) sacc[u][r] = -INFINITY;
                tmax = fmaxf(tmax, sacc[u][r]);
            }
        tmax = col_max4(tmax);                                   // finite: key kbase exists
        const float m_new = fmaxf(m, tmax);
        const float alpha = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - m_new);
        float psum = 0.f;
        bf16x8 pf[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            u32x4 w;
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const int e = 2 * e2;
                const float p0 = __builtin_amdgcn_exp2f(sacc[2 * c + (e >> 2)][e & 3] - m_new);
                const float p1 = __builtin_amdgcn_exp2f(sacc[2 * c + ((e + 1) >> 2)][(e + 1) & 3] - m_new);
                psum += p0 + p1;
                w[e2] = pack_bf2(p0, p1);
            }
            pf[c] = __builtin_bit_cast(bf16x8, w);
        }
        l = l * alpha + psum;
        m = m_new;
#pragma unroll
        for (int db = 0; db < 8; ++db) oacc[db] *= alpha;
        // O^T += V^T . P^T
#pragma unroll
        for (int db = 0; db < 8; ++db)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const char* va = vw + v_rd + 32 * c * VRS + 32 * db;
                const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(va));
                const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(va + 16 * VRS));
                const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                const u32x4 a = u32x4{l2[0], l2[1], h2[0], h2[1]};
                oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), pf[c], oacc[db], 0, 0, 0);
            }
    }

    // merge the four waves: (m, l, O) per query column through LDS
    l = col_sum4(l);
    if (g == 0) { cm[wid][li] = m; cl[wid][li] = l; }
    if (li < gsz) {
#pragma unroll
        for (int db = 0; db < 8; ++db) *reinterpret_cast<f32x4*>(&co[wid][li][16 * db + 4 * g]) = oacc[db];
    }
    __syncthreads();
    for (int idx = tid; idx < gsz * HD; idx += 256) {
        const int q = idx / HD, d = idx % HD;
        const float M = fmaxf(fmaxf(cm[0][q], cm[1][q]), fmaxf(cm[2][q], cm[3][q]));
        float o = 0.f, L = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float wt = (cm[w][q] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(cm[w][q] - M);
            o += co[w][q][d] * wt;
            L += cl[w][q] * wt;
        }
        const long slot = (long)split * p.hq + kvh * gsz + q;
        p.part_o[slot * HD + d] = (M == -INFINITY) ? 0.f : o;
        if (d == 0) { p.part_ml[slot * 2] = M; p.part_ml[slot * 2 + 1] = L; }
    }
}

}  // namespace

// single query row, bf16, hd 128, pre-scaled Q, keys split over nsplit workgroups per kv head; partials in the
// layout attn.hip's attn_combine_kernel reads (q_len = 1)
int cogs_k_attention_decode(hipStream_t st, const CogsAttn& a, float* part_o, float* part_ml) {
    if (a.head_dim != 128 || a.q_len != 1 || a.nsplit < 1 || a.hq % a.hkv || a.hq / a.hkv > 16) return COGS_E_UNSUPPORTED;
    if (a.ldk % 8 || a.ldv % 8) return COGS_E_INVALID;
    DecodeAttnArgs p;
    p.Q = (const bf16_t*)a.Q; p.K = (const bf16_t*)a.K; p.V = (const bf16_t*)a.V;
    p.ldk = a.ldk; p.ldv = a.ldv;
    p.kend = a.causal ? (a.q_pos0 + 1 < a.kv_len ? a.q_pos0 + 1 : a.kv_len) : a.kv_len;
    if (p.kend < 1) return COGS_E_INVALID;
    p.hq = a.hq; p.hkv = a.hkv; p.nsplit = a.nsplit; p.gsz = a.hq / a.hkv;
    if ((long)((p.kend + 63) >> 6) * (a.nsplit + 1) > 0x7fffffffL) return COGS_E_INVALID;
    p.part_o = part_o; p.part_ml = part_ml;
    hipLaunchKernelGGL(attn_decode_kernel, dim3(a.nsplit, a.hkv), dim3(256), 0, st, p);
    return COGS_LAUNCH_CHECK();
}
