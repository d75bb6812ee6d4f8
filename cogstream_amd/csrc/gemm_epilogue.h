// Fused epilogue shared by the MFMA GEMMs (gemm.hip) and the weight-streaming GEMV (gemv.hip).
// A caller hands over 4 CONSECUTIVE output columns n..n+3 of row m as fp32 accumulators.
//
//   bias      + b[n..n+3]
//   rope      columns < rope_cols: weight rows were packed so that (n, n+1) is the rotary pair
//             (d, d+hd/2) of rotate_half (model/modeling_videollama3_encoder.py:154-170; Qwen2
//             apply_rotary_pos_emb); cos/sin tables are [M, hd/2] fp32, or -- rope_sin == NULL -- ONE
//             interleaved table [M, hd/2, 2] = (cos, sin) in rope_cos (a lane's two pairs = one 16-byte load)
//   act       gelu_pytorch_tanh (ViT fc1), erf GELU (projector), SwiGLU on (gate_i, up_i)-interleaved
//             rows (Qwen2 MLP) -> two outputs at column n/2
//   residual  + R[m][n..n+3]
//   store     T or fp32 (lm_head logits)
//
// The feature set is a compile-time mask (EPI_*): a runtime-flag epilogue makes hipcc branch around every
// load and serialises 16 load->use round trips per tile (measured: +9 us per 256x128 tile for bias+residual).
// With the mask known, epilogue_tile() issues all bias / residual / rotary loads of a wave's 64x64 tile
// first and only then does the arithmetic and the stores.
#pragma once
#include "common.h"
#include "kernels.h"

#define EPI_BIAS 1
#define EPI_RES 2
#define EPI_ROPE 4
#define EPI_GELU_TANH 8
#define EPI_GELU_ERF 16
#define EPI_SWIGLU 32
#define EPI_F32OUT 64
#define EPI_GENERIC 128   // decide everything at run time (rare combinations)
#define EPI_ROPE_LUT 512  // with EPI_ROPE, ping-pong kernel only: rotary factors from the LDS-resident position LUT
#define EPI_NOSTORE 256   // diagnostics only (COGS_GEMM_NOSTORE): accumulators kept live, nothing written
// LayerNorm fused around the GEMMs (model/modeling_videollama3_encoder.py:382-391: x += attn(LN1(x)); x += mlp(LN2(x))):
//   EPI_ROWSTAT  the GEMM that PRODUCES the residual stream x (patch embed, out-proj, fc2; N = hidden) also writes, per
//                row and 64-column wave tile, the partial sums (sum x, sum x^2) of the values it STORES (the bf16-rounded
//                ones in bf16 mode: what the consumer will read): stat_part[M][N/64][2].
//                A tiny kernel turns them into (a, b) = (rstd, -rstd * mean) per row (cogs_k_ln_finalize).
//   EPI_LNFOLD   the GEMM that CONSUMES LN(x) reads x itself with W'' = rows of W * diag(gamma) CENTRED (their mean over k
//                subtracted, folded at load time: sum_k x_k W''[n][k] = sum_k (x_k - mean) W'[n][k]) and applies
//                y[r][n] = rstd_r * acc[r][n] + c_n, c_n = bias_n + sum_k beta_k W[n][k] -- algebraically LN(x) W^T + bias,
//                with no normalised copy of x ever written or read and no extra arithmetic in the epilogue (the bias add
//                becomes an fma). The term rstd * mean * (sum_k of the bf16-ROUNDED W'') is not evaluated: the packer
//                (weights.fold_layernorm) makes the rounded rows sum to zero as well, so there is nothing to neglect.
#define EPI_ROWSTAT 1024
#define EPI_LNFOLD 2048
// Head-major output (round 5; the ViT QKV GEMM): instead of C[m][n] the element of row m, column n goes to
//   C + ((n / hm_cols) * hm_rows * hm_cols  +  ((n % hm_cols) / hm_hd) * hm_rows * hm_hd  +  m * hm_hd  +  n % hm_hd)
// i.e. [q | k | v][head][row][head_dim]: every (head, frame) block of K and V is ONE contiguous run of rows, so the
// attention kernel's 1 KiB LDS-DMA pieces are whole 128-byte lines (54 B/clk per CU against 28 for the 144-byte runs at a
// 6 912-byte stride of the row-major layout: profiles/r4_ldsdma_rate.txt). 8 consecutive columns that start at a multiple
// of 8 never leave a head (hm_hd % 8 == 0), so the 16-byte stores of the wide paths stay whole.
#define EPI_HM 4096

struct EpiArgs {
    char* C; long ldc;          // elements per row
    const void* bias;
    const char* R; long ldr;    // elements per row
    int act;
    int out_f32;
    const float* rope_cos;
    const float* rope_sin;
    int rope_pairs;
    int rope_cols;
    int head_dim;
    const int* rope_rowpos;     // EPI_ROPE_LUT: [M] h | w << 16
    int rope_lut_lds;           //   byte offset of the LUT inside the kernel's dynamic LDS
    int rope_maxpos;            //   positions in the LUT (LUT row = rope_pairs/2 frequencies x (cos, sin))
    float q_scale;              // != 1: columns < q_cols are multiplied by it after bias/rope, before the (single) rounding
    int q_cols;                 //   (the attention kernels then take Q pre-scaled by softmax_scale*log2(e))
    float* stat_part;           // EPI_ROWSTAT: [M][stat_tiles][2] fp32
    int stat_tiles;             //   = N / 64
    const float* ln_ab;         // EPI_LNFOLD: [M][2] fp32 (rstd, -rstd * mean); the epilogues read rstd only (centred W)
    const float* col_c;         //   [N] fp32 (replaces bias)
    long hm_rows;               // EPI_HM: rows of the whole output (the head stride is hm_rows * hm_hd elements); 0 = row-major
    int hm_hd, hm_cols;         //   head_dim and columns per q / k / v block (= heads * head_dim)
};

// element offset of (row m, column n) in C: row-major, or head-major (see EPI_HM). HM is a compile-time choice in the
// specialised epilogues (the row-major kernels carry none of this) and a run-time one in the generic epilogue4.
template <bool HM>
__device__ __forceinline__ long c_elem(const EpiArgs& p, long m, int n) {
    if constexpr (!HM) return m * p.ldc + n;
    const int which = n / p.hm_cols, cw = n - which * p.hm_cols;
    const int head = cw / p.hm_hd, d = cw - head * p.hm_hd;
    return ((long)which * p.hm_cols + (long)head * p.hm_hd) * p.hm_rows + m * p.hm_hd + d;
}

// sum over the four lanes {r, r+16, r+32, r+48} that share a row of a 16x16 accumulator tile, result in all four
__device__ __forceinline__ float rowgroup_sum(float x) {
    const unsigned xb = __builtin_bit_cast(unsigned, x);
    const auto a = __builtin_amdgcn_permlane16_swap(xb, xb, false, false);
    const float y = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
    const unsigned yb = __builtin_bit_cast(unsigned, y);
    const auto b = __builtin_amdgcn_permlane32_swap(yb, yb, false, false);
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}

// (cos, sin) of the two rotary pairs pi, pi+1 of row m, from either table format
__device__ __forceinline__ void rope_load(const EpiArgs& p, long m, int pi, f32x2& c, f32x2& s) {
    if (p.rope_sin) {
        c = *reinterpret_cast<const f32x2*>(p.rope_cos + m * p.rope_pairs + pi);
        s = *reinterpret_cast<const f32x2*>(p.rope_sin + m * p.rope_pairs + pi);
    } else {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p.rope_cos + (m * p.rope_pairs + pi) * 2);
        c = f32x2{t[0], t[2]};
        s = f32x2{t[1], t[3]};
    }
}

// rotate the two pairs (v0, v1), (v2, v3) by (c0, s0), (c1, s1). ONE spelling with explicit fmaf for every epilogue: left
// to -ffp-contract the compiler fuses `a*c - b*s` one way in one epilogue and the other way in the next, and a row that
// takes the ragged-block epilogue in a frame-sharded encode then differs from the whole-clip encode in the last bit
__device__ __forceinline__ f32x4 rope_rot(f32x4 v, float c0, float s0, float c1, float s1) {
    f32x4 r;
    r[0] = fmaf(-v[1], s0, v[0] * c0);
    r[1] = fmaf(v[0], s0, v[1] * c0);
    r[2] = fmaf(-v[3], s1, v[2] * c1);
    r[3] = fmaf(v[2], s1, v[3] * c1);
    return r;
}

template <typename T>
__device__ __forceinline__ void epilogue4(const EpiArgs& p, int m, int n, f32x4 v) {
    if (p.bias) v += ld4_f<T>(reinterpret_cast<const T*>(p.bias) + n);
    if (p.rope_cos && n < p.rope_cols) {
        f32x2 c, s;
        rope_load(p, m, (n % p.head_dim) >> 1, c, s);
        v = rope_rot(v, c[0], s[0], c[1], s[1]);
    }
    if (n < p.q_cols) v *= p.q_scale;
    if (p.act == COGS_ACT_GELU_TANH) {
        v = gelu_tanh_4(v);
    } else if (p.act == COGS_ACT_GELU_ERF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
    } else if (p.act == COGS_ACT_SWIGLU) {
        const float o0 = silu_f(v[0]) * v[1];
        const float o1 = silu_f(v[2]) * v[3];
        T* cp = reinterpret_cast<T*>(p.C) + (long)m * p.ldc + (n >> 1);
        st_f<T>(cp, o0);
        st_f<T>(cp + 1, o1);
        return;
    }
    if (p.R) v += ld4_f<T>(reinterpret_cast<const T*>(p.R) + (long)m * p.ldr + n);
    if (p.out_f32)
        st4_f<float>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n, v);
    else
        st4_f<T>(reinterpret_cast<T*>(p.C) + (p.hm_rows ? c_elem<true>(p, m, n) : c_elem<false>(p, m, n)), v);
}

// raw storage of 4 consecutive elements (converted to fp32 only when used)
template <typename T> struct RawVec;
template <> struct RawVec<bf16_t> {
    typedef u32x2 type;
    static __device__ __forceinline__ f32x4 to_f32(u32x2 w) { return f32x4{bf_lo(w[0]), bf_hi(w[0]), bf_lo(w[1]), bf_hi(w[1])}; }
};
template <> struct RawVec<float> {
    typedef f32x4 type;
    static __device__ __forceinline__ f32x4 to_f32(f32x4 w) { return w; }
};

// Epilogue of one wave's 64x64 tile held as acc[mi][ni] (lane: row mb + 16 mi + (lane&15), columns
// nb + 16 ni + 4 (lane>>4) .. +3). Loads first, then math + stores.
template <typename T, int EPI>
__device__ __forceinline__ void epilogue_tile(const EpiArgs& p, int mb, int nb, int M, int N, int lane,
                                              f32x4 (&acc)[4][4]) {
    const int mrow = mb + (lane & 15);
    const int ncol = nb + ((lane >> 4) << 2);
    if constexpr ((EPI & EPI_NOSTORE) != 0) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) asm volatile("" ::"v"(acc[mi][ni]));
        return;
    } else if constexpr ((EPI & EPI_GENERIC) != 0) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = mrow + mi * 16;
            if (m >= M) continue;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = ncol + ni * 16;
                if (n < N) epilogue4<T>(p, m, n, acc[mi][ni]);
            }
        }
        return;
    } else {
        // Two half tiles (mi pairs): per half, phase 1 issues every load (clamped addresses keep the loads
        // unconditional), phase 2 does math + stores. Batching per half caps the live registers so hipcc does
        // not fall back to load->use round trips when the persistent loop keeps next-tile fragments live.
        //
        // WIDE path (bf16 output, N % 32 == 0): an MFMA lane owns 4 consecutive columns = 8 bytes, so a plain
        // store instruction writes 32-byte row segments -- measured at 2.6 TB/s, 27 % of an fc1 GEMM. Adjacent
        // n-tiles (ni, ni+1) are therefore paired with v_permlane16_swap (odd 16-lane rows of the first
        // register <-> even rows of the second): afterwards lane row g holds 8 consecutive columns
        // (16 bytes) of tile ni (g even) or ni+1 (g odd), one store covers 64 contiguous bytes per row, and the
        // residual is read through the same (self-inverse) exchange with 16-byte loads.
        using raw_t = typename RawVec<T>::type;
        constexpr bool CAN_WIDE = sizeof(T) == 2 && (EPI & (EPI_SWIGLU | EPI_F32OUT)) == 0;
        const bool wide = CAN_WIDE && (N & 31) == 0;
        const int g4 = lane >> 4;
        const int wcol = 16 * (g4 & 1) + 8 * (g4 >> 1);   // this lane's 8 columns inside a 32-column tile pair
        f32x4 bias_v[4];
        int nn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) nn[i] = min(ncol + i * 16, N - 4);
        if constexpr ((EPI & EPI_LNFOLD) != 0) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bias_v[ni] = *reinterpret_cast<const f32x4*>(p.col_c + nn[ni]);
        } else if constexpr ((EPI & EPI_BIAS) != 0) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bias_v[ni] = ld4_f<T>(reinterpret_cast<const T*>(p.bias) + nn[ni]);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            raw_t res_raw[2][4];
            u32x4 res_wide[2][2];
            f32x2 cs[2][4], sn[2][4];
            int mm[2];
            float rs2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) mm[i] = min(mrow + (2 * half + i) * 16, M - 1);
            if constexpr ((EPI & EPI_LNFOLD) != 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) rs2[i] = p.ln_ab[2 * (long)mm[i]];
            }
            if constexpr ((EPI & EPI_RES) != 0) {
                if (wide) {
                    if constexpr (CAN_WIDE) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int pr = 0; pr < 2; ++pr)
                                res_wide[i][pr] = *reinterpret_cast<const u32x4*>(
                                    reinterpret_cast<const T*>(p.R) + (long)mm[i] * p.ldr + min(nb + 32 * pr, N - 32) + wcol);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
                            res_raw[i][ni] = *reinterpret_cast<const raw_t*>(reinterpret_cast<const T*>(p.R) + (long)mm[i] * p.ldr + nn[ni]);
                }
            }
            if constexpr ((EPI & EPI_ROPE) != 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        rope_load(p, mm[i], (nn[ni] % p.head_dim) >> 1, cs[i][ni], sn[i][ni]);
                    }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int mi = 2 * half + i;
                const int m = mrow + mi * 16;
                f32x4 v[4];
                float st1 = 0.f, st2 = 0.f;   // EPI_ROWSTAT: this lane's share of (sum x, sum x^2) of row m in this tile
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int n = ncol + ni * 16;
                    v[ni] = acc[mi][ni];
                    if constexpr ((EPI & EPI_LNFOLD) != 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[ni][e] = fmaf(rs2[i], v[ni][e], bias_v[ni][e]);
                    } else if constexpr ((EPI & EPI_BIAS) != 0) v[ni] += bias_v[ni];
                    if constexpr ((EPI & EPI_ROPE) != 0) {
                        if (n < p.rope_cols) {
                            const f32x2 c = cs[i][ni], sx = sn[i][ni];
                            v[ni] = rope_rot(v[ni], c[0], sx[0], c[1], sx[1]);
                        }
                        if (n < p.q_cols) v[ni] *= p.q_scale;
                    }
                    if constexpr ((EPI & EPI_GELU_TANH) != 0) {
                        v[ni] = gelu_tanh_4(v[ni]);
                    }
                    if constexpr ((EPI & EPI_GELU_ERF) != 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[ni][e] = gelu_erf_f(v[ni][e]);
                    }
                }
                if (wide) {
                    if constexpr (CAN_WIDE) {
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            f32x4 va = v[2 * pr], vb = v[2 * pr + 1];
                            if constexpr ((EPI & EPI_RES) != 0) {
                                // residual arrives in the exchanged layout: exchange back, then add in fp32
                                const u32x4 rw = res_wide[i][pr];
                                const auto x0 = __builtin_amdgcn_permlane16_swap(rw[0], rw[2], false, false);
                                const auto x1 = __builtin_amdgcn_permlane16_swap(rw[1], rw[3], false, false);
                                va += f32x4{bf_lo(x0[0]), bf_hi(x0[0]), bf_lo(x1[0]), bf_hi(x1[0])};
                                vb += f32x4{bf_lo(x0[1]), bf_hi(x0[1]), bf_lo(x1[1]), bf_hi(x1[1])};
                            }
                            const unsigned a0 = pack_bf2(va[0], va[1]), a1 = pack_bf2(va[2], va[3]);
                            const unsigned b0 = pack_bf2(vb[0], vb[1]), b1 = pack_bf2(vb[2], vb[3]);
                            if constexpr ((EPI & EPI_ROWSTAT) != 0) {
                                // statistics of the STORED (bf16-rounded) values: they are what the consuming GEMM multiplies,
                                // and on rows dominated by their mean the rounding noise is a visible part of the variance.
                                // The SAME association as epilogue_pair_fast (a 32-column unit summed from zero, units added
                                // in order): a row's statistics must not depend on which epilogue its tile happened to take,
                                // or a frame-sharded encode would differ from the whole-clip encode in the last bit
                                if (nb + 32 * pr < N) {
                                    const f32x4 ra = {bf_lo(a0), bf_hi(a0), bf_lo(a1), bf_hi(a1)};
                                    const f32x4 rb = {bf_lo(b0), bf_hi(b0), bf_lo(b1), bf_hi(b1)};
                                    float u1 = 0.f, u2 = 0.f;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        u1 += ra[e] + rb[e];
                                        u2 = fmaf(ra[e], ra[e], fmaf(rb[e], rb[e], u2));
                                    }
                                    if (pr == 0) { st1 = u1; st2 = u2; } else { st1 += u1; st2 += u2; }
                                }
                            }
                            const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                            const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                            if (m < M && nb + 32 * pr < N)   // N % 32 == 0: a tile pair is entirely inside or outside
                                *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.C) + c_elem<(EPI & EPI_HM) != 0>(p, m, nb + 32 * pr + wcol)) =
                                    u32x4{s0[0], s1[0], s0[1], s1[1]};
                        }
                    }
                } else {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        const int n = ncol + ni * 16;
                        if constexpr ((EPI & EPI_RES) != 0) v[ni] += RawVec<T>::to_f32(res_raw[i][ni]);
                        if constexpr ((EPI & EPI_ROWSTAT) != 0) {
                            if (n < N) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) { st1 += v[ni][e]; st2 = fmaf(v[ni][e], v[ni][e], st2); }
                            }
                        }
                        if (m < M && n < N) {
                            if constexpr ((EPI & EPI_SWIGLU) != 0) {
                                T* cp = reinterpret_cast<T*>(p.C) + (long)m * p.ldc + (n >> 1);
                                st_f<T>(cp, silu_f(v[ni][0]) * v[ni][1]);
                                st_f<T>(cp + 1, silu_f(v[ni][2]) * v[ni][3]);
                            } else if constexpr ((EPI & EPI_F32OUT) != 0) {
                                st4_f<float>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n, v[ni]);
                            } else {
                                st4_f<T>(reinterpret_cast<T*>(p.C) + c_elem<(EPI & EPI_HM) != 0>(p, m, n), v[ni]);
                            }
                        }
                    }
                }
                if constexpr ((EPI & EPI_ROWSTAT) != 0) {
                    st1 = rowgroup_sum(st1);
                    st2 = rowgroup_sum(st2);
                    if (g4 == 0 && m < M && nb < N)
                        *reinterpret_cast<f32x2*>(p.stat_part + ((long)m * p.stat_tiles + (nb >> 6)) * 2) = f32x2{st1, st2};
                }
            }
        }
    }
}

// Interior wave tile on the wide path (bf16 output, whole 64x64 tile inside the matrix, 16-byte aligned rows,
// rotary either on or off for the whole tile): the same arithmetic as epilogue_tile() with lean addressing.
// The general path spends ~5000 VALU cycles per wave and tile on 64-bit `row * ld` products (quarter-rate
// v_mul_lo_u32 / v_mad_u64_u32) and per-store exec masking -- measured 8-10 % of a K = 1152 GEMM. Here the
// tile base is wave-uniform (SGPRs), a lane adds ONE 32-bit byte offset, rows advance by a uniform stride,
// nothing is predicated, and the rotary (cos, sin) pairs come from the interleaved table with one 16-byte load.
template <int EPI>
__device__ __forceinline__ void epilogue_tile_fast(const EpiArgs& p, int mb, int nb, int lane, f32x4 (&acc)[4][4]) {
    typedef bf16_t T;
    const int r = lane & 15, g4 = lane >> 4;
    const int wcol = 16 * (g4 & 1) + 8 * (g4 >> 1);
    char* cbase;
    unsigned c_lane, c_lane1;
    long c_row16;
    if constexpr ((EPI & EPI_HM) != 0) {     // see epilogue_pair_fast
        const int which = nb / p.hm_cols, cw0 = nb - which * p.hm_cols + wcol;
        cbase = p.C + ((long)which * p.hm_cols * p.hm_rows + (long)mb * p.hm_hd) * 2;
        const int h0 = cw0 / p.hm_hd, h1 = (cw0 + 32) / p.hm_hd;
        c_lane = (unsigned)(((long)h0 * p.hm_rows + r) * p.hm_hd + (cw0 - h0 * p.hm_hd)) * 2u;
        c_lane1 = (unsigned)(((long)h1 * p.hm_rows + r) * p.hm_hd + (cw0 + 32 - h1 * p.hm_hd)) * 2u;
        c_row16 = (long)p.hm_hd * 32;
    } else {
        cbase = p.C + ((long)mb * p.ldc + nb) * 2;
        c_lane = ((unsigned)r * (unsigned)p.ldc + (unsigned)wcol) * 2u;
        c_lane1 = 0;                   // row-major: tile pair 1 is 64 bytes further (an immediate in the store)
        c_row16 = p.ldc * 32;
    }
    const char* rbase = nullptr;
    unsigned r_lane = 0;
    long r_row16 = 0;
    if constexpr ((EPI & EPI_RES) != 0) {
        rbase = p.R + ((long)mb * p.ldr + nb) * 2;
        r_lane = ((unsigned)r * (unsigned)p.ldr + (unsigned)wcol) * 2u;
        r_row16 = p.ldr * 32;
    }
    f32x4 bias_v[4];
    if constexpr ((EPI & EPI_BIAS) != 0) {
        const char* bbase = reinterpret_cast<const char*>(p.bias) + (long)nb * 2;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bias_v[ni] = ld4_f<T>(reinterpret_cast<const T*>(bbase + (unsigned)(g4 * 8 + ni * 32)));
    }
    const bool q_tile = nb < p.q_cols;   // wave-uniform (epilogue_wave sends tiles straddling q_cols to the general path)
    const char* csbase = nullptr;
    unsigned cs_lane[4] = {0, 0, 0, 0};
    long cs_row16 = 0;
    if constexpr ((EPI & EPI_ROPE) != 0) {
        csbase = reinterpret_cast<const char*>(p.rope_cos) + (long)mb * p.rope_pairs * 8;
        cs_row16 = (long)p.rope_pairs * 128;
        const int nbmod = nb % p.head_dim;     // wave-uniform
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            int c = nbmod + 16 * ni + 4 * g4;  // < 2 * head_dim (head_dim >= 64)
            c -= c >= p.head_dim ? p.head_dim : 0;
            cs_lane[ni] = (unsigned)(r * p.rope_pairs * 8 + c * 4);
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        u32x4 res_wide[2][2];
        f32x4 cs4[2][4];
        if constexpr ((EPI & EPI_RES) != 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
                    res_wide[i][pr] = *reinterpret_cast<const u32x4*>(rbase + (2 * half + i) * r_row16 + r_lane + 64 * pr);
        }
        if constexpr ((EPI & EPI_ROPE) != 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    cs4[i][ni] = *reinterpret_cast<const f32x4*>(csbase + (2 * half + i) * cs_row16 + cs_lane[ni]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mi = 2 * half + i;
            f32x4 v[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                v[ni] = acc[mi][ni];
                if constexpr ((EPI & EPI_BIAS) != 0) v[ni] += bias_v[ni];
                if constexpr ((EPI & EPI_ROPE) != 0) {
                    const f32x4 t = cs4[i][ni];   // c0 s0 c1 s1
                    v[ni] = rope_rot(v[ni], t[0], t[1], t[2], t[3]);
                    if (q_tile) v[ni] *= p.q_scale;
                }
                if constexpr ((EPI & EPI_GELU_TANH) != 0) {
                    v[ni] = gelu_tanh_4(v[ni]);
                }
                if constexpr ((EPI & EPI_GELU_ERF) != 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[ni][e] = gelu_erf_f(v[ni][e]);
                }
            }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                f32x4 va = v[2 * pr], vb = v[2 * pr + 1];
                if constexpr ((EPI & EPI_RES) != 0) {
                    const u32x4 rw = res_wide[i][pr];
                    const auto x0 = __builtin_amdgcn_permlane16_swap(rw[0], rw[2], false, false);
                    const auto x1 = __builtin_amdgcn_permlane16_swap(rw[1], rw[3], false, false);
                    va += f32x4{bf_lo(x0[0]), bf_hi(x0[0]), bf_lo(x1[0]), bf_hi(x1[0])};
                    vb += f32x4{bf_lo(x0[1]), bf_hi(x0[1]), bf_lo(x1[1]), bf_hi(x1[1])};
                }
                const unsigned a0 = pack_bf2(va[0], va[1]), a1 = pack_bf2(va[2], va[3]);
                const unsigned b0 = pack_bf2(vb[0], vb[1]), b1 = pack_bf2(vb[2], vb[3]);
                const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                if constexpr ((EPI & EPI_HM) != 0) *reinterpret_cast<u32x4*>(cbase + mi * c_row16 + (pr ? c_lane1 : c_lane)) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                else *reinterpret_cast<u32x4*>(cbase + mi * c_row16 + c_lane + 64 * pr) = u32x4{s0[0], s1[0], s0[1], s1[1]};
            }
        }
    }
}

// Epilogue of one wave's 64x64 tile: the lean path when the (wave-uniform) conditions hold, else the general one.
template <typename T, int EPI>
__device__ __forceinline__ void epilogue_wave(const EpiArgs& p, int mb, int nb, int M, int N, int lane,
                                              f32x4 (&acc)[4][4]) {
    if constexpr (sizeof(T) == 2 && (EPI & (EPI_SWIGLU | EPI_F32OUT | EPI_GENERIC | EPI_NOSTORE | EPI_ROWSTAT | EPI_LNFOLD)) == 0) {
        bool fast = mb + 64 <= M && nb + 64 <= N && (N & 31) == 0 && (reinterpret_cast<unsigned long>(p.C) & 15) == 0;
        if constexpr ((EPI & EPI_HM) != 0) fast = fast && (p.hm_cols & 63) == 0 && (p.hm_hd & 7) == 0;
        else fast = fast && (p.ldc & 7) == 0;
        if constexpr ((EPI & EPI_RES) != 0) fast = fast && (p.ldr & 7) == 0 && (reinterpret_cast<unsigned long>(p.R) & 15) == 0;
        if constexpr ((EPI & EPI_ROPE) != 0) {
            fast = fast && p.rope_sin == nullptr && p.head_dim >= 64 && (nb + 64 <= p.rope_cols || nb >= p.rope_cols) &&
                   (nb + 64 <= p.q_cols || nb >= p.q_cols);
            if (fast) {
                if (nb < p.rope_cols) epilogue_tile_fast<EPI>(p, mb, nb, lane, acc);
                else epilogue_tile_fast<(EPI & ~EPI_ROPE)>(p, mb, nb, lane, acc);
                return;
            }
        } else {
            if (fast) { epilogue_tile_fast<EPI>(p, mb, nb, lane, acc); return; }
        }
    }
    epilogue_tile<T, EPI>(p, mb, nb, M, N, lane, acc);
}

// The ping-pong kernel's wave region (128 rows x 64 columns = acc0 | acc1) as FOUR 32-row batches, software-
// pipelined: the loads of batch b+2 (residual rows / rotary pairs) are issued BEFORE the stores of batch b. vmcnt
// counts loads and stores in one in-order queue, so in the plain per-half sequence (load, wait, math, store, load,
// ...) every wait for a load also waits for the stores issued before it -- a store round trip plus a load round trip
// per batch, which is what made the rope / residual epilogues cost 20-25 % of a K = 1152 tile.
template <int EPI>
__device__ __forceinline__ void epilogue_pair_fast(const EpiArgs& p, int mb, int nb, int lane, f32x4 (&acc0)[4][4],
                                                   f32x4 (&acc1)[4][4]) {
    typedef bf16_t T;
    // Region markers for the build-time check (cogstream_amd/build.py::check_epilogue_vmem_counts): the relaxed
    // vmcnt of the ping-pong kernel's next-tile waits is only right if hipcc emits exactly epi_pair_vmem_ops<EPI>()
    // vector-memory instructions between these two markers; the build disassembles gemm.o and counts them.
    // s_nop 8 / 9 = begin (without / with rotary loads), s_nop 10 = end. ("memory": nothing moves across.)
    if constexpr ((EPI & EPI_ROPE) != 0) asm volatile("s_nop 9" ::: "memory");
    else asm volatile("s_nop 8" ::: "memory");
    const int r = lane & 15, g4 = lane >> 4;
    const int wcol = 16 * (g4 & 1) + 8 * (g4 >> 1);
    char* cbase;
    unsigned c_lane, c_lane1;          // byte offsets of this lane's 16 bytes in tile pair 0 / 1 of a row block
    long c_row16;
    if constexpr ((EPI & EPI_HM) != 0) {
        // head-major: the wave tile's 64 columns lie inside ONE of q / k / v (hm_cols % 64 == 0, checked by the caller) and
        // touch at most two heads; a lane's 8 columns of a tile pair never leave a head. Row blocks advance by 16 rows of hd.
        const int which = nb / p.hm_cols, cw0 = nb - which * p.hm_cols + wcol;
        cbase = p.C + ((long)which * p.hm_cols * p.hm_rows + (long)mb * p.hm_hd) * 2;
        const int h0 = cw0 / p.hm_hd, h1 = (cw0 + 32) / p.hm_hd;
        c_lane = (unsigned)(((long)h0 * p.hm_rows + r) * p.hm_hd + (cw0 - h0 * p.hm_hd)) * 2u;     // < 2^32: checked on the host
        c_lane1 = (unsigned)(((long)h1 * p.hm_rows + r) * p.hm_hd + (cw0 + 32 - h1 * p.hm_hd)) * 2u;
        c_row16 = (long)p.hm_hd * 32;
    } else {
        cbase = p.C + ((long)mb * p.ldc + nb) * 2;
        c_lane = ((unsigned)r * (unsigned)p.ldc + (unsigned)wcol) * 2u;
        c_lane1 = 0;                   // row-major: tile pair 1 is 64 bytes further (an immediate in the store)
        c_row16 = p.ldc * 32;
    }
    const char* rbase = nullptr;
    unsigned r_lane = 0;
    long r_row16 = 0;
    if constexpr ((EPI & EPI_RES) != 0) {
        rbase = p.R + ((long)mb * p.ldr + nb) * 2;
        r_lane = ((unsigned)r * (unsigned)p.ldr + (unsigned)wcol) * 2u;
        r_row16 = p.ldr * 32;
    }
    f32x4 bias_v[4];
    if constexpr ((EPI & EPI_LNFOLD) != 0) {
        // y = rstd_r * acc + c_n (W rows are centred, so the mean term vanishes): c takes the place of the bias (fp32)
        const char* cbase2 = reinterpret_cast<const char*>(p.col_c) + (long)nb * 4;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bias_v[ni] = *reinterpret_cast<const f32x4*>(cbase2 + (unsigned)(g4 * 16 + ni * 64));
    } else if constexpr ((EPI & EPI_BIAS) != 0) {
        const char* bbase = reinterpret_cast<const char*>(p.bias) + (long)nb * 2;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bias_v[ni] = ld4_f<T>(reinterpret_cast<const T*>(bbase + (unsigned)(g4 * 8 + ni * 32)));
    }
    // EPI_LNFOLD: rstd of row mb + 16 blk + r. Loaded with the batch that first touches the row block (one 4-byte load
    // per row block, 8 per epilogue) into a ring indexed by blk & 7 for 4-unit batches (6 row blocks in flight) or
    // blk & 1 for the rotary kernels' 1-unit batches (2 in flight): the rotary epilogue has no registers to spare
    const char* rsbase = nullptr;
    if constexpr ((EPI & EPI_LNFOLD) != 0) rsbase = reinterpret_cast<const char*>(p.ln_ab) + ((long)mb + r) * 8;
    float rs_ring[8];
    char* stbase = nullptr;            // EPI_ROWSTAT: partials of row mb + 16 blk + r, this wave tile
    long st_row16 = 0;
    if constexpr ((EPI & EPI_ROWSTAT) != 0) {
        stbase = reinterpret_cast<char*>(p.stat_part) + (((long)mb + r) * p.stat_tiles + (nb >> 6)) * 8;
        st_row16 = (long)p.stat_tiles * 128;
    }
    const bool q_tile = nb < p.q_cols;
    const char* csbase = nullptr;
    unsigned cs_lane[4] = {0, 0, 0, 0};
    long cs_row16 = 0;
    if constexpr ((EPI & EPI_ROPE) != 0) {
        csbase = reinterpret_cast<const char*>(p.rope_cos) + (long)mb * p.rope_pairs * 8;
        cs_row16 = (long)p.rope_pairs * 128;
        const int nbmod = nb % p.head_dim;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            int c = nbmod + 16 * ni + 4 * g4;
            c -= c >= p.head_dim ? p.head_dim : 0;
            cs_lane[ni] = (unsigned)(r * p.rope_pairs * 8 + c * 4);
        }
    }
    // EPI_ROPE_LUT: per-lane LUT addressing (fixed for the tile) and the row positions of the lane's 8 row blocks
    extern __shared__ __attribute__((aligned(16))) char epi_smem[];
    const char* lut_base = nullptr;
    int lut_off[4] = {0, 0, 0, 0}, lut_row = 0, rowpos_v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool lut_w[4] = {false, false, false, false};
    if constexpr ((EPI & EPI_ROPE_LUT) != 0) {
        const int nf = p.rope_pairs >> 1;               // frequencies per axis
        lut_base = epi_smem + p.rope_lut_lds;
        lut_row = nf * 8;
        const int nbmod = nb % p.head_dim;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            int c = nbmod + 16 * ni + 4 * g4;
            c -= c >= p.head_dim ? p.head_dim : 0;
            const int pi = c >> 1;                       // even pair index inside the head
            lut_w[ni] = pi >= nf;
            lut_off[ni] = (pi - (lut_w[ni] ? nf : 0)) * 8;
        }
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) rowpos_v[blk] = p.rope_rowpos[mb + 16 * blk + r];
    }
    // The region is 16 units of (16-row block blk, 32-column tile pair pr); a unit ends in ONE 16-byte store per lane.
    // A batch is UPB consecutive units: 4 (two row blocks) normally, 1 with rotary loads -- two batches are in flight
    // and a rotary unit already holds 8 registers of (cos, sin), so finer batches keep the kernel out of spills.
    constexpr int UPB = (EPI & EPI_ROPE) != 0 ? 1 : 4;
    constexpr int NB = 16 / UPB;
    u32x4 res_wide[NB][UPB];
    f32x4 cs4[NB][UPB][2];
    float st_sum[2] = {0.f, 0.f}, st_sq[2] = {0.f, 0.f};   // EPI_ROWSTAT: running sums of the batch's (<= 2) row blocks
    auto load_batch = [&](const int b) {
#pragma unroll
        for (int j = 0; j < UPB; ++j) {
            const int u = UPB * b + j, blk = u >> 1, pr = u & 1;
            if constexpr ((EPI & EPI_LNFOLD) != 0) {
                if (pr == 0) rs_ring[UPB == 1 ? (blk & 7) : blk] = *reinterpret_cast<const float*>(rsbase + blk * 128);
            }
            if constexpr ((EPI & EPI_RES) != 0)
                res_wide[b][j] = *reinterpret_cast<const u32x4*>(rbase + blk * r_row16 + r_lane + 64 * pr);
            if constexpr ((EPI & EPI_ROPE) != 0) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int ni = 2 * pr + h2;
                    if constexpr ((EPI & EPI_ROPE_LUT) != 0) {
                        // LDS LUT: row = position (h for the first half of the head's pairs, w for the second)
                        const int rp = rowpos_v[blk];
                        const int pos = lut_w[ni] ? (rp >> 16) : (rp & 0xffff);
                        cs4[b][j][h2] = *reinterpret_cast<const f32x4*>(lut_base + lut_off[ni] + pos * lut_row);
                    } else {
                        cs4[b][j][h2] = *reinterpret_cast<const f32x4*>(csbase + blk * cs_row16 + cs_lane[ni]);
                    }
                }
            }
        }
    };
    u32x4 outv[UPB];
    auto math_batch = [&](const int b) {
#pragma unroll
        for (int j = 0; j < UPB; ++j) {
            const int u = UPB * b + j, blk = u >> 1, pr = u & 1, mi = blk & 3;
            f32x4 v[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int ni = 2 * pr + h2;
                v[h2] = (blk < 4) ? acc0[mi][ni] : acc1[mi][ni];
                if constexpr ((EPI & EPI_LNFOLD) != 0) {
                    const float rs = rs_ring[UPB == 1 ? (blk & 7) : blk];
#ifdef COGS_EPI_SCALAR_MATH
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[h2][e] = fmaf(rs, v[h2][e], bias_v[ni][e]);
#else
                    v[h2] = __builtin_elementwise_fma(f32x4{rs, rs, rs, rs}, v[h2], bias_v[ni]);
#endif
                } else if constexpr ((EPI & EPI_BIAS) != 0) v[h2] += bias_v[ni];
                if constexpr ((EPI & EPI_ROPE) != 0) {
                    const f32x4 t = cs4[b][j][h2];   // c0 s0 c1 s1
                    v[h2] = rope_rot(v[h2], t[0], t[1], t[2], t[3]);
                    if (q_tile) v[h2] *= p.q_scale;
                }
                if constexpr ((EPI & EPI_GELU_TANH) != 0) {
                    v[h2] = gelu_tanh_4(v[h2]);
                }
                if constexpr ((EPI & EPI_GELU_ERF) != 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[h2][e] = gelu_erf_f(v[h2][e]);
                }
            }
            f32x4 va = v[0], vb = v[1];
            if constexpr ((EPI & EPI_RES) != 0) {
                const u32x4 rw = res_wide[b][j];
                const auto x0 = __builtin_amdgcn_permlane16_swap(rw[0], rw[2], false, false);
                const auto x1 = __builtin_amdgcn_permlane16_swap(rw[1], rw[3], false, false);
                va += f32x4{bf_lo(x0[0]), bf_hi(x0[0]), bf_lo(x1[0]), bf_hi(x1[0])};
                vb += f32x4{bf_lo(x0[1]), bf_hi(x0[1]), bf_lo(x1[1]), bf_hi(x1[1])};
            }
            const unsigned a0 = pack_bf2(va[0], va[1]), a1 = pack_bf2(va[2], va[3]);
            const unsigned b0 = pack_bf2(vb[0], vb[1]), b1 = pack_bf2(vb[2], vb[3]);
            if constexpr ((EPI & EPI_ROWSTAT) != 0) {
                // of the STORED values (see epilogue_tile). Row block blk of the batch = (j >> 1) (UPB == 4: units
                // 4b..4b+3 = row blocks 2b, 2b+1, two units each)
                const f32x4 ra = {bf_lo(a0), bf_hi(a0), bf_lo(a1), bf_hi(a1)};
                const f32x4 rb = {bf_lo(b0), bf_hi(b0), bf_lo(b1), bf_hi(b1)};
                // (one association everywhere: a packed form of these sums was tried in round 5 -- no faster, and the generic
                // epilogues of ragged tiles must produce the same bits for the frame-shard identity)
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1 += ra[e] + rb[e];
                    s2 = fmaf(ra[e], ra[e], fmaf(rb[e], rb[e], s2));
                }
                if ((j & 1) == 0) { st_sum[j >> 1] = s1; st_sq[j >> 1] = s2; }
                else { st_sum[j >> 1] += s1; st_sq[j >> 1] += s2; }
            }
            const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
            outv[j] = u32x4{s0[0], s1[0], s0[1], s1[1]};
        }
    };
    // WHOLE-LINE stores (round 6; row-major output, 4-unit batches). A unit's store is 16 rows x 64 bytes: sixteen HALF lines per
    // instruction, which the CU's vector-memory path takes at ~30 B/clk against ~54 for whole lines -- and a plain tile's 128 store
    // instructions are its whole 5 k-cycle epilogue (profiles/r6_gemm_tile_stamps.txt). The two units of a 16-row block hold, per
    // lane (row r, column group), the first and the second 64 bytes of row r; lanes r and r ^ 8 swap one of them (8 v_mov_dpp
    // row_ror:8 with a bank mask per row block), after which lane r < 8 holds the first halves of rows r and r + 8 and lane r >= 8
    // the second halves of rows r - 8 and r: store A writes rows 0..7 of the block as whole 128-byte lines, store B rows 8..15.
    // Same bytes, same number of store instructions (the relaxed vmcnt counts stay).
    constexpr bool WHOLE_LINES = (EPI & EPI_HM) == 0 && UPB == 4;
    unsigned c_laneA = 0;
    if constexpr (WHOLE_LINES) c_laneA = ((unsigned)(r & 7) * (unsigned)p.ldc + (unsigned)wcol) * 2u + (unsigned)(r >> 3) * 64u;
    auto ror8_hi = [](unsigned keep, unsigned give) -> unsigned {      // lanes 8..15 of every row of 16: `give` of lane - 8; lanes 0..7: keep
        return (unsigned)__builtin_amdgcn_update_dpp((int)keep, (int)give, 0x128, 0xf, 0xc, false);
    };
    auto ror8_lo = [](unsigned keep, unsigned give) -> unsigned {      // lanes 0..7: `give` of lane + 8; lanes 8..15: keep
        return (unsigned)__builtin_amdgcn_update_dpp((int)keep, (int)give, 0x128, 0xf, 0x3, false);
    };
    auto store_batch = [&](const int b) {
        if constexpr (WHOLE_LINES) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {                    // the batch's two row blocks
                const int blk = 2 * b + k;
                const u32x4 o0 = outv[2 * k], o1 = outv[2 * k + 1];
                u32x4 da, db;
#pragma unroll
                for (int e = 0; e < 4; ++e) { da[e] = ror8_hi(o0[e], o1[e]); db[e] = ror8_lo(o1[e], o0[e]); }
                *reinterpret_cast<u32x4*>(cbase + blk * c_row16 + c_laneA) = da;
                *reinterpret_cast<u32x4*>(cbase + blk * c_row16 + (c_row16 >> 1) + c_laneA) = db;
            }
        } else {
#pragma unroll
        for (int j = 0; j < UPB; ++j) {
            const int u = UPB * b + j, blk = u >> 1, pr = u & 1;
            if constexpr ((EPI & EPI_HM) != 0) *reinterpret_cast<u32x4*>(cbase + blk * c_row16 + (pr ? c_lane1 : c_lane)) = outv[j];
            else *reinterpret_cast<u32x4*>(cbase + blk * c_row16 + c_lane + 64 * pr) = outv[j];
        }
        }
        if constexpr ((EPI & EPI_ROWSTAT) != 0) {
            static_assert((EPI & EPI_ROWSTAT) == 0 || UPB == 4, "row statistics are laid out for 4-unit batches");
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float t1 = rowgroup_sum(st_sum[k]), t2 = rowgroup_sum(st_sq[k]);
                // all four lanes of a row hold the totals; ONE of them writes (8 bytes per row and wave tile). The store
                // is issued by every lane with the other three pointed at the same address and value: no exec masking,
                // so the instruction count the relaxed vmcnt relies on stays fixed (2 per batch, 8 per epilogue)
                *reinterpret_cast<f32x2*>(stbase + (2 * b + k) * st_row16) = f32x2{t1, t2};
            }
        }
    };
    constexpr bool HAS_LOADS = (EPI & (EPI_RES | EPI_ROPE | EPI_LNFOLD)) != 0;
    // batches in flight: 2, or -- rotary (1-unit batches of two 16-byte table loads: round 6's tile stamps put a rotary tile's
    // epilogue at 15.5 k cycles against 5 k for a plain one, sixteen load round trips two at a time) -- COGS_ROPE_DEPTH
#ifndef COGS_ROPE_DEPTH
#define COGS_ROPE_DEPTH 4
#endif
    constexpr int DEPTH = (EPI & EPI_ROPE) != 0 ? COGS_ROPE_DEPTH : 2;
    if constexpr (HAS_LOADS) {
#pragma unroll
        for (int b = 0; b < DEPTH; ++b) load_batch(b);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        math_batch(b);
        if constexpr (HAS_LOADS) { if (b + DEPTH < NB) load_batch(b + DEPTH); }   // before this batch's stores (see above)
        store_batch(b);
    }
    asm volatile("s_nop 10" ::: "memory");
}

// SwiGLU (Qwen2 gate / up, weight rows interleaved: a lane's 4 columns are (gate_i, up_i, gate_i+1, up_i+1)) on the ping-pong
// kernel's wave region, round 6. The generic path stored its two outputs per lane and n-tile as two 2-byte stores (64
// global_store_short per wave and tile, each with its own 64-bit address and exec mask). Here a lane packs its two outputs of an
// n-tile into one dword, the four lanes that share a row TRANSPOSE their four dwords (n-tile x lane: two v_permlane16_swap, two
// v_permlane32_swap), and lane group t ends up with the 8 consecutive output columns of n-tile t: ONE 16-byte store per lane and
// 16-row block, 64 contiguous bytes per row, 8 store instructions per wave and tile -- and an exact count for the relaxed vmcnt
// of the next tile's first slab. Same arithmetic per element (silu_f(gate) * up, one rounding), so the bits do not change.
__device__ __forceinline__ void epilogue_pair_swiglu(const EpiArgs& p, int mb, int nb, int lane, f32x4 (&acc0)[4][4],
                                                     f32x4 (&acc1)[4][4]) {
    asm volatile("s_nop 8" ::: "memory");      // region markers of the build-time vmem-count check (see epilogue_pair_fast)
    const int r = lane & 15, g4 = lane >> 4;
    char* cbase = p.C + ((long)mb * p.ldc + (nb >> 1)) * 2;
    const unsigned c_lane = ((unsigned)r * (unsigned)p.ldc + (unsigned)(8 * g4)) * 2u;
    const long c_row16 = p.ldc * 32;
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
        unsigned d[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const f32x4 v = blk < 4 ? acc0[blk][ni] : acc1[blk - 4][ni];
            d[ni] = pack_bf2(silu_f(v[0]) * v[1], silu_f(v[2]) * v[3]);
        }
        // 4 x 4 transpose over the lane rows {r, r+16, r+32, r+48}: row t collects n-tile t's dwords of rows 0..3 in order
        const auto e = __builtin_amdgcn_permlane16_swap(d[0], d[1], false, false);     // e[0] = [d0@0 d1@0 d0@2 d1@2], e[1] = [d0@1 d1@1 d0@3 d1@3]
        const auto f = __builtin_amdgcn_permlane16_swap(d[2], d[3], false, false);
        const auto x = __builtin_amdgcn_permlane32_swap((unsigned)e[0], (unsigned)f[0], false, false);   // x[0] = [d0@0 d1@0 d2@0 d3@0], x[1] = [..@2]
        const auto y = __builtin_amdgcn_permlane32_swap((unsigned)e[1], (unsigned)f[1], false, false);   // y[0] = [..@1], y[1] = [..@3]
        *reinterpret_cast<u32x4*>(cbase + blk * c_row16 + c_lane) = u32x4{(unsigned)x[0], (unsigned)y[0], (unsigned)x[1], (unsigned)y[1]};
    }
    asm volatile("s_nop 10" ::: "memory");
}

// vector-memory instructions (loads + stores; they share the in-order vmcnt queue) one epilogue_pair_fast<EPI> issues
template <int EPI>
constexpr int epi_pair_vmem_ops() {
    if ((EPI & EPI_SWIGLU) != 0) return 8;      // epilogue_pair_swiglu: one 16-byte store per 16-row block, no loads
    return 16 + ((EPI & EPI_BIAS) ? 4 : 0) + ((EPI & EPI_RES) ? 16 : 0) +
           ((EPI & EPI_ROPE) ? ((EPI & EPI_ROPE_LUT) ? 8 : 32) : 0) + ((EPI & EPI_LNFOLD) ? 8 : 0) +
           ((EPI & EPI_ROWSTAT) ? 8 : 0);
}

// Returns the number of vector-memory instructions issued when that is known exactly (pipelined path), else -1.
template <typename T, int EPI>
__device__ __forceinline__ int epilogue_wave_pair(const EpiArgs& p, int mb, int nb, int M, int N, int lane,
                                                  f32x4 (&acc0)[4][4], f32x4 (&acc1)[4][4]) {
#ifndef COGS_EPI_NOPAIR   // (A/B builds; also the build's fallback when the vmem-count check fails)
    if constexpr (sizeof(T) == 2 && EPI == EPI_SWIGLU) {
        // N / 2 output columns: the wave's 64 input columns are 32 outputs = 64 bytes per row
        if (mb + 128 <= M && nb + 64 <= N && (p.ldc & 7) == 0 && (reinterpret_cast<unsigned long>(p.C) & 15) == 0) {
            epilogue_pair_swiglu(p, mb, nb, lane, acc0, acc1);
            return epi_pair_vmem_ops<EPI>();
        }
    }
    if constexpr (sizeof(T) == 2 && (EPI & (EPI_SWIGLU | EPI_F32OUT | EPI_GENERIC | EPI_NOSTORE)) == 0) {
        bool fast = mb + 128 <= M && nb + 64 <= N && (N & 31) == 0 && (reinterpret_cast<unsigned long>(p.C) & 15) == 0;
        if constexpr ((EPI & EPI_HM) != 0) fast = fast && (p.hm_cols & 63) == 0 && (p.hm_hd & 7) == 0;
        else fast = fast && (p.ldc & 7) == 0;
        if constexpr ((EPI & EPI_RES) != 0) fast = fast && (p.ldr & 7) == 0 && (reinterpret_cast<unsigned long>(p.R) & 15) == 0;
        if constexpr ((EPI & EPI_ROPE) != 0) {
            fast = fast && p.rope_sin == nullptr && p.head_dim >= 64 && (nb + 64 <= p.rope_cols || nb >= p.rope_cols) &&
                   (nb + 64 <= p.q_cols || nb >= p.q_cols);
            if (fast) {
                if (nb < p.rope_cols) { epilogue_pair_fast<EPI>(p, mb, nb, lane, acc0, acc1); return epi_pair_vmem_ops<EPI>(); }
                epilogue_pair_fast<(EPI & ~(EPI_ROPE | EPI_ROPE_LUT))>(p, mb, nb, lane, acc0, acc1);
                return epi_pair_vmem_ops<(EPI & ~(EPI_ROPE | EPI_ROPE_LUT))>();
            }
        } else {
            if (fast) { epilogue_pair_fast<EPI>(p, mb, nb, lane, acc0, acc1); return epi_pair_vmem_ops<EPI>(); }
        }
    }
#endif
    epilogue_wave<T, (EPI & ~EPI_ROPE_LUT)>(p, mb, nb, M, N, lane, acc0);          // boundary tiles: per-row table
    epilogue_wave<T, (EPI & ~EPI_ROPE_LUT)>(p, mb + 64, nb, M, N, lane, acc1);
    return -1;
}

inline int cogs_fill_epi(const CogsGemm& g, EpiArgs* e) {
    if (g.act == COGS_ACT_SWIGLU && (g.bias || g.residual || g.out_f32)) return COGS_E_INVALID;
    if (g.rope_cos && (g.head_dim <= 0 || g.head_dim % 4 != 0 || g.rope_cols % g.head_dim != 0)) return COGS_E_INVALID;
    if (g.q_scale != 1.f && (!g.rope_cos || g.q_cols % 4 != 0 || g.q_cols > g.rope_cols)) return COGS_E_INVALID;   // rope epilogues only
    e->C = (char*)g.C; e->ldc = g.ldc;
    e->bias = g.bias;
    e->R = (const char*)g.residual; e->ldr = g.ldr;
    e->act = g.act; e->out_f32 = g.out_f32;
    e->rope_cos = g.rope_cos; e->rope_sin = g.rope_sin;
    e->rope_pairs = g.head_dim / 2; e->rope_cols = g.rope_cols;
    e->head_dim = g.head_dim > 0 ? g.head_dim : 4;
    e->q_scale = g.q_scale; e->q_cols = g.q_scale != 1.f ? g.q_cols : 0;
    e->rope_rowpos = g.rope_rowpos; e->rope_lut_lds = 0; e->rope_maxpos = g.rope_maxpos;
    e->stat_part = g.row_stats; e->stat_tiles = g.N / 64;
    e->ln_ab = g.ln_ab; e->col_c = g.col_c;
    e->hm_rows = 0; e->hm_hd = 1; e->hm_cols = 1;
    if (g.hm_rows > 0) {
        // head-major output: bf16 q | k | v blocks of whole heads, every column a rotary / plain head column
        if (g.head_dim <= 0 || g.head_dim % 8 != 0 || g.hm_cols <= 0 || g.hm_cols % g.head_dim != 0 || g.N % g.hm_cols != 0 ||
            g.out_f32 || g.act != COGS_ACT_NONE || g.residual || g.row_stats || g.hm_rows < g.M ||
            (double)g.hm_rows * g.hm_cols * 2.0 >= 4294967296.0)          // a lane's offset inside one q / k / v block is 32 bits
            return COGS_E_INVALID;
        e->hm_rows = g.hm_rows; e->hm_hd = g.head_dim; e->hm_cols = g.hm_cols;
    }
    if (g.row_stats && (g.N % 64 != 0 || g.out_f32 || g.act == COGS_ACT_SWIGLU)) return COGS_E_INVALID;
    if (g.ln_ab && (!g.col_c || g.act == COGS_ACT_SWIGLU || g.N % 4 != 0)) return COGS_E_INVALID;
    return COGS_OK;
}

// the compile-time mask for a descriptor, or EPI_GENERIC when the combination has no specialisation
inline int cogs_epi_mask(const CogsGemm& g) {
    int m = 0;
    if (g.row_stats) m |= EPI_ROWSTAT;
    if (g.ln_ab) m |= EPI_LNFOLD | EPI_BIAS;     // col_c plays the bias role
    if (g.bias) m |= EPI_BIAS;
    if (g.residual) m |= EPI_RES;
    if (g.rope_cos) m |= EPI_ROPE;
    if (g.act == COGS_ACT_GELU_TANH) m |= EPI_GELU_TANH;
    if (g.act == COGS_ACT_GELU_ERF) m |= EPI_GELU_ERF;
    if (g.act == COGS_ACT_SWIGLU) m |= EPI_SWIGLU;
    if (g.out_f32) m |= EPI_F32OUT;
    if (g.hm_rows > 0) m |= EPI_HM;
    switch (m) {
        case 0: case EPI_BIAS: case EPI_RES: case EPI_BIAS | EPI_RES: case EPI_BIAS | EPI_ROPE:
        case EPI_BIAS | EPI_GELU_TANH: case EPI_BIAS | EPI_GELU_ERF: case EPI_SWIGLU: case EPI_F32OUT:
        case EPI_BIAS | EPI_ROWSTAT: case EPI_BIAS | EPI_RES | EPI_ROWSTAT:
        case EPI_BIAS | EPI_ROPE | EPI_LNFOLD: case EPI_BIAS | EPI_GELU_TANH | EPI_LNFOLD: case EPI_BIAS | EPI_LNFOLD:
        case EPI_BIAS | EPI_ROPE | EPI_HM: case EPI_BIAS | EPI_ROPE | EPI_LNFOLD | EPI_HM:
            return m;
        default:
            return EPI_GENERIC;
    }
}
