// MFMA GEMM with fused epilogues for the ViT encoder, projector and Qwen2 linears.
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )        A, W both K-contiguous ("TN")
//
// Replaces the torch ops the reference calls through nn.Linear / nn.Conv2d:
//   model/modeling_videollama3_encoder.py:194-210 (patch embed as GEMM), :246-248
//   (q/k/v), :275 (out_proj), :369-373 (fc1/gelu/fc2), :388-391 (residual adds);
//   model/cogreasoner_chat.py:179-211 (projector); the Qwen2 linears of
//   transformers' modeling_qwen2 (q/k/v(+bias), o, gate/up, down, lm_head).
//
// Design (gfx950):
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave,
//     4x4 MFMA tiles of 16x16), 128-byte K slab per step (64 bf16 / 32 fp32).
//   * both operands staged HBM->LDS by direct LDS-DMA (global_load_lds_dwordx4),
//     two LDS buffers; the LDS image is lane-linear, the XOR swizzle lives on the
//     per-lane SOURCE address and on the ds_read address (chunk ^= row&7), which
//     makes every ds_read_b128 fragment read conflict free.
//   * weights are the MFMA A operand, activations the B operand, so that a lane ends
//     up holding 4 CONSECUTIVE output columns of one row: bias / residual / rope /
//     swiglu epilogues and the 8-byte stores are all lane-local.
//   * the fp32 instantiation (parity mode) keeps the same byte layout and uses the
//     exact-f32 MFMA (v_mfma_f32_16x16x4_f32).
//   * blocks are remapped XCD-aware (bijective) and walked in groups of 8 row tiles
//     so that co-resident blocks of one XCD share A/W panels in its L2.
#include "common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int ROW_BYTES = 128;               // one tile row of the K slab
constexpr int TILE_BYTES = BM * ROW_BYTES;   // 16 KiB per operand per buffer
constexpr int GROUP_M = 8;

template <typename T> struct ElemCfg;
template <> struct ElemCfg<bf16_t> { static constexpr int BK = 64; };
template <> struct ElemCfg<float> { static constexpr int BK = 32; };

struct GemmArgs {
    const char* A; long lda;   // bytes per row
    const char* W; long ldw;   // bytes per row
    int M, N, K;
    int nbm, nbn;
    EpiArgs epi;
};

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BK = ElemCfg<T>::BK;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    // ---- block -> tile (XCD-aware, bijective; then grouped walk) ----
    const int nb = p.nbm * p.nbn;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nb >> 3, r = nb & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int per_group = GROUP_M * p.nbn;
    const int grp = bid / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = min(p.nbm - first_m, GROUP_M);
    const int tm = first_m + (bid % per_group) % gsz;
    const int tn = (bid % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging addresses: wave w issues 4 LDS-DMA pieces per operand; piece i
    //      covers tile rows w*32+i*8 .. +7, lane -> (row = +lane>>3, chunk = lane&7) ----
    const char* a_src[4];
    const char* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wid * 32 + i * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (r & 7);
        const int am = min(m0 + r, p.M - 1);
        const int wr = min(n0 + r, p.N - 1);
        a_src[i] = p.A + (long)am * p.lda + c * 16;
        w_src[i] = p.W + (long)wr * p.ldw + c * 16;
    }
    char* const lds_wave = smem + wid * 4096;  // + buf*32768 + (W ? 16384 : 0) + i*1024

    // fragment read offsets (within a 16-row group): row = lane&15, chunk = s*4 + lane>>4
    int foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
        foff[s] = (lane & 15) * ROW_BYTES + ((((s << 2) + (lane >> 4)) ^ (lane & 7)) << 4);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int KT = p.K / BK;

    // prologue: stage slab 0 into buffer 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        glds16(a_src[i], lds_wave + i * 1024);
        glds16(w_src[i], lds_wave + TILE_BYTES + i * 1024);
    }
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) {
            const long ko = (long)(kt + 1) * ROW_BYTES;
            char* dst = lds_wave + (cur ^ 1) * 2 * TILE_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                glds16(a_src[i] + ko, dst + i * 1024);
                glds16(w_src[i] + ko, dst + TILE_BYTES + i * 1024);
            }
        }
        const char* As = smem + cur * 2 * TILE_BYTES + wm * 64 * ROW_BYTES;
        const char* Ws = smem + cur * 2 * TILE_BYTES + TILE_BYTES + wn * 64 * ROW_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const u32x4*>(As + i * 16 * ROW_BYTES + foff[s]);
                wf[i] = *reinterpret_cast<const u32x4*>(Ws + i * 16 * ROW_BYTES + foff[s]);
            }
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, wf[ni]), __builtin_bit_cast(bf16x8, af[mi]),
                            acc[mi][ni], 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                __uint_as_float(wf[ni][e]), __uint_as_float(af[mi][e]),
                                acc[mi][ni], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m][n..n+3], m = ..+(lane&15), n = ..+4*(lane>>4) ----
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + ((lane >> 4) << 2);
            if (n >= p.N) continue;
            epilogue4<T>(p.epi, m, n, acc[mi][ni]);
        }
    }
}

}  // namespace

int cogs_k_gemv(hipStream_t st, const CogsGemm& g);

int cogs_k_gemm(hipStream_t st, const CogsGemm& g) {
    const int es = g.dtype == COGS_DT_BF16 ? 2 : 4;
    const int BK = g.dtype == COGS_DT_BF16 ? 64 : 32;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return COGS_E_INVALID;
    if (g.K % BK != 0 || g.N % 4 != 0) return COGS_E_INVALID;
    if ((g.lda * es) % 16 != 0 || (g.ldw * es) % 16 != 0) return COGS_E_INVALID;
    if (g.M == 1) return cogs_k_gemv(st, g);
    GemmArgs p;
    const int rc = cogs_fill_epi(g, &p.epi);
    if (rc != COGS_OK) return rc;
    p.A = (const char*)g.A; p.lda = g.lda * es;
    p.W = (const char*)g.W; p.ldw = g.ldw * es;
    p.M = g.M; p.N = g.N; p.K = g.K;
    p.nbm = (g.M + BM - 1) / BM; p.nbn = (g.N + BN - 1) / BN;
    const int grid = p.nbm * p.nbn;
    const size_t lds = 4 * TILE_BYTES;
    if (g.dtype == COGS_DT_BF16) {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr = true;
        }
        hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, dim3(grid), dim3(256), lds, st, p);
    } else {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr = true;
        }
        hipLaunchKernelGGL(gemm_tn_kernel<float>, dim3(grid), dim3(256), lds, st, p);
    }
    return COGS_LAUNCH_CHECK();
}
