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
#include <stdlib.h>

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

template <typename T, int EPI>
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

    // fragment sets are register double buffered (see the 256x128 kernel below for the TOUCH rationale)
    u32x4 af[2][4], wf[2][4];
    auto load_frags = [&](int cur, int s, int set) {
        const char* As = smem + cur * 2 * TILE_BYTES + wm * 64 * ROW_BYTES + foff[s];
        const char* Ws = smem + cur * 2 * TILE_BYTES + TILE_BYTES + wn * 64 * ROW_BYTES + foff[s];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[set][i] = *reinterpret_cast<const u32x4*>(As + i * 16 * ROW_BYTES);
            wf[set][i] = *reinterpret_cast<const u32x4*>(Ws + i * 16 * ROW_BYTES);
        }
    };
    auto mma = [&](int set) {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, wf[set][ni]), __builtin_bit_cast(bf16x8, af[set][mi]),
                        acc[mi][ni], 0, 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            __uint_as_float(wf[set][ni][e]), __uint_as_float(af[set][mi][e]),
                            acc[mi][ni], 0, 0, 0);
        }
    };
#define COGS_TOUCH1(S)                                                                                       \
    asm volatile("" : "+v"(af[S][0]), "+v"(af[S][1]), "+v"(af[S][2]), "+v"(af[S][3]), "+v"(wf[S][0]), \
                 "+v"(wf[S][1]), "+v"(wf[S][2]), "+v"(wf[S][3]))

    // prologue: stage slab 0 into buffer 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        glds16(a_src[i], lds_wave + i * 1024);
        glds16(w_src[i], lds_wave + TILE_BYTES + i * 1024);
    }
    __syncthreads();
    load_frags(0, 0, 0);

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
        COGS_TOUCH1(0);
        load_frags(cur, 1, 1);
        __builtin_amdgcn_s_setprio(1);
        mma(0);
        __builtin_amdgcn_s_setprio(0);
        COGS_TOUCH1(1);
        __syncthreads();   // next slab landed (vmcnt(0)) and nobody still reads this one
        if (kt + 1 < KT) load_frags(cur ^ 1, 0, 0);
        __builtin_amdgcn_s_setprio(1);
        mma(1);
        __builtin_amdgcn_s_setprio(0);
    }
#undef COGS_TOUCH1

    // ---- epilogue: lane holds C[m][n..n+3], m = ..+(lane&15), n = ..+4*(lane>>4) ----
    epilogue_tile<T, EPI>(p.epi, m0 + wm * 64, n0 + wn * 64, p.M, p.N, lane, acc);
}


// ---------------------------------------------------------------------------------------------
// Large-M variant: 256x128 output tile, 8 waves (4x2, 64x64 each), 3-deep LDS ring (3 x 48 KiB),
// LDS-DMA prefetch two K slabs ahead that stays in flight ACROSS the barrier: one raw s_barrier per
// K step, counted s_waitcnt vmcnt(6) (= the 6 pieces of the next slab may still be in flight), all
// LDS in one array (guide section 5, "Pipelining across barriers"). One workgroup per CU.
constexpr int BM2 = 256;
constexpr int SLAB2 = (BM2 + BN) * ROW_BYTES;   // 48 KiB per K slab (A 32 KiB then W 16 KiB)
constexpr int PIECES2 = SLAB2 / 1024 / 8;       // 6 LDS-DMA pieces per wave per slab

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_tn_256x128_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BK = ElemCfg<T>::BK;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

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
    const int m0 = tm * BM2, n0 = tn * BN;

    // piece pc (0..47) of a slab: pc < 32 -> A rows 8pc..8pc+7, else W rows 8(pc-32)..; wave w owns 6w..6w+5
    const char* src[PIECES2];
#pragma unroll
    for (int i = 0; i < PIECES2; ++i) {
        const int pc = wid * PIECES2 + i;
        const bool is_a = pc < 32;
        const int r = (is_a ? pc : pc - 32) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (r & 7);
        src[i] = is_a ? p.A + (long)min(m0 + r, p.M - 1) * p.lda + c * 16
                      : p.W + (long)min(n0 + r, p.N - 1) * p.ldw + c * 16;
    }
    char* const lds_wave = smem + wid * PIECES2 * 1024;

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
    auto stage = [&](int kt, int buf) {
        const long ko = (long)kt * ROW_BYTES;
        char* dst = lds_wave + buf * SLAB2;
#pragma unroll
        for (int i = 0; i < PIECES2; ++i) glds16(src[i] + ko, dst + i * 1024);
    };

    // fragment registers are double buffered: the ds_reads of the NEXT half step are issued before the
    // MFMAs of the current one, so LDS latency hides under the wave's own matrix work.
    u32x4 af[2][4], wf[2][4];
    auto load_frags = [&](int buf, int s, int set) {
        const char* As = smem + buf * SLAB2 + wm * 64 * ROW_BYTES + foff[s];
        const char* Ws = smem + buf * SLAB2 + BM2 * ROW_BYTES + wn * 64 * ROW_BYTES + foff[s];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[set][i] = *reinterpret_cast<const u32x4*>(As + i * 16 * ROW_BYTES);
            wf[set][i] = *reinterpret_cast<const u32x4*>(Ws + i * 16 * ROW_BYTES);
        }
    };
    auto mma = [&](int set) {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, wf[set][ni]), __builtin_bit_cast(bf16x8, af[set][mi]),
                        acc[mi][ni], 0, 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            __uint_as_float(wf[set][ni][e]), __uint_as_float(af[set][mi][e]),
                            acc[mi][ni], 0, 0, 0);
        }
    };

    stage(0, 0);
    if (KT > 1) stage(1, 1);
    if (KT > 2) stage(2, 2);
    if (KT > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (KT > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(0, 0, 0);
    int buf = 0;
    // TOUCH(set): makes the compiler place its wait for that fragment set HERE (before newer ds_reads are
    // issued): hipcc cannot count lgkmcnt across the loop back-edge and would otherwise wait lgkmcnt(0) right
    // after issuing the next set, exposing one LDS latency per step.
#define COGS_TOUCH(S)                                                                                        \
    asm volatile("" : "+v"(af[S][0]), "+v"(af[S][1]), "+v"(af[S][2]), "+v"(af[S][3]), "+v"(wf[S][0]), \
                 "+v"(wf[S][1]), "+v"(wf[S][2]), "+v"(wf[S][3]))
    for (int kt = 0; kt < KT; ++kt) {
        const int nbuf = buf == 2 ? 0 : buf + 1;
        COGS_TOUCH(0);
        load_frags(buf, 1, 1);
        __builtin_amdgcn_s_setprio(1);
        mma(0);
        __builtin_amdgcn_s_setprio(0);
        COGS_TOUCH(1);   // every read this wave made of slab kt is complete from here on
        if (kt + 1 < KT) {
            // slab kt+1 must have landed (own pieces: counted vmcnt; everyone's: barrier); after the barrier
            // no wave reads slab kt any more, so its buffer is restaged with slab kt+3
            if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 3 < KT) stage(kt + 3, buf);
            load_frags(nbuf, 0, 0);
        }
        __builtin_amdgcn_s_setprio(1);
        mma(1);
        __builtin_amdgcn_s_setprio(0);
        buf = nbuf;
    }
#undef COGS_TOUCH

    epilogue_tile<T, EPI>(p.epi, m0 + wm * 64, n0 + wn * 64, p.M, p.N, lane, acc);
}

}  // namespace

int cogs_k_gemv(hipStream_t st, const CogsGemm& g);

namespace {

template <typename T, int EPI>
void launch_small(hipStream_t st, const GemmArgs& p, int grid) {
    const size_t lds = 4 * TILE_BYTES;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((gemm_tn_kernel<T, EPI>), dim3(grid), dim3(256), lds, st, p);
}
template <typename T, int EPI>
void launch_big(hipStream_t st, const GemmArgs& p, int grid) {
    const size_t lds = 3 * SLAB2;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_256x128_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((gemm_tn_256x128_kernel<T, EPI>), dim3(grid), dim3(512), lds, st, p);
}
template <typename T>
void dispatch(hipStream_t st, const GemmArgs& p, int grid, int mask, bool big) {
#define COGS_EPI_CASE(E) \
    case E: if (big) launch_big<T, E>(st, p, grid); else launch_small<T, E>(st, p, grid); break;
    switch (mask) {
        COGS_EPI_CASE(0)
        COGS_EPI_CASE(EPI_BIAS)
        COGS_EPI_CASE(EPI_RES)
        COGS_EPI_CASE(EPI_BIAS | EPI_RES)
        COGS_EPI_CASE(EPI_BIAS | EPI_ROPE)
        COGS_EPI_CASE(EPI_BIAS | EPI_GELU_TANH)
        COGS_EPI_CASE(EPI_BIAS | EPI_GELU_ERF)
        COGS_EPI_CASE(EPI_SWIGLU)
        COGS_EPI_CASE(EPI_F32OUT)
        default: if (big) launch_big<T, EPI_GENERIC>(st, p, grid); else launch_small<T, EPI_GENERIC>(st, p, grid); break;
    }
#undef COGS_EPI_CASE
}

}  // namespace

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
    static const bool env_small = getenv("COGS_GEMM_SMALL") != nullptr;
    const bool big = g.M >= 512 && !g.force_small_tile && !env_small;
    p.nbm = (g.M + (big ? BM2 : BM) - 1) / (big ? BM2 : BM);
    p.nbn = (g.N + BN - 1) / BN;
    const int grid = p.nbm * p.nbn;
    const int mask = cogs_epi_mask(g);
    if (g.dtype == COGS_DT_BF16) dispatch<bf16_t>(st, p, grid, mask, big);
    else dispatch<float>(st, p, grid, mask, big);
    return COGS_LAUNCH_CHECK();
}
