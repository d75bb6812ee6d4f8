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
#include "debug.h"
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
    const float* rope_lut; int rope_lut_bytes;   // EPI_ROPE_LUT: global LUT copied to LDS behind the ring at kernel start
    int group_m;                 // ping-pong kernel: row blocks per group of the tile walk (L2 footprint of an XCD)
    int cf;                      // whole-line ping-pong kernel: tiles of a FULL group = group_m * nbn (+ its tall tiles: see TALL4)
    int ntiles;                  //   ... and the tiles of the whole launch
    int epi_serial;              // whole-line kernel, A/B runs (debug switch gemm_epi_serial): both groups' epilogues behind the tile's last barrier
    unsigned long long* trace;   // diagnostics (debug switch gemm_trace): per-tile s_memtime stamps of WG 0, waves 0 and 4
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
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    epilogue_wave<T, EPI>(p.epi, m0 + wm * 64, n0 + wn * 64, p.M, p.N, lane, acc);
}


// ---------------------------------------------------------------------------------------------
// Large-M variant: 256x128 output tile, 8 waves (4x2, 64x64 each), 3-deep LDS ring (3 x 48 KiB),
// LDS-DMA prefetch two K slabs ahead that stays in flight ACROSS the barrier: one raw s_barrier per
// K step, counted s_waitcnt vmcnt(6) (= the 6 pieces of the next slab may still be in flight), all
// LDS in one array (guide section 5, "Pipelining across barriers"). One workgroup per CU.
constexpr int BM2 = 256;
constexpr int SLAB2 = (BM2 + BN) * ROW_BYTES;   // 48 KiB per K slab (A 32 KiB then W 16 KiB)
constexpr int PIECES2 = SLAB2 / 1024 / 8;       // 6 LDS-DMA pieces per wave per slab
constexpr int PERSISTENT_WGS = 256;             // one workgroup per CU (MI355X: 256 CUs)

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_tn_256x128_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BK = ElemCfg<T>::BK;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int nb = p.nbm * p.nbn;
    const int KT = p.K / BK;

    // PERSISTENT: this workgroup walks tiles t = blockIdx.x, +gridDim.x, ... of the XCD-aware order; the
    // K-slab ring runs continuously ACROSS tiles, so the next tile's first slabs stream in while the
    // current tile finishes and runs its epilogue (no per-tile prologue bubble).
    auto tile_origin = [&](int t, int& m0, int& n0) {
        const int xcd = t & 7, q = nb >> 3, r = nb & 7;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
        const int per_group = GROUP_M * p.nbn;
        const int first_m = (bid / per_group) * GROUP_M;
        const int gsz = min(p.nbm - first_m, GROUP_M);
        m0 = (first_m + (bid % per_group) % gsz) * BM2;
        n0 = ((bid % per_group) / gsz) * BN;
    };

    // staging iterator: (tile st_t, slab st_kt) with that tile's per-lane source pointers.
    // piece pc (0..47) of a slab: pc < 32 -> A rows 8pc..8pc+7, else W rows 8(pc-32)..; wave w owns 6w..6w+5
    const char* src[PIECES2];
    int st_t = blockIdx.x, st_kt = 0;
    auto set_src = [&](int t) {
        int m0, n0;
        tile_origin(t, m0, n0);
#pragma unroll
        for (int i = 0; i < PIECES2; ++i) {
            const int pc = wid * PIECES2 + i;
            const bool is_a = pc < 32;
            const int r = (is_a ? pc : pc - 32) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (r & 7);
            src[i] = is_a ? p.A + (long)min(m0 + r, p.M - 1) * p.lda + c * 16
                          : p.W + (long)min(n0 + r, p.N - 1) * p.ldw + c * 16;
        }
    };
    char* const lds_wave = smem + wid * PIECES2 * 1024;
    // stage the iterator's slab into ring slot `slot` and advance; returns false when the stream is exhausted
    auto stage_next = [&](int slot) -> bool {
        if (st_t >= nb) return false;
        const long ko = (long)st_kt * ROW_BYTES;
        char* dst = lds_wave + slot * SLAB2;
#pragma unroll
        for (int i = 0; i < PIECES2; ++i) glds16(src[i] + ko, dst + i * 1024);
        if (++st_kt == KT) {
            st_kt = 0;
            st_t += gridDim.x;
            if (st_t < nb) set_src(st_t);
        }
        return true;
    };

    int foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
        foff[s] = (lane & 15) * ROW_BYTES + ((((s << 2) + (lane >> 4)) ^ (lane & 7)) << 4);

    // fragment registers are double buffered: the ds_reads of the NEXT half step are issued before the
    // MFMAs of the current one, so LDS latency hides under the wave's own matrix work.
    f32x4 acc[4][4];
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
    // TOUCH(set): makes the compiler place its wait for that fragment set HERE (before newer ds_reads are
    // issued): hipcc cannot count lgkmcnt across the loop back-edge and would otherwise wait lgkmcnt(0) right
    // after issuing the next set, exposing one LDS latency per step.
#define COGS_TOUCH(S)                                                                                        \
    asm volatile("" : "+v"(af[S][0]), "+v"(af[S][1]), "+v"(af[S][2]), "+v"(af[S][3]), "+v"(wf[S][0]), \
                 "+v"(wf[S][1]), "+v"(wf[S][2]), "+v"(wf[S][3]))

    if (st_t >= nb) return;
    set_src(st_t);
    // prologue: three slabs in flight, slab 0 landed
    int staged = 0;   // slabs staged so far minus slabs consumed so far (0..3)
    for (int i = 0; i < 3; ++i) staged += stage_next(i) ? 1 : 0;
    if (staged == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (staged == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(0, 0, 0);
    int buf = 0;
    int epi_stores = 0;   // store instructions the previous tile's epilogue left in flight

    for (int t = blockIdx.x; t < nb; t += gridDim.x) {
        int m0, n0;
        tile_origin(t, m0, n0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT; ++kt) {
            const int nbuf = buf == 2 ? 0 : buf + 1;
            COGS_TOUCH(0);
            load_frags(buf, 1, 1);
            __builtin_amdgcn_s_setprio(1);
            mma(0);
            __builtin_amdgcn_s_setprio(0);
            COGS_TOUCH(1);   // every read this wave made of the current slab is complete from here on
            --staged;        // current slab consumed; `staged` slabs follow it in the ring
            if (staged > 0) {
                // the next slab must have landed (own pieces: counted vmcnt -- at most the 6 pieces of the slab
                // after it may stay in flight; everyone's: barrier); after the barrier nobody reads the current
                // slab any more, so its ring slot is restaged
                // the 8 stores of the previous tile's (wide, interior) epilogue are newer than the pieces this wait is
                // about for the first two slabs of a tile: leave them in flight (see the ping-pong kernel)
                const bool relaxed = epi_stores == 8 && kt < 2;
                if (staged > 1) { if (relaxed) asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
                else { if (relaxed) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                staged += stage_next(buf) ? 1 : 0;
                load_frags(nbuf, 0, 0);
            }
            __builtin_amdgcn_s_setprio(1);
            mma(1);
            __builtin_amdgcn_s_setprio(0);
            buf = nbuf;
        }
        epilogue_wave<T, EPI>(p.epi, m0 + wm * 64, n0 + wn * 64, p.M, p.N, lane, acc);
        epi_stores = (sizeof(T) == 2 && (EPI & (EPI_SWIGLU | EPI_F32OUT | EPI_GENERIC | EPI_NOSTORE)) == 0 &&
                      (p.N & 31) == 0 && m0 + BM2 <= p.M && n0 + BN <= p.N) ? 8 : 0;
    }
#undef COGS_TOUCH
}


// ---------------------------------------------------------------------------------------------
// Ping-pong variant (bf16): 256x256 output tile, 8 waves = two groups of 4 (one wave of each group
// per SIMD), each wave 128x64. The K stream is cut into 32-wide K-tiles (64-byte LDS rows), 4-slot
// ring (4 x 32 KiB). Every K-tile is
//     L segment: ds_read the wave's fragments (12 x b128), issue its 4 LDS-DMA pieces, lgkmcnt(0)
//     --- s_barrier ---
//     C segment: 32 MFMA
//     --- s_barrier ---
// and group 1 runs ONE BARRIER BEHIND group 0, so on every SIMD one wave is in its C segment while the
// other is in its L segment. Persistent over tiles with a continuous K-tile stream (prefetch distance 3 K-tiles, one
// counted vmcnt per K-tile). The first version cut a K-tile into two such L/C pairs (upper / lower 64 rows: 16 fewer
// fragment registers); interval stamps (-DCOGS_GEMM_KSTAMPS build, tools/gemm_trace.py) put its L segments at 330-440
// cycles against 258 for 16 MFMAs, and with the staging thinned out (timing experiments) an interval still took
// ~100 cycles more than its MFMAs: the hand-over between the groups costs about as much as the imbalance. One pair
// per K-tile halves the hand-overs: -3..-6 % on every shape but one (in-run A/B: fc1+GELU 0.588 -> 0.569 ms,
// out-proj 0.224 -> 0.212, fc2 0.610 -> 0.594, Qwen2 gate/up 3.67 -> 3.45; down-proj unchanged). A 5-slot ring
// (distance 4) and `buffer_load ... lds` pieces instead of `global_load_lds` were both measured without gain.
constexpr int BM3 = 256, BN3 = 256;
constexpr int ROW3 = 64;                          // bytes per LDS row = 32 bf16
constexpr int SLOT3 = (BM3 + BN3) * ROW3;         // 32 KiB per K-tile
constexpr int RING3 = 4;                          // ring slots (5 = all 160 KiB of LDS: measured, no gain)
constexpr int DIST3 = RING3 - 1;                  // prefetch distance in K-tiles

__device__ __forceinline__ int swz3(int r) { return (0x78 >> (2 * ((r >> 2) & 3))) & 3; }   // [0,2,3,1]

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_tn_pp_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef bf16_t T;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2;          // 0: rows 0..127 of the tile, 1: rows 128..255; also the stagger group
    const int wc = wid & 3;
    const int nb = p.nbm * p.nbn;
    const int KT = p.K / 32;

    auto tile_origin = [&](int t, int& m0, int& n0) {
        const int xcd = t & 7, q = nb >> 3, r = nb & 7;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
        const int per_group = p.group_m * p.nbn;
        const int first_m = (bid / per_group) * p.group_m;
        const int gsz = min(p.nbm - first_m, p.group_m);
        m0 = (first_m + (bid % per_group) % gsz) * BM3;
        n0 = ((bid % per_group) / gsz) * BN3;
    };

    // staging: a K-tile is 32 pieces of 1 KiB (16 rows x 64 B); wave w owns pieces 4w..4w+3
    // (pieces 0..15 = A rows, 16..31 = W rows); lane -> row lane>>2, LDS chunk lane&3, source chunk swizzled
    const char* src[4];
    int st_t = blockIdx.x, st_kt = 0;
    auto set_src = [&](int t) {
        int m0, n0;
        tile_origin(t, m0, n0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pc = wid * 4 + i;
            const bool is_a = pc < 16;
            const int r = (is_a ? pc : pc - 16) * 16 + (lane >> 2);
            const int c = (lane & 3) ^ swz3(lane >> 2);
            src[i] = is_a ? p.A + (long)min(m0 + r, p.M - 1) * p.lda + c * 16
                          : p.W + (long)min(n0 + r, p.N - 1) * p.ldw + c * 16;
        }
    };
    char* const lds_wave = smem + wid * 4096;
    int st_slot = 0;
    // issue pieces 2h, 2h+1 of the staging iterator's K-tile; after h == 1 the iterator advances
    auto stage_half = [&](int h) {
        if (st_t >= nb) return;
        const long ko = (long)st_kt * ROW3;
        char* dst = lds_wave + st_slot * SLOT3;
        glds16(src[2 * h] + ko, dst + (2 * h) * 1024);
        glds16(src[2 * h + 1] + ko, dst + (2 * h + 1) * 1024);
        if (h == 1) {
            st_slot = st_slot + 1 == RING3 ? 0 : st_slot + 1;
            if (++st_kt == KT) {
                st_kt = 0;
                st_t += gridDim.x;
                if (st_t < nb) set_src(st_t);
            }
        }
    };

    // fragment read offsets: row r = lane&15, k-chunk g = lane>>4
    const int foff = (lane & 15) * ROW3 + (((lane >> 4) ^ swz3(lane & 15)) << 4);
    const int a_off = (grp * 128) * ROW3 + foff;                 // + mh*64*ROW3 + mi*16*ROW3
    const int w_off = BM3 * ROW3 + (wc * 64) * ROW3 + foff;      // + ni*16*ROW3

    f32x4 acc[2][4][4];
    u32x4 afr[4], afr2[4], wfr[4];   // A fragments of the upper / lower 64 rows, W fragments

    if (st_t >= nb) return;
    if constexpr ((EPI & EPI_ROPE_LUT) != 0) {
        // rotary LUT -> LDS behind the ring (published by the prologue barrier below)
        for (int i = tid * 16; i < p.rope_lut_bytes; i += 512 * 16)
            *reinterpret_cast<u32x4*>(smem + RING3 * SLOT3 + i) = *reinterpret_cast<const u32x4*>(
                reinterpret_cast<const char*>(p.rope_lut) + i);
    }
    set_src(st_t);
    const int my_tiles = (nb - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
    const int total = my_tiles * KT;          // K-tiles this workgroup consumes
    // prologue: K-tiles 0..DIST3-1 in flight (4 pieces each), K-tile 0 landed
    {
        int pre = total < DIST3 ? total : DIST3;
        for (int i = 0; i < pre; ++i) { stage_half(0); stage_half(1); }
        if (pre >= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (pre == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (pre == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr ((EPI & EPI_ROPE_LUT) != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LUT writes landed
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0

    // vmcnt counts loads, LDS-DMA AND stores in issue order. The 16 stores of a tile epilogue are NEWER than the
    // DMA pieces the next two K-tile waits are about, so those waits may leave them in flight (vmcnt(8+16));
    // with a plain vmcnt(8) every tile boundary stalls for the full write latency. The count is exact only for
    // interior tiles on the wide epilogue path (16 unpredicated store instructions); anything else keeps the
    // conservative count. `epi_stores` is what the previous tile's epilogue left in flight.
    // (A 256x128 ping-pong variant for N = 1152 was measured 10-15 % slower than the ring kernel: 64x64 per
    // wave makes the L segment -- 8 reads + 3 DMA pieces -- too heavy for 16 MFMAs.)
    int epi_ops = 0;   // vector-memory instructions of the previous tile's epilogue when known exactly, else 0
    constexpr int PAIR_OK = (EPI & (EPI_F32OUT | EPI_GENERIC | EPI_NOSTORE)) == 0 && ((EPI & EPI_SWIGLU) == 0 || EPI == EPI_SWIGLU);
    constexpr int OPS_A = epi_pair_vmem_ops<EPI>();                                   // rotary / plain tile
    constexpr int OPS_B = epi_pair_vmem_ops<(EPI & ~(EPI_ROPE | EPI_ROPE_LUT))>();    // tile right of rope_cols
    auto wait_next_ktile = [&](int ahead, int kt) {
        // K-tile g+1 has landed when at most the pieces of the K-tiles staged after it are outstanding:
        // min(ahead, DIST3) - 1 K-tiles of 4 pieces -- plus, for the first DIST3-1 K-tiles of a tile, every vector-
        // memory instruction of the previous tile's epilogue (stores AND its bias / residual / rotary loads: one
        // in-order queue), all of which were issued after those pieces. (The count must be an immediate; the
        // epilogue reports which of its two compile-time counts applies. A generic switch over the count costs
        // ~200 scalar cycles per K-tile -- measured slower than not relaxing at all.)
        if (PAIR_OK && epi_ops != 0 && kt < DIST3 - 1 && ahead >= DIST3) {
            // (the counter has 6 bits: a count clamped to 63 only waits for a few more operations than necessary)
            constexpr int WAIT_A = 4 * (DIST3 - 1) + OPS_A < 63 ? 4 * (DIST3 - 1) + OPS_A : 63;
            constexpr int WAIT_B = 4 * (DIST3 - 1) + OPS_B < 63 ? 4 * (DIST3 - 1) + OPS_B : 63;
            if (epi_ops == OPS_A) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_A) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_B) : "memory");
            return;
        }
        const int newer = (ahead < DIST3 ? ahead : DIST3) - 1;
        if (newer >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (newer == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    int slot = 0;
    int g = 0;   // index of the K-tile being consumed (0..total-1)
    const bool tracing = p.trace && blockIdx.x == 0 && wc == 0;
    int trace_i = 0;
    auto stamp = [&]() {
        if (tracing && trace_i < 96) {
            const unsigned long long tm = __builtin_amdgcn_s_memtime();
            if (lane == 0) p.trace[grp * 96 + trace_i] = tm;
            ++trace_i;
        }
    };
    for (int t = blockIdx.x; t < nb; t += gridDim.x) {
        int m0, n0;
        tile_origin(t, m0, n0);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        stamp();
#ifdef COGS_GEMM_KSTAMPS
        // diagnostic build only: where does an interval go? sums over the K-tiles of this tile, per wave group
        unsigned long long ks_L0 = 0, ks_W0 = 0, ks_C0 = 0, ks_X0 = 0;
        unsigned long long ks_prev = __builtin_amdgcn_s_memtime();
#define KSTAMP(acc_) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_ += now_ - ks_prev; ks_prev = now_; } while (0)
#else
#define KSTAMP(acc_) do {} while (0)
#endif
        for (int kt = 0; kt < KT; ++kt, ++g) {
            const char* base = smem + slot * SLOT3;
            // ---- L segment: the K-tile's 12 fragment reads and this wave's 4 staging pieces ----
#pragma unroll
            for (int i = 0; i < 4; ++i) wfr[i] = *reinterpret_cast<const u32x4*>(base + w_off + i * 16 * ROW3);
#pragma unroll
            for (int i = 0; i < 4; ++i) afr[i] = *reinterpret_cast<const u32x4*>(base + a_off + i * 16 * ROW3);
#pragma unroll
            for (int i = 0; i < 4; ++i) afr2[i] = *reinterpret_cast<const u32x4*>(base + a_off + 64 * ROW3 + i * 16 * ROW3);
            // K-tile g+3 -> ring slot of K-tile g-1: its last ds_reads (group 1, previous interval) were drained by
            // that wave's lgkmcnt(0) BEFORE the barrier that opened this interval (WAR safe).
            // (Measured: draining LDS reads before the barrier with prefetch distance 3 beats distance 2 with
            // the wait behind the barrier by 3-4 %.)
            stage_half(0);
            stage_half(1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(afr[0]), "+v"(afr[1]), "+v"(afr[2]), "+v"(afr[3]), "+v"(wfr[0]), "+v"(wfr[1]),
                         "+v"(wfr[2]), "+v"(wfr[3]));
            asm volatile("" : "+v"(afr2[0]), "+v"(afr2[1]), "+v"(afr2[2]), "+v"(afr2[3]));
            // K-tile g+1 must have landed before anyone reads it (interval after the next-but-one barrier for
            // group 0, after the next barrier for group 1): both groups wait at the end of THIS interval --
            // group 1 here (end of its L segment), group 0 at the end of its C segment below.
            const int ahead = total - 1 - g;   // K-tiles after the current one
            if (grp == 1) wait_next_ktile(ahead, kt);
            KSTAMP(ks_L0);
            __builtin_amdgcn_s_barrier();
            KSTAMP(ks_W0);
            // ---- C segment: 32 MFMAs (upper, then lower 64 rows of the wave's tile) ----
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[0][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, wfr[ni]), __builtin_bit_cast(bf16x8, afr[mi]), acc[0][mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[1][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, wfr[ni]), __builtin_bit_cast(bf16x8, afr2[mi]), acc[1][mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (grp == 0) wait_next_ktile(ahead, kt);
            KSTAMP(ks_C0);
            __builtin_amdgcn_s_barrier();
            KSTAMP(ks_X0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("" ::: "memory");
            slot = slot + 1 == RING3 ? 0 : slot + 1;
        }
#ifdef COGS_GEMM_KSTAMPS
        if (tracing && lane == 0 && t == (int)blockIdx.x) {   // first tile of workgroup 0
            unsigned long long* o = p.trace + 192 + grp * 8;
            o[0] = ks_L0; o[1] = ks_W0; o[2] = ks_C0; o[3] = ks_X0; o[4] = o[5] = o[6] = o[7] = 0;
        }
#endif
        // epilogue of this tile; it runs inside this group's next L interval, i.e. beside the other group's C
        stamp();
        stamp();
        const int ops = epilogue_wave_pair<T, EPI>(p.epi, m0 + grp * 128, n0 + wc * 64, p.M, p.N, lane, acc[0], acc[1]);
        stamp();
#ifdef COGS_EPI_CONSERVATIVE   // build fallback: the emitted vector-memory count could not be verified
        epi_ops = 0; (void)ops;
#else
        epi_ops = ops > 0 ? ops : 0;
#endif
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance group 1's extra barrier
}

// ---------------------------------------------------------------------------------------------
// Whole-line staging for the ping-pong kernel (round 4). The kernel above stages 32-wide K-tiles: an LDS-DMA piece is 16
// rows x 64 B, i.e. sixteen HALF 128-byte lines per wave instruction, and the CU's vector-memory path takes such pieces at
// 29-30 B/clk against 54-55 B/clk for pieces made of whole lines (tools/micro/ldsdma_rate.cpp, L2-resident rows, 8 waves
// per CU) -- at 30 B/clk a 256x256 tile's staging (32 KiB per 32-wide K-tile) costs as many cycles as its 1 024 cycles of
// MFMA. Here a piece is 8 rows x 128 B: the K stream is cut into 64-wide slabs (128-byte LDS rows, chunk ^= row & 7).
// Same wave layout, same two segments per 32-wide half of a slab (12 fragment reads | 32 MFMAs), group 1 one barrier
// behind group 0, so every output element sees the same MFMAs in the same order as in the kernel above (bit-identical
// results). LDS: a slab is two 32 KiB units, its 256 A rows and its 256 W rows; the ring holds FIVE units (all 160 KiB,
// so the rotary LUT variant stays with the kernel above) in the order A0 W0 A1 W1 A2 ...: while slab s is consumed, A(s),
// W(s), A(s+1) are resident or landing and the two positions slab s-1 left take W(s+1) -- issued in the FIRST load segment
// of slab s -- and A(s+2) -- issued in the SECOND. Every wave issues 4 pieces per load segment, so each of a slab's four
// intervals carries 16 pieces, as in the kernel above, but of whole lines. One counted vmcnt per slab: behind the units a
// slab needs, each wave has issued 12 newer pieces (A(s+2), W(s+2), A(s+3)), plus, across a tile boundary, the epilogue's
// loads and stores (exact-count rule as above, else a stricter count that waits for them too).
//
// TALL tiles (round 6). N = 1152 is 4.5 column blocks of 256 and N = 3456 is 13.5: as a whole 256x256 tile the ragged block
// spends half its MFMAs on columns that do not exist (9.9 % / 3.7 % of those launches' matrix-pipe cycles). It is walked as
// tiles of 384 rows x 128 columns instead, through the SAME ring with the SAME piece and wait counts: the "A unit" of a slab
// is rows 0..255 of the tile as always; the "W unit" carries rows 256..383 of the tile in its rows 0..127 (staged by waves
// 0..3) and the 128 weight rows in its rows 128..255 (waves 4..7). The 384 x 128 outputs are six 128 x 64 wave regions:
// wave (group g, column wc) takes rows 128 wc.., columns 64 g.. for wc < 3 (wc == 2 reads its A fragments from the W unit),
// and the two waves with wc == 3 -- one SIMD -- stage, keep the barriers, run their MFMAs on unused data and skip the epilogue.
// Every output element still sees the same MFMAs over the same K order, so a row's bits do not depend on which kind of tile
// produced it; three ragged half tiles cost two tile times instead of three (MFMAs spent on padding: 3.7 % instead of 11.1 % of
// the algorithmic count at N = 1152, 1.2 % instead of 3.7 % at N = 3456).
constexpr int ROW4 = 128;                         // bytes per LDS row = 64 bf16 = one line of the operand
constexpr int UNIT4 = 256 * ROW4;                 // 32 KiB: the A rows or the W rows of one slab
constexpr int RING4 = 5;
constexpr int TALL4 = 384;                        // rows of a tall tile (its columns: the <= 128 of the ragged column block)

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_tn_pp64_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert((EPI & EPI_ROPE_LUT) == 0, "no LDS left for the rotary LUT");
    typedef bf16_t T;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2;          // 0: rows 0..127 of the tile, 1: rows 128..255; also the stagger group
    const int wc = wid & 3;
    const int nb = p.ntiles;
    const int KS = p.K / 64;
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // Tile walk: groups of group_m row blocks; a group's whole 256x256 tiles column by column, then -- when N leaves a ragged
    // column block of <= 128 columns (p.cf > group_m * nbn; p.nbn then counts the WHOLE column blocks only) -- its TALL tiles:
    // 384 rows x 128 columns of that ragged block (kind 1, see TALL4 below). Only the last group can be short, so a tile's
    // group is bid / cf.
    auto tile_origin = [&](int t, int& m0, int& n0, int& kind) {
        const int xcd = t & 7, q = nb >> 3, r = nb & 7;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
        const int gi = bid / p.cf, idx = bid - gi * p.cf;
        const int first_m = gi * p.group_m;
        const int gsz = min(p.nbm - first_m, p.group_m);
        const int nnorm = gsz * p.nbn;
        if (idx < nnorm) {
            m0 = (first_m + idx % gsz) * BM3;
            n0 = (idx / gsz) * BN3;
            kind = 0;
        } else {
            m0 = first_m * BM3 + (idx - nnorm) * TALL4;
            n0 = p.nbn * BN3;
            kind = 1;
        }
    };

    // staging: a unit is 32 pieces of 1 KiB (8 rows x 128 B); wave w owns pieces 4w..4w+3 of EVERY unit (rows 32w..32w+31).
    // lane -> row lane>>3 of the piece, LDS chunk lane&7, source chunk (lane&7) ^ (row&7). Addresses: a uniform base
    // (operand + first row of the tile + K offset) + a 32-bit lane offset. The A and the W stream advance separately (the
    // A stream runs a unit ahead of the W stream and may already be in the next tile).
    unsigned long long base_a = 0, base_w = 0;
    int voff_a[4], voff_w[4];
    int sa_t = blockIdx.x, sa_ks = 0, sw_t = blockIdx.x, sw_ks = 0;
    auto set_src_a = [&](int t) {
        int m0, n0, kind;
        tile_origin(t, m0, n0, kind);
        base_a = (unsigned long long)p.A + (unsigned long long)m0 * p.lda;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = m0 + wid * 32 + i * 8 + (lane >> 3);
            voff_a[i] = (min(r, p.M - 1) - m0) * (int)p.lda + (((lane & 7) ^ (lane >> 3)) << 4);
        }
    };
    auto set_src_w = [&](int t) {
        int m0, n0, kind;
        tile_origin(t, m0, n0, kind);
        if (kind != 0 && wid < 4) {
            // tall tile, waves 0..3: rows 256..383 of the tile -> rows 0..127 of the W unit (the offset is unsigned: base = tile origin)
            base_w = (unsigned long long)p.A + (unsigned long long)m0 * p.lda;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = m0 + 256 + wid * 32 + i * 8 + (lane >> 3);
                voff_w[i] = (min(r, p.M - 1) - m0) * (int)p.lda + (((lane & 7) ^ (lane >> 3)) << 4);
            }
        } else {
            // weight rows: 256 of them over all eight waves, or (tall tile) 128 over waves 4..7 -> rows 128..255 of the unit
            const int w0 = kind != 0 ? (wid - 4) * 32 : wid * 32;
            base_w = (unsigned long long)p.W + (unsigned long long)n0 * p.ldw;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = n0 + w0 + i * 8 + (lane >> 3);
                voff_w[i] = (min(r, p.N - 1) - n0) * (int)p.ldw + (((lane & 7) ^ (lane >> 3)) << 4);
            }
        }
    };
    const unsigned lds_wave = smem_lds + wid * 4096;
    int ipos = 0;                                    // ring position of the next unit to issue
    auto issue4 = [&](unsigned long long b, const int (&voff)[4]) {
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)b), bhi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        const char* bp = (const char*)(((unsigned long long)bhi << 32) | blo);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_wave + ipos * UNIT4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(dst + i * 1024), "v"(voff[i]), "s"(bp) : "memory");
        ipos = ipos + 1 == RING4 ? 0 : ipos + 1;
    };
    auto issue_a = [&]() {
        if (sa_t >= nb) return;
        issue4(base_a + (unsigned long long)sa_ks * ROW4, voff_a);
        if (++sa_ks == KS) {
            sa_ks = 0;
            sa_t += gridDim.x;
            if (sa_t < nb) set_src_a(sa_t);
        }
    };
    auto issue_w = [&]() {
        if (sw_t >= nb) return;
        issue4(base_w + (unsigned long long)sw_ks * ROW4, voff_w);
        if (++sw_ks == KS) {
            sw_ks = 0;
            sw_t += gridDim.x;
            if (sw_t < nb) set_src_w(sw_t);
        }
    };

    // fragment read offsets inside a unit: row r = lane&15, k-chunk (lane>>4) of the slab's first half; the second half is ^ 64
    const int foff = (lane & 15) * ROW4 + (((lane >> 4) ^ (lane & 7)) << 4);
    int a_off, w_off;      // per tile: the wave's first A row / W row inside their units (+ 64*ROW4 + mi*16*ROW4, + ni*16*ROW4)
    int a_in_w = 0;        // tall tile, wc == 2: the wave's A rows are rows 0..127 of the W unit
    int idle = 0;          // tall tile, wc == 3: no region of the tile

    f32x4 acc[2][4][4];
    u32x4 afr[4], afr2[4], wfr[4];

    if (sa_t >= nb) return;
    set_src_a(sa_t);
    set_src_w(sw_t);
    const int my_tiles = (nb - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
    const int total = my_tiles * KS;          // slabs this workgroup consumes
    // prologue: A0 W0 A1 in flight, slab 0 landed (a stream that has run out issues nothing)
    issue_a(); issue_w(); issue_a();
    if (total >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0

    int epi_ops = 0;       // vector-memory instructions of the previous tile's epilogue when known exactly, else 0
    constexpr int PAIR_OK = (EPI & (EPI_F32OUT | EPI_GENERIC | EPI_NOSTORE)) == 0 && ((EPI & EPI_SWIGLU) == 0 || EPI == EPI_SWIGLU);
    constexpr int OPS_A = epi_pair_vmem_ops<EPI>();                                   // rotary / plain tile
    constexpr int OPS_B = epi_pair_vmem_ops<(EPI & ~(EPI_ROPE | EPI_ROPE_LUT))>();    // tile right of rope_cols
    // Slab g+1 = A(g+1), W(g+1) has landed when at most the 4 pieces of A(g+2) -- the only unit a wave issues after
    // W(g+1) -- are outstanding. Across a tile boundary W(g+1) goes out in FRONT of the previous tile's epilogue, whose
    // loads and stores are then newer than it and may stay in flight too when their number is known exactly.
    bool pre_issued = false;
    auto wait_mode = [&](int g, bool after_epilogue) -> int {      // 0: vmcnt(0), 1: vmcnt(4), 2: 4 + OPS_A, 3: 4 + OPS_B
        const bool a2 = 2 * g + 4 <= 2 * total - 1;                   // A(g+2) exists
        int m = a2 ? 1 : 0;
        if (after_epilogue && epi_ops != -2) m = (PAIR_OK && epi_ops != 0 && a2) ? (epi_ops == OPS_A ? 2 : 3) : 0;
        return m;
    };
    auto wait_next_slab = [&](int mode) {
        constexpr int WAIT_A = 4 + OPS_A < 63 ? 4 + OPS_A : 63;
        constexpr int WAIT_B = 4 + OPS_B < 63 ? 4 + OPS_B : 63;
        if (mode == 4) return;          // the other group's turn to wait
        if (mode == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (mode == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (mode == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_A) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_B) : "memory");
    };

    int ra = 0, rw = 1;    // ring positions of the current slab's A and W unit
    int is_g1;             // scalar copy of grp (kept out of the compiler's lane-mask booleans)
    asm volatile("s_mov_b32 %0, %1" : "=s"(is_g1) : "s"(grp));
    int g = 0;             // index of the slab being consumed (0..total-1)
    for (int t = blockIdx.x; t < nb; t += gridDim.x) {
        int m0, n0, kind;
        tile_origin(t, m0, n0, kind);
        {
            // the wave's 128 x 64 region of the tile: whole tile (group -> rows, wc -> columns), tall tile (wc -> rows, group -> columns)
            const int rg = kind != 0 ? wc : grp, cg = kind != 0 ? grp : wc;
            asm volatile("s_mov_b32 %0, %1" : "=s"(a_in_w) : "s"(__builtin_amdgcn_readfirstlane((int)(kind != 0 && wc == 2))));
            asm volatile("s_mov_b32 %0, %1" : "=s"(idle) : "s"(__builtin_amdgcn_readfirstlane((int)(kind != 0 && wc == 3))));
            a_off = ((kind != 0 && wc >= 2) ? 0 : rg * 128) * ROW4 + foff;
            w_off = (kind != 0 ? 128 + cg * 64 : cg * 64) * ROW4 + foff;
            m0 += rg * 128;
            n0 += cg * 64;
        }
#ifdef COGS_GEMM_TSTAMPS     // diagnostic build (tools/gemm_trace.py): K loop / tile boundary of workgroup 0, waves 0 and 4, per tile
        const bool ts_on = p.trace && blockIdx.x == 0 && wc == 0;
        const int ts_i = (t - (int)blockIdx.x) / (int)gridDim.x;
        if (ts_on && ts_i < 24) { const unsigned long long tm_ = __builtin_amdgcn_s_memtime(); if (lane == 0) p.trace[grp * 96 + 4 * ts_i] = tm_; }
#endif
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef COGS_GEMM_KSTAMPS
        // diagnostic build only: where does a slab's time go? sums over the slabs of this tile, per wave group
        unsigned long long ks_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long ks_prev = __builtin_amdgcn_s_memtime();
#define KSTAMP4(i_) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ks_sum[i_] += now_ - ks_prev; ks_prev = now_; } while (0)
#else
#define KSTAMP4(i_) do {} while (0)
#endif
        // One slab. STEADY = a slab that is neither the first of its tile nor one of the stream's last two: its wait is
        // always vmcnt(4), written without a single scalar instruction behind the MFMAs of C1. Measured with interval stamps
        // (-DCOGS_GEMM_KSTAMPS build): whatever DECIDES at that point -- the compiler's lane-mask conversion (v_cndmask /
        // v_cmp writing an SGPR) or a tree of s_cmp / taken s_cbranch on a pinned SGPR -- costs 180-340 cycles per slab
        // there, against 4 cycles for nothing. Group 1, whose wait stands at the end of L1, executes the C1 wait too: by
        // then it has at most A(g+2) outstanding, so it falls through. The other slabs take the general form, with the
        // count chosen at the top of the slab and pinned into SGPRs (opaque s_mov).
        auto slab = [&](auto steady_tag, auto last_tag, const int ks) __attribute__((always_inline)) {
            constexpr bool STEADY = decltype(steady_tag)::value;
            constexpr bool LAST = decltype(last_tag)::value;      // last slab of the tile: its closing barrier belongs to the tile end below
            const char* uw = smem + rw * UNIT4;
            const char* ua = a_in_w ? uw : smem + ra * UNIT4;
            int wm_l1 = 4, wm_c1 = 4;
            if constexpr (!STEADY) {
                const int wmode = wait_mode(g, ks == 0 && g > 0);
                asm volatile("s_mov_b32 %0, %1" : "=s"(wm_l1) : "s"(__builtin_amdgcn_readfirstlane(grp == 1 ? wmode : 4)));
                asm volatile("s_mov_b32 %0, %1" : "=s"(wm_c1) : "s"(__builtin_amdgcn_readfirstlane(grp == 0 ? wmode : 4)));
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int ao = a_off ^ (half << 6), wo = w_off ^ (half << 6);   // k-chunk index ^ 4
                // ---- L segment: 12 fragment reads and this wave's 4 pieces of W(g+1) (first half) / A(g+2) (second) ----
#pragma unroll
                for (int i = 0; i < 4; ++i) wfr[i] = *reinterpret_cast<const u32x4*>(uw + wo + i * 16 * ROW4);
#pragma unroll
                for (int i = 0; i < 4; ++i) afr[i] = *reinterpret_cast<const u32x4*>(ua + ao + i * 16 * ROW4);
#pragma unroll
                for (int i = 0; i < 4; ++i) afr2[i] = *reinterpret_cast<const u32x4*>(ua + ao + 64 * ROW4 + i * 16 * ROW4);
                // W(g+1) -> position of A(g-1), A(g+2) -> position of W(g-1): both groups drained their reads of slab
                // g-1 (lgkmcnt(0)) before the barrier that opened this slab's first interval (WAR safe)
                // (measured and dropped, in-run A/B on the encoder step: the pieces in FRONT of the reads -0.8 %; no s_setprio
                // around the MFMA segment +-0; the upper A fragments of a slab's second half requested from inside the first
                // half's MFMA segment, behind their last use, -0.3 %; the pieces between the MFMAs: see the header)
                if (half == 1) issue_a();
                else if (STEADY || !pre_issued) issue_w();
                if (half == 0) pre_issued = false;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(afr[0]), "+v"(afr[1]), "+v"(afr[2]), "+v"(afr[3]), "+v"(wfr[0]), "+v"(wfr[1]),
                             "+v"(wfr[2]), "+v"(wfr[3]));
                asm volatile("" : "+v"(afr2[0]), "+v"(afr2[1]), "+v"(afr2[2]), "+v"(afr2[3]));
                if (half == 1) {
                    if constexpr (STEADY) { if (is_g1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                    else wait_next_slab(wm_l1);
                }
                KSTAMP4(half * 6 + 0);
                __builtin_amdgcn_s_barrier();
                KSTAMP4(half * 6 + 1);
                // ---- C segment: 32 MFMAs. (The two waves a tall tile leaves without a region run them too, on whatever their
                // fragment reads returned, and drop the result: a run-time `if (!idle)` here -- one scalar branch in front of every
                // MFMA block -- cost EVERY GEMM 2-6 % (profiles/r6_gemm_idle_branch_ab.txt), and the same choice made once per
                // tile, around two copies of the K loop, made hipcc spill ~100 registers.) ----
                __builtin_amdgcn_s_setprio(1);
                {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[0][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, wfr[ni]), __builtin_bit_cast(bf16x8, afr[mi]), acc[0][mi][ni], 0, 0, 0);
#ifdef COGS_GEMM_KSTAMPS
                __builtin_amdgcn_sched_barrier(0);
                KSTAMP4(half * 6 + 2);
                __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[1][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, wfr[ni]), __builtin_bit_cast(bf16x8, afr2[mi]), acc[1][mi][ni], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                KSTAMP4(half * 6 + 3);
                if (half == 1) {
                    if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else wait_next_slab(wm_c1);
                }
                KSTAMP4(half * 6 + 4);
                if (!(LAST && half == 1)) __builtin_amdgcn_s_barrier();
                KSTAMP4(half * 6 + 5);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(0);
                asm volatile("" ::: "memory");
            }
            ra = ra + 2 >= RING4 ? ra + 2 - RING4 : ra + 2;
            rw = rw + 2 >= RING4 ? rw + 2 - RING4 : rw + 2;
        };
        // (four copies of the slab in a row -- first, steady, the stream's tail, last of the tile -- rather than one loop
        // with branches around them: a diamond made the register allocator spill ~190 registers; KS >= 2 by the launcher)
        int ks = 0;
        slab(std::false_type{}, std::false_type{}, ks);
        ++ks; ++g;
        for (; ks < KS - 1 && g < total - 2; ++ks, ++g) slab(std::true_type{}, std::false_type{}, ks);
        for (; ks < KS - 1; ++ks, ++g) slab(std::false_type{}, std::false_type{}, ks);
        slab(std::false_type{}, std::true_type{}, ks);
        ++ks; ++g;
#ifdef COGS_GEMM_KSTAMPS
        if (p.trace && blockIdx.x == 0 && wc == 0 && lane == 0 && t == (int)blockIdx.x) {   // first tile of workgroup 0
            unsigned long long* o = p.trace + 192 + grp * 12;
            for (int i = 0; i < 12; ++i) o[i] = ks_sum[i];
        }
#endif
        // Tile end. The barrier that closes the last slab stands BEFORE group 0's epilogue and BEHIND group 1's, so the
        // two epilogues share one interval (group 0: epilogue + first load segment of the next tile; group 1: its last
        // MFMA segment + epilogue) instead of taking one each while the other group waits at a barrier.
        // W of the next tile's second slab goes out in front of the epilogue, into the position of this tile's last A
        // unit: a wave stages rows 32w..32w+31 of it, rows that only its own group read (before this point).
#ifdef COGS_GEMM_TSTAMPS
        if (ts_on && ts_i < 24) { const unsigned long long tm_ = __builtin_amdgcn_s_memtime(); if (lane == 0) { p.trace[grp * 96 + 4 * ts_i + 1] = tm_; p.trace[grp * 96 + 4 * ts_i + 2] = tm_; } }
#endif
        if (grp == 0 || p.epi_serial) __builtin_amdgcn_s_barrier();
        issue_w();
        pre_issued = true;
        int ops = -2;      // -2: this wave had no region (tall tile) and issued nothing
        if (!idle) ops = epilogue_wave_pair<T, EPI>(p.epi, m0, n0, p.M, p.N, lane, acc[0], acc[1]);
#ifdef COGS_GEMM_TSTAMPS
        if (ts_on && ts_i < 24) { const unsigned long long tm_ = __builtin_amdgcn_s_memtime(); if (lane == 0) p.trace[grp * 96 + 4 * ts_i + 3] = tm_; }
#endif
        if (grp == 1 && !p.epi_serial) __builtin_amdgcn_s_barrier();
#ifdef COGS_EPI_CONSERVATIVE
        epi_ops = ops == -2 ? -2 : 0;
#else
        epi_ops = __builtin_amdgcn_readfirstlane(ops > 0 || ops == -2 ? ops : 0);
#endif
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance group 1's extra barrier
}

}  // namespace

// kernel launches issued by cogs_k_gemm on this thread (a round-aligned split is two): lets the profiler bracket
// report per-KERNEL averages that can be compared with rocprofv3's kernel stats
static thread_local long g_gemm_launches = 0;
// how many streams the caller is feeding with GEMMs of this size at the same time (cogs_vit_encode's two-stream mode: 2).
// Only the few-tile choice below reads it: with a second stream filling the CUs a launch leaves idle, a partly filled
// round costs its share of the chip, not a whole round.
static thread_local int g_co_streams = 1;
void cogs_k_gemm_co_streams(int streams) { g_co_streams = streams < 1 ? 1 : streams; }
long cogs_k_gemm_launch_count() { return g_gemm_launches; }

int cogs_k_gemv(hipStream_t st, const CogsGemm& g);

namespace {

template <typename T, int EPI>
void launch_small(hipStream_t st, const GemmArgs& p, int grid) {
    const size_t lds = 4 * TILE_BYTES;
    static std::atomic<uint64_t> attr_done{0};
    cogs_ensure_dyn_lds((const void*)gemm_tn_kernel<T, EPI>, (int)lds, attr_done);
    ++g_gemm_launches;
    g_cogs_debug.gemm_last_body = 1;
    hipLaunchKernelGGL((gemm_tn_kernel<T, EPI>), dim3(grid), dim3(256), lds, st, p);
}
template <typename T, int EPI>
void launch_big(hipStream_t st, const GemmArgs& p, int grid) {
    const size_t lds = 3 * SLAB2;
    static std::atomic<uint64_t> attr_done{0};
    cogs_ensure_dyn_lds((const void*)gemm_tn_256x128_kernel<T, EPI>, (int)lds, attr_done);
    const int dbg_wgs = (int)g_cogs_debug.gemm_wgs;
    const int wgs = dbg_wgs <= 0 ? grid : (grid < dbg_wgs ? grid : dbg_wgs);   // 0 = one tile per workgroup
    ++g_gemm_launches;
    g_cogs_debug.gemm_last_body = 2;
    hipLaunchKernelGGL((gemm_tn_256x128_kernel<T, EPI>), dim3(wgs), dim3(512), lds, st, p);
}
// debug switch gemm_pp64 = 0 keeps the 32-wide K-tile kernel (A/B runs); default: the whole-line kernel wherever it exists
// (every epilogue but the rotary LUT one, for which it has no LDS left -- cogs_k_gemm then takes the rotary factors from
// global memory instead: measured 1 % faster end to end than the LUT epilogue on the old body)
bool pp64_enabled() { return g_cogs_debug.gemm_pp64 != 0; }
// Persistent workgroups of the whole-line kernel for `tiles` tiles. 256 = one per CU; debug switch gemm_even_grid = 1: the
// fewest workgroups (a multiple of 8: the tile walk is XCD-aware) that still finish in ceil(tiles / 256) rounds, so that every
// workgroup walks the same number of tiles and the CUs the last round would leave idle are free from the start -- for the
// other stream's kernel, or for the clock.
int pp64_grid(int tiles) {
    if (tiles <= PERSISTENT_WGS) return tiles;
    if (g_cogs_debug.gemm_even_grid == 0) return PERSISTENT_WGS;
    const int rounds = (tiles + PERSISTENT_WGS - 1) / PERSISTENT_WGS;
    const int wgs = ((tiles + rounds - 1) / rounds + 7) / 8 * 8;
    return wgs < PERSISTENT_WGS ? wgs : PERSISTENT_WGS;
}
template <int EPI>
void launch_pp(hipStream_t st, const GemmArgs& p, int grid) {
    const size_t lds = RING3 * SLOT3 + ((EPI & EPI_ROPE_LUT) ? 28 * 1024 : 0);   // ring (+ rotary LUT, <= 28 KiB)
    static std::atomic<uint64_t> attr_done{0};
    cogs_ensure_dyn_lds((const void*)gemm_tn_pp_kernel<EPI>, (int)lds, attr_done);
    const bool env_trace = g_cogs_debug.gemm_trace != 0;
    if (env_trace) {
        static unsigned long long* dbuf = nullptr;
        if (!dbuf) (void)hipMalloc(&dbuf, 224 * 8);
        (void)hipMemsetAsync(dbuf, 0, 224 * 8, st);
        GemmArgs q = p;
        q.trace = dbuf;
        const bool env_pp64t = pp64_enabled();
        bool done64 = false;
        if constexpr ((EPI & EPI_ROPE_LUT) == 0) {
            if (env_pp64t && p.K >= 128) {
                static std::atomic<uint64_t> attr_done64t{0};
                cogs_ensure_dyn_lds((const void*)gemm_tn_pp64_kernel<EPI>, RING4 * UNIT4, attr_done64t);
                hipLaunchKernelGGL((gemm_tn_pp64_kernel<EPI>), dim3(grid < PERSISTENT_WGS ? grid : PERSISTENT_WGS), dim3(512), RING4 * UNIT4, st, q);
                done64 = true;
            }
        }
        if (!done64)
        hipLaunchKernelGGL((gemm_tn_pp_kernel<EPI>), dim3(grid < PERSISTENT_WGS ? grid : PERSISTENT_WGS), dim3(512), lds, st, q);
        unsigned long long h[224];
        (void)hipMemcpyAsync(h, dbuf, sizeof(h), hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        for (int g = 0; g < 2; ++g) {
            fprintf(stderr, "[gemm trace] EPI=%d M=%d N=%d K=%d group %d:", EPI, p.M, p.N, p.K, g);
            for (int i = 0; i + 3 < 96 && h[g * 96 + i + 3]; i += 4)
                fprintf(stderr, " k%llu/e%llu+%llu(+%llu to the next tile's top)", h[g * 96 + i + 1] - h[g * 96 + i], h[g * 96 + i + 2] - h[g * 96 + i + 1],
                        h[g * 96 + i + 3] - h[g * 96 + i + 2], (i + 4 < 96 && h[g * 96 + i + 4]) ? h[g * 96 + i + 4] - h[g * 96 + i + 3] : 0ull);
            fprintf(stderr, "\n");
        }
#ifdef COGS_GEMM_KSTAMPS
        for (int g = 0; g < 2; ++g) {
            const unsigned long long* o = h + 192 + g * 8;
            const double kt_n = p.K / 32.0;
            if (env_pp64t && (EPI & EPI_ROPE_LUT) == 0) {
                const unsigned long long* q = h + 192 + g * 12;
                const double sl = kt_n / 2;
                fprintf(stderr, "[gemm kstamps] group %d per 64-wide slab: L0 %.0f wait %.0f C0 %.0f+%.0f vm %.0f wait %.0f | L1 %.0f wait %.0f C1 %.0f+%.0f vm %.0f wait %.0f\n",
                        g, q[0] / sl, q[1] / sl, q[2] / sl, q[3] / sl, q[4] / sl, q[5] / sl, q[6] / sl, q[7] / sl, q[8] / sl, q[9] / sl, q[10] / sl, q[11] / sl);
            }
            else
            fprintf(stderr, "[gemm kstamps] group %d per K-tile: L %.0f wait %.0f C %.0f wait %.0f\n",
                    g, o[0] / kt_n, o[1] / kt_n, o[2] / kt_n, o[3] / kt_n);
        }
#endif
        return;
    }
    ++g_gemm_launches;
    const bool env_pp64 = pp64_enabled();
    if constexpr ((EPI & EPI_ROPE_LUT) == 0) {
        if (env_pp64 && p.K >= 128) {
            static std::atomic<uint64_t> attr_done64{0};
            cogs_ensure_dyn_lds((const void*)gemm_tn_pp64_kernel<EPI>, RING4 * UNIT4, attr_done64);
            g_cogs_debug.gemm_last_body = 4;
            hipLaunchKernelGGL((gemm_tn_pp64_kernel<EPI>), dim3(pp64_grid(grid)), dim3(512), RING4 * UNIT4, st, p);
            return;
        }
    }
    g_cogs_debug.gemm_last_body = 3;
    hipLaunchKernelGGL((gemm_tn_pp_kernel<EPI>), dim3(grid < PERSISTENT_WGS ? grid : PERSISTENT_WGS), dim3(512), lds, st, p);
}
void dispatch_pp(hipStream_t st, const GemmArgs& p, int grid, int mask) {
#define COGS_PP_CASE(E) case E: launch_pp<E>(st, p, grid); break;
    switch (mask) {
        COGS_PP_CASE(0)
        COGS_PP_CASE(EPI_BIAS)
        COGS_PP_CASE(EPI_RES)
        COGS_PP_CASE(EPI_BIAS | EPI_RES)
        COGS_PP_CASE(EPI_BIAS | EPI_ROPE)
        COGS_PP_CASE(EPI_BIAS | EPI_ROPE | EPI_ROPE_LUT)
        COGS_PP_CASE(EPI_BIAS | EPI_GELU_TANH)
        COGS_PP_CASE(EPI_BIAS | EPI_GELU_ERF)
        COGS_PP_CASE(EPI_BIAS | EPI_ROWSTAT)
        COGS_PP_CASE(EPI_BIAS | EPI_RES | EPI_ROWSTAT)
        COGS_PP_CASE(EPI_BIAS | EPI_LNFOLD)
        COGS_PP_CASE(EPI_BIAS | EPI_ROPE | EPI_LNFOLD)
        COGS_PP_CASE(EPI_BIAS | EPI_ROPE | EPI_ROPE_LUT | EPI_LNFOLD)
        COGS_PP_CASE(EPI_BIAS | EPI_GELU_TANH | EPI_LNFOLD)
        COGS_PP_CASE(EPI_BIAS | EPI_ROPE | EPI_HM)
        COGS_PP_CASE(EPI_BIAS | EPI_ROPE | EPI_LNFOLD | EPI_HM)
        COGS_PP_CASE(EPI_SWIGLU)
        COGS_PP_CASE(EPI_F32OUT)
        COGS_PP_CASE(EPI_NOSTORE)
        default: launch_pp<EPI_GENERIC>(st, p, grid); break;
    }
#undef COGS_PP_CASE
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
        COGS_EPI_CASE(EPI_BIAS | EPI_ROWSTAT)
        COGS_EPI_CASE(EPI_BIAS | EPI_RES | EPI_ROWSTAT)
        COGS_EPI_CASE(EPI_BIAS | EPI_LNFOLD)
        COGS_EPI_CASE(EPI_BIAS | EPI_ROPE | EPI_LNFOLD)
        COGS_EPI_CASE(EPI_BIAS | EPI_GELU_TANH | EPI_LNFOLD)
        COGS_EPI_CASE(EPI_BIAS | EPI_ROPE | EPI_HM)
        COGS_EPI_CASE(EPI_BIAS | EPI_ROPE | EPI_LNFOLD | EPI_HM)
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
    // the fused-LayerNorm features exist in the specialised MFMA epilogues only: the single-row GEMV and the run-time
    // (EPI_GENERIC) epilogue know neither, and silently dropping them would return un-normalised rows
    if ((g.row_stats || g.ln_ab) && (g.M == 1 || cogs_epi_mask(g) == EPI_GENERIC)) return COGS_E_UNSUPPORTED;
    if (g.hm_rows > 0 && (g.M == 1 || g.dtype != COGS_DT_BF16)) return COGS_E_UNSUPPORTED;       // head-major output: bf16 MFMA kernels only
    if (g.M == 1) { ++g_gemm_launches; g_cogs_debug.gemm_last_body = 6; return cogs_k_gemv(st, g); }
    if (g.rms_gamma) return COGS_E_UNSUPPORTED;   // fused RMSNorm exists for the single-token GEMV only
    GemmArgs p;
    p.trace = nullptr; p.rope_lut = nullptr; p.rope_lut_bytes = 0; p.group_m = GROUP_M;
    p.epi_serial = (int)g_cogs_debug.gemm_epi_serial;
    const int rc = cogs_fill_epi(g, &p.epi);
    if (rc != COGS_OK) return rc;
    p.A = (const char*)g.A; p.lda = g.lda * es;
    p.W = (const char*)g.W; p.ldw = g.ldw * es;
    p.M = g.M; p.N = g.N; p.K = g.K;
    const bool env_small = g_cogs_debug.gemm_small != 0;
    const bool env_nopp = g_cogs_debug.gemm_pingpong == 0;
    const int n_pad = (g.N + BN3 - 1) / BN3 * BN3;
    const int env_waste = (int)g_cogs_debug.gemm_pad_pct;
    const bool pp_fits = n_pad * 100 <= g.N * env_waste;   // default: <= 12 % of the MFMAs spent on N padding (N = 1152 -> 1280 measured faster than the 256x128 ring kernel)
    if (!env_nopp && !env_small && g.dtype == COGS_DT_BF16 && g.M >= 1024 && pp_fits && !g.force_small_tile &&
        !g.force_mid_tile) {
        p.nbm = (g.M + BM3 - 1) / BM3;
        p.nbn = (g.N + BN3 - 1) / BN3;
        const bool env_nostore = g_cogs_debug.gemm_nostore != 0;
        const bool env_nolut = g_cogs_debug.gemm_rope_lut == 0;
        // Tall tiles for a ragged column block of <= 128 columns (whole-line kernel only; see TALL4): p.nbn then counts the
        // whole column blocks, and a group of the walk must be a whole number of 384-row tiles.
        const int ragged_cols = g.N % BN3;
        const bool tall = g_cogs_debug.gemm_tall != 0 && pp64_enabled() && g.K >= 128 && g.N > BN3 && ragged_cols > 0 &&
                          ragged_cols <= 128;
        if (tall) p.nbn = g.N / BN3;
        // tile walk: groups of group_m row blocks x all column blocks. Measured (in-run A/B, cfg2 shapes): few column
        // blocks with a long K (fc2: 5 x K 4352) want small groups (2: 0.627 -> 0.591 ms), many column blocks with a
        // short K (fc1: 17 x K 1152) want 8 (2: +7 %)
        const int env_gm = (int)g_cogs_debug.gemm_group_m;
        const bool few_cols_long_k = (g.N + BN3 - 1) / BN3 <= 6 && g.K >= 2048;
        p.group_m = env_gm > 0 ? env_gm : (few_cols_long_k ? 2 : GROUP_M);
        if (tall) p.group_m = env_gm > 0 ? (env_gm + 2) / 3 * 3 : (few_cols_long_k ? 3 : 6);
        p.cf = p.group_m * p.nbn + (tall ? p.group_m * BM3 / TALL4 : 0);
        // tiles of the first `rows` rows (a whole number of row blocks, or all of M)
        auto count_tiles = [&](int rows) {
            const int nbm = (rows + BM3 - 1) / BM3;
            const int full = nbm / p.group_m, last = nbm % p.group_m;
            int n = full * p.cf + last * p.nbn;
            if (tall && last) n += (rows - full * p.group_m * BM3 + TALL4 - 1) / TALL4;
            return n;
        };
        // row blocks (with tall tiles: a multiple of three = whole tall tiles) whose tiles number at most `tiles`
        auto whole_blocks = [&](int tiles) {
            return tall ? tiles * 3 / (3 * p.nbn + 2) / 3 * 3 : tiles / p.nbn;
        };
        int pp_mask = cogs_epi_mask(g);
        p.rope_lut = nullptr; p.rope_lut_bytes = 0;
        if ((pp_mask & ~EPI_LNFOLD) == (EPI_BIAS | EPI_ROPE) && g.rope_lut && g.rope_rowpos && !g.rope_sin && !env_nolut && !pp64_enabled()) {
            const int lut_bytes = g.rope_maxpos * (g.head_dim / 4) * 8;
            if (lut_bytes > 0 && lut_bytes <= 28 * 1024 && g.rope_maxpos < 65536) {
                pp_mask |= EPI_ROPE_LUT;
                p.rope_lut = g.rope_lut; p.rope_lut_bytes = lut_bytes;
                p.epi.rope_lut_lds = RING3 * SLOT3;
            }
        }
        // Round-aligned split. The persistent kernel walks nbm*nbn tiles with 256 workgroups; when the last round
        // is mostly empty (N = 1152 at cfg2: 1155 tiles = 4.5 rounds, half the CUs idle for a whole tile time) the
        // leading row blocks that make up WHOLE rounds stay here and the remaining rows go to the 256x128 kernel,
        // whose half-size tiles fill the chip again (rows are independent: same arithmetic per row, same K order, so
        // a row's result does not depend on which kernel computed it).
        const bool env_nosplit = g_cogs_debug.gemm_split == 0;
        const int nb = count_tiles(g.M);
        p.ntiles = nb;
        const int rounds = nb / PERSISTENT_WGS, rem = nb % PERSISTENT_WGS;
        int mb_main = -1;      // >= 0: row blocks that stay in this kernel (0 = none: the whole GEMM goes to the ring kernel)
        // (calibrated on K = 1152 .. 4352, the ViT shapes; longer K -- the Qwen2 prompt pass at M = 2048 -- keeps the
        // plain ping-pong launch)
        const int env_co = (int)g_cogs_debug.gemm_co_streams;   // A/B runs
        const int S = env_co > 0 ? env_co : g_co_streams;
        if (!env_nosplit && !env_nostore && nb <= 2 * PERSISTENT_WGS && g.K <= 4352) {
            // Few tiles (one rank's share of a frame-sharded clip: M = 6 400 is 125 tiles for N = 1152 and 350 = 1.37
            // rounds for QKV): the time is rounds x one tile, so the choice is by rounds. Calibrated on the four ViT
            // shapes at M = 3 200 .. 14 784 (tools/cal_tiles.sh, rocprofv3; round 4 with the whole-line kernel,
            // tools/experiments/cring_sweep.sh): a round of 256x128 ring tiles takes 0.72 of a round of ping-pong tiles of
            // the same K; a second launch costs about 0.08 of one. With a second stream running the other half of the clip
            // (S = 2) the CUs a launch leaves idle are not lost, so a launch costs max(1, its share of the chip) and is never
            // split: measured at 3 200 / 3 696 / 6 400 / 14 784 rows per stream, the ring kernel wins the first two (65 and
            // 75 ping-pong tiles: -2 %, -7 % of the step), the ping-pong kernel the third (125 tiles: -12 %) and the unsplit
            // launch the last (-4 %).
            const int rbm = (g.M + BM2 - 1) / BM2, rbn = (g.N + BN - 1) / BN;
            auto launch_cost = [&](int tiles) {
                return S == 1 ? (float)((tiles + PERSISTENT_WGS - 1) / PERSISTENT_WGS)
                              : fmaxf(1.f, (float)S * (float)tiles / (float)PERSISTENT_WGS);
            };
            const float c_ring = (float)g_cogs_debug.gemm_ring_cost_permille * 1e-3f;   // 0.72; the switch is for calibration sweeps
            const float c_launch = 0.08f;
            const float cost_pp = launch_cost(nb);
            const float cost_ring = c_ring * launch_cost(rbm * rbn);
            float best = cost_pp;
            if (cost_ring < best) { best = cost_ring; mb_main = 0; }
            if (S == 1 && rounds == 1 && rem > 0) {
                const int mb = whole_blocks(PERSISTENT_WGS);           // whole row blocks inside the first round
                const int rows_rem = g.M - mb * BM3;
                if (mb > 0 && rows_rem >= 512) {     // the remainder must reach the 256x128 ring kernel the cost model prices
                    const float cost_split = 1.f + c_ring * launch_cost(((rows_rem + BM2 - 1) / BM2) * rbn) + c_launch;
                    if (cost_split < best) { best = cost_split; mb_main = mb; }
                }
            }
        } else if (S == 1 && !env_nosplit && !env_nostore && g.K >= 2048 && rounds >= 2 && rem > 0 && rem * 100 < PERSISTENT_WGS * 45) {
            // (many tiles: pays when a tile is long against a second launch and the last round is less than ~45 % full:
            // measured with the two-segment K loop at K = 1152: -2 %; K = 3584, last round 34 % full: +2 %; K = 18944,
            // 34 %: +6 %; K = 4352, 51 % full (ViT fc2): -2 %, so it no longer splits)
            const int mb = whole_blocks(rounds * PERSISTENT_WGS);    // whole row blocks within the full rounds
            if (mb > 0 && g.M - mb * BM3 >= 512) mb_main = mb;
        }
        const bool env_choice = g_cogs_debug.gemm_choice != 0;      // diagnostics: which body each shape gets
        if (env_choice) fprintf(stderr, "[gemm choice] M=%d N=%d K=%d: %d ping-pong tiles (%d rounds + %d), mb_main %d\n", g.M, g.N, g.K, nb, rounds, rem, mb_main);
        if (mb_main >= 0) {
            const int rows_main = mb_main * BM3;
            CogsGemm b = g;
            b.M = g.M - rows_main; b.force_mid_tile = 1;
            b.A = (const char*)g.A + (size_t)rows_main * g.lda * es;
            b.C = (char*)g.C + (size_t)rows_main * (g.hm_rows > 0 ? g.head_dim : g.ldc) * (g.out_f32 ? 4 : es);   // head-major: rows advance by head_dim inside every head block
            if (g.residual) b.residual = (const char*)g.residual + (size_t)rows_main * g.ldr * es;
            if (g.rope_rowpos) b.rope_rowpos = g.rope_rowpos + rows_main;
            if (g.row_stats) b.row_stats = g.row_stats + (size_t)rows_main * (g.N / 64) * 2;
            if (g.ln_ab) b.ln_ab = g.ln_ab + (size_t)rows_main * 2;
            if (g.rope_cos) {
                const size_t per_row = (size_t)(g.head_dim / 2) * (g.rope_sin ? 1 : 2);
                b.rope_cos = g.rope_cos + (size_t)rows_main * per_row;
                if (g.rope_sin) b.rope_sin = g.rope_sin + (size_t)rows_main * per_row;
            }
            if (mb_main > 0) {
                p.M = rows_main; p.nbm = mb_main;
                p.ntiles = count_tiles(rows_main);
                dispatch_pp(st, p, p.ntiles, pp_mask);
                const int rc2 = COGS_LAUNCH_CHECK();
                if (rc2 != COGS_OK) return rc2;
            }
            const int rc3 = cogs_k_gemm(st, b);
            if (mb_main > 0) g_cogs_debug.gemm_last_body = 5;
            return rc3;
        }
        dispatch_pp(st, p, nb, env_nostore ? EPI_NOSTORE : pp_mask);
        return COGS_LAUNCH_CHECK();
    }
    const bool big = g.M >= 512 && !g.force_small_tile && !env_small;
    p.nbm = (g.M + (big ? BM2 : BM) - 1) / (big ? BM2 : BM);
    p.nbn = (g.N + BN - 1) / BN;
    const int grid = p.nbm * p.nbn;
    const int mask = cogs_epi_mask(g);
    if (g.dtype == COGS_DT_BF16) dispatch<bf16_t>(st, p, grid, mask, big);
    else dispatch<float>(st, p, grid, mask, big);
    return COGS_LAUNCH_CHECK();
}
