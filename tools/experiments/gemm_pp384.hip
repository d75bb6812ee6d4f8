// Probe: the ping-pong body of csrc/gemm.hip (gemm_tn_pp_kernel) with a 256 x 384 tile -- a wave owns 128 x 96
// (192 accumulator + 56 fragment registers), 40 KiB per 32-wide K-tile, 4-slot ring = all 160 KiB of LDS. Per MAC it moves
// 0.83 of the 256 x 256 tile's bytes through the CU's vector-memory path (the resource the shipped kernel shares with the
// MFMA pipe at about equal load), and N = 1152 / 3456 are whole multiples of 384. Plain bf16 GEMM, simple stores. NEVER RUN:
// hipcc allocates 256 VGPRs and spills 40 registers inside the MFMA segment (-Rpass-analysis=kernel-resource-usage; scratch
// reloads count on vmcnt, which the staging waits rely on), so the probe stops at the compile.
#include "common.h"
#include <stdlib.h>
#include <atomic>

namespace {
constexpr int PBM = 256, PBN = 384;
constexpr int PROW = 64;
constexpr int PSLOT = (PBM + PBN) * PROW;      // 40 KiB
constexpr int PRING = 4, PDIST = PRING - 1;
constexpr int PPW = PSLOT / 1024 / 8;          // 5 pieces per wave and K-tile

struct PArgs {
    const char* A; long lda; const char* W; long ldw; char* C; long ldc;
    int M, N, K, nbm, nbn, group_m, nostore;
};
__device__ __forceinline__ void p_glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ int p_swz(int r) { return (0x78 >> (2 * ((r >> 2) & 3))) & 3; }

__global__ __launch_bounds__(512, 2) void gemm_pp384_kernel(PArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, wc = wid & 3;
    const int nb = p.nbm * p.nbn;
    const int KT = p.K / 32;
    auto tile_origin = [&](int t, int& m0, int& n0) {
        const int xcd = t & 7, q = nb >> 3, r = nb & 7;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
        const int per_group = p.group_m * p.nbn;
        const int first_m = (bid / per_group) * p.group_m;
        const int gsz = min(p.nbm - first_m, p.group_m);
        m0 = (first_m + (bid % per_group) % gsz) * PBM;
        n0 = ((bid % per_group) / gsz) * PBN;
    };
    const char* src[PPW];
    int st_t = blockIdx.x, st_kt = 0, st_slot = 0;
    auto set_src = [&](int t) {
        int m0, n0;
        tile_origin(t, m0, n0);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pc = wid * PPW + i;                 // 0..15 A rows, 16..39 W rows
            const bool is_a = pc < 16;
            const int r = (is_a ? pc : pc - 16) * 16 + (lane >> 2);
            const int c = (lane & 3) ^ p_swz(lane >> 2);
            src[i] = is_a ? p.A + (long)min(m0 + r, p.M - 1) * p.lda + c * 16
                          : p.W + (long)min(n0 + r, p.N - 1) * p.ldw + c * 16;
        }
    };
    char* const lds_wave = smem + wid * PPW * 1024;
    auto stage = [&]() {
        if (st_t >= nb) return;
        const long ko = (long)st_kt * PROW;
        char* dst = lds_wave + st_slot * PSLOT;
#pragma unroll
        for (int i = 0; i < PPW; ++i) p_glds16(src[i] + ko, dst + i * 1024);
        st_slot = st_slot + 1 == PRING ? 0 : st_slot + 1;
        if (++st_kt == KT) {
            st_kt = 0;
            st_t += gridDim.x;
            if (st_t < nb) set_src(st_t);
        }
    };
    const int foff = (lane & 15) * PROW + (((lane >> 4) ^ p_swz(lane & 15)) << 4);
    const int a_off = (grp * 128) * PROW + foff;
    const int w_off = PBM * PROW + (wc * 96) * PROW + foff;
    f32x4 acc[8][6];
    u32x4 afr[8], wfr[6];
    if (st_t >= nb) return;
    set_src(st_t);
    const int my_tiles = (nb - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
    const int total = my_tiles * KT;
    {
        const int pre = total < PDIST ? total : PDIST;
        for (int i = 0; i < pre; ++i) stage();
        if (pre >= 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (pre == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    auto wait_next = [&](int ahead) {
        const int newer = (ahead < PDIST ? ahead : PDIST) - 1;
        if (newer >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    int slot = 0, g = 0;
    for (int t = blockIdx.x; t < nb; t += gridDim.x) {
        int m0, n0;
        tile_origin(t, m0, n0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT; ++kt, ++g) {
            const char* base = smem + slot * PSLOT;
#pragma unroll
            for (int i = 0; i < 6; ++i) wfr[i] = *reinterpret_cast<const u32x4*>(base + w_off + i * 16 * PROW);
#pragma unroll
            for (int i = 0; i < 8; ++i) afr[i] = *reinterpret_cast<const u32x4*>(base + a_off + i * 16 * PROW);
            stage();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(afr[0]), "+v"(afr[1]), "+v"(afr[2]), "+v"(afr[3]), "+v"(afr[4]), "+v"(afr[5]), "+v"(afr[6]), "+v"(afr[7]));
            asm volatile("" : "+v"(wfr[0]), "+v"(wfr[1]), "+v"(wfr[2]), "+v"(wfr[3]), "+v"(wfr[4]), "+v"(wfr[5]));
            const int ahead = total - 1 - g;
            if (grp == 1) wait_next(ahead);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 6; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, wfr[ni]), __builtin_bit_cast(bf16x8, afr[mi]), acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (grp == 0) wait_next(ahead);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("" ::: "memory");
            slot = slot + 1 == PRING ? 0 : slot + 1;
        }
        const int mrow = m0 + grp * 128 + (lane & 15);
        const int ncol = n0 + wc * 96 + ((lane >> 4) << 2);
        if (p.nostore) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 6; ++ni) asm volatile("" ::"v"(acc[mi][ni]));
        } else {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int m = mrow + mi * 16;
#pragma unroll
                for (int ni = 0; ni < 6; ++ni) {
                    const int n = ncol + ni * 16;
                    if (m < p.M && n < p.N)
                        st4_f<bf16_t>(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n, acc[mi][ni]);
                }
            }
        }
        // the stores above are newer than the pieces the next waits are about: waited for too (conservative probe)
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
}
}  // namespace

extern "C" int cogs_x_gemm_pp384(void* stream, const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N,
                                 int K, int group_m, int nostore) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0 || N % 4 != 0) return -1;
    PArgs p;
    p.A = (const char*)A; p.lda = lda * 2; p.W = (const char*)W; p.ldw = ldw * 2; p.C = (char*)C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.nbm = (M + PBM - 1) / PBM; p.nbn = (N + PBN - 1) / PBN;
    p.group_m = group_m > 0 ? group_m : 8; p.nostore = nostore;
    const size_t lds = (size_t)PRING * PSLOT;
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)gemm_pp384_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
    const int nb = p.nbm * p.nbn;
    hipLaunchKernelGGL(gemm_pp384_kernel, dim3(nb < 256 ? nb : 256), dim3(512), lds, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
