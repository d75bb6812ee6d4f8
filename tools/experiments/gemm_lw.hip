// Loader-wave MFMA GEMM (bf16): C[M,N] = epilogue( A[M,K] . W[N,K]^T ), the ViT encoder's dense contractions
// (model/modeling_videollama3_encoder.py:194-210 patch embed, :246-248 q/k/v, :275 out_proj, :369-373 fc1 / fc2).
//
// Why another body. In the ping-pong kernel (gemm.hip) every wave stages its share of the next K-tile itself: 4 LDS-DMA
// pieces per wave and K-tile, and each 1 KiB piece holds the issuing wave for ~115 cycles (the CU's vector-memory path
// takes ~34 B/clk whoever asks) -- its load segment (650-770 cycles) is longer than its partner's 32 MFMAs (512), and
// with all staging removed the same MFMA stream runs 1.7x faster. Here the staging is
// taken out of the computing waves altogether:
//
//   * 768-thread workgroup = 8 COMPUTE waves + 4 LOADER waves, three waves per SIMD (two compute + one loader).
//   * tile 256 x 192 (N = 1152 = 6 tiles, 3456 = 18: no padded MFMAs on the ViT shapes), 32-wide K-tiles in the
//     64-byte-row swizzled LDS image of the ping-pong kernel, 5-slot ring, the loaders run 4 K-tiles ahead and issue
//     all 28 one-KiB LDS-DMA pieces of a K-tile (7 per loader wave), continuously across tiles (persistent grid).
//   * a compute wave owns 64 rows x 96 columns: per K-tile 10 ds_read_b128 and 24 v_mfma_f32_16x16x32_bf16; 96
//     accumulator + 40 fragment registers, which fits the 168 registers a wave gets at three waves per SIMD.
//   * compute waves never touch vmcnt inside the K loop; the two compute waves of a SIMD alternate read / MFMA
//     segments exactly like the ping-pong kernel (group 1 one barrier behind group 0), the loaders keep step with the
//     same barriers: K-tile j + 4 is issued in the interval after the last read of K-tile j - 1 (same slot), and a
//     counted s_waitcnt vmcnt in front of the loaders' second barrier guarantees K-tile j + 1 has landed.
//   * every output element accumulates its K products in the same order as in the other GEMM kernels (one
//     16x16x32 MFMA per 32-wide K-tile, ascending), so a row's result does not depend on the kernel that computed it.
#include "common.h"
#include "kernels.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

namespace {

constexpr int LBM = 256, LBN = 192;
constexpr int LROW = 64;                          // bytes per LDS row = 32 bf16
constexpr int LSLOT = (LBM + LBN) * LROW;         // 28 KiB per K-tile
constexpr int LPIECES = LSLOT / 1024;             // 28 pieces of 16 rows x 64 B
constexpr int LRING = 5;
constexpr int LDIST = LRING - 1;                  // loaders run 4 K-tiles ahead
constexpr int LPW = LPIECES / 4;                  // 7 pieces per loader wave and K-tile
constexpr int LW_WGS = 256;

struct LwArgs {
    const char* A; long lda;   // bytes per row
    const char* W; long ldw;
    char* C; long ldc;         // elements per row
    int M, N, K;
    int nbm, nbn, group_m;
    int nostore;
};

__device__ __forceinline__ void lw_glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ int lw_swz(int r) { return (0x78 >> (2 * ((r >> 2) & 3))) & 3; }   // [0,2,3,1]

template <int N>
__device__ __forceinline__ void lw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(768, 3) void gemm_lw_proto_kernel(LwArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = p.nbm * p.nbn;
    const int KT = p.K / 32;

    auto tile_origin = [&](int t, int& m0, int& n0) {
        const int xcd = t & 7, q = nb >> 3, r = nb & 7;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
        const int per_group = p.group_m * p.nbn;
        const int first_m = (bid / per_group) * p.group_m;
        const int gsz = min(p.nbm - first_m, p.group_m);
        m0 = (first_m + (bid % per_group) % gsz) * LBM;
        n0 = ((bid % per_group) / gsz) * LBN;
    };
    if ((int)blockIdx.x >= nb) return;
    const int my_tiles = (nb - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
    const int total = my_tiles * KT;          // K-tiles this workgroup consumes

    if (wid >= 8) {
        // ------------------------------------------------------------------ loader wave
        const int lw = wid - 8;
        const char* src[LPW];
        int st_t = blockIdx.x, st_kt = 0, st_slot = 0, issued = 0;
        auto set_src = [&](int t) {
            int m0, n0;
            tile_origin(t, m0, n0);
#pragma unroll
            for (int i = 0; i < LPW; ++i) {
                const int pc = lw * LPW + i;               // 0..15 = A rows, 16..27 = W rows
                const bool is_a = pc < 16;
                const int r = (is_a ? pc : pc - 16) * 16 + (lane >> 2);
                const int c = (lane & 3) ^ lw_swz(lane >> 2);
                src[i] = is_a ? p.A + (long)min(m0 + r, p.M - 1) * p.lda + c * 16
                              : p.W + (long)min(n0 + r, p.N - 1) * p.ldw + c * 16;
            }
        };
        auto issue_ktile = [&]() {
            if (issued >= total) return;
            const long ko = (long)st_kt * LROW;
            char* dst = smem + st_slot * LSLOT + lw * LPW * 1024;
#pragma unroll
            for (int i = 0; i < LPW; ++i) lw_glds16(src[i] + ko, dst + i * 1024);
            ++issued;
            st_slot = st_slot + 1 == LRING ? 0 : st_slot + 1;
            if (++st_kt == KT) {
                st_kt = 0;
                st_t += gridDim.x;
                if (st_t < nb) set_src(st_t);
            }
        };
        // wait until K-tile `k` (0-based index in this workgroup's stream) has landed: at most the pieces of the
        // K-tiles issued after it may be outstanding
        auto wait_landed = [&](int k) {
            const int newer = issued - 1 - k;
            if (newer >= 3) lw_wait_vm<3 * LPW>();
            else if (newer == 2) lw_wait_vm<2 * LPW>();
            else if (newer == 1) lw_wait_vm<LPW>();
            else lw_wait_vm<0>();
        };
        set_src(st_t);
        for (int i = 0; i < LDIST; ++i) issue_ktile();
        wait_landed(0);
        __builtin_amdgcn_s_barrier();                    // P: K-tile 0 visible to everyone
        for (int j = 0; j < total; ++j) {
            issue_ktile();                               // K-tile j + LDIST -> the slot K-tile j - 1 was read from
            __builtin_amdgcn_s_barrier();
            if (j + 1 < total) wait_landed(j + 1);
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();                    // balances group 1's extra barrier
        return;
    }

    // ---------------------------------------------------------------------- compute wave
    const int grp = wid >> 2;          // ping-pong group: the waves wid and wid + 4 share a SIMD
    const int rg = wid & 3;            // 64-row group of the tile
    const int cg = grp;                // 96-column group of the tile
    const int foff = (lane & 15) * LROW + (((lane >> 4) ^ lw_swz(lane & 15)) << 4);
    const int a_off = (rg * 64) * LROW + foff;                  // + mi*16*LROW
    const int w_off = LBM * LROW + (cg * 96) * LROW + foff;     // + ni*16*LROW

    f32x4 acc[4][6];
    u32x4 afr[4], wfr[6];

    __builtin_amdgcn_s_barrier();                        // P
    if (grp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0
    int slot = 0;
    for (int t = blockIdx.x; t < nb; t += gridDim.x) {
        int m0, n0;
        tile_origin(t, m0, n0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT; ++kt) {
            const char* base = smem + slot * LSLOT;
#pragma unroll
            for (int i = 0; i < 6; ++i) wfr[i] = *reinterpret_cast<const u32x4*>(base + w_off + i * 16 * LROW);
#pragma unroll
            for (int i = 0; i < 4; ++i) afr[i] = *reinterpret_cast<const u32x4*>(base + a_off + i * 16 * LROW);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(afr[0]), "+v"(afr[1]), "+v"(afr[2]), "+v"(afr[3]));
            asm volatile("" : "+v"(wfr[0]), "+v"(wfr[1]), "+v"(wfr[2]), "+v"(wfr[3]), "+v"(wfr[4]), "+v"(wfr[5]));
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 6; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, wfr[ni]), __builtin_bit_cast(bf16x8, afr[mi]), acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("" ::: "memory");
            slot = slot + 1 == LRING ? 0 : slot + 1;
        }
        // prototype epilogue: plain bf16 stores (lane: row mb + 16 mi + (lane&15), columns nb + 16 ni + 4 (lane>>4) ..+3)
        const int mrow = m0 + rg * 64 + (lane & 15);
        const int ncol = n0 + cg * 96 + ((lane >> 4) << 2);
        if (p.nostore) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 6; ++ni) asm volatile("" ::"v"(acc[mi][ni]));
        } else {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int m = mrow + mi * 16;
#pragma unroll
                for (int ni = 0; ni < 6; ++ni) {
                    const int n = ncol + ni * 16;
                    if (m < p.M && n < p.N)
                        st4_f<bf16_t>(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n, acc[mi][ni]);
                }
            }
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();          // balances group 1's extra barrier
}

}  // namespace

// experimental entry (tools/gemm_lw_ab.py): plain bf16 GEMM through the loader-wave body
extern "C" int cogs_x_gemm_lw(void* stream, const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N,
                              int K, int group_m, int nostore) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0 || N % 4 != 0) return COGS_E_INVALID;
    LwArgs p;
    p.A = (const char*)A; p.lda = lda * 2;
    p.W = (const char*)W; p.ldw = ldw * 2;
    p.C = (char*)C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.nbm = (M + LBM - 1) / LBM; p.nbn = (N + LBN - 1) / LBN;
    p.group_m = group_m > 0 ? group_m : 8;
    p.nostore = nostore;
    const size_t lds = (size_t)LRING * LSLOT;
    static std::atomic<uint64_t> attr_done{0};
    cogs_ensure_dyn_lds((const void*)gemm_lw_proto_kernel, (int)lds, attr_done);
    const int nb = p.nbm * p.nbn;
    hipLaunchKernelGGL(gemm_lw_proto_kernel, dim3(nb < LW_WGS ? nb : LW_WGS), dim3(768), lds, (hipStream_t)stream, p);
    return COGS_LAUNCH_CHECK();
}
