// How fast does a CU take in LDS-DMA pieces (global_load_lds_dwordx4, 1 KiB per wave instruction) from L2-resident rows,
// as a function of how a piece's 64 lanes are spread over rows? The GEMM K loop stages 16 rows x 64 B per piece (32-wide
// bf16 K-tiles); the alternatives take whole 128-byte lines (8 rows x 128 B), 4 rows x 256 B, or 1 KiB of one row.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/ldsdma_rate.cpp -o tools/micro/ldsdma_rate
// 256 workgroups x 8 waves, every wave keeps 8 pieces in flight; all workgroups walk the same 512-row panel (L2 hits).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int RB>   // bytes of one row a piece takes: 64, 128, 256, 1024
__global__ __launch_bounds__(512) void rate_kernel(const char* base, long ld, int kbytes, int iters, int wgs_per_panel) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int LPR = RB / 16;            // lanes per row
    constexpr int ROWS = 64 / LPR;          // rows per piece
    // a "slab" = 512 rows x RB bytes = 512*RB/1024 pieces, split over the 8 waves
    constexpr int PPW = 512 * RB / 1024 / 8;   // pieces per wave and slab: 4, 8, 16, 64
    const char* src0 = base + (long)(blockIdx.x / wgs_per_panel) * 512 * ld + (long)(lane / LPR) * ld + (lane % LPR) * 16;
    char* dst = smem + wid * 8192;
    int ko = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wid * PPW + i;
            glds16(src0 + (long)piece * ROWS * ld + ko, dst + (i & 7) * 1024);
            if ((i & 3) == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        ko += RB;
        if (ko + RB > kbytes) ko = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int RB>
void run(const char* name, const char* buf, long ld, int kbytes, int wgs_per_panel) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int slabs = 4096 * 64 / RB;                 // same bytes for every pattern: 4096 K-tiles of 32 KiB = 128 MiB per WG
    hipFuncSetAttribute((const void*)rate_kernel<RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(rate_kernel<RB>, dim3(256), dim3(512), 96 * 1024, 0, buf, ld, kbytes, slabs, wgs_per_panel);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = 4096.0 * 32768;
    printf("%-26s ld %5ld B, %3d WGs per panel: %7.3f ms  %6.1f GB/s per CU  (%5.1f B/clk at 2.4 GHz), chip %5.2f TB/s\n", name, ld, wgs_per_panel, best,
           bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 2.4e9, bytes * 256 / (best * 1e-3) / 1e12);
}

int main() {
    char* buf; const size_t sz = (size_t)256 * 512 * 8192;   // up to 256 panels of 512 rows x 8 KiB
    hipMalloc(&buf, sz); hipMemset(buf, 0x11, sz);
    for (int wpp : {256, 32, 8}) {
        for (long ld : {2304L, 7168L}) {
            const int kb = (int)ld;
            run<64>("16 rows x 64 B", buf, ld, kb, wpp);
            run<128>("8 rows x 128 B", buf, ld, kb, wpp);
            run<256>("4 rows x 256 B", buf, ld, kb, wpp);
            run<1024>("1 row x 1 KiB", buf, ld, kb, wpp);
        }
    }
    return 0;
}
