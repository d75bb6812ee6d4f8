// The GEMM epilogue's residual loads and C stores are 16 bytes per lane with FOUR lanes per row: one wave instruction
// touches 16 rows x 64 bytes -- sixteen half lines. Would the same bytes as 8 rows x 128 bytes (whole lines, eight lanes
// per row) move faster through the CU's vector-memory path? Each wave owns a 128-row x 128-byte region of a 256 x 256
// bf16 tile (as in gemm_tn_pp64_kernel: 8 waves, region = rows 128 g.., columns 64 wc..), 16 instructions per region and
// kind; 256 workgroups walk the tiles of an M x 1152 matrix (row stride 2 304 bytes) like the persistent GEMM does.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/epi_rw_rate.cpp -o tools/micro/epi_rw_rate && tools/micro/epi_rw_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// MODE bit 0: load, bit 1: store.  WHOLE: 0 = 16 rows x 64 B per instruction, 1 = 8 rows x 128 B
template <int MODE, int WHOLE>
__global__ __launch_bounds__(512) void rw_kernel(const char* R, char* C, long ld, int nbm, int nbn, int spin) {
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wid >> 2, wc = wid & 3;
    const int nb = nbm * nbn;
    for (int t = blockIdx.x; t < nb; t += gridDim.x) {
        const int tm = t % nbm, tn = t / nbm;
        const long base = ((long)tm * 256 + grp * 128) * ld + ((long)tn * 256 + wc * 64) * 2;
        u32x4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            long off;
            if (WHOLE) off = base + (long)(8 * i + (lane >> 3)) * ld + (lane & 7) * 16;                 // 8 rows x 128 B
            else off = base + (long)(16 * (i >> 1) + (lane & 15)) * ld + (i & 1) * 64 + (lane >> 4) * 16;   // 16 rows x 64 B
            if (MODE & 1) v[i] = *reinterpret_cast<const u32x4*>(R + off);
            else v[i] = u32x4{(uint32_t)off, 1u, 2u, 3u};
        }
        // stand-in for the K loop between epilogues (the rate question is about the burst, not the average)
        for (int s = 0; s < spin; ++s) asm volatile("s_sleep 8");
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            long off;
            if (WHOLE) off = base + (long)(8 * i + (lane >> 3)) * ld + (lane & 7) * 16;
            else off = base + (long)(16 * (i >> 1) + (lane & 15)) * ld + (i & 1) * 64 + (lane >> 4) * 16;
            if (MODE & 2) *reinterpret_cast<u32x4*>(C + off) = v[i];
            else if (v[i][1] == 0xdeadbeefu) *reinterpret_cast<u32x4*>(C + off) = v[i];
        }
    }
}

template <int MODE, int WHOLE>
static float run(const char* R, char* C, long ld, int nbm, int nbn, int spin, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rw_kernel<MODE, WHOLE>), dim3(256), dim3(512), 0, 0, R, C, ld, nbm, nbn, spin);
    hipDeviceSynchronize();
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rw_kernel<MODE, WHOLE>), dim3(256), dim3(512), 0, 0, R, C, ld, nbm, nbn, spin);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

#include <algorithm>
int main() {
    const int M = 59136, N = 1152;
    const long ld = (long)N * 2;
    const int nbm = M / 256, nbn = 4;     // the four whole column blocks
    char *R, *C;
    hipMalloc(&R, (size_t)M * ld); hipMalloc(&C, (size_t)M * ld);
    hipMemset(R, 1, (size_t)M * ld); hipMemset(C, 0, (size_t)M * ld);
    const double bytes = (double)nbm * nbn * 256 * 256 * 2;
    for (int spin : {0, 40}) {
        printf("spin %d (s_sleep 8 x spin between the loads and the stores of a tile)\n", spin);
#define RUN(MODE, WHOLE, name) { float t = run<MODE, WHOLE>(R, C, ld, nbm, nbn, spin, 9); \
        printf("  %-34s %.4f ms  %.2f TB/s per direction\n", name, t, bytes / t / 1e9); }
        RUN(1, 0, "loads,  16 rows x 64 B");
        RUN(1, 1, "loads,   8 rows x 128 B");
        RUN(2, 0, "stores, 16 rows x 64 B");
        RUN(2, 1, "stores,  8 rows x 128 B");
        RUN(3, 0, "load + store, 16 rows x 64 B");
        RUN(3, 1, "load + store,  8 rows x 128 B");
    }
    return 0;
}
