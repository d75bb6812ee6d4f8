// Is a weight matrix that a previous kernel has touched cheaper to stream than one that comes from HBM? (decode: the next
// GEMV's weights do not depend on the previous kernel's result, so a side stream could read them ahead into the Infinity Cache.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/mall_prefetch.cpp -o tools/micro/mall_prefetch
// For sizes 26 / 33 / 136 / 272 MB: the streaming kernel (non-temporal 16-byte loads, 64 KiB per workgroup, like the GEMV) timed
// cold (after 2 GB of other traffic) and after a prefetch pass of the same buffer with plain loads / with non-temporal loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float vf4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void stream(const vf4* __restrict__ w, float* out) {
    const vf4* p = w + (size_t)blockIdx.x * 4096 + threadIdx.x;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const vf4 v = NT ? __builtin_nontemporal_load(p + i * 256) : p[i * 256];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}
int main() {
    const size_t big = 2048ull << 20;
    char *flush, *buf; float* out;
    (void)hipMalloc(&flush, big); (void)hipMalloc(&buf, 512ull << 20); (void)hipMalloc(&out, 1 << 20);
    (void)hipMemset(flush, 1, big); (void)hipMemset(buf, 1, 512ull << 20);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (double mb : {25.7, 33.0, 135.8, 271.6}) {
        const int wgs = (int)(mb * 1e6 / 65536);
        for (int mode = 0; mode < 4; ++mode) {       // 0 cold, 1 after plain prefetch, 2 after non-temporal prefetch, 3 back to back (same kernel twice)
            std::vector<float> ts;
            for (int rep = 0; rep < 7; ++rep) {
                hipLaunchKernelGGL(stream<true>, dim3((int)(big / 65536)), dim3(256), 0, 0, (const vf4*)flush, out);
                if (mode == 1) hipLaunchKernelGGL(stream<false>, dim3(wgs), dim3(256), 0, 0, (const vf4*)buf, out);
                if (mode == 2 || mode == 3) hipLaunchKernelGGL(stream<true>, dim3(wgs), dim3(256), 0, 0, (const vf4*)buf, out);
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL(stream<true>, dim3(wgs), dim3(256), 0, 0, (const vf4*)buf, out);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            const char* names[] = {"cold (after 2 GB of other reads)", "after a prefetch pass, plain loads", "after a prefetch pass, non-temporal loads", "second of two identical passes"};
            printf("%6.1f MB  %-44s median %7.2f us  %6.2f TB/s\n", mb, names[mode], ts[3] * 1e3, mb * 1e6 / (ts[3] * 1e-3) / 1e12);
        }
    }
    return 0;
}
