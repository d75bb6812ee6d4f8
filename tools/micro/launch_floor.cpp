// launch-boundary price on this box: N dependent trivial kernels on one stream, eager vs hipGraph replay,
// and a chain shaped like one decode layer (6 kernels of 256..4736 workgroups doing a tiny amount of work).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
// a weight-streaming stand-in: every workgroup reads 64 KiB once (non-temporal) and writes one float
typedef float vf4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_k(const float4* __restrict__ w4, float* out) {
    const vf4* p = reinterpret_cast<const vf4*>(w4) + (size_t)blockIdx.x * 4096 + threadIdx.x;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const vf4 v = __builtin_nontemporal_load(p + i * 256); acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* d; hipMalloc(&d, 64 << 20); hipMemset(d, 0, 64 << 20);
    hipStream_t st; hipStreamCreate(&st);
    const int N = 2000;
    for (int grid : {1, 256, 2368, 4736}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipStreamSynchronize(st);
            double t0 = now();
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(touch, dim3(grid), dim3(256), 0, st, d, grid * 256);
            double t_issue = now() - t0;
            hipStreamSynchronize(st);
            double t1 = now() - t0;
            if (rep) printf("eager grid %5d: %.2f us per kernel (host issue %.2f us)\n", grid, t1 / N * 1e6, t_issue / N * 1e6);
        }
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(touch, dim3(grid), dim3(256), 0, st, d, grid * 256);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        double t0 = now();
        for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        printf("graph grid %5d: %.2f us per kernel\n", grid, (now() - t0) / 2000 * 1e6);
    }
    // dependent kernels that each stream 32 MiB (512 workgroups x 64 KiB) out of a 2 GiB buffer: eager vs graph replay
    {
        float4* w; hipMalloc(&w, (size_t)2 << 30); hipMemset(w, 0, (size_t)2 << 30);
        const int per = 512, nk = 60;                 // 60 kernels x 32 MiB = 1.9 GiB: nothing is re-read from cache
        for (int rep = 0; rep < 3; ++rep) {
            hipStreamSynchronize(st);
            double t0 = now();
            for (int i = 0; i < nk; ++i) hipLaunchKernelGGL(stream_k, dim3(per), dim3(256), 0, st, w + (size_t)i * per * 4096, d);
            hipStreamSynchronize(st);
            if (rep) printf("eager 32 MiB stream kernels: %.2f us per kernel\n", (now() - t0) / nk * 1e6);
        }
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < nk; ++i) hipLaunchKernelGGL(stream_k, dim3(per), dim3(256), 0, st, w + (size_t)i * per * 4096, d);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            hipGraphLaunch(ge, st);
            hipStreamSynchronize(st);
            printf("graph 32 MiB stream kernels: %.2f us per kernel\n", (now() - t0) / nk * 1e6);
        }
    }
    return 0;
}
