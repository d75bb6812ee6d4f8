// launch-boundary price on this box: N dependent trivial kernels on one stream, eager vs hipGraph replay,
// and a chain shaped like one decode layer (6 kernels of 256..4736 workgroups doing a tiny amount of work).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
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
    return 0;
}
