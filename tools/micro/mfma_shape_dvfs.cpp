// Which bf16 MFMA shape does the chip hold the higher clock on, under load, on RANDOM data (review item 3c, round 6)?
// v_mfma_f32_16x16x32_bf16 (16 cycles, 16 384 FLOP) against v_mfma_f32_32x32x16_bf16 (32 cycles, 32 768 FLOP): equal FLOP per cycle,
// so at equal clocks the two loops deliver the same TFLOP/s -- a difference in wall time is a difference in the clock the chip
// holds (MI355X_MICROARCH.md, DVFS give-back item 7). Every CU runs 1 or 2 waves per SIMD, each wave a back-to-back stream of
// MFMAs on four independent accumulator sets with operands in registers (random bf16 from a buffer; zeros as the control).
// The in-kernel clock is measured the guide's way: d(s_memtime) / d(s_memrealtime) * 100 MHz over the loop of one workgroup.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/mfma_shape_dvfs.cpp -o tools/micro/mfma_shape_dvfs && tools/micro/mfma_shape_dvfs
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int SHAPE>   // 16 or 32
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ data, float* out, unsigned long long* clk, int iters) {
    const uint4 ra = data[(blockIdx.x * blockDim.x + threadIdx.x) * 2], rb = data[(blockIdx.x * blockDim.x + threadIdx.x) * 2 + 1];
    const bf16x8 a = __builtin_bit_cast(bf16x8, ra), b = __builtin_bit_cast(bf16x8, rb);
    f32x4 c[8];
    f32x16 C[4];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) C[i][j] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (SHAPE == 16) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);      // 8 x 16 cycles
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) C[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, C[i], 0, 0, 0);      // 4 x 32 cycles
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0];
    for (int i = 0; i < 4; ++i) s += C[i][0];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
static void run(const char* what, const uint4* data, float* out, unsigned long long* clk, int waves_per_simd) {
    const int threads = 256 * waves_per_simd, wgs = 256, iters = 400000;      // ~ 0.1-0.2 s per launch
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<SHAPE>, dim3(wgs), dim3(threads), 0, 0, data, out, clk, iters);     // warm-up: lets the clock settle
    std::vector<float> ts;
    std::vector<double> ghz;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<SHAPE>, dim3(wgs), dim3(threads), 0, 0, data, out, clk, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
        std::vector<unsigned long long> h(2 * wgs);
        (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> g;
        for (int i = 0; i < wgs; ++i) g.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
        std::sort(g.begin(), g.end());
        ghz.push_back(g[wgs / 2]);
    }
    std::sort(ts.begin(), ts.end()); std::sort(ghz.begin(), ghz.end());
    const double flop = (double)wgs * (threads / 64) * iters * 8.0 * 16384.0;
    printf("%-44s %d wave(s)/SIMD: %8.2f ms  %7.1f TFLOP/s  in-kernel clock %.3f GHz (median workgroup)\n", what, waves_per_simd, ts[1],
           flop / ts[1] / 1e9, ghz[1]);
}

static uint32_t rand_bf1() { float f = (float)rand() / RAND_MAX * 2.f - 1.f; uint32_t u; std::memcpy(&u, &f, 4); return u >> 16; }
static uint32_t rand_bf2() { return rand_bf1() | (rand_bf1() << 16); }      // two random bf16 in [-1, 1)

int main() {
    const size_t n = 256 * 512 * 2;
    std::vector<uint4> h(n);
    srand(1);
    for (auto& v : h) v = uint4{rand_bf2(), rand_bf2(), rand_bf2(), rand_bf2()};
    uint4 *drand, *dzero; float* out; unsigned long long* clk;
    (void)hipMalloc(&drand, n * 16); (void)hipMalloc(&dzero, n * 16); (void)hipMalloc(&out, 64); (void)hipMalloc(&clk, 256 * 16);
    (void)hipMemcpy(drand, h.data(), n * 16, hipMemcpyHostToDevice);
    (void)hipMemset(dzero, 0, n * 16);
    for (int w : {1, 2}) {
        run<16>("16x16x32 bf16, random operands", drand, out, clk, w);
        run<32>("32x32x16 bf16, random operands", drand, out, clk, w);
        run<16>("16x16x32 bf16, zero operands (control)", dzero, out, clk, w);
        run<32>("32x32x16 bf16, zero operands (control)", dzero, out, clk, w);
    }
    return 0;
}
