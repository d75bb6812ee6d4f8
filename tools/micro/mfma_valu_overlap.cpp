// Does the matrix pipe of a gfx950 SIMD run beside its VALU / transcendental pipe? One wave (or two) per SIMD issues, per
// loop iteration (16 cycles of loop control included), M independent MFMAs and E independent v_exp_f32 / A independent v_add_f32, in a fixed interleaved order
// (asm volatile keeps it); the shader-clock cycles per iteration say whether the times add or overlap.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.cpp -o tools/micro/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

#define MF16(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MF32(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(one))
#define CVT(x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(y))

// MODE: 0 MFMA16 only (8 per iter), 1 exp only (8 per iter), 2 8 MFMA16 + 8 exp interleaved, 3 8 MFMA16 + 4 exp,
//       4 add only (32 per iter), 5 8 MFMA16 + 32 add, 6 MFMA32 only (4 per iter), 7 4 MFMA32 + 8 exp, 8 4 MFMA32 + 32 add,
//       9 8 MFMA16 + 4 exp + 16 add, 10 odd waves MFMA16 only / even waves exp only (two waves per SIMD), 11-14 dependent accumulator chains
template <int MODE>
__global__ void k(long long* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f); b[i] = (__bf16)(i * 0.01f); }
    f32x4 c[8];
    f32x16 C[4];
    float x[8], one = 1e-9f;
    for (int i = 0; i < 8; ++i) { c[i] = f32x4{0, 0, 0, 0}; x[i] = -1.f - i; }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) C[i][j] = 0.f;
    const int w = threadIdx.x >> 6;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) { for (int i = 0; i < 8; ++i) MF16(c[i]); }
        if constexpr (MODE == 1) { for (int i = 0; i < 8; ++i) EXP(x[i]); }
        if constexpr (MODE == 2) { for (int i = 0; i < 8; ++i) { MF16(c[i]); EXP(x[i]); } }
        if constexpr (MODE == 3) { for (int i = 0; i < 8; ++i) { MF16(c[i]); if (i & 1) EXP(x[i]); } }
        if constexpr (MODE == 4) { for (int r = 0; r < 4; ++r) for (int i = 0; i < 8; ++i) ADD(x[i]); }
        if constexpr (MODE == 5) { for (int i = 0; i < 8; ++i) { MF16(c[i]); ADD(x[0]); ADD(x[1]); ADD(x[2]); ADD(x[3]); } }
        if constexpr (MODE == 6) { for (int i = 0; i < 4; ++i) MF32(C[i]); }
        if constexpr (MODE == 7) { for (int i = 0; i < 4; ++i) { MF32(C[i]); EXP(x[2 * i]); EXP(x[2 * i + 1]); } }
        if constexpr (MODE == 8) { for (int i = 0; i < 4; ++i) { MF32(C[i]); for (int j = 0; j < 8; ++j) ADD(x[j]); } }
        if constexpr (MODE == 9) { for (int i = 0; i < 8; ++i) { MF16(c[i]); if (i & 1) EXP(x[i]); ADD(x[(i + 2) & 7]); ADD(x[(i + 4) & 7]); } }
        if constexpr (MODE == 11) { for (int i = 0; i < 4; ++i) MF32(C[0]); }                       // ONE accumulator: a dependent chain
        if constexpr (MODE == 12) { for (int i = 0; i < 8; ++i) MF16(c[0]); }
        if constexpr (MODE == 13) { for (int i = 0; i < 4; ++i) { MF32(C[0]); EXP(x[2 * i]); EXP(x[2 * i + 1]); } }     // chain + exps in its gaps
        if constexpr (MODE == 14) { for (int i = 0; i < 2; ++i) { MF32(C[0]); MF32(C[1]); } }          // two interleaved chains
        if constexpr (MODE == 10) {
            if (w & 4) { for (int i = 0; i < 8; ++i) MF16(c[i]); } else { for (int i = 0; i < 8; ++i) EXP(x[i]); }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");        // inline-asm MFMA results are read below: no compiler hazard handling
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0] + x[i];
    for (int i = 0; i < 4; ++i) s += C[i][0];
    if (s == 12345.f) out[1] = 1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { out[2 + 2 * w] = t0; out[3 + 2 * w] = t1; }
}

template <int MODE>
void run(const char* what, int threads, long long* d) {
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, 100);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, iters);
    long long h[18];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // a SIMD issues its oldest wave first, so one wave's own time says little with two waves: first start -> last end
    long long lo = h[2], hi = h[3], own_min = h[3] - h[2], own_max = own_min;
    for (int w = 0; w < threads / 64; ++w) {
        lo = h[2 + 2 * w] < lo ? h[2 + 2 * w] : lo; hi = h[3 + 2 * w] > hi ? h[3 + 2 * w] : hi;
        const long long o = h[3 + 2 * w] - h[2 + 2 * w];
        own_min = o < own_min ? o : own_min; own_max = o > own_max ? o : own_max;
    }
    printf("%-76s %d wave(s)/SIMD: %7.1f cycles per iteration (per wave: %.1f - %.1f)\n", what, threads / 256, (double)(hi - lo) / iters,
           (double)own_min / iters, (double)own_max / iters);
}

int main() {
    long long* d;
    (void)hipMalloc(&d, 256);
    for (int th : {256, 512}) {
        run<0>("8 MFMA 16x16x32", th, d);
        run<1>("8 v_exp_f32", th, d);
        run<2>("8 MFMA 16x16x32 + 8 v_exp_f32, interleaved", th, d);
        run<3>("8 MFMA 16x16x32 + 4 v_exp_f32 (the prompt kernel's ratio)", th, d);
        run<4>("32 v_add_f32", th, d);
        run<5>("8 MFMA 16x16x32 + 32 v_add_f32, interleaved", th, d);
        run<9>("8 MFMA 16x16x32 + 4 v_exp_f32 + 16 v_add_f32", th, d);
        run<6>("4 MFMA 32x32x16", th, d);
        run<7>("4 MFMA 32x32x16 + 8 v_exp_f32", th, d);
        run<8>("4 MFMA 32x32x16 + 32 v_add_f32", th, d);
        run<11>("4 MFMA 32x32x16 on ONE accumulator (dependent chain)", th, d);
        run<14>("4 MFMA 32x32x16 on two accumulators, alternating", th, d);
        run<13>("4 MFMA 32x32x16 on one accumulator + 8 v_exp_f32 between them", th, d);
        run<12>("8 MFMA 16x16x32 on ONE accumulator (dependent chain)", th, d);
    }
    run<10>("waves 4-7: 8 MFMA 16x16x32; waves 0-3: 8 v_exp_f32 (per SIMD one of each)", 512, d);
    return 0;
}
