// Shared device helpers for the cogstream_amd HIP kernels (gfx950 / CDNA4 only).
//
// Conventions used by every kernel in this directory:
//   * wavefront = 64 lanes, workgroups are multiples of 64 threads;
//   * bf16 tensors are passed as raw 16-bit patterns (bf16_t), converted with a
//     plain cast so that hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even,
//     NaN stays NaN);
//   * all reductions use a fixed order so results are reproducible run to run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define COGS_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t b) {
    return __uint_as_float(((uint32_t)b) << 16);
}
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}
// pack two floats into one dword of two bf16 (lo in bits 0..15)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// element load/store helpers templated on the storage type (bf16_t or float)
template <typename T> __device__ __forceinline__ float ld_f(const T* p);
template <> __device__ __forceinline__ float ld_f<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <> __device__ __forceinline__ float ld_f<float>(const float* p) { return *p; }
template <typename T> __device__ __forceinline__ void st_f(T* p, float v);
template <> __device__ __forceinline__ void st_f<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
template <> __device__ __forceinline__ void st_f<float>(float* p, float v) { *p = v; }

// load 4 consecutive elements as floats (8-byte / 16-byte aligned)
template <typename T> __device__ __forceinline__ f32x4 ld4_f(const T* p);
template <> __device__ __forceinline__ f32x4 ld4_f<bf16_t>(const bf16_t* p) {
    u32x2 w = *reinterpret_cast<const u32x2*>(p);
    f32x4 r = {bf_lo(w[0]), bf_hi(w[0]), bf_lo(w[1]), bf_hi(w[1])};
    return r;
}
template <> __device__ __forceinline__ f32x4 ld4_f<float>(const float* p) {
    return *reinterpret_cast<const f32x4*>(p);
}
template <typename T> __device__ __forceinline__ void st4_f(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4_f<bf16_t>(bf16_t* p, f32x4 v) {
    u32x2 w;
    w[0] = pack_bf2(v[0], v[1]);
    w[1] = pack_bf2(v[2], v[3]);
    *reinterpret_cast<u32x2*>(p) = w;
}
template <> __device__ __forceinline__ void st4_f<float>(float* p, f32x4 v) {
    *reinterpret_cast<f32x4*>(p) = v;
}

// load 8 consecutive elements as floats (bf16: one 16-B load, float: two)
template <typename T> __device__ __forceinline__ void ld8_f(const T* p, float (&o)[8]);
template <> __device__ __forceinline__ void ld8_f<bf16_t>(const bf16_t* p, float (&o)[8]) {
    u32x4 w = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[2 * i] = bf_lo(w[i]);
        o[2 * i + 1] = bf_hi(w[i]);
    }
}
template <> __device__ __forceinline__ void ld8_f<float>(const float* p, float (&o)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[i] = a[i];
        o[4 + i] = b[i];
    }
}
template <typename T> __device__ __forceinline__ void st8_f(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8_f<bf16_t>(bf16_t* p, const float (&v)[8]) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<u32x4*>(p) = w;
}
template <> __device__ __forceinline__ void st8_f<float>(float* p, const float (&v)[8]) {
    f32x4 a = {v[0], v[1], v[2], v[3]};
    f32x4 b = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}

// value of lane (i ^ X) for X in {8, 4, 2, 1}, by DPP (data-parallel primitives: a few cycles, no LDS). hipcc compiles
// __shfl_xor to ds_bpermute_b32 -- an LDS-crossbar round trip of ~100 cycles per step, four DEPENDENT ones at the end of
// every wave reduction (rocprofv3 / disassembly, round 3). Within a 16-lane row: xor 8 = rotate by 8; xor 4 = rotate by
// 12 for the lanes with bit 2 clear (banks 0 and 2) and by 4 for the others (banks 1 and 3); xor 2 / xor 1 = quad
// permutations. Same lanes paired in the same order as before: results are bit-identical.
template <int CTRL, int BANKS>
__device__ __forceinline__ float dpp_take(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, 0xf, BANKS, false));
}
template <int X>
__device__ __forceinline__ float lane_xor_dpp(float v) {
    static_assert(X == 8 || X == 4 || X == 2 || X == 1, "within a 16-lane row");
    if constexpr (X == 8) return dpp_take<0x128, 0xf>(v, v);                       // row_ror:8
    else if constexpr (X == 4) return dpp_take<0x124, 0xa>(dpp_take<0x12C, 0x5>(v, v), v);   // row_ror:12 | row_ror:4
    else if constexpr (X == 2) return dpp_take<0x4E, 0xf>(v, v);                   // quad_perm [2,3,0,1]
    else return dpp_take<0xB1, 0xf>(v, v);                                         // quad_perm [1,0,3,2]
}

// 64-lane butterfly reductions (fixed order -> deterministic). The xor-32 and xor-16 exchanges go through
// v_permlane32_swap / v_permlane16_swap (both operands = v: afterwards the two results hold "mine" and "partner's" in
// some order, and the operation is commutative), xor 8..1 through DPP (above). Same values, bit for bit, as the
// __shfl_xor butterfly they replace.
template <typename F>
__device__ __forceinline__ float wave_butterfly(float v, F op) {
    {
        const unsigned b = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
        v = op(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
    }
    {
        const unsigned b = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
        v = op(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
    }
    v = op(v, lane_xor_dpp<8>(v));
    v = op(v, lane_xor_dpp<4>(v));
    v = op(v, lane_xor_dpp<2>(v));
    v = op(v, lane_xor_dpp<1>(v));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    return wave_butterfly(v, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float wave_max(float v) {
    return wave_butterfly(v, [](float a, float b) { return fmaxf(a, b); });
}
__device__ __forceinline__ float wave_min(float v) {
    return wave_butterfly(v, [](float a, float b) { return fminf(a, b); });
}

// activations (fp32)
__device__ __forceinline__ float gelu_tanh_f(float x) {
    // 0.5*x*(1+tanh(u)), u = sqrt(2/pi)*(x+0.044715x^3)   (ACT2FN["gelu_pytorch_tanh"])
    //   = x * sigmoid(2u) = x / (1 + 2^(x*(a + b*x^2))),  a = -2*sqrt(2/pi)*log2(e), b = a*0.044715
    // 7 VALU incl. one v_exp_f32 and one v_rcp_f32 (1 ulp) instead of libm tanhf (~50 instructions); the
    // fp32 error (~2e-7 relative) is far below the bf16 output rounding and inside the fp32-mode tolerance
    const float a = -2.3022082f, b = -0.10294324f;
    const float t = x * fmaf(b, x * x, a);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
// the same on four values in packed fp32 arithmetic: 2 x (v_pk_mul, v_pk_fma, v_pk_mul, v_pk_add, v_pk_mul) + 4 v_exp + 4 v_rcp
// instead of 4 x 5 scalar VALU + the same transcendentals. Written on vectors because hipcc's SLP pass only packs about half of
// the scalar form inside the GEMM epilogues, and every VALU instruction there is taken from the matrix pipe
// (profiles/r5_prefill_attn_anatomy.txt section 4: ~2 cycles per instruction beside 16x16x32 MFMAs; v_exp / v_rcp run beside it).
// Bit-identical to gelu_tanh_f per element (same operations in the same order).
__device__ __forceinline__ f32x4 gelu_tanh_4(f32x4 x) {
    const float a = -2.3022082f, b = -0.10294324f;
    const f32x4 a4 = {a, a, a, a}, b4 = {b, b, b, b}, one = {1.f, 1.f, 1.f, 1.f};
#ifdef COGS_EPI_SCALAR_MATH      // A/B builds (tools/build_alt.sh gemm -DCOGS_EPI_SCALAR_MATH): the scalar form of rounds 1-4
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = gelu_tanh_f(x[e]);
    return y;
#endif
    const f32x4 t = x * __builtin_elementwise_fma(b4, x * x, a4);
    f32x4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = __builtin_amdgcn_exp2f(t[e]);
    d += one;
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = __builtin_amdgcn_rcpf(d[e]);
    return x * d;
}
__device__ __forceinline__ float gelu_erf_f(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
}
__device__ __forceinline__ float silu_f(float x) {
    // x * sigmoid(x) = x / (1 + 2^(-x log2 e)): v_exp_f32 + v_rcp_f32 (1 ulp) instead of an IEEE division (~10 VALU)
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// status codes shared with include/cogs.h
#define COGS_OK 0
#define COGS_E_INVALID (-1)
#define COGS_E_HIP (-2)
#define COGS_E_UNSUPPORTED (-3)
#define COGS_E_WORKSPACE (-4)

#define COGS_DT_BF16 0
#define COGS_DT_F32 1

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (kernel, device): raise it once per device the calling thread
// is on (bit per device ordinal). Racing threads may both set it -- the call is idempotent.
#include <atomic>
static inline void cogs_ensure_dyn_lds(const void* fn, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
}

static inline int cogs_hip_check(hipError_t e) { return e == hipSuccess ? COGS_OK : COGS_E_HIP; }
#define COGS_LAUNCH_CHECK() cogs_hip_check(hipGetLastError())
