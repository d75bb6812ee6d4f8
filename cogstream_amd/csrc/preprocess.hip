// GPU pre-processing: uint8 frames -> Pillow-exact bicubic resize -> x/255,
// (x-0.5)/0.5 -> merge-window-major [N,588] patch rows, written straight in the encoder's input dtype.
//
// Replaces Videollama3ImageProcessor._preprocess (model/image_processing_videollama3.py:235-347): PIL
// Image.resize(BICUBIC) + rescale + normalize + the reshape/transpose patchify (:326-345). The resample is
// Pillow's fixed-point algorithm restated exactly (two passes, horizontal first, 22-bit integer coefficients
// built on the host by cogstream_amd.processing.resample_coeffs, uint8 rounding after each pass), so the uint8
// image -- and, through the [3,256] value table (processing.pixel_value_table), pixel_values -- is bit-identical to the reference's host path.
// HBM-bound, integer/byte work: pass 1 reads the frames once, pass 2 reads the intermediate once and writes
// pixel_values once.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int PBITS = 22;

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// horizontal pass: one thread per (image row, output column), 3 channels
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ tmp,
                                                       long rows, int W, int TW, const int* __restrict__ bounds,
                                                       const int* __restrict__ kk, int ksize) {
    const long total = rows * TW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / TW;
        const int xx = (int)(i % TW);
        const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
        const uint8_t* p = in + (row * W + x0) * 3;
        const int* k = kk + (long)xx * ksize;
        int s0 = 1 << (PBITS - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < n; ++x) {
            const int c = k[x];
            s0 += p[3 * x] * c;
            s1 += p[3 * x + 1] * c;
            s2 += p[3 * x + 2] * c;
        }
        uint8_t* o = tmp + i * 3;
        o[0] = (uint8_t)clip8(s0 >> PBITS);
        o[1] = (uint8_t)clip8(s1 >> PBITS);
        o[2] = (uint8_t)clip8(s2 >> PBITS);
    }
}

// vertical pass + rescale/normalise + patchify: one thread per output element (element index fastest)
template <typename TO>
__global__ __launch_bounds__(256) void resize_v_patchify_kernel(const uint8_t* __restrict__ tmp, TO* __restrict__ out,
                                                                int T, int H, int TW, int gh, int gw, int ms,
                                                                const int* __restrict__ bounds,
                                                                const int* __restrict__ kk, int ksize,
                                                                const float* __restrict__ table) {
    const int per_frame = gh * gw;
    const long total = (long)T * per_frame * 588;
    const int wpr = gw / ms;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / 588;
        const int e = (int)(i % 588);
        const int c = e / 196, py = (e % 196) / 14, px = e % 14;
        const int f = (int)(r / per_frame), rr = (int)(r % per_frame);
        const int win = rr / (ms * ms), inw = rr % (ms * ms);
        const int prow = (win / wpr) * ms + inw / ms, pcol = (win % wpr) * ms + inw % ms;
        const int y = prow * 14 + py, x = pcol * 14 + px;
        const int y0 = bounds[2 * y], n = bounds[2 * y + 1];
        const uint8_t* p = tmp + (((long)f * H + y0) * TW + x) * 3 + c;
        const int* k = kk + (long)y * ksize;
        int s = 1 << (PBITS - 1);
        for (int j = 0; j < n; ++j) s += p[(long)j * TW * 3] * k[j];
        // rescale + normalise through the host-built per-channel byte table (reference arithmetic, exact)
        st_f<TO>(out + i, table[c * 256 + clip8(s >> PBITS)]);
    }
}

inline int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

int cogs_k_preprocess(hipStream_t st, const uint8_t* frames, int T, int H, int W, int th, int tw, int ms,
                      const int* bx, const int* kx, int ksx, const int* by, const int* ky, int ksy, void* out,
                      int out_dtype, const float* table, uint8_t* tmp) {
    if (T <= 0 || th % (14 * ms) || tw % (14 * ms)) return COGS_E_INVALID;
    const long rows = (long)T * H;
    hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for(rows * tw)), dim3(256), 0, st, frames, tmp, rows, W, tw, bx, kx, ksx);
    const int gh = th / 14, gw = tw / 14;
    const long total = (long)T * gh * gw * 588;
    if (out_dtype == COGS_DT_BF16)
        hipLaunchKernelGGL(resize_v_patchify_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, tmp, (bf16_t*)out, T,
                           H, tw, gh, gw, ms, by, ky, ksy, table);
    else
        hipLaunchKernelGGL(resize_v_patchify_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, tmp, (float*)out, T,
                           H, tw, gh, gw, ms, by, ky, ksy, table);
    return COGS_LAUNCH_CHECK();
}
