// Small HBM-bound helpers around the ViT encoder: patch-row packing and rotary tables.
//
//   pack_rows     pixel_values [N,588] (fp32 or bf16) -> compute dtype, K padded to a multiple of
//                 the GEMM slab (640) so that patch-embed (nn.Conv2d k=s=14,
//                 model/modeling_videollama3_encoder.py:194-210) is a plain row-major GEMM.
//   vit_rope_table cos/sin of the 2-D rotary angles in merge-window row order
//                 (rot_pos_emb, model/modeling_videollama3_encoder.py:405-434; VisionRotaryEmbedding
//                 :173-183): row r of a frame -> window r/ms^2, (dy,dx) inside it; angles are
//                 [h*inv_freq[0..n), w*inv_freq[0..n)].
//   llm_rope_table cos/sin(pos * inv_freq) for Qwen2 (theta 1e6, rotate_half pairs d, d+hd/2).
#include "common.h"
#include "kernels.h"

namespace {

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void pack_rows_kernel(const TI* __restrict__ in, long ld_in, TO* __restrict__ out,
                                                        long ld_out, int rows, int cols_in, int cols_out) {
    const int cpr = cols_out >> 2;  // 4-element groups per output row
    const long total = (long)rows * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cpr;
        const int c = (int)(i % cpr) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c + 3 < cols_in) {
            v = ld4_f<TI>(in + r * ld_in + c);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < cols_in) v[e] = ld_f<TI>(in + r * ld_in + c + e);
        }
        st4_f<TO>(out + r * ld_out + c, v);
    }
}

__global__ __launch_bounds__(256) void vit_rope_kernel(float* cos_t, float* sin_t, int row0, int t, int gh, int gw,
                                                       int ms, const float* inv_freq, int nf) {
    const int per_frame = gh * gw;
    const long total = (long)t * per_frame * 2 * nf;
    const int wpr = gw / ms;  // merge windows per row
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i % (2 * nf));
        const long row = i / (2 * nf);
        const int r = (int)(row % per_frame);
        const int win = r / (ms * ms), in = r % (ms * ms);
        const int hpos = (win / wpr) * ms + in / ms;
        const int wpos = (win % wpr) * ms + in % ms;
        const float ang = (f < nf) ? (float)hpos * inv_freq[f] : (float)wpos * inv_freq[f - nf];
        const long o = ((long)row0 + row) * (2 * nf) + f;
        if (sin_t) { cos_t[o] = cosf(ang); sin_t[o] = sinf(ang); }
        else { cos_t[2 * o] = cosf(ang); cos_t[2 * o + 1] = sinf(ang); }   // interleaved (cos, sin)
    }
}

// per-row (h | w << 16) position of the 2-D rotary embedding, same row order as vit_rope_kernel
__global__ __launch_bounds__(256) void vit_rowpos_kernel(int* rowpos, int row0, int t, int gh, int gw, int ms) {
    const int per_frame = gh * gw;
    const long total = (long)t * per_frame;
    const int wpr = gw / ms;
    for (long row = (long)blockIdx.x * blockDim.x + threadIdx.x; row < total; row += (long)gridDim.x * blockDim.x) {
        const int r = (int)(row % per_frame);
        const int win = r / (ms * ms), in = r % (ms * ms);
        const int hpos = (win / wpr) * ms + in / ms;
        const int wpos = (win % wpr) * ms + in % ms;
        rowpos[row0 + row] = hpos | (wpos << 16);
    }
}

// (cos, sin) of pos * inv_freq[f] for pos < maxpos: lut[pos][f][2]; the SAME expression as vit_rope_kernel, so a
// GEMM tile that reads the LUT and one that reads the per-row table see bit-identical factors
__global__ __launch_bounds__(256) void vit_rope_lut_kernel(float* lut, int maxpos, const float* inv_freq, int nf) {
    const int total = maxpos * nf;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int f = i % nf, pos = i / nf;
        const float ang = (float)pos * inv_freq[f];
        lut[2 * i] = cosf(ang);
        lut[2 * i + 1] = sinf(ang);
    }
}

__global__ __launch_bounds__(256) void llm_rope_kernel(float* cos_t, float* sin_t, const int* pos, int pos0, int rows,
                                                       const float* inv_freq, int nf) {
    const long total = (long)rows * nf;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i % nf);
        const int r = (int)(i / nf);
        const int pp = pos ? pos[r] : pos0 + r;
        const float ang = (float)pp * inv_freq[f];
        if (sin_t) { cos_t[i] = cosf(ang); sin_t[i] = sinf(ang); }
        else { cos_t[2 * i] = cosf(ang); cos_t[2 * i + 1] = sinf(ang); }   // interleaved (cos, sin)
    }
}

// cu_seqlens (modeling_videollama3_encoder.py:439-440) and the same-frame row ranges of the eager-global mode for one
// video, written on the device: frame f of the video covers rows [row0 + f*per, row0 + (f+1)*per)
__global__ __launch_bounds__(256) void vit_segments_kernel(int* __restrict__ cu, int* __restrict__ lo, int* __restrict__ hi,
                                                           int row0, int frame0, int t, int per) {
    const long total = (long)t * per;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 <= t) {
        if (i0 > 0) cu[frame0 + i0] = row0 + (int)i0 * per;
        else if (frame0 == 0) cu[0] = 0;
    }
    if (!lo) return;
    for (long i = i0; i < total; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i / per);
        lo[row0 + i] = row0 + f * per;
        hi[row0 + i] = row0 + (f + 1) * per;
    }
}

// positions restart at every segment: pos[i] = i - cu[s] for cu[s] <= i < cu[s+1]; one workgroup per segment
__global__ __launch_bounds__(256) void seg_positions_kernel(const int* __restrict__ cu, int* __restrict__ pos) {
    const int b = cu[blockIdx.x], e = cu[blockIdx.x + 1];
    for (int i = b + threadIdx.x; i < e; i += 256) pos[i] = i - b;
}

inline int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

int cogs_k_pack_rows(hipStream_t st, int in_dtype, int out_dtype, const void* in, long ld_in, void* out, long ld_out,
                     int rows, int cols_in, int cols_out) {
    if (rows <= 0) return COGS_OK;
    if (cols_out % 4 || cols_in > cols_out || ld_in % 4 || ld_out % 4) return COGS_E_INVALID;
    const int g = grid_for((long)rows * (cols_out / 4));
    if (in_dtype == COGS_DT_F32 && out_dtype == COGS_DT_BF16)
        hipLaunchKernelGGL((pack_rows_kernel<float, bf16_t>), dim3(g), dim3(256), 0, st, (const float*)in, ld_in,
                           (bf16_t*)out, ld_out, rows, cols_in, cols_out);
    else if (in_dtype == COGS_DT_BF16 && out_dtype == COGS_DT_BF16)
        hipLaunchKernelGGL((pack_rows_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)in, ld_in,
                           (bf16_t*)out, ld_out, rows, cols_in, cols_out);
    else if (in_dtype == COGS_DT_F32 && out_dtype == COGS_DT_F32)
        hipLaunchKernelGGL((pack_rows_kernel<float, float>), dim3(g), dim3(256), 0, st, (const float*)in, ld_in,
                           (float*)out, ld_out, rows, cols_in, cols_out);
    else if (in_dtype == COGS_DT_BF16 && out_dtype == COGS_DT_F32)
        hipLaunchKernelGGL((pack_rows_kernel<bf16_t, float>), dim3(g), dim3(256), 0, st, (const bf16_t*)in, ld_in,
                           (float*)out, ld_out, rows, cols_in, cols_out);
    else
        return COGS_E_INVALID;
    return COGS_LAUNCH_CHECK();
}

int cogs_k_vit_rope_table(hipStream_t st, float* cos_t, float* sin_t, int row0, int t, int gh, int gw, int ms,
                          const float* inv_freq, int n_freq) {
    if (t <= 0 || gh <= 0 || gw <= 0 || ms <= 0 || gh % ms || gw % ms) return COGS_E_INVALID;
    const long total = (long)t * gh * gw * 2 * n_freq;
    hipLaunchKernelGGL(vit_rope_kernel, dim3(grid_for(total)), dim3(256), 0, st, cos_t, sin_t, row0, t, gh, gw, ms,
                       inv_freq, n_freq);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_vit_rope_lut(hipStream_t st, int* rowpos, int row0, int t, int gh, int gw, int ms, float* lut, int maxpos,
                        const float* inv_freq, int n_freq) {
    if (t <= 0 || gh <= 0 || gw <= 0 || ms <= 0 || gh % ms || gw % ms) return COGS_E_INVALID;
    hipLaunchKernelGGL(vit_rowpos_kernel, dim3(grid_for((long)t * gh * gw)), dim3(256), 0, st, rowpos, row0, t, gh, gw, ms);
    if (lut)
        hipLaunchKernelGGL(vit_rope_lut_kernel, dim3(grid_for((long)maxpos * n_freq)), dim3(256), 0, st, lut, maxpos, inv_freq,
                           n_freq);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_vit_segments(hipStream_t st, int* cu, int* lo, int* hi, int row0, int frame0, int t, int per) {
    if (t <= 0 || per <= 0 || !cu) return COGS_E_INVALID;
    const long total = lo ? (long)t * per : (long)t + 1;
    long g = (total + 255) / 256, gmin = ((long)t + 1 + 255) / 256;   // the first t+1 threads write cu
    if (g < gmin) g = gmin;
    if (g > 4096 && gmin <= 4096) g = 4096;
    hipLaunchKernelGGL(vit_segments_kernel, dim3((unsigned)g), dim3(256), 0, st, cu, lo, hi, row0, frame0, t, per);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_seg_positions(hipStream_t st, const int* cu, int nseg, int* pos) {
    if (nseg <= 0) return COGS_OK;
    hipLaunchKernelGGL(seg_positions_kernel, dim3(nseg), dim3(256), 0, st, cu, pos);
    return COGS_LAUNCH_CHECK();
}

int cogs_k_llm_rope_table(hipStream_t st, float* cos_t, float* sin_t, const int* pos, int pos0, int rows,
                          const float* inv_freq, int n_freq) {
    if (rows <= 0) return COGS_OK;
    hipLaunchKernelGGL(llm_rope_kernel, dim3(grid_for((long)rows * n_freq)), dim3(256), 0, st, cos_t, sin_t, pos, pos0,
                       rows, inv_freq, n_freq);
    return COGS_LAUNCH_CHECK();
}
