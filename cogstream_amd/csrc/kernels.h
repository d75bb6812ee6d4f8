// Internal launcher interface between the C-ABI layer (capi.hip) and the kernel files.
// Every launcher enqueues on the given stream, never synchronises, never allocates, and
// returns a COGS_* status (0 = ok).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define COGS_ACT_NONE 0
#define COGS_ACT_GELU_TANH 1
#define COGS_ACT_GELU_ERF 2
#define COGS_ACT_SWIGLU 3

struct CogsGemm {
    int dtype = 0;                 // COGS_DT_*
    const void* A = nullptr; long lda = 0;   // [M,K], elements per row
    const void* W = nullptr; long ldw = 0;   // [N,K]
    void* C = nullptr; long ldc = 0;         // [M,N] (SWIGLU: [M,N/2])
    const void* bias = nullptr;              // [N]
    const void* residual = nullptr; long ldr = 0;
    int M = 0, N = 0, K = 0;
    int act = 0;
    int out_f32 = 0;
    const float* rope_cos = nullptr;         // [M, head_dim/2], or interleaved [M, head_dim/2, 2] when rope_sin == nullptr
    const float* rope_sin = nullptr;
    int rope_cols = 0;
    int head_dim = 0;
    int force_small_tile = 0;                // testing: always use the 128x128 kernel
    int force_mid_tile = 0;                  // internal: skip the ping-pong kernel (remainder rows of a round-aligned split)
    const void* rms_gamma = nullptr;         // M == 1 only: RMS-normalise A on the fly with this weight
    float rms_eps = 0.f;
    // M == 1 only: output columns [kv_col0, kv_col0 + kv_dim) go to kv_k[0..kv_dim), the next kv_dim to kv_v
    // (single-token decode writes the new K/V row straight into the cache; no kv_append launch)
    void* kv_k = nullptr; void* kv_v = nullptr; int kv_col0 = 0; int kv_dim = 0;
    // rope epilogues only: columns [0, q_cols) (the Q heads) are multiplied by q_scale before rounding
    float q_scale = 1.f; int q_cols = 0;
    // optional 2-D rotary LUT (ViT): rope_lut [maxpos][head_dim/4][2] fp32 = (cos, sin) of pos*inv_freq[f], rope_rowpos
    // [M] = h | w << 16. The ping-pong kernel keeps the LUT in LDS and reads the factors from there (first half of a
    // head's pairs rotates with h, second half with w); rope_cos must still be the matching per-row table.
    const float* rope_lut = nullptr; const int* rope_rowpos = nullptr; int rope_maxpos = 0;
    // LayerNorm fused around the GEMM (gemm_epilogue.h EPI_ROWSTAT / EPI_LNFOLD; bf16 kernels):
    float* row_stats = nullptr;              // != null: also write per-row partial (sum, sum of squares) of the outputs,
                                             //   [M][N/64][2] fp32 (N % 64 == 0)
    // head-major output (gemm_epilogue.h EPI_HM; bf16, rotary epilogues): hm_rows > 0 = rows of the whole output buffer, C is
    // [N / hm_cols][hm_cols / head_dim][hm_rows][head_dim] instead of [M][N] (ldc ignored); M <= hm_rows rows are written
    long hm_rows = 0; int hm_cols = 0;
    const float* ln_ab = nullptr;            // != null: y = ln_ab[r][0] * acc + col_c[n] instead of acc + bias. W = rows of
    const float* col_c = nullptr;            //   W0*diag(gamma) made ZERO-SUM over k, col_c = bias + W0.beta; bias must be null
                                             //   (include/cogs.h, cogs_gemm_desc.ln_ab, has the contract)
};
int cogs_k_gemm(hipStream_t st, const CogsGemm& g);
void cogs_k_gemm_co_streams(int streams);   // hint for the few-tile choice: streams fed with GEMMs of this size at once (thread-local)
// (a, b) = (rstd, -rstd * mean) per row from the EPI_ROWSTAT partials [rows][tiles][2]: ab [rows][2]
int cogs_k_ln_finalize(hipStream_t st, const float* stat_part, int rows, int tiles, int H, float eps, float* ab);
// (a, b) = (rstd, -rstd * mean) per row from the EPI_ROWSTAT partials: ab [rows][2]
int cogs_k_ln_finalize(hipStream_t st, const float* stat_part, int rows, int tiles, int H, float eps, float* ab);

struct CogsAttn {
    int dtype = 0;
    const void* Q = nullptr; const void* K = nullptr; const void* V = nullptr; void* O = nullptr;
    long ldq = 0, ldk = 0, ldv = 0, ldo = 0;
    const int* cu_seqlens = nullptr;  // device [nseg+1]; block-diagonal segments (q and k alike)
    int nseg = 1;
    int max_seqlen = 0;
    const int* row_lo = nullptr;      // device [q_len]: same-segment key range, enables the
    const int* row_hi = nullptr;      //   "global attention + bias" mode of the eager reference
    float bias = 0.f;
    int q_len = 0, kv_len = 0;
    int hq = 1, hkv = 1, head_dim = 0;
    float scale = 1.f;
    int causal = 0;
    int q_pos0 = 0;
    int force_rowwise = 0;
    int nsplit = 1;                   // >1: split the keys over blocks (decode); needs ws
    void* ws = nullptr;               // nsplit*q_len*hq*(head_dim+2) floats
    size_t ws_bytes = 0;
    int q_prescaled = 0;              // Q already multiplied by scale*log2(e) (bf16 MFMA kernels only; no bias mode)
    int uniform_seqlen = 0;           // internal hint: > 0 = every cu_seqlens segment has exactly this many rows, in order
    long head_stride = 0;             // > 0: head-major Q / K / V (include/cogs.h cogs_attn_desc.head_stride); ViT kernel only
};
int cogs_k_attention_vit(hipStream_t st, const struct CogsAttn& a);   // attn_vit.hip: block-diagonal, hd 72, pre-scaled Q
int cogs_k_attention_decode(hipStream_t st, const struct CogsAttn& a, float* part_o, float* part_ml);   // attn_decode.hip: one query row, hd 128, key-split partials
int cogs_k_attention(hipStream_t st, const CogsAttn& a);
long cogs_k_gemm_launch_count();   // kernels launched by cogs_k_gemm so far on this thread


// norms
int cogs_k_layernorm(hipStream_t st, int dtype, const void* x, void* y, const void* gamma, const void* beta,
                     int rows, int H, float eps);
int cogs_k_rmsnorm(hipStream_t st, int dtype, const void* x, void* y, const void* gamma, int rows, int H, float eps);
// y[r] = mean_{i<group} LN(x[group*r+i])   (post_layernorm + 2x2 bilinear merge)
int cogs_k_ln_merge(hipStream_t st, int dtype, const void* x, void* y, const void* gamma, const void* beta,
                    int out_rows, int group, int H, float eps);

// vision front-end helpers
int cogs_k_pack_rows(hipStream_t st, int in_dtype, int out_dtype, const void* in, long ld_in, void* out,
                     long ld_out, int rows, int cols_in, int cols_out);
int cogs_k_vit_rope_table(hipStream_t st, float* cos_t, float* sin_t, int row0, int t, int gh, int gw, int ms,
                          const float* inv_freq, int n_freq);
// cu_seqlens[frame0+1 .. frame0+t] (and cu[0] for the first video) + optional same-frame row ranges, on the device
int cogs_k_vit_segments(hipStream_t st, int* cu, int* lo, int* hi, int row0, int frame0, int t, int per);
// pos[i] = i - cu[segment(i)]
int cogs_k_seg_positions(hipStream_t st, const int* cu, int nseg, int* pos);
int cogs_k_vit_rope_lut(hipStream_t st, int* rowpos, int row0, int t, int gh, int gw, int ms, float* lut, int maxpos,
                        const float* inv_freq, int n_freq);
int cogs_k_llm_rope_table(hipStream_t st, float* cos_t, float* sin_t, const int* pos, int pos0, int rows,
                          const float* inv_freq, int n_freq);

// GPU pre-processing: Pillow-exact bicubic resize + normalise + patchify (tmp: T*H*tw*3 bytes)
int cogs_k_preprocess(hipStream_t st, const uint8_t* frames, int T, int H, int W, int th, int tw, int ms,
                      const int* bx, const int* kx, int ksx, const int* by, const int* ky, int ksy, void* out,
                      int out_dtype, const float* table, uint8_t* tmp);

// token compression
int cogs_k_pixdiff_mask(hipStream_t st, int dtype, const void* pix, int t, int tokens_per_frame, int row_elems,
                        float thr, int min_tokens, uint8_t* mask);
int cogs_k_mask_fix(hipStream_t st, uint8_t* mask, int t, int tokens_per_frame, int min_tokens, const uint8_t* minor);
int cogs_k_frame_mean_to_slot0(hipStream_t st, int dtype, void* feats, int P, int D, const int* frames, int n_frames);
int cogs_k_gather_rows(hipStream_t st, int dtype, const void* table_a, const void* table_b, const int64_t* idx,
                       void* out, int rows, int D);
int cogs_k_mean_rows(hipStream_t st, int dtype, const void* x, long ldx, int rows, int D, float* out);
int cogs_k_cosine(hipStream_t st, const float* a, const float* b, int n, int D, float* out);

// time-aware k-means
int cogs_k_kmeans_sqdist(hipStream_t st, int dtype, const void* feats, int T, long PD, const float* centres,
                         const int* centre_rows, int K, float* ws, int nslices, float* dist2);
int cogs_k_kmeans_pp_step(hipStream_t st, int dtype, const void* feats, int T, long PD, int row, int first,
                          float* nearest2, float* probs_host, float* ws, int nslices);
int cogs_k_kmeans_pp(hipStream_t st, int dtype, const void* feats, int T, long PD, int K, int first_row, const float* q, int* idx,
                     int* zero_flag, float* nearest2, float* ws, int nslices);
int cogs_k_kmeans_margins(hipStream_t st, int T, long PD, int K, float* ws, float* min_margin, int* rows_below);
int cogs_k_kmeans_lloyd(hipStream_t st, int dtype, const void* feats, const float* ts, int T, long PD, int K, float alpha,
                        int max_iter, float tol, const int* pool_host, int pool_len, float* centres, float* centre_ts,
                        int64_t* assign, int* iterations, int* reseeds_used, int* exhausted, float* ws, int nslices);
size_t cogs_k_kmeans_ws(int T, long PD, int K, int* nslices);
int cogs_k_kmeans_assign(hipStream_t st, const float* dist2, const float* ts, const float* centre_ts, int T, int K,
                         float alpha, int64_t* assign, int* counts);
int cogs_k_kmeans_update(hipStream_t st, int dtype, const void* feats, const float* ts, int T, long PD, int K,
                         const int64_t* assign, const int* reseed_rows, float* centres, float* centre_ts,
                         float* ws, int nblk, float* shift_out);
int cogs_k_kmeans_update_blocks(long PD);
int cogs_k_select_near(hipStream_t st, const float* dist2, const int64_t* assign, int T, int K, int n, int64_t* picks,
                       int* counts);

// LLM helpers
// append S rows of K and V (adjacent column blocks of the fused qkv buffer) to the two caches in one launch
int cogs_k_kv_append(hipStream_t st, int dtype, const void* k_src, const void* v_src, long ld_src, void* k_dst,
                     void* v_dst, long ld_dst, int rows, int cols);
int cogs_k_copy_cols(hipStream_t st, int dtype, const void* src, long ld_src, void* dst, long ld_dst, int rows, int cols);
int cogs_k_argmax(hipStream_t st, const float* logits, int n, int64_t* out, float* ws);
// HF logits processors on one fp32 row: repetition penalty over `prev` (gather-then-scatter, so
// duplicates are penalised once), allowed-id mask (others -> -inf), temperature; `tmp` >= n_prev floats
int cogs_k_logits_process(hipStream_t st, float* logits, int n, const int64_t* prev, int n_prev, float rep_penalty,
                          const int32_t* allowed, int n_allowed, float temperature, float* tmp);
// the top_k largest logits in descending order (ties: lower index first); ws >= 2*n floats
int cogs_k_topk(hipStream_t st, const float* logits, int n, int top_k, float* topk_val, int32_t* topk_idx, float* ws);
// TopK -> TopP -> softmax -> multinomial of one processed fp32 row (sample.hip); ws >= cogs_k_sample_ws() bytes
size_t cogs_k_sample_ws();
int cogs_k_sample(hipStream_t st, const float* logits, int n, float temperature, int top_k, double top_p, const float* draws, uint64_t seed,
                  uint64_t offset, int64_t* out_token, int32_t* kept_idx, float* kept_prob, int* n_kept, int kept_cap,
                  void* ws);
