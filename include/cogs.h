/*
 * cogs.h -- C ABI of libcogs_hip.so: the MI355X (gfx950) hot path of the CogReasoner
 * streaming-VQA inference pipeline (reference: LiamZhao326/CogStream, model/ package).
 *
 * The reference has no native ABI: its hot path is Python calling torch / transformers /
 * flash-attn. This header is the boundary a maintainer binds with
 * ctypes from the three reference seams
 *     vision_encoder(pixel_values, grid_sizes, merge_sizes)   model/cogreasoner_chat.py:270-274
 *     kmeans_with_time_min_max(features, timestamps, K)        model/cogreasoner_chat.py:283
 *     get_model()(inputs_embeds=..) / super().generate(..)     model/cogreasoner_chat.py:312-316,802
 * (the stub is shown in INTEGRATION.md; cogstream_amd/_lib.py is the binding this repo uses).
 *
 * Rules that hold for every entry point:
 *   - extern "C", plain pointers and sizes; returns a cogs_status (0 = ok, negative = error);
 *     never throws, never synchronises the device unless stated, never allocates
 *     caller-visible memory (workspace sizes are queried, then passed in);
 *   - all tensor pointers are DEVICE pointers owned by the caller (e.g. torch tensors'
 *     data_ptr()); small shape arrays marked "host" are host pointers read during the call;
 *   - `stream` is a hipStream_t; work is enqueued on it in call order;
 *   - `dtype` is the storage + arithmetic input type of activations/weights:
 *     COGS_DT_BF16 (production; fp32 accumulation) or COGS_DT_F32 (parity mode, exact-f32 MFMA);
 *   - a handle is bound to one device; calls on one handle are not thread-safe. Calls may be queued on a stream
 *     back to back without synchronising: every device table a call needs is built by kernels on that stream or
 *     travels through a pinned, event-guarded staging ring of the handle, and the workspace passed to a call is
 *     only touched by work queued on the call's stream (so one workspace = one stream at a time).
 */
#ifndef COGS_H_
#define COGS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int cogs_status;
#define COGS_OK 0
#define COGS_E_INVALID (-1)     /* bad argument / unsupported shape alignment */
#define COGS_E_HIP (-2)         /* a HIP runtime call or kernel launch failed */
#define COGS_E_UNSUPPORTED (-3)
#define COGS_E_WORKSPACE (-4)   /* workspace missing or too small */

#define COGS_DT_BF16 0
#define COGS_DT_F32 1

#define COGS_ACT_NONE 0
#define COGS_ACT_GELU_TANH 1    /* ACT2FN["gelu_pytorch_tanh"], ViT fc1 */
#define COGS_ACT_GELU_ERF 2     /* nn.GELU(), projector */
#define COGS_ACT_SWIGLU 3       /* silu(gate)*up on (gate_i, up_i)-interleaved weight rows */

/* ViT attention semantics (SURVEY.md headline fact 2) */
#define COGS_ATTN_BLOCK_DIAG 0        /* per-frame attention = flash_attn_varlen path (:309-312) */
#define COGS_ATTN_REF_EAGER_GLOBAL 1  /* eager path: global attention, +1.0 logit on same-frame pairs (:257-266) */

typedef struct cogs_ctx* cogs_handle;
typedef void* cogs_stream; /* hipStream_t */

const char* cogs_status_string(cogs_status s);
const char* cogs_version(void);

cogs_status cogs_create(int device, cogs_handle* out);
cogs_status cogs_destroy(cogs_handle h);   /* NULL is accepted (no-op) */

/* Optional per-kernel-class timing of the composite calls below (HIP events on `stream`, recorded
 * around every launch between begin and end). Used by bench.py for the roofline figure; adds a few
 * microseconds per launch, so it is never on inside a throughput measurement. */
#define COGS_PROF_GEMM 0
#define COGS_PROF_ATTN 1
#define COGS_PROF_NORM 2
#define COGS_PROF_OTHER 3
#define COGS_PROF_CLASSES 4
cogs_status cogs_profile_begin(cogs_handle h);
/* synchronises `stream`; ms_per_class / launches_per_class: host arrays [COGS_PROF_CLASSES]; launches are KERNELS
 * (a GEMM that the library splits in two counts twice), so ms / launches is comparable with rocprofv3's kernel stats */
cogs_status cogs_profile_end(cogs_handle h, cogs_stream stream, float* ms_per_class, int* launches_per_class);

/* Diagnostics. The library reads NOTHING from the environment: its A/B switches (which kernel body a shape gets, timing-
 * only modes, prints) live in one table whose defaults are the shipped behaviour (cogstream_amd/csrc/debug.h lists them;
 * cogs_debug_list() returns the same text: "name = default: meaning" per line). bench.py --debug, tools/ and the
 * bit-identity tests flip a switch by name, run, and flip it back inside one process. Process-wide and unsynchronised:
 * set between calls. Unknown names return COGS_E_INVALID. cogs_debug_get also reports "gemm_last_body". */
cogs_status cogs_debug_set(const char* name, int64_t value);
cogs_status cogs_debug_get(const char* name, int64_t* value);
const char* cogs_debug_list(void);

/* ------------------------------------------------------------------ operator level ---- */

/* DESCRIPTORS (cogs_gemm_desc, cogs_attn_desc, ...) MUST BE ZERO-INITIALISED before the caller fills in what it uses
 * (`cogs_gemm_desc d = {0};` / memset / ctypes' default constructor): they carry no size or version member, new optional
 * fields are appended at the end with 0 = "off" (round 5: cogs_gemm_desc.hm_rows / hm_cols, cogs_attn_desc.head_stride), and a
 * field left holding stack garbage switches a feature on. cogs_version() names the ABI revision: "0.2" added those three. */

/* C[M,N] = epilogue(A[M,K] . W[N,K]^T). nn.Linear / Conv2d(k=s=14) replacement
 * (model/modeling_videollama3_encoder.py:194-210,246-248,275,369-373; cogreasoner_chat.py:179-211).
 * K must be a multiple of 64 (bf16) / 32 (f32) -- pad with zeros; N a multiple of 4;
 * lda/ldw/ldc/ldr in elements, rows 16-byte aligned. bias [N] and residual [M,N] nullable.
 * rope: if rope_cos != NULL, columns [0, rope_cols) are rotated with tables [M, head_dim/2] (fp32), or --
 * rope_sin == NULL -- with ONE interleaved table [M, head_dim/2, 2] = (cos, sin) in rope_cos (faster);
 * weight rows must be packed so that columns (2i, 2i+1) of a head are the rotate_half pair
 * (i, i + head_dim/2). act = COGS_ACT_*; SWIGLU writes [M, N/2]. out_f32: store fp32. */
typedef struct {
    int dtype;
    const void* A; int64_t lda;
    const void* W; int64_t ldw;
    void* C; int64_t ldc;
    const void* bias;
    const void* residual; int64_t ldr;
    int M, N, K;
    int act;
    int out_f32;
    const float* rope_cos;
    const float* rope_sin;
    int rope_cols;
    int head_dim;
    /* optional 2-D rotary position LUT (ViT): rope_lut [rope_maxpos][head_dim/4][2] fp32 = (cos, sin)(pos * inv_freq[f]),
     * rope_rowpos [M] int32 = h | w << 16; the first half of a head's pairs rotates with h, the second with w. Large
     * bf16 GEMMs then read the factors from an LDS copy of the LUT; rope_cos must still hold the matching interleaved
     * per-row table (rope_sin NULL) for the tiles that do not take that path. */
    const float* rope_lut;
    const int32_t* rope_rowpos;
    int rope_maxpos;
    /* LayerNorm fused around the GEMM (nn.LayerNorm of modeling_videollama3_encoder.py:382-391; bf16/f32 alike):
     * row_stats != NULL: the GEMM also writes, per output row and 64-column block, (sum, sum of squares) of the values
     *   it stores (after the rounding to `dtype`): row_stats [M][N/64][2] fp32 (N % 64 == 0, no SWIGLU / fp32 output); cogs_ln_finalize turns them into
     *   ln_ab [M][2] = (rstd, -rstd * mean).
     * ln_ab != NULL: the GEMM computes LN(A) . W0^T + bias0 WITHOUT a normalised copy of A. What the kernels evaluate
     *   is exactly  y[r][n] = ln_ab[r][0] * acc[r][n] + col_c[n]  (before rope / activation; bias must be NULL), where
     *   acc = A . W^T on the UN-normalised rows of A. That equals LN(A) . W0^T + bias0 if and only if the caller passes
     *     W [n][k]  = W0[n][k] * gamma[k] - mean_k(W0[n][:] * gamma)      (every row of W sums to ZERO over k: the
     *                 mean term  -rstd * mean_r * sum_k W[n][k]  of the LayerNorm then vanishes and is NOT evaluated;
     *                 ln_ab[r][1] is not read), stored in `dtype`;
     *     col_c [n] = bias0[n] + sum_k W0[n][k] * beta[k]                 (fp32 [N]).
     *   A W whose rows do not sum to zero gives wrong results without any error. The rounding of W to bf16 leaves a
     *   row sum s_n != 0, i.e. an error rstd * mean * s_n that grows with |mean / std| of a row of A;
     *   cogstream_amd.weights.fold_layernorm removes it by moving a handful of weights per row one bf16 step so that
     *   the ROUNDED row sums to zero too (|s_n| < one step of the row's smallest weights). The four specialised
     *   epilogues that exist: {LNFOLD, LNFOLD + rope, LNFOLD + GELU_TANH} with M >= 2; anything else (M == 1,
     *   residual / GELU_ERF / SWIGLU / fp32 output together with ln_ab or row_stats) returns COGS_E_UNSUPPORTED. */
    float* row_stats;
    const float* ln_ab;
    const float* col_c;
    /* Head-major output (bf16, M >= 2, no activation / residual / row_stats / fp32 output; what the encoder's QKV GEMM uses):
     * hm_rows > 0 -> the N columns are N / hm_cols blocks (q | k | v) of hm_cols / head_dim heads each, and element (row m,
     * column n) is stored at  C[((n / hm_cols) * (hm_cols / head_dim) + (n % hm_cols) / head_dim) * hm_rows * head_dim
     * + m * head_dim + n % head_dim]  -- [block][head][row][head_dim], every head's rows contiguous (ldc is ignored; M <=
     * hm_rows; head_dim % 8 == 0; hm_cols % head_dim == 0; N % hm_cols == 0; hm_rows * hm_cols * 2 < 2^32).
     * cogs_attn_desc.head_stride reads that layout. 0 = row-major C[m][n]. */
    int64_t hm_rows;
    int hm_cols;
} cogs_gemm_desc;
cogs_status cogs_gemm(cogs_stream stream, const cogs_gemm_desc* d);
/* (rstd, -rstd * mean) per row from the row_stats partials of a GEMM whose N is the LayerNorm width H */
cogs_status cogs_ln_finalize(cogs_stream stream, const float* row_stats, int rows, int H, float eps, float* ln_ab);

/* Attention over token-major Q/K/V (row stride ld*, head h at column h*head_dim).
 * cu_seqlens (device int32 [nseg+1], nullable): block-diagonal segments as in
 * flash_attn_varlen_func (modeling_videollama3_encoder.py:309-312, cu_seqlens :439-440).
 * row_lo/row_hi (device int32 [q_len], nullable): eager-reference mode, every query sees all
 * keys and `bias` is added to the logits of keys in [row_lo[q], row_hi[q]) (:257-266).
 * causal: key j visible to query i iff j <= i + q_pos0 (Qwen2; q_pos0 = kv_len - q_len).
 * nsplit > 1 splits the keys over workgroups (decode); ws >= nsplit*q_len*hq*(head_dim+2)*4 bytes. */
typedef struct {
    int dtype;
    const void* Q; const void* K; const void* V; void* O;
    int64_t ldq, ldk, ldv, ldo;
    const int32_t* cu_seqlens; int nseg; int max_seqlen;
    const int32_t* row_lo; const int32_t* row_hi; float bias;
    int q_len, kv_len;
    int hq, hkv, head_dim;
    float scale;
    int causal; int q_pos0;
    int force_rowwise;      /* use the generic fp32-math kernel (any head_dim <= 256) */
    int nsplit; void* ws; size_t ws_bytes;
    int q_prescaled;        /* Q already carries scale*log2(e) (`scale` is then ignored): lets the bf16 kernels run the
                             * softmax without a per-score multiply; not with row_lo/row_hi */
    int64_t head_stride;    /* 0: token-major Q / K / V as described above (head h at column h*head_dim of a row). > 0: HEAD-major
                             * Q, K and V -- head h starts head_stride elements after head h-1 and holds its rows back to back
                             * (ldq = ldk = ldv = head_dim): the layout cogs_gemm_desc.hm_rows writes (head_stride = hm_rows *
                             * head_dim). Only the encoder's production shape takes it (bf16, head_dim 72, cu_seqlens, pre-scaled
                             * Q, hq == hkv, no mask modes); anything else returns COGS_E_UNSUPPORTED. O stays token-major. */
} cogs_attn_desc;
cogs_status cogs_attention(cogs_stream stream, const cogs_attn_desc* d);

/* nn.LayerNorm (modeling_videollama3_encoder.py:382-384,475) / Qwen2RMSNorm; H % 8 == 0 */
cogs_status cogs_layernorm(cogs_stream stream, int dtype, const void* x, void* y, const void* gamma,
                           const void* beta, int rows, int H, float eps);
cogs_status cogs_rmsnorm(cogs_stream stream, int dtype, const void* x, void* y, const void* gamma, int rows,
                         int H, float eps);
/* y[r] = mean_{i<group} LayerNorm(x[group*r+i]): post_layernorm + 2x2 bilinear merge (:482-501) */
cogs_status cogs_ln_merge(cogs_stream stream, int dtype, const void* x, void* y, const void* gamma,
                          const void* beta, int out_rows, int group, int H, float eps);

/* Pixel-difference keep-mask of merged visual tokens (model/cogreasoner_chat.py:405-422).
 * pix: one video's pixel_values viewed as [t, P, E] (E = merge^2 * 588); mask: uint8 [t*P].
 * minor (device uint8 [t], nullable): frames reduced to their token 0. */
cogs_status cogs_pixdiff_mask(cogs_stream stream, int dtype, const void* pix, int t, int P, int E, float thr,
                              int min_tokens, const uint8_t* minor, uint8_t* mask);
/* token 0 of each listed frame <- mean over the frame's P tokens, in place (cogreasoner_chat.py:434-447) */
cogs_status cogs_frame_mean_to_slot0(cogs_stream stream, int dtype, void* feats, int P, int D,
                                     const int32_t* frames, int n_frames);
/* out[r] = idx[r] >= 0 ? table_a[idx[r]] : table_b[-idx[r]-1]   (embed_tokens + masked scatter, :567-572) */
cogs_status cogs_gather_rows(cogs_stream stream, int dtype, const void* table_a, const void* table_b,
                             const int64_t* idx, void* out, int rows, int D);
/* out[D] (fp32) = mean over rows (torch.mean(last_hidden_state, dim=1), :317,323) */
cogs_status cogs_mean_rows(cogs_stream stream, int dtype, const void* x, int64_t ldx, int rows, int D, float* out);
/* out[i] = cos(a, b[i])  (F.cosine_similarity, :325) */
cogs_status cogs_cosine(cogs_stream stream, const float* a, const float* b, int n, int D, float* out);

/* Time-aware k-means steps (model/kmeans_with_time.py:4-137); the host keeps the RNG draws.
 * feats [T, PD] (dtype), centres fp32 [K, PD], ts/centre_ts fp32. Any T and K (the reference has no limit: a
 * 600-frame session clusters into K = 40). Distances are direct sums of (x - c)^2, not torch.cdist's
 * |x|^2 + |c|^2 - 2 x.c: see DESIGN.md section 2 (near-tie study) for where the reference's own rounding decides. */
cogs_status cogs_kmeans_workspace_bytes(int T, int64_t PD, int K, size_t* bytes);
/* dist2[T,K] = squared L2 distance to the centres; centre_rows (device int32 [K], nullable)
 * selects feature rows as centres instead of `centres` (k-means++ init, :46-50) */
cogs_status cogs_kmeans_sqdist(cogs_stream stream, int dtype, const void* feats, int T, int64_t PD,
                               const float* centres, const int32_t* centre_rows, int K, float* dist2,
                               void* ws, size_t ws_bytes);
/* per-row min-max normalised feature + time distance, argmin (:76-104); counts int32 [K] */
cogs_status cogs_kmeans_assign(cogs_stream stream, const float* dist2, const float* ts, const float* centre_ts,
                               int T, int K, float alpha, int64_t* assign, int32_t* counts);
/* centres/centre_ts <- cluster means, empty cluster k <- row reseed_rows[k] (:107-120);
 * shift_out[0] = sum_k ||dc_k||_2 + ||d centre_ts||_2 (:123-125) */
cogs_status cogs_kmeans_update(cogs_stream stream, int dtype, const void* feats, const float* ts, int T,
                               int64_t PD, int K, const int64_t* assign, const int32_t* reseed_rows,
                               float* centres, float* centre_ts, float* shift_out, void* ws, size_t ws_bytes);
/* select_additional_frames (model/cogreasoner_chat.py:50-64): per cluster k, the rows the event selection always keeps --
 * all members if there are at most n_extra (ascending row), else the n_extra members nearest to the centroid
 * (torch.cdist + topk(largest=False): ascending distance; equal distances -> lower row first). dist2 [T][K] fp32 as
 * cogs_kmeans_sqdist leaves it (squared distances order like distances), assign int64 [T]; out_idx int64 [K][n_extra]
 * (-1 padded), out_counts int32 [K] = min(members, n_extra). n_extra <= 8. */
cogs_status cogs_select_near_centroid(cogs_stream stream, const float* dist2, const int64_t* assign, int T, int K, int n_extra,
                                      int64_t* out_idx, int32_t* out_counts);
/* One k-means++ step (:46-60): squared distance of every row to feature row `row`, folded into nearest2 (device fp32 [T]:
 * first != 0 stores, else min). probs_host (pinned host fp32 [T], nullable): the updated nearest2 is copied there and the
 * stream is synchronised, so the caller can draw the next centre (torch.multinomial on the CPU generator) right away.
 * Workspace: cogs_kmeans_workspace_bytes(T, PD, 1). */
cogs_status cogs_kmeans_pp_step(cogs_stream stream, int dtype, const void* feats, int T, int64_t PD, int row, int first,
                                float* nearest2, float* probs_host, void* ws, size_t ws_bytes);
/* The whole k-means++ seeding (:41-62) in ONE call, no host round trip per centre. first_row = the first centre (the
 * caller's random.randint draw); idx (device int32 [K], all written): idx[0] = first_row, entry m the row drawn after m centres:
 * probs = (sqrt(nearest distance^2))^2 normalised by their sum, torch.multinomial(probs, 1) = argmax(probs / q), q ~
 * Exponential(1) (ATen's n_sample == 1 path) -- q_draws (device fp32 [K-1][T]) are those draws, made ahead on the
 * caller's CPU generator, one [T] row per centre, exactly as K - 1 multinomial calls would consume it. *zero_flag (device
 * int32) is set when a step finds every probability zero (the reference's random.randint branch, :57-59): the indices
 * are then not usable and the caller seeds step by step with cogs_kmeans_pp_step instead. nearest2: device fp32 [T]
 * scratch. The sum of the probabilities is taken in fp64 in a fixed order, not in torch's fp32 summation order: it is a
 * common divisor of all ratios. Workspace: cogs_kmeans_workspace_bytes(T, PD, 1). */
cogs_status cogs_kmeans_pp(cogs_stream stream, int dtype, const void* feats, int T, int64_t PD, int K, int first_row,
                           const float* q_draws, int32_t* idx, int32_t* zero_flag, float* nearest2, void* ws, size_t ws_bytes);
/* The whole Lloyd loop (:71-131) from given centres: per iteration distances -> assignment -> means / reseeds -> centre
 * shift, stopping after max_iter iterations or when the shift is <= tol. The library queues up to four iterations
 * between two synchronisations of the stream (it reads eight control words from host-mapped memory); every kernel of an
 * iteration queued behind the converged one returns without touching anything, so centres / assign are those of the
 * iteration the reference stops at. Empty clusters (:116-120: one random.randint(0, T-1) per empty cluster, ascending
 * cluster index, every iteration) take their rows from reseed_pool (HOST int32 [pool_len], pool_len <= 4096: the caller's
 * pre-drawn values, consumed in order, copied into host-mapped memory before the first launch; *reseeds_used reports how many). If an iteration
 * needs more than are left it is NOT committed: the call returns COGS_OK with *pool_exhausted = 1 and *iterations = the
 * completed ones; draw more and call again with the remaining iteration budget. T + K + 1 <= 16 384 (the assignments
 * and the cluster counters are staged in LDS). Workspace: cogs_kmeans_workspace_bytes(T, PD, K). */
cogs_status cogs_kmeans_lloyd(cogs_stream stream, int dtype, const void* feats, const float* ts, int T, int64_t PD, int K,
                              float alpha, int max_iter, float tol, const int32_t* reseed_pool, int pool_len,
                              float* centres, float* centre_ts, int64_t* assign, int* iterations, int* reseeds_used,
                              int* pool_exhausted, void* ws, size_t ws_bytes);
/* How close the assignments of the LAST cogs_kmeans_lloyd call on this workspace were (host outputs; synchronises).
 * Per row and iteration: the relative change of the row's feature distances that would flip the decision between its
 * best and second-best cluster -- with every time term equal this is (d2 - d1) / mean(d1, d2) of the two nearest feature
 * distances, the margin of the near-tie study; a decision the feature distances cannot flip is
 * +inf. *min_margin = the minimum over all rows and iterations, *rows_below = the number of (row, iteration) pairs
 * below 1e-3. Below that margin the reference's own torch.cdist rounding decides the row and no other summation order
 * reproduces it; at or above it the assignment equals the reference's. */
cogs_status cogs_kmeans_margins(cogs_stream stream, int T, int64_t PD, int K, void* ws, size_t ws_bytes, float* min_margin,
                                int32_t* rows_below);
/* dtype conversion / zero-padded row copy (features.to(float32), patch padding) */
cogs_status cogs_pack_rows(cogs_stream stream, int in_dtype, int out_dtype, const void* in, int64_t ld_in,
                           void* out, int64_t ld_out, int rows, int cols_in, int cols_out);

/* GPU pre-processing (image_processing_videollama3.py:235-347): uint8 frames [T,H,W,3] -> Pillow-exact bicubic
 * resize to (th,tw) -> value_table[c][byte] (fp32 [3,256], = rescale+normalize per byte) -> merge-window-major patch rows [T*(th/14)*(tw/14), 588] in out_dtype.
 * bounds_* [out,2] / coef_* [out,ks*] are Pillow's per-output tap tables (device int32; built on the host,
 * cogstream_amd.processing.resample_coeffs). ws >= T*H*tw*3 bytes. */
cogs_status cogs_preprocess_workspace_bytes(int T, int H, int tw, size_t* bytes);
cogs_status cogs_preprocess_frames(cogs_stream stream, const uint8_t* frames, int T, int H, int W, int th, int tw,
                                   int merge, const int32_t* bounds_x, const int32_t* coef_x, int ksx,
                                   const int32_t* bounds_y, const int32_t* coef_y, int ksy, const float* value_table,
                                   void* out, int out_dtype, void* ws, size_t ws_bytes);

/* logits post-processing (model/generation_config.json:2-12; qaselect_module_predict.py:86-103) */
cogs_status cogs_argmax(cogs_stream stream, const float* logits, int n, int64_t* out, void* ws /* >= 512 B */);
cogs_status cogs_logits_process(cogs_stream stream, float* logits, int n, const int64_t* prev, int n_prev,
                                float repetition_penalty, const int32_t* allowed, int n_allowed,
                                float temperature, float* tmp /* >= n_prev floats */);
cogs_status cogs_topk(cogs_stream stream, const float* logits, int n, int top_k, float* topk_val,
                      int32_t* topk_idx, float* ws /* >= n floats */);
/* One sampled token from a PROCESSED fp32 row (after cogs_logits_process's penalty / mask): TemperatureLogitsWarper
 * (scores / temperature, temperature > 0; 1 = off) -> TopKLogitsWarper (top_k <= 0: off; keeps
 * every score >= the k-th largest) -> TopPLogitsWarper (top_p >= 1: off; ascending cumulative probability <= 1 - top_p
 * is dropped, the best token always stays) -> softmax -> torch.multinomial(probs, 1) = argmax_i probs_i / q_i,
 * q ~ Exp(1) (GenerationMixin._sample of transformers 4.46.3 with model/generation_config.json:2-12; call site
 * model/cogreasoner_chat.py:802-807). `draws`: NULL -> q_i from Philox4x32-10 keyed (seed, offset, i), no host round
 * trip; else device float [n] holding the CPU generator's exponential draws of this step (parity with the
 * reference's CPU sampler). Optional kept_idx/kept_prob [kept_cap] + n_kept report the surviving ids and their
 * renormalised probabilities (any order). ws >= cogs_sample_workspace_bytes(). */
size_t cogs_sample_workspace_bytes(void);
cogs_status cogs_sample(cogs_stream stream, const float* logits, int n, float temperature, int top_k, double top_p,
                        const float* draws,
                        uint64_t seed, uint64_t offset, int64_t* out_token, int32_t* kept_idx, float* kept_prob,
                        int32_t* n_kept, int kept_cap, void* ws);

/* ------------------------------------------------------------------ vision encoder ---- */

/* Packed ViT weights (borrowed device pointers, `dtype` elements). HF names (A18):
 * model.vision_encoder.{embeddings.patch_embedding, encoder.layers.N.*, post_layernorm}.
 * Packing done once at load (cogstream_amd/weights.py): patch_w [hidden, patch_pad] zero padded;
 * qkv_w [3*hidden, hidden] = q,k,v rows stacked, q/k rows of each head interleaved as rotary pairs;
 * fc1_w [inter_pad, hidden], fc1_b [inter_pad], fc2_w [hidden, inter_pad] zero padded.
 * LayerNorm folding (bf16 production path; both pointers non-NULL and hidden % 64 == 0): qkv_w / fc1_w then hold
 * the rows of W * diag(ln_gamma) with their mean over k removed -- every row sums to zero, see cogs_gemm_desc.ln_ab for
 * the exact contract -- and qkv_c / fc1_c [rows] fp32 = bias + W . ln_beta; the encoder then never materialises LN(x):
 * the GEMMs that write the residual stream emit per-row statistics and the QKV / fc1 GEMMs apply them in their epilogue.
 * With the pointers NULL (fp32 parity mode) qkv_w / fc1_w are the plain weights and LayerNorm runs as its own kernel. */
typedef struct {
    const void *ln1_g, *ln1_b, *qkv_w, *qkv_b, *o_w, *o_b, *ln2_g, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
    const float *qkv_c, *fc1_c;
} cogs_vit_layer;
typedef struct {
    int dtype;
    int hidden, inter_pad, layers, heads, patch_dim, patch_pad;
    float ln_eps;
    const void *patch_w, *patch_b, *post_ln_g, *post_ln_b;
    const cogs_vit_layer* layer; /* host array [layers], copied by cogs_vit_load */
} cogs_vit_weights;
cogs_status cogs_vit_load(cogs_handle h, const cogs_vit_weights* w);
cogs_status cogs_vit_workspace_bytes(cogs_handle h, int64_t n_patches, size_t* bytes);
/* Videollama3VisionEncoderModel.forward (modeling_videollama3_encoder.py:479-510).
 * pixel_values [N,588] (pix_dtype fp32 or bf16), grid_sizes host int64 [V,3] = (t,gh,gw),
 * merge_sizes host int64 [V]; out_tokens [M, hidden] in the weights' dtype. */
cogs_status cogs_vit_encode(cogs_handle h, cogs_stream stream, const void* pixel_values, int pix_dtype,
                            const int64_t* grid_sizes, const int64_t* merge_sizes, int V, int attn_mode,
                            void* out_tokens, void* ws, size_t ws_bytes);
/* cogs_vit_encode runs a clip of two or more frames as `streams` contiguous frame ranges on as many streams (the caller's
 * and up to three the handle owns; frames are independent under per-frame attention, the tokens are bit-identical to a
 * one-stream encode): one range's kernels fill the partly empty last rounds of another's persistent GEMM launches. The
 * ranges are queued layer by layer in turn, so every stream starts within one layer of the first. streams = 1 keeps
 * everything on the caller's stream, 2 is the default, 3 and 4 are allowed. */
cogs_status cogs_vit_set_streams(cogs_handle h, int streams);

/* ----------------------------------------------------------------------- multi-GPU ------ */

/* The one data-path collective of the path (the reference never shards a clip, its only
 * distributed code is init + per-video replicas, evaluate/answer_generate.py:155,186-187): frames of a clip are encoded
 * (and projected) by the ranks of one node, and ONE all-gather reassembles the visual tokens in frame order. Every rank
 * contributes `local_rows` rows of `row_bytes` bytes (equal counts: a ragged frame split pads to the largest shard);
 * `global` receives world * local_rows rows in rank order. `nccl_comm` is an ncclComm_t the caller created (rccl.h; one
 * process per GPU); the call enqueues ncclAllGather on `stream`. librccl.so is bound lazily with dlopen -- the library
 * has no link-time dependency on it -- and COGS_E_UNSUPPORTED is returned when it cannot be loaded. In-place use
 * (local == global + rank * local_rows * row_bytes) is allowed, as for ncclAllGather. The Python host does the same
 * through torch.distributed (cogstream_amd/parallel.py), whose "nccl" backend is RCCL. */
cogs_status cogs_allgather_tokens(cogs_stream stream, void* nccl_comm, const void* local, size_t local_rows,
                                  size_t row_bytes, void* global);

/* MlpGeluProjector (model/cogreasoner_chat.py:199-211): Linear -> GELU(erf) -> Linear */
typedef struct {
    int dtype; int in_dim, out_dim;
    const void *w1, *b1, *w2, *b2;
} cogs_proj_weights;
cogs_status cogs_proj_load(cogs_handle h, const cogs_proj_weights* w);
cogs_status cogs_project(cogs_handle h, cogs_stream stream, const void* tokens, int M, void* out, void* ws,
                         size_t ws_bytes /* >= M*out_dim elements */);
/* encode_images (model/cogreasoner_chat.py:264-276: vision_encoder, then mm_projector) as ONE call: cogs_vit_encode with the
 * projector's two GEMMs of every frame range queued behind that range's encoder on ITS stream, instead of on the caller's
 * stream after the join (round 6: the ranges' projections overlap the other ranges' last layers). out_tokens [M, hidden] is
 * still written (callers gather / cache the encoder tokens); proj_out [M, out_dim]; proj_ws >= M * out_dim elements.
 * Results are the bits of cogs_vit_encode followed by cogs_project. */
cogs_status cogs_vit_encode_project(cogs_handle h, cogs_stream stream, const void* pixel_values, int pix_dtype,
                                    const int64_t* grid_sizes, const int64_t* merge_sizes, int V, int attn_mode,
                                    void* out_tokens, void* ws, size_t ws_bytes, void* proj_out, void* proj_ws,
                                    size_t proj_ws_bytes);

/* ------------------------------------------------------------------------- Qwen2 ------ */

/* Packed Qwen2 weights. HF names: model.layers.N.{input_layernorm, self_attn.{q,k,v,o}_proj,
 * post_attention_layernorm, mlp.{gate,up,down}_proj}, model.norm, lm_head (untied).
 * qkv_w [(hq+2*hkv)*hd, hidden] with q/k head rows interleaved as rotary pairs, qkv_b likewise;
 * gu_w [2*inter, hidden] rows interleaved (gate_i, up_i); down_w [hidden, inter];
 * lm_head [vocab, hidden] (vocab % 4 == 0). */
typedef struct {
    const void *in_ln, *qkv_w, *qkv_b, *o_w, *post_ln, *gu_w, *down_w;
} cogs_llm_layer;
typedef struct {
    int dtype;
    int hidden, inter, layers, heads, kv_heads, head_dim, vocab;
    float rms_eps, rope_theta;
    const void *final_norm, *lm_head;
    const cogs_llm_layer* layer; /* host array [layers] */
} cogs_llm_weights;
cogs_status cogs_llm_load(cogs_handle h, const cogs_llm_weights* w);

/* KV cache, caller-owned: k and v are [layers][max_len][kv_heads*head_dim] in the weights' dtype */
typedef struct {
    void* k; void* v;
    int max_len;
    int len;      /* tokens already cached; advanced by cogs_llm_forward */
} cogs_kv;
cogs_status cogs_llm_workspace_bytes(cogs_handle h, int max_tokens, int max_context, size_t* bytes);
/* One Qwen2Model forward over S new tokens given as embeddings [S, hidden] at positions
 * kv->len .. kv->len+S-1 (causal over cache + new tokens); kv == NULL: stateless forward
 * (event summaries, cogreasoner_chat.py:303-316). Outputs, each nullable:
 *   last_logits  fp32 [vocab]   lm_head(norm(h_last))          (generate / decode)
 *   pooled_mean  fp32 [hidden]  mean_t norm(h_t)                (cogreasoner_chat.py:317,323)
 *   hidden_out   [S, hidden]    norm(h)  (last_hidden_state) */
cogs_status cogs_llm_forward(cogs_handle h, cogs_stream stream, const void* embeds, int S, cogs_kv* kv,
                             float* last_logits, float* pooled_mean, void* hidden_out, void* ws,
                             size_t ws_bytes);

/* nseg independent sequences stored back to back in embeds [S,H] (cu_seqlens_host [nseg+1], host memory, 0 .. S):
 * one stateless forward -- positions and causal attention restart at every sequence -- and the mean over each
 * sequence of the final hidden states -> pooled_means [nseg, H] fp32. Replaces the K sequential
 * self.get_model()(inputs_embeds=...).last_hidden_state.mean(dim=1) calls of select_events_based_on_summary
 * (model/cogreasoner_chat.py:303-322: one per event, one for the question). Synchronises the stream once (host
 * staging of the position table). Workspace: cogs_llm_workspace_bytes(h, S, S). */
cogs_status cogs_llm_forward_segments(cogs_handle h, cogs_stream stream, const void* embeds, int S,
                                      const int32_t* cu_seqlens_host, int nseg, float* pooled_means, void* ws,
                                      size_t ws_bytes);

#ifdef __cplusplus
}
#endif
#endif /* COGS_H_ */
