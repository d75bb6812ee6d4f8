// C-ABI layer of libcogs_hip.so (see include/cogs.h): argument checking, workspace carving and
// the per-model launch sequences (ViT encoder, projector, Qwen2 forward). No torch types, no
// exceptions, no stream synchronisation (host tables travel through a pinned, event-guarded ring).
#include "../../include/cogs.h"
#include "common.h"
#include "kernels.h"
#include "debug.h"

#include <math.h>
#include <stdint.h>
#include <new>
#include <stdlib.h>
#include <dlfcn.h>
#include <atomic>
#include <string.h>
#include <vector>

CogsDebug g_cogs_debug;

struct cogs_ctx {
    int device = 0;
    // ViT
    bool vit_ok = false;
    cogs_vit_weights vit{};
    std::vector<cogs_vit_layer> vit_layers;
    float* vit_inv_freq = nullptr;  // device [head_dim/4]
    int vit_nfreq = 0;
    // projector
    bool proj_ok = false;
    cogs_proj_weights proj{};
    // LLM
    bool llm_ok = false;
    cogs_llm_weights llm{};
    std::vector<cogs_llm_layer> llm_layers;
    float* llm_inv_freq = nullptr;  // device [head_dim/2]
    int llm_nfreq = 0; float llm_theta = 0.f;
    // pinned host staging for small host->device tables (segment boundaries): a ring of slots, each guarded by an
    // event recorded behind the copy that reads it, so a slot is never rewritten while a queued copy still needs it
    static constexpr int STAGE_SLOTS = 4;
    void* stage_host[STAGE_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    size_t stage_cap[STAGE_SLOTS] = {0, 0, 0, 0};
    hipEvent_t stage_ev[STAGE_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    bool stage_busy[STAGE_SLOTS] = {false, false, false, false};
    int stage_next = 0;
    // second stream of the frame-split encode (cogs_vit_encode): two halves of a small clip run side by side
    std::vector<hipStream_t> aux_streams;
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_joins;
    int vit_streams = 2;                  // cogs_vit_set_streams: 1 = never split
    // optional per-kernel-class event profiling (bench/roofline only)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;      // pairs
    std::vector<int> prof_cls;
    std::vector<int> prof_n;              // kernels inside the bracket
    size_t prof_used = 0;
};

// RAII event bracket around one launch; a no-op unless cogs_profile_begin() was called
struct ProfScope {
    cogs_ctx* c; hipStream_t st; size_t slot = 0; bool on; long launches0 = 0;
    ProfScope(cogs_ctx* c_, hipStream_t st_, int cls) : c(c_), st(st_), on(c_->prof_on) {
        if (!on) return;
        launches0 = cogs_k_gemm_launch_count();
        if (c->prof_used * 2 + 2 > c->prof_ev.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            c->prof_ev.push_back(a); c->prof_ev.push_back(b);
        }
        slot = c->prof_used++;
        if (c->prof_cls.size() < c->prof_used) c->prof_cls.resize(c->prof_used);
        c->prof_cls[slot] = cls;
        (void)hipEventRecord(c->prof_ev[2 * slot], st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(c->prof_ev[2 * slot + 1], st);
        const long n = cogs_k_gemm_launch_count() - launches0;   // a GEMM bracket may hold two kernels (split)
        if (c->prof_n.size() < c->prof_used) c->prof_n.resize(c->prof_used);
        c->prof_n[slot] = n > 0 ? (int)n : 1;
    }
};
#define PROF(cls) ProfScope _prof_scope(h, st, cls)

// copy `bytes` of host data to the device on `st` through the handle's pinned ring (no stream synchronisation)
static int stage_h2d(cogs_ctx* c, hipStream_t st, void* dst, const void* src, size_t bytes) {
    const int s = c->stage_next;
    c->stage_next = (s + 1) % cogs_ctx::STAGE_SLOTS;
    if (c->stage_busy[s]) {
        if (hipEventSynchronize(c->stage_ev[s]) != hipSuccess) return COGS_E_HIP;
        c->stage_busy[s] = false;
    }
    if (c->stage_cap[s] < bytes) {
        if (c->stage_host[s]) (void)hipHostFree(c->stage_host[s]);
        c->stage_host[s] = nullptr; c->stage_cap[s] = 0;
        const size_t cap = bytes < 4096 ? 4096 : bytes * 2;
        if (hipHostMalloc(&c->stage_host[s], cap, hipHostMallocDefault) != hipSuccess) return COGS_E_HIP;
        c->stage_cap[s] = cap;
    }
    if (!c->stage_ev[s] && hipEventCreateWithFlags(&c->stage_ev[s], hipEventDisableTiming) != hipSuccess) return COGS_E_HIP;
    memcpy(c->stage_host[s], src, bytes);
    if (hipMemcpyAsync(dst, c->stage_host[s], bytes, hipMemcpyHostToDevice, st) != hipSuccess) return COGS_E_HIP;
    if (hipEventRecord(c->stage_ev[s], st) != hipSuccess) return COGS_E_HIP;
    c->stage_busy[s] = true;
    return COGS_OK;
}

namespace {

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
inline int esize(int dtype) { return dtype == COGS_DT_BF16 ? 2 : 4; }

struct Carver {
    char* base; size_t off = 0; size_t cap;
    Carver(void* p, size_t c) : base((char*)p), cap(c) {}
    void* take(size_t bytes) {
        const size_t o = off;
        off = align_up(off + bytes);
        return base ? base + o : nullptr;
    }
};

#define COGS_TRY(x)                      \
    do {                                 \
        const int _rc = (x);             \
        if (_rc != COGS_OK) return _rc;  \
    } while (0)

CogsGemm to_gemm(const cogs_gemm_desc* d) {
    CogsGemm g;
    g.dtype = d->dtype;
    g.A = d->A; g.lda = d->lda; g.W = d->W; g.ldw = d->ldw; g.C = d->C; g.ldc = d->ldc;
    g.bias = d->bias; g.residual = d->residual; g.ldr = d->ldr;
    g.M = d->M; g.N = d->N; g.K = d->K; g.act = d->act; g.out_f32 = d->out_f32;
    g.rope_cos = d->rope_cos; g.rope_sin = d->rope_sin; g.rope_cols = d->rope_cols; g.head_dim = d->head_dim;
    g.rope_lut = d->rope_lut; g.rope_rowpos = d->rope_rowpos; g.rope_maxpos = d->rope_maxpos;
    g.row_stats = d->row_stats; g.ln_ab = d->ln_ab; g.col_c = d->col_c;
    g.hm_rows = d->hm_rows; g.hm_cols = d->hm_cols;
    return g;
}

}  // namespace

extern "C" {

const char* cogs_status_string(cogs_status s) {
    switch (s) {
        case COGS_OK: return "ok";
        case COGS_E_INVALID: return "invalid argument";
        case COGS_E_HIP: return "HIP runtime error";
        case COGS_E_UNSUPPORTED: return "unsupported configuration";
        case COGS_E_WORKSPACE: return "workspace missing or too small";
        default: return "unknown status";
    }
}

const char* cogs_version(void) { return "cogstream_amd 0.2 (gfx950)"; }     // 0.2: cogs_gemm_desc.hm_rows / hm_cols, cogs_attn_desc.head_stride

// ---------------------------------------------------------------- diagnostics (csrc/debug.h)
// accepted range of a switch: on / off unless listed here
static bool cogs_debug_value_ok(const char* name, int64_t v) {
    struct Range { const char* name; int64_t lo, hi; };
    static const Range ranges[] = {
        {"gemm_wgs", 0, 256}, {"gemm_pad_pct", 100, 400}, {"gemm_group_m", 0, 64}, {"gemm_co_streams", 0, 8},
        {"gemm_ring_cost_permille", 1, 4000}, {"gemv_small_n", 0, 1 << 30}, {"attn_vit", 0, 2}, {"attn_uniform_hint", 0, 1 << 20}, {"attn_prio", 0, 2},
        {"attn_nq", 0, 2}, {"vit_split_max", 0, INT64_MAX}, {"llm_split_keys", 0, 1 << 24}, {"km_row_groups", 0, 1 << 16},
    };
    for (const Range& r : ranges)
        if (strcmp(name, r.name) == 0) return v >= r.lo && v <= r.hi;
    return v == 0 || v == 1;
}

cogs_status cogs_debug_set(const char* name, int64_t value) {
    if (!name) return COGS_E_INVALID;
    if (!cogs_debug_value_ok(name, value)) return COGS_E_INVALID;
    if (value != 0 && (strcmp(name, "gemm_nostore") == 0 || strcmp(name, "gemm_trace") == 0))
        fprintf(stderr, "[cogs] debug switch %s = %lld: a timing / tracing mode -- cogs_gemm %s until it is set back to 0\n", name,
                (long long)value, strcmp(name, "gemm_nostore") == 0 ? "does NOT write its result" : "synchronises and prints per-tile stamps");
#define COGS_DBG_SET(n, dflt, doc) if (strcmp(name, #n) == 0) { g_cogs_debug.n = value; return COGS_OK; }
    COGS_DEBUG_SWITCHES(COGS_DBG_SET)
#undef COGS_DBG_SET
    return COGS_E_INVALID;
}

cogs_status cogs_debug_get(const char* name, int64_t* value) {
    if (!name || !value) return COGS_E_INVALID;
#define COGS_DBG_GET(n, dflt, doc) if (strcmp(name, #n) == 0) { *value = g_cogs_debug.n; return COGS_OK; }
    COGS_DEBUG_SWITCHES(COGS_DBG_GET)
#undef COGS_DBG_GET
    if (strcmp(name, "gemm_last_body") == 0) { *value = g_cogs_debug.gemm_last_body; return COGS_OK; }
    if (strcmp(name, "attn_last_kernel") == 0) { *value = g_cogs_debug.attn_last_kernel; return COGS_OK; }
    if (strcmp(name, "attn_vit_last_end") == 0) { *value = g_cogs_debug.attn_vit_last_end; return COGS_OK; }
    return COGS_E_INVALID;
}

const char* cogs_debug_list(void) {
    return
#define COGS_DBG_DOC(n, dflt, doc) #n " = " #dflt ": " doc "\n"
        COGS_DEBUG_SWITCHES(COGS_DBG_DOC)
#undef COGS_DBG_DOC
        "gemm_last_body (read only): body the last cogs_gemm dispatched to -- 1 128x128, 2 256x128 ring, 3 K-tile ping-pong, "
        "4 whole-line ping-pong, 5 ping-pong + ring (split launch), 6 GEMV\n"
        "attn_last_kernel (read only): kernel of the last cogs_attention -- 1 general MFMA, 2 ViT unpipelined, 3 ViT pipelined, "
        "4 single-token decode, 5 prompt LDS-DMA, 7 row-wise fp32, 8 ViT pipelined on head-major K/V\n"
        "attn_vit_last_end (read only): ragged end of the last pipelined ViT attention launch -- 10 x tiles behind the four-tile loop + "
        "32-key blocks of the last tile (41 .. 72: compile-time shape), 0 the run-time form\n";
}

cogs_status cogs_create(int device, cogs_handle* out) {
    if (!out) return COGS_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return COGS_E_HIP;
    cogs_ctx* c = new (std::nothrow) cogs_ctx();
    if (!c) return COGS_E_HIP;
    c->device = device;
    *out = c;
    return COGS_OK;
}

cogs_status cogs_profile_begin(cogs_handle h) {
    if (!h) return COGS_E_INVALID;
    h->prof_on = true;
    h->prof_used = 0;
    return COGS_OK;
}

cogs_status cogs_profile_end(cogs_handle h, cogs_stream stream, float* ms_per_class, int* launches_per_class) {
    if (!h || !ms_per_class || !launches_per_class) return COGS_E_INVALID;
    h->prof_on = false;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return COGS_E_HIP;
    for (int i = 0; i < COGS_PROF_CLASSES; ++i) { ms_per_class[i] = 0.f; launches_per_class[i] = 0; }
    for (size_t i = 0; i < h->prof_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->prof_ev[2 * i], h->prof_ev[2 * i + 1]) != hipSuccess) return COGS_E_HIP;
        const int cls = h->prof_cls[i];
        if (cls >= 0 && cls < COGS_PROF_CLASSES) { ms_per_class[cls] += ms; launches_per_class[cls] += i < h->prof_n.size() ? h->prof_n[i] : 1; }
    }
    h->prof_used = 0;
    return COGS_OK;
}

cogs_status cogs_destroy(cogs_handle h) {
    if (!h) return COGS_OK;
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    for (int i = 0; i < cogs_ctx::STAGE_SLOTS; ++i) {
        if (h->stage_ev[i]) { (void)hipEventSynchronize(h->stage_ev[i]); (void)hipEventDestroy(h->stage_ev[i]); }
        if (h->stage_host[i]) (void)hipHostFree(h->stage_host[i]);
    }
    if (h->vit_inv_freq) (void)hipFree(h->vit_inv_freq);
    if (h->llm_inv_freq) (void)hipFree(h->llm_inv_freq);
    for (hipStream_t s2 : h->aux_streams) { (void)hipStreamSynchronize(s2); (void)hipStreamDestroy(s2); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (hipEvent_t e : h->ev_joins) (void)hipEventDestroy(e);
    delete h;
    return COGS_OK;
}

// ---------------------------------------------------------------- operator level

cogs_status cogs_gemm(cogs_stream stream, const cogs_gemm_desc* d) {
    if (!d || !d->A || !d->W || !d->C) return COGS_E_INVALID;
    if (d->ln_ab && d->bias) return COGS_E_INVALID;     // col_c already contains the bias
    return cogs_k_gemm((hipStream_t)stream, to_gemm(d));
}

cogs_status cogs_ln_finalize(cogs_stream stream, const float* row_stats, int rows, int H, float eps, float* ln_ab) {
    if (!row_stats || !ln_ab || rows <= 0 || H <= 0 || H % 64) return COGS_E_INVALID;
    return cogs_k_ln_finalize((hipStream_t)stream, row_stats, rows, H / 64, H, eps, ln_ab);
}

cogs_status cogs_attention(cogs_stream stream, const cogs_attn_desc* d) {
    if (!d || !d->Q || !d->K || !d->V || !d->O) return COGS_E_INVALID;
    CogsAttn a;
    a.dtype = d->dtype; a.Q = d->Q; a.K = d->K; a.V = d->V; a.O = d->O;
    a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo;
    a.cu_seqlens = d->cu_seqlens; a.nseg = d->nseg; a.max_seqlen = d->max_seqlen;
    a.row_lo = d->row_lo; a.row_hi = d->row_hi; a.bias = d->bias;
    a.q_len = d->q_len; a.kv_len = d->kv_len; a.hq = d->hq; a.hkv = d->hkv; a.head_dim = d->head_dim;
    a.scale = d->scale; a.causal = d->causal; a.q_pos0 = d->q_pos0; a.force_rowwise = d->force_rowwise;
    a.nsplit = d->nsplit > 1 ? d->nsplit : 1; a.ws = d->ws; a.ws_bytes = d->ws_bytes;
    a.q_prescaled = d->q_prescaled;
    a.head_stride = d->head_stride;
    a.uniform_seqlen = (int)g_cogs_debug.attn_uniform_hint;      // 0 unless a test sets it (checked against q_len / max_seqlen by the kernel's launcher)
    return cogs_k_attention((hipStream_t)stream, a);
}

cogs_status cogs_layernorm(cogs_stream stream, int dtype, const void* x, void* y, const void* gamma, const void* beta,
                           int rows, int H, float eps) {
    return cogs_k_layernorm((hipStream_t)stream, dtype, x, y, gamma, beta, rows, H, eps);
}
cogs_status cogs_rmsnorm(cogs_stream stream, int dtype, const void* x, void* y, const void* gamma, int rows, int H,
                         float eps) {
    return cogs_k_rmsnorm((hipStream_t)stream, dtype, x, y, gamma, rows, H, eps);
}
cogs_status cogs_ln_merge(cogs_stream stream, int dtype, const void* x, void* y, const void* gamma, const void* beta,
                          int out_rows, int group, int H, float eps) {
    return cogs_k_ln_merge((hipStream_t)stream, dtype, x, y, gamma, beta, out_rows, group, H, eps);
}

cogs_status cogs_pixdiff_mask(cogs_stream stream, int dtype, const void* pix, int t, int P, int E, float thr,
                              int min_tokens, const uint8_t* minor, uint8_t* mask) {
    if (!pix || !mask) return COGS_E_INVALID;
    COGS_TRY(cogs_k_pixdiff_mask((hipStream_t)stream, dtype, pix, t, P, E, thr, min_tokens, mask));
    return cogs_k_mask_fix((hipStream_t)stream, mask, t, P, min_tokens, minor);
}
cogs_status cogs_frame_mean_to_slot0(cogs_stream stream, int dtype, void* feats, int P, int D, const int32_t* frames,
                                     int n_frames) {
    return cogs_k_frame_mean_to_slot0((hipStream_t)stream, dtype, feats, P, D, frames, n_frames);
}
cogs_status cogs_gather_rows(cogs_stream stream, int dtype, const void* ta, const void* tb, const int64_t* idx,
                             void* out, int rows, int D) {
    return cogs_k_gather_rows((hipStream_t)stream, dtype, ta, tb, idx, out, rows, D);
}
cogs_status cogs_mean_rows(cogs_stream stream, int dtype, const void* x, int64_t ldx, int rows, int D, float* out) {
    return cogs_k_mean_rows((hipStream_t)stream, dtype, x, ldx, rows, D, out);
}
cogs_status cogs_cosine(cogs_stream stream, const float* a, const float* b, int n, int D, float* out) {
    return cogs_k_cosine((hipStream_t)stream, a, b, n, D, out);
}

cogs_status cogs_kmeans_workspace_bytes(int T, int64_t PD, int K, size_t* bytes) {
    if (!bytes || T <= 0 || PD <= 0 || K <= 0) return COGS_E_INVALID;
    *bytes = align_up(cogs_k_kmeans_ws(T, PD, K, nullptr));
    return COGS_OK;
}
cogs_status cogs_kmeans_sqdist(cogs_stream stream, int dtype, const void* feats, int T, int64_t PD,
                               const float* centres, const int32_t* centre_rows, int K, float* dist2, void* ws,
                               size_t ws_bytes) {
    int ns = 0;
    const size_t need = cogs_k_kmeans_ws(T, PD, K, &ns);
    if (!ws || ws_bytes < need) return COGS_E_WORKSPACE;
    if (!centres && !centre_rows) return COGS_E_INVALID;
    return cogs_k_kmeans_sqdist((hipStream_t)stream, dtype, feats, T, PD, centres, centre_rows, K, (float*)ws, ns, dist2);
}
cogs_status cogs_kmeans_assign(cogs_stream stream, const float* dist2, const float* ts, const float* centre_ts, int T,
                               int K, float alpha, int64_t* assign, int32_t* counts) {
    return cogs_k_kmeans_assign((hipStream_t)stream, dist2, ts, centre_ts, T, K, alpha, assign, counts);
}
cogs_status cogs_kmeans_update(cogs_stream stream, int dtype, const void* feats, const float* ts, int T, int64_t PD,
                               int K, const int64_t* assign, const int32_t* reseed_rows, float* centres,
                               float* centre_ts, float* shift_out, void* ws, size_t ws_bytes) {
    const size_t need = cogs_k_kmeans_ws(T, PD, K, nullptr);
    if (!ws || ws_bytes < need) return COGS_E_WORKSPACE;
    return cogs_k_kmeans_update((hipStream_t)stream, dtype, feats, ts, T, PD, K, assign, reseed_rows, centres,
                                centre_ts, (float*)ws, cogs_k_kmeans_update_blocks(PD), shift_out);
}
cogs_status cogs_select_near_centroid(cogs_stream stream, const float* dist2, const int64_t* assign, int T, int K, int n_extra,
                                      int64_t* out_idx, int32_t* out_counts) {
    if (!dist2 || !assign || !out_idx || !out_counts) return COGS_E_INVALID;
    return cogs_k_select_near((hipStream_t)stream, dist2, assign, T, K, n_extra, out_idx, out_counts);
}
cogs_status cogs_kmeans_pp_step(cogs_stream stream, int dtype, const void* feats, int T, int64_t PD, int row, int first,
                                float* nearest2, float* probs_host, void* ws, size_t ws_bytes) {
    int ns = 0;
    const size_t need = cogs_k_kmeans_ws(T, PD, 1, &ns);
    if (!ws || ws_bytes < need) return COGS_E_WORKSPACE;
    if (!feats || !nearest2 || T <= 0) return COGS_E_INVALID;
    return cogs_k_kmeans_pp_step((hipStream_t)stream, dtype, feats, T, PD, row, first, nearest2, probs_host, (float*)ws, ns);
}
cogs_status cogs_kmeans_pp(cogs_stream stream, int dtype, const void* feats, int T, int64_t PD, int K, int first_row,
                           const float* q_draws, int32_t* idx, int32_t* zero_flag, float* nearest2, void* ws, size_t ws_bytes) {
    int ns = 0;
    const size_t need = cogs_k_kmeans_ws(T, PD, 1, &ns);
    if (!ws || ws_bytes < need) return COGS_E_WORKSPACE;
    if (!feats || !q_draws || !idx || !zero_flag || !nearest2 || T <= 0 || K <= 0) return COGS_E_INVALID;
    return cogs_k_kmeans_pp((hipStream_t)stream, dtype, feats, T, PD, K, first_row, q_draws, idx, zero_flag, nearest2, (float*)ws, ns);
}
cogs_status cogs_kmeans_margins(cogs_stream stream, int T, int64_t PD, int K, void* ws, size_t ws_bytes, float* min_margin,
                                int32_t* rows_below) {
    const size_t need = cogs_k_kmeans_ws(T, PD, K, nullptr);
    if (!ws || ws_bytes < need) return COGS_E_WORKSPACE;
    return cogs_k_kmeans_margins((hipStream_t)stream, T, PD, K, (float*)ws, min_margin, rows_below);
}
cogs_status cogs_kmeans_lloyd(cogs_stream stream, int dtype, const void* feats, const float* ts, int T, int64_t PD, int K,
                              float alpha, int max_iter, float tol, const int32_t* reseed_pool, int pool_len,
                              float* centres, float* centre_ts, int64_t* assign, int* iterations, int* reseeds_used,
                              int* pool_exhausted, void* ws, size_t ws_bytes) {
    int ns = 0;
    const size_t need = cogs_k_kmeans_ws(T, PD, K, &ns);
    if (!ws || ws_bytes < need) return COGS_E_WORKSPACE;
    if (!feats || !ts || !centres || !centre_ts || !assign || (pool_len > 0 && !reseed_pool)) return COGS_E_INVALID;
    return cogs_k_kmeans_lloyd((hipStream_t)stream, dtype, feats, ts, T, PD, K, alpha, max_iter, tol, reseed_pool, pool_len,
                               centres, centre_ts, assign, iterations, reseeds_used, pool_exhausted, (float*)ws, ns);
}
cogs_status cogs_pack_rows(cogs_stream stream, int in_dtype, int out_dtype, const void* in, int64_t ld_in, void* out,
                           int64_t ld_out, int rows, int cols_in, int cols_out) {
    return cogs_k_pack_rows((hipStream_t)stream, in_dtype, out_dtype, in, ld_in, out, ld_out, rows, cols_in, cols_out);
}

cogs_status cogs_preprocess_workspace_bytes(int T, int H, int tw, size_t* bytes) {
    if (!bytes || T <= 0 || H <= 0 || tw <= 0) return COGS_E_INVALID;
    *bytes = align_up((size_t)T * H * tw * 3);
    return COGS_OK;
}
cogs_status cogs_preprocess_frames(cogs_stream stream, const uint8_t* frames, int T, int H, int W, int th, int tw,
                                   int merge, const int32_t* bounds_x, const int32_t* coef_x, int ksx,
                                   const int32_t* bounds_y, const int32_t* coef_y, int ksy, const float* value_table,
                                   void* out, int out_dtype, void* ws, size_t ws_bytes) {
    if (!frames || !out || !bounds_x || !coef_x || !bounds_y || !coef_y || !value_table) return COGS_E_INVALID;
    if (!ws || ws_bytes < (size_t)T * H * tw * 3) return COGS_E_WORKSPACE;
    return cogs_k_preprocess((hipStream_t)stream, frames, T, H, W, th, tw, merge, bounds_x, coef_x, ksx, bounds_y,
                             coef_y, ksy, out, out_dtype, value_table, (uint8_t*)ws);
}

cogs_status cogs_argmax(cogs_stream stream, const float* logits, int n, int64_t* out, void* ws) {
    if (!ws) return COGS_E_WORKSPACE;
    return cogs_k_argmax((hipStream_t)stream, logits, n, out, (float*)ws);
}
cogs_status cogs_logits_process(cogs_stream stream, float* logits, int n, const int64_t* prev, int n_prev,
                                float repetition_penalty, const int32_t* allowed, int n_allowed, float temperature,
                                float* tmp) {
    if (n_prev > 0 && !tmp) return COGS_E_WORKSPACE;
    return cogs_k_logits_process((hipStream_t)stream, logits, n, prev, n_prev, repetition_penalty, allowed, n_allowed,
                                 temperature, tmp);
}
size_t cogs_sample_workspace_bytes(void) { return cogs_k_sample_ws(); }
cogs_status cogs_sample(cogs_stream stream, const float* logits, int n, float temperature, int top_k, double top_p,
                        const float* draws,
                        uint64_t seed, uint64_t offset, int64_t* out_token, int32_t* kept_idx, float* kept_prob,
                        int32_t* n_kept, int kept_cap, void* ws) {
    return cogs_k_sample((hipStream_t)stream, logits, n, temperature, top_k, top_p, draws, seed, offset, out_token, kept_idx, kept_prob,
                         n_kept, kept_cap, ws);
}
cogs_status cogs_topk(cogs_stream stream, const float* logits, int n, int top_k, float* topk_val, int32_t* topk_idx,
                      float* ws) {
    if (!ws) return COGS_E_WORKSPACE;
    return cogs_k_topk((hipStream_t)stream, logits, n, top_k, topk_val, topk_idx, ws);
}

// ---------------------------------------------------------------- vision encoder

cogs_status cogs_vit_load(cogs_handle h, const cogs_vit_weights* w) {
    if (!h || !w || !w->layer || w->layers <= 0 || w->heads <= 0) return COGS_E_INVALID;
    if (w->hidden % w->heads || (w->hidden / w->heads) % 4) return COGS_E_INVALID;
    const int slab = w->dtype == COGS_DT_BF16 ? 64 : 32;
    if (w->hidden % slab || w->inter_pad % slab || w->patch_pad % slab || w->inter_pad % 4) return COGS_E_INVALID;
    h->vit = *w;
    h->vit_layers.assign(w->layer, w->layer + w->layers);
    h->vit.layer = h->vit_layers.data();
    // VisionRotaryEmbedding(head_dim // 2): inv_freq = 1 / theta^(arange(0, dim, 2)/dim), dim = head_dim/2
    const int hd = w->hidden / w->heads;
    const int dim = hd / 2;
    const int nf = dim / 2;
    std::vector<float> inv(nf);
    for (int i = 0; i < nf; ++i) inv[i] = 1.0f / powf(10000.0f, (float)(2 * i) / (float)dim);
    if (h->vit_inv_freq) (void)hipFree(h->vit_inv_freq);
    if (hipMalloc(&h->vit_inv_freq, nf * sizeof(float)) != hipSuccess) return COGS_E_HIP;
    if (hipMemcpy(h->vit_inv_freq, inv.data(), nf * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return COGS_E_HIP;
    h->vit_nfreq = nf;
    h->vit_ok = true;
    return COGS_OK;
}

static size_t vit_carve(const cogs_vit_weights& w, int64_t N, int nframes, Carver& c, void** xpad, void** x, void** ln,
                        void** big, float** rc, float** rs, int32_t** cu, int32_t** lo, int32_t** hi) {
    const size_t es = esize(w.dtype);
    const int hd = w.hidden / w.heads;
    *xpad = c.take((size_t)N * w.patch_pad * es);
    *x = c.take((size_t)N * w.hidden * es);
    *ln = c.take((size_t)N * w.hidden * es);
    const size_t bigcols = (size_t)(4 * w.hidden > w.inter_pad ? 4 * w.hidden : w.inter_pad);
    *big = c.take((size_t)N * bigcols * es);  // [qkv | attn_out] then reused as the MLP hidden
    *rc = (float*)c.take((size_t)N * (hd / 2) * 2 * sizeof(float));   // interleaved (cos, sin) table
    *rs = nullptr;
    *cu = (int32_t*)c.take((size_t)(nframes + 1) * sizeof(int32_t));
    *lo = (int32_t*)c.take((size_t)N * sizeof(int32_t));
    *hi = (int32_t*)c.take((size_t)N * sizeof(int32_t));
    (void)c.take(28 * 1024);   // rotary position LUT (directly behind `hi`)
    return c.off;
}

cogs_status cogs_vit_set_streams(cogs_handle h, int streams) {
    if (!h || streams < 1 || streams > 4) return COGS_E_INVALID;
    h->vit_streams = streams;
    return COGS_OK;
}

cogs_status cogs_vit_workspace_bytes(cogs_handle h, int64_t n_patches, size_t* bytes) {
    if (!h || !h->vit_ok || !bytes || n_patches <= 0) return COGS_E_INVALID;
    Carver c(nullptr, 0);
    void *a, *b, *d, *e; float *rc, *rs; int32_t *cu, *lo, *hi;
    // frames <= patches; size the cu_seqlens array for the worst case
    // + room for the further sets of fixed-size tables when a clip is encoded as up to four frame ranges (cogs_vit_encode)
    *bytes = vit_carve(h->vit, n_patches, (int)n_patches, c, &a, &b, &d, &e, &rc, &rs, &cu, &lo, &hi) + 256 * 1024;
    return COGS_OK;
}

// One contiguous run of frames of a clip on one stream. The encoder is queued in three kinds of step -- begin (tables,
// patch embedding), layer(l), finish (post-LayerNorm + merge) -- so that cogs_vit_encode can ALTERNATE the steps of several
// ranges between their streams: queued one whole range after the other (rounds 3-4), the second stream's first kernel
// reached the GPU ~250 launches = 0.9 ms after the first stream's, and ran that much alone at the end -- a tenth of a
// frame-sharded rank's 8.5 ms step.
struct VitRange {
    cogs_handle h; hipStream_t st;
    const void* pixel_values; int pix_dtype;
    std::vector<int64_t> grid, merge;     // [V][3], [V]
    int attn_mode; void* out_tokens; void* ws; size_t ws_bytes;
    // derived by begin()
    int64_t N = 0; int nframes = 0, V = 0;
    void *xpad = nullptr, *x = nullptr, *ln = nullptr, *big = nullptr;
    float *rc = nullptr, *rs = nullptr, *lut = nullptr, *stat_part = nullptr, *ln_ab = nullptr;
    int32_t *cu = nullptr, *lo = nullptr, *hi = nullptr;
    bool fold = false, use_lut = false, prescale_q = false, head_major = false;
    int maxpos = 0, max_seq = 0, uniform_seq = 0;
    float scale = 1.f;

    cogs_status begin();
    cogs_status layer(int l);
    cogs_status finish();
};

// Frames are independent under block-diagonal attention (per-frame attention, RoPE and 2x2 merge:
// modeling_videollama3_encoder.py:309-312,427,487-501), so a clip is encoded as two halves on two streams: the second
// half's kernels take the CUs the first half's ragged last rounds leave idle (every persistent GEMM launch ends in a
// partly filled round of tiles). Same arithmetic per row, bit-identical tokens (tests/test_gpu_models.py). One rank's
// share of a frame-sharded clip (8 frames of 22 x 42 patches: 145 tiles for the N = 1152 shapes) gains 10 %, the whole
// 64-frame clip 2.7 % (round 4, same box, interleaved: 60.8 -> 59.2 ms; 256 frames at 140 x 280: 47.3 -> 46.5 ms).
// Not while the per-kernel profiler is on: overlapping launches stretch every bracket, so bench.py's `roofline` and
// `breakdown_ms` describe the kernels one at a time. cogs_vit_set_streams(h, 1) or the debug switch vit_split_max switch it off.

static cogs_status cogs_project_rows(cogs_handle h, hipStream_t st, const void* tokens, int M, void* out, void* ws);

// proj_out != NULL: every frame range also runs the projector on its own tokens, on its own stream (cogs_vit_encode_project)
static cogs_status vit_encode_impl(cogs_handle h, cogs_stream stream, const void* pixel_values, int pix_dtype,
                                   const int64_t* grid_sizes, const int64_t* merge_sizes, int V, int attn_mode,
                                   void* out_tokens, void* ws, size_t ws_bytes, void* proj_out, void* proj_ws) {
    if (!h || !h->vit_ok || !pixel_values || !grid_sizes || !merge_sizes || V <= 0 || !out_tokens) return COGS_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    int64_t N = 0, nframes = 0;
    for (int v = 0; v < V; ++v) {
        const int64_t t = grid_sizes[3 * v], gh = grid_sizes[3 * v + 1], gw = grid_sizes[3 * v + 2], ms = merge_sizes[v];
        if (t <= 0 || gh <= 0 || gw <= 0 || ms <= 0 || gh % ms || gw % ms) return COGS_E_INVALID;
        N += t * gh * gw;
        nframes += t;
    }
    const cogs_vit_weights& w = h->vit;
    auto whole = [&]() {
        VitRange r{h, st, pixel_values, pix_dtype, std::vector<int64_t>(grid_sizes, grid_sizes + 3 * V),
                   std::vector<int64_t>(merge_sizes, merge_sizes + V), attn_mode, out_tokens, ws, ws_bytes};
        COGS_TRY(r.begin());
        for (int l = 0; l < w.layers; ++l) COGS_TRY(r.layer(l));
        COGS_TRY(r.finish());
        if (proj_out) {
            int64_t toks = 0;
            for (int v = 0; v < V; ++v) toks += grid_sizes[3 * v] * grid_sizes[3 * v + 1] * grid_sizes[3 * v + 2] / (merge_sizes[v] * merge_sizes[v]);
            COGS_TRY(cogs_project_rows(h, st, out_tokens, (int)toks, proj_out, proj_ws));
        }
        return COGS_OK;
    };
    const int64_t split_max = g_cogs_debug.vit_split_max;
    int S = h->vit_streams;
    if (S > nframes) S = (int)nframes;
    if (S < 2 || h->prof_on || attn_mode != COGS_ATTN_BLOCK_DIAG || N > split_max) return whole();

    // cut the frames into S contiguous runs at the frame boundaries nearest to r / S of the patches
    std::vector<VitRange> R;
    std::vector<int64_t> tok_begin;                       // first token row of every range, then the total
    {
        std::vector<int64_t> frame_rows, frame_v;       // per frame: patches, video
        for (int v = 0; v < V; ++v)
            for (int64_t f = 0; f < grid_sizes[3 * v]; ++f) { frame_rows.push_back(grid_sizes[3 * v + 1] * grid_sizes[3 * v + 2]); frame_v.push_back(v); }
        std::vector<int64_t> cum(nframes + 1, 0);
        for (int64_t f = 0; f < nframes; ++f) cum[f + 1] = cum[f] + frame_rows[f];
        std::vector<int64_t> cuts(1, 0);
        for (int r = 1; r < S; ++r) {
            int64_t best = cuts.back() + 1;
            for (int64_t f = cuts.back() + 1; f <= nframes - (S - r); ++f)
                if (llabs(cum[f] * S - N * r) < llabs(cum[best] * S - N * r)) best = f;
            cuts.push_back(best);
        }
        cuts.push_back(nframes);
        size_t ws_off = 0;
        int64_t tok_off = 0;
        tok_begin.clear();
        const size_t pes = pix_dtype == COGS_DT_BF16 ? 2 : 4;
        bool ok = ws != nullptr;
        for (int r = 0; r < S && ok; ++r) {
            VitRange q{h, nullptr, (const char*)pixel_values + (size_t)cum[cuts[r]] * w.patch_dim * pes, pix_dtype, {}, {}, attn_mode,
                       (char*)out_tokens + (size_t)tok_off * w.hidden * esize(w.dtype), (char*)ws + ws_off, 0};
            int64_t rows = 0, toks = 0;
            for (int64_t f = cuts[r]; f < cuts[r + 1]; ++f) {       // frames -> (t, gh, gw) runs per video
                const int v = (int)frame_v[f];
                if (!q.merge.empty() && q.grid[q.grid.size() - 2] == grid_sizes[3 * v + 1] && q.grid.back() == grid_sizes[3 * v + 2] &&
                    f > cuts[r] && frame_v[f - 1] == v) ++q.grid[q.grid.size() - 3];
                else { q.grid.insert(q.grid.end(), {1, grid_sizes[3 * v + 1], grid_sizes[3 * v + 2]}); q.merge.push_back(merge_sizes[v]); }
                rows += frame_rows[f];
                toks += frame_rows[f] / (merge_sizes[v] * merge_sizes[v]);
            }
            Carver c(nullptr, 0);
            void *p0, *p1, *p2, *p3; float *q0, *q1; int32_t *i0, *i1, *i2;
            q.ws_bytes = vit_carve(w, rows, (int)(cuts[r + 1] - cuts[r]), c, &p0, &p1, &p2, &p3, &q0, &q1, &i0, &i1, &i2);
            ws_off += q.ws_bytes;
            tok_begin.push_back(tok_off);
            tok_off += toks;
            ok = ws_off <= ws_bytes && rows > 0;
            R.push_back(std::move(q));
        }
        tok_begin.push_back(tok_off);
        if (!ok) return whole();
    }
    while ((int)h->aux_streams.size() < S - 1) {          // streams and events of the handle, created once
        hipStream_t s2 = nullptr;
        hipEvent_t e2 = nullptr;
        if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess) {
            if (s2) (void)hipStreamDestroy(s2);       // whatever was created goes back; the clip is encoded on the caller's stream alone
            return whole();
        }
        h->aux_streams.push_back(s2);
        h->ev_joins.push_back(e2);
    }
    if (!h->ev_fork && hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) return whole();
    if (hipEventRecord(h->ev_fork, st) != hipSuccess) return COGS_E_HIP;
    R[0].st = st;
    for (int r = 1; r < S; ++r) {
        R[r].st = h->aux_streams[r - 1];
        if (hipStreamWaitEvent(R[r].st, h->ev_fork, 0) != hipSuccess) return COGS_E_HIP;
    }
    cogs_k_gemm_co_streams(S);
    cogs_status rc = COGS_OK;
    for (int r = 0; r < S && rc == COGS_OK; ++r) rc = R[r].begin();
    for (int l = 0; l < w.layers && rc == COGS_OK; ++l)
        for (int r = 0; r < S && rc == COGS_OK; ++r) rc = R[r].layer(l);
    for (int r = 0; r < S && rc == COGS_OK; ++r) rc = R[r].finish();
    if (proj_out) {
        // the projector's two GEMMs of a range run behind that range's encoder, on its stream: rows are independent, so the
        // projected tokens are the bits cogs_project gives on the whole clip, and no range waits for the join to be projected
        const size_t tes = (size_t)w.hidden * esize(w.dtype), pes2 = (size_t)h->proj.out_dim * esize(h->proj.dtype);
        for (int r = 0; r < S && rc == COGS_OK; ++r)
            rc = cogs_project_rows(h, R[r].st, (const char*)out_tokens + (size_t)tok_begin[r] * tes, (int)(tok_begin[r + 1] - tok_begin[r]),
                                   (char*)proj_out + (size_t)tok_begin[r] * pes2, (char*)proj_ws + (size_t)tok_begin[r] * pes2);
    }
    cogs_k_gemm_co_streams(1);
    // join unconditionally: the caller's stream must not run ahead of anything queued on the other ones
    for (int r = 1; r < S; ++r)
        if (hipEventRecord(h->ev_joins[r - 1], R[r].st) != hipSuccess || hipStreamWaitEvent(st, h->ev_joins[r - 1], 0) != hipSuccess) return COGS_E_HIP;
    return rc;
}

cogs_status cogs_vit_encode(cogs_handle h, cogs_stream stream, const void* pixel_values, int pix_dtype,
                            const int64_t* grid_sizes, const int64_t* merge_sizes, int V, int attn_mode,
                            void* out_tokens, void* ws, size_t ws_bytes) {
    return vit_encode_impl(h, stream, pixel_values, pix_dtype, grid_sizes, merge_sizes, V, attn_mode, out_tokens, ws, ws_bytes,
                           nullptr, nullptr);
}

cogs_status cogs_vit_encode_project(cogs_handle h, cogs_stream stream, const void* pixel_values, int pix_dtype,
                                    const int64_t* grid_sizes, const int64_t* merge_sizes, int V, int attn_mode,
                                    void* out_tokens, void* ws, size_t ws_bytes, void* proj_out, void* proj_ws,
                                    size_t proj_ws_bytes) {
    if (!h || !h->proj_ok || !h->vit_ok || !proj_out || !grid_sizes || !merge_sizes || V <= 0) return COGS_E_INVALID;
    if (h->proj.in_dim != h->vit.hidden || h->proj.dtype != h->vit.dtype) return COGS_E_INVALID;
    int64_t toks = 0;
    for (int v = 0; v < V; ++v) {
        const int64_t ms = merge_sizes[v];
        if (ms <= 0) return COGS_E_INVALID;
        toks += grid_sizes[3 * v] * grid_sizes[3 * v + 1] * grid_sizes[3 * v + 2] / (ms * ms);
    }
    if (!proj_ws || proj_ws_bytes < (size_t)toks * h->proj.out_dim * esize(h->proj.dtype)) return COGS_E_WORKSPACE;
    return vit_encode_impl(h, stream, pixel_values, pix_dtype, grid_sizes, merge_sizes, V, attn_mode, out_tokens, ws, ws_bytes,
                           proj_out, proj_ws);
}

cogs_status VitRange::begin() {
    const cogs_vit_weights& w = h->vit;
    const int dt = w.dtype;
    const int H = w.hidden, hd = H / w.heads;
    V = (int)merge.size();
    const int64_t* grid_sizes = grid.data();
    const int64_t* merge_sizes = merge.data();
    N = 0; nframes = 0;
    for (int v = 0; v < V; ++v) {
        const int64_t t = grid_sizes[3 * v], gh = grid_sizes[3 * v + 1], gw = grid_sizes[3 * v + 2], ms = merge_sizes[v];
        if (t <= 0 || gh <= 0 || gw <= 0 || ms <= 0 || gh % ms || gw % ms) return COGS_E_INVALID;
        N += t * gh * gw;
        nframes += (int)t;
    }
    if (N > 0x7fffffff) return COGS_E_INVALID;
    Carver c(ws, ws_bytes);
    const size_t need = vit_carve(w, N, nframes, c, &xpad, &x, &ln, &big, &rc, &rs, &cu, &lo, &hi);
    if (!ws || ws_bytes < need) return COGS_E_WORKSPACE;
    // LayerNorm folded into the GEMMs (bf16 production path, cogs_vit_layer.qkv_c ...): the buffer that would hold LN(x)
    // carries the per-row partial statistics [N][H/64][2] and the per-row (rstd, -rstd*mean) [N][2] instead
    fold = dt == COGS_DT_BF16 && H % 64 == 0;
    for (int l = 0; l < w.layers && fold; ++l) {
        const cogs_vit_layer& L = h->vit_layers[l];
        fold = L.qkv_c && L.fc1_c;
    }
    stat_part = (float*)ln;
    ln_ab = stat_part + (size_t)N * (H / 64) * 2;

    // rotary position LUT for the ping-pong QKV GEMM (bf16, block-diagonal): (cos, sin)[pos][freq], kept in LDS there
    maxpos = 0;
    for (int v = 0; v < V; ++v) {
        const int gh = (int)grid_sizes[3 * v + 1], gw = (int)grid_sizes[3 * v + 2];
        maxpos = gh > maxpos ? gh : maxpos;
        maxpos = gw > maxpos ? gw : maxpos;
    }
    lut = (float*)(hi + N);   // carved right behind `hi` (vit_carve)
    lut = (float*)(((uintptr_t)lut + 255) & ~(uintptr_t)255);
    use_lut = dt == COGS_DT_BF16 && attn_mode == COGS_ATTN_BLOCK_DIAG && hd % 4 == 0 && maxpos * (hd / 4) * 8 <= 27 * 1024;

    // cu_seqlens (:439-440), same-frame ranges for the eager-global mode, rotary tables (:405-434): all written by
    // kernels on `st` from the grid -- no host staging, so back-to-back encodes with different grids cannot race
    max_seq = 0; uniform_seq = 0;
    {
        bool alike = true;
        for (int v = 1; v < V; ++v)
            alike = alike && grid_sizes[3 * v + 1] * grid_sizes[3 * v + 2] == grid_sizes[1] * grid_sizes[2];
        if (alike) uniform_seq = (int)(grid_sizes[1] * grid_sizes[2]);
    }
    {
        int64_t row = 0;
        int frame0 = 0;
        const bool need_rows = attn_mode == COGS_ATTN_REF_EAGER_GLOBAL;
        for (int v = 0; v < V; ++v) {
            const int t = (int)grid_sizes[3 * v], gh = (int)grid_sizes[3 * v + 1], gw = (int)grid_sizes[3 * v + 2];
            const int per = gh * gw;
            if (per > max_seq) max_seq = per;
            { PROF(COGS_PROF_OTHER); COGS_TRY(cogs_k_vit_rope_table(st, rc, rs, (int)row, t, gh, gw, (int)merge_sizes[v], h->vit_inv_freq, h->vit_nfreq)); }
            if (use_lut) {   // `lo` is free in block-diagonal mode: per-row positions; the LUT itself once (first video)
                PROF(COGS_PROF_OTHER);
                COGS_TRY(cogs_k_vit_rope_lut(st, lo, (int)row, t, gh, gw, (int)merge_sizes[v], v == 0 ? lut : nullptr, maxpos,
                                             h->vit_inv_freq, h->vit_nfreq));
            }
            { PROF(COGS_PROF_OTHER); COGS_TRY(cogs_k_vit_segments(st, cu, need_rows ? lo : nullptr, need_rows ? hi : nullptr, (int)row, frame0, t, per)); }
            row += (int64_t)t * per;
            frame0 += t;
        }
    }

    // patch embed: conv2d k=s=14 == GEMM over the 588-element rows (:202-210)
    { PROF(COGS_PROF_OTHER); COGS_TRY(cogs_k_pack_rows(st, pix_dtype, dt, pixel_values, w.patch_dim, xpad, w.patch_pad, (int)N, w.patch_dim, w.patch_pad)); }
    {
        CogsGemm g; g.dtype = dt;
        g.A = xpad; g.lda = w.patch_pad; g.W = w.patch_w; g.ldw = w.patch_pad; g.C = x; g.ldc = H;
        g.bias = w.patch_b; g.M = (int)N; g.N = H; g.K = w.patch_pad;
        if (fold) g.row_stats = stat_part;
        { PROF(COGS_PROF_GEMM); COGS_TRY(cogs_k_gemm(st, g)); }
    }
    scale = 1.0f / sqrtf((float)hd);
    // bf16 MFMA attention kernels take Q pre-multiplied by scale*log2(e): the QKV GEMM epilogue folds the factor in
    // before its single rounding (same relative rounding error as rounding q itself) and the softmax needs no
    // per-score multiply. Not in the parity modes (fp32, or the eager-global bias mode).
    prescale_q = dt == COGS_DT_BF16 && attn_mode == COGS_ATTN_BLOCK_DIAG && (hd == 72 || hd == 128);
    // Round 5: the production path keeps q, k, v HEAD-major between the QKV GEMM and the attention kernel --
    // [q | k | v][head][row][hd] in the same N x 3H elements -- so that a (frame, head) block of K / V is one contiguous
    // run and the attention kernel's LDS-DMA pieces are whole 128-byte lines (csrc/gemm_epilogue.h EPI_HM, csrc/attn_vit.hip).
    // Only that kernel reads the layout; the parity modes (fp32, eager-global) stay token-major.
    head_major = prescale_q && hd == 72 && g_cogs_debug.gemm_headmajor != 0 && g_cogs_debug.attn_vit != 0 &&
                 (double)N * H * 2.0 < 4294967296.0;
    return COGS_OK;
}

cogs_status VitRange::layer(int l) {
    const cogs_vit_weights& w = h->vit;
    const int dt = w.dtype;
    const size_t es = esize(dt);
    const int H = w.hidden, hd = H / w.heads;
    char* qkv = (char*)big;
    char* att = qkv + (size_t)N * 3 * H * es;
    const cogs_vit_layer& L = h->vit_layers[l];
    if (fold) { PROF(COGS_PROF_NORM); COGS_TRY(cogs_k_ln_finalize(st, stat_part, (int)N, H / 64, H, w.ln_eps, ln_ab)); }
    else { PROF(COGS_PROF_NORM); COGS_TRY(cogs_k_layernorm(st, dt, x, ln, L.ln1_g, L.ln1_b, (int)N, H, w.ln_eps)); }
    {
        CogsGemm g; g.dtype = dt;
        g.A = fold ? x : ln; g.lda = H; g.W = L.qkv_w; g.ldw = H; g.C = qkv; g.ldc = 3 * H;
        g.M = (int)N; g.N = 3 * H; g.K = H;
        if (fold) { g.ln_ab = ln_ab; g.col_c = L.qkv_c; }
        else g.bias = L.qkv_b;
        g.rope_cos = rc; g.rope_sin = rs; g.rope_cols = 2 * H; g.head_dim = hd;
        if (prescale_q) { g.q_scale = scale * 1.4426950408889634f; g.q_cols = H; }
        if (use_lut) { g.rope_lut = lut; g.rope_rowpos = lo; g.rope_maxpos = maxpos; }
        if (head_major) { g.hm_rows = N; g.hm_cols = H; }
        { PROF(COGS_PROF_GEMM); COGS_TRY(cogs_k_gemm(st, g)); }
    }
    {
        CogsAttn a; a.dtype = dt;
        a.Q = qkv; a.K = qkv + (size_t)H * es; a.V = qkv + (size_t)2 * H * es; a.O = att;
        a.ldq = a.ldk = a.ldv = 3 * H; a.ldo = H;
        if (head_major) {
            a.K = qkv + (size_t)N * H * es; a.V = qkv + (size_t)2 * N * H * es;
            a.ldq = a.ldk = a.ldv = hd; a.head_stride = (long)N * hd;
        }
        a.q_len = (int)N; a.kv_len = (int)N; a.hq = a.hkv = w.heads; a.head_dim = hd; a.scale = scale;
        a.q_prescaled = prescale_q;
        if (attn_mode == COGS_ATTN_REF_EAGER_GLOBAL) { a.row_lo = lo; a.row_hi = hi; a.bias = 1.0f; }
        else {
            a.cu_seqlens = cu; a.nseg = nframes; a.max_seqlen = max_seq;
            a.uniform_seqlen = uniform_seq;      // all frames alike (one video, or videos of one grid): see attn_vit.hip
        }
        { PROF(COGS_PROF_ATTN); COGS_TRY(cogs_k_attention(st, a)); }
    }
    {
        CogsGemm g; g.dtype = dt;
        g.A = att; g.lda = H; g.W = L.o_w; g.ldw = H; g.C = x; g.ldc = H;
        g.bias = L.o_b; g.residual = x; g.ldr = H; g.M = (int)N; g.N = H; g.K = H;
        if (fold) g.row_stats = stat_part;
        { PROF(COGS_PROF_GEMM); COGS_TRY(cogs_k_gemm(st, g)); }
    }
    if (fold) { PROF(COGS_PROF_NORM); COGS_TRY(cogs_k_ln_finalize(st, stat_part, (int)N, H / 64, H, w.ln_eps, ln_ab)); }
    else { PROF(COGS_PROF_NORM); COGS_TRY(cogs_k_layernorm(st, dt, x, ln, L.ln2_g, L.ln2_b, (int)N, H, w.ln_eps)); }
    {
        CogsGemm g; g.dtype = dt;
        g.A = fold ? x : ln; g.lda = H; g.W = L.fc1_w; g.ldw = H; g.C = big; g.ldc = w.inter_pad;
        g.M = (int)N; g.N = w.inter_pad; g.K = H; g.act = COGS_ACT_GELU_TANH;
        if (fold) { g.ln_ab = ln_ab; g.col_c = L.fc1_c; }
        else g.bias = L.fc1_b;
        { PROF(COGS_PROF_GEMM); COGS_TRY(cogs_k_gemm(st, g)); }
    }
    {
        CogsGemm g; g.dtype = dt;
        g.A = big; g.lda = w.inter_pad; g.W = L.fc2_w; g.ldw = w.inter_pad; g.C = x; g.ldc = H;
        g.bias = L.fc2_b; g.residual = x; g.ldr = H; g.M = (int)N; g.N = H; g.K = w.inter_pad;
        if (fold && l + 1 < w.layers) g.row_stats = stat_part;     // the last layer feeds post_layernorm (own kernel)
        { PROF(COGS_PROF_GEMM); COGS_TRY(cogs_k_gemm(st, g)); }
    }
    return COGS_OK;
}

// post_layernorm + per-video 2x2 merge (:482-510)
cogs_status VitRange::finish() {
    const cogs_vit_weights& w = h->vit;
    const int dt = w.dtype;
    const size_t es = esize(dt);
    const int H = w.hidden;
    int64_t row = 0, orow = 0;
    for (int v = 0; v < V; ++v) {
        const int64_t n = grid[3 * v] * grid[3 * v + 1] * grid[3 * v + 2];
        const int grp = (int)(merge[v] * merge[v]);
        { PROF(COGS_PROF_NORM); COGS_TRY(cogs_k_ln_merge(st, dt, (char*)x + (size_t)row * H * es, (char*)out_tokens + (size_t)orow * H * es,
                                 w.post_ln_g, w.post_ln_b, (int)(n / grp), grp, H, w.ln_eps)); }
        row += n;
        orow += n / grp;
    }
    return COGS_OK;
}

// ---------------------------------------------------------------- multi-GPU
// RCCL is bound at first use: libcogs_hip.so itself does not link against it (a single-GPU user never loads it)
typedef int (*cogs_nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
static cogs_nccl_allgather_fn cogs_bind_allgather() {
    static std::atomic<int> state{0};      // 0 = not tried, 1 = bound, -1 = unavailable
    static cogs_nccl_allgather_fn fn = nullptr;
    const int s = state.load(std::memory_order_acquire);
    if (s != 0) return s > 0 ? fn : nullptr;
    void* lib = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    cogs_nccl_allgather_fn f = lib ? (cogs_nccl_allgather_fn)dlsym(lib, "ncclAllGather") : nullptr;
    fn = f;
    state.store(f ? 1 : -1, std::memory_order_release);
    return f;
}
cogs_status cogs_allgather_tokens(cogs_stream stream, void* nccl_comm, const void* local, size_t local_rows,
                                  size_t row_bytes, void* global) {
    if (!nccl_comm || !local || !global || local_rows == 0 || row_bytes == 0) return COGS_E_INVALID;
    cogs_nccl_allgather_fn f = cogs_bind_allgather();
    if (!f) return COGS_E_UNSUPPORTED;
    const int nccl_uint8 = 1;              // ncclUint8 (rccl.h): the payload is moved as bytes, whatever its dtype
    return f(local, global, local_rows * row_bytes, nccl_uint8, nccl_comm, (hipStream_t)stream) == 0 ? COGS_OK : COGS_E_HIP;
}

cogs_status cogs_proj_load(cogs_handle h, const cogs_proj_weights* w) {
    if (!h || !w || !w->w1 || !w->w2) return COGS_E_INVALID;
    const int slab = w->dtype == COGS_DT_BF16 ? 64 : 32;
    if (w->in_dim % slab || w->out_dim % slab) return COGS_E_INVALID;
    h->proj = *w;
    h->proj_ok = true;
    return COGS_OK;
}

cogs_status cogs_project(cogs_handle h, cogs_stream stream, const void* tokens, int M, void* out, void* ws,
                         size_t ws_bytes) {
    if (!h || !h->proj_ok || !tokens || !out || M <= 0) return COGS_E_INVALID;
    const cogs_proj_weights& w = h->proj;
    if (!ws || ws_bytes < (size_t)M * w.out_dim * esize(w.dtype)) return COGS_E_WORKSPACE;
    return cogs_project_rows(h, (hipStream_t)stream, tokens, M, out, ws);
}

// the projector's two GEMMs on M rows (arguments checked by the callers; ws holds M x out_dim elements)
static cogs_status cogs_project_rows(cogs_handle h, hipStream_t st, const void* tokens, int M, void* out, void* ws) {
    if (M <= 0) return COGS_OK;
    const cogs_proj_weights& w = h->proj;
    CogsGemm g; g.dtype = w.dtype;
    g.A = tokens; g.lda = w.in_dim; g.W = w.w1; g.ldw = w.in_dim; g.C = ws; g.ldc = w.out_dim;
    g.bias = w.b1; g.M = M; g.N = w.out_dim; g.K = w.in_dim; g.act = COGS_ACT_GELU_ERF;
    { PROF(COGS_PROF_GEMM); COGS_TRY(cogs_k_gemm(st, g)); }
    CogsGemm g2; g2.dtype = w.dtype;
    g2.A = ws; g2.lda = w.out_dim; g2.W = w.w2; g2.ldw = w.out_dim; g2.C = out; g2.ldc = w.out_dim;
    g2.bias = w.b2; g2.M = M; g2.N = w.out_dim; g2.K = w.out_dim;
    { PROF(COGS_PROF_GEMM); COGS_TRY(cogs_k_gemm(st, g2)); }
    return COGS_OK;
}

// ---------------------------------------------------------------- Qwen2

cogs_status cogs_llm_load(cogs_handle h, const cogs_llm_weights* w) {
    if (!h || !w || !w->layer || w->layers <= 0) return COGS_E_INVALID;
    if (w->heads % w->kv_heads || w->head_dim % 8 || w->vocab % 4) return COGS_E_INVALID;
    const int slab = w->dtype == COGS_DT_BF16 ? 64 : 32;
    if (w->hidden % slab || w->inter % slab || (w->heads * w->head_dim) % slab) return COGS_E_INVALID;
    h->llm = *w;
    h->llm_layers.assign(w->layer, w->layer + w->layers);
    h->llm.layer = h->llm_layers.data();
    // Qwen2RotaryEmbedding: inv_freq = 1 / theta^(arange(0, hd, 2)/hd)
    const int nf = w->head_dim / 2;
    std::vector<float> inv(nf);
    for (int i = 0; i < nf; ++i) inv[i] = 1.0f / powf(w->rope_theta, (float)(2 * i) / (float)w->head_dim);
    // re-pointing the handle at another weight set of the same shape (adapter switch) keeps the device table
    if (!h->llm_inv_freq || h->llm_nfreq != nf || h->llm_theta != w->rope_theta) {
        if (h->llm_inv_freq) (void)hipFree(h->llm_inv_freq);
        h->llm_inv_freq = nullptr;
        if (hipMalloc(&h->llm_inv_freq, nf * sizeof(float)) != hipSuccess) return COGS_E_HIP;
        if (hipMemcpy(h->llm_inv_freq, inv.data(), nf * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return COGS_E_HIP;
        h->llm_nfreq = nf; h->llm_theta = w->rope_theta;
    }
    h->llm_ok = true;
    return COGS_OK;
}

static int llm_nsplit(int ctx) {
    // decode attention: the keys are split over workgroups (x kv heads) and combined by a second kernel
    // measured (ms/token, 7B, kv heads 4): context 15 360: 256 keys per split 3.62, 128: 3.67, 64: 3.66, 512: 3.72;
    // context 2 048: 256: 3.51, 128: 3.42, 64: 3.39 -- i.e. at least ~32 splits, of 64..256 keys
    // => at least 32 splits while a split keeps 64 keys, 256 keys per split beyond that (monotonic in ctx: the
    // workspace is sized with the largest context)
    const int env_keys = (int)g_cogs_debug.llm_split_keys;
    int n;
    if (env_keys > 0) n = (ctx + env_keys - 1) / env_keys;
    else {
        n = (ctx + 255) / 256;
        if (n < 32) n = 32;
        const int cap = (ctx + 63) / 64;
        if (n > cap) n = cap;
    }
    if (n < 1) n = 1;
    if (n > 128) n = 128;
    return n;
}

static size_t llm_carve(const cogs_llm_weights& w, int S, int max_ctx, Carver& c, void** x, void** ln, void** qkv,
                        void** att, void** act, float** rc, float** rs, void** split, size_t* split_bytes,
                        int32_t** pos = nullptr, int32_t** cu = nullptr) {
    const size_t es = esize(w.dtype);
    const int qd = (w.heads + 2 * w.kv_heads) * w.head_dim;
    *x = c.take((size_t)S * w.hidden * es);
    *ln = c.take((size_t)S * w.hidden * es);
    *qkv = c.take((size_t)S * qd * es);
    *att = c.take((size_t)S * w.heads * w.head_dim * es);
    *act = c.take((size_t)S * w.inter * es);
    *rc = (float*)c.take((size_t)S * (w.head_dim / 2) * 2 * sizeof(float));   // interleaved (cos, sin) table
    *rs = nullptr;
    *split_bytes = (size_t)llm_nsplit(max_ctx) * w.heads * (w.head_dim + 2) * sizeof(float);
    *split = c.take(*split_bytes);
    int32_t* pp = (int32_t*)c.take((size_t)S * sizeof(int32_t));          // per-row positions (segmented forward)
    int32_t* cc = (int32_t*)c.take(((size_t)S + 1) * sizeof(int32_t));    // cu_seqlens (at most S segments)
    if (pos) *pos = pp;
    if (cu) *cu = cc;
    return c.off;
}

cogs_status cogs_llm_workspace_bytes(cogs_handle h, int max_tokens, int max_context, size_t* bytes) {
    if (!h || !h->llm_ok || !bytes || max_tokens <= 0) return COGS_E_INVALID;
    Carver c(nullptr, 0);
    void *a, *b, *d, *e, *f, *sp; float *rc, *rs; size_t sb;
    *bytes = llm_carve(h->llm, max_tokens, max_context, c, &a, &b, &d, &e, &f, &rc, &rs, &sp, &sb);
    return COGS_OK;
}

// one implementation behind cogs_llm_forward and cogs_llm_forward_segments: nseg > 0 = stateless forward over
// nseg independent sequences stored back to back (causal attention and positions restart at every cu_host[s])
static cogs_status llm_forward_impl(cogs_handle h, cogs_stream stream, const void* embeds, int S, cogs_kv* kv,
                                    float* last_logits, float* pooled_mean, void* hidden_out, void* ws, size_t ws_bytes,
                                    const int32_t* cu_host, int nseg, float* pooled_seg) {
    if (!h || !h->llm_ok || !embeds || S <= 0) return COGS_E_INVALID;
    const cogs_llm_weights& w = h->llm;
    hipStream_t st = (hipStream_t)stream;
    const int dt = w.dtype;
    const size_t es = esize(dt);
    const int H = w.hidden, hd = w.head_dim, kvd = w.kv_heads * hd, qd_q = w.heads * hd, qd = qd_q + 2 * kvd;
    const int pos0 = kv ? kv->len : 0;
    const int ctx = pos0 + S;
    if (kv && (!kv->k || !kv->v || ctx > kv->max_len)) return COGS_E_INVALID;
    Carver c(ws, ws_bytes);
    void *x, *ln, *qkv, *att, *act, *split; float *rc, *rs; size_t split_bytes;
    int32_t *pos_d = nullptr, *cu_d = nullptr;
    const size_t need = llm_carve(w, S, ctx, c, &x, &ln, &qkv, &att, &act, &rc, &rs, &split, &split_bytes, &pos_d, &cu_d);
    if (!ws || ws_bytes < need) return COGS_E_WORKSPACE;
    int max_seg = 0;
    if (nseg > 0) {
        if (kv || !cu_host || cu_host[0] != 0 || cu_host[nseg] != S || nseg > S) return COGS_E_INVALID;
        for (int sgm = 0; sgm < nseg; ++sgm) {
            const int b = cu_host[sgm], e = cu_host[sgm + 1];
            if (e <= b) return COGS_E_INVALID;
            max_seg = e - b > max_seg ? e - b : max_seg;
        }
        // segment boundaries through the pinned ring (the caller's array may die when this call returns);
        // per-row positions from them on the device
        COGS_TRY(stage_h2d(h, st, cu_d, cu_host, (size_t)(nseg + 1) * sizeof(int32_t)));
        { PROF(COGS_PROF_OTHER); COGS_TRY(cogs_k_seg_positions(st, cu_d, nseg, pos_d)); }
    }

    // single-token decode reads the new embedding row in place (layer 0's qkv input and residual source), everything
    // else copies the prompt rows into the residual stream first
    const bool embeds_in_place = (S == 1 && nseg == 0);
    if (!embeds_in_place && hipMemcpyAsync(x, embeds, (size_t)S * H * es, hipMemcpyDeviceToDevice, st) != hipSuccess) return COGS_E_HIP;
    { PROF(COGS_PROF_OTHER); COGS_TRY(cogs_k_llm_rope_table(st, rc, rs, nseg > 0 ? pos_d : nullptr, pos0, S, h->llm_inv_freq, hd / 2)); }
    const float scale = 1.0f / sqrtf((float)hd);
    const bool prescale_q = dt == COGS_DT_BF16 && hd == 128;   // see cogs_vit_encode
    for (int l = 0; l < w.layers; ++l) {
        const cogs_llm_layer& L = h->llm_layers[l];
        const bool fuse_norm = (S == 1);   // single-token decode: RMSNorm runs inside the GEMV prologue
        const void* xin = (embeds_in_place && l == 0) ? embeds : (const void*)x;    // the residual stream entering this layer
        if (!fuse_norm) { PROF(COGS_PROF_NORM); COGS_TRY(cogs_k_rmsnorm(st, dt, x, ln, L.in_ln, S, H, w.rms_eps)); }
        {
            CogsGemm g; g.dtype = dt;
            g.A = fuse_norm ? xin : ln; g.lda = H; g.W = L.qkv_w; g.ldw = H; g.C = qkv; g.ldc = qd;
            g.bias = L.qkv_b; g.M = S; g.N = qd; g.K = H;
            g.rope_cos = rc; g.rope_sin = rs; g.rope_cols = qd_q + kvd; g.head_dim = hd;
            if (prescale_q) { g.q_scale = scale * 1.4426950408889634f; g.q_cols = qd_q; }
            if (fuse_norm) { g.rms_gamma = L.in_ln; g.rms_eps = w.rms_eps; }
            if (kv && S == 1) {   // single-token decode: the GEMV writes the new K / V row straight into the cache
                g.kv_k = (char*)kv->k + (((size_t)l * kv->max_len) + pos0) * kvd * es;
                g.kv_v = (char*)kv->v + (((size_t)l * kv->max_len) + pos0) * kvd * es;
                g.kv_col0 = qd_q; g.kv_dim = kvd;
            }
            { PROF(COGS_PROF_GEMM); COGS_TRY(cogs_k_gemm(st, g)); }
        }
        const char* kp = (const char*)qkv + (size_t)qd_q * es;
        const char* vp = kp + (size_t)kvd * es;
        long ldkv = qd;
        if (kv) {
            char* kc = (char*)kv->k + ((size_t)l * kv->max_len) * kvd * es;
            char* vc = (char*)kv->v + ((size_t)l * kv->max_len) * kvd * es;
            if (S != 1) {
                PROF(COGS_PROF_OTHER);
                COGS_TRY(cogs_k_kv_append(st, dt, kp, vp, qd, kc + (size_t)pos0 * kvd * es, vc + (size_t)pos0 * kvd * es,
                                          kvd, S, kvd));
            }
            kp = kc; vp = vc; ldkv = kvd;
        }
        {
            CogsAttn a; a.dtype = dt;
            a.Q = qkv; a.K = kp; a.V = vp; a.O = att;
            a.ldq = qd; a.ldk = ldkv; a.ldv = ldkv; a.ldo = qd_q;
            a.q_len = S; a.kv_len = ctx; a.hq = w.heads; a.hkv = w.kv_heads; a.head_dim = hd; a.scale = scale;
            a.causal = 1; a.q_pos0 = pos0; a.q_prescaled = prescale_q;
            if (nseg > 0) { a.cu_seqlens = cu_d; a.nseg = nseg; a.max_seqlen = max_seg; }
            if (S == 1 && dt == COGS_DT_BF16 && hd == 128) {
                a.nsplit = llm_nsplit(ctx); a.ws = split; a.ws_bytes = split_bytes;
            }
            { PROF(COGS_PROF_ATTN); COGS_TRY(cogs_k_attention(st, a)); }
        }
        {
            CogsGemm g; g.dtype = dt;
            g.A = att; g.lda = qd_q; g.W = L.o_w; g.ldw = qd_q; g.C = x; g.ldc = H;
            g.residual = xin; g.ldr = H; g.M = S; g.N = H; g.K = qd_q;
            { PROF(COGS_PROF_GEMM); COGS_TRY(cogs_k_gemm(st, g)); }
        }
        if (!fuse_norm) { PROF(COGS_PROF_NORM); COGS_TRY(cogs_k_rmsnorm(st, dt, x, ln, L.post_ln, S, H, w.rms_eps)); }
        {
            CogsGemm g; g.dtype = dt;
            g.A = fuse_norm ? x : ln; g.lda = H; g.W = L.gu_w; g.ldw = H; g.C = act; g.ldc = w.inter;
            g.M = S; g.N = 2 * w.inter; g.K = H; g.act = COGS_ACT_SWIGLU;
            if (fuse_norm) { g.rms_gamma = L.post_ln; g.rms_eps = w.rms_eps; }
            { PROF(COGS_PROF_GEMM); COGS_TRY(cogs_k_gemm(st, g)); }
        }
        {
            CogsGemm g; g.dtype = dt;
            g.A = act; g.lda = w.inter; g.W = L.down_w; g.ldw = w.inter; g.C = x; g.ldc = H;
            g.residual = x; g.ldr = H; g.M = S; g.N = H; g.K = w.inter;
            { PROF(COGS_PROF_GEMM); COGS_TRY(cogs_k_gemm(st, g)); }
        }
    }
    if (kv) kv->len = ctx;
    if (nseg > 0) {   // final norm over every row, then one mean per sequence
        if (!pooled_seg) return COGS_E_INVALID;
        { PROF(COGS_PROF_NORM); COGS_TRY(cogs_k_rmsnorm(st, dt, x, ln, w.final_norm, S, H, w.rms_eps)); }
        for (int sgm = 0; sgm < nseg; ++sgm) {
            PROF(COGS_PROF_OTHER);
            COGS_TRY(cogs_k_mean_rows(st, dt, (const char*)ln + (size_t)cu_host[sgm] * H * es, H,
                                      cu_host[sgm + 1] - cu_host[sgm], H, pooled_seg + (size_t)sgm * H));
        }
        return COGS_OK;
    }
    const bool need_all = pooled_mean || hidden_out;
    if (need_all) {
        void* hn = hidden_out ? hidden_out : ln;
        { PROF(COGS_PROF_NORM); COGS_TRY(cogs_k_rmsnorm(st, dt, x, hn, w.final_norm, S, H, w.rms_eps)); }
        if (pooled_mean) { PROF(COGS_PROF_OTHER); COGS_TRY(cogs_k_mean_rows(st, dt, hn, H, S, H, pooled_mean)); }
        if (last_logits) {
            CogsGemm g; g.dtype = dt;
            g.A = (char*)hn + (size_t)(S - 1) * H * es; g.lda = H; g.W = w.lm_head; g.ldw = H; g.C = last_logits;
            g.ldc = w.vocab; g.M = 1; g.N = w.vocab; g.K = H; g.out_f32 = 1;
            { PROF(COGS_PROF_GEMM); COGS_TRY(cogs_k_gemm(st, g)); }
        }
    } else if (last_logits) {
        CogsGemm g; g.dtype = dt;
        g.rms_gamma = w.final_norm; g.rms_eps = w.rms_eps;   // final RMSNorm fused into the lm_head GEMV
        g.A = (char*)x + (size_t)(S - 1) * H * es; g.lda = H; g.W = w.lm_head; g.ldw = H; g.C = last_logits; g.ldc = w.vocab;
        g.M = 1; g.N = w.vocab; g.K = H; g.out_f32 = 1;
        { PROF(COGS_PROF_GEMM); COGS_TRY(cogs_k_gemm(st, g)); }
    }
    return COGS_OK;
}

cogs_status cogs_llm_forward(cogs_handle h, cogs_stream stream, const void* embeds, int S, cogs_kv* kv,
                             float* last_logits, float* pooled_mean, void* hidden_out, void* ws, size_t ws_bytes) {
    return llm_forward_impl(h, stream, embeds, S, kv, last_logits, pooled_mean, hidden_out, ws, ws_bytes, nullptr, 0, nullptr);
}

cogs_status cogs_llm_forward_segments(cogs_handle h, cogs_stream stream, const void* embeds, int S,
                                      const int32_t* cu_seqlens_host, int nseg, float* pooled_means, void* ws,
                                      size_t ws_bytes) {
    if (nseg <= 0 || !cu_seqlens_host || !pooled_means) return COGS_E_INVALID;
    return llm_forward_impl(h, stream, embeds, S, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes, cu_seqlens_host, nseg,
                            pooled_means);
}

}  // extern "C"
