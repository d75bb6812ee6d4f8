// Diagnostic switches of libcogs_hip.so: ONE table, changed only through cogs_debug_set (include/cogs.h).
//
// Nothing in the library reads the environment. Every entry below has the SHIPPED behaviour as its default; the
// measurement tools (bench.py --debug, tools/*) and the A/B tests flip an entry by name, run, and flip it back -- in one
// process, so both sides of a comparison run on the same device (rounds 1-4 read ~28 COGS_* environment variables into
// function-local statics spread over the kernel files; a value could only be chosen before the first call).
// The table is process-wide. Its entries are relaxed atomics: another host thread driving its own handle may read a switch (or
// write one of the two last_* reports) while this one sets it without a data race -- which value a call running at that moment
// sees is still unspecified, so set a switch between calls. cogs_debug_set checks the value against the entry's range
// (csrc/capi.hip) and warns on stderr when a timing-only switch that changes results is turned on.
#pragma once
#include <atomic>

//   X(name, default, meaning)
#define COGS_DEBUG_SWITCHES(X)                                                                                              \
    /* ---- GEMM (csrc/gemm.hip) ---- */                                                                                    \
    X(gemm_pp64, 1, "1: whole-line ping-pong kernel (gemm_tn_pp64_kernel); 0: the 32-wide K-tile body it replaced")         \
    X(gemm_tall, 1, "whole-line kernel: 1 a ragged column block of <= 128 columns as 384x128 tiles (round 6), 0 as padded 256x256 tiles") \
    X(gemm_even_grid, 0, "whole-line kernel: 1 the fewest persistent workgroups that keep the round count (every workgroup the same number of tiles), 0 always 256") \
    X(gemm_pingpong, 1, "0: never take a ping-pong kernel (256x128 ring / 128x128 kernels only)")                          \
    X(gemm_small, 0, "1: always the 128x128 kernel")                                                                        \
    X(gemm_wgs, 256, "persistent workgroups of the 256x128 ring kernel (0: one tile per workgroup)")                        \
    X(gemm_pad_pct, 112, "ping-pong kernel is taken while N padded to 256 <= this percentage of N")                         \
    X(gemm_rope_lut, 1, "K-tile body only: rotary factors from the LDS position LUT when the caller supplies one")          \
    X(gemm_group_m, 0, "> 0: row blocks per group of the ping-pong tile walk (0: chosen per shape)")                        \
    X(gemm_split, 1, "0: no round-aligned / few-tile split of a launch")                                                    \
    X(gemm_co_streams, 0, "> 0: overrides the co-running-streams hint of the few-tile choice")                              \
    X(gemm_ring_cost_permille, 720, "few-tile choice: cost of a round of ring tiles relative to a round of ping-pong tiles") \
    X(gemm_epi_serial, 0, "1: the two wave groups' epilogues in separate intervals (the order before round 4)")             \
    X(gemm_nostore, 0, "1: timing only -- K loops without epilogues (results are NOT written)")                             \
    X(gemm_trace, 0, "1: per-tile s_memtime stamps of workgroup 0 printed to stderr (synchronises)")                        \
    X(gemm_choice, 0, "1: print which body every ping-pong-eligible shape gets")                                            \
    X(gemm_headmajor, 1, "ViT encode: QKV GEMM writes q/k/v head-major [which][head][row][hd] for the attention kernel")    \
    X(gemv_small_n, 1024, "GEMV: 2 output rows per wave while N / 16 is below this (else 4)") \
    X(gemv_dot2, 1, "bf16 GEMV (decode): 1 products on v_dot2c_f32_bf16 from the packed registers (round 5), 0 unpack + fp32 fma per element") \
    /* ---- attention (csrc/attn.hip, attn_vit.hip, attn_decode.hip) ---- */                                                \
    X(attn_vit, 2, "ViT block-diagonal attention: 2 pipelined LDS-DMA kernel, 1 unpipelined, 0 general kernel")            \
    X(attn_vit_early, 0, "1: pipelined ViT kernel issues tiles 1-2 before its first wait (the order before round 4)")       \
    X(attn_vit_len, 1, "ViT attention: 0 the run-time form of the ragged end also for one-video launches (round 6: its shape as template parameters)") \
    X(attn_uniform_hint, 0, "tests: > 0 = cogs_attention treats every cu_seqlens segment as exactly this many rows (the hint cogs_vit_encode derives from the grid itself)") \
    X(attn_decode, 1, "0: single-token attention through the general split-KV kernel")                                      \
    X(attn_combine32, 1, "split-KV combine: 1 one block per (head, 32-column slice), 8 loads in flight per thread; 0 one block per head") \
    X(attn_prefill_dma, 1, "0: Qwen2 prompt attention through the general register-staged kernel")                          \
    X(attn_prefill_deep, 1, "prompt LDS-DMA kernel: 1 fragment reads ordered 6-8 ahead of their MFMAs + running maximum deferred to 2^6 (round 5: 1.82 -> 1.64 ms), 0 the round-4 kernel") \
    X(attn_prio, 2, "prompt attention wave priorities: 0 none, 1 MFMA phases raised, 2 softmax phase raised")               \
    X(attn_light_first, 0, "1: causal query tiles issued lightest first")                                                   \
    X(attn_nq, 0, "general kernel: 1 / 2 forces 16 / 32 query rows per wave")                                              \
    /* ---- encoder / LLM drivers (csrc/capi.hip), k-means ---- */                                                          \
    X(vit_split_max, 1099511627776LL, "clips above this many patches are encoded on one stream")                           \
    X(llm_split_keys, 0, "> 0: keys per split of the decode attention (0: chosen from the context)")                        \
    X(km_row_groups, 0, "> 0: row groups of the k-means distance pass (0: sized to fill the chip)")

struct CogsDebug {
#define COGS_DBG_FIELD(name, dflt, doc) std::atomic<long long> name{dflt};
    COGS_DEBUG_SWITCHES(COGS_DBG_FIELD)
#undef COGS_DBG_FIELD
    // read-only reports (cogs_debug_get): what the last cogs_gemm of this process dispatched to
    // 0 none yet, 1 128x128, 2 256x128 ring, 3 K-tile ping-pong, 4 whole-line ping-pong, 5 ping-pong + ring (split), 6 GEMV
    std::atomic<long long> gemm_last_body{0};
    // ... and the last cogs_attention: 1 general MFMA kernel, 2 ViT unpipelined, 3 ViT pipelined (row-major K/V), 4 single-token
    // decode (+ combine), 5 prompt LDS-DMA kernel, 7 row-wise fp32 kernel, 8 ViT pipelined, head-major K/V (6 and 9 were the archived
    // ping-pong / 64-rows-per-wave prompt kernels: tools/experiments/attn_prefill_variants.hip)
    std::atomic<long long> attn_last_kernel{0};
    // ... and which ragged end the last pipelined ViT launch had: 10 R + blocks of the last tile (41 .. 72), 0 = the run-time form
    std::atomic<long long> attn_vit_last_end{0};
};
extern CogsDebug g_cogs_debug;     // capi.hip
