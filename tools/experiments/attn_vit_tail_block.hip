// ARCHIVED EXPERIMENT (round 6) -- not compiled into the library. The key-split form of the ViT attention's short last query block,
// as it stood in csrc/attn_vit.hip (called at the top of attn_vit_pipe_kernel: `if (p.split_tail && qe - q0 <= 32) {
// attn_vit_tail_block<HD>(p, smem, head, qs, qe, q0); return; }`, debug switch attn_vit_tail). Correct (it passed the edge tests of
// tests/test_gpu_ops.py plus a test of its own against the fp32 softmax, both K/V layouts, mixed segments) and step-neutral:
// profiles/r6_attn_vit_tail.txt holds the measurements and why it cannot win -- a workgroup that must pull a (frame, head)'s whole
// K and V (266 KB) through 74 KiB of LDS is bound by memory latency x bytes / bytes in flight (13 000 ticks for the walk alone), and
// its four busy waves slow the ordinary workgroup that shares the CU by what the shorter lifetime gives back.
// Needs the definitions of csrc/attn_vit.hip (VitAttnArgs with an `int split_tail`, f32x16, RESCALE_THR, bf16_round, LIFE_NOW / LIFE_ADD).

// ---------------------------------------------------------------------------------------------------------------
// Round 6: the ragged last query block, split over the KEYS.
// 924 patches are 7 query blocks of 128 rows and one of 28: as an ordinary workgroup that eighth block computes on ONE wave
// (three only stage) yet holds its half CU for a whole workgroup time -- skipping it altogether takes 11 % off the launch
// (profiles/r6_attn_vit_tail.txt) where its rows are 3 % of the work. Here a query block of <= 32 rows goes to a workgroup whose
// four waves all own the SAME rows and every fourth key tile each (tile t -> wave t % 4): 4 instead of 15 tiles deep, no
// workgroup barrier inside the walk, one merge through LDS at the end. Called by the pipelined kernel for such a block
// (p.split_tail), in the SAME launch: as a launch of its own behind the main kernel it re-read every K and V from HBM (272 MB per
// layer at cfg2, the other query blocks of its (frame, head) long gone from the L2) and the attention got 8 % SLOWER.
//   * a wave is a complete little attention of its own over 32-key blocks: K and V blocks (32 rows x 144 bytes, the plain image
//     the pipelined kernel reads) go global -> LDS by DMA into the wave's PRIVATE buffers, two slots each, so the block after
//     next is requested as soon as a slot's fragments are in registers: a whole tile ahead, and the only vector-memory
//     instructions in the walk are those DMA pieces -- every vmcnt wait is counted by hand, the wave waits on nobody else;
//   * same products, pad-column bookkeeping and deferred maximum as above; the partial results (O^T relative to the wave's
//     reference, the reference, the denominator row) meet in LDS and are combined as sum_w 2^(m_w - m) O_w, m = max m_w.
// LDS: 4 waves x 4 x 4 608 B + the two constant chunks, inside the pipelined kernel's request (two workgroups per CU).
template <int HD>
__device__ __forceinline__ void attn_vit_tail_block(const VitAttnArgs& p, char* const smem, const int head, const int qs, const int qe, const int q0) {
    constexpr int NW = 4;
    static_assert(HD % 8 == 0 && HD % 16 == 8 && HD < 96, "pad column HD must open a fresh 16-byte chunk inside the last k-step");
    constexpr int KS = (HD + 8) / 16, DB = (HD + 8 + 31) / 32, CH = HD / 8;
    constexpr int RS = HD * 2, BLK = 32 * RS;                    // a 32-key block image: 4 608 B
    constexpr int NP = (BLK + 1023) / 1024;                      // DMA pieces per block: 4 whole KiB + one that ends with the block
    constexpr int WAVE_LDS = 4 * BLK;                            // [K slot 0 | K slot 1 | V slot 0 | V slot 1]; later the merge record
    constexpr int C_ONE = NW * WAVE_LDS, C_ZERO = C_ONE + 16;
    constexpr int NACC = DB * 16;                                // accumulator registers per lane
    static_assert(BLK >= 1024 && (NACC + 1) * 256 <= WAVE_LDS, "piece layout; merge record fits the wave's buffers");
    static_assert(NW * WAVE_LDS + 64 <= 4 * (2 * 64 * RS + 64), "fits the pipelined kernel's LDS request");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r32 = lane & 31;
    const int len = qe - qs;
    const int nt = (len + 63) >> 6;
    unsigned long long lf0 = 0, lf1 = 0, lf2 = 0, lf3 = 0;
    LIFE_NOW(lf0);

    if (tid == 0) {
        *reinterpret_cast<u32x4*>(smem + C_ONE) = u32x4{0x00003f80u, 0, 0, 0};
        *reinterpret_cast<u32x4*>(smem + C_ZERO) = u32x4{0, 0, 0, 0};
    }
    const int qrow = q0 + r32;
    const bool qok = qrow < qe;
    u32x4 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 16 * s + 8 * h;
        qf[s] = u32x4{0, 0, 0, 0};
        if (qok && k < HD) qf[s] = *reinterpret_cast<const u32x4*>(p.Q + (long)qrow * p.ldq + head * p.head_stride + k);
    }

    const bf16_t* kbase = p.K + (long)qs * p.ldk + head * p.head_stride;
    const bf16_t* vbase = p.V + (long)qs * p.ldv + head * p.head_stride;
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    char* const my_lds = smem + wid * WAVE_LDS;
    auto uniform_ptr = [](const bf16_t* q) -> const bf16_t* {
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const bf16_t*)(((unsigned long long)hi << 32) | lo);
    };
    auto dma16 = [&](const bf16_t* base, int off_bytes, unsigned lds) {      // see attn_vit_pipe_kernel (M0, hidden from hipcc)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(lds), "v"(off_bytes), "s"(base) : "memory");
    };
    // The wave's j-th block: half j & 1 of its tile wid + NW (j >> 1), i.e. block g = 2 tile + half of the segment; it exists while
    // 32 g < len (monotone in j). Keys past the end read the last row (finite; their scores are masked).
    auto blk_of = [&](int j) -> int { return 2 * (wid + NW * (j >> 1)) + (j & 1); };
    auto exists = [&](int j) -> bool { return 32 * blk_of(j) < len; };
    // one matrix block -> LDS: NP pieces of 1 KiB, the last one placed so that it ENDS with the block (it repeats half of its
    // predecessor: the same bytes twice); chunk c = (row c / 9, 16-byte column c % 9) of the image
    auto issue_blk = [&](const bf16_t* base, long ld, int g, unsigned lds) {
        const int valid = len - 32 * g;                          // >= 1
        const bf16_t* src = uniform_ptr(base + (long)g * 32 * ld);
        int ln = lane;
        asm volatile("" : "+v"(ln));      // the piece offsets are computed HERE, every block (hoisted out of the loop they get spilled)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int start = i + 1 < NP ? 1024 * i : BLK - 1024;
            const int c = start / 16 + ln;
            const int row = min(c / CH, valid - 1);
            dma16(src, (row * (int)ld + (c % CH) * 8) * 2, lds + start);
        }
    };
    const unsigned my_lds_a = __builtin_amdgcn_readfirstlane(smem_lds + wid * WAVE_LDS);
    auto issue_k = [&](int j) { issue_blk(kbase, p.ldk, blk_of(j), my_lds_a + (j & 1) * BLK); };
    auto issue_v = [&](int j) { issue_blk(vbase, p.ldv, blk_of(j), my_lds_a + (2 + (j & 1)) * BLK); };
    // ... and of a FULL block (every block of the steady walk): the piece offsets are lane constants, kept in registers (computed
    // per block they were ~100 vector instructions of a block's ~250)
    int pk_off[NP], pv_off[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = (i + 1 < NP ? 1024 * i : BLK - 1024) / 16 + lane;
        pk_off[i] = ((c / CH) * (int)p.ldk + (c % CH) * 8) * 2;
        pv_off[i] = ((c / CH) * (int)p.ldv + (c % CH) * 8) * 2;
    }
    auto issue_full = [&](const bf16_t* base, long ld, const int (&off)[NP], int g, unsigned lds) {
        const bf16_t* src = uniform_ptr(base + (long)g * 32 * ld);
#pragma unroll
        for (int i = 0; i < NP; ++i) dma16(src, off[i], lds + (i + 1 < NP ? 1024 * i : BLK - 1024));
    };
    auto issue_k_full = [&](int j) { issue_full(kbase, p.ldk, pk_off, blk_of(j), my_lds_a + (j & 1) * BLK); };
    auto issue_v_full = [&](int j) { issue_full(vbase, p.ldv, pv_off, blk_of(j), my_lds_a + (2 + (j & 1)) * BLK); };

    // per-lane LDS read offsets inside a block image (as in the pipelined kernel)
    const int k_rd = r32 * RS + h * 16;
    const int v_d = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int v_rd = (4 * h + ((lane & 15) >> 2)) * RS + v_d * 2;
    constexpr int DL = 32 * (DB - 1);
    const bool v_last_real = DL + v_d < HD;
    const int v_last_const = DL + v_d == HD ? C_ONE : C_ZERO;
    auto read_k = [&](const char* img, u32x4 (&kf)[KS]) {
#pragma unroll
        for (int ks = 0; ks < KS - 1; ++ks) kf[ks] = *reinterpret_cast<const u32x4*>(img + k_rd + ks * 32);
        kf[KS - 1] = *reinterpret_cast<const u32x4*>(h ? smem + C_ONE : img + k_rd + (KS - 1) * 32);
    };
    auto read_v = [&](const char* img, int b, int s2) -> u32x4 {
        const char* a0 = img + v_rd + 16 * s2 * RS + b * 64;
        const char* a1 = a0 + 8 * RS;
        if (b == DB - 1) {
            a0 = v_last_real ? a0 : smem + v_last_const;
            a1 = v_last_real ? a1 : smem + v_last_const;
        }
        const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(a0));
        const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(a1));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        return u32x4{l2[0], l2[1], h2[0], h2[1]};
    };

    f32x16 oacc[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[b][r] = 0.f;
    float sh = 0.f;
    bool first = true;

    // blocks 0 and 1 on their way behind the Q loads (hipcc's own wait for Q, a vmcnt(0), covers them as well)
    if (exists(0)) { issue_k(0); issue_v(0); }
    if (exists(1)) { issue_k(1); issue_v(1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qf[s]));          // Q loads complete: no compiler wait inside the walk
    __syncthreads();                                             // the constant chunks (every wave comes by here exactly once)

    // STEADY: blocks j + 1 and j + 2 exist, block j is full. Vector-memory queue, oldest first, at the top: K(j) V(j) K(j+1) V(j+1)
    auto block = [&](const int j, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        const char* kimg = my_lds + (j & 1) * BLK;
        const char* vimg = my_lds + (2 + (j & 1)) * BLK;
        if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * NP) : "memory");      // K(j) landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        u32x4 kf[KS];
        read_k(kimg, kf);
        f32x16 sc;
        auto qk = [&]() {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                f32x16 c0;
#pragma unroll
                for (int r = 0; r < 16; ++r) c0[r] = 0.f;
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[s]), __builtin_bit_cast(bf16x8, qf[s]),
                                                             s == 0 ? c0 : sc, 0, 0, 0);
            }
            if constexpr (!STEADY) {                             // the segment's last block may be ragged
                int vh = len - 32 * blk_of(j) - 4 * h;           // key = (r & 3) + 8 (r >> 2) + 4 h >= valid, as ONE per-lane bound against
                asm volatile("" : "+v"(vh));                     // constants (sixteen per-lane key numbers get hoisted and spilled)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) >= vh) sc[r] = -INFINITY;
            }
        };
        qk();
        float d = sc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) d = fmaxf(d, sc[r]);
        {
            const unsigned db = __builtin_bit_cast(unsigned, d);
            const auto sw = __builtin_amdgcn_permlane32_swap(db, db, false, false);
            d = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
        }
        if (first || __any(d > RESCALE_THR)) {                   // wave-uniform: move the reference, rescale O, recompute the block's scores
            float dd = first ? d : fmaxf(d, 0.f);
            if (!(dd > -INFINITY)) dd = 0.f;
            const float m_new = (first || d > RESCALE_THR) ? bf16_round(sh + dd) : sh;
            if (!first) {
                const float al = __builtin_amdgcn_exp2f(sh - m_new);
#pragma unroll
                for (int b = 0; b < DB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[b][r] *= al;
            }
            sh = m_new;
            const unsigned bits = __float_as_uint(-sh) >> 16;    // exact: sh is a bf16 value
            if (h) qf[KS - 1][0] = (qf[KS - 1][0] & 0xffff0000u) | bits;
            first = false;
            qk();
        }
        if constexpr (STEADY) {
            if (exists(j + 3)) issue_k_full(j + 2); else issue_k(j + 2);      // the K slot's fragments are in registers
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * NP) : "memory");      // V(j) landed; behind it K(j+1) V(j+1) K(j+2)
        }
        u32x4 vf[DB][2];
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) vf[b][s2] = read_v(vimg, b, s2);
        u32x4 pf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int w = 0; w < 4; ++w)
                pf[s2][w] = pack_bf2(__builtin_amdgcn_exp2f(sc[8 * s2 + 2 * w]), __builtin_amdgcn_exp2f(sc[8 * s2 + 2 * w + 1]));
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int b = 0; b < DB; ++b)
                oacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[b][s2]), __builtin_bit_cast(bf16x8, pf[s2]),
                                                                 oacc[b], 0, 0, 0);
        if constexpr (STEADY) { if (exists(j + 3)) issue_v_full(j + 2); else issue_v(j + 2); }      // the V slot's fragments are in registers
    };
    LIFE_NOW(lf1);
    int j = 0;
    for (; exists(j + 2); ++j) block(j, std::true_type{});
    for (; exists(j); ++j) block(j, std::false_type{});          // the last two: everything has been requested, one plain wait each
    LIFE_NOW(lf2);

    // ---- merge: every wave leaves [register][lane] fp32 records of O^T and its reference in its own (now idle) buffers
    float* rec = reinterpret_cast<float*>(my_lds);
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) rec[(b * 16 + r) * 64 + lane] = oacc[b][r];
    rec[NACC * 64 + lane] = first ? -INFINITY : sh;              // a wave without a block (fewer tiles than waves) weighs nothing
    __syncthreads();
    float mw[NW], m = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        mw[w] = reinterpret_cast<const float*>(smem + w * WAVE_LDS)[NACC * 64 + lane];
        m = fmaxf(m, mw[w]);
    }
    float fw[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) fw[w] = __builtin_amdgcn_exp2f(mw[w] - m);      // wave 0 always has block 0: m is finite
    constexpr int LB = HD / 32, LR = HD % 32;
    constexpr int LH = (LR >> 2) & 1, LREG = (LR & 3) + 4 * (LR >> 3);
    float l = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) l += fw[w] * reinterpret_cast<const float*>(smem + w * WAVE_LDS)[(LB * 16 + LREG) * 64 + r32 + 32 * LH];
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    bf16_t* orow = p.O + (long)qrow * p.ldo + head * HD;
    // output groups (d-block b, half gp): 16 consecutive d; wave w takes group w, wave 0 the fifth as well
    constexpr int NGRP = (HD + 15) / 16;
    for (int g = wid; g < NGRP; g += NW) {
        const int b = g >> 1, gp = g & 1;
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) a += fw[w] * reinterpret_cast<const float*>(smem + w * WAVE_LDS)[(b * 16 + 8 * gp + i) * 64 + lane];
            o[i] = a * inv;
        }
        const unsigned e0 = pack_bf2(o[0], o[1]), e1 = pack_bf2(o[2], o[3]), o0 = pack_bf2(o[4], o[5]), o1 = pack_bf2(o[6], o[7]);
        const auto s0 = __builtin_amdgcn_permlane32_swap(e0, o0, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(e1, o1, false, false);
        const int d0 = 32 * b + 16 * gp + 8 * h;
        if (qok && d0 < HD) *reinterpret_cast<u32x4*>(orow + d0) = u32x4{(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};
    }
    LIFE_NOW(lf3);
    LIFE_ADD(2, lf3 - lf0); LIFE_ADD(3, 1); LIFE_ADD(4, lf1 - lf0); LIFE_ADD(5, lf2 - lf1); LIFE_ADD(6, lf3 - lf2);
}

