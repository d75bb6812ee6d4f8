// ARCHIVE (not built, not shipped): two prompt-attention kernels that were built, tested at full size and measured SLOWER than
// the shipped attn_prefill_dma_kernel<1> (csrc/attn.hip) in rounds 4-5, kept here for their measurements' sake:
//   attn_prefill64_kernel   64 query rows per wave, one wave per SIMD: 2.94 vs 1.97 ms per layer at 15 395 tokens
//                           (profiles/r5_attn_prefill64_ab.txt)
//   attn_prefill_pp_kernel  ping-pong form, bit-identical to attn_prefill_dma_kernel<0>: 1.89 vs 1.81-1.86 ms
//                           (profiles/r5_prefill_attn_anatomy.txt)
// They were members of csrc/attn.hip's anonymous namespace (AttnArgs, dma16, colgroup_max, k_swz ... come from there) behind
// the debug switches attn_prefill64 / attn_prefill_pp until round 6; the round-5 tree (git: 5a24324) builds and tests them.
// ---------------------------------------------------------------------------------------------------------------
// Qwen2 prompt attention with 64 query rows per wave (round 5): the construction of the encoder's pipelined kernel
// (attn_vit.hip) at head dim 128. The kernel above reads every K / V fragment from LDS for 32 query rows -- 48 LDS reads
// per 1 024 cycles of MFMA, each wave's softmax in front of its own PV product -- and sits at ~50 % of the matrix pipe
// whatever the staging (register-staged, LDS-DMA, ping-pong: DESIGN.md 4.4). Here
//   * a workgroup is 4 waves x 64 query rows = 256 rows of one (query head, sequence), ONE wave per SIMD with the whole
//     512-register file: O^T 2 x 4 x 16 accumulators, Q fragments of both 32-row blocks (64), two score sets (64);
//   * v_mfma_f32_32x32x16_bf16, S^T[key][q] = K.Q^T and O^T[d][q] += V^T[d][key].P^T[key][q] as in attn_vit.hip (the score
//     accumulator registers of a lane ARE the B fragment of the PV product), every K and V^T fragment feeding the MFMAs
//     of BOTH 32-row blocks: 24 LDS reads per 1 024 MFMA cycles;
//   * software pipeline at 32-key blocks: sub-step j = {P(j) = exp2(S(j)), row sums} beside {S(j+1) = K(j+1).Q^T, 16
//     MFMAs}, then {O += V(j)^T.P(j), 16 MFMAs} beside {max over S(j+1), K fragments of block j+2};
//   * the reference maximum is the accumulators' initial value (-m: no per-score subtraction) and moves only when a
//     block exceeds it by more than 2^6 (wave-uniform rare path: O and l rescaled, the pending scores shifted);
//   * K / V tiles (64 keys x 256 B each) by LDS-DMA into a ring of four 32 KiB slots, three tiles ahead, one counted
//     vmcnt + one barrier per tile; bank swizzles on the SOURCE side: K chunk ^= row & 15 (conflict-free ds_read_b128 over
//     the 32 keys of an A fragment), V chunk ^= (row & 3) << 2 (the four rows a transposing read touches per half wave
//     fall into four different 64-byte bank quarters);
//   * causal / key-range masks only on the blocks that need them (the last five tiles of a workgroup at most).
// STATUS (round 5): correct (tests/test_gpu_fullsize.py runs it at 15 395 tokens and in the 19-sequence form against fp32
// torch), measured, OFF by default (debug switch attn_prefill64 = 1): 2.94 ms per layer against 1.97 ms for the kernel
// above on the same box (tools/attn_prefill_ab.py). What the disassembly says: with one wave per SIMD the register file is
// 256 vector + 256 accumulator registers and hipcc decides what lives where -- it keeps the Q fragments in vector
// registers and both score sets in accumulator registers, so every 32-key block pays 32 v_accvgpr_write (the -m initial
// values) and 32 v_accvgpr_read (scores back for the exponentials): ~2 300 vector-issue cycles per 64-key tile against
// 2 048 of MFMA; and 30 spilled registers, 14 of them reloaded inside the tile loop -- scratch loads count on vmcnt, so
// the compiler's wait for them (vmcnt(0)) drains the three-tile LDS-DMA prefetch in every sub-step, with no second wave
// on the SIMD to cover it. Pinning Q or O to the accumulator half with "a"-constrained asm made the allocation worse (417
// -- 602 spills); only the rare-path O rescale written through single v_accvgpr_read / write pairs helped (393 -> 30).
// What it needs is the guide's form: O, Q (and the -m splat) in asm-owned accumulator ranges and the MFMAs issued from
// inline asm with literal registers, the softmax alone in compiler-allocated vector registers.
__global__ __launch_bounds__(256, 1) void attn_prefill64_kernel(AttnArgs p) {
    constexpr int HD = 128, KS = 8, DB = 4, NQ = 2, NS = 4;
    constexpr int RS = 256, TILE = 64 * RS, STAGE = 2 * TILE;            // 32 KiB per tile: [K | V]
    constexpr float THR = 6.0f;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // NS * STAGE = 128 KiB

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r32 = lane & 31;
    const int seg = blockIdx.z, head = blockIdx.y;
    const int kvh = head / (p.hq / p.hkv);
    int qs = 0, qe = p.q_len, ks = 0, ke = p.kv_len;
    if (p.cu) { qs = p.cu[seg]; qe = p.cu[seg + 1]; ks = qs; ke = qe; }
    const int qt = p.heavy_first ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int q0 = qs + qt * 256;
    if (q0 >= qe) return;
    const int kend = min(ke, ks + (q0 - qs) + 255 + p.q_pos0 + 1);       // causal: nothing beyond the last row's diagonal
    const int len = kend - ks;                                          // >= 1 (q_pos0 >= 0)
    const int nt = (len + 63) >> 6, nblk = (len + 31) >> 5;
    // blocks [0, nfree) need no mask for ANY row of the workgroup: wholly inside the key range and at or left of the
    // first row's diagonal
    int nfree = (q0 - qs) + p.q_pos0 - 31 >= 0 ? ((q0 - qs) + p.q_pos0 - 31) / 32 + 1 : 0;
    nfree = min(nfree, min((ke - ks) >> 5, nblk));

    const bf16_t* Qp = reinterpret_cast<const bf16_t*>(p.Q);
    const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.K);
    const bf16_t* Vp = reinterpret_cast<const bf16_t*>(p.V);

    // Q fragments (B operand of S^T = K.Q^T): lane (q = r32, h) holds Q[q][16 s + 8 h + 0..7] of both 32-row blocks
    int qrow[NQ];
    bool qok[NQ];
    u32x4 qf[NQ][KS];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        qrow[qi] = q0 + wid * 64 + qi * 32 + r32;
        qok[qi] = qrow[qi] < qe;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qf[qi][s] = u32x4{0, 0, 0, 0};
            if (qok[qi]) qf[qi][s] = *reinterpret_cast<const u32x4*>(Qp + (long)qrow[qi] * p.ldq + head * HD + 16 * s + 8 * h);
        }
    }
    const bool wave_active = q0 + wid * 64 < qe;

    // ---- staging: a tile is 16 + 16 pieces of 1 KiB (4 rows x 256 B); wave w issues pieces w, w + 4, w + 8, w + 12 of K and V
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    int st_row[4], k_off[4], v_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * (wid + 4 * i) + (lane >> 4);
        st_row[i] = row;
        k_off[i] = (row * (int)p.ldk + kvh * HD + ((lane & 15) ^ (row & 15)) * 8) * 2;          // bytes from the tile's first row
        v_off[i] = (row * (int)p.ldv + kvh * HD + ((lane & 15) ^ ((row & 3) << 2)) * 8) * 2;
    }
    auto uniform_ptr = [](const bf16_t* q) -> const bf16_t* {
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const bf16_t*)(((unsigned long long)hi << 32) | lo);
    };
    auto dma16 = [&](const bf16_t* base, int off_bytes, unsigned lds) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(lds), "v"(off_bytes), "s"(base) : "memory");
    };
    const bf16_t* k_next = Kp + (long)ks * p.ldk;
    const bf16_t* v_next = Vp + (long)ks * p.ldv;
    const long k_step = 64 * p.ldk, v_step = 64 * p.ldv;
    auto issue_tile = [&](int t) {                                       // tiles are issued strictly in order
        const int valid = ke - (ks + t * 64);                            // >= 1
        const bf16_t* kb = uniform_ptr(k_next);
        const bf16_t* vb = uniform_ptr(v_next);
        k_next += k_step; v_next += v_step;
        const unsigned st = __builtin_amdgcn_readfirstlane(smem_lds + (t & (NS - 1)) * STAGE);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ko = k_off[i], vo = v_off[i];
            if (valid < 64) {                                            // rows past the key range repeat its last row
                const int back = st_row[i] - min(st_row[i], valid - 1);
                ko -= back * (int)p.ldk * 2;
                vo -= back * (int)p.ldv * 2;
            }
            dma16(kb, ko, st + (wid + 4 * i) * 1024);
            dma16(vb, vo, st + TILE + (wid + 4 * i) * 1024);
        }
    };
    auto wait_tiles = [&](int newer) {                                   // all but the `newer` newest tiles of this wave landed
        if (newer <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    };

    // ---- per-lane LDS read offsets inside a stage
    // K fragment of k-step s: lane (key r32, h) reads logical chunk 2 s + h of row 32 kb + r32, stored at chunk ^ (row & 15)
    int kx[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) kx[s] = r32 * RS + (((2 * s + h) ^ (r32 & 15)) << 4);
    // V^T fragment: lane supplies the 8-byte piece (key 4 h + q4 [+ 8], d = 32 b + 16 ((lane >> 4) & 1) + 4 (lane & 3) .. + 3);
    // logical chunk 4 b + 2 ((lane >> 4) & 1) + ((lane & 3) >> 1), stored at chunk ^ (q4 << 2)
    const int q4 = (lane & 15) >> 2;
    int vx[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b)
        vx[b] = TILE + (4 * h + q4) * RS + (((4 * (b ^ q4)) + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1)) << 4) + 8 * (lane & 1);

    f32x16 oacc[NQ][DB];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qi][b][r] = 0.f;
    float sh[NQ] = {0.f, 0.f}, l_run[NQ] = {0.f, 0.f};

    auto stage = [&](int t) -> const char* { return smem + (t & (NS - 1)) * STAGE; };
    auto read_k = [&](const char* st, int kb, u32x4 (&kf)[KS]) {
#pragma unroll
        for (int s = 0; s < KS; ++s) kf[s] = *reinterpret_cast<const u32x4*>(st + kx[s] + kb * 32 * RS);
    };
    auto read_v = [&](const char* st, int b, int kb, int s2) -> u32x4 {
        const char* a0 = st + vx[b] + (32 * kb + 16 * s2) * RS;
        const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(a0));
        const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(a0 + 8 * RS));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        return u32x4{l2[0], l2[1], h2[0], h2[1]};
    };
    auto qk_blk = [&](const u32x4 (&kf)[KS], f32x16& sx, const int qi) {           // S - m of one (key block, row block)
        f32x16 c0;
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = -sh[qi];
#pragma unroll
        for (int s = 0; s < KS; ++s)
            sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[s]), __builtin_bit_cast(bf16x8, qf[qi][s]),
                                                         s == 0 ? c0 : sx, 0, 0, 0);
    };
    auto mask_blk = [&](f32x16& sx, const int jb, const int qi) {                   // keys outside the range / right of the diagonal
        const int lim = min(ke - ks - 1, (qrow[qi] - qs) + p.q_pos0);               // last visible key of this lane's row, relative
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * jb + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (key > lim) sx[r] = -INFINITY;
        }
    };
    auto own_max = [&](const f32x16& sx) -> float {
        float d = sx[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) d = fmaxf(d, sx[r]);
        return d;
    };
    auto pair_max = [&](float d) -> float {
        const unsigned db = __builtin_bit_cast(unsigned, d);
        const auto sw = __builtin_amdgcn_permlane32_swap(db, db, false, false);
        return fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
    };

    // ---- prologue
    issue_tile(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qf[qi][s]));          // Q loads complete before the loop
    __builtin_amdgcn_s_barrier();
    if (nt > 1) issue_tile(1);
    if (nt > 2) issue_tile(2);

    f32x16 sa[NQ], sb[NQ];
    u32x4 kf[KS];                                                                  // K fragments of the NEXT block to be multiplied
    if (wave_active) {
        read_k(stage(0), 0, kf);
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            qk_blk(kf, sa[qi], qi);                                                // shift 0
            if (nfree == 0) mask_blk(sa[qi], 0, qi);
            float d = pair_max(own_max(sa[qi]));
            if (!(d > -INFINITY)) d = 0.f;                                         // a row that sees no key of block 0 (never: key 0 is visible)
            sh[qi] = d;
#pragma unroll
            for (int r = 0; r < 16; ++r) sa[qi][r] -= d;
        }
        read_k(stage(0), 1, kf);
    }

    // sub-step j: consumes sc = S(j) - m, produces sn = S(j+1) - m from kf, leaves the fragments of block j+2 in kf.
    // KIND 1: block j+1 needs no mask; 2: it does; 0: j is the last block
    auto substep = [&](f32x16 (&sc)[NQ], f32x16 (&sn)[NQ], const int j, auto kind_tag) {
        constexpr int KIND = decltype(kind_tag)::value;
        const char* st_c = stage(j >> 1);
        u32x4 vf[DB][2];
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) vf[b][s2] = read_v(st_c, b, j & 1, s2);
        __builtin_amdgcn_sched_barrier(0);
        // phase 1: S(j+1) beside P(j)
        if constexpr (KIND != 0) {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) qk_blk(kf, sn[qi], qi);
        }
        u32x4 pf[NQ][2];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            float psum = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float p0 = __builtin_amdgcn_exp2f(sc[qi][8 * s2 + 2 * w]), p1 = __builtin_amdgcn_exp2f(sc[qi][8 * s2 + 2 * w + 1]);
                    psum += p0 + p1;
                    pf[qi][s2][w] = pack_bf2(p0, p1);
                }
            l_run[qi] += psum;
        }
        __builtin_amdgcn_sched_barrier(0);
        // phase 2: O += V(j)^T.P(j) beside the maximum of S(j+1) and the K fragments of block j+2
        if constexpr (KIND != 0) read_k(stage((j + 2) >> 1), (j + 2) & 1, kf);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int b = 0; b < DB; ++b)
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi)
                    oacc[qi][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[b][s2]),
                                                                        __builtin_bit_cast(bf16x8, pf[qi][s2]), oacc[qi][b], 0, 0, 0);
        if constexpr (KIND != 0) {
            if constexpr (KIND == 2) {
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) mask_blk(sn[qi], j + 1, qi);
            }
            const float d0 = own_max(sn[0]), d1 = own_max(sn[1]);
            if (__builtin_expect(__any(fmaxf(d0, d1) > THR), 0)) {
                // rare (wave-uniform): move the reference of the rows that need it; O and l, complete up to block j, are
                // multiplied by 2^-(m_new - m), the pending scores shifted by the same amount
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the PV MFMAs' results, read below by inline asm hipcc does not pad for
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    const float d = pair_max(qi == 0 ? d0 : d1);
                    const float up = d > THR ? d : 0.f;
                    const float al = __builtin_amdgcn_exp2f(-up);
                    sh[qi] += up;
                    l_run[qi] *= al;
                    // O lives in the accumulator half of the register file (one wave per SIMD: 256 + 256 registers). Written as
                    // plain C++ (oacc *= al) this rare branch made hipcc keep O where vector instructions reach it and spill
                    // 393 registers in the whole kernel; element by element through ONE temporary it spills 30, none in the loop
#pragma unroll
                    for (int b = 0; b < DB; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float tmp;
                            asm volatile("v_accvgpr_read_b32 %1, %0\n\ts_nop 1\n\tv_mul_f32 %1, %1, %2\n\ts_nop 1\n\tv_accvgpr_write_b32 %0, %1\n\ts_nop 1"
                                         : "+a"(oacc[qi][b][r]), "=&v"(tmp) : "v"(al));
                        }
#pragma unroll
                    for (int r = 0; r < 16; ++r) sn[qi][r] -= up;
                }
            }
        }
    };
    auto tile_head = [&](const int t) {
        wait_tiles(t + 2 < nt ? 1 : 0);       // outstanding, oldest first: tile t+1, tile t+2; tile t+1 must have landed
        __builtin_amdgcn_s_barrier();         // K(t+1), V(t) visible to all; slot of tile t-1 no longer read by anyone
        if (t + 3 < nt) issue_tile(t + 3);
    };
    using Free = std::integral_constant<int, 1>;
    using Masked = std::integral_constant<int, 2>;
    using Last = std::integral_constant<int, 0>;
    int t = 0;
    for (; 2 * t + 2 < nfree; ++t) {          // blocks 2t+1 and 2t+2 need no mask
        tile_head(t);
        if (wave_active) {
            substep(sa, sb, 2 * t, Free{});
            substep(sb, sa, 2 * t + 1, Free{});
        }
    }
    for (; t < nt; ++t) {                     // the masked end: block kinds decided at run time (wave-uniform)
        tile_head(t);
        if (!wave_active) continue;
        {
            const int j = 2 * t;
            if (j + 1 < nfree) substep(sa, sb, j, Free{});
            else if (j + 1 < nblk) substep(sa, sb, j, Masked{});
            else substep(sa, sb, j, Last{});
        }
        if (2 * t + 1 < nblk) {
            const int j = 2 * t + 1;
            if (j + 1 < nfree) substep(sb, sa, j, Free{});
            else if (j + 1 < nblk) substep(sb, sa, j, Masked{});
            else substep(sb, sa, j, Last{});
        }
    }
    if (!wave_active) return;

    // epilogue: lane (q, h) holds d = 32 b + 8 g4 + 4 h + 0..3; pack and exchange between the halves so that a lane owns 8
    // consecutive d (attn_vit.hip)
    bf16_t* Op = reinterpret_cast<bf16_t*>(p.O);
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        float l = l_run[qi];
        l += __shfl_xor(l, 32, 64);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        bf16_t* orow = Op + (long)qrow[qi] * p.ldo + head * HD;
#pragma unroll
        for (int b = 0; b < DB; ++b) {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                unsigned e0 = pack_bf2(oacc[qi][b][8 * gp + 0] * inv, oacc[qi][b][8 * gp + 1] * inv);
                unsigned e1 = pack_bf2(oacc[qi][b][8 * gp + 2] * inv, oacc[qi][b][8 * gp + 3] * inv);
                unsigned o0 = pack_bf2(oacc[qi][b][8 * gp + 4] * inv, oacc[qi][b][8 * gp + 5] * inv);
                unsigned o1 = pack_bf2(oacc[qi][b][8 * gp + 6] * inv, oacc[qi][b][8 * gp + 7] * inv);
                const auto s0 = __builtin_amdgcn_permlane32_swap(e0, o0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(e1, o1, false, false);
                const int d0 = 32 * b + 16 * gp + 8 * h;
                if (qok[qi]) *reinterpret_cast<u32x4*>(orow + d0) = u32x4{(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Ping-pong form of the prompt attention above (round 4): 8 waves = 256 query rows of one head per workgroup, two
// groups of four waves, one wave of each group per SIMD. In the kernel above the two waves of a SIMD belong to different
// workgroups and fall into step: per tile a SIMD spends the SUM of a wave's 1 024 MFMA cycles and its ~900 cycles of
// softmax VALU, twice (43 % matrix pipe). Here workgroup barriers keep the groups half a tile apart:
//     group 0:  M(t) | V(t) | M(t+1) | V(t+1) ...        M(t) = O += V(t-1)^T.P(t-1), then S(t) = K(t).Q^T  (64 MFMAs)
//     group 1:   -   | M(t) | V(t)   | M(t+1) ...        V(t) = P(t) = softmax numerators of S(t)           (VALU) + staging
// so on every SIMD one wave feeds the matrix pipe while the other runs the VALU stream. Same arithmetic per element as
// the kernel above, in the same order (PV of tile t simply happens one phase later), hence bit-identical outputs.
// K/V: five 32 KiB tile slots (all of LDS), tile t+3 is issued in V(t) -- four DMA pieces per wave -- into the slot tile
// t-2 left (its last read, PV(t-2) in M(t-1), lies two intervals back for either group). Tile t+1 has to be complete by
// the end of the interval in which group 0 runs V(t) and group 1 runs M(t): group 0 waits at the end of V(t) (two newer
// tiles in flight: vmcnt(8)), group 1 at the end of M(t) (one: vmcnt(4)); the last three tiles of a block run a second
// copy of the loop body with vmcnt(0).
__global__ __launch_bounds__(512, 2) void attn_prefill_pp_kernel(AttnArgs p) {
    constexpr int HD = 128, NQ = 2, KS = 4, DT = 8, NSLOT = 5;
    constexpr int K_LDS = 64 * 256, V_LDS = 64 * 256, BUF = K_LDS + V_LDS;     // 32 KiB per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, wq = wid & 3;
    const int g = lane >> 4, li = lane & 15;
    const int seg = blockIdx.z;
    const int gsz = p.hq / p.hkv;
    const int kvh = blockIdx.y / gsz;

    int qs = 0, qe = p.q_len, ks = 0, ke = p.kv_len;
    if (p.cu) { qs = p.cu[seg]; qe = p.cu[seg + 1]; ks = qs; ke = qe; }
    const int qt = p.heavy_first ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int q0 = qs + qt * 256;
    if (q0 >= qe) return;
    const int gq0 = q0 + grp * 128;                       // first query row of this group

    const bf16_t* Qp = reinterpret_cast<const bf16_t*>(p.Q);
    const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.K);
    const bf16_t* Vp = reinterpret_cast<const bf16_t*>(p.V);

    bf16x8 qf[NQ][KS];
    int qrow[NQ];
    bool qok[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        qrow[qi] = gq0 + wq * 32 + qi * 16 + li;
        qok[qi] = qrow[qi] < qe;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 v = {0, 0, 0, 0};
            if (qok[qi]) v = *reinterpret_cast<const u32x4*>(Qp + (long)qrow[qi] * p.ldq + blockIdx.y * HD + 32 * s + 8 * g);
            qf[qi][s] = __builtin_bit_cast(bf16x8, v);
        }
    }
    // keys the workgroup stages (its last row's causal range) and keys this group needs
    const int last_row = min(q0 + 255, qe - 1);
    const int kend_wg = min(ke, ks + (last_row - qs) + p.q_pos0 + 1);
    const int nt = (kend_wg - ks + 63) / 64;
    const int kend = gq0 < qe ? min(ke, ks + (gq0 - qs) + 127 + p.q_pos0 + 1) : ks;      // this group's causal range
    const int nt_g = (kend - ks + 63) / 64;                                                // <= nt

    f32x4 oacc[DT][NQ];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) oacc[d][qi] = f32x4{0, 0, 0, 0};
    float l_run[NQ], m_ref[NQ];
    bool first_tile = true;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) { l_run[qi] = 0.f; m_ref[qi] = 0.f; }

    // ---- staging by LDS-DMA: wave w issues pieces w and w + 8 (4 tile rows each) of K and of V
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    int st_row[2], k_off[2], v_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 4 * (wid + 8 * i) + (lane >> 4);
        st_row[i] = row;
        k_off[i] = (row * (int)p.ldk + kvh * HD + ((lane & 15) ^ k_swz(row)) * 8) * 2;     // bytes from the tile's first row
        v_off[i] = (row * (int)p.ldv + kvh * HD + ((lane & 15) ^ v_swz(row)) * 8) * 2;
    }
    auto uniform_ptr = [](const bf16_t* q) -> const bf16_t* {
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const bf16_t*)(((unsigned long long)hi << 32) | lo);
    };
    auto dma16 = [&](const bf16_t* base, int off_bytes, unsigned lds) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(lds), "v"(off_bytes), "s"(base) : "memory");
    };
    int is_slot = 0;                                     // slot of the next tile to issue (tiles are issued in order)
    auto issue_tile = [&](int kt) {
        const int kbase = ks + kt * 64;
        const int valid = ke - kbase;                               // >= 1
        const bf16_t* kb = uniform_ptr(Kp + (long)kbase * p.ldk);
        const bf16_t* vb = uniform_ptr(Vp + (long)kbase * p.ldv);
        const unsigned st = __builtin_amdgcn_readfirstlane(smem_lds + is_slot * BUF);
        is_slot = is_slot + 1 == NSLOT ? 0 : is_slot + 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int ko = k_off[i], vo = v_off[i];
            if (valid < 64) {                                        // rows past the key range repeat its last row
                const int back = st_row[i] - min(st_row[i], valid - 1);
                ko -= back * (int)p.ldk * 2;
                vo -= back * (int)p.ldv * 2;
            }
            dma16(kb, ko, st + (wid + 8 * i) * 1024);
            dma16(vb, vo, st + K_LDS + (wid + 8 * i) * 1024);
        }
    };

    // per-lane LDS read addressing (as in the kernel above)
    const int krow0 = 8 * (li >> 2) + (li & 3);
    const int vr = 8 * g + (li >> 2);
    const int v_sw = v_swz(vr);
    const int v_base = vr * 256 + ((li & 3) & 1) * 8;
    const int v_ch = (li & 3) >> 1;

    const bool wave_active = gq0 + wq * 32 < qe;
    // leading tiles [0, t_mid) of this group need no mask: inside the key range and left of its causal diagonal
    int t_mid;
    {
        int full = (ke - ks) / 64;
        const int lim = (gq0 - qs) + p.q_pos0 - 63;
        full = min(full, lim >= 0 ? lim / 64 + 1 : 0);
        t_mid = max(0, min(full, nt_g));
    }

    f32x4 sacc[4][NQ];
    bf16x8 pf[2][NQ];
    float post_alpha[NQ] = {1.f, 1.f};
    int rescale[NQ] = {0, 0};         // scalar copies of "post_alpha != 1 somewhere in the wave", decided in the V phase: a vote
                                      // (VALU writing an SGPR) right behind the MFMAs of the PV product would wait for the matrix pipe

    // The M phase: O += V(t-1)^T.P(t-1) (eight batches, one d-block each: 4 transposing reads, 4 MFMAs), then the
    // deferred rescale, then S(t) = K(t).Q^T - m (eight batches: one 16-key group x two k-steps: 2 fragment reads, 4
    // MFMAs). In this phase the wave is the only one of its SIMD that feeds the matrix pipe, so LDS latency has to be
    // hidden inside the wave: one software pipeline over the (up to) sixteen batches, the two fragments of batch i + 3
    // requested before the MFMAs of batch i (four fragment sets of 8 registers; three batches = 192 cycles of MFMA cover
    // the read). Measured with phase stamps on a block whose other group was idle: with only the next 8-MFMA batch
    // requested ahead the phase took 2 070 cycles for its 1 024 cycles of MFMA.
    // (Masked tiles: the kernel above skips the MFMAs of a tile's second 32 keys when they all lie beyond the causal
    // diagonal; here they run -- their P is exactly 0 and the V rows are finite, their S is masked to -inf by the
    // softmax -- so the result is the same bits and the pipeline has no branches.)
    auto m_phase = [&](auto do_pv_tag, auto do_qk_tag, const int pv_slot, const int qk_slot) __attribute__((always_inline)) {
        constexpr bool DO_PV = decltype(do_pv_tag)::value, DO_QK = decltype(do_qk_tag)::value;
        constexpr int FIRST = DO_PV ? 0 : 8, END = DO_QK ? 16 : 8, AHEAD = 3;
        const char* Vs = smem + pv_slot * BUF + K_LDS;
        const char* Ks = smem + qk_slot * BUF;
        u32x4 fr[AHEAD + 1][2];
        // batch i < 8: d-blocks 2(i>>1), 2(i>>1)+1 against key half u = i & 1; batch i >= 8: key groups 2((i-8)>>2),
        // 2((i-8)>>2)+1 against k-step s = (i-8) & 3 -- four independent MFMAs per batch, and the batch that continues an
        // accumulator follows four MFMAs later (an MFMA's result is ready for a dependent one after ~2 issue slots)
        auto load_batch = [&](const int i, u32x4 (&dst)[2]) {
            if (i < 8) {
                const int u = i & 1;
#pragma unroll
                for (int dd = 0; dd < 2; ++dd) {
                    const int d = 2 * (i >> 1) + dd;
                    const char* va = Vs + u * 32 * 256 + v_base + ((((2 * d + v_ch) ^ v_sw)) << 4);
                    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(va));
                    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(va + 4 * 256));
                    u32x4 w;
                    u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                    w[0] = l2[0]; w[1] = l2[1]; w[2] = h2[0]; w[3] = h2[1];
                    dst[dd] = w;
                }
            } else {
                const int sidx = (i - 8) & 3;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int ut = 2 * ((i - 8) >> 2) + e;
                    const int krow = 32 * (ut >> 1) + 4 * (ut & 1) + krow0;
                    dst[e] = *reinterpret_cast<const u32x4*>(Ks + krow * 256 + (((4 * sidx + g) ^ k_swz(krow)) << 4));
                }
            }
        };
        auto compute_batch = [&](const int i, const u32x4 (&src)[2]) {
            if (i < 8) {
                const int u = i & 1;
#pragma unroll
                for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) {
                        const int d = 2 * (i >> 1) + dd;
                        oacc[d][qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, src[dd]), pf[u][qi], oacc[d][qi], 0, 0, 0);
                    }
            } else {
                const int sidx = (i - 8) & 3;
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) {
                        const int ut = 2 * ((i - 8) >> 2) + e;
                        const float nm = -m_ref[qi];
                        const f32x4 c0 = f32x4{nm, nm, nm, nm};
                        sacc[ut][qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, src[e]), qf[qi][sidx], sidx == 0 ? c0 : sacc[ut][qi], 0, 0, 0);
                    }
            }
        };
#pragma unroll
        for (int i = FIRST; i < FIRST + AHEAD && i < END; ++i) load_batch(i, fr[i % (AHEAD + 1)]);
#pragma unroll
        for (int i = FIRST; i < END; ++i) {
            if (i + AHEAD < END) load_batch(i + AHEAD, fr[(i + AHEAD) % (AHEAD + 1)]);
            __builtin_amdgcn_sched_barrier(0);
            compute_batch(i, fr[i % (AHEAD + 1)]);
            __builtin_amdgcn_sched_barrier(0);
            if (DO_PV && i == 7) {                          // the rescale that belongs behind the PV product (rare)
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    if (rescale[qi]) {
                        l_run[qi] *= post_alpha[qi];
#pragma unroll
                        for (int d = 0; d < DT; ++d) oacc[d][qi] *= post_alpha[qi];
                    }
                }
            }
        }
    };
    // P(t) from S(t): exactly the softmax body of the kernel above
    auto softmax_tile = [&](const int kt) {
        const int kbase = ks + kt * 64;
        const bool masked = kt >= t_mid;
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            float sv[4][4];
            post_alpha[qi] = 1.f;
            float d = -INFINITY;
            if (masked) {
                const int qloc = qrow[qi] - qs;
#pragma unroll
                for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kbase + 32 * (ut >> 1) + 8 * g + 4 * (ut & 1) + r;
                        const bool valid = key < ke && (key - ks) <= qloc + p.q_pos0;
                        const float sc = valid ? sacc[ut][qi][r] : -INFINITY;
                        sv[ut][r] = sc;
                        d = fmaxf(d, sc);
                    }
            } else {
#pragma unroll
                for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sv[ut][r] = sacc[ut][qi][r]; d = fmaxf(d, sacc[ut][qi][r]); }
            }
            d = colgroup_max(d);
            if (first_tile) {
                const float d0 = (d == -INFINITY) ? 0.f : d;
#pragma unroll
                for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sv[ut][r] -= d0;
                asm volatile("" ::: "memory");
                m_ref[qi] = d0;
                d = 0.f;
            }
            float psum = 0.f;
#pragma unroll
            for (int ut = 0; ut < 4; ++ut)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(sv[ut][r]);
                    sv[ut][r] = pv;
                    psum += pv;
                }
            l_run[qi] += psum;
            int moved = 0;
            if (__any(d > 0.f)) {
                const float dd = fmaxf(d, 0.f);
                post_alpha[qi] = __builtin_amdgcn_exp2f(-dd);
                m_ref[qi] += dd;
                moved = 1;
            }
            asm volatile("s_mov_b32 %0, %1" : "=s"(rescale[qi]) : "s"(__builtin_amdgcn_readfirstlane(moved)));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                u32x4 w;
                w[0] = pack_bf2(sv[2 * u][0], sv[2 * u][1]);
                w[1] = pack_bf2(sv[2 * u][2], sv[2 * u][3]);
                w[2] = pack_bf2(sv[2 * u + 1][0], sv[2 * u + 1][1]);
                w[3] = pack_bf2(sv[2 * u + 1][2], sv[2 * u + 1][3]);
                pf[u][qi] = __builtin_bit_cast(bf16x8, w);
            }
        }
        first_tile = false;
    };
    // prologue: tiles 0, 1, 2 in flight, tile 0 landed
    {
        const int pre = nt < 3 ? nt : 3;
        for (int i = 0; i < pre; ++i) issue_tile(i);
        if (pre >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (pre == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qf[qi][s]));     // Q loads complete before the loop
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one phase behind group 0

    // The tile loop, written as a straight sequence (first tile: QK only | tiles with PV and QK | the PV of the last
    // computed tile | tiles this group only stages and synchronises for) -- a choice between phase bodies INSIDE one loop
    // is a diamond over 130 live accumulator registers and made hipcc spill ~50 of them.
    // Waits: group 1 needs tile t + 1 complete at the end of M(t) (one newer tile of its own in flight: vmcnt(4)), group 0
    // at the end of V(t) (two newer: vmcnt(8)); each group executes the other's wait too (it only asks its own share of a
    // tile a phase early, or nothing), so no branch on the group stands behind the MFMAs. The last three tiles wait for
    // everything.
    int slot = 0, pslot = 0;                               // slots of tile t and of tile t - 1
    auto sync_m = [&](const int t) {
        if (t < nt - 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto v_tail = [&](const int t) {
        if (t + 3 < nt) issue_tile(t + 3);
        if (t < nt - 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        pslot = slot;
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
    };
    const int n_act = wave_active ? nt_g : 0;              // tiles this wave computes
    int t = 0;
    if (n_act > 0) {
        if (p.pp_prio == 1) __builtin_amdgcn_s_setprio(1);
        m_phase(std::false_type{}, std::true_type{}, pslot, slot);
        sync_m(0);
        if (p.pp_prio == 1) __builtin_amdgcn_s_setprio(0);
        softmax_tile(0);
        v_tail(0);
#ifdef COGS_ATTN_PP_STAMPS
        unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#define PPSTAMP(i_) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_sum[i_] += now_ - st_prev; st_prev = now_; } while (0)
#else
#define PPSTAMP(i_) do {} while (0)
#endif
        for (t = 1; t < n_act; ++t) {
            if (p.pp_prio == 1) __builtin_amdgcn_s_setprio(1);
            m_phase(std::true_type{}, std::true_type{}, pslot, slot);
            PPSTAMP(0);
            sync_m(t);
            PPSTAMP(1);
            if (p.pp_prio == 1) __builtin_amdgcn_s_setprio(0);
            softmax_tile(t);
            PPSTAMP(2);
            v_tail(t);
            PPSTAMP(3);
        }
#ifdef COGS_ATTN_PP_STAMPS
        if (p.part_o && (blockIdx.x == 10 || blockIdx.x == 0) && blockIdx.y == 0 && blockIdx.z == 0 && wq == 0 && lane == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(p.part_o) + grp * 8 + (blockIdx.x == 0 ? 16 : 0);
            for (int i = 0; i < 6; ++i) o[i] = st_sum[i];
            o[6] = (unsigned long long)(n_act - 1);
        }
#endif
        if (p.pp_prio == 1) __builtin_amdgcn_s_setprio(1);
        m_phase(std::true_type{}, std::false_type{}, pslot, slot);       // the last computed tile's PV: M(n_act)
        if (p.pp_prio == 1) __builtin_amdgcn_s_setprio(0);
        if (t < nt) { sync_m(t); v_tail(t); ++t; }
    }
    for (; t < nt; ++t) { sync_m(t); v_tail(t); }
    if (grp == 0) __builtin_amdgcn_s_barrier();          // balance group 1's extra barrier

    bf16_t* Op = reinterpret_cast<bf16_t*>(p.O);
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        float l = l_run[qi];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (!qok[qi]) continue;
        const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            f32x4 v = oacc[d][qi] * inv;
            st4_f<bf16_t>(Op + (long)qrow[qi] * p.ldo + blockIdx.y * HD + 16 * d + 4 * g, v);
        }
    }
}

