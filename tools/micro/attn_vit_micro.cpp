// Stand-alone timing / stamp harness for csrc/attn_vit.hip (ViT block-diagonal attention, hd 72):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DCOGS_ATTN_STAMPS] -I cogstream_amd/csrc tools/micro/attn_vit_micro.cpp -o tools/micro/attn_vit_micro
//   attn_vit_micro [frames] [rows per frame] [layout: 0 token-major fused qkv, 1 head-major q | k | v]
#include "../../cogstream_amd/csrc/attn_vit.hip"
CogsDebug g_cogs_debug;
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
int main(int argc, char** argv) {
    const int nseg = argc > 1 ? atoi(argv[1]) : 64, seg = argc > 2 ? atoi(argv[2]) : 924, heads = 16, hd = 72;
    const int hm = argc > 3 ? atoi(argv[3]) : 1;
    const long L = (long)nseg * seg, H = heads * hd;
    std::vector<uint16_t> h(L * 3 * H);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (long i = 0; i < L; ++i)
        for (long c = 0; c < 3 * H; ++c) {
            float v = nd(rng) * (c < H ? 1.4426950408889634f / sqrtf(72.f) : 1.f);
            uint32_t u; std::memcpy(&u, &v, 4); h[i * 3 * H + c] = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
        }
    uint16_t *qkv, *out; int* cu;
    hipMalloc(&qkv, h.size() * 2); hipMalloc(&out, L * H * 2); hipMalloc(&cu, (nseg + 1) * 4);
    if (hm) {      // [row][q | k | v][head][hd] -> [q | k | v][head][row][hd]
        std::vector<uint16_t> t(h.size());
        for (long i = 0; i < L; ++i)
            for (int w = 0; w < 3; ++w)
                for (int hh = 0; hh < heads; ++hh)
                    std::memcpy(&t[((long)(w * heads + hh) * L + i) * hd], &h[i * 3 * H + w * H + hh * hd], hd * 2);
        h.swap(t);
    }
    hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    std::vector<int> hc(nseg + 1); for (int i = 0; i <= nseg; ++i) hc[i] = i * seg;
    hipMemcpy(cu, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    CogsAttn a; a.dtype = COGS_DT_BF16; a.Q = qkv; a.K = qkv + H; a.V = qkv + 2 * H; a.O = out;
    a.ldq = a.ldk = a.ldv = 3 * H; a.ldo = H; a.cu_seqlens = cu; a.nseg = nseg; a.max_seqlen = seg;
    a.uniform_seqlen = seg;
    if (hm) {
        a.K = qkv + L * H; a.V = qkv + 2 * L * H; a.ldq = a.ldk = a.ldv = hd; a.head_stride = L * hd;
    }
    a.q_len = a.kv_len = (int)L; a.hq = a.hkv = heads; a.head_dim = hd; a.q_prescaled = 1;
    if (argc > 4) g_cogs_debug.attn_vit_len = atoi(argv[4]);      // 0: the run-time-length kernel for every frame size
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) cogs_k_attention_vit(0, a);
    hipDeviceSynchronize();
    float best = 1e9f, sum = 0;
    for (int i = 0; i < 20; ++i) {
        hipEventRecord(e0, 0); cogs_k_attention_vit(0, a); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
    }
    const double fl = 4.0 * nseg * heads * (double)seg * seg * hd;
    printf("attn_vit %dx%d: mean %.4f ms min %.4f ms  %.1f TFLOP/s (min)\n", nseg, seg, sum / 20, best, fl / best / 1e9);
#ifdef COGS_LIFE_STAMPS
    {
        unsigned long long lf[8];
        hipMemcpyFromSymbol(lf, HIP_SYMBOL(g_av_life), sizeof(lf));
        printf("workgroup lifetime (wave 0, s_memtime ticks): %.0f on average over %llu workgroups\n", lf[1] ? (double)lf[0] / lf[1] : 0.0, lf[1]);
    }
#endif
#ifdef COGS_ATTN_STAMPS
    unsigned long long st[8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_attn_stamps), sizeof(st));
    const double n = (double)st[5];
    printf("per full tile, wave 0 of one workgroup (cycles): barrier wait %.0f | stage %.0f | QK (to results) %.0f | softmax %.0f | PV issue %.0f | sum %.0f\n",
           st[0] / n, st[1] / n, st[2] / n, st[3] / n, st[4] / n, (st[0] + st[1] + st[2] + st[3] + st[4]) / n);
#endif
#ifdef COGS_PHASE_STAMPS
    {
        unsigned long long st[8];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_attn_stamps), sizeof(st));
        const double n = (double)st[3] + 1e-9;
        printf("workgroup 3000, wave 0, %d main-loop tiles, shader cycles per tile: wait for tile t+1 %.0f | barrier %.0f | issue of tile t+3 %.0f | sub-step 0 %.0f | sub-step 1 %.0f | sum %.0f\n",
               (int)n, st[0] / n, st[1] / n, st[2] / n, st[4] / n, st[5] / n, (st[0] + st[1] + st[2] + st[4] + st[5]) / n);
        unsigned long long tl[8];
        hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_tail_stamps), sizeof(tl));
        printf("   the ragged end of the same wave (cycles): tile heads (wait + barrier + issue) %llu in all | sub-steps in order: %llu %llu %llu %llu %llu\n",
               tl[0], tl[1], tl[2], tl[3], tl[4], tl[5]);
    }
#endif
#ifdef COGS_PIPE_STAMPS
    {
        unsigned long long st[8];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_attn_stamps), sizeof(st));
#ifdef COGS_PIPE_STAMPS2
        printf("workgroup 3000, wave 0 (shader cycles): entry->segment bounds %llu | ->Q loads issued %llu | ->3 tiles issued %llu | ->Q and tiles landed %llu | ->S(0) ready %llu | main loop %llu | tail %llu\n",
               st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[7] - st[6]);
#else
        printf("workgroup 3000, wave 0 (shader cycles): entry->Q and tile 0 landed %llu | ->S(0) ready %llu | main loop %llu | tail %llu | epilogue+store %llu\n",
               st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4]);
#endif
    }
#endif
    return 0;
}
