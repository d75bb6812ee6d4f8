// Phase stamps of the sampler's merge kernel (csrc/sample.hip):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCOGS_SAMPLE_STAMPS -I cogstream_amd/csrc tools/micro/sample_micro.cpp -o tools/micro/sample_micro
#include "../../cogstream_amd/csrc/sample.hip"
#include <cstdio>
#include <random>
#include <vector>
int main() {
    const int n = 152064;
    std::vector<float> h(n);
    std::mt19937 rng(3); std::normal_distribution<float> nd(0.f, 3.f);
    for (auto& v : h) v = nd(rng);
    float* lg; long long* tok; void* ws;
    hipMalloc(&lg, n * 4); hipMalloc(&tok, 8); hipMalloc(&ws, cogs_k_sample_ws());
    hipMemcpy(lg, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) cogs_k_sample(0, lg, n, 0.7f, 20, 0.8, nullptr, 1234, i, (int64_t*)tok, nullptr, nullptr, nullptr, 0, ws);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    printf("cogs_sample (slice top-k + merge): %.1f us per call\n", best / 20 * 1e3);
#ifdef COGS_SAMPLE_STAMPS
    unsigned long long st[8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_sample_stamps), sizeof(st));
    printf("merge kernel, thread 0 (shader cycles): prefilter %llu | compaction %llu | ranks %llu | ties %llu | exp + draws %llu | serial tail %llu\n",
           st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5]);
#endif
    return 0;
}
