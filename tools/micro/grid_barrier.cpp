// Cost of a grid-wide barrier inside one persistent kernel on MI355X (decode-path study):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/grid_barrier.cpp -o tools/micro/grid_barrier
// Every workgroup: (optional agent-scope release fence) -> atomic arrive -> spin on the generation word (bounded) ->
// (optional acquire fence). Prints microseconds per barrier for 256 / 512 / 1024 workgroups of 256 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Bar { unsigned count; unsigned gen; unsigned err; unsigned pad; };

template <bool FENCE>
__global__ __launch_bounds__(256) void bar_kernel(Bar* b, int iters, float* sink, const float* src) {
    const unsigned nwg = gridDim.x;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        acc += src[(blockIdx.x * 256 + threadIdx.x + it) & 0xffff];            // some traffic between barriers
        if (FENCE) __threadfence();                                            // release: make this phase's stores visible
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned g = __hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned arrived = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (arrived == nwg - 1) {
                __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                long spins = 0;
                while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 20000000L) { b->err = 1; break; }            // bounded: never hangs the box
                }
            }
        }
        __syncthreads();
        if (FENCE) __threadfence();                                            // acquire side
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

// Hierarchical form: one arrival counter per XCD (blockIdx % 8), the last arriver of an XCD arrives at the top counter,
// the last of those bumps the generation word; counters and the polled word sit on separate 256-byte lines.
struct HBar { unsigned xcd[8][64]; unsigned top[64]; unsigned gen[64]; unsigned err[64]; };

template <bool FENCE>
__global__ __launch_bounds__(256) void hbar_kernel(HBar* b, int iters, float* sink, const float* src) {
    const unsigned nwg = gridDim.x, x = blockIdx.x & 7, per = nwg / 8;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        acc += src[(blockIdx.x * 256 + threadIdx.x + it) & 0xffff];
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned g = __hip_atomic_load(&b->gen[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool last = false;
            if (__hip_atomic_fetch_add(&b->xcd[x][0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == per - 1) {
                __hip_atomic_store(&b->xcd[x][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 7) {
                    __hip_atomic_store(&b->top[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(&b->gen[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    last = true;
                }
            }
            if (!last) {
                long spins = 0;
                while (__hip_atomic_load(&b->gen[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > 5000000L) { b->err[0] = 1; break; }
                }
            }
        }
        __syncthreads();
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    Bar* b; float *sink, *src;
    hipMalloc(&b, sizeof(Bar)); hipMalloc(&sink, 1024 * 256 * 4); hipMalloc(&src, 65536 * 4);
    hipMemset(src, 0, 65536 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200;
    for (int fence = 0; fence < 2; ++fence)
        for (int nwg : {256, 512, 1024}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(b, 0, sizeof(Bar));
                hipEventRecord(e0, 0);
                if (fence) hipLaunchKernelGGL(bar_kernel<true>, dim3(nwg), dim3(256), 0, 0, b, iters, sink, src);
                else hipLaunchKernelGGL(bar_kernel<false>, dim3(nwg), dim3(256), 0, 0, b, iters, sink, src);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            Bar hb; hipMemcpy(&hb, b, sizeof(Bar), hipMemcpyDeviceToHost);
            printf("%4d workgroups, fences %d: %.2f us per barrier (err %u)\n", nwg, fence, best * 1e3 / iters, hb.err);
        }
    HBar* hb; hipMalloc(&hb, sizeof(HBar));
    for (int fence = 0; fence < 2; ++fence)
        for (int nwg : {256, 512}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(hb, 0, sizeof(HBar));
                hipEventRecord(e0, 0);
                if (fence) hipLaunchKernelGGL(hbar_kernel<true>, dim3(nwg), dim3(256), 0, 0, hb, iters, sink, src);
                else hipLaunchKernelGGL(hbar_kernel<false>, dim3(nwg), dim3(256), 0, 0, hb, iters, sink, src);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            HBar h; hipMemcpy(&h, hb, sizeof(HBar), hipMemcpyDeviceToHost);
            printf("hierarchical, %4d workgroups, fences %d: %.2f us per barrier (err %u)\n", nwg, fence, best * 1e3 / iters, h.err[0]);
        }
    return 0;
}
