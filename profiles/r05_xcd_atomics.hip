// Microbenchmark (round 5): returning atomics on hot counters -- one counter for the whole chip vs one per XCD, at agent scope
// (what atomicAdd means in HIP) vs workgroup scope (executed in the issuing XCD's L2: valid when only one XCD touches the line).
//   hipcc --offload-arch=gfx950 -O3 profiles/r05_xcd_atomics.hip -o profiles/_bin/xcd_atomics && profiles/_bin/xcd_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int MODE>   // 0: one counter, agent scope   1: per-XCD counter, agent scope   2: per-XCD counter, workgroup scope
__global__ void hot(int* counters, int n_hot, int iters, int* sink, int* xcc_seen) {
    const int lane = threadIdx.x & 63;
    const int x = xcc_id();
    if (threadIdx.x == 0) atomicOr(xcc_seen, 1 << x);
    int acc = 0;
    for (int it = 0; it < iters; ++it) {
        const int c = (blockIdx.x * 7 + it) % n_hot;                    // a handful of hot lists
        int* p = counters + (MODE == 0 ? c * 64 : (c * 8 + x) * 64);     // one 256-B line per counter
        if (lane == 0) {
            if (MODE == 2) acc += __hip_atomic_fetch_add(p, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else acc += __hip_atomic_fetch_add(p, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (lane == 0 && acc == 0x7fffffff) *sink = acc;
}

template <int MODE>
static void run(const char* name, int n_hot) {
    const int blocks = 2048, threads = 256, iters = 12;
    int *counters, *sink, *seen;
    const size_t n = (size_t)n_hot * 8 * 64;
    hipMalloc(&counters, n * 4); hipMalloc(&sink, 4); hipMalloc(&seen, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(counters, 0, n * 4); hipMemset(seen, 0, 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL(hot<MODE>, dim3(blocks), dim3(threads), 0, 0, counters, n_hot, iters, sink, seen);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    std::vector<int> h(n); int hs;
    hipMemcpy(h.data(), counters, n * 4, hipMemcpyDeviceToHost); hipMemcpy(&hs, seen, 4, hipMemcpyDeviceToHost);
    long long sum = 0;
    for (size_t i = 0; i < n; i += 64) sum += h[i];
    const long long want = 3ll * blocks * (threads / 64) * iters;
    printf("%-42s hot counters %3d: %8.1f us  %6.2f G atomics/s  sum %s (xcc mask 0x%x)\n", name, n_hot, best * 1e3,
           want / 3 / (best * 1e-3) / 1e9, sum == want ? "ok" : "WRONG", hs);
    hipFree(counters); hipFree(sink); hipFree(seen);
}

int main() {
    for (int n_hot : {4, 32, 256}) {
        run<0>("one counter per list, agent scope", n_hot);
        run<1>("one counter per list and XCD, agent scope", n_hot);
        run<2>("one counter per list and XCD, workgroup scope", n_hot);
    }
    return 0;
}
