// Round 6 microbenchmark: are packed fp32 VALU operations (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) twice the rate of the scalar forms on gfx950?
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off profiles/r06_pk_f32_rate.hip -o /tmp/pk && /tmp/pk
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    f2* v = (f2*)a;
    f2 s2 = {s, s}, h2 = {0.5f, 0.5f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // scalar mul + add (no contraction): 32 VALU per round
#pragma unroll
            for (int i = 0; i < 16; ++i) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s)); asm volatile("v_add_f32 %0, 0.5, %0" : "+v"(a[i])); }
        } else if (MODE == 1) {     // packed mul + add: 16 VALU per round, same flops
#pragma unroll
            for (int i = 0; i < 8; ++i) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(s2)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(h2)); }
        } else if (MODE == 2) {     // scalar fma
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(a[i]) : "v"(s));      // (inline asm: the compiler packs a plain loop by itself)
        } else {                    // packed fma
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(s2), "v"(h2));
        }
    }
    float r = 0; for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> float run(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256 * 8, 256>>>(d, 10, 0.999f);
    hipEventRecord(e0); k<MODE><<<256 * 8, 256>>>(d, iters, 0.999f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 20000;
    const double lane_ops = 256.0 * 8 * 256 * iters * 16;      // (element, operation-pair) count
    float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters);
    printf("scalar mul+add %.3f ms (%.1f G elem-pairs/s) | packed mul+add %.3f ms (%.1f) | scalar fma %.3f ms (%.1f G fma/s) | packed fma %.3f ms (%.1f)\n",
           t0, lane_ops / t0 * 1e-6, t1, lane_ops / t1 * 1e-6, t2, lane_ops / t2 * 1e-6, t3, lane_ops / t3 * 1e-6);
    return 0;
}
