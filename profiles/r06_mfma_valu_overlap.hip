// r06_mfma_valu_overlap.hip -- do v_mfma_f32_32x32x2_f32 chains and fp32 VALU work overlap on one SIMD of gfx950, and how?
//   hipcc --offload-arch=gfx950 -O2 profiles/r06_mfma_valu_overlap.hip -o profiles/_bin/r06_mfma_valu_overlap
// One workgroup of 8 waves per CU (2 waves per SIMD), 256 workgroups.  Per "tile": 120 dependent MFMAs (the decoder chain of
// render.hip) and NV independent v_fma_f32 (the gather / OneBlob work around it: ~1750 VALU per tile).  Modes:
//   mfma       every wave: chains only                               valu      every wave: VALU only
//   serial     every wave: VALU block then chain (what the frame kernels do)
//   split      waves 0-3 chains only, waves 4-7 VALU only (one of each per SIMD)   -> do the two pipes run side by side?
//   inter1     every wave: ONE chain with FILL v_fma between consecutive MFMAs (same accumulator)
//   inter2     every wave: TWO chains alternating (different accumulators) with FILL v_fma after every MFMA
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N>
__device__ __forceinline__ void valu_block(float (&v)[8], float k) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], k, 1.0f);
}

__device__ __forceinline__ f32x16 chain(f32x16 acc, float a, float b, int n) {
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    return acc;
}

template <int MODE, int FILL>
__global__ __launch_bounds__(512) void k(float* out, int tiles, float a, float b) {
    const int wv = threadIdx.x >> 6;
    f32x16 acc, acc2;
    for (int q = 0; q < 16; ++q) { acc[q] = 0.0f; acc2[q] = 1.0f; }
    float v[8];
    for (int q = 0; q < 8; ++q) v[q] = (float)(threadIdx.x + q);
    for (int t = 0; t < tiles; ++t) {
        if (MODE == 0) acc = chain(acc, a, b, 120);
        else if (MODE == 1) { for (int r = 0; r < 14; ++r) valu_block<125>(v, a); }
        else if (MODE == 2) { for (int r = 0; r < 14; ++r) valu_block<125>(v, a); acc = chain(acc, a, v[0] * 0.0f + b, 120); }
        else if (MODE == 3) {
            if (wv < 4) acc = chain(acc, a, b, 120);
            else { for (int r = 0; r < 14; ++r) valu_block<125>(v, a); }
        } else if (MODE == 4) {
            for (int i = 0; i < 120; i += 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                    valu_block<FILL>(v, a);
                }
            }
        } else {
            for (int i = 0; i < 60; i += 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                    valu_block<FILL>(v, a);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
                    valu_block<FILL>(v, a);
                }
            }
        }
    }
    float s = 0.0f;
    for (int q = 0; q < 16; ++q) s += acc[q] + acc2[q];
    for (int q = 0; q < 8; ++q) s += v[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int FILL>
static void run(const char* name, float* out, int tiles, double per_tile_scale) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, FILL>), dim3(256), dim3(512), 0, 0, out, 4, 1.0f, 0.5f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<MODE, FILL>), dim3(256), dim3(512), 0, 0, out, tiles, 1.0f, 0.5f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %8.2f us per tile-round (%d rounds); %s\n", name, 1e3 * ms / tiles * per_tile_scale, tiles, "2 waves per SIMD");
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 256 * 512 * 4));
    const int T = 400;
    printf("per round every wave does: 120 MFMA (f32 32x32x2) and / or 1750 v_fma_f32, as the mode says\n");
    run<0, 0>("mfma   (chains only)", out, T, 1.0);
    run<1, 0>("valu   (1750 v_fma only)", out, T, 1.0);
    run<2, 0>("serial (valu then chain)", out, T, 1.0);
    run<3, 0>("split  (4 waves chain, 4 valu)", out, T, 1.0);
    run<4, 4>("inter1 fill 4  (1 chain, 480 fma)", out, T, 1.0);
    run<4, 8>("inter1 fill 8  (1 chain, 960 fma)", out, T, 1.0);
    run<4, 14>("inter1 fill 14 (1 chain, 1680 fma)", out, T, 1.0);
    run<5, 4>("inter2 fill 4  (2 chains = 2 tiles, 960 fma)", out, T / 2, 0.5);
    run<5, 7>("inter2 fill 7  (2 chains, 1680 fma)", out, T / 2, 0.5);
    run<5, 10>("inter2 fill 10 (2 chains, 2400 fma)", out, T / 2, 0.5);
    run<5, 14>("inter2 fill 14 (2 chains, 3360 fma)", out, T / 2, 0.5);
    return 0;
}
