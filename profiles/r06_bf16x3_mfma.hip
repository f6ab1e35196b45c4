// r06_bf16x3_mfma.hip -- could the decoder's fp32 MFMA chain (v_mfma_f32_32x32x2_f32: 64 cycles, excludes VALU issue) be replaced by
// bf16 MFMAs on three-way bf16 splits of both operands (a = a1 + a2 + a3, 6 products a1b1 a1b2 a2b1 a1b3 a3b1 a2b2: ~2^-24 relative)?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off profiles/r06_bf16x3_mfma.hip -o profiles/_bin/bf16x3 && profiles/_bin/bf16x3
// (1) accuracy of C = A (32 x K) . B (K x 32), K = 112, against a double-precision host product: fp32 MFMA chain vs the 6-product split;
// (2) time per "tile" at 2 waves per SIMD on every CU: 120 fp32 MFMAs vs 15 K-steps x 6 bf16 MFMAs + the split of 96 activation values per lane,
//     each alone and with the ~1400 VALU instructions of a decode tile beside it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int K = 112;

// three truncated bf16 pieces of 8 floats -> three packed operands (4 dwords each)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
    unsigned h1[8], h2[8], h3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned b = __float_as_uint(x[j]);
        h1[j] = b & 0xffff0000u;
        const float r1 = x[j] - __uint_as_float(h1[j]);
        h2[j] = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(h2[j]);
        h3[j] = __float_as_uint(r2) & 0xffff0000u;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        p1[q] = (h1[2 * q] >> 16) | h1[2 * q + 1];
        p2[q] = (h2[2 * q] >> 16) | h2[2 * q + 1];
        p3[q] = (h3[2 * q] >> 16) | h3[2 * q + 1];
    }
}
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 six(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 c) {
    c = mf(a[2], b[0], c); c = mf(a[0], b[2], c); c = mf(a[1], b[1], c);      // smallest terms first
    c = mf(a[1], b[0], c); c = mf(a[0], b[1], c); c = mf(a[0], b[0], c);
    return c;
}

// accuracy: one wave.  A [32][K], B [K][32] row-major fp32; C32 / C16 [32][32]
__global__ void acc_kernel(const float* A, const float* B, float* C32, float* C16) {
    const int l = threadIdx.x, n = l & 31, h = l >> 5;
    f32x16 c; for (int q = 0; q < 16; ++q) c[q] = 0.f;
    for (int k = 0; k < K; k += 2) c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[n * K + k + h], B[(k + h) * 32 + n], c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) C32[((i / 4) * 8 + h * 4 + (i % 4)) * 32 + n] = c[i];
    for (int q = 0; q < 16; ++q) c[q] = 0.f;
    for (int k = 0; k < K; k += 16) {
        float xa[8], xb[8];
        for (int j = 0; j < 8; ++j) { xa[j] = A[n * K + k + 8 * h + j]; xb[j] = B[(k + 8 * h + j) * 32 + n]; }
        u32x4 a[3], b[3];
        split8(xa, a[0], a[1], a[2]); split8(xb, b[0], b[1], b[2]);
        c = six(a, b, c);
    }
    for (int i = 0; i < 16; ++i) C16[((i / 4) * 8 + h * 4 + (i % 4)) * 32 + n] = c[i];
}

template <int NV>
__device__ __forceinline__ void valu_block(float (&v)[8], float k) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], k, 1.0f);
}
// MODE 0: 120 fp32 MFMAs; 1: 15 steps x 6 bf16 MFMAs, weights pre-split (LDS-resident in the real kernel), activations split per step (8 values);
// 2 / 3: the same with 1400 VALU per tile in front (what a decode tile issues besides the chain)
template <int MODE>
__global__ __launch_bounds__(512) void time_kernel(float* out, int tiles, float s) {
    f32x16 c; for (int q = 0; q < 16; ++q) c[q] = 0.f;
    float v[8]; for (int q = 0; q < 8; ++q) v[q] = (float)(threadIdx.x + q) * 1e-3f;
    u32x4 w[3];
    for (int q = 0; q < 4; ++q) { w[0][q] = 0x3f803f80u + threadIdx.x; w[1][q] = 0x3b803b80u; w[2][q] = 0x37803780u; }
    for (int t = 0; t < tiles; ++t) {
        if (MODE >= 2) { for (int r = 0; r < 14; ++r) valu_block<100>(v, s); }
        if (MODE == 0 || MODE == 2) {
            for (int i = 0; i < 120; i += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(s, v[j] , c, 0, 0, 0);
            }
        } else {
#pragma unroll 1
            for (int st = 0; st < 15; ++st) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = v[j] + c[j];                 // (activations depend on the running state)
                u32x4 b[3];
                split8(x, b[0], b[1], b[2]);
                c = six(w, b, c);
            }
        }
    }
    float r = 0; for (int q = 0; q < 16; ++q) r += c[q]; for (int q = 0; q < 8; ++q) r += v[q];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int MODE> float run(float* d, int tiles) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    time_kernel<MODE><<<256, 512>>>(d, 4, 0.999f);
    CHECK(hipEventRecord(e0)); time_kernel<MODE><<<256, 512>>>(d, tiles, 0.999f); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1000.0f / tiles;
}
int main() {
    std::vector<float> A(32 * K), B(K * 32), c32(1024), c16(1024);
    srand(1);
    for (auto& x : A) x = ((rand() % 20001) - 10000) * 1e-4f * 0.3f;        // weights ~U(-0.3, 0.3)
    for (auto& x : B) x = ((rand() % 20001) - 10000) * 1e-4f;               // activations ~U(-1, 1)
    float *dA, *dB, *dC, *dD, *dO;
    CHECK(hipMalloc(&dA, A.size() * 4)); CHECK(hipMalloc(&dB, B.size() * 4)); CHECK(hipMalloc(&dC, 4096)); CHECK(hipMalloc(&dD, 4096)); CHECK(hipMalloc(&dO, 256 * 512 * 4));
    CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    acc_kernel<<<1, 64>>>(dA, dB, dC, dD);
    CHECK(hipMemcpy(c32.data(), dC, 4096, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(c16.data(), dD, 4096, hipMemcpyDeviceToHost));
    double e32 = 0, e16 = 0, m32 = 0, m16 = 0, scale = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        double ref = 0, mag = 0;
        for (int k = 0; k < K; ++k) { ref += (double)A[m * K + k] * B[k * 32 + n]; mag += fabs((double)A[m * K + k] * B[k * 32 + n]); }
        const double d32 = fabs(c32[m * 32 + n] - ref) / mag, d16 = fabs(c16[m * 32 + n] - ref) / mag;
        e32 += d32; e16 += d16; m32 = fmax(m32, d32); m16 = fmax(m16, d16); scale += mag;
    }
    printf("accuracy, K = %d, error / sum |a b|: fp32 MFMA chain mean %.3g max %.3g | bf16 x3 (6 products) mean %.3g max %.3g   (2^-24 = %.3g)\n", K, e32 / 1024, m32, e16 / 1024, m16, ldexp(1.0, -24));
    for (int rep = 0; rep < 2; ++rep)
        printf("us per tile, 2 waves per SIMD: fp32 chain %.2f | bf16 x3 + split %.2f | 1400 VALU + fp32 chain %.2f | 1400 VALU + bf16 x3 %.2f\n",
               run<0>(dO, 400), run<1>(dO, 400), run<2>(dO, 400), run<3>(dO, 400));
    return 0;
}
