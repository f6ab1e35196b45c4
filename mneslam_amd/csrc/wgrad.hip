// wgrad.hip -- decoder weight gradients from the per-sample tape: four skinny GEMMs
//     dW[o][i] = sum_t dY[t][o] * X[t][i]          (K = number of contributing samples, ~1e5)
// on the matrix cores with v_mfma_f32_32x32x2_f32 (exact fp32 at the vector rate).  Operands are
// loaded straight from the tape in fragment layout -- lane l reads dY[t0+(l>>5)][o0+(l&31)] and
// X[t0+(l>>5)][i0+(l&31)], i.e. two coalesced 128-B row segments per instruction, no LDS.
// Each wave owns a contiguous slice of tape rows and writes one partial result; a second tiny
// kernel sums the partials in a fixed order (deterministic given the tape).
#include "mne_device.h"
#include "mne_launch.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmDesc { int offA, OUT, offB, IN, poff; };

template <int TM, int TN>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradArgs a, GemmDesc g, int row_stride, int nparam) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nw = gridDim.x * (blockDim.x >> 6);
    const int n = *a.tape_rows;
    int per = (n + nw - 1) / nw;
    per = (per + 1) & ~1;
    const int t0 = gw * per;
    const int t1 = (t0 + per < n) ? t0 + per : n;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int q = 0; q < TN; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][q][e] = 0.0f;
    const int col = lane & 31, kk = lane >> 5;
    for (int t = t0; t < t1; t += 2) {
        const int tt = t + kk;
        const bool ok = tt < t1;
        const float* row = a.tape + (size_t)(ok ? tt : t) * row_stride;
        float av[TM], bv[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            const int o = 32 * m + col;
            av[m] = (ok && o < g.OUT) ? row[g.offA + o] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < TN; ++q) {
            const int i = 32 * q + col;
            bv[q] = (ok && i < g.IN) ? row[g.offB + i] : 0.0f;
        }
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int q = 0; q < TN; ++q)
                acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[q], acc[m][q], 0, 0, 0);
    }
    float* out = a.partials + (size_t)gw * nparam + g.poff;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int q = 0; q < TN; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int o = 32 * m + (e & 3) + 8 * (e >> 2) + 4 * kk;
                const int i = 32 * q + col;
                if (o < g.OUT && i < g.IN) out[o * g.IN + i] = acc[m][q][e];
            }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradArgs a, int nparam) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nparam) return;
    float s = 0.0f;
    for (int w = 0; w < a.n_waves; ++w) s += a.partials[(size_t)w * nparam + e];
    a.grad_out[e] = s;
}

// scalar cross-check of the MFMA path (impl = 1): one thread per output element
__global__ __launch_bounds__(256) void wgrad_scalar_kernel(WgradArgs a, GemmDesc g, int row_stride) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= g.OUT * g.IN) return;
    const int o = e / g.IN, i = e % g.IN;
    const int n = *a.tape_rows;
    float s = 0.0f;
    for (int t = 0; t < n; ++t) {
        const float* row = a.tape + (size_t)t * row_stride;
        s = fmaf(row[g.offA + o], row[g.offB + i], s);
    }
    a.grad_out[g.poff + e] = s;
}

#define MNE_WGRAD_BLOCKS 128      // x 4 waves = 512 partial results

int mne_wgrad_waves(void) { return MNE_WGRAD_BLOCKS * 4; }

template <int HID, int HIDC, bool CP>
static int launch_wgrad(WgradArgs a, int impl, hipStream_t st) {
    typedef DecDims<HID, HIDC, CP> D;
    const GemmDesc g1 = {D::T_DH, HID, D::T_X, MNE_IN1, D::P_SDF0};
    const GemmDesc g2 = {D::T_DOUT, MNE_OUT1, D::T_H, HID, D::P_SDF1};
    const GemmDesc g3 = {D::T_DHC, HIDC, D::T_CIN, D::CIN, D::P_COL0};
    const GemmDesc g4 = {D::T_DC, 3, D::T_HC, HIDC, D::P_COL1};
    if (impl == 1) {
        const GemmDesc gs[4] = {g1, g2, g3, g4};
        for (int k = 0; k < 4; ++k)
            MNE_LAUNCH(wgrad_scalar_kernel, (gs[k].OUT * gs[k].IN + 255) / 256, 256, 0, st, a, gs[k], D::ROW);
        return 0;
    }
    a.n_waves = MNE_WGRAD_BLOCKS * 4;
    MNE_LAUNCH((wgrad_mfma_kernel<HID / 32, 4>), MNE_WGRAD_BLOCKS, 256, 0, st, a, g1, D::ROW, D::NPARAM);
    MNE_LAUNCH((wgrad_mfma_kernel<1, HID / 32>), MNE_WGRAD_BLOCKS, 256, 0, st, a, g2, D::ROW, D::NPARAM);
    MNE_LAUNCH((wgrad_mfma_kernel<HIDC / 32, D::CINP / 32>), MNE_WGRAD_BLOCKS, 256, 0, st, a, g3, D::ROW, D::NPARAM);
    MNE_LAUNCH((wgrad_mfma_kernel<1, HIDC / 32>), MNE_WGRAD_BLOCKS, 256, 0, st, a, g4, D::ROW, D::NPARAM);
    MNE_LAUNCH(wgrad_reduce_kernel, (D::NPARAM + 255) / 256, 256, 0, st, a, D::NPARAM);
    return 0;
}

int mne_launch_wgrad(const mne_scene_t& sc, WgradArgs a, int impl, hipStream_t st) {
    const bool cp = sc.n_sets == 2;
    if (sc.hidden == 32 && sc.hidden_color == 32) return cp ? launch_wgrad<32, 32, true>(a, impl, st) : launch_wgrad<32, 32, false>(a, impl, st);
    if (sc.hidden == 64 && sc.hidden_color == 64) return cp ? launch_wgrad<64, 64, true>(a, impl, st) : launch_wgrad<64, 64, false>(a, impl, st);
    return -2;
}
