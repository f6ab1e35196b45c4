// wgrad.hip -- decoder weight gradients from the per-sample tape: four skinny GEMMs
//     dW[o][i] = sum_t dY[t][o] * X[t][i]          (K = number of contributing samples, ~1e5)
// on the matrix cores with v_mfma_f32_32x32x2_f32 (exact fp32 at the vector rate).  Operands are
// loaded straight from the tape in fragment layout -- lane l reads dY[t0+(l>>5)][o0+(l&31)] and
// X[t0+(l>>5)][i0+(l&31)], i.e. two coalesced 128-B row segments per instruction, no LDS.  Eight
// tape rows (4 k-steps) are loaded per loop trip before their MFMAs issue, so each wave keeps
// 4*(TM+TN) loads in flight.  Each wave owns a contiguous slice of tape rows and writes one partial
// result; a second small kernel sums the partials in a fixed order (deterministic given the tape).
#include "mne_device.h"
#include "mne_launch.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// offA/OUT: tape column and width of dY; offB/IN: tape column and width of X; skip: one tape column
// of X that is not a weight input (the sdf slot inside the colour-net input), -1 = none;
// poff/ld: offset and row length of the matrix inside the decoder parameter buffer.
struct GemmDesc { int offA, OUT, offB, IN, skip, poff, ld; };

#define WG_KSTEPS 4

template <int TM, int TN>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradArgs a, GemmDesc g, int row_stride, int nparam) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nw = gridDim.x * (blockDim.x >> 6);
    const int n = *a.tape_rows;
    int per = (n + nw - 1) / nw;
    per = (per + 2 * WG_KSTEPS - 1) / (2 * WG_KSTEPS) * (2 * WG_KSTEPS);
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
    for (int t = t0; t < t1; t += 2 * WG_KSTEPS) {
        float av[WG_KSTEPS][TM], bv[WG_KSTEPS][TN];
#pragma unroll
        for (int ks = 0; ks < WG_KSTEPS; ++ks) {
            const int tt = t + 2 * ks + kk;
            const bool ok = tt < t1;
            const float* row = a.tape + (size_t)(ok ? tt : t0) * row_stride;
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                const int o = 32 * m + col;
                av[ks][m] = (ok && o < g.OUT) ? row[g.offA + o] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < TN; ++q) {
                const int i = 32 * q + col;
                bv[ks][q] = (ok && i < g.IN) ? row[g.offB + i] : 0.0f;
            }
        }
#pragma unroll
        for (int ks = 0; ks < WG_KSTEPS; ++ks)
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int q = 0; q < TN; ++q)
                    acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks][m], bv[ks][q], acc[m][q], 0, 0, 0);
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
                if (o < g.OUT && i < g.IN && i != g.skip) out[o * g.ld + (g.skip >= 0 && i > g.skip ? i - 1 : i)] = acc[m][q][e];
            }
}

// All four GEMMs in ONE pass over the tape (HID = HIDC = 32): every wave reads each of its tape rows once --
// A operands dh(32) | dout(16) | dhc(32) | dc(4), B operands x(112) | h(32) | cin(CINP) | hc(32) -- and
// keeps the 8 (or 10) 32x32 accumulator tiles of dW1, dW2, dV1, dV2 in registers.
// The pass is latency-bound (a wave issues one batch of row loads, waits, multiplies), so the tape is cut
// finely: 4 waves per workgroup, each with its own slice, 2*WG_KS rows per batch; the four waves'
// accumulators are then summed tile by tile through 12 KiB of LDS (plain stores/loads, fixed order) and
// the workgroup writes ONE partial result, so the second-stage reduction stays small.
#ifndef WG_KS
#define WG_KS 4
#endif
#ifndef WG_FUSED_BLOCKS
#define WG_FUSED_BLOCKS 256          // x 4 waves; measured best next to the concurrent tile_adam_kernel (profiles/r01_wgrad_variants.txt)
#endif
template <int HID, int HIDC, bool CP>
__global__ __launch_bounds__(256) void wgrad_fused_kernel(WgradArgs a) {
    typedef DecDims<HID, HIDC, CP> D;
    static_assert(HID == 32 && HIDC == 32, "fused weight-gradient kernel is built for the 2x32 decoders");
    constexpr int TNC = D::CINP / 32;
    constexpr int KS = WG_KS;
    constexpr int NTILE = 4 + 1 + TNC + 1;
    __shared__ float red[3][64 * 16];                            // one 32x32 tile of waves 1..3
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wv;
    const int nw = gridDim.x * 4;
    const int n = *a.tape_rows;
    int per = (n + nw - 1) / nw;
    per = (per + 2 * KS - 1) / (2 * KS) * (2 * KS);
    const int t0 = gw * per < n ? gw * per : n;
    const int t1 = (t0 + per < n) ? t0 + per : n;
    f32x16 acc[NTILE];                                           // w1[0..3] | w2 | v1[0..TNC) | v2
#pragma unroll
    for (int q = 0; q < NTILE; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    const int col = lane & 31, kk = lane >> 5;
    for (int t = t0; t < t1; t += 2 * KS) {
        float adh[KS], ado[KS], adc[KS], adq[KS], bx[KS][4], bh[KS], bc[KS][TNC], bhc[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int tt = t + 2 * ks + kk;
            const bool ok = tt < t1;
            const float* row = a.tape + (size_t)(ok ? tt : t0) * D::ROW;
            adh[ks] = ok ? row[D::T_DH + col] : 0.f;
            ado[ks] = (ok && col < MNE_OUT1) ? row[D::T_DOUT + col] : 0.f;
            adc[ks] = ok ? row[D::T_DHC + col] : 0.f;
            adq[ks] = (ok && col < 3) ? row[D::T_DC + col] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) bx[ks][q] = (ok && 32 * q + col < MNE_IN1) ? row[D::T_X + 32 * q + col] : 0.f;
            bh[ks] = ok ? row[D::T_H + col] : 0.f;
#pragma unroll
            for (int q = 0; q < TNC; ++q) bc[ks][q] = ok ? row[D::T_CIN + 32 * q + col] : 0.f;
            bhc[ks] = ok ? row[D::T_HC + col] : 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(adh[ks], bx[ks][q], acc[q], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(ado[ks], bh[ks], acc[4], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < TNC; ++q) acc[5 + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(adc[ks], bc[ks][q], acc[5 + q], 0, 0, 0);
            acc[5 + TNC] = __builtin_amdgcn_mfma_f32_32x32x2f32(adq[ks], bhc[ks], acc[5 + TNC], 0, 0, 0);
        }
    }
    // ---- sum of the four waves, tile by tile, in wave order
#pragma unroll
    for (int q = 0; q < NTILE; ++q) {
        if (wv > 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[wv - 1][e * 64 + lane] = acc[q][e];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int w = 0; w < 3; ++w)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[q][e] += red[w][e * 64 + lane];
        }
        __syncthreads();
    }
    if (wv != 0) return;
    float* out = a.partials + (size_t)blockIdx.x * D::NPARAM;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int o = (e & 3) + 8 * (e >> 2) + 4 * kk;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (32 * q + col < MNE_IN1) out[D::P_SDF0 + o * MNE_IN1 + 32 * q + col] = acc[q][e];
        if (o < MNE_OUT1) out[D::P_SDF1 + o * HID + col] = acc[4][e];
#pragma unroll
        for (int q = 0; q < TNC; ++q) {
            const int i = 32 * q + col;                      // tape column of the colour-net input
            if (i != D::CINB) out[D::P_COL0 + o * D::CIN + (i > D::CINB ? i - 1 : i)] = acc[5 + q][e];
        }
        if (o < 3) out[D::P_COL1 + o * HIDC + col] = acc[5 + TNC][e];
    }
}

// The same single pass for the 2x64 decoders: the 64 hidden rows of dW1 / dV1 need 8 + 2*TNC accumulator tiles,
// too many for one wave, so a pair of waves shares a tape slice -- wave parity hh takes hidden rows [32 hh, 32 hh + 32) of
// dW1 and dV1; the even wave also accumulates dW2 (16 x 64), the odd one dV2 (3 x 64).  Waves 2,3 of the workgroup
// work on a second slice and are summed into waves 0,1 through LDS; one partial per workgroup.
template <bool CP>
__global__ __launch_bounds__(256) void wgrad_fused64_kernel(WgradArgs a) {
    typedef DecDims<64, 64, CP> D;
    constexpr int TNC = D::CINP / 32;
    constexpr int KS = 2;
    constexpr int NTILE = 4 + TNC + 2;
    __shared__ float red[2][64 * 16];                            // one 32x32 tile of waves 2 and 3
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int hh = wv & 1;                                       // hidden-row half of this wave
    const int slice = blockIdx.x * 2 + (wv >> 1);
    const int nslice = gridDim.x * 2;
    const int n = *a.tape_rows;
    int per = (n + nslice - 1) / nslice;
    per = (per + 2 * KS - 1) / (2 * KS) * (2 * KS);
    const int t0 = slice * per < n ? slice * per : n;
    const int t1 = (t0 + per < n) ? t0 + per : n;
    f32x16 acc[NTILE];                                           // w1[0..3] | v1[0..TNC) | (w2 or v2)[0..1]
#pragma unroll
    for (int q = 0; q < NTILE; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    const int col = lane & 31, kk = lane >> 5;
    // the small GEMM of this wave: even = dout (16 rows) x h (64), odd = dc (3 rows) x hc (64)
    const int offA2 = hh ? D::T_DC : D::T_DOUT, rowsA2 = hh ? 3 : MNE_OUT1, offB2 = hh ? D::T_HC : D::T_H;
    for (int t = t0; t < t1; t += 2 * KS) {
        float adh[KS], adc[KS], a2[KS], bx[KS][4], bc[KS][TNC], b2[KS][2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int tt = t + 2 * ks + kk;
            const bool ok = tt < t1;
            const float* row = a.tape + (size_t)(ok ? tt : t0) * D::ROW;
            adh[ks] = ok ? row[D::T_DH + 32 * hh + col] : 0.f;
            adc[ks] = ok ? row[D::T_DHC + 32 * hh + col] : 0.f;
            a2[ks] = (ok && col < rowsA2) ? row[offA2 + col] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) bx[ks][q] = (ok && 32 * q + col < MNE_IN1) ? row[D::T_X + 32 * q + col] : 0.f;
#pragma unroll
            for (int q = 0; q < TNC; ++q) bc[ks][q] = ok ? row[D::T_CIN + 32 * q + col] : 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q) b2[ks][q] = ok ? row[offB2 + 32 * q + col] : 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(adh[ks], bx[ks][q], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < TNC; ++q) acc[4 + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(adc[ks], bc[ks][q], acc[4 + q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[4 + TNC + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[ks], b2[ks][q], acc[4 + TNC + q], 0, 0, 0);
        }
    }
    // ---- waves 2,3 -> waves 0,1 (same hh), tile by tile
#pragma unroll
    for (int q = 0; q < NTILE; ++q) {
        if (wv >= 2) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[wv - 2][e * 64 + lane] = acc[q][e];
        }
        __syncthreads();
        if (wv < 2) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[q][e] += red[wv][e * 64 + lane];
        }
        __syncthreads();
    }
    if (wv >= 2) return;
    float* out = a.partials + (size_t)blockIdx.x * D::NPARAM;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int lo = (e & 3) + 8 * (e >> 2) + 4 * kk;          // row inside the 32-row tile
        const int o = 32 * hh + lo;                              // hidden unit
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (32 * q + col < MNE_IN1) out[D::P_SDF0 + o * MNE_IN1 + 32 * q + col] = acc[q][e];
#pragma unroll
        for (int q = 0; q < TNC; ++q) {
            const int i = 32 * q + col;                          // tape column of the colour-net input
            if (i != D::CINB) out[D::P_COL0 + o * D::CIN + (i > D::CINB ? i - 1 : i)] = acc[4 + q][e];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (hh == 0) { if (lo < MNE_OUT1) out[D::P_SDF1 + lo * 64 + 32 * q + col] = acc[4 + TNC + q][e]; }
            else if (lo < 3) out[D::P_COL1 + lo * 64 + 32 * q + col] = acc[4 + TNC + q][e];
        }
    }
}

// 32 parameters per block, 8 groups of partials per parameter, fixed summation order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradArgs a, int nparam) {
    __shared__ float part[8][32];
    const int e = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
    float s = 0.0f;
    if (e < nparam) {
#pragma unroll 8
        for (int w = grp; w < a.n_waves; w += 8) s += a.partials[(size_t)w * nparam + e];
    }
    part[grp][threadIdx.x & 31] = s;
    __syncthreads();
    if (grp == 0 && e < nparam) {
        float tot = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) tot += part[k][threadIdx.x];
        a.grad_out[e] = tot;
    }
}

// scalar cross-check of the MFMA path (impl = 1): one thread per output element
__global__ __launch_bounds__(256) void wgrad_scalar_kernel(WgradArgs a, GemmDesc g, int row_stride) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= g.OUT * g.ld) return;
    const int o = e / g.ld, c = e % g.ld;
    const int i = (g.skip >= 0 && c >= g.skip) ? c + 1 : c;       // tape column of parameter column c
    const int n = *a.tape_rows;
    float s = 0.0f;
    for (int t = 0; t < n; ++t) {
        const float* row = a.tape + (size_t)t * row_stride;
        s = fmaf(row[g.offA + o], row[g.offB + i], s);
    }
    a.grad_out[g.poff + e] = s;
}

#define MNE_WGRAD_BLOCKS 256      // x 4 waves = 1024 partial results

int mne_wgrad_waves(void) { return MNE_WGRAD_BLOCKS * 4; }

template <int HID, int HIDC, bool CP>
static int launch_wgrad(WgradArgs a, int impl, hipStream_t st) {
    typedef DecDims<HID, HIDC, CP> D;
    const GemmDesc g1 = {D::T_DH, HID, D::T_X, MNE_IN1, -1, D::P_SDF0, MNE_IN1};
    const GemmDesc g2 = {D::T_DOUT, MNE_OUT1, D::T_H, HID, -1, D::P_SDF1, HID};
    const GemmDesc g3 = {D::T_DHC, HIDC, D::T_CIN, D::CINP, D::CINB, D::P_COL0, D::CIN};
    const GemmDesc g4 = {D::T_DC, 3, D::T_HC, HIDC, -1, D::P_COL1, HIDC};
    if (impl == 1) {
        const GemmDesc gs[4] = {g1, g2, g3, g4};
        for (int k = 0; k < 4; ++k)
            MNE_LAUNCH(wgrad_scalar_kernel, (gs[k].OUT * gs[k].ld + 255) / 256, 256, 0, st, a, gs[k], D::ROW);
        return 0;
    }
    if constexpr (HID == 32 && HIDC == 32) {
        if (impl == 0) {
            // one partial per WORKGROUP: as many workgroups as there are partial slots, fewer for short tapes
            int blocks = a.n_waves * (64 / (2 * WG_KS * 4));      // n_waves = 64-row chunks of the caller's bound: >= 2*WG_KS rows per wave
            blocks = blocks < 1 ? 1 : (blocks > WG_FUSED_BLOCKS ? WG_FUSED_BLOCKS : blocks);
            a.n_waves = blocks;
            MNE_LAUNCH((wgrad_fused_kernel<HID, HIDC, CP>), blocks, 256, 0, st, a);
            MNE_LAUNCH(wgrad_reduce_kernel, (D::NPARAM + 31) / 32, 256, 0, st, a, D::NPARAM);
            return 0;
        }
    }
    if constexpr (HID == 64 && HIDC == 64) {
        if (impl == 0) {
            int blocks = a.n_waves * (64 / (2 * 2 * 2));          // >= 4 rows per slice; two slices per workgroup
            blocks = blocks < 1 ? 1 : (blocks > WG_FUSED_BLOCKS ? WG_FUSED_BLOCKS : blocks);
            a.n_waves = blocks;
            MNE_LAUNCH((wgrad_fused64_kernel<CP>), blocks, 256, 0, st, a);
            MNE_LAUNCH(wgrad_reduce_kernel, (D::NPARAM + 31) / 32, 256, 0, st, a, D::NPARAM);
            return 0;
        }
    }
    int blocks = (a.n_waves + 3) / 4;                    // caller's bound on the tape length
    blocks = blocks < 1 ? 1 : (blocks > MNE_WGRAD_BLOCKS ? MNE_WGRAD_BLOCKS : blocks);
    a.n_waves = blocks * 4;
    MNE_LAUNCH((wgrad_mfma_kernel<HID / 32, 4>), blocks, 256, 0, st, a, g1, D::ROW, D::NPARAM);
    MNE_LAUNCH((wgrad_mfma_kernel<1, HID / 32>), blocks, 256, 0, st, a, g2, D::ROW, D::NPARAM);
    MNE_LAUNCH((wgrad_mfma_kernel<HIDC / 32, D::CINP / 32>), blocks, 256, 0, st, a, g3, D::ROW, D::NPARAM);
    MNE_LAUNCH((wgrad_mfma_kernel<1, HIDC / 32>), blocks, 256, 0, st, a, g4, D::ROW, D::NPARAM);
    MNE_LAUNCH(wgrad_reduce_kernel, (D::NPARAM + 31) / 32, 256, 0, st, a, D::NPARAM);
    return 0;
}

int mne_launch_wgrad(const mne_scene_t& sc, WgradArgs a, int impl, hipStream_t st) {
    const bool cp = sc.n_sets == 2;
    if (sc.hidden == 32 && sc.hidden_color == 32) return cp ? launch_wgrad<32, 32, true>(a, impl, st) : launch_wgrad<32, 32, false>(a, impl, st);
    if (sc.hidden == 64 && sc.hidden_color == 64) return cp ? launch_wgrad<64, 64, true>(a, impl, st) : launch_wgrad<64, 64, false>(a, impl, st);
    return -2;
}
