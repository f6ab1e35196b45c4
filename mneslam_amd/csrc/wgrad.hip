// wgrad.hip -- decoder weight gradients from the per-sample tape: four skinny GEMMs
//     dW[o][i] = sum_t dY[t][o] * X[t][i]          (K = samples of the tiles that received gradient, ~1e5)
// on the matrix cores with v_mfma_f32_32x32x2_f32 (exact fp32 at the vector rate).  Operands are
// loaded straight from the tape in fragment layout -- lane l reads dY[t0+(l>>5)][o0+(l&31)] and
// X[t0+(l>>5)][i0+(l&31)], i.e. two coalesced 128-B row segments per instruction, no LDS.
//
// Rows: tape row = ray * S + sample; ray_kernel leaves in ray_tiles[r] the number of leading 32-sample tiles of ray r
// whose rows are complete (forward half by the decode, backward half -- all zeros for samples without gradient -- by
// ray_kernel).  Every wave owns a contiguous range of RAYS and walks their rows in order, so the summation order
// is fixed by the batch (deterministic given the tape); a second small kernel sums the partials in a fixed order.
#include "mne_device.h"
#include "mne_launch.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// The four GEMMs: G = 0 dW1 = DH x X, 1 dW2 = DOUT x H, 2 dV1 = DHC x [pos|(cf)|out16], 3 dV2 = DC x HC.
template <int HID, int HIDC, bool CP, int G>
struct Gemm {
    typedef DecDims<HID, HIDC, CP> D;
    static constexpr int OUT = G == 0 ? HID : G == 1 ? MNE_OUT1 : G == 2 ? HIDC : 3;
    static constexpr int IN = G == 0 ? MNE_IN1 : G == 1 ? HID : G == 2 ? D::CINP : HIDC;      // GEMM view of the input
    static constexpr int OFFA = G == 0 ? D::T_DH : G == 1 ? D::T_DOUT : G == 2 ? D::T_DHC : D::T_DC;
    static constexpr int POFF = G == 0 ? D::P_SDF0 : G == 1 ? D::P_SDF1 : G == 2 ? D::P_COL0 : D::P_COL1;
    static constexpr int LD = G == 0 ? MNE_IN1 : G == 1 ? HID : G == 2 ? D::CIN : HIDC;       // row length of the parameter
    static constexpr int TM = (OUT + 31) / 32, TN = (IN + 31) / 32;
    // tape column of input element c
    __host__ __device__ static constexpr int bcol(int c) {
        return G == 0 ? D::T_X + c : G == 1 ? D::T_H + c : G == 2 ? D::cin_col(c) : D::T_HC + c;
    }
    // parameter column of input element c; -1 = not a weight input (the sdf slot of out16 inside the colour input)
    __host__ __device__ static constexpr int pcol(int c) {
        return G != 2 ? c : (c < D::CINB ? c : (c == D::CINB ? -1 : c - 1));
    }
};

// rays [r0, r1) of wave `gw` out of `nw`
__device__ __forceinline__ void ray_range(int R, int gw, int nw, int& r0, int& r1) {
    const int per = (R + nw - 1) / nw;
    r0 = gw * per < R ? gw * per : R;
    r1 = r0 + per < R ? r0 + per : R;
}
__device__ __forceinline__ int ray_rows(const WgradArgs& a, int r) {
    const int n = a.ray_tiles[r] * 32;
    return n < a.S ? n : a.S;
}

#define WG_KSTEPS 4

template <int HID, int HIDC, bool CP, int G>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradArgs a) {
    typedef Gemm<HID, HIDC, CP, G> M;
    typedef DecDims<HID, HIDC, CP> D;
    constexpr int TM = M::TM, TN = M::TN;
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int r0, r1;
    ray_range(a.R, gw, gridDim.x * (blockDim.x >> 6), r0, r1);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int q = 0; q < TN; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][q][e] = 0.0f;
    const int col = lane & 31, kk = lane >> 5;
    int bc[TN];
#pragma unroll
    for (int q = 0; q < TN; ++q) bc[q] = 32 * q + col < M::IN ? M::bcol(32 * q + col) : -1;
    for (int r = r0; r < r1; ++r) {
        const int n = ray_rows(a, r);
        const float* base = a.tape + (size_t)r * a.S * D::ROW;
        for (int t = 0; t < n; t += 2 * WG_KSTEPS) {
            float av[WG_KSTEPS][TM], bv[WG_KSTEPS][TN];
#pragma unroll
            for (int ks = 0; ks < WG_KSTEPS; ++ks) {
                const int tt = t + 2 * ks + kk;
                const bool ok = tt < n;
                const float* row = base + (size_t)(ok ? tt : 0) * D::ROW;
#pragma unroll
                for (int m = 0; m < TM; ++m) {
                    const int o = 32 * m + col;
                    av[ks][m] = (ok && o < M::OUT) ? row[M::OFFA + o] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < TN; ++q) bv[ks][q] = (ok && bc[q] >= 0) ? row[bc[q]] : 0.0f;
            }
#pragma unroll
            for (int ks = 0; ks < WG_KSTEPS; ++ks)
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int q = 0; q < TN; ++q)
                        acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks][m], bv[ks][q], acc[m][q], 0, 0, 0);
        }
    }
    float* out = a.partials + (size_t)gw * D::NPARAM + M::POFF;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int q = 0; q < TN; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int o = 32 * m + (e & 3) + 8 * (e >> 2) + 4 * kk;
                const int i = 32 * q + col;
                const int pc = i < M::IN ? M::pcol(i) : -1;
                if (o < M::OUT && pc >= 0) out[o * M::LD + pc] = acc[m][q][e];
            }
}

// All four GEMMs in ONE pass over the tape (HID = HIDC = 32): every wave reads each of its tape rows once --
// A operands dh(32) | dout(16) | dhc(32) | dc(4), B operands x(112) | h(32) | [pos|(cf)|out16] | hc(32) -- and
// keeps the 8 (or 10) 32x32 accumulator tiles of dW1, dW2, dV1, dV2 in registers.
// The pass is latency-bound (a wave issues one batch of row loads, waits, multiplies), so the rays are cut
// finely: 4 waves per workgroup, each with its own ray range, 2*WG_KS rows per batch; the four waves'
// accumulators are then summed tile by tile through 12 KiB of LDS (plain stores/loads, fixed order) and
// the workgroup writes ONE partial result, so the second-stage reduction stays small.
#ifndef WG_KS
#define WG_KS 4
#endif
#ifndef WG_KS64
#define WG_KS64 4                    // row pairs per batch of the 2x64 kernel (2 / 4 / 8: profiles/r04_wgrad64_ks.txt, r04_hash_slice_size.txt)
#endif
#ifndef WG_FUSED_BLOCKS
#define WG_FUSED_BLOCKS 256          // x 4 waves; one partial result per workgroup
#endif
// PART: the four GEMMs are split over TWO kinds of workgroups that share a ray range and a partial-result slot --
// part 0 accumulates dW1 = DH x X (4 tiles: reads dh and x), part 1 dW2, dV1, dV2 (4-5 tiles: reads dout, h, dhc, the colour
// input, dc, hc).  One wave holding all 8-10 accumulator tiles needed 200 registers: a single such wave per SIMD and,
// worse, no room beside it for a second workgroup of the plane update that runs concurrently on the other stream
// (tile_adam_kernel 198 us alone, 249 us next to it; profiles/r03_tile_adam_ablation.txt).  Each part reads about half
// of a tape row, so the tape is still read once.
template <int HID, int HIDC, bool CP, int PART>
__device__ __forceinline__ void wgrad_fused_part(const WgradArgs& a, float (*red)[64 * 16]) {
    typedef DecDims<HID, HIDC, CP> D;
    constexpr int TNC = D::CINP / 32;
    constexpr int KS = WG_KS;
    constexpr int NTILE = PART == 0 ? 4 : 1 + TNC + 1;           // w1[0..3]  |  w2, v1[0..TNC), v2
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int slot = blockIdx.x >> 1, n_slots = gridDim.x >> 1;
    int r0, r1;
    ray_range(a.R, slot * 4 + wv, n_slots * 4, r0, r1);
    f32x16 acc[NTILE];
#pragma unroll
    for (int q = 0; q < NTILE; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    const int col = lane & 31, kk = lane >> 5;
    int ccol[TNC];                                               // tape columns of this lane's colour-input elements
#pragma unroll
    for (int q = 0; q < TNC; ++q) ccol[q] = D::cin_col(32 * q + col);
    for (int r = r0; r < r1; ++r) {
        const int n = ray_rows(a, r);
        const float* base = a.tape + (size_t)r * a.S * D::ROW;
        for (int t = 0; t < n; t += 2 * KS) {
            if (PART == 0) {
                float adh[KS], bx[KS][4];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int tt = t + 2 * ks + kk;
                    const bool ok = tt < n;
                    const float* row = base + (size_t)(ok ? tt : 0) * D::ROW;
                    adh[ks] = ok ? row[D::T_DH + col] : 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) bx[ks][q] = (ok && 32 * q + col < MNE_IN1) ? row[D::T_X + 32 * q + col] : 0.f;
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(adh[ks], bx[ks][q], acc[q], 0, 0, 0);
            } else {
                float ado[KS], adc[KS], adq[KS], bh[KS], bc[KS][TNC], bhc[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int tt = t + 2 * ks + kk;
                    const bool ok = tt < n;
                    const float* row = base + (size_t)(ok ? tt : 0) * D::ROW;
                    ado[ks] = (ok && col < MNE_OUT1) ? row[D::T_DOUT + col] : 0.f;
                    adc[ks] = ok ? row[D::T_DHC + col] : 0.f;
                    adq[ks] = (ok && col < 3) ? row[D::T_DC + col] : 0.f;
                    bh[ks] = ok ? row[D::T_H + col] : 0.f;
#pragma unroll
                    for (int q = 0; q < TNC; ++q) bc[ks][q] = ok ? row[ccol[q]] : 0.f;
                    bhc[ks] = ok ? row[D::T_HC + col] : 0.f;
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ado[ks], bh[ks], acc[0], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < TNC; ++q) acc[1 + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(adc[ks], bc[ks][q], acc[1 + q], 0, 0, 0);
                    acc[1 + TNC] = __builtin_amdgcn_mfma_f32_32x32x2f32(adq[ks], bhc[ks], acc[1 + TNC], 0, 0, 0);
                }
            }
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
    float* out = a.partials + (size_t)slot * D::NPARAM;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int o = (e & 3) + 8 * (e >> 2) + 4 * kk;
        if (PART == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (32 * q + col < MNE_IN1) out[D::P_SDF0 + o * MNE_IN1 + 32 * q + col] = acc[q][e];
        } else {
            if (o < MNE_OUT1) out[D::P_SDF1 + o * HID + col] = acc[0][e];
#pragma unroll
            for (int q = 0; q < TNC; ++q) {
                const int i = 32 * q + col;                      // element of the colour-net input [pos|(cf)|out16]
                if (i != D::CINB) out[D::P_COL0 + o * D::CIN + (i > D::CINB ? i - 1 : i)] = acc[1 + q][e];
            }
            if (o < 3) out[D::P_COL1 + o * HIDC + col] = acc[1 + TNC][e];
        }
    }
}

// (second launch bound = workgroups per CU the register allocation must allow: 4 -> 128 registers per lane, with colour planes
// 3 -> 168: the concurrent plane update keeps its two workgroups per CU (2 x 2 waves x 88 registers per SIMD))
template <int HID, int HIDC, bool CP>
__global__ __launch_bounds__(256, CP ? 3 : 4) void wgrad_fused_kernel(WgradArgs a) {
    static_assert(HID == 32 && HIDC == 32, "fused weight-gradient kernel is built for the 2x32 decoders");
    __shared__ float red[3][64 * 16];                            // one 32x32 tile of waves 1..3
    if (blockIdx.x & 1) wgrad_fused_part<HID, HIDC, CP, 1>(a, red);
    else wgrad_fused_part<HID, HIDC, CP, 0>(a, red);
}

// The same single pass for the 2x64 decoders: the 64 hidden rows of dW1 / dV1 need 8 + 2*TNC accumulator tiles,
// too many for one wave, so a pair of waves shares a ray range -- wave parity hh takes hidden rows [32 hh, 32 hh + 32) of
// dW1 and dV1; the even wave also accumulates dW2 (16 x 64), the odd one dV2 (3 x 64).  Waves 2,3 of the workgroup
// work on a second range and are summed into waves 0,1 through LDS; one partial per workgroup.
// Round 3: like the 2x32 kernel it comes in two register-light PARTS selected by workgroup parity (it held 256 VGPRs +
// 144-240 AGPRs at one wave per SIMD, which kept the concurrent plane update down to one workgroup on its CUs:
// tile_adam_kernel 330 us beside it against 229 us beside the 2x32 kernel).  PART 0 = dW1 (4 tiles), PART 1 = dV1 (TNC
// tiles) + dW2 / dV2 (2 tiles); the two parts read disjoint tape columns, so the rows are still read once.
template <bool CP, int PART>
__device__ __forceinline__ void wgrad_fused64_part(const WgradArgs& a, float (*red)[64 * 16]) {
    typedef DecDims<64, 64, CP> D;
    constexpr int TNC = D::CINP / 32;
    constexpr int KS = WG_KS64;
    constexpr int NTILE = PART == 0 ? 4 : TNC + 2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int hh = wv & 1;                                       // hidden-row half of this wave
    const int slot = blockIdx.x >> 1, n_slots = gridDim.x >> 1;
    int r0, r1;
    ray_range(a.R, slot * 2 + (wv >> 1), n_slots * 2, r0, r1);
    f32x16 acc[NTILE];                                           // PART 0: w1[0..3]   PART 1: v1[0..TNC) | (w2 or v2)[0..1]
#pragma unroll
    for (int q = 0; q < NTILE; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    const int col = lane & 31, kk = lane >> 5;
    // the small GEMM of this wave: even = dout (16 rows) x h (64), odd = dc (3 rows) x hc (64)
    const int offA2 = hh ? D::T_DC : D::T_DOUT, rowsA2 = hh ? 3 : MNE_OUT1, offB2 = hh ? D::T_HC : D::T_H;
    int ccol[TNC];
#pragma unroll
    for (int q = 0; q < TNC; ++q) ccol[q] = D::cin_col(32 * q + col);
    for (int r = r0; r < r1; ++r) {
        const int n = ray_rows(a, r);
        const float* base = a.tape + (size_t)r * a.S * D::ROW;
        for (int t = 0; t < n; t += 2 * KS) {
            if constexpr (PART == 0) {
                float adh[KS], bx[KS][4];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int tt = t + 2 * ks + kk;
                    const bool ok = tt < n;
                    const float* row = base + (size_t)(ok ? tt : 0) * D::ROW;
                    adh[ks] = ok ? row[D::T_DH + 32 * hh + col] : 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) bx[ks][q] = (ok && 32 * q + col < MNE_IN1) ? row[D::T_X + 32 * q + col] : 0.f;
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(adh[ks], bx[ks][q], acc[q], 0, 0, 0);
            } else {
                float adc[KS], a2[KS], bc[KS][TNC], b2[KS][2];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int tt = t + 2 * ks + kk;
                    const bool ok = tt < n;
                    const float* row = base + (size_t)(ok ? tt : 0) * D::ROW;
                    adc[ks] = ok ? row[D::T_DHC + 32 * hh + col] : 0.f;
                    a2[ks] = (ok && col < rowsA2) ? row[offA2 + col] : 0.f;
#pragma unroll
                    for (int q = 0; q < TNC; ++q) bc[ks][q] = ok ? row[ccol[q]] : 0.f;
#pragma unroll
                    for (int q = 0; q < 2; ++q) b2[ks][q] = ok ? row[offB2 + 32 * q + col] : 0.f;
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int q = 0; q < TNC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(adc[ks], bc[ks][q], acc[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[TNC + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[ks], b2[ks][q], acc[TNC + q], 0, 0, 0);
                }
            }
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
    float* out = a.partials + (size_t)slot * D::NPARAM;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int lo = (e & 3) + 8 * (e >> 2) + 4 * kk;          // row inside the 32-row tile
        const int o = 32 * hh + lo;                              // hidden unit
        if constexpr (PART == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (32 * q + col < MNE_IN1) out[D::P_SDF0 + o * MNE_IN1 + 32 * q + col] = acc[q][e];
        } else {
#pragma unroll
            for (int q = 0; q < TNC; ++q) {
                const int i = 32 * q + col;                      // element of the colour-net input
                if (i != D::CINB) out[D::P_COL0 + o * D::CIN + (i > D::CINB ? i - 1 : i)] = acc[q][e];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (hh == 0) { if (lo < MNE_OUT1) out[D::P_SDF1 + lo * 64 + 32 * q + col] = acc[TNC + q][e]; }
                else if (lo < 3) out[D::P_COL1 + lo * 64 + 32 * q + col] = acc[TNC + q][e];
            }
        }
    }
}

template <bool CP>
__global__ __launch_bounds__(256, 2) void wgrad_fused64_kernel(WgradArgs a) {
    __shared__ float red[2][64 * 16];                            // one 32x32 tile of waves 2 and 3
    if (blockIdx.x & 1) wgrad_fused64_part<CP, 1>(a, red);
    else wgrad_fused64_part<CP, 0>(a, red);
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

// scalar cross-check of the MFMA path (impl = 1): one thread per parameter, rows in tape order
template <int HID, int HIDC, bool CP, int G>
__global__ __launch_bounds__(256) void wgrad_scalar_kernel(WgradArgs a) {
    typedef Gemm<HID, HIDC, CP, G> M;
    typedef DecDims<HID, HIDC, CP> D;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M::OUT * M::LD) return;
    const int o = e / M::LD, pc = e % M::LD;
    int c = pc;                                                   // input element with parameter column pc
    if (G == 2 && pc >= D::CINB) c = pc + 1;
    const int tc = M::bcol(c);
    float s = 0.0f;
    for (int r = 0; r < a.R; ++r) {
        const int n = ray_rows(a, r);
        const float* base = a.tape + (size_t)r * a.S * D::ROW;
        for (int t = 0; t < n; ++t) s = fmaf(base[(size_t)t * D::ROW + M::OFFA + o], base[(size_t)t * D::ROW + tc], s);
    }
    a.grad_out[M::POFF + e] = s;
}

#define MNE_WGRAD_BLOCKS 256      // x 4 waves = 1024 partial results

int mne_wgrad_waves(void) { return MNE_WGRAD_BLOCKS * 4; }

// partial results the fused pass leaves for n_rays rays
static int fused_partials(int hid, int n_rays) {
    const int ranges_per_block = (hid == 32) ? 4 : 2;
    const int blocks = (n_rays + ranges_per_block - 1) / ranges_per_block;
    return blocks < 1 ? 1 : (blocks > WG_FUSED_BLOCKS ? WG_FUSED_BLOCKS : blocks);
}
int mne_wgrad_partial_count(const mne_scene_t& sc, int n_rays) { return fused_partials(sc.hidden, n_rays); }

template <int HID, int HIDC, bool CP>
static int launch_wgrad(WgradArgs a, int impl, hipStream_t st) {
    typedef DecDims<HID, HIDC, CP> D;
    if (impl == 1) {
        MNE_LAUNCH((wgrad_scalar_kernel<HID, HIDC, CP, 0>), (HID * MNE_IN1 + 255) / 256, 256, 0, st, a);
        MNE_LAUNCH((wgrad_scalar_kernel<HID, HIDC, CP, 1>), (MNE_OUT1 * HID + 255) / 256, 256, 0, st, a);
        MNE_LAUNCH((wgrad_scalar_kernel<HID, HIDC, CP, 2>), (HIDC * D::CIN + 255) / 256, 256, 0, st, a);
        MNE_LAUNCH((wgrad_scalar_kernel<HID, HIDC, CP, 3>), (3 * HIDC + 255) / 256, 256, 0, st, a);
        return 0;
    }
    // one partial per WORKGROUP (fused kernels) or per wave (impl 2); never more waves than rays
    if (impl == 0 || impl == 3) {          // 3: without the reduction (the caller continues with mne_decoder_update)
        const int blocks = fused_partials(HID, a.R);
        a.n_waves = blocks;
        if constexpr (HID == 32 && HIDC == 32) MNE_LAUNCH((wgrad_fused_kernel<HID, HIDC, CP>), 2 * blocks, 256, 0, st, a);   // two parts per slot
        else MNE_LAUNCH((wgrad_fused64_kernel<CP>), 2 * blocks, 256, 0, st, a);                                             // likewise
        if (impl == 0) MNE_LAUNCH(wgrad_reduce_kernel, (D::NPARAM + 31) / 32, 256, 0, st, a, D::NPARAM);
        return 0;
    }
    int blocks = (a.R + 3) / 4;
    blocks = blocks < 1 ? 1 : (blocks > MNE_WGRAD_BLOCKS ? MNE_WGRAD_BLOCKS : blocks);
    a.n_waves = blocks * 4;
    MNE_LAUNCH((wgrad_mfma_kernel<HID, HIDC, CP, 0>), blocks, 256, 0, st, a);
    MNE_LAUNCH((wgrad_mfma_kernel<HID, HIDC, CP, 1>), blocks, 256, 0, st, a);
    MNE_LAUNCH((wgrad_mfma_kernel<HID, HIDC, CP, 2>), blocks, 256, 0, st, a);
    MNE_LAUNCH((wgrad_mfma_kernel<HID, HIDC, CP, 3>), blocks, 256, 0, st, a);
    MNE_LAUNCH(wgrad_reduce_kernel, (D::NPARAM + 31) / 32, 256, 0, st, a, D::NPARAM);
    return 0;
}

int mne_launch_wgrad(const mne_scene_t& sc, WgradArgs a, int impl, hipStream_t st) {
    const bool cp = sc.n_sets == 2;
    if (sc.hidden == 32 && sc.hidden_color == 32) return cp ? launch_wgrad<32, 32, true>(a, impl, st) : launch_wgrad<32, 32, false>(a, impl, st);
    if (sc.hidden == 64 && sc.hidden_color == 64) return cp ? launch_wgrad<64, 64, true>(a, impl, st) : launch_wgrad<64, 64, false>(a, impl, st);
    return -2;
}
