// mlp_mfma.h -- the tiny decoder MLPs on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
//
// Formulation: every layer is computed TRANSPOSED,  Y^T[out][point] = W[out][in] . X^T[in][point],
// so the point index stays lane-local across the whole chain (the CDNA analogue of the swapped
// QK^T trick).  A 32-point tile is owned by one wave with TWO LANES PER POINT: lane l and l+32 both
// belong to point (l & 31); for the MFMA B operand (B[k = l>>5][col = l&31]) lane l<32 supplies
// input channel c_lo(s) of its point at k-step s and lane l+32 supplies channel c_hi(s).  The
// reduction order over input channels is free, so channels are paired the way the data is laid out:
//   plane features : lanes<32 hold the coarse level (channels 0..31), lanes>=32 the fine level
//   OneBlob        : lanes<32 hold pos[0..23], lanes>=32 pos[24..47]
//   hidden vectors : the MFMA C/D layout itself -- lane l holds rows (r&3)+8(r>>2)+4(l>>5) of its
//                    point for r = 0..15, i.e. exactly a (lo, hi) pairing with k-step = register r,
//                    so the next layer consumes the accumulator registers directly as B operands.
// The A operand (A[row = l&31][k = l>>5]) is a weight; all weights are pre-permuted once per call by
// pack_decoder_kernel into "A tables": 64 floats per k-step in lane order, read with one coalesced
// 256-B load per MFMA (L1/L2 resident, shared by every wave).
//
// Reference semantics: model/decoder.py:143-175 (ColorSDFNet_v2) / :110-141 (ColorSDFNet), bias-free
// Linear -> ReLU -> Linear for both nets.
#pragma once
#include "mne_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MNE_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// f32x16 accumulator tile (rows = mfma_row(e, hf) of a 32-row tile, column = this lane's point) -> columns
// [col0, col0 + 32) of the point's LDS row; nreg = 8 writes only the 16 valid rows of a 16-row result
__device__ __forceinline__ void acc_to_row(float* prow, int col0, const f32x16& v, int hf, int nreg = 16) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (4 * q < nreg)
            *(float4*)(prow + col0 + 8 * q + 4 * hf) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// A-table access: row `step` of the packed tables, this lane's element.
//   GTAB = false: `atab` is an LDS (or, in small forward-only kernels, a plain global) pointer: one ds_read / global_load
//                 with an immediate offset per MFMA operand.
//   GTAB = true : the tables of the largest decoder (2x64 + colour planes, 124 KiB) do not fit beside the per-wave LDS of
//                 the decode / autograd-backward kernels and are read through L2 with BUFFER loads: one shared 128-bit
//                 resource in SGPRs, one VGPR (lane * 4), the row offset as scalar offset.  With flat global loads the
//                 13-bit immediate reaches 16 rows, so the compiler kept ~20 64-bit row base addresses live across the
//                 MFMA chains and spilled 2-6 KiB per lane (profiles/r02 spill table).
template <bool GTAB, int BIAS = 0>
struct ATabRef {
    const float* A;
    __device__ __forceinline__ ATabRef(const float* atab, int lane) : A(atab + lane - BIAS * 64) {}
    __device__ __forceinline__ float at(int step) const { return A[step * 64]; }
};
#ifndef MNE_HOST_EMU
template <int BIAS>
struct ATabRef<true, BIAS> {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;
    __device__ __forceinline__ ATabRef(const float* atab, int lane)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc((void*)atab, 0, 0x7fffffff, 0x00020000)), voff(lane * 4) {}
    __device__ __forceinline__ float at(int step) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (step - BIAS) * 256, 0));
    }
};
#endif

// row of a 32-row MFMA tile held in accumulator register r by a lane of half h (= lane >> 5)
__host__ __device__ constexpr int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int HID, int HIDC, bool CP>
struct ATab {
    typedef DecDims<HID, HIDC, CP> D;
    static constexpr int NT = HID / 32, NTC = HIDC / 32;
    static constexpr int L1S = 56;                                   // 32 feature + 24 OneBlob steps
    static constexpr int C1S = 24 + (CP ? 32 : 0) + 8;               // OneBlob + (colour features) + geo
    static constexpr int OFF_L1 = 0;                                 // [NT][L1S]
    static constexpr int OFF_L2 = OFF_L1 + NT * L1S;                 // [16*NT]
    static constexpr int OFF_C1 = OFF_L2 + 16 * NT;                  // [NTC][C1S]
    static constexpr int OFF_C2 = OFF_C1 + NTC * C1S;                // [16*NTC]
    static constexpr int FWD_STEPS = OFF_C2 + 16 * NTC;
    static constexpr int OFF_B1 = FWD_STEPS;                         // [NTC][2]     d hc   = V2^T dc
    static constexpr int OFF_B2 = OFF_B1 + NTC * 2;                  // [16*NTC]     d out16 rows m (geo part of V1^T dhc)
    static constexpr int OFF_B2C = OFF_B2 + 16 * NTC;                // CP: [2][16*NTC]  d colour features
    static constexpr int OFF_B3 = OFF_B2C + (CP ? 2 * 16 * NTC : 0); // [NT][8]      d h    = W2^T dout
    static constexpr int OFF_B4 = OFF_B3 + NT * 8;                   // [2][16*NT]   d features = W1^T dh
    static constexpr int TOTAL = OFF_B4 + 2 * 16 * NT;               // everything the mapping iteration needs
    // ray gradients only (R13, pose optimisation): d OneBlob[k] = sum_j W1[j][64+k] dh[j] + sum_j V1[j][k] dhc[j]
    static constexpr int OFF_P1 = TOTAL;                             // [2][16*NT]   sdf-net part, rows = pos channel
    static constexpr int OFF_P2 = OFF_P1 + 2 * 16 * NT;              // [2][16*NTC]  colour-net part
    static constexpr int TOTAL_RAYGRAD = OFF_P2 + 2 * 16 * NTC;
};

// hidden-unit index consumed at k-step s (0..HID/2-1) by half h when the B operand is an accumulator
__host__ __device__ constexpr int hid_of_step(int s, int h) { return 32 * (s >> 4) + mfma_row(s & 15, h); }

// One element of the A tables: step in [0, TOTAL), lane l in [0, 64).
template <int HID, int HIDC, bool CP>
__device__ inline float atab_value(const mne_scene_t& sc, int step, int l) {
    typedef ATab<HID, HIDC, CP> T;
    typedef DecDims<HID, HIDC, CP> D;
    const int i = l & 31, h = l >> 5;
    const float* W1 = sc.w_sdf0;   // [HID][112]
    const float* W2 = sc.w_sdf1;   // [16][HID]
    const float* V1 = sc.w_col0;   // [HIDC][CIN]
    const float* V2 = sc.w_col1;   // [3][HIDC]
    if (step < T::OFF_L2) {                                   // sdf layer 1
        const int t = step / T::L1S, s = step % T::L1S;
        const int ch = s < 32 ? h * 32 + s : MNE_FEAT + h * 24 + (s - 32);
        return W1[(32 * t + i) * MNE_IN1 + ch];
    }
    if (step < T::OFF_C1) {                                   // sdf layer 2 (16 valid rows)
        const int s = step - T::OFF_L2;
        return i < MNE_OUT1 ? W2[i * HID + hid_of_step(s, h)] : 0.0f;
    }
    if (step < T::OFF_C2) {                                   // colour layer 1
        const int t = (step - T::OFF_C1) / T::C1S, s = (step - T::OFF_C1) % T::C1S;
        const int row = 32 * t + i;
        if (s < 24) return V1[row * D::CIN + h * 24 + s];
        if (CP && s < 56) return V1[row * D::CIN + MNE_POS + h * 32 + (s - 24)];
        const int m = mfma_row(s - (CP ? 56 : 24), h);        // row of out16: 0 = sdf (not an input)
        return m >= 1 ? V1[row * D::CIN + D::CINB + m - 1] : 0.0f;
    }
    if (step < T::FWD_STEPS) {                                // colour layer 2 (3 valid rows)
        const int s = step - T::OFF_C2;
        return i < 3 ? V2[i * HIDC + hid_of_step(s, h)] : 0.0f;
    }
    if (step < T::OFF_B2) {                                   // d hc[j] = sum_c V2[c][j] dc[c]
        const int t = (step - T::OFF_B1) / 2, s = (step - T::OFF_B1) % 2;
        const int c = 2 * s + h;
        return c < 3 ? V2[c * HIDC + 32 * t + i] : 0.0f;
    }
    if (step < T::OFF_B2C) {                                  // d out16[m] = sum_j V1[j][CINB+m-1] dhc[j]
        const int s = step - T::OFF_B2;
        return (i >= 1 && i < MNE_OUT1) ? V1[hid_of_step(s, h) * D::CIN + D::CINB + i - 1] : 0.0f;
    }
    if (CP && step < T::OFF_B3) {                             // d colour feature[k] = sum_j V1[j][48+k] dhc[j]
        const int rt = (step - T::OFF_B2C) / (16 * T::NTC), s = (step - T::OFF_B2C) % (16 * T::NTC);
        return V1[hid_of_step(s, h) * D::CIN + MNE_POS + 32 * rt + i];
    }
    if (step < T::OFF_B4) {                                   // d h[j] = sum_m W2[m][j] dout[m]
        const int t = (step - T::OFF_B3) / 8, s = (step - T::OFF_B3) % 8;
        return W2[mfma_row(s, h) * HID + 32 * t + i];
    }
    if (step < T::OFF_P1) {                                   // d feature[k] = sum_j W1[j][k] dh[j]
        const int rt = (step - T::OFF_B4) / (16 * T::NT), s = (step - T::OFF_B4) % (16 * T::NT);
        return W1[hid_of_step(s, h) * MNE_IN1 + 32 * rt + i];
    }
    if (step < T::OFF_P2) {                                   // d pos[k] (sdf net): W1[j][64 + k]
        const int rt = (step - T::OFF_P1) / (16 * T::NT), s = (step - T::OFF_P1) % (16 * T::NT);
        const int k = 32 * rt + i;
        return k < MNE_POS ? W1[hid_of_step(s, h) * MNE_IN1 + MNE_FEAT + k] : 0.0f;
    }
    {                                                         // d pos[k] (colour net): V1[j][k]
        const int rt = (step - T::OFF_P2) / (16 * T::NTC), s = (step - T::OFF_P2) % (16 * T::NTC);
        const int k = 32 * rt + i;
        return k < MNE_POS ? V1[hid_of_step(s, h) * D::CIN + k] : 0.0f;
    }
}

// Register state of one 32-point tile after the forward chain.
template <int HID, int HIDC>
struct MlpState {
    f32x16 h[HID / 32];     // relu(W1 x)           rows = hidden units
    f32x16 out;             // W2 h                 rows 0..15 = (sdf, geo15)
    f32x16 hc[HIDC / 32];   // relu(V1 [pos,(cf),geo])
    f32x16 rgb;             // V2 hc                rows 0..2 = raw rgb
};

// OneBlob of this lane's half of the 48 channels: half 0 -> pos[0..23] = dim0 (16) + dim1 bins 0..7,
// half 1 -> pos[24..47] = dim1 bins 8..15 + dim2 (16).
// LEAN: the middle dimension's eight bins of this half only (oneblob8: 25 instead of 32 cumulative values per lane, same
// bits).  Off for the one kernel that has no register to spare for the extra live range (decode of 2x64 + colour planes).
template <bool LEAN = true>
__device__ __forceinline__ void oneblob_half(const float u[3], int h, float (&pos)[24]) {
    float full[MNE_NB], part[LEAN ? 8 : MNE_NB];
    const float xf = h == 0 ? u[0] : u[2];
    // x in [1/64, 59/64]: for every bin the two periodic wrap terms have |16 t| >= 1.25, where the clamped quartic is exactly
    // 0 and 1 (its value there is 0.023 outside [0, 1]; checked over 2.2 M fp32 values incl. every value next to the bounds)
    const bool in_f = xf >= 0.015625f && xf <= 0.921875f, in_p = u[1] >= 0.015625f && u[1] <= 0.921875f;
    const bool interior = __ballot(!(in_f && in_p)) == 0ull;        // wave-uniform fast path
    oneblob16<LEAN>(xf, full, interior);
    if constexpr (LEAN) oneblob8(u[1], h * 8, part, interior);
    else oneblob16<false>(u[1], part, interior);
    if constexpr (!LEAN) {
#pragma unroll
        for (int idx = 0; idx < 24; ++idx) {
            const float lo = idx < 16 ? full[idx & 15] : part[(idx - 16) & 15];      // half 0: dim0 | dim1[0..7]
            const float hi = idx < 8 ? part[(8 + idx) & 15] : full[(idx - 8) & 15];   // half 1: dim1[8..15] | dim2
            pos[idx] = h == 0 ? lo : hi;
        }
        return;
    }
    // static register indices only (a lane-dependent index would put pos[] in scratch)
#pragma unroll
    for (int idx = 0; idx < 24; ++idx) {
        const float lo = idx < 16 ? full[idx & 15] : part[(idx - 16) & 7];       // half 0: dim0 | dim1[0..7]
        const float hi = idx < 8 ? part[idx & 7] : full[(idx - 8) & 15];          // half 1: dim1[8..15] | dim2
        pos[idx] = h == 0 ? lo : hi;
    }
}

// frow / cfrow: LDS feature rows of this lane's POINT (64 floats each); the lane reads its level half.
// Two halves -- the sdf net (needs frow) and the colour net (needs cfrow, pos and the sdf net's out) -- so that a forward-only
// caller can keep ONE set of feature rows in LDS and gather the colour planes between the two (decode_tile<..., SEQF>).
template <int HID, int HIDC, bool CP, bool GTAB = false>
__device__ __forceinline__ void mlp_forward_sdf(const float* frow, const float (&pos)[24], const float* atab, int lane,
                                                MlpState<HID, HIDC>& S) {
    typedef ATab<HID, HIDC, CP> T;
    const int h = lane >> 5;
    const ATabRef<GTAB> A(atab, lane);
    const float* fh = frow + h * 32;
#pragma unroll
    for (int t = 0; t < T::NT; ++t) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 x = *(const float4*)(fh + 4 * q);
            acc = MNE_MFMA(A.at(T::OFF_L1 + t * T::L1S + 4 * q + 0), x.x, acc);
            acc = MNE_MFMA(A.at(T::OFF_L1 + t * T::L1S + 4 * q + 1), x.y, acc);
            acc = MNE_MFMA(A.at(T::OFF_L1 + t * T::L1S + 4 * q + 2), x.z, acc);
            acc = MNE_MFMA(A.at(T::OFF_L1 + t * T::L1S + 4 * q + 3), x.w, acc);
        }
#pragma unroll
        for (int s = 0; s < 24; ++s) acc = MNE_MFMA(A.at(T::OFF_L1 + t * T::L1S + 32 + s), pos[s], acc);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = fmaxf(acc[e], 0.0f);
        S.h[t] = acc;
    }
    {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 16 * T::NT; ++s) acc = MNE_MFMA(A.at(T::OFF_L2 + s), S.h[s >> 4][s & 15], acc);
        S.out = acc;
    }
}

template <int HID, int HIDC, bool CP, bool GTAB = false>
__device__ __forceinline__ void mlp_forward_color(const float* cfrow, const float (&pos)[24], const float* atab, int lane,
                                                  MlpState<HID, HIDC>& S) {
    typedef ATab<HID, HIDC, CP> T;
    const int h = lane >> 5;
    const ATabRef<GTAB> A(atab, lane);
    const float* ch = cfrow + h * 32;
#pragma unroll
    for (int t = 0; t < T::NTC; ++t) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        const int base = T::OFF_C1 + t * T::C1S;
#pragma unroll
        for (int s = 0; s < 24; ++s) acc = MNE_MFMA(A.at(base + s), pos[s], acc);
        if (CP) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 x = *(const float4*)(ch + 4 * q);
                acc = MNE_MFMA(A.at(base + 24 + 4 * q + 0), x.x, acc);
                acc = MNE_MFMA(A.at(base + 24 + 4 * q + 1), x.y, acc);
                acc = MNE_MFMA(A.at(base + 24 + 4 * q + 2), x.z, acc);
                acc = MNE_MFMA(A.at(base + 24 + 4 * q + 3), x.w, acc);
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = MNE_MFMA(A.at(base + 24 + (CP ? 32 : 0) + s), S.out[s], acc);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = fmaxf(acc[e], 0.0f);
        S.hc[t] = acc;
    }
    {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 16 * T::NTC; ++s) acc = MNE_MFMA(A.at(T::OFF_C2 + s), S.hc[s >> 4][s & 15], acc);
        S.rgb = acc;
    }
}

template <int HID, int HIDC, bool CP, bool GTAB = false>
__device__ __forceinline__ void mlp_forward_mfma(const float* frow, const float* cfrow, const float (&pos)[24],
                                                 const float* atab, int lane, MlpState<HID, HIDC>& S) {
    mlp_forward_sdf<HID, HIDC, CP, GTAB>(frow, pos, atab, lane, S);
    mlp_forward_color<HID, HIDC, CP, GTAB>(cfrow, pos, atab, lane, S);
}

// ReLU pattern of this lane's hidden rows: bit (16 t + e) = (S.h[t][e] > 0), same for hc.  Saved by the forward
// (8 bytes per lane) so that the backward chain needs no activations.
template <int HID, int HIDC>
__device__ __forceinline__ void relu_masks(const MlpState<HID, HIDC>& S, unsigned& mh, unsigned& mhc) {
    mh = 0u; mhc = 0u;
#pragma unroll
    for (int t = 0; t < HID / 32; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) mh |= (S.h[t][e] > 0.0f ? 1u : 0u) << (16 * t + e);
#pragma unroll
    for (int t = 0; t < HIDC / 32; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) mhc |= (S.hc[t][e] > 0.0f ? 1u : 0u) << (16 * t + e);
}

// Backward data path.  ds/dc: d(total)/d(sdf), d(total)/d(raw rgb) of this lane's point (identical on
// both lanes of the pair); mh / mhc: relu_masks of the forward.  Outputs: dh, dout, dhc (tape) and d(feature)
// rows written to LDS (dfrow / dcfrow = the point's rows; each lane writes the rows it holds).
// BIAS (backward functions): `atab` points at table step BIAS (the training kernel stages only the backward steps in LDS;
// a pointer biased at run time instead pushed the ds_read offsets of the 2x64 tables past the 64 KiB immediate range and
// cost an address register per MFMA operand).
template <int HID, int HIDC, bool CP, int BIAS = 0, bool GTAB = false>
__device__ __forceinline__ void mlp_backward_color(unsigned mhc, float ds, const float (&dc)[3], const float* atab, int lane,
                                                   f32x16& dout, f32x16 (&dhc)[HIDC / 32], float* dcfrow) {
    typedef ATab<HID, HIDC, CP> T;
    const int h = lane >> 5;
    const ATabRef<GTAB, BIAS> A(atab, lane);
#pragma unroll
    for (int t = 0; t < T::NTC; ++t) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        acc = MNE_MFMA(A.at(T::OFF_B1 + 2 * t + 0), h ? dc[1] : dc[0], acc);
        acc = MNE_MFMA(A.at(T::OFF_B1 + 2 * t + 1), h ? 0.0f : dc[2], acc);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = ((mhc >> (16 * t + e)) & 1u) ? acc[e] : 0.0f;
        dhc[t] = acc;
    }
    {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 16 * T::NTC; ++s) acc = MNE_MFMA(A.at(T::OFF_B2 + s), dhc[s >> 4][s & 15], acc);
        if (h == 0) acc[0] = ds;            // row m = 0 is the sdf output
        dout = acc;
    }
    if (CP) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
            for (int s = 0; s < 16 * T::NTC; ++s)
                acc = MNE_MFMA(A.at(T::OFF_B2C + rt * 16 * T::NTC + s), dhc[s >> 4][s & 15], acc);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(float4*)(dcfrow + 32 * rt + 8 * q + 4 * h) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
    }
}

template <int HID, int HIDC, bool CP, int BIAS = 0, bool GTAB = false>
__device__ __forceinline__ void mlp_backward_sdf(unsigned mh, const float* atab, int lane, f32x16 (&dh)[HID / 32],
                                                 const f32x16& dout, float* dfrow) {
    typedef ATab<HID, HIDC, CP> T;
    const int h = lane >> 5;
    const ATabRef<GTAB, BIAS> A(atab, lane);
#pragma unroll
    for (int t = 0; t < T::NT; ++t) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = MNE_MFMA(A.at(T::OFF_B3 + 8 * t + s), dout[s], acc);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = ((mh >> (16 * t + e)) & 1u) ? acc[e] : 0.0f;
        dh[t] = acc;
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 16 * T::NT; ++s)
            acc = MNE_MFMA(A.at(T::OFF_B4 + rt * 16 * T::NT + s), dh[s >> 4][s & 15], acc);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(float4*)(dfrow + 32 * rt + 8 * q + 4 * h) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
}

// the whole chain: colour net (writes the colour planes' d(feature) rows), then sdf net (geometry planes' rows)
template <int HID, int HIDC, bool CP, int BIAS = 0, bool GTAB = false>
__device__ __forceinline__ void mlp_backward_mfma(unsigned mh, unsigned mhc, float ds, const float (&dc)[3],
                                                  const float* atab, int lane, f32x16 (&dh)[HID / 32],
                                                  f32x16& dout, f32x16 (&dhc)[HIDC / 32], float* dfrow, float* dcfrow) {
    mlp_backward_color<HID, HIDC, CP, BIAS, GTAB>(mhc, ds, dc, atab, lane, dout, dhc, dcfrow);
    mlp_backward_sdf<HID, HIDC, CP, BIAS, GTAB>(mh, atab, lane, dh, dout, dfrow);
}

// d(total)/d(OneBlob channel) rows of this lane's point -> LDS row dprow[0..63] (48 used): both nets'
// first layers, chained into one accumulator per 32-row tile.  Ray-gradient variant only.
template <int HID, int HIDC, bool CP, int BIAS = 0, bool GTAB = false>
__device__ __forceinline__ void mlp_backward_dpos(const f32x16 (&dh)[HID / 32], const f32x16 (&dhc)[HIDC / 32],
                                                  const float* atab, int lane, float* dprow) {
    typedef ATab<HID, HIDC, CP> T;
    const int h = lane >> 5;
    const ATabRef<GTAB, BIAS> A(atab, lane);                   // atab starts at table step BIAS (as in the other backward chains)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 16 * T::NT; ++s)
            acc = MNE_MFMA(A.at(T::OFF_P1 + rt * 16 * T::NT + s), dh[s >> 4][s & 15], acc);
#pragma unroll
        for (int s = 0; s < 16 * T::NTC; ++s)
            acc = MNE_MFMA(A.at(T::OFF_P2 + rt * 16 * T::NTC + s), dhc[s >> 4][s & 15], acc);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(float4*)(dprow + 32 * rt + 8 * q + 4 * h) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
}

// d(OneBlob bins of one dim)/dx for this lane's half, laid out like oneblob_half's pos[24]:
// quartic kernel derivative 15/16 (1-u^2)^2 * 16 on |u| < 1 (u = 16 t), with the periodic wrap terms.
__device__ __forceinline__ float quartic_pdf(float t) {
    const float u = t * 16.0f;
    const float w = 1.0f - u * u;
    return (u > -1.0f && u < 1.0f) ? 0.9375f * w * w * 16.0f : 0.0f;
}

__device__ __forceinline__ void oneblob16_dx(float x, float* dout /*16*/) {
    float c[MNE_NB];
#pragma unroll
    for (int b = 0; b < MNE_NB; ++b) {
        const float t = (float)b * 0.0625f - x;
        c[b] = -((quartic_pdf(t) + quartic_pdf(t - 1.0f)) + quartic_pdf(t + 1.0f));    // d cdf(l_b - x) / dx
    }
#pragma unroll
    for (int b = 0; b < MNE_NB - 1; ++b) dout[b] = c[b + 1] - c[b];
    dout[MNE_NB - 1] = c[0] - c[MNE_NB - 1];
}

// sum_k dpos[k] * d pos[k] / d u[dim] over this lane's 24 channels -> contributions to du[0..2]
__device__ __forceinline__ void oneblob_half_backward(const float u[3], int h, const float* dprow, float (&du)[3]) {
    float dfull[MNE_NB], dpart[MNE_NB];
    oneblob16_dx(h == 0 ? u[0] : u[2], dfull);
    oneblob16_dx(u[1], dpart);
    float s_full = 0.0f, s_part = 0.0f;
#pragma unroll
    for (int idx = 0; idx < 24; ++idx) {
        const float g = dprow[h * 24 + idx];
        const float lo_f = idx < 16 ? dfull[idx & 15] : 0.0f, lo_p = idx < 16 ? 0.0f : dpart[(idx - 16) & 15];
        const float hi_p = idx < 8 ? dpart[(8 + idx) & 15] : 0.0f, hi_f = idx < 8 ? 0.0f : dfull[(idx - 8) & 15];
        s_full += g * (h == 0 ? lo_f : hi_f);
        s_part += g * (h == 0 ? lo_p : hi_p);
    }
    du[0] = h == 0 ? s_full : 0.0f;
    du[1] = s_part;
    du[2] = h == 0 ? 0.0f : s_full;
}
