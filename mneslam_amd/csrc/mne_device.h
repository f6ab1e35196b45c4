// mne_device.h -- device helpers shared by the mapping-path kernels (gfx950 / wave64).
//
// Numerics policy: the translation units are compiled with -ffp-contract=off, so the coordinate
// pipeline (points, normalisation, bilinear indices/weights, z jitter) rounds exactly like the
// reference's eager fp32 torch ops; FMAs appear only where written as fmaf() (feature
// accumulation, MLP dot products).
#pragma once
#include "mne_platform.h"
#include "mneslam_hip.h"
#include "mne_launch.h"

#define MNE_WAVE 64
#define MNE_C 32      // channels per plane (model.c_dim)
#define MNE_NB 16     // OneBlob bins per dim (pos.n_bins)
#define MNE_POS 48    // 3 * MNE_NB
#define MNE_FEAT 64   // 2 levels * MNE_C
#define MNE_GEO 15    // decoder.geo_feat_dim
#define MNE_OUT1 16   // 1 sdf + 15 geo
#define MNE_FS 68     // LDS feature-row stride in floats (64 + 4: 16-B aligned, b128 reads conflict-free)
#define MNE_IN1 112   // MNE_FEAT + MNE_POS

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- Philox4x32-10 (counter-based RNG for the on-device jitter) -------------------------------
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t offset, uint64_t elem) {
    uint64_t ctr = offset + (elem >> 2);
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0u, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    uint32_t w = (elem & 3) == 0 ? c0 : (elem & 3) == 1 ? c1 : (elem & 3) == 2 ? c2 : c3;
    return (float)(w >> 8) * (1.0f / 16777216.0f);      // [0,1), 24 bits like torch.rand
}

// ---- Adam bias corrections from the device clock (graph replay): same double arithmetic as the host path --------
__device__ __forceinline__ void clock_bias(const Clock& clk, double lr, int step, float& step_size, float& bc2_sqrt) {
    int t = step + *clk.step_offset;
    t = t < 1 ? 1 : (t > clk.n_table ? clk.n_table : t);
    step_size = (float)(lr / clk.bias_table[2 * (t - 1)]);
    bc2_sqrt = (float)sqrt(clk.bias_table[2 * (t - 1) + 1]);
}

// ---- one Adam element (torch.optim.Adam single-tensor arithmetic; constants prepared on the host in double) ------
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const PlaneOpt& o) {
    if (o.wd != 0.0f) g = g + o.wd * p;
    m = m + (g - m) * o.omb1;
    v = v * o.b2 + o.omb2 * (g * g);
    const float denom = sqrtf(v) / o.bc2_sqrt + o.eps;
    p = p - o.step_size * (m / denom);
}

// ---- coordinates --------------------------------------------------------------------------------
// Plane lookup uses the EXTENDED bound -> [-1,1] (model/utils.py:38-40); OneBlob uses the RAW
// bounding box -> [0,1], in fp64 when the box is float64 (model/scene_rep.py:292).
__device__ __forceinline__ float unit_coord(float p, double bb_lo, double bb_hi, bool is_f64) {
    if (is_f64) return (float)(((double)p - bb_lo) / (bb_hi - bb_lo));
    const float lo = (float)bb_lo, hi = (float)bb_hi;
    return (p - lo) / (hi - lo);
}

__device__ __forceinline__ void point_coords(const mne_scene_t& sc, const float p[3], float pn[3], float u[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        pn[k] = ((p[k] - sc.bound_lo[k]) / (sc.bound_hi[k] - sc.bound_lo[k])) * 2.0f - 1.0f;
        u[k] = unit_coord(p[k], sc.bb_lo[k], sc.bb_hi[k], sc.bb_is_f64 != 0);
    }
}

// ---- bilinear corner set (ATen grid_sampler_2d: bilinear, align_corners=True, border) ----------
struct Bilin {
    int o00, o01, o10, o11;       // float offsets of the corner rows (nw, ne, sw, se), clamped in-range
    float w00, w01, w10, w11;     // weights; 0 for corners outside the plane (skipped by ATen)
    int ix0, iy0;                 // the integer NW corner ("bit-exact indices" of this path)
};

__device__ __forceinline__ float unnormalize_clip(float g, int size) {
    float v = ((g + 1.0f) / 2.0f) * (float)(size - 1);
    return fminf((float)(size - 1), fmaxf(v, 0.0f));
}

__device__ __forceinline__ void bilin_setup(float gx, float gy, int H, int W, Bilin& b) {
    float fx = unnormalize_clip(gx, W), fy = unnormalize_clip(gy, H);
    float x0 = floorf(fx), y0 = floorf(fy);
    float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    float nw = (x1 - fx) * (y1 - fy), ne = (fx - x0) * (y1 - fy);
    float sw = (x1 - fx) * (fy - y0), se = (fx - x0) * (fy - y0);
    int ix0 = (int)x0, iy0 = (int)y0;
    bool xin = ix0 + 1 < W, yin = iy0 + 1 < H;
    int ix1 = xin ? ix0 + 1 : ix0, iy1 = yin ? iy0 + 1 : iy0;
    b.ix0 = ix0; b.iy0 = iy0;
    b.o00 = (iy0 * W + ix0) * MNE_C; b.o01 = (iy0 * W + ix1) * MNE_C;
    b.o10 = (iy1 * W + ix0) * MNE_C; b.o11 = (iy1 * W + ix1) * MNE_C;
    b.w00 = nw; b.w01 = xin ? ne : 0.0f; b.w10 = yin ? sw : 0.0f; b.w11 = (xin && yin) ? se : 0.0f;
}

__device__ __forceinline__ void orient_coords(int orient, float px, float py, float pz, float& gx, float& gy) {
    // xy -> (x,y); xz -> (x,z); yz -> (y,z): first coordinate indexes W (model/scene_rep.py:43-47)
    gx = orient == MNE_YZ ? py : px;
    gy = orient == MNE_XY ? py : pz;
}

// ---- OneBlob (oracle/oneblob.py is the spec; tinycudann's published quartic-kernel OneBlob) ----
__device__ __forceinline__ float quartic_cdf(float t) {
    float u = t * 16.0f;
    float u2 = u * u;
    float u4 = u2 * u2;
    float poly = (0.9375f * u) * ((1.0f - 0.6666666666666666f * u2) + 0.2f * u4) + 0.5f;
    return fminf(1.0f, fmaxf(poly, 0.0f));
}

// Two cumulative values at once (bins b, b + 1): the same fp32 operations on the same inputs as quartic_cdf, element by element
// (-ffp-contract=off: no fused multiply-add either way), written on float2 so that the multiplies and adds issue as
// v_pk_mul_f32 / v_pk_add_f32 -- two values per VALU instruction.  Round 6: the OneBlob was ~400 scalar VALU instructions per lane
// and 32-sample tile (25 cumulative values x 15 operations), a fifth of what a decode wave issues.
typedef float mne_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mne_f2 quartic_cdf2(mne_f2 t) {
    const mne_f2 u = t * 16.0f;
    const mne_f2 u2 = u * u;
    const mne_f2 u4 = u2 * u2;
    const mne_f2 poly = (0.9375f * u) * ((1.0f - 0.6666666666666666f * u2) + 0.2f * u4) + 0.5f;
    mne_f2 r;
    r.x = fminf(1.0f, fmaxf(poly.x, 0.0f));
    r.y = fminf(1.0f, fmaxf(poly.y, 0.0f));
    return r;
}
__device__ __forceinline__ mne_f2 oneblob_cum2(float b0, float b1, float x, bool interior) {
    mne_f2 t;
    t.x = b0 * 0.0625f - x;
    t.y = b1 * 0.0625f - x;
    if (interior) return (quartic_cdf2(t) + 0.0f) + 1.0f;
    return (quartic_cdf2(t) + quartic_cdf2(t - 1.0f)) + quartic_cdf2(t + 1.0f);
}
__device__ __forceinline__ float oneblob_cum1(float b, float x, bool interior) {
    const float t = b * 0.0625f - x;
    return interior ? (quartic_cdf(t) + 0.0f) + 1.0f : (quartic_cdf(t) + quartic_cdf(t - 1.0f)) + quartic_cdf(t + 1.0f);
}

// `interior` (wave-uniform): every lane's x lies in [1/64, 59/64], where the two periodic wrap terms are
// exactly 0 and 1 (|u| >= 1.25: the clamp saturates with a margin of 0.023), so only the central kernel is evaluated.
// PACKED = false: the scalar form, for the one kernel without a register to spare for even-aligned pairs (decode of 2x64 + colour planes).
template <bool PACKED = true>
__device__ __forceinline__ void oneblob16(float x, float* out /*16*/, bool interior = false) {
    float c[MNE_NB];
    if constexpr (PACKED) {
#pragma unroll
        for (int b = 0; b < MNE_NB; b += 2) {
            const mne_f2 v = oneblob_cum2((float)b, (float)(b + 1), x, interior);
            c[b] = v.x; c[b + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int b = 0; b < MNE_NB; ++b) c[b] = oneblob_cum1((float)b, x, interior);
    }
#pragma unroll
    for (int b = 0; b < MNE_NB - 1; ++b) out[b] = c[b + 1] - c[b];
    out[MNE_NB - 1] = (c[0] + 1.0f) - c[MNE_NB - 1];
}

// Eight consecutive bins [base, base + 8) of the same encoding, base = 0 or 8 (lane-dependent): the nine cumulative values
// they need instead of all sixteen -- same expressions on the same inputs, so the values are bit-identical to
// oneblob16's (the half-wave layout of the MLP kernels needs only half of the middle dimension's bins per lane).
__device__ __forceinline__ void oneblob8(float x, int base, float* out /*8*/, bool interior = false) {
    float c[9];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {                          // base + j + 1 <= 15: no wrap inside a pair
        const mne_f2 v = oneblob_cum2((float)(base + j), (float)(base + j + 1), x, interior);
        c[j] = v.x; c[j + 1] = v.y;
    }
    c[8] = oneblob_cum1((float)((base + 8) & (MNE_NB - 1)), x, interior);     // base = 8: the ninth value is c[0] (periodic wrap, below)
#pragma unroll
    for (int j = 0; j < 7; ++j) out[j] = c[j + 1] - c[j];
    out[7] = base ? (c[8] + 1.0f) - c[7] : c[8] - c[7];        // bin 15 wraps around: (c[0] + 1) - c[15]
}

// ---- gather: tri-plane features of the 64 staged points -> LDS rows ---------------------------
// Lane layout: 8 lanes x float4 cover one 128-B corner row, 8 points per pass (coalesced rows).
// pn: LDS [NPTS][4] normalised points; feat: LDS [NSETS][NPTS][MNE_FS].
// The 8 lanes of a point share the bilinear set-up work: lane cg computes the corner set of ONE of the six
// (level, orientation) planes of the set (lanes 6, 7 recompute planes 0, 1), and the eight values of plane k
// are then broadcast inside the 8-lane group (ds_swizzle, no LDS memory).  Same fp32 operations on the same
// inputs as a per-lane set-up, so indices and weights are bit-identical; 6x fewer set-up instructions.
// Load scheduling is explicit: per pass and plane set, (1) the 48 broadcasts, (2) all corner-row loads of a group of
// planes back to back (MNE_GATHER_INFLIGHT = 12: one level, or 24: both levels -> that many float4 in flight per lane),
// (3) the blends.  Left to the compiler, every load queued behind its own ds_swizzle and the kernel ran 40 % slower
// (profiles/r02_order_ab.txt); the scheduling barriers pin the three phases.
#ifndef MNE_GATHER_INFLIGHT
#define MNE_GATHER_INFLIGHT 12
#endif
#ifndef MNE_INLINE_GATHER_NLV
#define MNE_INLINE_GATHER_NLV 1     // levels requested together by the gathers INSIDE the tile kernels (gather_chunk)
#endif
// Tri-plane features of ONE point for the 8 lanes that share it (cg = lane & 7: float4 chunk of the 128-B rows); the
// blended rows go to out + set * set_stride + level * 32 + cg * 4 (LDS row or tape row).
// Four channels of one corner row at element offset e: planes are stored in fp32, or -- mne_scene_t.plane_f16, uniform over
// the launch -- in IEEE half precision (a corner row is then 64 bytes: 8 lanes x 8 B); arithmetic is fp32 either way.
__device__ __forceinline__ float4 half4_to_float4(uint2 u) {
    union { uint2 u; _Float16 h[4]; } r;
    r.u = u;
    return make_float4((float)r.h[0], (float)r.h[1], (float)r.h[2], (float)r.h[3]);
}
__device__ __forceinline__ uint2 float4_to_half4(float4 v) {                  // round to nearest even
    union { uint2 u; _Float16 h[4]; } r;
    r.h[0] = (_Float16)v.x; r.h[1] = (_Float16)v.y; r.h[2] = (_Float16)v.z; r.h[3] = (_Float16)v.w;
    return r.u;
}
// MNE_PLANE_BUFFER_LOADS: the corner rows through BUFFER loads -- the plane's base in a (wave-uniform) resource descriptor, the
// element offset as one 32-bit VGPR: one VALU instruction of address arithmetic per load instead of three (add, sign extension,
// 64-bit shift-add).  Same bytes loaded.  The frame kernels are bound by what their waves ISSUE: render_img 36.2 -> 34.7 ms
// per frame pair; the mapping iteration +0.3 % (office0) / +0.8 % (ScanNet) / 0 (INS Indoor), profiles/r05_buffer_loads.txt.
// Opt-in per call site (BUF): gather_kernel and the frame kernels; the tile kernels of the largest decoder have no register to spare
// for the descriptors (decode_kernel<64, 64, colour planes> spilled 12 B per lane with it).
#ifndef MNE_PLANE_BUFFER_LOADS
#define MNE_PLANE_BUFFER_LOADS 1
#endif
template <bool F16, bool BUF = false>
__device__ __forceinline__ float4 plane_row4(const mne_plane_t& pl, int e) {
#if MNE_PLANE_BUFFER_LOADS && !defined(MNE_HOST_EMU)
  if constexpr (BUF) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pl.data, 0, 0x7fffffff, 0x00020000);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    if (F16) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, e * 2, 0, 0);
        return half4_to_float4(make_uint2(v.x, v.y));
    }
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, e * 4, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  }
#endif
    if (F16) return half4_to_float4(*(const uint2*)((const _Float16*)pl.data + e));
    return *(const float4*)((const float*)pl.data + e);
}

template <int NSETS, bool F16, int NLV = MNE_GATHER_INFLIGHT / 12, int SET0 = 0, bool BUF = false>       // NLV: levels loaded together (12 corner rows each)
__device__ __forceinline__ void gather_slot_t(const mne_scene_t& sc, float px, float py, float pz, int cg, float* out, int set_stride) {   // plane sets SET0 .. SET0 + NSETS - 1
    const int kmine = cg < 6 ? cg : cg - 6;                        // this lane's plane: k = lvl * 3 + ori
    const int lvl_m = kmine >= 3 ? 1 : 0, ori_m = kmine - 3 * lvl_m;
#pragma unroll
    for (int set = SET0; set < SET0 + NSETS; ++set) {
        int Hm = sc.plane[set][0][0].h, Wm = sc.plane[set][0][0].w;
#pragma unroll
        for (int k = 1; k < 6; ++k) {
            Hm = kmine == k ? sc.plane[set][k % 3][k / 3].h : Hm;
            Wm = kmine == k ? sc.plane[set][k % 3][k / 3].w : Wm;
        }
        float gx, gy;
        orient_coords(ori_m, px, py, pz, gx, gy);
        Bilin bm;
        bilin_setup(gx, gy, Hm, Wm, bm);
#pragma unroll
        for (int l0 = 0; l0 < 2; l0 += NLV) {
            int off[3 * NLV][4];
            float wgt[3 * NLV][4];
#pragma unroll
            for (int j = 0; j < 3 * NLV; ++j) {
                const int k = l0 * 3 + j;                               // k = lvl * 3 + ori
                off[j][0] = mne_bcast8(bm.o00, k); off[j][1] = mne_bcast8(bm.o01, k);
                off[j][2] = mne_bcast8(bm.o10, k); off[j][3] = mne_bcast8(bm.o11, k);
                wgt[j][0] = mne_bcast8(bm.w00, k); wgt[j][1] = mne_bcast8(bm.w01, k);
                wgt[j][2] = mne_bcast8(bm.w10, k); wgt[j][3] = mne_bcast8(bm.w11, k);
            }
            MNE_SCHED_BARRIER();
            float4 v[3 * NLV][4];
#pragma unroll
            for (int j = 0; j < 3 * NLV; ++j) {
                const int k = l0 * 3 + j;
                const mne_plane_t& pl = sc.plane[set][k % 3][k / 3];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[j][q] = plane_row4<F16, BUF>(pl, cg * 4 + off[j][q]);
            }
            MNE_SCHED_BARRIER();
#pragma unroll
            for (int lv = 0; lv < NLV; ++lv) {
                float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int ori = 0; ori < 3; ++ori) {
                    const int j = lv * 3 + ori;
                    float4 acc;
                    acc.x = v[j][0].x * wgt[j][0]; acc.y = v[j][0].y * wgt[j][0]; acc.z = v[j][0].z * wgt[j][0]; acc.w = v[j][0].w * wgt[j][0];
#pragma unroll
                    for (int q = 1; q < 4; ++q) {
                        acc.x = fmaf(v[j][q].x, wgt[j][q], acc.x); acc.y = fmaf(v[j][q].y, wgt[j][q], acc.y);
                        acc.z = fmaf(v[j][q].z, wgt[j][q], acc.z); acc.w = fmaf(v[j][q].w, wgt[j][q], acc.w);
                    }
                    sum.x += acc.x; sum.y += acc.y; sum.z += acc.z; sum.w += acc.w;       // xy + xz + yz
                }
                *(float4*)(out + (set - SET0) * set_stride + (l0 + lv) * MNE_C + cg * 4) = sum;
            }
        }
    }
}

template <int NSETS, int NLV = MNE_GATHER_INFLIGHT / 12, int SET0 = 0, bool BUF = false>
__device__ __forceinline__ void gather_slot(const mne_scene_t& sc, float px, float py, float pz, int cg, float* out, int set_stride) {
    if (sc.plane_f16) gather_slot_t<NSETS, true, NLV, SET0, BUF>(sc, px, py, pz, cg, out, set_stride);       // (uniform over the launch)
    else gather_slot_t<NSETS, false, NLV, SET0, BUF>(sc, px, py, pz, cg, out, set_stride);
}

// (NLV = 2 -- both levels' 24 corner rows requested together -- was measured for the gathers INSIDE the tile kernels, where the
// registers are free: no effect on the deferred decode, the resolver's extension tile or render_img, profiles/r05_inline_gather_levels.txt)
template <int NSETS, int NPTS, int NLV = 1, int SET0 = 0, bool BUF = false>
__device__ __forceinline__ void gather_chunk(const mne_scene_t& sc, const float* pn, float* feat, int lane) {
    const int cg = lane & 7;
#pragma unroll 1
    for (int it = 0; it < NPTS / 8; ++it) {
        const int slot = it * 8 + (lane >> 3);
        gather_slot<NSETS, NLV, SET0, BUF>(sc, pn[slot * 4 + 0], pn[slot * 4 + 1], pn[slot * 4 + 2], cg, feat + slot * MNE_FS, NPTS * MNE_FS);
    }
}

// ---- d(total)/d(normalised point) through the bilinear lookups (ray gradients, R13) -----------------
// Same lane layout as gather_chunk.  For every staged point: sum over planes/levels of
//   dfeat . d feat / d (ix, iy) * (size-1)/2   (ATen grid_sampler_2d_backward's grid gradient: zero
// where the coordinate was clipped, out-of-range corners contribute nothing), reduced over the 8 lanes
// of a row group; result dpn[slot][0..2] = d/d(p_nor x,y,z).
template <int NSETS, int NPTS>
__device__ __forceinline__ void gather_coord_grad(const mne_scene_t& sc, const float* pn, const float* dfeat, float* dpn, int lane) {
    const int cg = lane & 7;
#pragma unroll 1
    for (int it = 0; it < NPTS / 8; ++it) {
        const int slot = it * 8 + (lane >> 3);
        const float px = pn[slot * 4 + 0], py = pn[slot * 4 + 1], pz = pn[slot * 4 + 2];
        float g3[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int set = 0; set < NSETS; ++set) {
#pragma unroll
            for (int lvl = 0; lvl < 2; ++lvl) {
                const float4 df = *(const float4*)(dfeat + set * NPTS * MNE_FS + slot * MNE_FS + lvl * MNE_C + cg * 4);
#pragma unroll
                for (int ori = 0; ori < 3; ++ori) {
                    const mne_plane_t& pl = sc.plane[set][ori][lvl];
                    float gx, gy;
                    orient_coords(ori, px, py, pz, gx, gy);
                    Bilin b;
                    bilin_setup(gx, gy, pl.h, pl.w, b);
                    const float ux = ((gx + 1.0f) / 2.0f) * (float)(pl.w - 1), uy = ((gy + 1.0f) / 2.0f) * (float)(pl.h - 1);
                    const float fx = fminf((float)(pl.w - 1), fmaxf(ux, 0.0f)), fy = fminf((float)(pl.h - 1), fmaxf(uy, 0.0f));
                    const float x0 = floorf(fx), y0 = floorf(fy);
                    float4 v00, v01, v10, v11;
                    if (sc.plane_f16) {
                        v00 = plane_row4<true>(pl, cg * 4 + b.o00); v01 = plane_row4<true>(pl, cg * 4 + b.o01);
                        v10 = plane_row4<true>(pl, cg * 4 + b.o10); v11 = plane_row4<true>(pl, cg * 4 + b.o11);
                    } else {
                        v00 = plane_row4<false>(pl, cg * 4 + b.o00); v01 = plane_row4<false>(pl, cg * 4 + b.o01);
                        v10 = plane_row4<false>(pl, cg * 4 + b.o10); v11 = plane_row4<false>(pl, cg * 4 + b.o11);
                    }
                    const bool xin = b.ix0 + 1 < pl.w, yin = b.iy0 + 1 < pl.h;
                    // dot(dfeat, corner) over this lane's 4 channels; absent corners count as zero
                    const float d00 = df.x * v00.x + df.y * v00.y + df.z * v00.z + df.w * v00.w;
                    const float d01 = xin ? df.x * v01.x + df.y * v01.y + df.z * v01.z + df.w * v01.w : 0.0f;
                    const float d10 = yin ? df.x * v10.x + df.y * v10.y + df.z * v10.z + df.w * v10.w : 0.0f;
                    const float d11 = (xin && yin) ? df.x * v11.x + df.y * v11.y + df.z * v11.z + df.w * v11.w : 0.0f;
                    const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
                    float gix = -d00 * (y1 - fy) + d01 * (y1 - fy) - d10 * (fy - y0) + d11 * (fy - y0);
                    float giy = -d00 * (x1 - fx) - d01 * (fx - x0) + d10 * (x1 - fx) + d11 * (fx - x0);
                    // clip gradient (ATen clip_coordinates_set_grad): zero at/below 0 and at/above size-1
                    gix *= (ux <= 0.0f || ux >= (float)(pl.w - 1)) ? 0.0f : (float)(pl.w - 1) / 2.0f;
                    giy *= (uy <= 0.0f || uy >= (float)(pl.h - 1)) ? 0.0f : (float)(pl.h - 1) / 2.0f;
                    // xy -> (x,y); xz -> (x,z); yz -> (y,z)
                    if (ori == MNE_XY) { g3[0] += gix; g3[1] += giy; }
                    else if (ori == MNE_XZ) { g3[0] += gix; g3[2] += giy; }
                    else { g3[1] += gix; g3[2] += giy; }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            g3[k] += __shfl_xor(g3[k], 1); g3[k] += __shfl_xor(g3[k], 2); g3[k] += __shfl_xor(g3[k], 4);
        }
        if (cg == 0) { dpn[slot * 4 + 0] = g3[0]; dpn[slot * 4 + 1] = g3[1]; dpn[slot * 4 + 2] = g3[2]; }
    }
}

// ---- scatter: d(feature) rows in LDS -> atomic adds into the plane gradients -------------------
// Lane layout: 32 lanes = the 32 channels of one corner row (one 128-B line per half-wave).
// `live`: bit s set = slot s holds a sample that receives gradient (the others have an all-zero row).
template <int NSETS, int NPTS>
__device__ __forceinline__ void scatter_chunk(const mne_scene_t& sc, const float* pn, const float* dfeat,
                                              unsigned live, int lane, int set_lo = 0, int set_hi = NSETS,
                                              bool one_buffer = false) {      // one_buffer: dfeat holds the rows of set_lo only
    const int c = lane & 31, half = lane >> 5;
#pragma unroll 1
    for (int it = 0; it < NPTS / 2; ++it) {
        const int slot = it * 2 + half;
        if ((live >> slot) & 1u) {
            const float px = pn[slot * 4 + 0], py = pn[slot * 4 + 1], pz = pn[slot * 4 + 2];
#pragma unroll
            for (int set = 0; set < NSETS; ++set) {
                if (set < set_lo || set >= set_hi) continue;
#pragma unroll
                for (int lvl = 0; lvl < 2; ++lvl) {
                    const float g = dfeat[(one_buffer ? 0 : set) * NPTS * MNE_FS + slot * MNE_FS + lvl * MNE_C + c];
#pragma unroll
                    for (int ori = 0; ori < 3; ++ori) {
                        const mne_plane_t& pl = sc.plane[set][ori][lvl];
                        float gx, gy;
                        orient_coords(ori, px, py, pz, gx, gy);
                        Bilin b;
                        bilin_setup(gx, gy, pl.h, pl.w, b);
                        float* base = pl.grad + c;
                        unsafeAtomicAdd(base + b.o00, g * b.w00);
                        if (b.w01 != 0.0f) unsafeAtomicAdd(base + b.o01, g * b.w01);
                        if (b.w10 != 0.0f) unsafeAtomicAdd(base + b.o10, g * b.w10);
                        if (b.w11 != 0.0f) unsafeAtomicAdd(base + b.o11, g * b.w11);
                    }
                }
            }
        }
    }
}

// ---- decoder dimensions and the tape row --------------------------------------------------------------
// One tape row per SAMPLE (row index = ray * S + sample; only rows of decoded tiles are ever touched):
//   forward part, written by whoever decodes the sample (decode_kernel, or ray_kernel's on-demand decode):
//     X[112] = plane features (64) | OneBlob (48)      H[HID] = relu(W1 x)      OUT[16] = (sdf, geo15)
//     HC[HIDC] = relu(V1 [pos,(cf),geo])               CF[64] = colour-plane features (colour planes only)
//   backward part, written by ray_kernel:
//     DH[HID] | DOUT[16] | DHC[HIDC] | DC[4] = d(total)/d(pre-activations) -- the dY operands of the decoder
//     weight-gradient GEMMs (wgrad.hip) -- then DFEAT[64 per plane set] = d(total)/d(plane features) and PN[4] =
//     the normalised point, read by the binned plane update (tile_adam.hip).
template <int HID, int HIDC, bool CP>
struct DecDims {
    static constexpr int CINB = CP ? (MNE_POS + MNE_FEAT) : MNE_POS;   // where geo starts in the colour input
    static constexpr int CIN = CINB + MNE_GEO;                           // 63 or 127
    static constexpr int CINP = CINB + MNE_OUT1;                         // GEMM view of the colour input: [pos | (cf) | out16]
    // forward half: [feat 64 | pos 48 | out 16 | h HID | hc HIDC | (cf 64)] -- 64-float chunks [pos|out], [h|hc] (staged stores)
    static constexpr int T_X = 0;
    static constexpr int T_OUT = T_X + MNE_IN1;
    static constexpr int T_H = T_OUT + MNE_OUT1;
    static constexpr int T_HC = T_H + HID;
    static constexpr int T_CF = T_HC + HIDC;
    static constexpr int T_FWD_END = T_CF + (CP ? MNE_FEAT : 0);
    // backward half: [dh HID | dhc HIDC | dout 16 | dc 4 | pn 4 | pad 8 | dfeat 64 per plane set]
    static constexpr int T_DH = T_FWD_END;
    static constexpr int T_DHC = T_DH + HID;
    static constexpr int T_DOUT = T_DHC + HIDC;
    static constexpr int T_DC = T_DOUT + MNE_OUT1;
    static constexpr int T_PN = T_DC + 4;
    static constexpr int T_DFEAT = T_DOUT + 32;
    static constexpr int ROW = T_DFEAT + (CP ? 2 : 1) * MNE_FEAT;
    // tape column of element c (0 <= c < CINP) of the colour-net input [pos48 | (cf64) | out16]
    __host__ __device__ static constexpr int cin_col(int c) {
        return c < MNE_POS ? T_X + MNE_FEAT + c : (CP && c < MNE_POS + MNE_FEAT) ? T_CF + (c - MNE_POS) : T_OUT + (c - CINB);
    }
    // decoder parameter buffer (order of decoder.parameters()): col0 | col1 | sdf0 | sdf1
    static constexpr int P_COL0 = 0;
    static constexpr int P_COL1 = P_COL0 + HIDC * CIN;
    static constexpr int P_SDF0 = P_COL1 + 3 * HIDC;
    static constexpr int P_SDF1 = P_SDF0 + HID * MNE_IN1;
    static constexpr int NPARAM = P_SDF1 + MNE_OUT1 * HID;
};

// ---- staged tape stores ----------------------------------------------------------------------------------
// Tape rows are 1.4-2 KB apart, so a store of "what each lane holds" touches 64 different cache lines with 16 bytes
// each (store-issue-bound, ~7 B/clk/CU).  Everything that goes to the tape is therefore first laid out as rows in LDS
// ([32 points][MNE_FS], columns [0, NCOL)) and then written by groups of 8 lanes x float4 = one full 128-B line per
// point and instruction (8 lines per wave instruction).  `live`: bit s = point s is written (32 points: a 32-bit mask --
// a 64-bit one made the compiler keep four per-lane 64-bit bit constants alive across the whole tile loop).
template <int NCOL>
__device__ __forceinline__ void store_rows(const float* rows, float* tape_rows0, int row_stride, int tcol,
                                           unsigned live, int lane) {
    const int cg = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int slot = it * 8 + (lane >> 3);
        if ((live >> slot) & 1u) {
#pragma unroll
            for (int hfc = 0; hfc < NCOL / 32; ++hfc)
                *(float4*)(tape_rows0 + (size_t)slot * row_stride + tcol + hfc * 32 + cg * 4) =
                    *(const float4*)(rows + slot * MNE_FS + hfc * 32 + cg * 4);
        }
    }
}

// load_rows in two halves, so that independent work (OneBlob: pure VALU) can run while the rows are in flight:
// rows_fetch issues the coalesced loads into registers (eight float4 per lane for 64 columns), rows_commit writes them to
// the LDS rows.  (Plain reference-to-array parameters: a wrapper struct ended up in scratch.)
template <int NCOL>
__device__ __forceinline__ void rows_fetch(float4 (&v)[4][NCOL / 32], const float* tape_rows0, int row_stride, int tcol, unsigned live, int lane) {
    const int cg = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int slot = it * 8 + (lane >> 3);
        const int src = ((live >> slot) & 1u) ? slot : 0;              // rows beyond the ray's last sample: any valid row
#pragma unroll
        for (int hfc = 0; hfc < NCOL / 32; ++hfc)
            v[it][hfc] = *(const float4*)(tape_rows0 + (size_t)src * row_stride + tcol + hfc * 32 + cg * 4);
    }
}
template <int NCOL>
__device__ __forceinline__ void rows_commit(const float4 (&v)[4][NCOL / 32], float* rows, int lane) {
    const int cg = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int slot = it * 8 + (lane >> 3);
#pragma unroll
        for (int hfc = 0; hfc < NCOL / 32; ++hfc) *(float4*)(rows + slot * MNE_FS + hfc * 32 + cg * 4) = v[it][hfc];
    }
}

// the reverse of store_rows: tape columns [tcol, tcol + NCOL) of the tile's rows -> LDS rows, one batch of coalesced loads
template <int NCOL>
__device__ __forceinline__ void load_rows(float* rows, const float* tape_rows0, int row_stride, int tcol,
                                          unsigned live, int lane) {
    const int cg = lane & 7;
    float4 v[4][NCOL / 32];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int slot = it * 8 + (lane >> 3);
        const int src = ((live >> slot) & 1u) ? slot : 0;              // rows beyond the ray's last sample: any valid row
#pragma unroll
        for (int hfc = 0; hfc < NCOL / 32; ++hfc)
            v[it][hfc] = *(const float4*)(tape_rows0 + (size_t)src * row_stride + tcol + hfc * 32 + cg * 4);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int slot = it * 8 + (lane >> 3);
#pragma unroll
        for (int hfc = 0; hfc < NCOL / 32; ++hfc) *(float4*)(rows + slot * MNE_FS + hfc * 32 + cg * 4) = v[it][hfc];
    }
}
