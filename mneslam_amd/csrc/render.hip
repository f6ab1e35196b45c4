// render.hip -- z sampling, fused tri-plane gather -> OneBlob -> MFMA tiny-MLP -> SDF compositing
// (forward) and its backward (loss gradients -> MFMA MLP backward -> plane-gradient scatter/append +
// decoder tape) for the MNE-SLAM mapping iteration on gfx950.
//
// Work decomposition of one training iteration (mne_render_fused):
//   decode_kernel  one wave per (ray, 32-sample tile), two lanes per point, persistent grid: coalesced gather
//                  (8 lanes x 16 B per 128-B corner row) -> feature rows in LDS -> OneBlob in registers -> MFMA
//                  chain (mlp_mfma.h) -> raw (r,g,b,sdf), the ReLU bit masks and the forward half of the tape row.
//                  EARLY RAY TERMINATION: only the tiles a ray needs a priori are decoded -- those holding a sample
//                  with a loss mask (z <= target depth + truncation; known from z and the target depth alone,
//                  counted by sample_z_kernel) and at least the first tile.
//   ray_kernel     one wave per ray: finds the first SDF sign change among the decoded samples; if the render
//                  window (z < z_first + trunc) or the search itself runs past the decoded prefix, the wave decodes
//                  further tiles ON DEMAND (same code as decode_kernel) until the ray is resolved -- so the result is
//                  exactly the reference's, whatever the scene looks like.  Then compositing (weights, maps, loss
//                  partial sums; shuffles/ballot) and, for training, the backward of the ray's own tiles: loss /
//                  compositing gradients -> MFMA backward chain from the saved ReLU masks (no forward recompute, no
//                  re-gather) -> backward half of the tape row -> plane-gradient appends (binned) or atomics.
// Samples beyond the last one a ray needs are never touched: they have zero weight and no loss term in the reference.
//
// Reference semantics: model/scene_rep.py:28-53,183-230,351-419,475-611; model/decoder.py:110-175;
// model/utils.py:27-41,117-185 (include/mneslam_hip.h maps each entry point).
#include "mlp_mfma.h"
#include "mne_launch.h"
#include "mne_sampler.h"

#define RAYS_PER_WG 4
#define TILE 32

// -----------------------------------------------------------------------------------------------
// z sampling + mask counts: one wave per ray, linspace tables staged in LDS once per workgroup
// -----------------------------------------------------------------------------------------------
// z samples (+ per-ray mask counts) of ray r by one wave; d = its target depth (unused without depth guidance); tab =
// the staged linspace tables, `vals` = S floats of this wave's LDS
__device__ __forceinline__ void sample_z_ray(const ZArgs& a, int r, float d, int lane, const float* tab, float* vals) {
    const int S = a.S;
    const uint64_t z_offset = a.offset + (a.clk.iteration ? *a.clk.iteration * a.clk.z_offset_stride : 0ull);
    if (a.has_d) {
        const float* uni = tab;
        const float* surf = tab + a.n_a;
        const float* inval = tab + a.n_a + a.n_b;
        const bool invalid = d <= 0.0f;                             // scene_rep.py:365
        // stable merge of two ascending sequences by rank (== torch.sort of their concatenation)
        for (int e = lane; e < S; e += MNE_WAVE) {
            if (e < a.n_a) {
                const float v = uni[e];
                int lo = 0, hi = a.n_b;                             // #b strictly below v
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const float bv = invalid ? inval[mid] : surf[mid] + d;
                    if (bv < v) lo = mid + 1; else hi = mid;
                }
                vals[e + lo] = v;
            } else {
                const int j = e - a.n_a;
                const float v = invalid ? inval[j] : surf[j] + d;
                int lo = 0, hi = a.n_a;                             // #a at or below v
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (uni[mid] <= v) lo = mid + 1; else hi = mid;
                }
                vals[j + lo] = v;
            }
        }
    } else {
        for (int e = lane; e < S; e += MNE_WAVE) vals[e] = tab[e];
    }
    MNE_WAVE_SYNC();
    int n_front = 0, n_center = 0, n_tail = 0, n_cofs = 0, n_cosdf = 0, n_need = 0;
    for (int i = lane; i < S; i += MNE_WAVE) {
        float z = vals[i];
        if (a.perturb > 0.0f) {                                     // scene_rep.py:377-381
            const float zm = vals[i > 0 ? i - 1 : 0], zp = vals[i < S - 1 ? i + 1 : S - 1];
            const float lower = i > 0 ? 0.5f * (z + zm) : z;
            const float upper = i < S - 1 ? 0.5f * (zp + z) : z;
            const uint64_t e = (uint64_t)r * (uint64_t)S + (uint64_t)i;
            const float uu = a.u ? a.u[e] : philox_uniform(a.seed, z_offset, e);
            z = lower + (upper - lower) * uu;
        }
        a.z_vals[(size_t)r * S + i] = z;
        if (a.has_d) {
            if (d > 0.0f) {                                         // ESLAM masks, scene_rep.py:489-499 (d>0 rays, :589)
                const bool front = z < (d - a.e_T), back = z > (d + a.e_T);
                const bool center = (z > (d - a.e_T04)) && (z < (d + a.e_T04));
                n_front += front;
                n_center += center;
                n_tail += (!front && !back && !center);
            }
            const bool cf = z < (d - a.co_T), cb = z > (d + a.co_T);   // Co-SLAM masks, model/utils.py:131-137
            n_cofs += cf;
            n_cosdf += (!cf && !cb && d > 0.0f);
            // samples that can carry a loss term whatever the decoder says: not behind either truncation band
            // (z sorted: a prefix of the ray).  The render kernels decode at least these (early ray termination).
            // A ray without valid depth has no loss term but its first SDF sign change can sit anywhere: all its samples.
            n_need += (d > 0.0f) ? ((!(z > (d + a.e_T)) || !cb) ? 1 : 0) : 1;
        }
    }
    if (a.has_d) {
        int sums[6] = {n_front, n_center, n_tail, n_cofs, n_cosdf, n_need};
#pragma unroll
        for (int k = 0; k < 6; ++k)
            for (int m = 32; m >= 1; m >>= 1) sums[k] += __shfl_xor(sums[k], m);
        if (lane == 0) {                 // per-ray counts; summed by counts_reduce_kernel (no same-address atomics)
            int* rc = a.ray_counts + (size_t)r * MNE_N_COUNT;
            rc[MNE_C_VALID] = (d > 0.0f && d < a.depth_trunc) ? 1 : 0;          // scene_rep.py:570
            rc[MNE_C_E_FRONT] = sums[0]; rc[MNE_C_E_CENTER] = sums[1]; rc[MNE_C_E_TAIL] = sums[2];
            rc[MNE_C_CO_FS] = sums[3]; rc[MNE_C_CO_SDF] = sums[4]; rc[MNE_C_NEED] = sums[5]; rc[7] = 0;
        }
    }
}

__device__ __forceinline__ int z_tab_entries(const ZArgs& a) { return a.has_d ? a.n_a + 2 * a.n_b : a.S; }

__global__ __launch_bounds__(256) void sample_z_kernel(ZArgs a) {
    MNE_DYN_LDS(lds_raw);
    const int S = a.S, n_tab = z_tab_entries(a);
    float* tab = (float*)lds_raw;                                   // [n_tab]
    for (int i = threadIdx.x; i < n_tab; i += blockDim.x) tab[i] = a.tables[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.x * RAYS_PER_WG + w;
    if (r >= a.R) return;                                           // whole wave leaves together
    sample_z_ray(a, r, a.has_d ? a.target_d[r] : 0.0f, lane, tab, tab + ((n_tab + 3) & ~3) + w * ((S + 3) & ~3));
}

// Batches of more rays than this (whole frames: 816 k rays) neither fit the balanced decode schedule (its tile-count prefix of
// every ray sits in LDS) nor a one-workgroup reduction: their counts are summed by many workgroups (integer atomics) and slot
// MNE_C_TILE0 is left unwritten -- launch_decode applies the same bound.  (Round 5 first ran the one-workgroup form on frames:
// 3.3 ms of a 22 ms depth-guided render, profiles/r05_render_img_experiments.txt.)
#ifndef MNE_BALANCED_MAX_RAYS
#define MNE_BALANCED_MAX_RAYS 16384
#endif
// sum of the per-ray mask counts: all 256 threads of ONE workgroup (gridDim.x == 1), or this workgroup's stripe of the rays
// added to a.counts (zeroed by the launcher)
__device__ __forceinline__ void counts_reduce_block(const ZArgs& a, int (*part)[MNE_N_COUNT]) {
    const int t = threadIdx.x;
    const bool striped = gridDim.x > 1;
    int acc[MNE_N_COUNT];
    for (int k = 0; k < MNE_N_COUNT; ++k) acc[k] = 0;
    for (long long r = (long long)blockIdx.x * 256 + t; r < a.R; r += (long long)gridDim.x * 256)
        for (int k = 0; k < MNE_N_COUNT; ++k) acc[k] += a.ray_counts[(size_t)r * MNE_N_COUNT + k];
    for (int k = 0; k < MNE_N_COUNT; ++k) part[t][k] = acc[k];
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (t < st)
            for (int k = 0; k < MNE_N_COUNT; ++k) part[t][k] += part[t + st][k];
        __syncthreads();
    }
    if (striped) {
        if (t < MNE_N_COUNT && t != MNE_C_TILE0 && part[0][t]) atomicAdd(a.counts + t, part[0][t]);
        return;
    }
    if (t < MNE_N_COUNT) a.counts[t] = part[0][t];
    if (a.R > MNE_BALANCED_MAX_RAYS) return;
    // Exclusive prefix of the rays' a-priori tile counts -> slot MNE_C_TILE0 of every ray's counts (decode_kernel deals its
    // tile tasks out from it, evenly over the waves): 256 rays per round, wave scan + the four waves' sums through LDS.
    __syncthreads();
    const int ntile = (a.S + 31) / 32, lane = t & 63, wave = t >> 6;
    int base = 0;
    for (int r0 = 0; r0 < a.R; r0 += 256) {
        const int r = r0 + t;
        int v = 0;
        if (r < a.R) {
            const int tl = (a.ray_counts[(size_t)r * MNE_N_COUNT + MNE_C_NEED] + 31) / 32;
            v = tl < 1 ? 1 : (tl > ntile ? ntile : tl);
        }
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d); if (lane >= d) inc += u; }
        if (lane == 63) part[0][wave] = inc;
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int sw = part[0][w]; woff += w < wave ? sw : 0; total += sw; }
        if (r < a.R) a.ray_counts[(size_t)r * MNE_N_COUNT + MNE_C_TILE0] = base + woff + inc - v;
        base += total;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void counts_reduce_kernel(ZArgs a) {
    __shared__ int part[256][MNE_N_COUNT];
    counts_reduce_block(a, part);
}

__device__ __forceinline__ void loss_coef_thread(const LossArgs& a);

// batch_kernel + counts_coef_kernel: the per-iteration batch preparation (R1-R3 + the loss coefficients) in TWO launches
// instead of four (sample_rays, sample_z, counts_reduce, loss_coef: ~70 us of launch latency in the decoder's dependency
// chain, profiles/r03_timeline_mid.txt) -- ray draw and pose rotation (sampler.hip) + z samples + per-ray mask counts;
// then the sum of the counts and the d(total)/d(sample) coefficients.  (One launch with a "last workgroup" epilogue was
// measured and is WORSE: its agent-scope release fences write back the L2 under the concurrent plane update, which lost
// 20 us -- profiles/r03_fused_small_kernels.txt.)
__global__ __launch_bounds__(256) void batch_kernel(SampleRaysArgs sr, ZArgs a) {
    MNE_DYN_LDS(lds_raw);
    const int S = a.S, n_tab = z_tab_entries(a);
    float* tab = (float*)lds_raw;                                   // [n_tab]
    for (int i = threadIdx.x; i < n_tab; i += blockDim.x) tab[i] = a.tables[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.x * RAYS_PER_WG + w;
    if (r < a.R) {
        const float d = sample_ray(sr, r, lane == 0);              // every lane draws the (same) ray, lane 0 stores it
        sample_z_ray(a, r, d, lane, tab, tab + ((n_tab + 3) & ~3) + w * ((S + 3) & ~3));
    }
}

// sum of the per-ray counts + the loss coefficients: the second (one-workgroup) launch of the batch preparation
__global__ __launch_bounds__(256) void counts_coef_kernel(ZArgs a, LossArgs lc) {
    __shared__ int part[256][MNE_N_COUNT];
    counts_reduce_block(a, part);
    __syncthreads();
    if (threadIdx.x == 0 && lc.coef) loss_coef_thread(lc);
}

// -----------------------------------------------------------------------------------------------
// decoder packing: the MFMA A-operand tables of mlp_mfma.h (one 64-float row per k-step)
// -----------------------------------------------------------------------------------------------
template <int HID, int HIDC, bool CP>
__global__ __launch_bounds__(256) void pack_decoder_kernel(mne_scene_t sc, float* pk) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ATab<HID, HIDC, CP>::TOTAL_RAYGRAD * 64) pk[t] = atab_value<HID, HIDC, CP>(sc, t >> 6, t & 63);
}

// -----------------------------------------------------------------------------------------------
// render kernels
// -----------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 f32x16_zero() { f32x16 v; for (int q = 0; q < 16; ++q) v[q] = 0.0f; return v; }

struct SampleMasks { bool e_front, e_center, e_tail, co_fs, co_sdf; };

__device__ __forceinline__ SampleMasks sample_masks(float z, float d, bool has_t, const RenderArgs& a) {
    SampleMasks m = {false, false, false, false, false};
    if (has_t) {
        if (d > 0.0f) {
            const bool front = z < (d - a.e_T), back = z > (d + a.e_T);
            const bool center = (z > (d - a.e_T04)) && (z < (d + a.e_T04));
            m.e_front = front; m.e_center = center; m.e_tail = !front && !back && !center;
        }
        const bool cf = z < (d - a.win_f), cb = z > (d + a.win_f);
        m.co_fs = cf; m.co_sdf = !cf && !cb && d > 0.0f;
    }
    return m;
}

#ifndef MNE_DECODE_WIDE_GATHER
#define MNE_DECODE_WIDE_GATHER 1  // see decode_tile<..., WIDE>
#endif
// per-wave LDS of the tile code: pn[32][4] | feat[NSETS][32][FS]  (+ ray-gradient variant: dpos[32][64] | dpn[32][4])
__host__ __device__ inline size_t tile_wave_lds_bytes(int nsets, bool raygrad = false) {
    size_t b = (size_t)(TILE * 4 + nsets * TILE * MNE_FS) * sizeof(float);
    if (raygrad) b += (size_t)(TILE * 64 + TILE * 4) * sizeof(float);
    return b;
}

// number of leading tiles of ray r that are decoded a priori (see the file header)
// a-priori tiles of ray r from its loss-mask count alone (what the exact early termination decodes tile-parallel)
__device__ __forceinline__ int apriori_tiles(const RenderArgs& a, int r, int ntile) {
    const int need = a.ray_counts[(size_t)r * MNE_N_COUNT + MNE_C_NEED];
    const int t = (need + TILE - 1) / TILE;
    return t < 1 ? 1 : (t > ntile ? ntile : t);
}
__device__ __forceinline__ int prefix_tiles(const RenderArgs& a, int r, int ntile, bool listed = false) {
    if (listed) return ntile;                    // second pass over the deferred rays: everything is decoded
    if (a.tile_need) return a.tile_need[r];      // backward of an earlier forward call: the tiles its backward will walk
    if (!a.ray_counts) return a.prefix_default < ntile ? a.prefix_default : ntile;
    if (a.adapt && a.adapt[0]) return ntile;     // adaptive schedule, mode 1: most rays would be deferred -> decode everything a priori
    const int need = a.ray_counts[(size_t)r * MNE_N_COUNT + MNE_C_NEED];
    const int t = (need + TILE - 1) / TILE;
    return t < 1 ? 1 : (t > ntile ? ntile : t);
}

// Decode tile c of ray r with the calling wave: raw -> a.raw (when given), ReLU masks -> a.relu_mask, forward half of
// the tape rows -> a.tape (staged through the wave's LDS rows: full-line stores, see store_rows).  Returns (r,g,b,sdf)
// of this lane's point (valid lanes); pnv/u are its coordinates, relu its mask words.
// PRE: the plane features of the tile are already in the tape rows (gather_kernel): they are loaded back into the LDS
// rows with one batch of coalesced loads instead of being gathered here (8 dependent rounds of corner-row loads).
// SEQF (forward-only launches of a model with colour planes: no tape, no pre-gathered rows): `feat` is ONE set of 32 rows -- the
// geometry planes are gathered, the sdf net runs, then the colour planes are gathered into the same rows for the colour net.
// Same operations on the same values; what it buys is LDS: 9.2 instead of 17.9 KB per wave, i.e. 8 waves per CU beside the
// 38.9 KB of tables instead of 6 (decode_kernel) / 5 (ray_kernel<..., 0> at 256 samples per ray) -- render_img, DESIGN.md 3.6.
// WIDE (decode_kernel of 2x32 decoders without colour planes, the fused training launch): the inline gather requests both levels' 24 corner
// rows together and goes through buffer loads, as gather_kernel does -- the kernel has the registers (249 of 256, no scratch), the other
// callers of this function do not (round 6, profiles/r06_inline_gather.txt: +1.1 % on office0, nothing elsewhere).
template <int HID, int HIDC, bool CP, bool GTAB = false, bool SEQF = false, bool WIDE = false>
__device__ __forceinline__ float4 decode_tile(const RenderArgs& a, int r, int c, int lane, float* pn, float* feat,
                                              const float* atab, float (&pnv)[3], float (&u)[3], uint2& relu, bool PRE = false) {
    typedef DecDims<HID, HIDC, CP> D;
    constexpr int NSETS = CP ? 2 : 1;
    constexpr int NT = HID / 32, NTC = HIDC / 32;
    const int S = a.S, pt = lane & 31, hf = lane >> 5;
    const int i = c * TILE + pt;
    const bool valid = i < S;
    const float z = a.z_vals[(size_t)r * S + (valid ? i : S - 1)];
    float p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = a.rays_o[r * 3 + k] + a.rays_d[r * 3 + k] * z;      // scene_rep.py:384
    point_coords(a.sc, p, pnv, u);
    const int n_here = S - c * TILE;
    const unsigned live = n_here >= TILE ? 0xffffffffu : ((1u << n_here) - 1u);
    float* tape0 = a.tape ? a.tape + ((size_t)r * S + (size_t)c * TILE) * D::ROW : nullptr;
    MNE_WAVE_SYNC();                                       // earlier LDS reads of this wave are done
    if constexpr (SEQF) {
        static_assert(CP, "SEQF: the one-set form exists for models with colour planes");
        if (hf == 0) *(float4*)(pn + pt * 4) = make_float4(pnv[0], pnv[1], pnv[2], 0.0f);
        MNE_WAVE_SYNC();
        gather_chunk<1, TILE, MNE_INLINE_GATHER_NLV, 0>(a.sc, pn, feat, lane);
        MNE_WAVE_SYNC();
        const float* row = feat + pt * MNE_FS;
        float pos[24];
        oneblob_half<!(HID == 64 && CP)>(u, hf, pos);
        MlpState<HID, HIDC> st;
        mlp_forward_sdf<HID, HIDC, CP, GTAB>(row, pos, atab, lane, st);
        MNE_WAVE_SYNC();                                   // every lane has read its geometry row
        gather_chunk<1, TILE, MNE_INLINE_GATHER_NLV, 1>(a.sc, pn, feat, lane);
        MNE_WAVE_SYNC();
        mlp_forward_color<HID, HIDC, CP, GTAB>(row, pos, atab, lane, st);
        const float4 rw = make_float4(st.rgb[0], st.rgb[1], st.rgb[2], st.out[0]);
        relu = make_uint2(0u, 0u);
        if (a.relu_mask) relu_masks<HID, HIDC>(st, relu.x, relu.y);
        if (valid) {
            const size_t e = (size_t)r * S + i;
            if (a.raw && hf == 0) *(float4*)(a.raw + e * 4) = rw;
            if (a.relu_mask) *(uint2*)(a.relu_mask + e * 4 + hf * 2) = relu;
        }
        return rw;
    }
    // Pre-gathered rows, 2x32 decoders (registers to spare): the row loads are issued here and land in LDS only after the
    // OneBlob below (3 us of VALU work that does not depend on them) -- 2 us of load latency per tile off the chain.
    constexpr bool OVERLAP = HID == 32 && HIDC == 32;
    float4 q0[4][MNE_FEAT / 32], q1[4][MNE_FEAT / 32];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int k = 0; k < MNE_FEAT / 32; ++k) q0[it][k] = q1[it][k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool late_rows = OVERLAP && PRE;
    if (late_rows) {
        if (!CP && a.ext_rows) rows_fetch<MNE_FEAT>(q0, a.ext_rows + ((size_t)r * S + (size_t)c * TILE) * a.ext_stride, a.ext_stride, 0, live, lane);
        else rows_fetch<MNE_FEAT>(q0, tape0, D::ROW, D::T_X, live, lane);
        if (CP) rows_fetch<MNE_FEAT>(q1, tape0, D::ROW, D::T_CF, live, lane);
    } else if (PRE) {
        if (!CP && a.ext_rows) load_rows<MNE_FEAT>(feat, a.ext_rows + ((size_t)r * S + (size_t)c * TILE) * a.ext_stride, a.ext_stride, 0, live, lane);
        else load_rows<MNE_FEAT>(feat, tape0, D::ROW, D::T_X, live, lane);
        if (CP) load_rows<MNE_FEAT>(feat + TILE * MNE_FS, tape0, D::ROW, D::T_CF, live, lane);
        MNE_WAVE_SYNC();
    } else {
        if (hf == 0) *(float4*)(pn + pt * 4) = make_float4(pnv[0], pnv[1], pnv[2], 0.0f);
        MNE_WAVE_SYNC();
        gather_chunk<NSETS, TILE, WIDE ? 2 : MNE_INLINE_GATHER_NLV, 0, WIDE>(a.sc, pn, feat, lane);
        MNE_WAVE_SYNC();
        if (tape0) {                                       // plane features: straight from the gathered rows
            store_rows<MNE_FEAT>(feat, tape0, D::ROW, D::T_X, live, lane);
            if (CP) store_rows<MNE_FEAT>(feat + TILE * MNE_FS, tape0, D::ROW, D::T_CF, live, lane);
        }
    }
    float* frow = feat + pt * MNE_FS;
    const float* cfrow = feat + TILE * MNE_FS + pt * MNE_FS;
    float pos[24];
    oneblob_half<!(HID == 64 && CP)>(u, hf, pos);
    if (late_rows) {
        rows_commit<MNE_FEAT>(q0, feat, lane);
        if (CP) rows_commit<MNE_FEAT>(q1, feat + TILE * MNE_FS, lane);
        MNE_WAVE_SYNC();
    }
    MlpState<HID, HIDC> st;
    mlp_forward_mfma<HID, HIDC, CP, GTAB>(frow, cfrow, pos, atab, lane, st);
    const float4 rw = make_float4(st.rgb[0], st.rgb[1], st.rgb[2], st.out[0]);      // rows 0..3 live in the lower half
    relu = make_uint2(0u, 0u);
    if (a.relu_mask) relu_masks<HID, HIDC>(st, relu.x, relu.y);
    if (valid) {
        const size_t e = (size_t)r * S + i;
        if (a.raw && hf == 0) *(float4*)(a.raw + e * 4) = rw;
        if (a.relu_mask) *(uint2*)(a.relu_mask + e * 4 + hf * 2) = relu;
    }
    if (tape0) {
        // [pos 48 | out 16] -> tape columns 64..127, then [h | hc]; the feature rows are dead after the forward chain
        MNE_WAVE_SYNC();
#pragma unroll
        for (int q = 0; q < 6; ++q)
            *(float4*)(frow + hf * 24 + 4 * q) = make_float4(pos[4 * q], pos[4 * q + 1], pos[4 * q + 2], pos[4 * q + 3]);
        acc_to_row(frow, MNE_POS, st.out, hf, 8);
        MNE_WAVE_SYNC();
        store_rows<64>(feat, tape0, D::ROW, D::T_X + MNE_FEAT, live, lane);
        MNE_WAVE_SYNC();
        if (HID == 32 && HIDC == 32) {
            acc_to_row(frow, 0, st.h[0], hf);
            acc_to_row(frow, 32, st.hc[0], hf);
            MNE_WAVE_SYNC();
            store_rows<64>(feat, tape0, D::ROW, D::T_H, live, lane);
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc_to_row(frow, 32 * t, st.h[t], hf);
            MNE_WAVE_SYNC();
            store_rows<HID>(feat, tape0, D::ROW, D::T_H, live, lane);
            MNE_WAVE_SYNC();
#pragma unroll
            for (int t = 0; t < NTC; ++t) acc_to_row(frow, 32 * t, st.hc[t], hf);
            MNE_WAVE_SYNC();
            store_rows<HIDC>(feat, tape0, D::ROW, D::T_HC, live, lane);
        }
    }
    return rw;
}

// Waves per workgroup decode_kernel is compiled for: 8 = 256 registers per lane, no spills in any decoder shape (a
// 12-wave build for the inline-gather form hid a little more latency and spilled 16-376 B per lane: dropped in round 3,
// see DESIGN.md section 9 for why spills are treated as defects here).
#ifndef DECODE_WPB
#define DECODE_WPB 8
#endif
// Persistent tile kernels walk their tasks with a grid stride; consecutive tasks are consecutive rays -- in a frame render
// neighbouring pixels, whose samples fall into the same plane cells.  Workgroups are dealt to the 8 XCDs round robin, so with
// the plain stride every XCD (own L2) sees every 8th group of rays and fetches the same plane lines as its neighbours.
// xcd_block() renumbers the workgroups so that the 32 workgroups of one XCD take ADJACENT task groups: one contiguous run of
// rays per XCD and round.  (grid a multiple of 8 only; any other grid keeps its numbering.)  Measured on frames: 39.66 vs 39.65 ms
// per pair, i.e. nothing by itself -- the frame kernels are not bound by their fetches; kept because it is what lets the Z-order ray
// walk of render_img keep an XCD's rays in one 16 x 16 pixel patch (HBM traffic of the frame kernels / 4).
#ifndef MNE_XCD_TASK_MAP
#define MNE_XCD_TASK_MAP 1
#endif
__device__ __forceinline__ int xcd_block() {
    const int b = blockIdx.x, g = gridDim.x;
    if (!MNE_XCD_TASK_MAP || (g & 7)) return b;
    return (b & 7) * (g >> 3) + (b >> 3);
}
#ifndef MNE_SEQ_FORWARD
#define MNE_SEQ_FORWARD 1       // forward-only launches with colour planes: one set of LDS feature rows per wave (decode_tile<..., SEQF>)
#endif
#ifndef MNE_FUSED_GATHER
#define MNE_FUSED_GATHER 1      // training launches without colour planes gather inline in decode_kernel (no gather_kernel); see launch_render
#endif
#ifndef MNE_WAVE_MAJOR
#define MNE_WAVE_MAJOR 1        // first-pass tasks / rays of launches on caller-supplied features dealt wave-major over the workgroups (decode_kernel, ray_kernel<..., 4>); 0: workgroup-major everywhere
#endif
#ifndef MNE_DECODE_BALANCED
#define MNE_DECODE_BALANCED 1   // decode_kernel: the rays' real tiles dealt evenly to the waves (0: fixed stride over the (tile, ray) slots)
#endif
// -----------------------------------------------------------------------------------------------
// gather_kernel: tri-plane features of the a-priori samples straight into their tape rows.  The gather is a chain of
// dependent load rounds; inside decode_kernel (12 waves per CU, LDS- and register-bound) it took 21 of the 34 us a tile
// needs.  Here it runs by itself: one wave = 8 consecutive samples of a ray, 8 lanes per sample, no LDS, few registers,
// 32 waves per CU -- the whole batch is in flight at once.  decode_kernel<PRE> then starts from the rows.
// -----------------------------------------------------------------------------------------------
template <bool CP, bool F16>
__global__ __launch_bounds__(256) void gather_kernel(RenderArgs a, int chunks_per_ray) {
    typedef DecDims<32, 32, CP> D0;                        // only the hidden-size independent columns are used (T_X = 0)
    constexpr int NSETS = CP ? 2 : 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long task = (long long)blockIdx.x * 4 + wv;
    if (task >= (long long)a.R * chunks_per_ray) return;
    const int k = (int)(task / a.R), r = (int)(task % a.R);       // chunk-major: the skipped chunks cluster at the end
    const int S = a.S, ntile = (S + TILE - 1) / TILE;
    const int n_samp = prefix_tiles(a, r, ntile) * TILE;
    const int i = k * 8 + (lane >> 3);
    if (k * 8 >= n_samp || k * 8 >= S) return;                    // whole wave leaves together
    const int ii = i < S ? i : S - 1;
    const float z = a.z_vals[(size_t)r * S + ii];
    float p[3], pnv[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        p[q] = a.rays_o[r * 3 + q] + a.rays_d[r * 3 + q] * z;                                      // scene_rep.py:384
        pnv[q] = ((p[q] - a.sc.bound_lo[q]) / (a.sc.bound_hi[q] - a.sc.bound_lo[q])) * 2.0f - 1.0f;  // = point_coords' pn
    }
    float* row = a.tape + ((size_t)r * S + ii) * a.tape_row + a.tape_tx;
    // rows of samples beyond S (last chunk of a ray) are computed on the clamped sample and written twice: harmless
    gather_slot_t<NSETS, F16, MNE_GATHER_INFLIGHT / 12, 0, true>(a.sc, pnv[0], pnv[1], pnv[2], lane & 7, row, a.tape_tcf - a.tape_tx);
}

// WPB: waves per workgroup the kernel is compiled for (12 for the fused gather+MLP form: latency hiding matters most;
// 8 for the pre-gathered form: 256 registers per lane, no spills -- measured 0.531 vs 0.538 ms per iteration)
// Tiles the resolver wave may decode beyond a ray's a-priori prefix before it leaves the rest to the deferred pass: the
// extension is SERIAL in one wave (34 us per tile); INS Indoor has 33 tiles per ray and rays that cross empty space:
// unbounded 606 it/s, 4 tiles 791, 2 tiles 828, 1 tile 846; office0 / ScanNet within noise (profiles/r02_resolver_ext.txt).
template <int HID, int HIDC, bool CP, bool ALDS, int WPB, bool SEQF = false>
__global__ __launch_bounds__(64 * WPB) void decode_kernel(RenderArgs a, int pre, int sched) {
    typedef ATab<HID, HIDC, CP> T;
    constexpr int NSETS = (CP && !SEQF) ? 2 : 1;           // sets of LDS feature rows per wave
    constexpr int TAB_FLOATS = ALDS ? T::FWD_STEPS * 64 : 0;
    MNE_DYN_LDS(lds_raw);
    const int wpb = blockDim.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0 && !a.ray_list) {   // counters of this call, reset before any consumer runs
        if (a.tape_rows) *a.tape_rows = 0;
        if (a.bins.spill_count) *a.bins.spill_count = 0;
        if (a.defer_count) *a.defer_count = 0;
        if (a.long_count) *a.long_count = 0;
        if (a.heavy_count) *a.heavy_count = 0;
    }
    if (a.ray_list && *a.ray_list_count == 0) return;      // second pass with nothing deferred (the usual case)
    if (ALDS) {                                            // stage the A tables: the only block-wide step
        float4* dst = (float4*)lds_raw;
        const float4* src = (const float4*)a.packed;
        for (int i = threadIdx.x; i < TAB_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const float* atab = ALDS ? (const float*)lds_raw : a.packed;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ntile = (a.S + TILE - 1) / TILE;
    float* pn = (float*)(lds_raw + (size_t)TAB_FLOATS * sizeof(float) + (size_t)wv * tile_wave_lds_bytes(NSETS));
    float* feat = pn + TILE * 4;
    // persistent waves over the (tile, ray) tasks, TILE-major: the a-priori tiles of all rays come first, so the
    // skipped tasks (tiles beyond a ray's prefix) cluster at the end and the real ones spread evenly over the waves.
    // List mode (second pass): the REMAINING tiles of the rays the training kernel deferred.
    const int n_rays = a.ray_list ? *a.ray_list_count : a.R;
    // BALANCED schedule (sched != 0; first pass of a depth-guided batch in the prefix schedule): about half of the (tile, ray)
    // slots are tiles beyond their ray's a-priori prefix, and with a fixed stride over the slots some waves get four real
    // tiles and others two -- the launch takes as long as its busiest wave (70 us for 2.4 tiles of 12.6 us per wave on
    // average, profiles/r05_decode_balance.txt).  The exclusive prefix of the rays' tile counts (slot MNE_C_TILE0 of ray_counts, left
    // by the batch preparation) is staged in LDS and wave w decodes the real tiles w, w + W, ...: tile g belongs to the ray
    // found by a binary search, every wave gets the same number of tiles +- 1.  (A device-side queue over the slots was
    // measured first: 8600 returning atomics on one address made the launch 126 us instead of 70.)
    bool balanced = sched != 0 && !a.ray_list && a.ray_counts && !(a.adapt && a.adapt[0]);
    int* tstart = (int*)(lds_raw + (size_t)TAB_FLOATS * sizeof(float) + (size_t)wpb * tile_wave_lds_bytes(NSETS));
    long long ntask_l = (long long)n_rays * ntile;
    if (balanced) {
        for (int r = threadIdx.x; r < a.R; r += blockDim.x) tstart[r] = a.ray_counts[(size_t)r * MNE_N_COUNT + MNE_C_TILE0];
        __syncthreads();
        // The prefix must be THIS call's: written by mne_sample_z / mne_sample_batch for exactly these R rays, starting at 0.
        // A caller that renders a sub-range of a larger batch (pointer offset into ray_counts) or fills the counts itself has
        // another start or total: the fixed-stride schedule then (workgroup-uniform, every thread reads the same words).
        const long long total = (long long)tstart[a.R - 1] + apriori_tiles(a, a.R - 1, ntile);
        if (tstart[0] != 0 || total < a.R || total > (long long)a.R * ntile) balanced = false;
        else ntask_l = total;
    }
    // Task numbering.  The launch is bound by what a SIMD issues -- 7.2 us per tile and SIMD whether two or three waves share it
    // (profiles/r06_decode_waves.txt) -- and workgroup-major numbering (wave w of workgroup b: tasks b * wpb + w, ...) puts the last,
    // partial round on the first workgroups only (office0: 4952 tiles on 2048 waves, the 856 third tasks on workgroups 0..106).  Wave-major
    // numbering (w * gridDim + b) spreads it over all CUs and SIMDs: first-pass decode 60 -> 57 us, ray pass 74 -> 72 us on office0,
    // 94 -> 86 / 100 -> 93 us on ScanNet -- and the deferred pass behind them, beside the list appends on the other stream, loses the same
    // time (same-box A/B, profiles/r06_wave_major.txt: office0 +-0, INS Indoor -1..2 %).  It is used where no append kernel runs beside the
    // ray pass: the hash-grid iteration (caller-supplied features), +1.5 %.
    for (long long task = (MNE_WAVE_MAJOR && !a.ray_list && a.ext_feat) ? (long long)wv * gridDim.x + xcd_block() : (long long)xcd_block() * wpb + wv; task < ntask_l;
         task += (long long)gridDim.x * wpb) {
        int c, r;
        if (balanced) {
            int lo = 0, hi = a.R - 1;                              // last ray whose first tile is <= task
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (tstart[mid] <= (int)task) lo = mid; else hi = mid - 1;
            }
            r = lo; c = (int)task - tstart[lo];
        } else {
            c = (int)(task / n_rays);
            r = a.ray_list ? a.ray_list[(int)(task % n_rays)] : (int)(task % n_rays);
            if (a.ray_list ? (c < (a.dec_tiles ? a.dec_tiles[r] : prefix_tiles(a, r, ntile))) : (c >= prefix_tiles(a, r, ntile))) continue;
        }
        float pnv[3], u[3];
        uint2 relu;
        // The wave that decodes the LAST a-priori tile of a ray extends the prefix on the spot while the ray is visibly
        // unresolved: no sign change from this tile on, or the truncation window behind the sign change runs past the
        // tile.  (A heuristic that saves a second pass: ray_kernel checks the whole prefix exactly and defers whatever
        // is still unresolved.)  dec_tiles[r] = tiles decoded in the end.  One decode_tile call site for both uses.
        const bool resolver = a.dec_tiles && a.ray_counts && !a.ray_list && !a.tile_need && c == prefix_tiles(a, r, ntile) - 1;
        int cc = c;
        bool found = false, have_carry = false, pre_now = pre != 0 || a.ext_feat != 0;   // ext_feat: every row's features are the caller's
        float z_lim = 0.0f, s_carry = 0.0f;
        const float* zr = a.z_vals + (size_t)r * a.S;
        while (true) {
            const float4 rw = decode_tile<HID, HIDC, CP, !ALDS, SEQF, MNE_DECODE_WIDE_GATHER && HID == 32 && HIDC == 32 && !CP>(a, r, cc, lane, pn, feat, atab, pnv, u, relu, pre_now);
            if (!resolver) break;
            const int i0 = cc * TILE, n_in = a.S - i0 < TILE ? a.S - i0 : TILE;
            const float s_me = rw.w;                                   // valid on lanes < 32 (rows 0..3 of the result)
            const float s_nx = __shfl_down(s_me, 1);
            if (!found) {
                const bool cr = lane < 32 && (lane & 31) + 1 < n_in && s_me * s_nx < 0.0f;
                const unsigned long long m = __ballot(cr);
                int f = m ? i0 + __ffsll(m) - 1 : -1;
                if (have_carry && s_carry * __shfl(s_me, 0) < 0.0f) f = i0 - 1;       // pair across the tile boundary
                if (f >= 0) { found = true; z_lim = zr[f] + a.win_f; }
            }
            if (cc + 1 >= ntile) break;
            if (found && !(zr[i0 + n_in] < z_lim)) break;              // the window ends before the next tile
            if (cc - c >= (a.ext_feat ? MNE_RESOLVER_MAX_EXT_FEAT : ResolverExt<CP>::MAX)) break;                 // a long unresolved stretch is cheaper tile-parallel (deferred pass)
            s_carry = __shfl(s_me, n_in - 1); have_carry = true;
            ++cc;
            pre_now = a.ext_feat != 0;                                 // tiles beyond the prefix were not pre-gathered
        }
        if (resolver && lane == 0) a.dec_tiles[r] = cc + 1;
    }
}

// -----------------------------------------------------------------------------------------------
// decode_frame_kernel: the forward-only decode of whole frames (render_img) -- decode_kernel without its tape, ReLU-mask,
// resolver and pre-gathered-row paths, for 2x32 decoders without colour planes (what the frame renders of the default scene use).
//
// What bounds decode_kernel on a frame (profiles/r05_render_img_experiments.txt, r05_render_sq_counters.txt): per 32-sample tile
// a wave issues 120 dependent MFMAs (64 cycles each), ~1750 VALU and ~270 LDS instructions and waits 54 % of its cycles; the
// matrix pipe is 43 % busy at two waves per SIMD.  Not HBM: a Z-order ray walk cut the plane fetches 4x and changed nothing.
// Overlapping the chain of tile k with the gather of tile k + 1 inside ONE wave was built and measured (patches/
// r05_decode_pipe_kernel.patch): nothing at equal occupancy -- fillers between the MFMAs of one accumulator chain cost the chain
// its back-to-back issue.  What does pay is a third wave per SIMD, and that is a matter of registers: without the training-side
// code the tile needs 124 instead of 189, so 12 waves per CU (LDS: 30.7 KB of tables + 12 x 9.2 KB) instead of 8: -14 % per
// launch; 14 waves (the LDS limit; uneven over the SIMDs) are slower again.
// Same operations on the same values in the same order as decode_tile: bit-equal results.
// -----------------------------------------------------------------------------------------------
#ifndef MNE_DECODE_FRAME
#define MNE_DECODE_FRAME 1
#endif
#ifndef MNE_FRAME_WPB
#define MNE_FRAME_WPB 12        // waves per workgroup (= the register budget: 512 / ceil(waves / 4) = 168 per lane)
#endif
#ifndef MNE_FRAME_MIN_TILES
#define MNE_FRAME_MIN_TILES 8   // launches with fewer rays per wave than this stay with decode_kernel (balanced schedule, fewer waves to fill)
#endif

struct FrameTile {
    float4 x[8];            // this lane's half of its point's feature row
    float pos[24];
    f32x16 acc, h, out, hc;
};

// MFMA number M (0..119) of the forward chain of mlp_forward_sdf + mlp_forward_color for <32, 32, no colour planes>
template <int M>
__device__ __forceinline__ void frame_mfma(FrameTile& s, const ATabRef<false>& A) {
    typedef ATab<32, 32, false> T;
    if constexpr (M == 0 || M == 56 || M == 72 || M == 104) s.acc = f32x16_zero();
    if constexpr (M < 32) {
        const float4 q = s.x[M / 4];
        s.acc = MNE_MFMA(A.at(T::OFF_L1 + M), (M % 4 == 0 ? q.x : M % 4 == 1 ? q.y : M % 4 == 2 ? q.z : q.w), s.acc);
    } else if constexpr (M < 56) {
        s.acc = MNE_MFMA(A.at(T::OFF_L1 + M), s.pos[M - 32], s.acc);
        if constexpr (M == 55) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s.h[e] = fmaxf(s.acc[e], 0.0f);
        }
    } else if constexpr (M < 72) {
        s.acc = MNE_MFMA(A.at(T::OFF_L2 + (M - 56)), s.h[M - 56], s.acc);
        if constexpr (M == 71) s.out = s.acc;
    } else if constexpr (M < 96) {
        s.acc = MNE_MFMA(A.at(T::OFF_C1 + (M - 72)), s.pos[M - 72], s.acc);
    } else if constexpr (M < 104) {
        s.acc = MNE_MFMA(A.at(T::OFF_C1 + 24 + (M - 96)), s.out[M - 96], s.acc);
        if constexpr (M == 103) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s.hc[e] = fmaxf(s.acc[e], 0.0f);
        }
    } else {
        s.acc = MNE_MFMA(A.at(T::OFF_C2 + (M - 104)), s.hc[M - 104], s.acc);
    }
}
template <int M0, int M1>
__device__ __forceinline__ void frame_mfma_range(FrameTile& s, const ATabRef<false>& A) {
    if constexpr (M0 < M1) {
        frame_mfma<M0>(s, A);
        frame_mfma_range<M0 + 1, M1>(s, A);
    }
}

// One 32-sample tile of ray r, forward only, by the calling wave: normalised points -> inline gather into the wave's LDS rows -> OneBlob
// -> the bare MFMA chain.  Returns (r, g, b, sdf) of this lane's sample (rows 0..3 of the colour result and row 0 of the sdf result
// live in the lower half-wave) and stores it to a.raw.
__device__ __forceinline__ float4 frame_tile(const RenderArgs& a, int r, int c, int lane, float* pn, float* feat, const ATabRef<false>& A,
                                             FrameTile& st) {
    const int S = a.S, pt = lane & 31, hf = lane >> 5;
    const int i = c * TILE + pt;
    const float z = a.z_vals[(size_t)r * S + (i < S ? i : S - 1)];
    float p[3], pnv[3], u[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = a.rays_o[r * 3 + k] + a.rays_d[r * 3 + k] * z;      // scene_rep.py:384
    point_coords(a.sc, p, pnv, u);
    MNE_WAVE_SYNC();                                       // the previous tile's LDS reads are done
    *(float4*)(pn + pt * 4) = make_float4(pnv[0], pnv[1], pnv[2], 0.0f);       // (both lanes of a point store the same value)
    MNE_WAVE_SYNC();
    gather_chunk<1, TILE, MNE_INLINE_GATHER_NLV, 0, true>(a.sc, pn, feat, lane);
    MNE_WAVE_SYNC();
    oneblob_half<true>(u, hf, st.pos);
    const float* frow = feat + pt * MNE_FS + hf * 32;
#pragma unroll
    for (int q = 0; q < 8; ++q) st.x[q] = *(const float4*)(frow + 4 * q);
    frame_mfma_range<0, 120>(st, A);
    const float4 rw = make_float4(st.acc[0], st.acc[1], st.acc[2], st.out[0]);
    if (i < S && hf == 0) *(float4*)(a.raw + ((size_t)r * S + i) * 4) = rw;
    return rw;
}

template <int WPB>
__global__ __launch_bounds__(64 * WPB) void decode_frame_kernel(RenderArgs a, int sched) {
    typedef ATab<32, 32, false> T;
    constexpr int TAB_FLOATS = T::FWD_STEPS * 64;
    MNE_DYN_LDS(lds_raw);
    const int wpb = blockDim.x >> 6;
    {
        float4* dst = (float4*)lds_raw;
        const float4* src = (const float4*)a.packed;
        for (int i = threadIdx.x; i < TAB_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const float* atab = (const float*)lds_raw;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int S = a.S, ntile = (S + TILE - 1) / TILE;
    float* pn = (float*)(lds_raw + (size_t)TAB_FLOATS * sizeof(float) + (size_t)wv * tile_wave_lds_bytes(1));
    float* feat = pn + TILE * 4;
    const int n_rays = a.R;
    bool balanced = sched != 0 && a.ray_counts && !(a.adapt && a.adapt[0]);
    int* tstart = (int*)(lds_raw + (size_t)TAB_FLOATS * sizeof(float) + (size_t)wpb * tile_wave_lds_bytes(1));
    long long ntask_l = (long long)n_rays * ntile;
    if (balanced) {
        for (int r = threadIdx.x; r < a.R; r += blockDim.x) tstart[r] = a.ray_counts[(size_t)r * MNE_N_COUNT + MNE_C_TILE0];
        __syncthreads();
        const long long total = (long long)tstart[a.R - 1] + apriori_tiles(a, a.R - 1, ntile);       // (a foreign prefix: see decode_kernel)
        if (tstart[0] != 0 || total < a.R || total > (long long)a.R * ntile) balanced = false;
        else ntask_l = total;
    }
    const ATabRef<false> A(atab, lane);
    FrameTile st;
    // decode_kernel's two schedules over the (tile, ray) tasks: tiles of the rays' a-priori prefixes only
    for (long long task = (long long)xcd_block() * wpb + wv; task < ntask_l; task += (long long)gridDim.x * wpb) {
        int c, r;
        if (balanced) {
            int lo = 0, hi = a.R - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (tstart[mid] <= (int)task) lo = mid; else hi = mid - 1;
            }
            r = lo; c = (int)task - tstart[lo];
        } else {
            c = (int)(task / n_rays);
            r = (int)(task % n_rays);
            if (c >= prefix_tiles(a, r, ntile)) continue;
        }
        frame_tile(a, r, c, lane, pn, feat, A, st);
    }
}

// -----------------------------------------------------------------------------------------------
// compositing of one ray (wave-wide): raws[i] = (r,g,b,sdf) of the first D samples in LDS
// -----------------------------------------------------------------------------------------------
struct RayGrad { float denom, z_lim, Aq, g_rgb[3], g_dep; };

// first adjacent sign change among samples [from, D) (pairs (i, i+1), both < D); -1 when none
__device__ __forceinline__ int first_crossing(const float* raws, int from, int D, int lane) {
    for (int base = from; base < D - 1; base += MNE_WAVE) {
        const int i = base + lane;
        const bool cr = (i < D - 1) && (raws[4 * (i + 1) + 3] * raws[4 * i + 3] < 0.0f);
        const unsigned long long m = __ballot(cr);
        if (m) return base + __ffsll(m) - 1;
    }
    return -1;
}

// The two rules the training ray kernel and bin_kernel must agree on, sample for sample:
// (1) the decoded prefix [0, Dn) RESOLVES a ray when the first sign change is known and the render window behind it ends
//     inside the prefix (z sorted), or when everything is decoded;
__device__ __forceinline__ bool ray_resolved(const float* zr, int first, int Dn, int S, float win_f) {
    return Dn >= S || (first >= 0 && !(zr[Dn] < zr[first] + win_f));
}
// (2) a sample receives gradient when it lies inside the render window or an active loss mask selects it (exact).
__device__ __forceinline__ bool sample_contrib(float z, float z_lim, const SampleMasks& mk, bool use_e, bool use_co) {
    return (z < z_lim) || (use_e && (mk.e_front || mk.e_center || mk.e_tail)) || (use_co && (mk.co_fs || mk.co_sdf));
}

// Maps, loss partial sums and (WITH_GRAD) the per-ray constants of the gradient.  `first` = index of the first sign
// change (0 when the whole ray has none, scene_rep.py:195-199); all samples with z < z_lim are among the first D.
template <bool WITH_GRAD>
__device__ __forceinline__ void composite_ray(const RenderArgs& a, int r, int lane, const float* raws, const float* zr, int D,
                                              int first, RayGrad& G) {
    const bool has_t = a.target_d != nullptr;
    const float td = has_t ? a.target_d[r] : 0.0f;
    const float z_lim = zr[first] + a.win_f;                                // scene_rep.py:200
    float wsum = 0.0f;
    for (int i = lane; i < D; i += MNE_WAVE) {
        const float s = raws[4 * i + 3];
        const float wt = sigmoidf_(s / a.trunc_f) * sigmoidf_(-s / a.trunc_f);
        wsum += (zr[i] < z_lim) ? wt : 0.0f;
    }
    wsum = wave_sum(wsum);
    const float denom = wsum + 1e-8f;                                       // scene_rep.py:203
    float m_rgb[3] = {0.f, 0.f, 0.f}, m_depth = 0.f, m_acc = 0.f;
    float l_efs = 0.f, l_ec = 0.f, l_et = 0.f, l_cofs = 0.f, l_cosdf = 0.f;
    for (int i = lane; i < D; i += MNE_WAVE) {
        const float4 rw = *(const float4*)(raws + 4 * i);
        const float s = rw.w, z = zr[i];
        const float wt = sigmoidf_(s / a.trunc_f) * sigmoidf_(-s / a.trunc_f);
        const float w = ((z < z_lim) ? wt : 0.0f) / denom;
        m_rgb[0] += w * sigmoidf_(rw.x); m_rgb[1] += w * sigmoidf_(rw.y); m_rgb[2] += w * sigmoidf_(rw.z);
        m_depth += w * z;
        m_acc += w;
        const SampleMasks mk = sample_masks(z, td, has_t, a);
        const float sm1 = s - 1.0f;
        const float e_res = (z + s * a.e_T) - td;                          // scene_rep.py:503-507
        const float c_res = (z + s * a.win_f) - td;                        // model/utils.py:177
        l_efs += mk.e_front ? sm1 * sm1 : 0.f;
        l_ec += mk.e_center ? e_res * e_res : 0.f;
        l_et += mk.e_tail ? e_res * e_res : 0.f;
        l_cofs += mk.co_fs ? sm1 * sm1 : 0.f;
        l_cosdf += mk.co_sdf ? c_res * c_res : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) m_rgb[k] = wave_sum(m_rgb[k]);
    m_depth = wave_sum(m_depth);
    m_acc = wave_sum(m_acc);
    if (a.depth_var || a.disp) {
        float var = 0.f;
        for (int i = lane; i < D; i += MNE_WAVE) {
            const float s = raws[4 * i + 3], z = zr[i];
            const float wt = sigmoidf_(s / a.trunc_f) * sigmoidf_(-s / a.trunc_f);
            const float w = ((z < z_lim) ? wt : 0.0f) / denom;
            const float dz = z - m_depth;
            var += w * (dz * dz);
        }
        var = wave_sum(var);
        if (lane == 0) {
            if (a.depth_var) a.depth_var[r] = var;
            if (a.disp) a.disp[r] = 1.0f / fmaxf(1e-10f, m_depth / m_acc);   // scene_rep.py:224
        }
    }
    if (lane == 0) {
        if (a.rgb) { a.rgb[r * 3 + 0] = m_rgb[0]; a.rgb[r * 3 + 1] = m_rgb[1]; a.rgb[r * 3 + 2] = m_rgb[2]; }
        if (a.depth) a.depth[r] = m_depth;
        if (a.acc) a.acc[r] = m_acc;
    }
    const bool valid_ray = has_t && td > 0.0f && td < a.depth_trunc;
    float trgb[3] = {0.f, 0.f, 0.f};
    if (a.target_rgb) { trgb[0] = a.target_rgb[r * 3 + 0]; trgb[1] = a.target_rgb[r * 3 + 1]; trgb[2] = a.target_rgb[r * 3 + 2]; }
    if (a.ray_sums) {
        l_efs = wave_sum(l_efs); l_ec = wave_sum(l_ec); l_et = wave_sum(l_et);
        l_cofs = wave_sum(l_cofs); l_cosdf = wave_sum(l_cosdf);
        if (lane == 0) {
            float* rs = a.ray_sums + (size_t)r * MNE_N_LOSS;
            const float e0 = m_rgb[0] - trgb[0], e1 = m_rgb[1] - trgb[1], e2 = m_rgb[2] - trgb[2];
            const float ed = m_depth - td;
            rs[MNE_L_RGB] = e0 * e0 + e1 * e1 + e2 * e2;
            rs[MNE_L_DEPTH] = valid_ray ? ed * ed : 0.0f;
            rs[MNE_L_CO_SDF] = l_cosdf; rs[MNE_L_CO_FS] = l_cofs;
            rs[MNE_L_E_FS] = l_efs; rs[MNE_L_E_CENTER] = l_ec; rs[MNE_L_E_TAIL] = l_et;
            rs[MNE_L_PSNR] = 0.0f;
        }
    }
    if (WITH_GRAD) {
        float cf_rgb = a.coef ? a.coef[MNE_L_RGB] : 0.0f, cf_dep = a.coef ? a.coef[MNE_L_DEPTH] : 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            G.g_rgb[k] = (a.target_rgb ? cf_rgb * (m_rgb[k] - trgb[k]) : 0.0f) + (a.g_rgb ? a.g_rgb[r * 3 + k] : 0.0f);
        G.g_dep = (valid_ray ? cf_dep * (m_depth - td) : 0.0f) + (a.g_depth ? a.g_depth[r] : 0.0f);
        G.Aq = G.g_rgb[0] * m_rgb[0] + G.g_rgb[1] * m_rgb[1] + G.g_rgb[2] * m_rgb[2] + G.g_dep * m_depth;
        G.denom = denom; G.z_lim = z_lim;
    }
}

// all samples decoded: 4 rays per workgroup, raws staged in LDS (forward calls that must return raw)
__global__ __launch_bounds__(256) void composite_kernel(RenderArgs a) {
    MNE_DYN_LDS(lds_raw);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= a.R) return;
    const int S = a.S;
    float* raws = (float*)lds_raw + (size_t)wv * ((S + 3) & ~3) * 4;
    const float4* src = (const float4*)(a.raw_in + (size_t)r * S * 4);
    for (int i = lane; i < S; i += MNE_WAVE) *(float4*)(raws + 4 * i) = src[i];
    MNE_WAVE_SYNC();
    const int f = first_crossing(raws, 0, S, lane);
    RayGrad G;
    composite_ray<false>(a, r, lane, raws, a.z_vals + (size_t)r * S, S, f < 0 ? 0 : f, G);
}

// -----------------------------------------------------------------------------------------------
// ray_kernel: resolve (on-demand decode) -> composite -> backward of the ray's own tiles
//   MODE 0: forward only (maps; early ray termination)
//        1: training iteration (raw / masks / tape of every tile it touches exist: written by decode_kernel or by its own
//           resolve step)
//        2: backward of an EARLIER forward call (raw_in complete; forward tape rows of tiles beyond the a-priori prefix
//           are produced on demand inside the backward loop)        3: = 2 with ray gradients (R13)
// -----------------------------------------------------------------------------------------------
#ifndef MAX_WPB_RAY
#define MAX_WPB_RAY 8
#endif
#ifndef MAX_WPB_HOT
#define MAX_WPB_HOT 12
#endif

// Run grouping of the list appends.  The samples of a tile lie on ONE ray in order, and a straight line visits every
// cell block of an axis-aligned lattice in one contiguous stretch -- so lanes that append to the same list are
// CONSECUTIVE lanes.  A run start is a lane whose list differs from its predecessor's (lanes 0 and 32 always: the two
// halves handle different planes); leader, rank and length of a lane's run then follow from one ballot with bit
// arithmetic, in constant time whatever the number of distinct lists (the earlier leader-election loop cost one
// ballot round per distinct list: 11 us for a tile of 32 spread-out samples).
__device__ __forceinline__ unsigned run_meta(int w_, int lane) {
    const int prev = __shfl_up(w_, 1);
    const bool start = ((lane & 31) == 0) || (w_ != prev);
    const unsigned long long sm = __ballot(start);
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);      // bits 0..lane
    const int leader = 63 - __clzll(sm & upto);
    const unsigned long long above = sm & ~upto;
    const int next = above ? __ffsll(above) - 1 : 64;
    return (unsigned)leader | ((unsigned)(lane - leader) << 8) | ((unsigned)(next - leader) << 16);
}

// MODE 0: forward only (maps; early ray termination)
//      1: (unused)
//      2: backward with every capability: resolves rays by decoding on demand, produces missing forward tape rows inside
//         the backward loop.  Used for the backward of an EARLIER forward call (raw_in complete) and, driven by a.ray_list,
//         for the rays the training kernel deferred.            3: = 2 with ray gradients (R13)
//      4: TRAINING kernel (the hot one): no decode code at all -- 12 waves per CU instead of 8, every ray of the batch
//         resident at once.  A ray whose decoded prefix does not resolve it (no sign change yet, or the render window
//         runs past the prefix) is pushed to a.defer_list and handled afterwards by a MODE 2 launch.
#ifndef MNE_HOT64CP_WPB
#define MNE_HOT64CP_WPB 8       // 2x64 + colour planes: 46 KiB of backward tables in LDS leave room for 9 waves; 8 = 256 registers
#endif
// waves per workgroup a ray_kernel instantiation is compiled for (= its register budget: 512 / ceil(waves / 4) per lane)
// (MODE 3, ray gradients: 4 waves = the whole register file of a SIMD per wave -- its extra d(OneBlob) / d(coordinate) stages
// spilled 0.4-6 KiB per lane at 8 waves; it serves the 100-iteration pose loops of loop closure, not the mapping iteration)
// (MODE 2, the autograd path's backward: the training kernel's tile body and tables since round 6, so its shape as well)
#define RAY_WPB(HID, CP, MODE) (((MODE) == 4 || (MODE) == 2) ? (((HID) == 64 && (CP)) ? MNE_HOT64CP_WPB : MAX_WPB_HOT) : (MODE) == 3 ? 4 : MAX_WPB_RAY)

// One 32-sample tile of the TRAINING backward of ray r (tile c, samples [32 c, 32 c + 32) of the ray's Dn decoded ones): loss
// and compositing gradients of every sample from the ray's constants G -> MFMA backward chain from the saved ReLU masks ->
// backward half of the tape rows (d(pre-activations) for the decoder's weight gradients, d(feature) rows for the plane / table
// update).  zsrc / rawsrc: the ray's z samples and raw (r,g,b,sdf) rows -- the wave's LDS copies in ray_kernel, global memory
// in heavy_bwd_kernel.  pn / feat: the wave's LDS rows; atab: the backward tables (step BIAS onwards).
template <int HID, int HIDC, bool CP, int BIAS, bool GTAB>
__device__ __forceinline__ void hot_backward_tile(const RenderArgs& a, int r, int c, int Dn, const RayGrad& G, float td,
                                                  const float (&ro)[3], const float (&rd)[3], const float (&cf)[MNE_N_LOSS],
                                                  bool use_e, bool use_co, const float* zsrc, const float* rawsrc, int lane,
                                                  float* pn, float* feat, const float* atab) {
    typedef DecDims<HID, HIDC, CP> D;
    constexpr int NT = HID / 32, NTC = HIDC / 32;
    constexpr bool SEQ = CP, EARLY_DHC = HID == 64;       // see ray_kernel: one set of LDS rows / 32 registers less across the sdf chain
    const int S = a.S, pt = lane & 31, hf = lane >> 5;
    const bool has_t = a.target_d != nullptr;
    const int i = c * TILE + pt;
    const bool valid = i < Dn;
    const int ii = valid ? i : Dn - 1;
    const float z = zsrc[ii];
    float p[3], pnv[3], u[3];
    const size_t e = (size_t)r * S + ii;
#pragma unroll
    for (int q = 0; q < 3; ++q) p[q] = ro[q] + rd[q] * z;
    point_coords(a.sc, p, pnv, u);
    const uint2 mk2 = *(const uint2*)(a.relu_mask + e * 4 + hf * 2);
    // ---- d(total)/d(raw) of this point (both lanes of the pair compute the same values)
    const float4 rw = *(const float4*)(rawsrc + 4 * ii);
    const float s = rw.w;
    float ds = 0.0f, dc[3] = {0.f, 0.f, 0.f};
    bool contrib = false;
    if (valid) {
        if (z < G.z_lim) {
            const float pp = sigmoidf_(s / a.trunc_f), qq = sigmoidf_(-s / a.trunc_f);
            const float wt = pp * qq;
            const float w = wt / G.denom;
            const float sg[3] = {sigmoidf_(rw.x), sigmoidf_(rw.y), sigmoidf_(rw.z)};
            const float dLdw = G.g_rgb[0] * sg[0] + G.g_rgb[1] * sg[1] + G.g_rgb[2] * sg[2] + G.g_dep * z;
            ds += ((dLdw - G.Aq) / G.denom) * (wt * (qq - pp) / a.trunc_f);
#pragma unroll
            for (int q = 0; q < 3; ++q) dc[q] = G.g_rgb[q] * w * (sg[q] * (1.0f - sg[q]));
        }
        const SampleMasks mk = sample_masks(z, td, has_t, a);
        const float e_res = (z + s * a.e_T) - td, c_res = (z + s * a.win_f) - td;
        if (mk.e_front) ds += cf[MNE_L_E_FS] * (s - 1.0f);
        if (mk.e_center) ds += cf[MNE_L_E_CENTER] * e_res;
        if (mk.e_tail) ds += cf[MNE_L_E_TAIL] * e_res;
        if (mk.co_fs) ds += cf[MNE_L_CO_FS] * (s - 1.0f);
        if (mk.co_sdf) ds += cf[MNE_L_CO_SDF] * c_res;
        contrib = sample_contrib(z, G.z_lim, mk, use_e, use_co);
    }
    MNE_WAVE_SYNC();                                      // feat rows are about to be overwritten
    // ---- MFMA backward chain from the saved ReLU masks; d(feature) rows land in this point's LDS rows.
    // A sample without gradient has ds = dc = 0 and therefore an all-zero backward row.
    float* frow = feat + pt * MNE_FS;
    f32x16 dh[NT], dout, dhc[NTC];
    const bool live = valid && contrib;
    const unsigned live_rows = (unsigned)__ballot(live && hf == 0), valid_rows = (unsigned)__ballot(valid && hf == 0);
    float* tape0 = a.tape + ((size_t)r * S + (size_t)c * TILE) * D::ROW;
    if (hf == 0) *(float4*)(pn + pt * 4) = make_float4(pnv[0], pnv[1], pnv[2], 0.0f);
    if (SEQ || EARLY_DHC) {
        // colour net first; what it produces leaves for the tape before the sdf net's chain starts, so that its registers
        // (and, with colour planes, its LDS rows) are free again
        mlp_backward_color<HID, HIDC, CP, BIAS, GTAB>(mk2.y, ds, dc, atab, lane, dout, dhc, frow);
        MNE_WAVE_SYNC();
        if (SEQ) {
            if (a.ext_feat) store_rows<MNE_FEAT>(feat, tape0, D::ROW, D::T_DFEAT + MNE_FEAT, valid_rows, lane);
            else if (a.plane_grads) store_rows<MNE_FEAT>(feat, tape0, D::ROW, D::T_DFEAT + MNE_FEAT, live_rows, lane);
            MNE_WAVE_SYNC();
        }
        if (EARLY_DHC) {
#pragma unroll
            for (int t = 0; t < NTC; ++t) acc_to_row(frow, 32 * t, dhc[t], hf);
            MNE_WAVE_SYNC();
            store_rows<HIDC>(feat, tape0, D::ROW, D::T_DHC, valid_rows, lane);
            MNE_WAVE_SYNC();
        }
        mlp_backward_sdf<HID, HIDC, CP, BIAS, GTAB>(mk2.x, atab, lane, dh, dout, frow);
    } else {
        mlp_backward_mfma<HID, HIDC, CP, BIAS, GTAB>(mk2.x, mk2.y, ds, dc, atab, lane, dh, dout, dhc, frow, frow);
    }
    MNE_WAVE_SYNC();
    // ---- backward half of the tape rows, staged through the LDS rows (full-line stores, see store_rows)
    if (a.ext_feat) store_rows<MNE_FEAT>(feat, tape0, D::ROW, D::T_DFEAT, valid_rows, lane);         // caller-owned encoding: every valid sample
    else if (a.plane_grads) store_rows<MNE_FEAT>(feat, tape0, D::ROW, D::T_DFEAT, live_rows, lane);   // the samples that receive gradient
    MNE_WAVE_SYNC();
    if (HID == 32 && HIDC == 32) {
        acc_to_row(frow, 0, dh[0], hf);
        acc_to_row(frow, 32, dhc[0], hf);
        MNE_WAVE_SYNC();
        store_rows<64>(feat, tape0, D::ROW, D::T_DH, valid_rows, lane);
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc_to_row(frow, 32 * t, dh[t], hf);
        MNE_WAVE_SYNC();
        store_rows<HID>(feat, tape0, D::ROW, D::T_DH, valid_rows, lane);
        if (!EARLY_DHC) {
            MNE_WAVE_SYNC();
#pragma unroll
            for (int t = 0; t < NTC; ++t) acc_to_row(frow, 32 * t, dhc[t], hf);
            MNE_WAVE_SYNC();
            store_rows<HIDC>(feat, tape0, D::ROW, D::T_DHC, valid_rows, lane);
        }
    }
    MNE_WAVE_SYNC();
    acc_to_row(frow, 0, dout, hf, 8);                     // [dout 16 | dc 4 | pn 4 | pad 8]
    if (hf == 0) {
        *(float4*)(frow + 16) = make_float4(dc[0], dc[1], dc[2], 0.0f);
        *(float4*)(frow + 20) = make_float4(pnv[0], pnv[1], pnv[2], live ? 1.0f : 0.0f);   // pn.w: the sample receives gradient (scatter_kernel)
    } else {
        *(float4*)(frow + 24) = make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)(frow + 28) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    MNE_WAVE_SYNC();
    store_rows<32>(feat, tape0, D::ROW, D::T_DOUT, valid_rows, lane);
}

// first adjacent sign change among samples [from, D) of a ray whose sdf values sit in `sdf` (LDS); -1 when none
__device__ __forceinline__ int first_crossing_s(const float* sdf, int from, int D, int lane) {
    for (int base = from; base < D - 1; base += MNE_WAVE) {
        const int i = base + lane;
        const bool cr = (i < D - 1) && (sdf[i + 1] * sdf[i] < 0.0f);
        const unsigned long long m = __ballot(cr);
        if (m) return base + __ffsll(m) - 1;
    }
    return -1;
}

// ray_frame_kernel: ray_kernel<32, 32, no colour planes, MODE 0> of whole frames -- resolve every ray from its a-priori prefix, decoding
// further tiles on demand (frame_tile), then composite.  What differs is the LDS: ray_kernel keeps a ray's raw rows and z samples in
// LDS (20 B per sample: 5 KB of the 14.3 KB per wave at 256 samples), which holds it at 8 waves per CU; here only the sdf values stay in
// LDS (the sign-change search), the compositing reads the raw rows back from global memory -- written by the decode launch and by this
// very wave (L1 is write-through and shared by the workgroup: a drain of the wave's stores is all the hand-off needs) -- and z from
// global memory: 10.2 KB per wave, 12 waves per CU.  Same values through the same compositing code: bit-equal maps.
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void ray_frame_kernel(RenderArgs a) {
    typedef ATab<32, 32, false> T;
    constexpr int TAB_FLOATS = T::FWD_STEPS * 64;
    MNE_DYN_LDS(lds_raw);
    const int wpb = blockDim.x >> 6;
    {
        float4* dst = (float4*)lds_raw;
        const float4* src = (const float4*)a.packed;
        for (int i = threadIdx.x; i < TAB_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const float* atab = (const float*)lds_raw;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pt = lane & 31, hf = lane >> 5;
    const int S = a.S, ntile = (S + TILE - 1) / TILE;
    const size_t wave_bytes = tile_wave_lds_bytes(1) + (size_t)((S + 3) & ~3) * sizeof(float);
    float* pn = (float*)(lds_raw + (size_t)TAB_FLOATS * sizeof(float) + (size_t)wv * wave_bytes);
    float* feat = pn + TILE * 4;
    float* sdf = feat + TILE * MNE_FS;                            // [Spad] this ray's sdf values
    const ATabRef<false> A(atab, lane);
    FrameTile st;
    for (int r = xcd_block() * wpb + wv; r < a.R; r += gridDim.x * wpb) {
        const float* zsrc = a.z_vals + (size_t)r * S;
        const float* rawsrc = a.raw + (size_t)r * S * 4;
        int t_dec = prefix_tiles(a, r, ntile);                   // tiles the decode launch wrote
        int Dn = t_dec * TILE < S ? t_dec * TILE : S;
        MNE_WAVE_SYNC();                                          // previous ray's LDS reads are done
        for (int i = lane; i < Dn; i += MNE_WAVE) sdf[i] = rawsrc[4 * i + 3];
        MNE_WAVE_SYNC();
        int first = -1, from = 0;
        while (true) {
            if (first < 0) first = first_crossing_s(sdf, from, Dn, lane);
            if (ray_resolved(zsrc, first, Dn, S, a.win_f)) break;
            const float4 rw = frame_tile(a, r, t_dec, lane, pn, feat, A, st);      // (Dn is a multiple of TILE here)
            const int i = t_dec * TILE + pt;
            if (i < S && hf == 0) sdf[i] = rw.w;
            MNE_WAVE_SYNC();
            from = Dn > 0 ? Dn - 1 : 0;
            ++t_dec;
            Dn = t_dec * TILE < S ? t_dec * TILE : S;
        }
        MNE_DRAIN_STORES();                                       // this wave's raw rows have left for L1 / L2 before its lanes read them back
        MNE_WAVE_SYNC();
        RayGrad G;
        composite_ray<false>(a, r, lane, rawsrc, zsrc, Dn, first < 0 ? 0 : first, G);
    }
}

template <int HID, int HIDC, bool CP, bool ALDS, int MODE>
__global__ __launch_bounds__(64 * RAY_WPB(HID, CP, MODE)) void ray_kernel(RenderArgs a) {
    typedef DecDims<HID, HIDC, CP> D;
    typedef ATab<HID, HIDC, CP> T;
    constexpr int NSETS = CP ? 2 : 1;
    constexpr int NT = HID / 32, NTC = HIDC / 32;
    constexpr bool HOT = MODE == 4, BWD = MODE >= 1, RAYGRAD = MODE == 3;
    // A tables in LDS: everything the mode needs; the backward modes stage only the backward steps (round 6: none of them
    // decodes any more -- tile_need_kernel + the decode launch make every tape row MODE 2 / 3 walk)
    constexpr int TAB_FIRST = BWD ? T::FWD_STEPS : 0;
    constexpr int TAB_LAST = RAYGRAD ? T::TOTAL_RAYGRAD : BWD ? T::TOTAL : T::FWD_STEPS;
    constexpr int TAB_FLOATS = ALDS ? (TAB_LAST - TAB_FIRST) * 64 : 0;
    MNE_DYN_LDS(lds_raw);
    const int wpb = blockDim.x >> 6;
    if (HOT && a.adapt && a.adapt_update && blockIdx.x == 0 && threadIdx.x == 0) {
        // Adaptive schedule (mne_fused_opts_t::adapt_state), decided by the LAST ray launch of a call for the NEXT call:
        // n = rays the a-priori prefix did not (mode 0: the deferred list) / would not (mode 1: counted below) resolve
        const int mode = a.adapt[0];
        const int n = mode ? a.adapt[1] : *a.ray_list_count;
        a.adapt[0] = mode ? (n * 16 >= a.R ? 1 : 0) : (n * 8 > a.R ? 1 : 0);
        a.adapt[1] = 0;
        a.adapt[2] = n;                       // (diagnostics: what the decision was taken on)
    }
    if (a.ray_list && *a.ray_list_count == 0) return;      // second pass with nothing deferred (the usual case)
    if (ALDS) {
        float4* dst = (float4*)lds_raw;
        const float4* src = (const float4*)(a.packed + TAB_FIRST * 64);
        for (int i = threadIdx.x; i < TAB_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const float* atab = ALDS ? (const float*)lds_raw : a.packed;      // ALDS: table step TAB_FIRST onwards (BIAS below)
    constexpr int BIAS = ALDS ? TAB_FIRST : 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int S = a.S, ntile = (S + TILE - 1) / TILE;
    const int L = (a.lds_samples > 0 && a.lds_samples < S) ? a.lds_samples : S;       // samples the per-wave LDS arrays hold
    const int Spad = (L + 3) & ~3;
    // Training kernel with colour planes: ONE set of 32 LDS rows per wave -- the colour net's d(feature) rows leave for
    // the tape before the sdf net's rows are produced -- so that more waves fit beside the tables (ScanNet: 6 -> 11 per CU,
    // i.e. all 2150 rays in one round instead of two).
    constexpr bool SEQ = HOT && CP;
    constexpr bool SEQF = MNE_SEQ_FORWARD && MODE == 0 && CP;       // forward only: one set of rows, the two nets' gathers in turn (decode_tile)
    // 2x64 decoders: d(hidden) of the colour net is written to the tape before the sdf net's backward (32 registers less
    // across its chain); not with ray gradients, which need both nets' d(hidden) for the OneBlob input gradient
    constexpr bool EARLY_DHC = HID == 64 && BWD && !RAYGRAD;
    constexpr int LSETS = (SEQ || SEQF) ? 1 : NSETS;
    const size_t wave_bytes = (size_t)Spad * 5 * sizeof(float) + tile_wave_lds_bytes(LSETS, RAYGRAD);
    unsigned char* my = lds_raw + (size_t)TAB_FLOATS * sizeof(float) + (size_t)wv * wave_bytes;
    float* raws = (float*)my;                                     // [Spad][4]
    float* zr = raws + (size_t)Spad * 4;                          // [Spad] this ray's z samples
    float* pn = zr + Spad;
    float* feat = pn + TILE * 4;
    float* dposL = feat + LSETS * TILE * MNE_FS;                  // RAYGRAD: [32][64] d OneBlob rows
    float* dpnL = dposL + TILE * 64;                              // RAYGRAD: [32][4]  d normalised point
    const int pt = lane & 31, hf = lane >> 5;
    const bool has_t = a.target_d != nullptr;
    float cf[MNE_N_LOSS];
#pragma unroll
    for (int q = 0; q < MNE_N_LOSS; ++q) cf[q] = (BWD && a.coef) ? a.coef[q] : 0.0f;
    const bool use_e = cf[MNE_L_E_FS] != 0.f || cf[MNE_L_E_CENTER] != 0.f || cf[MNE_L_E_TAIL] != 0.f;
    const bool use_co = cf[MNE_L_CO_FS] != 0.f || cf[MNE_L_CO_SDF] != 0.f;
    const int n_items = a.ray_list ? *a.ray_list_count : a.R;
    // (training with caller-supplied features -- see decode_kernel's task numbering: wave w of workgroup b takes rays w * gridDim + b, ...,
    // with launch_ray's grid of min(CUs, rays) every CU gets its share of a 2150-ray batch; forward frames keep consecutive rays -- Z-order
    // neighbours -- in one workgroup)
    for (int item = ((MODE == 4 && MNE_WAVE_MAJOR && !a.ray_list && a.ext_feat) ? wv * (int)gridDim.x + (int)blockIdx.x : (MODE == 0 ? xcd_block() : (int)blockIdx.x) * wpb + wv); item < n_items;
         item += gridDim.x * wpb) {
        const int r = a.ray_list ? a.ray_list[item] : item;
        const float td = has_t ? a.target_d[r] : 0.0f;
        float ro[3], rd[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { ro[q] = a.rays_o[r * 3 + q]; rd[q] = a.rays_d[r * 3 + q]; }
        MNE_WAVE_SYNC();                                          // previous ray's LDS reads are done
        // ---- this ray's z and the raw of its known samples -> LDS, one batch of loads (raw beyond the decoded prefix
        // is never looked at): everything (backward of an earlier forward) or the decoded prefix
        int t_dec = (a.dec_tiles && (!a.ray_list || a.list_keeps_prefix) && a.ray_counts)
                        ? a.dec_tiles[r] : prefix_tiles(a, r, ntile, a.ray_list != nullptr);   // tiles with raw / masks / tape rows written
        int Dn = a.raw_in ? S : (t_dec * TILE < S ? t_dec * TILE : S);
        if (L < S && Dn + 1 > L) {                                // (HOT only) the decoded prefix does not fit this launch's LDS:
            if (lane == 0) a.long_list[atomicAdd(a.long_count, 1)] = r;            // next pass: the same kernel sized for S
            continue;
        }
        {
            const float4* src = (const float4*)((a.raw_in ? a.raw_in : a.raw) + (size_t)r * S * 4);
            const float* zsrc = a.z_vals + (size_t)r * S;
            for (int i = lane; i < L; i += MNE_WAVE) {
                zr[i] = zsrc[i];
                if (i < Dn) *(float4*)(raws + 4 * i) = src[i];
            }
        }
        MNE_WAVE_SYNC();
        // ---- resolve: first sign change + every sample inside the render window must be known
        int first = -1, from = 0;
        bool deferred = false;
        while (true) {
            if (first < 0) first = first_crossing(raws, from, Dn, lane);
            if (ray_resolved(zr, first, Dn, S, a.win_f)) break;
            if (HOT) { deferred = true; break; }
            if constexpr (BWD) break;                             // MODE 2 / 3: raw of the forward call, all S samples: always resolved above
            if constexpr (MODE == 0) {
                // decode the next tile on demand (Dn is a multiple of TILE here)
                float pnv[3], u[3];
                uint2 relu;
                const float4 rw = decode_tile<HID, HIDC, CP, !ALDS, SEQF>(a, r, t_dec, lane, pn, feat, atab, pnv, u, relu, a.ext_feat != 0);
                const int i = t_dec * TILE + pt;
                if (i < S && hf == 0) *(float4*)(raws + 4 * i) = rw;
                MNE_WAVE_SYNC();
                from = Dn > 0 ? Dn - 1 : 0;
                ++t_dec;
                Dn = t_dec * TILE < S ? t_dec * TILE : S;
            }
        }
        if (HOT && deferred) {
            if (lane == 0) a.defer_list[atomicAdd(a.defer_count, 1)] = r;
            continue;
        }
        if (HOT && a.adapt && a.adapt[0] && (!a.ray_list || a.list_keeps_prefix)) {
            // mode 1 (every sample was decoded a priori): would the a-priori prefix plus the resolver's extension have
            // resolved this ray?  Feeds the decision to go back to mode 0.
            int t_ap = apriori_tiles(a, r, ntile) + (a.ext_feat ? MNE_RESOLVER_MAX_EXT_FEAT : ResolverExt<CP>::MAX);
            const int D_ap = t_ap * TILE < S ? t_ap * TILE : S;
            const bool resolved = D_ap >= S || (first >= 0 && first + 1 < D_ap && !(zr[D_ap] < zr[first] + a.win_f));
            if (!resolved && lane == 0) atomicAdd(a.adapt + 1, 1);
        }
        RayGrad G;
        composite_ray<BWD>(a, r, lane, raws, zr, Dn, first < 0 ? 0 : first, G);
        if (!BWD) continue;
        // ---- samples that can receive gradient: render window or an active loss mask (exact)
        int last = -1, n_contrib = 0;
        for (int base = 0; base < Dn; base += MNE_WAVE) {
            const int i = base + lane;
            bool f = false;
            if (i < Dn) {
                const float z = zr[i];
                f = sample_contrib(z, G.z_lim, sample_masks(z, td, has_t, a), use_e, use_co);
            }
            const unsigned long long m = __ballot(f);
            if (m) last = base + 63 - __clzll(m);
            n_contrib += __popcll(m);
        }
        const int nb = last < 0 ? 0 : last / TILE + 1;
        if (lane == 0) {
            if (a.ray_tiles) a.ray_tiles[r] = nb;
            if (a.tape_rows && n_contrib) atomicAdd(a.tape_rows, n_contrib);
        }
        if (HOT && a.heavy_min > 0 && nb > a.heavy_min) {
            // A ray with many backward tiles: its tiles go to heavy_bwd_kernel, tile-parallel and evenly dealt, instead of being
            // walked one after the other by this wave (INS Indoor: up to 33 tiles per ray; the launch was as long as its
            // longest ray).  What the tiles need of the ray: the gradient constants and the decoded sample count.
            if (lane == 0) {
                a.heavy_list[atomicAdd(a.heavy_count, 1)] = r;
                float4* rec = (float4*)(a.heavy_rec + (size_t)r * 8);
                rec[0] = make_float4(G.denom, G.z_lim, G.Aq, G.g_dep);
                rec[1] = make_float4(G.g_rgb[0], G.g_rgb[1], G.g_rgb[2], __uint_as_float((unsigned)Dn));
            }
            continue;
        }
        float ray_do[3] = {0.f, 0.f, 0.f}, ray_dd[3] = {0.f, 0.f, 0.f};
        for (int c = 0; c < nb; ++c) {
            if constexpr (HOT || MODE == 2) {                     // the training kernel's tile body (shared with heavy_bwd_kernel);
                // MODE 2 -- the autograd path's backward -- is the same tile without the ray-gradient stages below
                hot_backward_tile<HID, HIDC, CP, BIAS, !ALDS>(a, r, c, Dn, G, td, ro, rd, cf, use_e, use_co, zr, raws, lane, pn, feat, atab);
                continue;
            }
            const int i = c * TILE + pt;
            const bool valid = i < Dn;
            const int ii = valid ? i : Dn - 1;
            const float z = zr[ii];
            float p[3], pnv[3], u[3];
            const size_t e = (size_t)r * S + ii;
            uint2 mk2;
            // (MODE 2 / 3, the backward of an earlier forward call: the tape rows and ReLU masks of every tile walked here were
            // made by the decode launch -- tile_need_kernel told it which; until round 6 tiles beyond the a-priori prefix were
            // decoded HERE, which put the forward chain's registers on top of the backward's: 20-240 B of scratch per lane)
#pragma unroll
            for (int q = 0; q < 3; ++q) p[q] = ro[q] + rd[q] * z;
            point_coords(a.sc, p, pnv, u);
            mk2 = *(const uint2*)(a.relu_mask + e * 4 + hf * 2);
            // ---- d(total)/d(raw) of this point (both lanes of the pair compute the same values)
            const float4 rw = *(const float4*)(raws + 4 * ii);
            const float s = rw.w;
            float ds = 0.0f, dc[3] = {0.f, 0.f, 0.f};
            bool contrib = false;
            if (valid) {
                if (z < G.z_lim) {
                    const float pp = sigmoidf_(s / a.trunc_f), qq = sigmoidf_(-s / a.trunc_f);
                    const float wt = pp * qq;
                    const float w = wt / G.denom;
                    const float sg[3] = {sigmoidf_(rw.x), sigmoidf_(rw.y), sigmoidf_(rw.z)};
                    const float dLdw = G.g_rgb[0] * sg[0] + G.g_rgb[1] * sg[1] + G.g_rgb[2] * sg[2] + G.g_dep * z;
                    ds += ((dLdw - G.Aq) / G.denom) * (wt * (qq - pp) / a.trunc_f);
#pragma unroll
                    for (int q = 0; q < 3; ++q) dc[q] = G.g_rgb[q] * w * (sg[q] * (1.0f - sg[q]));
                    contrib = true;
                }
                const SampleMasks mk = sample_masks(z, td, has_t, a);
                const float e_res = (z + s * a.e_T) - td, c_res = (z + s * a.win_f) - td;
                if (mk.e_front) ds += cf[MNE_L_E_FS] * (s - 1.0f);
                if (mk.e_center) ds += cf[MNE_L_E_CENTER] * e_res;
                if (mk.e_tail) ds += cf[MNE_L_E_TAIL] * e_res;
                if (mk.co_fs) ds += cf[MNE_L_CO_FS] * (s - 1.0f);
                if (mk.co_sdf) ds += cf[MNE_L_CO_SDF] * c_res;
                contrib = sample_contrib(z, G.z_lim, mk, use_e, use_co);
            }
            MNE_WAVE_SYNC();                                      // feat rows are about to be overwritten
            // ---- MFMA backward chain from the saved ReLU masks; d(feature) rows land in this point's LDS rows.
            // A sample without gradient has ds = dc = 0 and therefore an all-zero backward row.
            float* frow = feat + pt * MNE_FS;
            float* cfrow = SEQ ? frow : feat + TILE * MNE_FS + pt * MNE_FS;
            f32x16 dh[NT], dout, dhc[NTC];
            const bool live = valid && contrib;
            const unsigned live_rows = (unsigned)__ballot(live && hf == 0), valid_rows = (unsigned)__ballot(valid && hf == 0);
            float* tape0 = a.tape + ((size_t)r * S + (size_t)c * TILE) * D::ROW;
            if (hf == 0) *(float4*)(pn + pt * 4) = make_float4(pnv[0], pnv[1], pnv[2], 0.0f);
            if (SEQ || EARLY_DHC) {
                // colour net first; what it produces leaves for the tape before the sdf net's chain starts, so that
                // its registers (and, with colour planes, its LDS rows) are free again
                mlp_backward_color<HID, HIDC, CP, BIAS, !ALDS>(mk2.y, ds, dc, atab, lane, dout, dhc, cfrow);
                MNE_WAVE_SYNC();
                if (SEQ) {
                    if (a.ext_feat) store_rows<MNE_FEAT>(feat, tape0, D::ROW, D::T_DFEAT + MNE_FEAT, valid_rows, lane);
                    else if (a.plane_grads) store_rows<MNE_FEAT>(feat, tape0, D::ROW, D::T_DFEAT + MNE_FEAT, live_rows, lane);
                    MNE_WAVE_SYNC();
                }
                if (EARLY_DHC) {
#pragma unroll
                    for (int t = 0; t < NTC; ++t) acc_to_row(frow, 32 * t, dhc[t], hf);
                    MNE_WAVE_SYNC();
                    store_rows<HIDC>(feat, tape0, D::ROW, D::T_DHC, valid_rows, lane);
                    MNE_WAVE_SYNC();
                }
                mlp_backward_sdf<HID, HIDC, CP, BIAS, !ALDS>(mk2.x, atab, lane, dh, dout, frow);
            } else {
                mlp_backward_mfma<HID, HIDC, CP, BIAS, !ALDS>(mk2.x, mk2.y, ds, dc, atab, lane, dh, dout, dhc, frow, cfrow);
            }
            if (RAYGRAD) {
                // d(total)/d(point) = through the OneBlob input + through the bilinear plane coordinates; every
                // point of the tile belongs to this ray: summed over the wave, stored once at the end
                float* dprow = dposL + pt * 64;
                mlp_backward_dpos<HID, HIDC, CP, BIAS, !ALDS>(dh, dhc, atab, lane, dprow);
                MNE_WAVE_SYNC();
                // (a caller-owned encoding -- hash / dense grid -- differentiates its own features: mne_hash_ray_grad adds
                // that part from the d(feature) rows this kernel leaves in the tape; here only the OneBlob input's share)
                if (!a.ext_feat) gather_coord_grad<NSETS, TILE>(a.sc, pn, feat, dpnL, lane);
                MNE_WAVE_SYNC();
                float du[3];
                oneblob_half_backward(u, hf, dprow, du);
#pragma unroll
                for (int q = 0; q < 3; ++q) du[q] += __shfl_xor(du[q], 32);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float inv_bb = a.sc.bb_is_f64 ? (float)(1.0 / (a.sc.bb_hi[q] - a.sc.bb_lo[q]))
                                                        : 1.0f / ((float)a.sc.bb_hi[q] - (float)a.sc.bb_lo[q]);
                    const float dplane = a.ext_feat ? 0.0f : dpnL[pt * 4 + q] * (2.0f / (a.sc.bound_hi[q] - a.sc.bound_lo[q]));
                    const float dp = (valid && contrib && hf == 0) ? dplane + du[q] * inv_bb : 0.0f;
                    ray_do[q] += dp;
                    ray_dd[q] += z * dp;
                }
            }
            MNE_WAVE_SYNC();
            const unsigned tape_rows_mask = valid_rows;
            // ---- backward half of the tape rows, staged through the LDS rows (full-line stores, see store_rows)
            if (a.ext_feat) {                                     // caller-owned encoding: d(feature) rows of every valid sample
#pragma unroll                                                    // (all-zero rows for samples without gradient)
                for (int set = 0; set < LSETS; ++set)
                    store_rows<MNE_FEAT>(feat + set * TILE * MNE_FS, tape0, D::ROW, D::T_DFEAT + set * MNE_FEAT, valid_rows, lane);
            } else if (a.plane_grads) {                           // d(feature) rows of the samples that receive gradient: read by the
#pragma unroll                                                    // binned plane update (tile_adam.hip) or by scatter_kernel (atomics);
                for (int set = 0; set < LSETS; ++set)             // 0 = the caller wants no plane gradients (pose-only loops)
                    store_rows<MNE_FEAT>(feat + set * TILE * MNE_FS, tape0, D::ROW, D::T_DFEAT + set * MNE_FEAT, live_rows, lane);
            }
            MNE_WAVE_SYNC();
            if (HID == 32 && HIDC == 32) {
                acc_to_row(frow, 0, dh[0], hf);
                acc_to_row(frow, 32, dhc[0], hf);
                MNE_WAVE_SYNC();
                store_rows<64>(feat, tape0, D::ROW, D::T_DH, tape_rows_mask, lane);
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc_to_row(frow, 32 * t, dh[t], hf);
                MNE_WAVE_SYNC();
                store_rows<HID>(feat, tape0, D::ROW, D::T_DH, tape_rows_mask, lane);
                if (!EARLY_DHC) {
                    MNE_WAVE_SYNC();
#pragma unroll
                    for (int t = 0; t < NTC; ++t) acc_to_row(frow, 32 * t, dhc[t], hf);
                    MNE_WAVE_SYNC();
                    store_rows<HIDC>(feat, tape0, D::ROW, D::T_DHC, tape_rows_mask, lane);
                }
            }
            MNE_WAVE_SYNC();
            acc_to_row(frow, 0, dout, hf, 8);                     // [dout 16 | dc 4 | pn 4 | pad 8]
            if (hf == 0) {
                *(float4*)(frow + 16) = make_float4(dc[0], dc[1], dc[2], 0.0f);
                *(float4*)(frow + 20) = make_float4(pnv[0], pnv[1], pnv[2], live ? 1.0f : 0.0f);   // pn.w: the sample receives gradient (bin_kernel)
            } else {
                *(float4*)(frow + 24) = make_float4(0.f, 0.f, 0.f, 0.f);
                *(float4*)(frow + 28) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            MNE_WAVE_SYNC();
            store_rows<32>(feat, tape0, D::ROW, D::T_DOUT, tape_rows_mask, lane);
            // (the list appends of the binned plane update are bin_kernel's: it reads pn and the `live` flag written above)
        }
        if (RAYGRAD) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { ray_do[q] = wave_sum(ray_do[q]); ray_dd[q] = wave_sum(ray_dd[q]); }
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (a.d_rays_o) a.d_rays_o[r * 3 + q] = ray_do[q];
                    if (a.d_rays_d) a.d_rays_d[r * 3 + q] = ray_dd[q];
                }
            }
        }
    }
}

// -----------------------------------------------------------------------------------------------
// heavy_bwd_kernel: the backward tiles of the rays the training kernel put on its heavy list (more than heavy_min tiles:
// long rays -- INS Indoor samples 1045 points per ray), TILE-parallel: the exclusive prefix of the listed rays' tile counts
// is built in LDS by every workgroup, wave w takes tiles w, w + W, ... and finds the ray of a tile by binary search (the
// schedule of decode_kernel).  One wave per ray made the ray launches as long as the longest ray's 10-30 tiles in sequence
// (Indoor: 134 + 153 + 81 us for the three ray launches of an iteration, profiles/r05_timeline_indoor.txt) while most waves
// had long finished; here every wave gets the same number of tiles, and no wave carries a ray's raw / z arrays in LDS, so
// twelve waves fit on a CU whatever S is.
// -----------------------------------------------------------------------------------------------
template <int HID, int HIDC, bool CP>
__global__ __launch_bounds__(64 * RAY_WPB(HID, CP, 4)) void heavy_bwd_kernel(RenderArgs a) {
    typedef ATab<HID, HIDC, CP> T;
    constexpr int TAB_FIRST = T::FWD_STEPS, TAB_FLOATS = (T::TOTAL - T::FWD_STEPS) * 64;
    MNE_DYN_LDS(lds_raw);
    const int n_heavy = *a.heavy_count;
    if (n_heavy == 0) return;
    {
        float4* dst = (float4*)lds_raw;
        const float4* src = (const float4*)(a.packed + TAB_FIRST * 64);
        for (int i = threadIdx.x; i < TAB_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
    }
    const float* atab = (const float*)lds_raw;
    const int wpb = blockDim.x >> 6, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int* tstart = (int*)(lds_raw + (size_t)TAB_FLOATS * sizeof(float) + (size_t)wpb * tile_wave_lds_bytes(1));     // [n_heavy + 1]
    for (int j = threadIdx.x; j < n_heavy; j += blockDim.x) tstart[j] = a.ray_tiles[a.heavy_list[j]];
    __syncthreads();
    if (wv == 0) {                                         // exclusive prefix: lane l scans a contiguous stretch, the wave its 64 sums
        const int per = (n_heavy + 63) / 64, j0 = lane * per < n_heavy ? lane * per : n_heavy, j1 = j0 + per < n_heavy ? j0 + per : n_heavy;
        int sum = 0;
        for (int j = j0; j < j1; ++j) sum += tstart[j];
        int inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d); if (lane >= d) inc += v; }
        int run = inc - sum;
        for (int j = j0; j < j1; ++j) { const int v = tstart[j]; tstart[j] = run; run += v; }
        if (lane == 63) tstart[n_heavy] = inc;
    }
    __syncthreads();
    const int total = tstart[n_heavy];
    float* pn = (float*)(lds_raw + (size_t)TAB_FLOATS * sizeof(float) + (size_t)wv * tile_wave_lds_bytes(1));
    float* feat = pn + TILE * 4;
    const bool has_t = a.target_d != nullptr;
    float cf[MNE_N_LOSS];
#pragma unroll
    for (int q = 0; q < MNE_N_LOSS; ++q) cf[q] = a.coef ? a.coef[q] : 0.0f;
    const bool use_e = cf[MNE_L_E_FS] != 0.f || cf[MNE_L_E_CENTER] != 0.f || cf[MNE_L_E_TAIL] != 0.f;
    const bool use_co = cf[MNE_L_CO_FS] != 0.f || cf[MNE_L_CO_SDF] != 0.f;
    for (int task = (int)blockIdx.x * wpb + wv; task < total; task += gridDim.x * wpb) {      // (wave-major: see decode_kernel)
        int lo = 0, hi = n_heavy - 1;                      // last listed ray whose first tile is <= task
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tstart[mid] <= task) lo = mid; else hi = mid - 1;
        }
        const int r = a.heavy_list[lo], c = task - tstart[lo];
        const float4 g0 = *(const float4*)(a.heavy_rec + (size_t)r * 8), g1 = *(const float4*)(a.heavy_rec + (size_t)r * 8 + 4);
        RayGrad G;
        G.denom = g0.x; G.z_lim = g0.y; G.Aq = g0.z; G.g_dep = g0.w;
        G.g_rgb[0] = g1.x; G.g_rgb[1] = g1.y; G.g_rgb[2] = g1.z;
        const int Dn = (int)__float_as_uint(g1.w);
        const float td = has_t ? a.target_d[r] : 0.0f;
        float ro[3], rd[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { ro[q] = a.rays_o[r * 3 + q]; rd[q] = a.rays_d[r * 3 + q]; }
        hot_backward_tile<HID, HIDC, CP, TAB_FIRST, false>(a, r, c, Dn, G, td, ro, rd, cf, use_e, use_co, a.z_vals + (size_t)r * a.S,
                                                           a.raw + (size_t)r * a.S * 4, lane, pn, feat, atab);
    }
}

// -----------------------------------------------------------------------------------------------
// List appends of the binned plane update (tile_adam.hip).
// append_tile: the 32 samples of one (ray, tile) held by a wave, lane = (sample, plane level): every sample that receives
// gradient (`live`) is appended to the list of every 16x16-cell plane tile its 2x2 footprints touch.
// bin_kernel: one wave per ray, SELF-SUFFICIENT: it derives "which samples receive gradient" from the decode's outputs alone
// (raw sdf of the decoded prefix, z, target depth, loss coefficients) with the same rules as the training ray kernel
// (first_crossing_g / ray_resolved / sample_contrib are shared), so it does not depend on ray_kernel and the host can run
// it on a second stream BESIDE the backward kernels (mne_tile_bin).  Until round 3 the appends were the tail of
// ray_kernel's tile loop: their 100+ live registers forced spills in every training instantiation and the returning
// atomics (one per run of samples in the same list; ~60 us of same-address serialisation on the tiles around the camera
// centres, profiles/r03_bin_ablation.txt) sat on the critical path of every ray.
// -----------------------------------------------------------------------------------------------
template <int NSETS>
__device__ __forceinline__ void append_tile(const RenderArgs& a, const float (&pnv)[3], bool live, size_t e, int lane, int hf) {
#pragma unroll 1
    for (int set = 0; set < NSETS; ++set) {
        // The sample is appended to the list of every plane tile its 2x2 footprints touch (tile_adam.hip): one
        // returning atomic per RUN of lanes with the same list (run_meta), issued back to back for all 12 (24)
        // (plane, corner-tile) slots of the lane before any result is consumed; then one 32-byte entry per append.
        constexpr int NJ = 3;                              // the three orientations of one plane set, this lane's level
        int want[NJ * 4];
        unsigned meta[NJ * 4];                             // leader lane | rank << 8 | run length << 16
        int cix[NJ], ciy[NJ];                              // NW corner of the footprint in each of the lane's planes
        int pcap[NJ];                                      // list capacity / list offset of the lane's planes
        long long loff[NJ];
        float wq[NJ][4];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int pidx = set * 6 + 2 * j + hf;         // planes in [set][orient][level] order
            const int ori = (pidx % 6) / 2;
            const mne_plane_t& pl = a.sc.plane[pidx / 6][ori][pidx % 2];
            float gx, gy;
            orient_coords(ori, pnv[0], pnv[1], pnv[2], gx, gy);
            Bilin b;
            bilin_setup(gx, gy, pl.h, pl.w, b);
            wq[j][0] = b.w00; wq[j][1] = b.w01; wq[j][2] = b.w10; wq[j][3] = b.w11;
            const int ix1 = b.ix0 + 1 < pl.w ? b.ix0 + 1 : b.ix0, iy1 = b.iy0 + 1 < pl.h ? b.iy0 + 1 : b.iy0;
            const int tx0 = b.ix0 / MNE_TILE, tx1 = ix1 / MNE_TILE, ty0 = b.iy0 / MNE_TILE, ty1 = iy1 / MNE_TILE;
            const int base = a.bins.tile_base[pidx], ntx = a.bins.ntx[pidx];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tx = (q & 1) ? tx1 : tx0, ty = (q & 2) ? ty1 : ty0;
                const bool dup = ((q & 1) && tx1 == tx0) || ((q & 2) && ty1 == ty0);   // same tile again
                const int w_ = (live && !dup) ? base + ty * ntx + tx : -1;
                want[j * 4 + q] = w_;
                meta[j * 4 + q] = run_meta(w_, lane);
            }
            cix[j] = b.ix0; ciy[j] = b.iy0;
            pcap[j] = a.bins.pcap[pidx]; loff[j] = a.bins.list_off[pidx];
        }
        int first_slot[NJ * 4];
        // this wave's segment of every list: its XCD (more than eight segments: the workgroup index picks among the XCD's)
        const int xcc = MNE_LIST_SEGMENTS > 8 ? mne_xcc_id() + 8 * (int)((blockIdx.x >> 3) % (MNE_LIST_SEGMENTS / 8 > 0 ? MNE_LIST_SEGMENTS / 8 : 1))
                                              : mne_xcc_id() % MNE_LIST_SEGMENTS;
#pragma unroll
        for (int q = 0; q < NJ * 4; ++q) {
            first_slot[q] = 0;
            if (want[q] >= 0 && (int)(meta[q] & 255u) == lane)
                first_slot[q] = atomicAdd(a.bins.counts + want[q] * MNE_LIST_SEGMENTS + xcc, (int)(meta[q] >> 16));      // this XCD's segment of the list
        }
        const unsigned trow = (unsigned)e;                 // tape row of this sample
#pragma unroll
        for (int eq = 0; eq < NJ * 4; ++eq) {
            const int f0 = __shfl(first_slot[eq], (int)(meta[eq] & 255u));
            if (want[eq] >= 0) {
                const int slot = f0 + (int)((meta[eq] >> 8) & 255u);
                unsigned* dst = nullptr;
                const int seg_cap = pcap[eq >> 2] / MNE_LIST_SEGMENTS;
                if (slot < seg_cap) dst = a.bins.lists + (size_t)(loff[eq >> 2] + (long long)want[eq] * pcap[eq >> 2] + (long long)xcc * seg_cap + slot) * MNE_ENTRY_WORDS;
                else {
                    const int sp = atomicAdd(a.bins.spill_count, 1);
                    if (sp < a.bins.spill_cap) dst = a.bins.spill + (size_t)sp * MNE_ENTRY_WORDS;
                    else atomicAdd(a.bins.dropped, 1);              // caller-sized spill area too small: reported, never silent
                }
                if (dst) {
                    const int j = eq >> 2, q = eq & 3;
                    // corner relative to the tile this entry goes to (+1: 0 = one cell before the tile)
                    const int tx = (cix[j] + (q & 1)) / MNE_TILE, ty = (ciy[j] + (q >> 1)) / MNE_TILE;
                    const unsigned corner = (unsigned)(cix[j] - tx * MNE_TILE + 1) | ((unsigned)(ciy[j] - ty * MNE_TILE + 1) << 8);
                    *(uint4*)dst = make_uint4(trow, corner, __float_as_uint(wq[j][0]), __float_as_uint(wq[j][1]));
                    *(uint4*)(dst + 4) = make_uint4(__float_as_uint(wq[j][2]), __float_as_uint(wq[j][3]), (unsigned)want[eq], 0u);
                }
            }
        }
    }
}

// first adjacent sign change among samples [0, D) of a ray whose raw (r,g,b,sdf) rows start at `raw_ray` (global memory)
__device__ __forceinline__ int first_crossing_g(const float* raw_ray, int D, int lane) {
    for (int base = 0; base < D - 1; base += MNE_WAVE) {
        const int i = base + lane;
        const bool in = i < D - 1;
        const float s0 = raw_ray[4 * (in ? i : 0) + 3], s1 = raw_ray[4 * (in ? i + 1 : 0) + 3];
        const unsigned long long m = __ballot(in && (s1 * s0 < 0.0f));
        if (m) return base + __ffsll(m) - 1;
    }
    return -1;
}

// tile_need_kernel: backward of an EARLIER forward call (mne_render_backward: the autograd path, the pose loops of loop
// closure) -- which tiles of every ray will the backward walk?  Exactly ray_kernel<..., 2 | 3>'s own rule on the forward
// call's raw: first sign change over all samples (undecoded ones are NaN there and never form a crossing), render window
// z < z[first] + sc_factor * trunc, loss masks; need[r] = max(a-priori prefix, last contributing tile + 1).  The decode
// launch then makes the forward half of exactly those tape rows tile-parallel, and the backward kernel carries no decode.
__global__ __launch_bounds__(256) void tile_need_kernel(RenderArgs a, int* need) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.R) return;
    const int S = a.S, ntile = (S + TILE - 1) / TILE;
    const bool has_t = a.target_d != nullptr;
    const float td = has_t ? a.target_d[r] : 0.0f;
    const bool use_e = a.coef && (a.coef[MNE_L_E_FS] != 0.f || a.coef[MNE_L_E_CENTER] != 0.f || a.coef[MNE_L_E_TAIL] != 0.f);
    const bool use_co = a.coef && (a.coef[MNE_L_CO_FS] != 0.f || a.coef[MNE_L_CO_SDF] != 0.f);
    const float* zr = a.z_vals + (size_t)r * S;
    const int first = first_crossing_g(a.raw_in + (size_t)r * S * 4, S, lane);
    const float z_lim = zr[first < 0 ? 0 : first] + a.win_f;
    int last = -1;
    for (int base = 0; base < S; base += MNE_WAVE) {
        const int i = base + lane;
        bool f = false;
        if (i < S) {
            const float z = zr[i];
            f = sample_contrib(z, z_lim, sample_masks(z, td, has_t, a), use_e, use_co);
        }
        const unsigned long long m = __ballot(f);
        if (m) last = base + 63 - __clzll(m);
    }
    const int nb = last < 0 ? 0 : last / TILE + 1;
    const int t0 = prefix_tiles(a, r, ntile);                  // (a.tile_need is NULL in this launch: the a-priori prefix)
    if (lane == 0) need[r] = nb > t0 ? nb : t0;
}

// INVARIANTS of pass 0 running on another stream beside the backward kernels and the deferred pass (exercised by
// tests/test_kernels_hostemu.py::test_bench_path_step_external_bin_with_many_deferred_rays):
//   (1) a ray's decoded-tile count dec_tiles[r] is written ONLY by the resolver wave of the first decode launch (decode_kernel:
//       `resolver` requires !a.ray_list): the list decode of the deferred pass never touches it -- otherwise pass 0 could see a
//       deferred ray as fully decoded and append it a second time;
//   (2) pass 0 takes every ray's prefix from dec_tiles (or, without per-ray counts, from prefix_default), never from the
//       adaptive-schedule word adapt[0], which the LAST ray launch of the render call rewrites for the next call while pass 0
//       may still be running (prefix_tiles() reads it: bin_kernel therefore calls it only for list passes, where it returns
//       ntile before looking at adapt, and without dec_tiles).
#ifndef BIN_WPR
#define BIN_WPR 4               // waves of a workgroup that share one ray's tiles (1, 2 or 4)
#endif
#ifndef BIN_WG_PER_CU
#define BIN_WG_PER_CU 8         // grid cap of the appends (x 4 waves = every wave slot); long-ray batches: one workgroup per CU
#endif
template <bool CP>
__global__ __launch_bounds__(256) void bin_kernel(RenderArgs a) {
    constexpr int NSETS = CP ? 2 : 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pt = lane & 31, hf = lane >> 5;
    const int S = a.S, ntile = (S + TILE - 1) / TILE;
    if (a.ray_list && *a.ray_list_count == 0) return;      // deferred-ray pass with nothing deferred (the usual case)
    const bool has_t = a.target_d != nullptr;
    float cf[MNE_N_LOSS];
#pragma unroll
    for (int q = 0; q < MNE_N_LOSS; ++q) cf[q] = a.coef ? a.coef[q] : 0.0f;
    const bool use_e = cf[MNE_L_E_FS] != 0.f || cf[MNE_L_E_CENTER] != 0.f || cf[MNE_L_E_TAIL] != 0.f;
    const bool use_co = cf[MNE_L_CO_FS] != 0.f || cf[MNE_L_CO_SDF] != 0.f;
    const int n_items = a.ray_list ? *a.ray_list_count : a.R;
    // BIN_WPR waves share a ray: each derives the ray's first sign change / render window itself (a few loads) and takes
    // every BIN_WPR-th decoded tile.  With one wave per ray the kernel was as long as the longest ray's walk -- a chain of
    // atomic round trips per tile: 72 us on office0 (<= 4 tiles per ray), 176 us on ScanNet, 249 us on INS Indoor (up to 33
    // tiles per ray) -- beside a ray kernel that needs the same time or less (profiles/r05_bin_waves_per_ray.txt).
    constexpr int RPB = 4 / BIN_WPR;                                   // rays per workgroup
    const int wr = wv / BIN_WPR, wt = wv % BIN_WPR;
    for (int item = blockIdx.x * RPB + wr; item < n_items; item += gridDim.x * RPB) {
        const int r = a.ray_list ? a.ray_list[item] : item;
        const float td = has_t ? a.target_d[r] : 0.0f;
        const float* zr = a.z_vals + (size_t)r * S;
        const float* raw_ray = a.raw + (size_t)r * S * 4;
        const int t_dec = (a.dec_tiles && !a.ray_list && a.ray_counts) ? a.dec_tiles[r] : prefix_tiles(a, r, ntile, a.ray_list != nullptr);
        const int Dn = t_dec * TILE < S ? t_dec * TILE : S;
        const int first = first_crossing_g(raw_ray, Dn, lane);
        if (!ray_resolved(zr, first, Dn, S, a.win_f)) continue;       // the deferred pass finishes this ray; its appends come with it
        const float z_lim = zr[first < 0 ? 0 : first] + a.win_f;
        float ro[3], rd[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { ro[q] = a.rays_o[r * 3 + q]; rd[q] = a.rays_d[r * 3 + q]; }
        // The rays of one keyframe start at one point: every ray's first tile lands in the lists of the same few plane
        // tiles, and reservations on one counter are served one after the other (profiles/r03_bin_ablation.txt).  Each ray
        // therefore starts at a different tile of its own and wraps around, which spreads those reservations over the
        // kernel's run time instead of queueing them all at its start.
        const int c0 = (int)((unsigned)r % (unsigned)t_dec);
        for (int k = wt; k < t_dec; k += BIN_WPR) {
            const int c = c0 + k < t_dec ? c0 + k : c0 + k - t_dec;
            const int i = c * TILE + pt;
            const bool valid = i < Dn;
            const int ii = valid ? i : Dn - 1;
            const float z = zr[ii];
            const bool live = valid && sample_contrib(z, z_lim, sample_masks(z, td, has_t, a), use_e, use_co);
            if (__ballot(live) == 0ull) continue;                       // nothing of this tile receives gradient
            float p[3], pnv[3], u[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) p[q] = ro[q] + rd[q] * z;
            point_coords(a.sc, p, pnv, u);
            append_tile<NSETS>(a, pnv, live, (size_t)r * S + ii, lane, hf);
        }
    }
}

// -----------------------------------------------------------------------------------------------
// scatter_kernel: plane gradients by global atomics (scatter="atomics" and the autograd path), the counterpart of
// bin_kernel + tile_adam_kernel: one wave per back-propagated (ray, tile); the d(feature) rows ray_kernel left in the tape
// and the normalised points go through the wave's LDS rows into scatter_chunk (one 128-B line of atomics per half-wave).
// -----------------------------------------------------------------------------------------------
template <bool CP>
__global__ __launch_bounds__(256) void scatter_kernel(RenderArgs a, int ntile) {
    constexpr int NSETS = CP ? 2 : 1;
    MNE_DYN_LDS(lds_raw);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pt = lane & 31, hf = lane >> 5;
    const int S = a.S;
    float* pn = (float*)(lds_raw + (size_t)wv * tile_wave_lds_bytes(NSETS));
    float* feat = pn + TILE * 4;
    const long long ntask = (long long)a.R * ntile;
    for (long long task = (long long)blockIdx.x * 4 + wv; task < ntask; task += (long long)gridDim.x * 4) {
        const int c = (int)(task / a.R), r = (int)(task % a.R);
        if (c >= a.ray_tiles[r]) continue;                             // whole wave together
        const int i = c * TILE + pt;
        const bool valid = i < S;
        const float* tape0 = a.tape + ((size_t)r * S + (size_t)c * TILE) * a.tape_row;
        const float4 pn4 = *(const float4*)(tape0 + (size_t)(valid ? pt : 0) * a.tape_row + a.tape_tpn);
        const unsigned live_rows = (unsigned)__ballot(valid && pn4.w != 0.0f && hf == 0);
        MNE_WAVE_SYNC();                                               // the previous task's LDS reads are done
        if (hf == 0) *(float4*)(pn + pt * 4) = pn4;
#pragma unroll
        for (int set = 0; set < NSETS; ++set)
            load_rows<MNE_FEAT>(feat + set * TILE * MNE_FS, tape0, a.tape_row, a.tape_tdfeat + set * MNE_FEAT, live_rows, lane);
        MNE_WAVE_SYNC();
        scatter_chunk<NSETS, TILE>(a.sc, pn, feat, live_rows, lane);
    }
}

// -----------------------------------------------------------------------------------------------
// loss scalars / coefficients (single small block; deterministic summation order)
// -----------------------------------------------------------------------------------------------
// body of the loss scalars: all 256 threads of ONE workgroup
__device__ __forceinline__ void loss_finalize_block(const LossArgs& a, double (*part)[MNE_N_LOSS]) {
    const int t = threadIdx.x;
    double acc[MNE_N_LOSS];
    for (int k = 0; k < MNE_N_LOSS; ++k) acc[k] = 0.0;
    for (int r = t; r < a.R; r += 256)
        for (int k = 0; k < MNE_N_LOSS; ++k) acc[k] += (double)a.ray_sums[(size_t)r * MNE_N_LOSS + k];
    for (int k = 0; k < MNE_N_LOSS; ++k) part[t][k] = acc[k];
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (t < st)
            for (int k = 0; k < MNE_N_LOSS; ++k) part[t][k] += part[t + st][k];
        __syncthreads();
    }
    if (t == 0) {
        const double n_valid = a.counts[MNE_C_VALID], n_ef = a.counts[MNE_C_E_FRONT], n_ec = a.counts[MNE_C_E_CENTER];
        const double n_et = a.counts[MNE_C_E_TAIL], n_cf = a.counts[MNE_C_CO_FS], n_cs = a.counts[MNE_C_CO_SDF];
        const double rs = (double)a.R * (double)a.S;
        const float fs_w = 1.0f - (float)n_cf / (float)(n_cf + n_cs);      // model/utils.py:141-145 (0/0 -> NaN)
        const float sdf_w = 1.0f - (float)n_cs / (float)(n_cf + n_cs);
        const float rgb_loss = (float)(part[0][MNE_L_RGB] / (3.0 * a.R));
        a.losses[MNE_L_RGB] = rgb_loss;
        a.losses[MNE_L_DEPTH] = (float)(part[0][MNE_L_DEPTH] / n_valid);   // mean of empty -> 0/0 = NaN
        a.losses[MNE_L_CO_SDF] = (float)(part[0][MNE_L_CO_SDF] / rs) * sdf_w;
        a.losses[MNE_L_CO_FS] = (float)(part[0][MNE_L_CO_FS] / rs) * fs_w;
        a.losses[MNE_L_E_FS] = (float)(part[0][MNE_L_E_FS] / n_ef);
        a.losses[MNE_L_E_CENTER] = (float)(part[0][MNE_L_E_CENTER] / n_ec);
        a.losses[MNE_L_E_TAIL] = (float)(part[0][MNE_L_E_TAIL] / n_et);
        a.losses[MNE_L_PSNR] = -10.0f * logf(rgb_loss) / logf(10.0f);      // model/utils.py:43-47
    }
}

__global__ __launch_bounds__(256) void loss_finalize_kernel(LossArgs a) {
    __shared__ double part[256][MNE_N_LOSS];
    loss_finalize_block(a, part);
}

// d(total)/d(sample) coefficients: ONE thread
__device__ __forceinline__ void loss_coef_thread(const LossArgs& a) {
    const float n_valid = (float)a.counts[MNE_C_VALID], n_ef = (float)a.counts[MNE_C_E_FRONT];
    const float n_ec = (float)a.counts[MNE_C_E_CENTER], n_et = (float)a.counts[MNE_C_E_TAIL];
    const float n_cf = (float)a.counts[MNE_C_CO_FS], n_cs = (float)a.counts[MNE_C_CO_SDF];
    const float rs = (float)a.R * (float)a.S;
    const float fs_w = 1.0f - n_cf / (n_cf + n_cs), sdf_w = 1.0f - n_cs / (n_cf + n_cs);
    const float* g = a.grad_losses;
    // d mean((x-t)^2) / dx = 2 (x-t) / N ; a term whose selection is empty gets no gradient
    a.coef[MNE_L_RGB] = g[MNE_L_RGB] * 2.0f / (3.0f * (float)a.R);
    a.coef[MNE_L_DEPTH] = n_valid > 0.f ? g[MNE_L_DEPTH] * 2.0f / n_valid : 0.0f;
    a.coef[MNE_L_CO_SDF] = (n_cf + n_cs) > 0.f ? g[MNE_L_CO_SDF] * sdf_w * 2.0f * a.co_T / rs : 0.0f;
    a.coef[MNE_L_CO_FS] = (n_cf + n_cs) > 0.f ? g[MNE_L_CO_FS] * fs_w * 2.0f / rs : 0.0f;
    a.coef[MNE_L_E_FS] = n_ef > 0.f ? g[MNE_L_E_FS] * 2.0f / n_ef : 0.0f;
    a.coef[MNE_L_E_CENTER] = n_ec > 0.f ? g[MNE_L_E_CENTER] * 2.0f * a.e_T / n_ec : 0.0f;
    a.coef[MNE_L_E_TAIL] = n_et > 0.f ? g[MNE_L_E_TAIL] * 2.0f * a.e_T / n_et : 0.0f;
    a.coef[MNE_L_PSNR] = 0.0f;
}

__global__ void loss_coef_kernel(LossArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    loss_coef_thread(a);
}

// -----------------------------------------------------------------------------------------------
// decoder_update_kernel: fixed-order sum of the partial weight gradients (wgrad.hip), torch.optim.Adam on the 6 k decoder
// parameters and the loss scalars of the iteration in ONE launch (were three: wgrad_reduce, adam, loss_finalize -- 45 us of
// launch latency in the decoder's dependency chain beside the plane update, profiles/r03_timeline_mid.txt); the caller
// follows with pack_decoder_kernel for the next render's tables.
// -----------------------------------------------------------------------------------------------
template <int HID, int HIDC, bool CP>
__global__ __launch_bounds__(256) void decoder_update_kernel(DecUpdateArgs a) {
    typedef DecDims<HID, HIDC, CP> D;
    __shared__ float part[8][32];
    __shared__ double lpart[256][MNE_N_LOSS];
    const int tid = threadIdx.x;
    const int e = blockIdx.x * 32 + (tid & 31), grp = tid >> 5;
    float s = 0.0f;
    if (e < D::NPARAM) {
#pragma unroll 8
        for (int w = grp; w < a.n_partials; w += 8) s += a.partials[(size_t)w * D::NPARAM + e];
    }
    part[grp][tid & 31] = s;
    __syncthreads();
    if (grp == 0 && e < D::NPARAM) {
        float g = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) g += part[k][tid];
        a.grad_out[e] = g;
        // decoder.parameters() order: col0 | col1 | sdf0 | sdf1
        const int t = e < D::P_COL1 ? 0 : e < D::P_SDF0 ? 1 : e < D::P_SDF1 ? 2 : 3;
        const int off = e - (t == 0 ? D::P_COL0 : t == 1 ? D::P_COL1 : t == 2 ? D::P_SDF0 : D::P_SDF1);
        float* P = (float*)(t == 0 ? a.sc.w_col0 : t == 1 ? a.sc.w_col1 : t == 2 ? a.sc.w_sdf0 : a.sc.w_sdf1);
        PlaneOpt o = a.opt;
        if (a.clk.bias_table) clock_bias(a.clk, o.lr, o.step, o.step_size, o.bc2_sqrt);
        float p = P[off], m = a.m[t][off], v = a.v[t][off];
        adam_elem(p, g, m, v, o);
        P[off] = p; a.m[t][off] = m; a.v[t][off] = v;
    }
    if (blockIdx.x == 0 && a.fin.losses) loss_finalize_block(a.fin, lpart);
}

template <int HID, int HIDC, bool CP>
static int launch_decoder_update(const DecUpdateArgs& a, hipStream_t st) {
    typedef DecDims<HID, HIDC, CP> D;
    MNE_LAUNCH((decoder_update_kernel<HID, HIDC, CP>), (D::NPARAM + 31) / 32, 256, 0, st, a);
    return 0;
}

// -----------------------------------------------------------------------------------------------
// point queries (forward only): 32 points per wave, same gather / MFMA building blocks
// -----------------------------------------------------------------------------------------------
template <int HID, int HIDC, bool CP>
__global__ __launch_bounds__(256) void query_kernel(QueryArgs a) {
    constexpr int NSETS = CP ? 2 : 1;
    MNE_DYN_LDS(lds_raw);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long tile = (long long)blockIdx.x * RAYS_PER_WG + wv;
    if (tile * TILE >= a.n) return;
    float* pn = (float*)lds_raw + (size_t)wv * (TILE * 4 + NSETS * TILE * MNE_FS);
    float* feat = pn + TILE * 4;
    const int pt = lane & 31, hf = lane >> 5;
    const long long i = tile * TILE + pt;
    const bool valid = i < a.n;
    const long long ii = valid ? i : a.n - 1;
    const float p[3] = {a.pts[ii * 3 + 0], a.pts[ii * 3 + 1], a.pts[ii * 3 + 2]};
    float pnv[3], u[3];
    point_coords(a.sc, p, pnv, u);
    if (a.flags & MNE_QUERY_PTS_NORMALISED) { pnv[0] = p[0]; pnv[1] = p[1]; pnv[2] = p[2]; }
    if (hf == 0) *(float4*)(pn + pt * 4) = make_float4(pnv[0], pnv[1], pnv[2], 0.0f);
    MNE_WAVE_SYNC();
    if (a.corner_idx && valid) {
        // integer NW corners of every plane: the same bilin_setup on the same staged fp32 coordinates as the
        // gather below (lane half = plane level)
        const float qx = pn[pt * 4 + 0], qy = pn[pt * 4 + 1], qz = pn[pt * 4 + 2];
#pragma unroll
        for (int set = 0; set < NSETS; ++set)
#pragma unroll
            for (int ori = 0; ori < 3; ++ori) {
                const mne_plane_t& pl = a.sc.plane[set][ori][hf];
                float gx, gy;
                orient_coords(ori, qx, qy, qz, gx, gy);
                Bilin b;
                bilin_setup(gx, gy, pl.h, pl.w, b);
                int* dst = a.corner_idx + ((i * (3 * NSETS) + set * 3 + ori) * 2 + hf) * 2;
                dst[0] = b.ix0; dst[1] = b.iy0;
            }
    }
    if (a.ext_rows) {                                       // caller-supplied feature rows (hash / dense grid model)
        const unsigned live = (unsigned)__ballot(valid && hf == 0);
        load_rows<MNE_FEAT>(feat, a.ext_rows + (size_t)tile * TILE * MNE_FEAT, MNE_FEAT, 0, live, lane);
    } else {
        gather_chunk<NSETS, TILE>(a.sc, pn, feat, lane);
    }
    MNE_WAVE_SYNC();
    if (a.feat_out && valid) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            *(float4*)(a.feat_out + i * MNE_FEAT + hf * 32 + 4 * q) = *(const float4*)(feat + pt * MNE_FS + hf * 32 + 4 * q);
    }
    if (a.raw || a.geo) {
        float pos[24];
        oneblob_half(u, hf, pos);
        MlpState<HID, HIDC> st;
        mlp_forward_mfma<HID, HIDC, CP>(feat + pt * MNE_FS, feat + TILE * MNE_FS + pt * MNE_FS, pos, a.packed, lane, st);
        if (valid) {
            if (a.raw && hf == 0) *(float4*)(a.raw + i * 4) = make_float4(st.rgb[0], st.rgb[1], st.rgb[2], st.out[0]);
            if (a.geo) {
                // out16 rows held by this lane: m = (r&3) + 8(r>>2) + 4 hf, r = 0..7; geo index = m-1
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int m = mfma_row(rr, 0) + 4 * hf;
                    if (m >= 1) a.geo[i * MNE_GEO + m - 1] = st.out[rr];
                }
            }
        }
    }
}

// -----------------------------------------------------------------------------------------------
// stand-alone OneBlob (the get_encoder('OneBlob') module surface)
// -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void oneblob_kernel(long long n_elems, const float* x, float* out) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one (point, dim) per thread
    if (e >= n_elems) return;
    float o[MNE_NB];
    oneblob16(x[e], o);
#pragma unroll
    for (int q = 0; q < MNE_NB / 4; ++q)
        *(float4*)(out + e * MNE_NB + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}

int mne_launch_oneblob(long long n, int dims, const float* x, float* out, hipStream_t st) {
    const long long ne = n * dims;
    MNE_LAUNCH(oneblob_kernel, (unsigned)((ne + 255) / 256), 256, 0, st, ne, x, out);
    return 0;
}

// -----------------------------------------------------------------------------------------------
// host-side launchers (called from capi.hip)
// -----------------------------------------------------------------------------------------------
// Workgroup shape of the tile kernels: the A tables (staged in LDS except for the largest decoder,
// which reads them through L2) plus one private region per wave.  The grid is persistent: at most
// one workgroup per CU (the LDS footprint allows no more), each wave striding over the tile tasks.
#define MNE_NUM_CU 256
#ifndef MNE_HOT_LDS_SAMPLES
#define MNE_HOT_LDS_SAMPLES 256
#endif
template <int HID, int HIDC, bool CP> struct WgShape {
    static constexpr bool ALDS = !(HID == 64 && CP);
};

// LDS of decode_kernel (tables: forward steps) / ray_kernel (tables by mode; + raws[Spad][4] per wave)
template <int HID, int HIDC, bool CP>
static size_t table_bytes(int mode) {
    typedef ATab<HID, HIDC, CP> T;
    if (!WgShape<HID, HIDC, CP>::ALDS) return 0;
    return (size_t)(mode == 3 ? T::TOTAL_RAYGRAD : mode >= 1 ? T::TOTAL : T::FWD_STEPS) * 64 * sizeof(float);   // mode 4: see launch_ray
}
template <int HID, int HIDC, bool CP>
static int fit_waves(size_t tab, size_t per_wave, int max_wpb) {
    int fit = 0;
    for (int k = 1; k <= max_wpb; ++k)
        if (tab + (size_t)k * per_wave <= MNE_LDS_MAX) fit = k;
    return fit;
}

static size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
// backward workspace: ReLU masks [R*S][4] u32 | deferred-ray list [R] | its length [1]
size_t mne_render_workspace(int R, int S) {
    return align16((size_t)R * S * 4 * sizeof(unsigned)) + 5 * align16((size_t)R * sizeof(int)) + align16((size_t)R * 8 * sizeof(float)) + 32;
}
static void carve_workspace(RenderArgs& a, void* ws) {
    unsigned char* p = (unsigned char*)ws;
    a.relu_mask = (unsigned*)p; p += align16((size_t)a.R * a.S * 4 * sizeof(unsigned));
    a.defer_list = (int*)p; p += align16((size_t)a.R * sizeof(int));
    a.dec_tiles = (int*)p; p += align16((size_t)a.R * sizeof(int));
    a.long_list = (int*)p; p += align16((size_t)a.R * sizeof(int));
    a.heavy_list = (int*)p; p += align16((size_t)a.R * sizeof(int));
    a.heavy_rec = (float*)p; p += align16((size_t)a.R * 8 * sizeof(float));
    a.defer_count = (int*)p;
    a.heavy_count = (int*)(p + 8);
    a.long_count = (int*)(p + 16);
}

int mne_half_bits_for(long long n);
int mne_launch_batch(SampleRaysArgs sr, unsigned long long seed, unsigned long long iteration, const ZArgs& a, const LossArgs& lc,
                     hipStream_t st) {
    sr.half_bits_kf = mne_half_bits_for(sr.n_kf_rays);
    sr.half_bits_cur = mne_half_bits_for(sr.n_cur_rays);
    sr.seed = seed; sr.iteration = iteration;
    const int n_tab = a.has_d ? a.n_a + 2 * a.n_b : a.S;
    const size_t lds = (size_t)(((n_tab + 3) & ~3) + RAYS_PER_WG * ((a.S + 3) & ~3)) * sizeof(float);
    MNE_LAUNCH(batch_kernel, (a.R + RAYS_PER_WG - 1) / RAYS_PER_WG, 256, lds, st, sr, a);
    MNE_LAUNCH(counts_coef_kernel, 1, 256, 0, st, a, lc);
    return 0;
}

int mne_launch_sample_z(const ZArgs& a, hipStream_t st) {
    const int n_tab = a.has_d ? a.n_a + 2 * a.n_b : a.S;
    const size_t lds = (size_t)(((n_tab + 3) & ~3) + RAYS_PER_WG * ((a.S + 3) & ~3)) * sizeof(float);
    MNE_LAUNCH(sample_z_kernel, (a.R + RAYS_PER_WG - 1) / RAYS_PER_WG, 256, lds, st, a);
    if (a.has_d) {
        if (a.R > MNE_BALANCED_MAX_RAYS) {                 // whole frames: a stripe of the rays per workgroup
            (void)hipMemsetAsync(a.counts, 0, MNE_N_COUNT * sizeof(int), st);
            MNE_LAUNCH(counts_reduce_kernel, MNE_NUM_CU, 256, 0, st, a);
        } else MNE_LAUNCH(counts_reduce_kernel, 1, 256, 0, st, a);
    }
    return 0;
}

template <int HID, int HIDC, bool CP>
static int launch_pack(const mne_scene_t& sc, float* pk, hipStream_t st) {
    const int n = ATab<HID, HIDC, CP>::TOTAL_RAYGRAD * 64;
    MNE_LAUNCH((pack_decoder_kernel<HID, HIDC, CP>), (n + 255) / 256, 256, 0, st, sc, pk);
    return 0;
}

static int ray_lds_cap(const RenderArgs& a) { return a.lds_samples > 0 ? a.lds_samples : MNE_HOT_LDS_SAMPLES; }   // (tests pass a small cap)

template <int HID, int HIDC, bool CP, int MODE>
static int launch_ray(RenderArgs a, hipStream_t st, int max_blocks = MNE_NUM_CU) {
    typedef WgShape<HID, HIDC, CP> W;
    typedef ATab<HID, HIDC, CP> T;
    // A tables in LDS: always for the training kernel (backward steps only: at most 46 KiB); the other modes of the
    // largest decoder (2x64 + colour planes: 124 KiB of tables) read them through L2
    constexpr bool ALDS = W::ALDS || MODE == 4 || MODE == 2;
    size_t tab = table_bytes<HID, HIDC, CP>(MODE);
    if (MODE == 4 || MODE == 2) tab = (size_t)(T::TOTAL - T::FWD_STEPS) * 64 * sizeof(float);
    else if (MODE == 3 && tab) tab -= (size_t)T::FWD_STEPS * 64 * sizeof(float);
    // Training kernel, first pass: LDS for MNE_HOT_LDS_SAMPLES samples per ray instead of S (INS Indoor: S = 1045 would
    // leave room for 4 waves per CU, i.e. 1024 of 2150 rays at a time); the few rays whose decoded prefix is longer
    // go to the second pass, which is sized for S.
    const int cap = ray_lds_cap(a);
    a.lds_samples = (MODE == 4 && !a.ray_list && a.S > cap) ? cap : 0;
    const int L = a.lds_samples ? a.lds_samples : a.S;
    const size_t per_wave = (size_t)((L + 3) & ~3) * 5 * sizeof(float) + tile_wave_lds_bytes((CP && MODE != 4 && !(MNE_SEQ_FORWARD && MODE == 0)) ? 2 : 1, MODE == 3);
    const int wpb = fit_waves<HID, HIDC, CP>(tab, per_wave, RAY_WPB(HID, CP, MODE));
    if (wpb < 1) return -4;
    const size_t lds = tab + (size_t)wpb * per_wave;
    if (lds > 64 * 1024) MNE_SET_MAX_LDS((ray_kernel<HID, HIDC, CP, ALDS, MODE>), MNE_LDS_MAX);
    long long grid = ((long long)a.R + wpb - 1) / wpb;
#ifndef MNE_HOST_EMU                                        // (the host emulator pays per workgroup: it keeps the dense grid, same item loop)
    if (MODE == 4 && MNE_WAVE_MAJOR && !a.ray_list && a.ext_feat) grid = a.R;      // training, first pass, caller-supplied features: rays are dealt wave-major over the workgroups (ray_kernel's item loop)
#endif
    if (grid > max_blocks) grid = max_blocks;
    MNE_LAUNCH((ray_kernel<HID, HIDC, CP, ALDS, MODE>), (unsigned)grid, 64 * wpb, lds, st, a);
    return 0;
}

// Timing marks (mne_fused_opts_t::timing_events): events a training render records between its kernels, so that a benchmark
// can time gather / decode / ray / deferred pass / binning live without a profiler.  Per-call state, owned by the caller.
static inline void mark(const RenderHost& h, int i, hipStream_t st) {
    if (i < h.n_marks && h.marks && h.marks[i]) (void)hipEventRecord((hipEvent_t)h.marks[i], st);
}

// The deferred pass (decode -> ray -> appends of the rays the training kernel could not resolve) sits on the critical path of
// every iteration, and in steady state its list is EMPTY or a handful of rays (an untrained map, where it is not, switches to
// decoding every sample a priori: adapt_state): what it costs then is dispatching three full-size grids that leave at once.
// Its kernels loop over the list with a grid stride, so a small grid is enough: 64 workgroups decode ~500 tile tasks / hold
// ~380 rays at a time.  (Measured: +1 % on the first iterations of a fresh map, nothing in steady state -- an empty launch
// costs its ~5 us whatever its grid, profiles/r04_list_pass_grid.txt.)
#ifndef MNE_LIST_PASS_BLOCKS
#define MNE_LIST_PASS_BLOCKS 64
#endif

#ifndef MNE_HEAVY_NTILE
#define MNE_HEAVY_NTILE 8       // rays of more than this many 32-sample tiles: the heavy list exists
#endif
#ifndef MNE_HEAVY_TILES
#define MNE_HEAVY_TILES 2       // ... and takes the rays with more than this many backward tiles (0: never)
#endif
template <int HID, int HIDC, bool CP>
static int launch_heavy(const RenderArgs& a, hipStream_t st) {
    typedef ATab<HID, HIDC, CP> T;
    const size_t tab = (size_t)(T::TOTAL - T::FWD_STEPS) * 64 * sizeof(float);
    const size_t sched = align16((size_t)(a.R + 1) * sizeof(int));
    const int wpb = fit_waves<HID, HIDC, CP>(tab + sched, tile_wave_lds_bytes(1), RAY_WPB(HID, CP, 4));
    if (wpb < 1) return -4;
    const size_t lds = tab + (size_t)wpb * tile_wave_lds_bytes(1) + sched;
    if (lds > 64 * 1024) MNE_SET_MAX_LDS((heavy_bwd_kernel<HID, HIDC, CP>), MNE_LDS_MAX);
    MNE_LAUNCH((heavy_bwd_kernel<HID, HIDC, CP>), MNE_NUM_CU, 64 * wpb, lds, st, a);
    return 0;
}

// pre = true: the a-priori tiles' plane features are gathered by gather_kernel first (needs the tape)
template <int HID, int HIDC, bool CP>
static int launch_decode(RenderArgs d, hipStream_t st, const RenderHost& host, bool pre = false) {
    typedef DecDims<HID, HIDC, CP> D;
    typedef WgShape<HID, HIDC, CP> W;
    pre = pre && d.tape && !d.ray_list;
    if (pre && !d.ext_feat) {
        d.tape_row = D::ROW; d.tape_tx = D::T_X; d.tape_tcf = D::T_CF;
        const int chunks = (d.S + 7) / 8;
        const long long waves = (long long)d.R * chunks;
        mark(host, 0, st);
        if (d.sc.plane_f16) MNE_LAUNCH((gather_kernel<CP, true>), (unsigned)((waves + 3) / 4), 256, 0, st, d, chunks);
        else MNE_LAUNCH((gather_kernel<CP, false>), (unsigned)((waves + 3) / 4), 256, 0, st, d, chunks);
        mark(host, 1, st);
    }
    const size_t tab = table_bytes<HID, HIDC, CP>(0);
    // forward-only launch of a model with colour planes (no tape to fill, no pre-gathered rows): one set of feature rows per wave
    const bool seqf = MNE_SEQ_FORWARD && CP && !d.tape && !pre;
    const int nsets = (CP && !seqf) ? 2 : 1;
    const int wpb = fit_waves<HID, HIDC, CP>(tab, tile_wave_lds_bytes(nsets), DECODE_WPB);
    if (wpb < 1) return -4;
    size_t lds = tab + (size_t)wpb * tile_wave_lds_bytes(nsets);
    // balanced tile schedule (see decode_kernel): needs the tile-count prefix of mne_sample_z / mne_sample_batch and room for
    // it in LDS; the adaptive "decode everything" state is checked on the device
    const size_t sched_bytes = align16((size_t)d.R * sizeof(int));
    const int sched = (MNE_DECODE_BALANCED && d.ray_counts && !d.ray_list && !d.tile_need && d.R <= MNE_BALANCED_MAX_RAYS &&
                       lds + sched_bytes <= MNE_LDS_MAX) ? 1 : 0;
    if (sched) lds += sched_bytes;
    const long long ntask = (long long)d.R * ((d.S + TILE - 1) / TILE);
    long long grid = (ntask + wpb - 1) / wpb;
    if (grid > MNE_NUM_CU) grid = MNE_NUM_CU;
    if (d.ray_list && grid > MNE_LIST_PASS_BLOCKS) grid = MNE_LIST_PASS_BLOCKS;      // deferred pass: see MNE_LIST_PASS_BLOCKS
    // gfx950 has 160 KiB of LDS per CU; above 64 KiB HIP wants an opt-in
    if constexpr (MNE_DECODE_FRAME && HID == 32 && HIDC == 32 && !CP) {
        // whole frames (forward only, a priori prefix, no caller-supplied features): the lean kernel at 12 waves per CU
        const int fwpb = fit_waves<HID, HIDC, CP>(tab, tile_wave_lds_bytes(1), MNE_FRAME_WPB);
        const int min_tiles = d.frame_min_tiles > 0 ? d.frame_min_tiles : d.frame_min_tiles < 0 ? 0 : MNE_FRAME_MIN_TILES;
        const bool frame = fwpb >= 1 && !d.tape && !pre && !d.relu_mask && !d.dec_tiles && !d.ray_list && !d.ext_feat && d.raw &&
                           (long long)d.R >= (long long)MNE_NUM_CU * fwpb * min_tiles;
        if (frame) {
            size_t flds = tab + (size_t)fwpb * tile_wave_lds_bytes(1);
            const int fsched = (sched && flds + sched_bytes <= MNE_LDS_MAX) ? 1 : 0;
            if (fsched) flds += sched_bytes;
            long long fgrid = (ntask + fwpb - 1) / fwpb;
            if (fgrid > MNE_NUM_CU) fgrid = MNE_NUM_CU;
            if (flds > 64 * 1024) MNE_SET_MAX_LDS((decode_frame_kernel<MNE_FRAME_WPB>), MNE_LDS_MAX);
            MNE_LAUNCH((decode_frame_kernel<MNE_FRAME_WPB>), (unsigned)fgrid, 64 * fwpb, flds, st, d, fsched);
            return 0;
        }
    }
    if constexpr (CP) {
        if (seqf) {
            if (lds > 64 * 1024) MNE_SET_MAX_LDS((decode_kernel<HID, HIDC, CP, W::ALDS, DECODE_WPB, true>), MNE_LDS_MAX);
            MNE_LAUNCH((decode_kernel<HID, HIDC, CP, W::ALDS, DECODE_WPB, true>), (unsigned)grid, 64 * wpb, lds, st, d, 0, sched);
            return 0;
        }
    }
    if (lds > 64 * 1024) MNE_SET_MAX_LDS((decode_kernel<HID, HIDC, CP, W::ALDS, DECODE_WPB>), MNE_LDS_MAX);
    MNE_LAUNCH((decode_kernel<HID, HIDC, CP, W::ALDS, DECODE_WPB>), (unsigned)grid, 64 * wpb, lds, st, d, pre ? 1 : 0, sched);
    return 0;
}

// List appends of the binned plane update for the rays of `pass`: 0 = every ray the decoded prefix resolves (needs the
// decode's outputs only), 1 = the rays of the deferred list (after the deferred pass decoded their remaining tiles).
template <bool CP>
static int launch_bin(RenderArgs b, int pass, hipStream_t st) {
    if (pass) { b.ray_list = b.defer_list; b.ray_list_count = b.defer_count; }
    else { b.ray_list = nullptr; b.ray_list_count = nullptr; }
    constexpr int rpb = 4 / BIN_WPR;
    long long grid = ((long long)b.R + rpb - 1) / rpb;
    // The appends are bound by the returning atomics, not by the number of waves -- and pass 0 runs BESIDE the ray kernels,
    // which need wave slots of their own: with every slot of the chip taken by this kernel (2048 workgroups) INS Indoor's
    // long-ray pass sat waiting until the appends had drained (profiles/r05_timeline_indoor_heavy.txt).
    // (measured, profiles/r05_bin_grid_cap.txt: Indoor 1074-1094 it/s with 8 workgroups per CU, 1135 with one; office0 prefers
    // the full grid -- its appends are as long as its ray kernel, 2113-2129 vs 2020 it/s)
    const int per_cu = (b.S + TILE - 1) / TILE > MNE_HEAVY_NTILE ? 1 : BIN_WG_PER_CU;
    if (grid > (long long)MNE_NUM_CU * per_cu) grid = (long long)MNE_NUM_CU * per_cu;
    if (pass && grid > MNE_LIST_PASS_BLOCKS) grid = MNE_LIST_PASS_BLOCKS;
    MNE_LAUNCH((bin_kernel<CP>), (unsigned)grid, 256, 0, st, b);
    return 0;
}

int mne_launch_bin(RenderArgs a, int pass, void* workspace, hipStream_t st) {
    carve_workspace(a, workspace);
    return a.sc.n_sets == 2 ? launch_bin<true>(a, pass, st) : launch_bin<false>(a, pass, st);
}

// After the ray kernels of a backward call: global atomics into plane[].grad from the d(feature) rows they left in the
// tape (scatter="atomics" and the autograd path); nothing when no plane gradient is wanted, the encoding is the caller's
// (ext_feat) or the plane update is binned (bin_kernel + tile_adam_kernel).
template <int HID, int HIDC, bool CP>
static void launch_plane_pass(const RenderArgs& a, hipStream_t st) {
    typedef DecDims<HID, HIDC, CP> D;
    if (!a.plane_grads || a.ext_feat || a.bins.lists) return;
    RenderArgs b = a;
    b.tape_row = D::ROW; b.tape_tpn = D::T_PN; b.tape_tdfeat = D::T_DFEAT;
    const int ntile = (a.S + TILE - 1) / TILE;
    long long grid = ((long long)a.R * ntile + 3) / 4;
    if (grid > MNE_NUM_CU * 8) grid = MNE_NUM_CU * 8;
    const size_t lds = 4 * tile_wave_lds_bytes(CP ? 2 : 1);
    if (lds > 64 * 1024) MNE_SET_MAX_LDS((scatter_kernel<CP>), MNE_LDS_MAX);
    MNE_LAUNCH((scatter_kernel<CP>), (unsigned)grid, 256, lds, st, b, ntile);
}

template <int HID, int HIDC, bool CP>
static int launch_render(RenderArgs a, int mode, void* workspace, const RenderHost& host, hipStream_t st) {
    a.plane_grads = (a.bins.lists != nullptr || a.sc.plane[0][0][0].grad != nullptr) ? 1 : 0;
    typedef WgShape<HID, HIDC, CP> W;
    const bool raygrad = a.d_rays_o != nullptr || a.d_rays_d != nullptr;
    if (raygrad && mode != 3) return -5;
    if (mode >= 2) {
        if (!workspace) return -6;
        carve_workspace(a, workspace);
    }
    if (mode == 2 && a.ext_feat && host.ext_grid && !host.features_pregathered) {       // hash-grid rows of the tiles the first pass can decode
        GridArgs g = *host.ext_grid;
        g.ray_counts = a.ray_counts; g.ray_list = nullptr; g.ray_list_count = nullptr;
        mark(host, 0, st);
        mne_launch_hash_rows(g, 0, st);
        mark(host, 1, st);
    }
    if (mode == 3 && a.raw_in) {                           // backward of an earlier forward: the tiles the backward walks, then their decode
        int* need = a.heavy_list;                          // (workspace array of the training call's heavy-ray list: unused in this mode)
        MNE_LAUNCH(tile_need_kernel, (unsigned)((a.R + 3) / 4), 256, 0, st, a, need);
        a.tile_need = need;
    }
    {   // decode: every tile (mode 0), or the a-priori prefix of every ray
        RenderArgs d = a;
        if (mode == 0) { d.ray_counts = nullptr; d.prefix_default = 1 << 30; }
        if (mode == 3) d.raw = nullptr;                    // raw of the forward call stays untouched
        // Training launches: the a-priori tiles' plane features either come pre-gathered (gather_kernel, then decode_kernel<PRE> reads the
        // rows back) or are gathered INLINE by the decode waves -- north_star's fused sample -> gather -> MLP form.  Round 2 measured the
        // split form 1.3 % ahead and kept it for everything; on the final tree (corner rows through buffer loads, balanced tile schedule) the
        // fused form wins without colour planes -- office0 driver form +2 %, apartment +2-4 %, INS Indoor +5 % -- and ties with them (48 corner
        // rows per sample in one wave): profiles/r06_inline_gather.txt.  MNE_FUSED_GATHER=0 builds the split form everywhere (A/B).
        const bool pre = mode >= 2 && (CP || !MNE_FUSED_GATHER);
        if (mode == 2 && !pre && !d.ext_feat) { mark(host, 0, st); mark(host, 1, st); }      // (timing marks of the gather that does not run: the decode's bracket starts here)
        if (int rc = launch_decode<HID, HIDC, CP>(d, st, host, pre)) return rc;
    }
    if (mode == 0) {
        const size_t clds = (size_t)4 * ((a.S + 3) & ~3) * 4 * sizeof(float);
        if (clds > MNE_LDS_MAX) return -4;
        if (clds > 64 * 1024) MNE_SET_MAX_LDS(composite_kernel, MNE_LDS_MAX);
        a.raw_in = a.raw;
        MNE_LAUNCH(composite_kernel, (a.R + 3) / 4, 256, clds, st, a);
        return 0;
    }
    if (mode == 1) {
        if constexpr (MNE_DECODE_FRAME && HID == 32 && HIDC == 32 && !CP) {
            // whole frames: the lean kernel at 12 waves per CU (see ray_frame_kernel)
            const size_t tab = table_bytes<HID, HIDC, CP>(0);
            const size_t per_wave = tile_wave_lds_bytes(1) + (size_t)((a.S + 3) & ~3) * sizeof(float);
            const int wpb = fit_waves<HID, HIDC, CP>(tab, per_wave, MNE_FRAME_WPB);
            const int min_tiles = a.frame_min_tiles > 0 ? a.frame_min_tiles : a.frame_min_tiles < 0 ? 0 : MNE_FRAME_MIN_TILES;
            if (wpb >= 1 && !a.raw_in && !a.ext_feat && a.raw && !a.ray_list && !a.dec_tiles &&
                (long long)a.R >= (long long)MNE_NUM_CU * wpb * min_tiles) {
                const size_t lds = tab + (size_t)wpb * per_wave;
                long long grid = ((long long)a.R + wpb - 1) / wpb;
                if (grid > MNE_NUM_CU) grid = MNE_NUM_CU;
                if (lds > 64 * 1024) MNE_SET_MAX_LDS((ray_frame_kernel<MNE_FRAME_WPB>), MNE_LDS_MAX);
                MNE_LAUNCH((ray_frame_kernel<MNE_FRAME_WPB>), (unsigned)grid, 64 * wpb, lds, st, a);
                return 0;
            }
        }
        return launch_ray<HID, HIDC, CP, 0>(a, st);
    }
    if (mode == 2) {
        // training: every ray in the lean kernel; the few it cannot resolve from the decoded prefix are finished by the
        // full kernel, driven by the deferred list (a handful of workgroups that leave at once when the list is empty)
        mark(host, 2, st);
        if (host.ev_after_decode) (void)hipEventRecord((hipEvent_t)host.ev_after_decode, st);      // the caller's bin pass 0 may start
        // long rays (more than MNE_HEAVY_NTILE tiles per ray, e.g. INS Indoor's 33): a ray with more than MNE_HEAVY_TILES
        // backward tiles is only composited by the ray kernels; heavy_bwd_kernel walks the listed rays' tiles afterwards
        {
            const int ntile_min = a.heavy_ntile > 0 ? a.heavy_ntile : MNE_HEAVY_NTILE;       // (tests lower both on small S)
            const int tiles = a.heavy_min > 0 ? a.heavy_min : MNE_HEAVY_TILES;
            a.heavy_min = (tiles > 0 && (a.S + TILE - 1) / TILE > ntile_min && a.heavy_ntile >= 0) ? tiles : 0;
        }
        if (int rc = launch_ray<HID, HIDC, CP, 4>(a, st)) return rc;
        if (ray_lds_cap(a) < a.S) {        // long rays: those whose decoded prefix exceeded the first pass's LDS, same kernel sized for S
            RenderArgs l = a;
            l.ray_list = a.long_list; l.ray_list_count = a.long_count; l.list_keeps_prefix = 1;
            if (int rc = launch_ray<HID, HIDC, CP, 4>(l, st)) return rc;
        }
        mark(host, 3, st);
        RenderArgs d = a;
        d.ray_list = a.defer_list; d.ray_list_count = a.defer_count;
        if (a.ext_feat && host.ext_grid && a.ray_counts) {    // the deferred rays' remaining hash-grid rows
            GridArgs g = *host.ext_grid;
            g.ray_counts = a.ray_counts; g.ray_list = a.defer_list; g.ray_list_count = a.defer_count;
            mne_launch_hash_rows(g, 0, st);
        }
        launch_decode<HID, HIDC, CP>(d, st, host);         // their remaining tiles, tile-parallel
        d.adapt_update = 1;                                // the last ray launch of the call decides the next call's schedule
        if (int rc = launch_ray<HID, HIDC, CP, 4>(d, st, MNE_LIST_PASS_BLOCKS / 2)) return rc;    // the same lean kernel: now every listed ray resolves
        if (a.heavy_min)
            if (int rc = launch_heavy<HID, HIDC, CP>(a, st)) return rc;
        mark(host, 4, st);
        if (a.bins.lists) {                                // list appends: the resolved rays' (pass 0) here unless the host runs
            if (!host.external_bin) launch_bin<CP>(a, 0, st);   // them beside the backward (mne_tile_bin on a second stream);
            launch_bin<CP>(a, 1, st);                      // the deferred rays' always here, right behind their ray pass
        }
        launch_plane_pass<HID, HIDC, CP>(a, st);           // scatter="atomics"
        mark(host, 5, st);
        return 0;
    }
    {
        const int rc = raygrad ? launch_ray<HID, HIDC, CP, 3>(a, st) : launch_ray<HID, HIDC, CP, 2>(a, st);
        if (rc) return rc;
        launch_plane_pass<HID, HIDC, CP>(a, st);
        return 0;
    }
}

template <int HID, int HIDC, bool CP>
static int launch_query(const QueryArgs& a, hipStream_t st) {
    const size_t lds = (size_t)RAYS_PER_WG * (TILE * 4 + (CP ? 2 : 1) * TILE * MNE_FS) * sizeof(float);
    const long long tiles = (a.n + TILE - 1) / TILE;
    MNE_LAUNCH((query_kernel<HID, HIDC, CP>), (unsigned)((tiles + RAYS_PER_WG - 1) / RAYS_PER_WG), 256, lds, st, a);
    return 0;
}

#define MNE_DISPATCH(sc, CALL, BAD)                                                           \
    do {                                                                                   \
        const bool cp_ = (sc).n_sets == 2;                                                 \
        if ((sc).hidden == 32 && (sc).hidden_color == 32) { if (cp_) { CALL(32, 32, true); } else { CALL(32, 32, false); } } \
        else if ((sc).hidden == 64 && (sc).hidden_color == 64) { if (cp_) { CALL(64, 64, true); } else { CALL(64, 64, false); } } \
        else return BAD;                                                                   \
    } while (0)

int mne_launch_pack(const mne_scene_t& sc, float* pk, hipStream_t st) {
#define CALL(H, HC, CPV) return launch_pack<H, HC, CPV>(sc, pk, st)
    MNE_DISPATCH(sc, CALL, -2);
#undef CALL
    return -2;
}

int mne_launch_render(const RenderArgs& a, int mode, void* workspace, const RenderHost& host, hipStream_t st) {
#define CALL(H, HC, CPV) return launch_render<H, HC, CPV>(a, mode, workspace, host, st)
    MNE_DISPATCH(a.sc, CALL, -2);
#undef CALL
    return -2;
}

int mne_launch_query(const QueryArgs& a, hipStream_t st) {
#define CALL(H, HC, CPV) return launch_query<H, HC, CPV>(a, st)
    MNE_DISPATCH(a.sc, CALL, -2);
#undef CALL
    return -2;
}

int mne_launch_decoder_update(const DecUpdateArgs& a, hipStream_t st) {
#define CALL(H, HC, CPV) return launch_decoder_update<H, HC, CPV>(a, st)
    MNE_DISPATCH(a.sc, CALL, -2);
#undef CALL
    return -2;
}

int mne_launch_loss_finalize(const LossArgs& a, hipStream_t st) {
    MNE_LAUNCH(loss_finalize_kernel, 1, 256, 0, st, a);
    return 0;
}

int mne_launch_loss_coef(const LossArgs& a, hipStream_t st) {
    MNE_LAUNCH(loss_coef_kernel, 1, 64, 0, st, a);
    return 0;
}

size_t mne_render_lds_bytes(const mne_scene_t& sc, int S, int bwd) {
    (void)sc; (void)bwd;
    return (size_t)4 * ((S + 3) & ~3) * 4 * sizeof(float);      // composite_kernel stages raw[S][4] of 4 rays
}

size_t mne_dims_packed(const mne_scene_t& sc) {
#define CALL(H, HC, CPV) return (size_t)ATab<H, HC, CPV>::TOTAL_RAYGRAD * 64
    MNE_DISPATCH(sc, CALL, 0);
#undef CALL
    return 0;
}
size_t mne_dims_tape_row(const mne_scene_t& sc) {
#define CALL(H, HC, CPV) return (size_t)DecDims<H, HC, CPV>::ROW
    MNE_DISPATCH(sc, CALL, 0);
#undef CALL
    return 0;
}
size_t mne_dims_tape_dfeat(const mne_scene_t& sc) {
#define CALL(H, HC, CPV) return (size_t)DecDims<H, HC, CPV>::T_DFEAT
    MNE_DISPATCH(sc, CALL, 0);
#undef CALL
    return 0;
}
size_t mne_dims_tape_pn(const mne_scene_t& sc) {
#define CALL(H, HC, CPV) return (size_t)DecDims<H, HC, CPV>::T_PN
    MNE_DISPATCH(sc, CALL, 0);
#undef CALL
    return 0;
}
size_t mne_dims_nparam(const mne_scene_t& sc) {
#define CALL(H, HC, CPV) return (size_t)DecDims<H, HC, CPV>::NPARAM
    MNE_DISPATCH(sc, CALL, 0);
#undef CALL
    return 0;
}
