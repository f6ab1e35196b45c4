// render.hip -- z sampling, fused tri-plane gather -> OneBlob -> MFMA tiny-MLP -> SDF compositing
// (forward) and its backward (loss gradients -> MFMA MLP backward -> plane-gradient scatter +
// decoder tape) for the MNE-SLAM mapping iteration on gfx950.
//
// Work decomposition: ONE WAVE PER RAY, four independent waves per 256-thread workgroup (they never
// barrier with each other; each owns a private LDS region and hands data between its own lanes
// with MNE_WAVE_SYNC).  Samples are processed in tiles of 32 points, two lanes per point:
//   pass 1  all S samples: coalesced gather (8 lanes x 16 B per 128-B corner row, 12 rows in flight
//           per lane) -> per-point feature rows in LDS -> OneBlob in registers -> MFMA chain
//           (mlp_mfma.h) -> raw (r,g,b,sdf) to global and to LDS.
//   pass 2  per-ray reductions with shuffles/ballot (lane per sample): first SDF sign change,
//           truncated sigmoid-product weights, rgb/depth/acc/var maps, loss partial sums.
//   pass 3  (backward) ballot/prefix-sum compaction of the samples that can receive gradient
//           (render window or loss masks), forward recompute on the compacted tiles, loss and
//           compositing gradients, MFMA backward chain, one tape row per sample for the decoder
//           weight-gradient GEMM, half-wave-per-row atomic scatter into the plane gradients.
//
// Reference semantics: model/scene_rep.py:28-53,183-230,351-419,475-611; model/decoder.py:110-175;
// model/utils.py:27-41,117-185 (include/mneslam_hip.h maps each entry point).
#include "mlp_mfma.h"
#include "mne_launch.h"

#define RAYS_PER_WG 4
#define TILE 32

// -----------------------------------------------------------------------------------------------
// z sampling + mask counts: one wave per ray, linspace tables staged in LDS once per workgroup
// -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_z_kernel(ZArgs a) {
    MNE_DYN_LDS(lds_raw);
    const int S = a.S, n_tab = a.has_d ? a.n_a + 2 * a.n_b : S;
    float* tab = (float*)lds_raw;                                   // [n_tab]
    for (int i = threadIdx.x; i < n_tab; i += blockDim.x) tab[i] = a.tables[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.x * RAYS_PER_WG + w;
    if (r >= a.R) return;                                           // whole wave leaves together
    float* vals = tab + ((n_tab + 3) & ~3) + w * ((S + 3) & ~3);    // [S] sorted samples of this ray
    float d = 0.0f;
    if (a.has_d) {
        d = a.target_d[r];
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
    int n_front = 0, n_center = 0, n_tail = 0, n_cofs = 0, n_cosdf = 0;
    for (int i = lane; i < S; i += MNE_WAVE) {
        float z = vals[i];
        if (a.perturb > 0.0f) {                                     // scene_rep.py:377-381
            const float zm = vals[i > 0 ? i - 1 : 0], zp = vals[i < S - 1 ? i + 1 : S - 1];
            const float lower = i > 0 ? 0.5f * (z + zm) : z;
            const float upper = i < S - 1 ? 0.5f * (zp + z) : z;
            const uint64_t e = (uint64_t)r * (uint64_t)S + (uint64_t)i;
            const float uu = a.u ? a.u[e] : philox_uniform(a.seed, a.offset, e);
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
        }
    }
    if (a.has_d) {
        int sums[5] = {n_front, n_center, n_tail, n_cofs, n_cosdf};
#pragma unroll
        for (int k = 0; k < 5; ++k)
            for (int m = 32; m >= 1; m >>= 1) sums[k] += __shfl_xor(sums[k], m);
        if (lane == 0) {                 // per-ray counts; summed by counts_reduce_kernel (no same-address atomics)
            int* rc = a.ray_counts + (size_t)r * MNE_N_COUNT;
            rc[MNE_C_VALID] = (d > 0.0f && d < a.depth_trunc) ? 1 : 0;          // scene_rep.py:570
            rc[MNE_C_E_FRONT] = sums[0]; rc[MNE_C_E_CENTER] = sums[1]; rc[MNE_C_E_TAIL] = sums[2];
            rc[MNE_C_CO_FS] = sums[3]; rc[MNE_C_CO_SDF] = sums[4]; rc[6] = 0; rc[7] = 0;
        }
    }
}

__global__ __launch_bounds__(256) void counts_reduce_kernel(ZArgs a) {
    __shared__ int part[256][MNE_N_COUNT];
    const int t = threadIdx.x;
    int acc[MNE_N_COUNT];
    for (int k = 0; k < MNE_N_COUNT; ++k) acc[k] = 0;
    for (int r = t; r < a.R; r += 256)
        for (int k = 0; k < MNE_N_COUNT; ++k) acc[k] += a.ray_counts[(size_t)r * MNE_N_COUNT + k];
    for (int k = 0; k < MNE_N_COUNT; ++k) part[t][k] = acc[k];
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (t < st)
            for (int k = 0; k < MNE_N_COUNT; ++k) part[t][k] += part[t + st][k];
        __syncthreads();
    }
    if (t < MNE_N_COUNT) a.counts[t] = part[0][t];
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
// render kernels (tile-parallel):
//   decode_kernel     one wave per 32-sample tile of a ray: gather -> OneBlob -> MFMA forward -> raw
//   composite_kernel  one wave per ray: SDF compositing, maps, loss partial sums; for the backward also
//                     the per-ray constants of the gradient and the compacted list of samples that can
//                     receive gradient (ballot + prefix popcount)
//   scan_kernel       exclusive prefix of the per-ray counts (deterministic tape order, no atomics)
//   backward_kernel   one wave per 32 contributing samples, packed ACROSS rays (full tiles): forward
//                     recompute, loss/compositing gradients, MFMA backward, tape row, scatter/append
// A ray is therefore never a serial chain of tiles: the batch exposes R*S/32 + P'/32 independent wave tasks.
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

// per-wave LDS of the tile kernels: pn[32][4] | feat[NSETS][32][FS]  (+ ray-gradient variant: dpos[32][64] | dpn[32][4])
__host__ __device__ inline size_t tile_wave_lds_bytes(int nsets, bool raygrad = false) {
    size_t b = (size_t)(TILE * 4 + nsets * TILE * MNE_FS) * sizeof(float);
    if (raygrad) b += (size_t)(TILE * 64 + TILE * 4) * sizeof(float);
    return b;
}

// ray_ctx[r][16]: constants of one ray's gradient, written by composite_kernel
enum { RC_DENOM = 0, RC_ZLIM = 1, RC_AQ = 2, RC_GR = 3, RC_GG = 4, RC_GB = 5, RC_GDEP = 6, RC_N = 16 };

#ifndef MAX_WPB
#define MAX_WPB 12
#endif
template <int HID, int HIDC, bool CP, bool ALDS>
__global__ __launch_bounds__(64 * MAX_WPB) void decode_kernel(RenderArgs a) {
    typedef ATab<HID, HIDC, CP> T;
    constexpr int NSETS = CP ? 2 : 1;
    constexpr int TAB_FLOATS = ALDS ? T::FWD_STEPS * 64 : 0;
    MNE_DYN_LDS(lds_raw);
    const int wpb = blockDim.x >> 6;
    if (ALDS) {                                            // stage the A tables: the only block-wide step
        float4* dst = (float4*)lds_raw;
        const float4* src = (const float4*)a.packed;
        for (int i = threadIdx.x; i < TAB_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const float* atab = ALDS ? (const float*)lds_raw : a.packed;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int S = a.S, ntile = (S + TILE - 1) / TILE;
    const long long ntask = (long long)a.R * ntile;
    float* pn = (float*)(lds_raw + (size_t)TAB_FLOATS * sizeof(float) + (size_t)wv * tile_wave_lds_bytes(NSETS));
    float* feat = pn + TILE * 4;
    const int pt = lane & 31, hf = lane >> 5;
    // persistent waves: a wave strides over the (ray, tile) tasks; nothing below is block-wide
    for (long long task = (long long)blockIdx.x * wpb + wv; task < ntask; task += (long long)gridDim.x * wpb) {
        const int r = (int)(task / ntile), c = (int)(task % ntile);
        const int i = c * TILE + pt;
        const bool valid = i < S;
        const float z = a.z_vals[(size_t)r * S + (valid ? i : S - 1)];
        float p[3], pnv[3], u[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = a.rays_o[r * 3 + k] + a.rays_d[r * 3 + k] * z;      // scene_rep.py:384
        point_coords(a.sc, p, pnv, u);
        MNE_WAVE_SYNC();                                       // previous task's LDS reads are done
        if (hf == 0) *(float4*)(pn + pt * 4) = make_float4(pnv[0], pnv[1], pnv[2], 0.0f);
        MNE_WAVE_SYNC();
        gather_chunk<NSETS, TILE>(a.sc, pn, feat, lane, a.dbg);
        MNE_WAVE_SYNC();
        float pos[24];
        oneblob_half(u, hf, pos);
        MlpState<HID, HIDC> st;
        if (!MNE_ABL(a.dbg, 8)) mlp_forward_mfma<HID, HIDC, CP>(feat + pt * MNE_FS, feat + TILE * MNE_FS + pt * MNE_FS, pos, atab, lane, st);
        else { st.rgb[0] = st.rgb[1] = st.rgb[2] = pos[0]; st.out[0] = pos[1]; }
        if (valid && hf == 0)                                  // rows 0..3 live in the lower half
            *(float4*)(a.raw + ((size_t)r * S + i) * 4) = make_float4(st.rgb[0], st.rgb[1], st.rgb[2], st.out[0]);
    }
}

// One wave per ray.  LDS per wave: raws[Spad][4].
template <bool BWD>
__device__ __forceinline__ void composite_ray(const RenderArgs& a, int r, int lane, float* raws) {
    const int S = a.S, Spad = (S + 3) & ~3;
    const bool has_t = a.target_d != nullptr;
    const float td = has_t ? a.target_d[r] : 0.0f;
    const float* zr = a.z_vals + (size_t)r * S;
    {
        const float4* src = (const float4*)(a.raw_in + (size_t)r * S * 4);
        for (int i = lane; i < S; i += MNE_WAVE) *(float4*)(raws + 4 * i) = src[i];
    }
    MNE_WAVE_SYNC();
    // first adjacent sign change (argmax of a 0/1 mask = first occurrence, 0 when none), scene_rep.py:195-199
    int first = 0;
    {
        const int nchunk = (S + MNE_WAVE - 1) / MNE_WAVE;
        for (int c = 0; c < nchunk; ++c) {
            const int i = c * MNE_WAVE + lane;
            const bool cr = (i < S - 1) && (raws[4 * (i + 1) + 3] * raws[4 * i + 3] < 0.0f);
            const unsigned long long m = __ballot(cr);
            if (m) { first = c * MNE_WAVE + __ffsll(m) - 1; break; }
        }
    }
    const float z_min = zr[first];
    const float z_lim = z_min + a.win_f;                                   // scene_rep.py:200
    float wsum = 0.0f;
    for (int i = lane; i < S; i += MNE_WAVE) {
        const float s = raws[4 * i + 3];
        const float wt = sigmoidf_(s / a.trunc_f) * sigmoidf_(-s / a.trunc_f);
        wsum += (zr[i] < z_lim) ? wt : 0.0f;
    }
    wsum = wave_sum(wsum);
    const float denom = wsum + 1e-8f;                                       // scene_rep.py:203
    float m_rgb[3] = {0.f, 0.f, 0.f}, m_depth = 0.f, m_acc = 0.f;
    float l_efs = 0.f, l_ec = 0.f, l_et = 0.f, l_cofs = 0.f, l_cosdf = 0.f;
    for (int i = lane; i < S; i += MNE_WAVE) {
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
        for (int i = lane; i < S; i += MNE_WAVE) {
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
    if (BWD) {
        float cf[MNE_N_LOSS];
#pragma unroll
        for (int k = 0; k < MNE_N_LOSS; ++k) cf[k] = a.coef ? a.coef[k] : 0.0f;
        float g_rgb[3], g_dep;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            g_rgb[k] = (a.target_rgb ? cf[MNE_L_RGB] * (m_rgb[k] - trgb[k]) : 0.0f) + (a.g_rgb ? a.g_rgb[r * 3 + k] : 0.0f);
        g_dep = (valid_ray ? cf[MNE_L_DEPTH] * (m_depth - td) : 0.0f) + (a.g_depth ? a.g_depth[r] : 0.0f);
        const float Aq = g_rgb[0] * m_rgb[0] + g_rgb[1] * m_rgb[1] + g_rgb[2] * m_rgb[2] + g_dep * m_depth;
        const bool use_e = cf[MNE_L_E_FS] != 0.f || cf[MNE_L_E_CENTER] != 0.f || cf[MNE_L_E_TAIL] != 0.f;
        const bool use_co = cf[MNE_L_CO_FS] != 0.f || cf[MNE_L_CO_SDF] != 0.f;
        // compaction of the samples that can receive a non-zero gradient (wave ballot + prefix popcount)
        unsigned short* list = a.clist + (size_t)r * Spad;
        int n_contrib = 0;
        const int nchunk = (S + MNE_WAVE - 1) / MNE_WAVE;
        for (int c = 0; c < nchunk; ++c) {
            const int i = c * MNE_WAVE + lane;
            bool f = false;
            if (i < S) {
                const float z = zr[i];
                const SampleMasks mk = sample_masks(z, td, has_t, a);
                f = (z < z_lim) || (use_e && (mk.e_front || mk.e_center || mk.e_tail)) ||
                    (use_co && (mk.co_fs || mk.co_sdf));
            }
            const unsigned long long m = __ballot(f);
            if (f) list[n_contrib + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
            n_contrib += __popcll(m);
        }
        if (lane == 0) {
            a.ccount[r] = MNE_ABL(a.dbg, 16) ? 0 : n_contrib;
            float* rc = a.ray_ctx + (size_t)r * RC_N;
            rc[RC_DENOM] = denom; rc[RC_ZLIM] = z_lim; rc[RC_AQ] = Aq;
            rc[RC_GR] = g_rgb[0]; rc[RC_GG] = g_rgb[1]; rc[RC_GB] = g_rgb[2]; rc[RC_GDEP] = g_dep;
        }
    }
}

// exclusive prefix sum of ccount[R] -> coffset[R+1], total -> tape_rows, first ray of every 32-row tile of
// the compacted list -> tile_ray; executed by ONE workgroup of NT threads (part = NT ints of LDS)
template <int NT>
__device__ __forceinline__ void scan_counts(const RenderArgs& a, int* part) {
    const int tid = threadIdx.x;
    const int per = (a.R + NT - 1) / NT;
    const int b0 = tid * per < a.R ? tid * per : a.R, b1 = (b0 + per < a.R) ? b0 + per : a.R;
    int s = 0;
    for (int i = b0; i < b1; ++i) s += a.ccount[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < NT; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;                        // exclusive prefix of this thread's chunk
    for (int i = b0; i < b1; ++i) {
        const int c0 = run, c1 = run + a.ccount[i];
        a.coffset[i] = c0;
        // ray i owns the 32-row tiles whose first row lies in [c0, c1): backward_kernel starts its search there
        for (int t = (c0 + TILE - 1) / TILE; t * TILE < c1; ++t) a.tile_ray[t] = i;
        run = c1;
    }
    if (tid == NT - 1) { a.coffset[a.R] = part[NT - 1]; *a.tape_rows = part[NT - 1]; }
}

// 4 rays per workgroup.  (Folding the prefix sum into the last workgroup to finish -- ticket counter plus
// agent-scope fences -- was measured: every workgroup's release fence writes its XCD's L2 back and the
// kernel went from 10 us to 54 us, so the scan stays a separate 5 us launch.)
template <bool BWD>
__global__ __launch_bounds__(256) void composite_kernel(RenderArgs a) {
    MNE_DYN_LDS(lds_raw);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (BWD && blockIdx.x == 0 && threadIdx.x == 0 && a.bins.spill_count) *a.bins.spill_count = 0;   // before any append of this call
    if (r < a.R) composite_ray<BWD>(a, r, lane, (float*)lds_raw + (size_t)wv * ((a.S + 3) & ~3) * 4);
}

__global__ __launch_bounds__(1024) void scan_kernel(RenderArgs a) {
    __shared__ int part[1024];
    scan_counts<1024>(a, part);
}

#ifndef MAX_WPB_BWD
#define MAX_WPB_BWD 8
#endif
template <int HID, int HIDC, bool CP, bool ALDS, bool RAYGRAD>
__global__ __launch_bounds__(64 * MAX_WPB_BWD) void backward_kernel(RenderArgs a) {
    typedef DecDims<HID, HIDC, CP> D;
    typedef ATab<HID, HIDC, CP> T;
    constexpr int NSETS = CP ? 2 : 1;
    constexpr int NT = HID / 32, NTC = HIDC / 32;
    constexpr int TAB_FLOATS = ALDS ? (RAYGRAD ? T::TOTAL_RAYGRAD : T::TOTAL) * 64 : 0;
    MNE_DYN_LDS(lds_raw);
    const int wpb = blockDim.x >> 6;
    const int total = a.coffset[a.R];
    if ((long long)blockIdx.x * wpb * TILE >= total) return;      // whole workgroup beyond the compacted list
    if (ALDS) {
        float4* dst = (float4*)lds_raw;
        const float4* src = (const float4*)a.packed;
        for (int i = threadIdx.x; i < TAB_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const float* atab = ALDS ? (const float*)lds_raw : a.packed;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned char* my = lds_raw + (size_t)TAB_FLOATS * sizeof(float) + (size_t)wv * tile_wave_lds_bytes(NSETS, RAYGRAD);
    float* pn = (float*)my;
    float* feat = pn + TILE * 4;
    float* dposL = feat + NSETS * TILE * MNE_FS;                  // RAYGRAD: [32][64] d OneBlob rows
    float* dpnL = dposL + TILE * 64;                              // RAYGRAD: [32][4]  d normalised point
    const int pt = lane & 31, hf = lane >> 5;
    const int S = a.S, Spad = (S + 3) & ~3;
    const bool has_t = a.target_d != nullptr;
    float cf[MNE_N_LOSS];
#pragma unroll
    for (int q = 0; q < MNE_N_LOSS; ++q) cf[q] = a.coef ? a.coef[q] : 0.0f;
    const long long ntile = ((long long)total + TILE - 1) / TILE;
    // persistent waves over the tiles of the compacted list; nothing below is block-wide
    for (long long tile = (long long)blockIdx.x * wpb + wv; tile < ntile; tile += (long long)gridDim.x * wpb) {
        const int k = (int)(tile * TILE) + pt;                    // position in the compacted list = tape row
        const bool valid = k < total;
        const int kk = valid ? k : total - 1;
        // owning ray = last r with coffset[r] <= kk.  The tile's first row belongs to tile_ray[tile]; the
        // following offsets are fetched with one coalesced load and scanned with wave-uniform reads
        // (a tile of 32 rows rarely spans more than two or three rays).
        int lo = a.tile_ray[tile];
        if (!MNE_ABL(a.dbg, 1024)) {
            const int r0 = lo;
            const int cnext = a.coffset[(r0 + 1 + lane < a.R) ? r0 + 1 + lane : a.R];     // coffset[R] = total > kk
            const int k_last = (int)(tile * TILE) + TILE - 1;
            bool open_end = true;
            for (int j = 0; j < MNE_WAVE; ++j) {
                const int v = __shfl(cnext, j);
                if (v > k_last) { open_end = false; break; }
                lo += (v <= kk) ? 1 : 0;
            }
            if (open_end) {                                       // > 64 rays (mostly empty ones) inside this tile
                int hi = a.R;
                if (lo < r0 + MNE_WAVE) hi = lo + 1;              // this lane's ray was already found
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (a.coffset[mid] <= kk) lo = mid; else hi = mid;
                }
            }
        }
        const int r = lo;
        const int i = MNE_ABL(a.dbg, 1024) ? kk % S : a.clist[(size_t)r * Spad + (kk - a.coffset[r])];
        const float z = a.z_vals[(size_t)r * S + i];
        const float td = has_t ? a.target_d[r] : 0.0f;
        const float* rc = a.ray_ctx + (size_t)r * RC_N;
        const float denom = rc[RC_DENOM], z_lim = rc[RC_ZLIM], Aq = rc[RC_AQ], g_dep = rc[RC_GDEP];
        const float g_rgb[3] = {rc[RC_GR], rc[RC_GG], rc[RC_GB]};
        float p[3], pnv[3], u[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) p[q] = a.rays_o[r * 3 + q] + a.rays_d[r * 3 + q] * z;
        point_coords(a.sc, p, pnv, u);
        MNE_WAVE_SYNC();                                          // previous tile's LDS reads are done
        if (hf == 0) *(float4*)(pn + pt * 4) = make_float4(pnv[0], pnv[1], pnv[2], 0.0f);
        MNE_WAVE_SYNC();
        gather_chunk<NSETS, TILE>(a.sc, pn, feat, lane, a.dbg);
        MNE_WAVE_SYNC();
        float* frow = feat + pt * MNE_FS;
        float* cfrow = feat + TILE * MNE_FS + pt * MNE_FS;
        float pos[24];
        oneblob_half(u, hf, pos);
        MlpState<HID, HIDC> st;
        if (!MNE_ABL(a.dbg, 512)) mlp_forward_mfma<HID, HIDC, CP>(frow, cfrow, pos, atab, lane, st);
        else {
            st.out = f32x16_zero(); st.rgb = f32x16_zero(); st.rgb[0] = pos[0];
            for (int t = 0; t < NT; ++t) st.h[t] = f32x16_zero();
            for (int t = 0; t < NTC; ++t) st.hc[t] = f32x16_zero();
        }
        // ---- d(total)/d(raw) of this point (both lanes of the pair compute the same values)
        const float4 rw = *(const float4*)(a.raw_in + ((size_t)r * S + i) * 4);
        const float s = rw.w;
        float ds = 0.0f, dc[3] = {0.f, 0.f, 0.f};
        if (valid) {
            if (z < z_lim) {
                const float pp = sigmoidf_(s / a.trunc_f), qq = sigmoidf_(-s / a.trunc_f);
                const float wt = pp * qq;
                const float w = wt / denom;
                const float sg[3] = {sigmoidf_(rw.x), sigmoidf_(rw.y), sigmoidf_(rw.z)};
                const float dLdw = g_rgb[0] * sg[0] + g_rgb[1] * sg[1] + g_rgb[2] * sg[2] + g_dep * z;
                ds += ((dLdw - Aq) / denom) * (wt * (qq - pp) / a.trunc_f);
#pragma unroll
                for (int q = 0; q < 3; ++q) dc[q] = g_rgb[q] * w * (sg[q] * (1.0f - sg[q]));
            }
            const SampleMasks mk = sample_masks(z, td, has_t, a);
            const float e_res = (z + s * a.e_T) - td, c_res = (z + s * a.win_f) - td;
            if (mk.e_front) ds += cf[MNE_L_E_FS] * (s - 1.0f);
            if (mk.e_center) ds += cf[MNE_L_E_CENTER] * e_res;
            if (mk.e_tail) ds += cf[MNE_L_E_TAIL] * e_res;
            if (mk.co_fs) ds += cf[MNE_L_CO_FS] * (s - 1.0f);
            if (mk.co_sdf) ds += cf[MNE_L_CO_SDF] * c_res;
        }
        // ---- tape: forward activations of this point (each lane writes the part it holds)
        float* row = a.tape + (size_t)kk * D::ROW;
        const bool tape_on = valid && !MNE_ABL(a.dbg, 2);
        if (tape_on) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                *(float4*)(row + D::T_X + hf * 32 + 4 * q) = *(const float4*)(frow + hf * 32 + 4 * q);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const float4 pv = make_float4(pos[4 * q], pos[4 * q + 1], pos[4 * q + 2], pos[4 * q + 3]);
                *(float4*)(row + D::T_X + MNE_FEAT + hf * 24 + 4 * q) = pv;
                *(float4*)(row + D::T_CIN + hf * 24 + 4 * q) = pv;
            }
            if (CP) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    *(float4*)(row + D::T_CIN + MNE_POS + hf * 32 + 4 * q) = *(const float4*)(cfrow + hf * 32 + 4 * q);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
                *(float4*)(row + D::T_CIN + D::CINB + 8 * q + 4 * hf) =
                    make_float4(st.out[4 * q], st.out[4 * q + 1], st.out[4 * q + 2], st.out[4 * q + 3]);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(float4*)(row + D::T_H + 32 * t + 8 * q + 4 * hf) =
                        make_float4(st.h[t][4 * q], st.h[t][4 * q + 1], st.h[t][4 * q + 2], st.h[t][4 * q + 3]);
#pragma unroll
            for (int t = 0; t < NTC; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(float4*)(row + D::T_HC + 32 * t + 8 * q + 4 * hf) =
                        make_float4(st.hc[t][4 * q], st.hc[t][4 * q + 1], st.hc[t][4 * q + 2], st.hc[t][4 * q + 3]);
            if (hf == 0) *(float4*)(row + D::T_DC) = make_float4(dc[0], dc[1], dc[2], 0.0f);
        }
        // ---- MFMA backward chain; d(feature) rows overwrite this point's LDS feature rows
        f32x16 dh[NT], dout, dhc[NTC];
        if (!MNE_ABL(a.dbg, 4096)) mlp_backward_mfma<HID, HIDC, CP>(st, ds, dc, atab, lane, dh, dout, dhc, frow, cfrow);
        else {
            dout = f32x16_zero();
            for (int t = 0; t < NT; ++t) dh[t] = f32x16_zero();
            for (int t = 0; t < NTC; ++t) dhc[t] = f32x16_zero();
        }
        if (tape_on) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                *(float4*)(row + D::T_DOUT + 8 * q + 4 * hf) = make_float4(dout[4 * q], dout[4 * q + 1], dout[4 * q + 2], dout[4 * q + 3]);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(float4*)(row + D::T_DH + 32 * t + 8 * q + 4 * hf) =
                        make_float4(dh[t][4 * q], dh[t][4 * q + 1], dh[t][4 * q + 2], dh[t][4 * q + 3]);
#pragma unroll
            for (int t = 0; t < NTC; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(float4*)(row + D::T_DHC + 32 * t + 8 * q + 4 * hf) =
                        make_float4(dhc[t][4 * q], dhc[t][4 * q + 1], dhc[t][4 * q + 2], dhc[t][4 * q + 3]);
        }
        if (RAYGRAD) {
            // d(total)/d(point) = through the OneBlob input + through the bilinear plane coordinates;
            // the rays of a tile differ, so each point adds its share to its ray with atomics ([R][3])
            float* dprow = dposL + pt * 64;
            mlp_backward_dpos<HID, HIDC, CP>(dh, dhc, atab, lane, dprow);
            MNE_WAVE_SYNC();
            gather_coord_grad<NSETS, TILE>(a.sc, pn, feat, dpnL, lane);
            MNE_WAVE_SYNC();
            float du[3];
            oneblob_half_backward(u, hf, dprow, du);
#pragma unroll
            for (int q = 0; q < 3; ++q) du[q] += __shfl_xor(du[q], 32);
            if (valid && hf == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float inv_bb = a.sc.bb_is_f64 ? (float)(1.0 / (a.sc.bb_hi[q] - a.sc.bb_lo[q]))
                                                        : 1.0f / ((float)a.sc.bb_hi[q] - (float)a.sc.bb_lo[q]);
                    const float dp = dpnL[pt * 4 + q] * (2.0f / (a.sc.bound_hi[q] - a.sc.bound_lo[q])) + du[q] * inv_bb;
                    if (a.d_rays_o) unsafeAtomicAdd(a.d_rays_o + r * 3 + q, dp);
                    if (a.d_rays_d) unsafeAtomicAdd(a.d_rays_d + r * 3 + q, z * dp);
                }
            }
        }
        MNE_WAVE_SYNC();
        const int n_here = total - (int)(tile * TILE);
        if (a.bins.lists) {
            // binned scatter: d(feature) + normalised point go to the tape row, and the sample is
            // appended to the list of every plane tile its 2x2 footprint touches (tile_adam.hip)
            if (tape_on) {
#pragma unroll
                for (int set = 0; set < NSETS; ++set)
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        *(float4*)(row + D::T_DFEAT + set * MNE_FEAT + hf * 32 + 4 * q) =
                            *(const float4*)(feat + set * TILE * MNE_FS + pt * MNE_FS + hf * 32 + 4 * q);
                if (hf == 0) *(float4*)(row + D::T_PN) = *(const float4*)(pn + pt * 4);
            }
            if (!MNE_ABL(a.dbg, 1)) {
                // One returning atomic per DISTINCT tile list per wave: lanes that append to the same
                // list are grouped with ballots and the group leader reserves the whole run of slots.
                // Three phases so that all reservations of a tile are in flight together: (A) grouping,
                // registers only; (B) the leaders' atomics, back to back; (C) slots and entry writes.
                constexpr int NQ = NSETS * 3 * 4;
                int want[NQ];
                unsigned meta[NQ];                                 // leader lane | rank << 8 | group size << 16
#pragma unroll
                for (int j = 0; j < NSETS * 3; ++j) {
                    const int pidx = 2 * j + hf;                   // planes in [set][orient][level] order
                    const int ori = (pidx % 6) / 2;
                    const mne_plane_t& pl = a.sc.plane[pidx / 6][ori][pidx % 2];
                    float gx, gy;
                    orient_coords(ori, pnv[0], pnv[1], pnv[2], gx, gy);
                    Bilin b;
                    bilin_setup(gx, gy, pl.h, pl.w, b);
                    const int ix1 = b.ix0 + 1 < pl.w ? b.ix0 + 1 : b.ix0, iy1 = b.iy0 + 1 < pl.h ? b.iy0 + 1 : b.iy0;
                    const int tx0 = b.ix0 / MNE_TILE, tx1 = ix1 / MNE_TILE, ty0 = b.iy0 / MNE_TILE, ty1 = iy1 / MNE_TILE;
                    const int base = a.bins.tile_base[pidx], ntx = a.bins.ntx[pidx];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int tx = (q & 1) ? tx1 : tx0, ty = (q & 2) ? ty1 : ty0;
                        const bool dup = ((q & 1) && tx1 == tx0) || ((q & 2) && ty1 == ty0);   // same tile again
                        const int w_ = (valid && !dup) ? base + ty * ntx + tx : -1;
                        unsigned long long todo = __ballot(w_ >= 0);
                        unsigned m_ = 0;
                        while (todo) {
                            const int leader = __ffsll(todo) - 1;
                            const int t = __shfl(w_, leader);
                            const unsigned long long same = __ballot(w_ == t);
                            if (w_ == t)
                                m_ = (unsigned)leader | ((unsigned)__popcll(same & ((1ull << lane) - 1ull)) << 8) |
                                     ((unsigned)__popcll(same) << 16);
                            todo &= ~same;
                        }
                        want[j * 4 + q] = w_;
                        meta[j * 4 + q] = m_;
                    }
                }
                int first[NQ];
#pragma unroll
                for (int e = 0; e < NQ; ++e) {
                    first[e] = 0;
                    if (want[e] >= 0 && (int)(meta[e] & 255u) == lane) first[e] = atomicAdd(a.bins.counts + want[e], (int)(meta[e] >> 16));
                }
                const unsigned trow = (unsigned)kk;
#pragma unroll
                for (int j = 0; j < NSETS * 3; ++j) {
                    const int pidx = 2 * j + hf;
                    const int ori = (pidx % 6) / 2;
                    const mne_plane_t& pl = a.sc.plane[pidx / 6][ori][pidx % 2];
                    float gx, gy;
                    orient_coords(ori, pnv[0], pnv[1], pnv[2], gx, gy);
                    Bilin b;
                    bilin_setup(gx, gy, pl.h, pl.w, b);
                    const int ix1 = b.ix0 + 1 < pl.w ? b.ix0 + 1 : b.ix0, iy1 = b.iy0 + 1 < pl.h ? b.iy0 + 1 : b.iy0;
                    const int tx0 = b.ix0 / MNE_TILE, tx1 = ix1 / MNE_TILE, ty0 = b.iy0 / MNE_TILE, ty1 = iy1 / MNE_TILE;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int e = j * 4 + q;
                        const int tx = (q & 1) ? tx1 : tx0, ty = (q & 2) ? ty1 : ty0;
                        const int f0 = __shfl(first[e], (int)(meta[e] & 255u));
                        if (want[e] >= 0) {
                            const int slot = f0 + (int)((meta[e] >> 8) & 255u);
                            unsigned* dst = nullptr;
                            if (slot < a.bins.cap) dst = a.bins.lists + ((size_t)want[e] * a.bins.cap + slot) * MNE_ENTRY_WORDS;
                            else {
                                const int sp = atomicAdd(a.bins.spill_count, 1);
                                if (sp < a.bins.spill_cap) { dst = a.bins.spill + (size_t)sp * MNE_SPILL_WORDS; *dst++ = (unsigned)want[e]; }
                                else atomicAdd(a.bins.dropped, 1);          // caller-sized spill area too small: reported, never silent
                            }
                            if (dst) {
                                dst[0] = trow;
                                dst[1] = (unsigned)(b.ix0 - tx * MNE_TILE + 1) | ((unsigned)(b.iy0 - ty * MNE_TILE + 1) << 8);
                                dst[2] = __float_as_uint(b.w00); dst[3] = __float_as_uint(b.w01);
                                dst[4] = __float_as_uint(b.w10); dst[5] = __float_as_uint(b.w11);
                            }
                        }
                    }
                }
            }
        } else if (a.sc.plane[0][0][0].grad) {                    // NULL: the caller wants no plane gradients (pose-only loops)
            scatter_chunk<NSETS, TILE>(a.sc, pn, feat, n_here < TILE ? n_here : TILE, lane, a.dbg);
        }
    }
}

// -----------------------------------------------------------------------------------------------
// loss scalars / coefficients (single small block; deterministic summation order)
// -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_finalize_kernel(LossArgs a) {
    __shared__ double part[256][MNE_N_LOSS];
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

__global__ void loss_coef_kernel(LossArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
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
    gather_chunk<NSETS, TILE>(a.sc, pn, feat, lane);
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
#define MNE_LDS_MAX (160 * 1024)
#define MNE_NUM_CU 256
template <int HID, int HIDC, bool CP> struct WgShape {
    static constexpr bool ALDS = !(HID == 64 && CP);
};

template <int HID, int HIDC, bool CP>
static size_t tile_lds_total(bool bwd, int wpb, bool raygrad = false) {
    typedef WgShape<HID, HIDC, CP> W;
    typedef ATab<HID, HIDC, CP> T;
    const size_t tab = W::ALDS ? (size_t)(raygrad ? T::TOTAL_RAYGRAD : bwd ? T::TOTAL : T::FWD_STEPS) * 64 * sizeof(float) : 0;
    return tab + (size_t)wpb * tile_wave_lds_bytes(CP ? 2 : 1, raygrad);
}

template <int HID, int HIDC, bool CP>
static int choose_wpb(bool bwd, bool raygrad, int max_wpb) {
    int fit = 0;
    for (int k = 1; k <= max_wpb; ++k)
        if (tile_lds_total<HID, HIDC, CP>(bwd, k, raygrad) <= MNE_LDS_MAX) fit = k;
    return fit;
}

static size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
size_t mne_render_workspace(int R, int S) {
    const size_t Spad = (size_t)((S + 3) & ~3);
    return align16((size_t)R * RC_N * sizeof(float)) + align16((size_t)R * sizeof(int)) +
           align16((size_t)(R + 1) * sizeof(int)) + align16((size_t)R * Spad * sizeof(unsigned short)) +
           align16(((size_t)R * Spad / TILE + 2) * sizeof(int));
}
static void carve_workspace(RenderArgs& a, void* ws) {
    unsigned char* p = (unsigned char*)ws;
    const size_t Spad = (size_t)((a.S + 3) & ~3);
    a.ray_ctx = (float*)p; p += align16((size_t)a.R * RC_N * sizeof(float));
    a.ccount = (int*)p; p += align16((size_t)a.R * sizeof(int));
    a.coffset = (int*)p; p += align16((size_t)(a.R + 1) * sizeof(int));
    a.clist = (unsigned short*)p; p += align16((size_t)a.R * Spad * sizeof(unsigned short));
    a.tile_ray = (int*)p;
}

int mne_launch_sample_z(const ZArgs& a, hipStream_t st) {
    const int n_tab = a.has_d ? a.n_a + 2 * a.n_b : a.S;
    const size_t lds = (size_t)(((n_tab + 3) & ~3) + RAYS_PER_WG * ((a.S + 3) & ~3)) * sizeof(float);
    MNE_LAUNCH(sample_z_kernel, (a.R + RAYS_PER_WG - 1) / RAYS_PER_WG, 256, lds, st, a);
    if (a.has_d) MNE_LAUNCH(counts_reduce_kernel, 1, 256, 0, st, a);
    return 0;
}

template <int HID, int HIDC, bool CP>
static int launch_pack(const mne_scene_t& sc, float* pk, hipStream_t st) {
    const int n = ATab<HID, HIDC, CP>::TOTAL_RAYGRAD * 64;
    MNE_LAUNCH((pack_decoder_kernel<HID, HIDC, CP>), (n + 255) / 256, 256, 0, st, sc, pk);
    return 0;
}

template <int HID, int HIDC, bool CP>
static int launch_render(RenderArgs a, int pass1, int bwd, void* workspace, hipStream_t st) {
    typedef WgShape<HID, HIDC, CP> W;
    const bool raygrad = a.d_rays_o != nullptr || a.d_rays_d != nullptr;
    if (raygrad && !(bwd && !pass1)) return -5;
    if (!pass1 && !bwd) return -1;
    if (pass1) {                                           // raw = decoder(points of every sample)
        const int wpb = choose_wpb<HID, HIDC, CP>(false, false, MAX_WPB);
        if (wpb < 1) return -4;
        const size_t lds = tile_lds_total<HID, HIDC, CP>(false, wpb);
        if (lds > 64 * 1024)        // gfx950 has 160 KiB of LDS per CU; above 64 KiB HIP wants an opt-in
            MNE_SET_MAX_LDS((decode_kernel<HID, HIDC, CP, W::ALDS>), MNE_LDS_MAX);
        const long long ntask = (long long)a.R * ((a.S + TILE - 1) / TILE);
        long long grid = (ntask + wpb - 1) / wpb;
        if (grid > MNE_NUM_CU) grid = MNE_NUM_CU;
        MNE_LAUNCH((decode_kernel<HID, HIDC, CP, W::ALDS>), (unsigned)grid, 64 * wpb, lds, st, a);
        a.raw_in = a.raw;
    }
    const size_t clds = (size_t)4 * ((a.S + 3) & ~3) * 4 * sizeof(float);
    if (clds > MNE_LDS_MAX) return -4;
    if (clds > 64 * 1024) {
        MNE_SET_MAX_LDS((composite_kernel<false>), MNE_LDS_MAX);
        MNE_SET_MAX_LDS((composite_kernel<true>), MNE_LDS_MAX);
    }
    if (!bwd) {
        MNE_LAUNCH((composite_kernel<false>), (a.R + 3) / 4, 256, clds, st, a);
        return 0;
    }
    if (!workspace) return -6;
    carve_workspace(a, workspace);
    MNE_LAUNCH((composite_kernel<true>), (a.R + 3) / 4, 256, clds, st, a);
    MNE_LAUNCH(scan_kernel, 1, 1024, 0, st, a);
    const int wpb = choose_wpb<HID, HIDC, CP>(true, raygrad, MAX_WPB_BWD);
    if (wpb < 1) return -4;
    const size_t lds = tile_lds_total<HID, HIDC, CP>(true, wpb, raygrad);
    const long long ntile = ((long long)a.R * a.S + TILE - 1) / TILE;       // upper bound; the kernel reads the real count
    long long grid = (ntile + wpb - 1) / wpb;
    if (grid > MNE_NUM_CU) grid = MNE_NUM_CU;
    if (raygrad) {
        if (lds > 64 * 1024) MNE_SET_MAX_LDS((backward_kernel<HID, HIDC, CP, W::ALDS, true>), MNE_LDS_MAX);
        MNE_LAUNCH((backward_kernel<HID, HIDC, CP, W::ALDS, true>), (unsigned)grid, 64 * wpb, lds, st, a);
    } else {
        if (lds > 64 * 1024) MNE_SET_MAX_LDS((backward_kernel<HID, HIDC, CP, W::ALDS, false>), MNE_LDS_MAX);
        MNE_LAUNCH((backward_kernel<HID, HIDC, CP, W::ALDS, false>), (unsigned)grid, 64 * wpb, lds, st, a);
    }
    return 0;
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

int mne_launch_render(const RenderArgs& a, int pass1, int bwd, void* workspace, hipStream_t st) {
#define CALL(H, HC, CPV) return launch_render<H, HC, CPV>(a, pass1, bwd, workspace, st)
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
