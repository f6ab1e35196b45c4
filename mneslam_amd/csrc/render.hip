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
// render kernel
// -----------------------------------------------------------------------------------------------
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

// per-wave LDS: pn[32][4] | feat[NSETS][32][FS] | raws[Spad][4] | list[Spad] (ushort)
// (+ ray-gradient variant: dpos[32][64] | dpn[32][4])
__host__ __device__ inline size_t render_wave_lds_bytes(int S, int nsets, bool raygrad = false) {
    const size_t Spad = (size_t)((S + 3) & ~3);
    size_t b = (size_t)(TILE * 4 + nsets * TILE * MNE_FS) * sizeof(float) + Spad * 4 * sizeof(float) + Spad * sizeof(unsigned short);
    b = (b + 15) & ~(size_t)15;
    if (raygrad) b += (size_t)(TILE * 64 + TILE * 4) * sizeof(float);
    return b;
}

// a.rpw = rays (= waves) per workgroup (chosen at launch so that one round of workgroups covers
// the batch); ALDS = the MFMA A-operand tables are staged in LDS once per
// workgroup (one ds_read_b32 per MFMA) instead of being re-read from global memory per MFMA.
#define MAX_RPW 10
template <int HID, int HIDC, bool CP, bool PASS1, bool BWD, bool ALDS, bool RAYGRAD = false>
__global__ __launch_bounds__(64 * MAX_RPW) void render_kernel(RenderArgs a) {
    const int RPW = a.rpw;                                 // waves (= rays) in this workgroup
    typedef DecDims<HID, HIDC, CP> D;
    typedef ATab<HID, HIDC, CP> T;
    constexpr int NSETS = CP ? 2 : 1;
    constexpr int NT = HID / 32, NTC = HIDC / 32;
    constexpr int TAB_FLOATS = ALDS ? (RAYGRAD ? T::TOTAL_RAYGRAD : BWD ? T::TOTAL : T::FWD_STEPS) * 64 : 0;
    MNE_DYN_LDS(lds_raw);
    if (ALDS) {                                            // stage the A tables: the only block-wide step
        float4* dst = (float4*)lds_raw;
        const float4* src = (const float4*)a.packed;
        for (int i = threadIdx.x; i < TAB_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const float* atab = ALDS ? (const float*)lds_raw : a.packed;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * RPW + wv;
    if (r >= a.R) return;                                  // whole wave leaves together; no block barriers below
    const int S = a.S;
    const int Spad = (S + 3) & ~3;
    unsigned char* my = lds_raw + (size_t)TAB_FLOATS * sizeof(float) + (size_t)wv * render_wave_lds_bytes(S, NSETS, RAYGRAD);
    float* pn = (float*)my;                                // [32][4]
    float* feat = pn + TILE * 4;                           // [NSETS][32][MNE_FS]
    float* raws = feat + NSETS * TILE * MNE_FS;            // [Spad][4]  (r,g,b,sdf)
    unsigned short* list = (unsigned short*)(raws + Spad * 4);   // [S] compacted sample ids (backward)
    float* dposL = (float*)(my + render_wave_lds_bytes(S, NSETS, false));   // RAYGRAD: [32][64] d OneBlob rows
    float* dpnL = dposL + TILE * 64;                                          // RAYGRAD: [32][4]  d normalised point
    const int pt = lane & 31, hf = lane >> 5;

    const float o[3] = {a.rays_o[r * 3 + 0], a.rays_o[r * 3 + 1], a.rays_o[r * 3 + 2]};
    const float dv[3] = {a.rays_d[r * 3 + 0], a.rays_d[r * 3 + 1], a.rays_d[r * 3 + 2]};
    const bool has_t = a.target_d != nullptr;
    const float td = has_t ? a.target_d[r] : 0.0f;
    const float* zr = a.z_vals + (size_t)r * S;

    if (!PASS1) {
        const float4* src = (const float4*)(a.raw_in + (size_t)r * S * 4);
        for (int i = lane; i < S; i += MNE_WAVE) *(float4*)(raws + 4 * i) = src[i];
    }

    // ------------------------------------------------------------------ pass 1: decode all samples
    if (PASS1) {
        const int ntile = (S + TILE - 1) / TILE;
#pragma unroll 1
        for (int c = 0; c < ntile; ++c) {
            const int i = c * TILE + pt;
            const bool valid = i < S;
            const float z = zr[valid ? i : S - 1];
            float p[3], pnv[3], u[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) p[k] = o[k] + dv[k] * z;          // scene_rep.py:384
            point_coords(a.sc, p, pnv, u);
            if (hf == 0) *(float4*)(pn + pt * 4) = make_float4(pnv[0], pnv[1], pnv[2], 0.0f);
            MNE_WAVE_SYNC();
            gather_chunk<NSETS, TILE>(a.sc, pn, feat, lane, a.dbg);
            MNE_WAVE_SYNC();
            float pos[24];
            oneblob_half(u, hf, pos);
            MlpState<HID, HIDC> st;
            if (!(a.dbg & 8)) mlp_forward_mfma<HID, HIDC, CP>(feat + pt * MNE_FS, feat + TILE * MNE_FS + pt * MNE_FS, pos, atab, lane, st);
            else { st.rgb[0] = st.rgb[1] = st.rgb[2] = pos[0]; st.out[0] = pos[1]; }
            if (valid && hf == 0) {                                        // rows 0..3 live in the lower half
                const float4 rw = make_float4(st.rgb[0], st.rgb[1], st.rgb[2], st.out[0]);
                *(float4*)(a.raw + ((size_t)r * S + i) * 4) = rw;
                *(float4*)(raws + 4 * i) = rw;
            }
            MNE_WAVE_SYNC();
        }
    }
    MNE_WAVE_SYNC();

    // ------------------------------------------------------------------ pass 2: compositing (lane per sample)
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

    // ------------------------------------------------------------------ pass 3: backward
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
        int n_contrib = 0;
        {
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
        }
        MNE_WAVE_SYNC();
        int tape_base = 0;
        if (lane == 0 && n_contrib > 0) tape_base = atomicAdd(a.tape_rows, n_contrib);
        tape_base = __shfl(tape_base, 0);
        float go[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f};       // RAYGRAD: d/d rays_o, d/d rays_d of this lane's points
        const int ntile = (a.dbg & 16) ? 0 : (n_contrib + TILE - 1) / TILE;
#pragma unroll 1
        for (int cc = 0; cc < ntile; ++cc) {
            const int k = cc * TILE + pt;
            const bool valid = k < n_contrib;
            const int i = list[valid ? k : n_contrib - 1];
            const float z = zr[i];
            float p[3], pnv[3], u[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) p[q] = o[q] + dv[q] * z;
            point_coords(a.sc, p, pnv, u);
            if (hf == 0) *(float4*)(pn + pt * 4) = make_float4(pnv[0], pnv[1], pnv[2], 0.0f);
            MNE_WAVE_SYNC();
            gather_chunk<NSETS, TILE>(a.sc, pn, feat, lane, a.dbg);
            MNE_WAVE_SYNC();
            float* frow = feat + pt * MNE_FS;
            float* cfrow = feat + TILE * MNE_FS + pt * MNE_FS;
            float pos[24];
            oneblob_half(u, hf, pos);
            MlpState<HID, HIDC> st;
            mlp_forward_mfma<HID, HIDC, CP>(frow, cfrow, pos, atab, lane, st);
            // ---- d(total)/d(raw) of this point (both lanes of the pair compute the same values)
            const float4 rw = *(const float4*)(raws + 4 * i);
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
            float* row = a.tape + (size_t)(tape_base + (valid ? k : 0)) * D::ROW;
            const bool tape_on = valid && !(a.dbg & 2);
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
            mlp_backward_mfma<HID, HIDC, CP>(st, ds, dc, atab, lane, dh, dout, dhc, frow, cfrow);
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
                // d(total)/d(point) = through the OneBlob input + through the bilinear plane coordinates
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
                        go[q] += dp;
                        gd[q] += z * dp;
                    }
                }
            }
            MNE_WAVE_SYNC();
            const int n_here = n_contrib - cc * TILE;
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
                if (!(a.dbg & 1)) {
                    // One returning atomic per DISTINCT tile per wave: lanes that append to the same list
                    // (consecutive samples of a ray mostly do) are grouped with ballots and the group
                    // leader reserves the whole run of slots.
                    const unsigned trow = (unsigned)(tape_base + (valid ? k : 0));
#pragma unroll
                    for (int j = 0; j < NSETS * 3; ++j) {
                        const int pidx = 2 * j + hf;                       // planes in [set][orient][level] order
                        const int set = pidx / 6, ori = (pidx % 6) / 2, lvl = pidx % 2;
                        const mne_plane_t& pl = a.sc.plane[set][ori][lvl];
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
                            const int want = (valid && !dup) ? base + ty * ntx + tx : -1;
                            unsigned long long todo = __ballot(want >= 0);
                            int slot = -1;
                            while (todo) {
                                const int leader = __ffsll(todo) - 1;
                                const int t = __shfl(want, leader);
                                const unsigned long long same = __ballot(want == t);
                                int first = 0;
                                if (lane == leader) first = atomicAdd(a.bins.counts + t, __popcll(same));
                                first = __shfl(first, leader);
                                if (want == t) slot = first + __popcll(same & ((1ull << lane) - 1ull));
                                todo &= ~same;
                            }
                            if (want >= 0) {
                                unsigned* dst = nullptr;
                                if (slot < a.bins.cap) dst = a.bins.lists + ((size_t)want * a.bins.cap + slot) * MNE_ENTRY_WORDS;
                                else {
                                    const int sp = atomicAdd(a.bins.spill_count, 1);
                                    if (sp < a.bins.spill_cap) { dst = a.bins.spill + (size_t)sp * MNE_SPILL_WORDS; *dst++ = (unsigned)want; }
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
            } else {
                scatter_chunk<NSETS, TILE>(a.sc, pn, feat, n_here < TILE ? n_here : TILE, lane, a.dbg);
            }
            MNE_WAVE_SYNC();
        }
        if (RAYGRAD) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { go[q] = wave_sum(go[q]); gd[q] = wave_sum(gd[q]); }
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (a.d_rays_o) a.d_rays_o[r * 3 + q] = go[q];
                    if (a.d_rays_d) a.d_rays_d[r * 3 + q] = gd[q];
                }
            }
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
// Workgroup shape: the A tables (staged in LDS except for the largest decoder, which reads them
// through L2) plus one private region per ray.  Rays per workgroup are chosen at launch: enough that
// ONE round of workgroups (<= 256, one per CU) covers the batch when the 160 KiB LDS allows it --
// a second, nearly empty round would double the kernel time -- otherwise as many as fit.
#define MNE_LDS_MAX (160 * 1024)
#define MNE_NUM_CU 256
template <int HID, int HIDC, bool CP> struct WgShape {
    static constexpr bool ALDS = !(HID == 64 && CP);
};

template <int HID, int HIDC, bool CP>
static size_t render_lds_total(int S, bool bwd, int rpw, bool raygrad = false) {
    typedef WgShape<HID, HIDC, CP> W;
    typedef ATab<HID, HIDC, CP> T;
    const size_t tab = W::ALDS ? (size_t)(raygrad ? T::TOTAL_RAYGRAD : bwd ? T::TOTAL : T::FWD_STEPS) * 64 * sizeof(float) : 0;
    return tab + (size_t)rpw * render_wave_lds_bytes(S, CP ? 2 : 1, raygrad);
}

template <int HID, int HIDC, bool CP>
static int choose_rpw(int R, int S, bool bwd, bool raygrad = false) {
    int fit = 0;
    for (int k = 1; k <= MAX_RPW; ++k)
        if (render_lds_total<HID, HIDC, CP>(S, bwd, k, raygrad) <= MNE_LDS_MAX) fit = k;
    if (fit == 0) return 0;
    const int want = (R + MNE_NUM_CU - 1) / MNE_NUM_CU;      // rays per CU for a single round
    return want <= fit ? (want < 1 ? 1 : want) : fit;
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
static int launch_render(RenderArgs a, int pass1, int bwd, hipStream_t st) {
    typedef WgShape<HID, HIDC, CP> W;
    const bool raygrad = a.d_rays_o != nullptr || a.d_rays_d != nullptr;
    if (raygrad && !(bwd && !pass1)) return -5;
    a.rpw = choose_rpw<HID, HIDC, CP>(a.R, a.S, bwd != 0, raygrad);
    if (a.rpw < 1) return -4;
    const size_t lds = render_lds_total<HID, HIDC, CP>(a.S, bwd != 0, a.rpw, raygrad);
    const int grid = (a.R + a.rpw - 1) / a.rpw;
    if (lds > 64 * 1024) {          // gfx950 has 160 KiB of LDS per CU; above 64 KiB HIP wants an opt-in
        MNE_SET_MAX_LDS((render_kernel<HID, HIDC, CP, true, false, W::ALDS>), MNE_LDS_MAX);
        MNE_SET_MAX_LDS((render_kernel<HID, HIDC, CP, false, true, W::ALDS>), MNE_LDS_MAX);
        MNE_SET_MAX_LDS((render_kernel<HID, HIDC, CP, true, true, W::ALDS>), MNE_LDS_MAX);
        MNE_SET_MAX_LDS((render_kernel<HID, HIDC, CP, false, true, W::ALDS, true>), MNE_LDS_MAX);
    }
    if (raygrad) {
        MNE_LAUNCH((render_kernel<HID, HIDC, CP, false, true, W::ALDS, true>), grid, 64 * a.rpw, lds, st, a);
        return 0;
    }
    if (pass1 && !bwd) MNE_LAUNCH((render_kernel<HID, HIDC, CP, true, false, W::ALDS>), grid, 64 * a.rpw, lds, st, a);
    else if (!pass1 && bwd) MNE_LAUNCH((render_kernel<HID, HIDC, CP, false, true, W::ALDS>), grid, 64 * a.rpw, lds, st, a);
    else if (pass1 && bwd) MNE_LAUNCH((render_kernel<HID, HIDC, CP, true, true, W::ALDS>), grid, 64 * a.rpw, lds, st, a);
    else return -1;
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

int mne_launch_render(const RenderArgs& a, int pass1, int bwd, hipStream_t st) {
#define CALL(H, HC, CPV) return launch_render<H, HC, CPV>(a, pass1, bwd, st)
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
#define CALL(H, HC, CPV) return render_lds_total<H, HC, CPV>(S, bwd != 0, 1)
    MNE_DISPATCH(sc, CALL, 0);
#undef CALL
    return 0;
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
