// render.hip -- z sampling, fused tri-plane gather -> OneBlob -> tiny MLP -> SDF compositing
// (forward), and its backward (loss gradients -> MLP backward -> plane-gradient scatter + decoder
// tape), for the MNE-SLAM mapping iteration on gfx950.
//
// Work decomposition: ONE WAVE (64 lanes, a 64-thread workgroup) PER RAY.
//   pass 1  all S samples in chunks of 64: coalesced gather (8 lanes x 16 B per 128-B corner row)
//           -> per-point feature rows in LDS -> lane-per-point OneBlob + MLP (weights via scalar
//           loads) -> raw[R][S][4].
//   pass 2  per-ray reductions with wave shuffles/ballot: first SDF sign change, truncated
//           sigmoid-product weights, rgb/depth/acc/var maps, per-ray loss partial sums.
//   pass 3  (backward) ballot/prefix-sum compaction of the samples that receive gradient
//           (render window or loss masks; SURVEY.md section 7 "early termination must be exact"),
//           forward recompute on the compacted samples, loss/compositing gradients, MLP backward,
//           one tape row per sample for the decoder weight-gradient GEMM, and half-wave-per-row
//           atomic scatter into the plane gradients.
//
// Reference semantics: model/scene_rep.py:28-53,183-230,351-419,475-611; model/decoder.py:143-175;
// model/utils.py:27-41,117-185 (see include/mneslam_hip.h for the per-entry-point mapping).
#include "mne_device.h"
#include "mne_launch.h"

// -----------------------------------------------------------------------------------------------
// z sampling + mask counts
// -----------------------------------------------------------------------------------------------

__global__ __launch_bounds__(64) void sample_z_kernel(ZArgs a) {
    MNE_DYN_LDS(lds_raw);
    float* vals = (float*)lds_raw;                 // [S] sorted samples of this ray
    const int lane = threadIdx.x, r = blockIdx.x, S = a.S;
    float d = 0.0f;
    if (a.has_d) {
        d = a.target_d[r];
        const float* uni = a.tables;
        const float* surf = a.tables + a.n_a;
        const float* inval = a.tables + a.n_a + a.n_b;
        const bool invalid = d <= 0.0f;            // scene_rep.py:365
        // stable merge of two ascending sequences by rank (== torch.sort of their concatenation)
        for (int e = lane; e < S; e += MNE_WAVE) {
            if (e < a.n_a) {
                const float v = uni[e];
                int lo = 0, hi = a.n_b;             // #b strictly below v
                while (lo < hi) {
                    int mid = (lo + hi) >> 1;
                    float bv = invalid ? inval[mid] : surf[mid] + d;
                    if (bv < v) lo = mid + 1; else hi = mid;
                }
                vals[e + lo] = v;
            } else {
                const int j = e - a.n_a;
                const float v = invalid ? inval[j] : surf[j] + d;
                int lo = 0, hi = a.n_a;             // #a at or below v
                while (lo < hi) {
                    int mid = (lo + hi) >> 1;
                    if (uni[mid] <= v) lo = mid + 1; else hi = mid;
                }
                vals[j + lo] = v;
            }
        }
    } else {
        for (int e = lane; e < S; e += MNE_WAVE) vals[e] = a.tables[e];
    }
    __syncthreads();
    int n_front = 0, n_center = 0, n_tail = 0, n_cofs = 0, n_cosdf = 0;
    for (int i = lane; i < S; i += MNE_WAVE) {
        float z = vals[i];
        if (a.perturb > 0.0f) {                     // scene_rep.py:377-381
            const float zm = vals[i > 0 ? i - 1 : 0], zp = vals[i < S - 1 ? i + 1 : S - 1];
            const float lower = i > 0 ? 0.5f * (z + zm) : z;
            const float upper = i < S - 1 ? 0.5f * (zp + z) : z;
            const uint64_t e = (uint64_t)r * (uint64_t)S + (uint64_t)i;
            const float uu = a.u ? a.u[e] : philox_uniform(a.seed, a.offset, e);
            z = lower + (upper - lower) * uu;
        }
        a.z_vals[(size_t)r * S + i] = z;
        if (a.has_d) {
            if (d > 0.0f) {                         // ESLAM masks, scene_rep.py:489-499 (rays with d>0, :589)
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
        if (lane == 0) {
            if (d > 0.0f && d < a.depth_trunc) atomicAdd(a.counts + MNE_C_VALID, 1);   // scene_rep.py:570
            if (sums[0]) atomicAdd(a.counts + MNE_C_E_FRONT, sums[0]);
            if (sums[1]) atomicAdd(a.counts + MNE_C_E_CENTER, sums[1]);
            if (sums[2]) atomicAdd(a.counts + MNE_C_E_TAIL, sums[2]);
            if (sums[3]) atomicAdd(a.counts + MNE_C_CO_FS, sums[3]);
            if (sums[4]) atomicAdd(a.counts + MNE_C_CO_SDF, sums[4]);
        }
    }
}

// -----------------------------------------------------------------------------------------------
// decoder packing (transposes so that the forward's inner loop reads contiguous weight rows)
// -----------------------------------------------------------------------------------------------
template <int HID, int HIDC, bool CP>
__global__ void pack_decoder_kernel(mne_scene_t sc, float* pk) {
    typedef DecDims<HID, HIDC, CP> D;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < MNE_IN1 * HID) { int k = t / HID, j = t % HID; pk[D::OFF_W1T + t] = sc.w_sdf0[j * MNE_IN1 + k]; }
    if (t < HID * MNE_OUT1) { int j = t / MNE_OUT1, m = t % MNE_OUT1; pk[D::OFF_W2T + t] = sc.w_sdf1[m * HID + j]; }
    if (t < D::CIN * HIDC) { int k = t / HIDC, j = t % HIDC; pk[D::OFF_V1T + t] = sc.w_col0[j * D::CIN + k]; }
    if (t < HIDC * 4) { int j = t / 4, c = t % 4; pk[D::OFF_V2T + t] = c < 3 ? sc.w_col1[c * HIDC + j] : 0.0f; }
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

template <int HID, int HIDC, bool CP, bool PASS1, bool BWD>
__global__ __launch_bounds__(64) void render_kernel(RenderArgs a) {
    typedef DecDims<HID, HIDC, CP> D;
    constexpr int NSETS = CP ? 2 : 1;
    MNE_DYN_LDS(lds_raw);
    const int lane = threadIdx.x, r = blockIdx.x, S = a.S;
    const int Spad = (S + 3) & ~3;
    float* pn = (float*)lds_raw;                     // [64][4]
    float* feat = pn + 64 * 4;                       // [NSETS][64][MNE_FS]
    float* zs = feat + NSETS * 64 * MNE_FS;          // [Spad]
    float* sdfs = zs + Spad;                         // [Spad]
    unsigned short* list = (unsigned short*)(sdfs + Spad);   // [S] compacted sample ids (backward)
    const mne_cptr pk = MNE_CPTR(a.packed);

    const float o[3] = {a.rays_o[r * 3 + 0], a.rays_o[r * 3 + 1], a.rays_o[r * 3 + 2]};
    const float dv[3] = {a.rays_d[r * 3 + 0], a.rays_d[r * 3 + 1], a.rays_d[r * 3 + 2]};
    const bool has_t = a.target_d != nullptr;
    const float td = has_t ? a.target_d[r] : 0.0f;
    const float* rawp = PASS1 ? a.raw : a.raw_in;

    for (int i = lane; i < S; i += MNE_WAVE) zs[i] = a.z_vals[(size_t)r * S + i];
    if (!PASS1)
        for (int i = lane; i < S; i += MNE_WAVE) sdfs[i] = rawp[((size_t)r * S + i) * 4 + 3];
    __syncthreads();

    // ------------------------------------------------------------------ pass 1: decode all samples
    if (PASS1) {
        const int nchunk = (S + MNE_WAVE - 1) / MNE_WAVE;
#pragma unroll 1
        for (int c = 0; c < nchunk; ++c) {
            const int i = c * MNE_WAVE + lane;
            const bool valid = i < S;
            const float z = zs[valid ? i : S - 1];
            float p[3], pnv[3], u[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) p[k] = o[k] + dv[k] * z;         // scene_rep.py:384
            point_coords(a.sc, p, pnv, u);
            pn[lane * 4 + 0] = pnv[0]; pn[lane * 4 + 1] = pnv[1]; pn[lane * 4 + 2] = pnv[2];
            __syncthreads();
            gather_chunk<NSETS>(a.sc, pn, feat, lane);
            __syncthreads();
            float pos[MNE_POS];
            oneblob16(u[0], pos); oneblob16(u[1], pos + 16); oneblob16(u[2], pos + 32);
            float h[HID], out[MNE_OUT1], hc[HIDC], rgbr[3];
            mlp_forward<HID, HIDC, CP>(feat + lane * MNE_FS, feat + 64 * MNE_FS + lane * MNE_FS, pos, pk, h, out, hc, rgbr);
            if (valid) {
                *(float4*)(a.raw + ((size_t)r * S + i) * 4) = make_float4(rgbr[0], rgbr[1], rgbr[2], out[0]);
                sdfs[i] = out[0];
            }
            __syncthreads();
        }
    }

    // ------------------------------------------------------------------ pass 2: compositing
    // first adjacent sign change (argmax of a 0/1 mask = first occurrence, 0 when none), scene_rep.py:195-199
    int first = 0;
    {
        const int nchunk = (S + MNE_WAVE - 1) / MNE_WAVE;
        for (int c = 0; c < nchunk; ++c) {
            const int i = c * MNE_WAVE + lane;
            const bool cr = (i < S - 1) && (sdfs[i + 1] * sdfs[i] < 0.0f);
            const unsigned long long m = __ballot(cr);
            if (m) { first = c * MNE_WAVE + __ffsll(m) - 1; break; }
        }
    }
    const float z_min = zs[first];
    const float z_lim = z_min + a.win_f;                                   // scene_rep.py:200
    float wsum = 0.0f;
    for (int i = lane; i < S; i += MNE_WAVE) {
        const float s = sdfs[i];
        const float wt = sigmoidf_(s / a.trunc_f) * sigmoidf_(-s / a.trunc_f);
        wsum += (zs[i] < z_lim) ? wt : 0.0f;
    }
    wsum = wave_sum(wsum);
    const float denom = wsum + 1e-8f;                                       // scene_rep.py:203
    float m_rgb[3] = {0.f, 0.f, 0.f}, m_depth = 0.f, m_acc = 0.f;
    float l_efs = 0.f, l_ec = 0.f, l_et = 0.f, l_cofs = 0.f, l_cosdf = 0.f;
    for (int i = lane; i < S; i += MNE_WAVE) {
        const float s = sdfs[i], z = zs[i];
        const float wt = sigmoidf_(s / a.trunc_f) * sigmoidf_(-s / a.trunc_f);
        const float w = ((z < z_lim) ? wt : 0.0f) / denom;
        const float4 rw = *(const float4*)(rawp + ((size_t)r * S + i) * 4);
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
            const float s = sdfs[i], z = zs[i];
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
        const float A = g_rgb[0] * m_rgb[0] + g_rgb[1] * m_rgb[1] + g_rgb[2] * m_rgb[2] + g_dep * m_depth;
        const bool use_e = cf[MNE_L_E_FS] != 0.f || cf[MNE_L_E_CENTER] != 0.f || cf[MNE_L_E_TAIL] != 0.f;
        const bool use_co = cf[MNE_L_CO_FS] != 0.f || cf[MNE_L_CO_SDF] != 0.f;
        // compaction of the samples that can receive a non-zero gradient
        int n_contrib = 0;
        {
            const int nchunk = (S + MNE_WAVE - 1) / MNE_WAVE;
            for (int c = 0; c < nchunk; ++c) {
                const int i = c * MNE_WAVE + lane;
                bool f = false;
                if (i < S) {
                    const float z = zs[i];
                    const SampleMasks mk = sample_masks(z, td, has_t, a);
                    f = (z < z_lim) || (use_e && (mk.e_front || mk.e_center || mk.e_tail)) ||
                        (use_co && (mk.co_fs || mk.co_sdf));
                }
                const unsigned long long m = __ballot(f);
                if (f) list[n_contrib + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
                n_contrib += __popcll(m);
            }
        }
        __syncthreads();
        int tape_base = 0;
        if (lane == 0 && n_contrib > 0) tape_base = atomicAdd(a.tape_rows, n_contrib);
        tape_base = __shfl(tape_base, 0);
        float go[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f};
        const int nck = (n_contrib + MNE_WAVE - 1) / MNE_WAVE;
#pragma unroll 1
        for (int cc = 0; cc < nck; ++cc) {
            const int k = cc * MNE_WAVE + lane;
            const bool valid = k < n_contrib;
            const int i = list[valid ? k : n_contrib - 1];
            const float z = zs[i];
            float p[3], pnv[3], u[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) p[q] = o[q] + dv[q] * z;
            point_coords(a.sc, p, pnv, u);
            pn[lane * 4 + 0] = pnv[0]; pn[lane * 4 + 1] = pnv[1]; pn[lane * 4 + 2] = pnv[2];
            __syncthreads();
            gather_chunk<NSETS>(a.sc, pn, feat, lane);
            __syncthreads();
            float* frow = feat + lane * MNE_FS;
            float* cfrow = feat + 64 * MNE_FS + lane * MNE_FS;
            float pos[MNE_POS];
            oneblob16(u[0], pos); oneblob16(u[1], pos + 16); oneblob16(u[2], pos + 32);
            float h[HID], out[MNE_OUT1], hc[HIDC], rgbr[3];
            mlp_forward<HID, HIDC, CP>(frow, cfrow, pos, pk, h, out, hc, rgbr);
            // ---- d(total)/d(raw) for this sample
            const float s = out[0];
            float ds = 0.0f, dc[3] = {0.f, 0.f, 0.f};
            if (valid) {
                if (z < z_lim) {
                    const float pp = sigmoidf_(s / a.trunc_f), qq = sigmoidf_(-s / a.trunc_f);
                    const float wt = pp * qq;
                    const float w = wt / denom;
                    const float sg[3] = {sigmoidf_(rgbr[0]), sigmoidf_(rgbr[1]), sigmoidf_(rgbr[2])};
                    const float dLdw = g_rgb[0] * sg[0] + g_rgb[1] * sg[1] + g_rgb[2] * sg[2] + g_dep * z;
                    ds += ((dLdw - A) / denom) * (wt * (qq - pp) / a.trunc_f);
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
            // ---- MLP backward (weights in nn.Linear layout [out][in], scalar loads)
            const mne_cptr W1 = MNE_CPTR(a.sc.w_sdf0), W2 = MNE_CPTR(a.sc.w_sdf1);
            const mne_cptr V1 = MNE_CPTR(a.sc.w_col0), V2 = MNE_CPTR(a.sc.w_col1);
            float dhc[HIDC];
#pragma unroll
            for (int j = 0; j < HIDC; ++j) {
                float t = V2[0 * HIDC + j] * dc[0];
                t = fmaf(V2[1 * HIDC + j], dc[1], t);
                t = fmaf(V2[2 * HIDC + j], dc[2], t);
                dhc[j] = hc[j] > 0.0f ? t : 0.0f;
            }
            float dout[MNE_OUT1];
            dout[0] = ds;
#pragma unroll
            for (int g = 0; g < MNE_GEO; ++g) dout[1 + g] = 0.0f;
#pragma unroll
            for (int j = 0; j < HIDC; ++j)
#pragma unroll
                for (int g = 0; g < MNE_GEO; ++g)
                    dout[1 + g] = fmaf(V1[j * D::CIN + D::CINB + g], dhc[j], dout[1 + g]);
            float dh[HID];
#pragma unroll
            for (int j = 0; j < HID; ++j) dh[j] = 0.0f;
#pragma unroll
            for (int m = 0; m < MNE_OUT1; ++m)
#pragma unroll
                for (int j = 0; j < HID; ++j) dh[j] = fmaf(W2[m * HID + j], dout[m], dh[j]);
#pragma unroll
            for (int j = 0; j < HID; ++j) dh[j] = h[j] > 0.0f ? dh[j] : 0.0f;
            // ---- tape row for the decoder weight-gradient GEMM
            if (valid) {
                float* row = a.tape + (size_t)(tape_base + k) * D::ROW;
#pragma unroll
                for (int q = 0; q < MNE_FEAT / 4; ++q) *(float4*)(row + D::T_X + 4 * q) = *(const float4*)(frow + 4 * q);
#pragma unroll
                for (int q = 0; q < MNE_POS / 4; ++q)
                    *(float4*)(row + D::T_X + MNE_FEAT + 4 * q) = make_float4(pos[4 * q], pos[4 * q + 1], pos[4 * q + 2], pos[4 * q + 3]);
#pragma unroll
                for (int q = 0; q < HID / 4; ++q) {
                    *(float4*)(row + D::T_H + 4 * q) = make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
                    *(float4*)(row + D::T_DH + 4 * q) = make_float4(dh[4 * q], dh[4 * q + 1], dh[4 * q + 2], dh[4 * q + 3]);
                }
#pragma unroll
                for (int q = 0; q < MNE_OUT1 / 4; ++q)
                    *(float4*)(row + D::T_DOUT + 4 * q) = make_float4(dout[4 * q], dout[4 * q + 1], dout[4 * q + 2], dout[4 * q + 3]);
                // colour-net input [pos | (colour features) | geo | 0-pad]
#pragma unroll
                for (int q = 0; q < MNE_POS / 4; ++q)
                    *(float4*)(row + D::T_CIN + 4 * q) = make_float4(pos[4 * q], pos[4 * q + 1], pos[4 * q + 2], pos[4 * q + 3]);
                if (CP) {
#pragma unroll
                    for (int q = 0; q < MNE_FEAT / 4; ++q)
                        *(float4*)(row + D::T_CIN + MNE_POS + 4 * q) = *(const float4*)(cfrow + 4 * q);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int b = 1 + 4 * q;      // geo index base (out[1..15]) + one zero pad
                    *(float4*)(row + D::T_CIN + D::CINB + 4 * q) =
                        make_float4(out[b], out[b + 1], out[b + 2], (b + 3 < MNE_OUT1) ? out[(b + 3) & 15] : 0.0f);
                }
#pragma unroll
                for (int q = 0; q < HIDC / 4; ++q) {
                    *(float4*)(row + D::T_HC + 4 * q) = make_float4(hc[4 * q], hc[4 * q + 1], hc[4 * q + 2], hc[4 * q + 3]);
                    *(float4*)(row + D::T_DHC + 4 * q) = make_float4(dhc[4 * q], dhc[4 * q + 1], dhc[4 * q + 2], dhc[4 * q + 3]);
                }
                *(float4*)(row + D::T_DC) = make_float4(dc[0], dc[1], dc[2], 0.0f);
            }
            // ---- d(feature): geometry planes via sdf-net layer 1, colour planes via colour-net layer 1
            {
                float dx[MNE_FEAT];
#pragma unroll
                for (int q = 0; q < MNE_FEAT; ++q) dx[q] = 0.0f;
#pragma unroll
                for (int j = 0; j < HID; ++j)
#pragma unroll
                    for (int q = 0; q < MNE_FEAT; ++q) dx[q] = fmaf(W1[j * MNE_IN1 + q], dh[j], dx[q]);
#pragma unroll
                for (int q = 0; q < MNE_FEAT / 4; ++q)
                    *(float4*)(frow + 4 * q) = make_float4(dx[4 * q], dx[4 * q + 1], dx[4 * q + 2], dx[4 * q + 3]);
            }
            if (CP) {
                float dx[MNE_FEAT];
#pragma unroll
                for (int q = 0; q < MNE_FEAT; ++q) dx[q] = 0.0f;
#pragma unroll
                for (int j = 0; j < HIDC; ++j)
#pragma unroll
                    for (int q = 0; q < MNE_FEAT; ++q) dx[q] = fmaf(V1[j * D::CIN + MNE_POS + q], dhc[j], dx[q]);
#pragma unroll
                for (int q = 0; q < MNE_FEAT / 4; ++q)
                    *(float4*)(cfrow + 4 * q) = make_float4(dx[4 * q], dx[4 * q + 1], dx[4 * q + 2], dx[4 * q + 3]);
            }
            __syncthreads();
            const int n_here = n_contrib - cc * MNE_WAVE;
            scatter_chunk<NSETS>(a.sc, pn, feat, n_here < MNE_WAVE ? n_here : MNE_WAVE, lane);
            __syncthreads();
        }
        (void)go; (void)gd;
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
// point queries (forward only): lane per point, same gather/MLP building blocks
// -----------------------------------------------------------------------------------------------

template <int HID, int HIDC, bool CP>
__global__ __launch_bounds__(64) void query_kernel(QueryArgs a) {
    constexpr int NSETS = CP ? 2 : 1;
    MNE_DYN_LDS(lds_raw);
    float* pn = (float*)lds_raw;
    float* feat = pn + 64 * 4;
    const int lane = threadIdx.x;
    const long long i = (long long)blockIdx.x * MNE_WAVE + lane;
    const bool valid = i < a.n;
    const long long ii = valid ? i : a.n - 1;
    const float p[3] = {a.pts[ii * 3 + 0], a.pts[ii * 3 + 1], a.pts[ii * 3 + 2]};
    float pnv[3], u[3];
    point_coords(a.sc, p, pnv, u);
    if (a.flags & MNE_QUERY_PTS_NORMALISED) { pnv[0] = p[0]; pnv[1] = p[1]; pnv[2] = p[2]; }
    pn[lane * 4 + 0] = pnv[0]; pn[lane * 4 + 1] = pnv[1]; pn[lane * 4 + 2] = pnv[2];
    __syncthreads();
    gather_chunk<NSETS>(a.sc, pn, feat, lane);
    __syncthreads();
    if (a.feat_out && valid) {
        for (int q = 0; q < MNE_FEAT / 4; ++q)
            *(float4*)(a.feat_out + i * MNE_FEAT + 4 * q) = *(const float4*)(feat + lane * MNE_FS + 4 * q);
    }
    if (a.raw || a.geo) {
        float pos[MNE_POS];
        oneblob16(u[0], pos); oneblob16(u[1], pos + 16); oneblob16(u[2], pos + 32);
        float h[HID], out[MNE_OUT1], hc[HIDC], rgbr[3];
        mlp_forward<HID, HIDC, CP>(feat + lane * MNE_FS, feat + 64 * MNE_FS + lane * MNE_FS, pos, MNE_CPTR(a.packed), h, out, hc, rgbr);
        if (valid) {
            if (a.raw) *(float4*)(a.raw + i * 4) = make_float4(rgbr[0], rgbr[1], rgbr[2], out[0]);
            if (a.geo)
                for (int g = 0; g < MNE_GEO; ++g) a.geo[i * MNE_GEO + g] = out[1 + g];
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
static size_t render_lds_bytes(int S, int nsets) {
    const int Spad = (S + 3) & ~3;
    return (size_t)(64 * 4 + nsets * 64 * MNE_FS + 2 * Spad) * sizeof(float) + (size_t)((S + 1) & ~1) * sizeof(unsigned short);
}

int mne_launch_sample_z(const ZArgs& a, hipStream_t st) {
    MNE_LAUNCH(sample_z_kernel, a.R, 64, (size_t)a.S * sizeof(float), st, a);
    return 0;
}

template <int HID, int HIDC, bool CP>
static int launch_pack(const mne_scene_t& sc, float* pk, hipStream_t st) {
    const int n = MNE_IN1 * HID > DecDims<HID, HIDC, CP>::CIN * HIDC ? MNE_IN1 * HID : DecDims<HID, HIDC, CP>::CIN * HIDC;
    MNE_LAUNCH((pack_decoder_kernel<HID, HIDC, CP>), (n + 255) / 256, 256, 0, st, sc, pk);
    return 0;
}

template <int HID, int HIDC, bool CP>
static int launch_render(const RenderArgs& a, int pass1, int bwd, hipStream_t st) {
    const size_t lds = render_lds_bytes(a.S, CP ? 2 : 1);
    if (pass1 && !bwd) MNE_LAUNCH((render_kernel<HID, HIDC, CP, true, false>), a.R, 64, lds, st, a);
    else if (!pass1 && bwd) MNE_LAUNCH((render_kernel<HID, HIDC, CP, false, true>), a.R, 64, lds, st, a);
    else if (pass1 && bwd) MNE_LAUNCH((render_kernel<HID, HIDC, CP, true, true>), a.R, 64, lds, st, a);
    else return -1;
    return 0;
}

template <int HID, int HIDC, bool CP>
static int launch_query(const QueryArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(64 * 4 + (CP ? 2 : 1) * 64 * MNE_FS) * sizeof(float);
    MNE_LAUNCH((query_kernel<HID, HIDC, CP>), (unsigned)((a.n + 63) / 64), 64, lds, st, a);
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

size_t mne_dims_packed(const mne_scene_t& sc) {
#define CALL(H, HC, CPV) return (size_t)DecDims<H, HC, CPV>::PACKED
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
size_t mne_dims_nparam(const mne_scene_t& sc) {
#define CALL(H, HC, CPV) return (size_t)DecDims<H, HC, CPV>::NPARAM
    MNE_DISPATCH(sc, CALL, 0);
#undef CALL
    return 0;
}
