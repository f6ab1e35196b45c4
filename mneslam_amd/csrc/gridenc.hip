// gridenc.hip -- multiresolution hash / dense grid encoding (the tinycudann replacement surface behind
// get_encoder('HashGrid' | 'dense'), model/encodings.py:13-46; spec: oracle/hashgrid.py, parity unpinned).
//
// Pure HBM gather/scatter: per point and level, 8 corners x F floats.  One thread per (point, level),
// level-major blocks so that a workgroup's lookups stay inside one level's table (the coarse levels are
// L2-resident; the hashed fine levels are random 8-byte reads -- the latency is covered by occupancy:
// 20 VGPRs, 8 waves/SIMD).  Backward scatters with global_atomic_add_f32 (F*8 per point-level).
#include "mne_device.h"
#include "mne_launch.h"

__device__ __forceinline__ uint32_t grid_index(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t res, uint32_t size, bool dense) {
    uint32_t idx;
    if (dense) idx = cx + cy * res + cz * res * res;
    else idx = (cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u);
    return idx % size;
}

template <bool BWD>
__global__ __launch_bounds__(256) void grid_kernel(GridArgs a) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int level = blockIdx.y;
    if (t >= a.n) return;
    const float scale = a.scale[level];
    const uint32_t res = a.res[level], size = a.size[level], off = a.offset[level];
    const bool dense = (unsigned long long)res * res * res <= size;
    float frac[3];
    uint32_t cell[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float pos = fmaf(scale, a.x[t * 3 + d], 0.5f);
        const float fl = floorf(pos);
        cell[d] = (uint32_t)(int)fl;
        frac[d] = pos - fl;
    }
    const int F = a.n_features;
    float acc[MNE_GRID_MAX_F];
#pragma unroll
    for (int f = 0; f < MNE_GRID_MAX_F; ++f) acc[f] = BWD ? a.dout[t * a.out_dim + level * F + (f < F ? f : 0)] : 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float w = 1.0f;
        uint32_t cc[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if ((c >> d) & 1) { cc[d] = cell[d] + 1u; w *= frac[d]; }
            else { cc[d] = cell[d]; w *= 1.0f - frac[d]; }
        }
        const uint32_t idx = grid_index(cc[0], cc[1], cc[2], res, size, dense);
        if (!BWD && a.idx_out) a.idx_out[(t * a.n_levels + level) * 8 + c] = idx;
        const size_t base = ((size_t)off + idx) * F;
        if (BWD) {
            for (int f = 0; f < F; ++f) unsafeAtomicAdd(a.dparams + base + f, w * acc[f]);
        } else {
            for (int f = 0; f < F; ++f) acc[f] = fmaf(w, a.params[base + f], acc[f]);
        }
    }
    if (!BWD)
        for (int f = 0; f < F; ++f) a.out[t * a.out_dim + level * F + f] = acc[f];
}

// ---- fused form: the encoding of a ray batch's samples, written as (read from) rows of the render tape ----------
// One lane per (sample, level): a wave covers 4 samples x 16 levels, so the 16 levels x 2 features of a sample are one
// 128-byte line of its tape row (full-line stores; the level-major kernel above would write 8 bytes per line).  The
// sample position is recomputed from the ray and z exactly as decode_tile does (render.hip), x = the OneBlob input.
// Backward: d(table) += w * d(feature) with global_atomic_add_f32; rows past a ray's last backward tile were never
// written by ray_kernel and are skipped, all-zero rows (samples without gradient) issue no atomics.
template <bool BWD>
__global__ __launch_bounds__(256) void hash_rows_kernel(GridArgs a) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int level = (int)(gid & 15);
    const long long row = gid >> 4;
    if (row >= (long long)a.R * a.S || level >= a.n_levels) return;
    const int r = (int)(row / a.S), s = (int)(row % a.S);
    if (BWD && a.ray_tiles && s >= a.ray_tiles[r] * 32) return;
    const float z = a.z_vals[row];
    const float scale = a.scale[level];
    const uint32_t res = a.res[level], size = a.size[level], off = a.offset[level];
    const bool dense = (unsigned long long)res * res * res <= size;
    float frac[3];
    uint32_t cell[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float p = a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z;              // scene_rep.py:384
        const float x = unit_coord(p, a.bb_lo[d], a.bb_hi[d], a.bb_is_f64 != 0);
        const float pos = fmaf(scale, x, 0.5f);
        const float fl = floorf(pos);
        cell[d] = (uint32_t)(int)fl;
        frac[d] = pos - fl;
    }
    float* trow = a.tape + (size_t)row * a.row_stride;
    float2 acc = make_float2(0.f, 0.f);
    if (BWD) {
        acc = *(const float2*)(trow + a.col_d + level * 2);
        if (acc.x == 0.0f && acc.y == 0.0f) return;
    }
    const float2* table = (const float2*)a.params + off;
    float2 v[8];
    float w[8];
    uint32_t idx[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        w[c] = 1.0f;
        uint32_t cc[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if ((c >> d) & 1) { cc[d] = cell[d] + 1u; w[c] *= frac[d]; }
            else { cc[d] = cell[d]; w[c] *= 1.0f - frac[d]; }
        }
        idx[c] = grid_index(cc[0], cc[1], cc[2], res, size, dense);
        if (!BWD) v[c] = table[idx[c]];                                            // 8 independent 8-byte reads in flight
    }
    if (BWD) {
        float* g = a.dparams + ((size_t)off) * 2;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            unsafeAtomicAdd(g + (size_t)idx[c] * 2, w[c] * acc.x);
            unsafeAtomicAdd(g + (size_t)idx[c] * 2 + 1, w[c] * acc.y);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) { acc.x = fmaf(w[c], v[c].x, acc.x); acc.y = fmaf(w[c], v[c].y, acc.y); }   // corner order of grid_kernel
        *(float2*)(trow + a.col_x + level * 2) = acc;
    }
}

int mne_launch_hash_rows(const GridArgs& a, int bwd, hipStream_t st) {
    const long long n = (long long)a.R * a.S * 16;
    if (n <= 0) return 0;
    if (bwd) MNE_LAUNCH(hash_rows_kernel<true>, (unsigned)((n + 255) / 256), 256, 0, st, a);
    else MNE_LAUNCH(hash_rows_kernel<false>, (unsigned)((n + 255) / 256), 256, 0, st, a);
    return 0;
}

int mne_launch_grid(const GridArgs& a, int bwd, hipStream_t st) {
    const unsigned gx = (unsigned)((a.n + 255) / 256);
    if (bwd) hipLaunchOrEmu2D(grid_kernel<true>, gx, a.n_levels, 256, st, a);
    else hipLaunchOrEmu2D(grid_kernel<false>, gx, a.n_levels, 256, st, a);
    return 0;
}
