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

// grid_index without the integer division: a hashed level's size is the power of two T; a dense index of a point inside the
// bounding box is below 2 * size (cell + 1 <= res: res * (1 + res + res^2) < 2 res^3); anything else (points outside the box:
// wrapped cells) takes the division.  Same value as grid_index for every input.
__device__ __forceinline__ uint32_t grid_index_fast(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t res, uint32_t size, bool dense) {
    uint32_t idx;
    if (dense) idx = cx + cy * res + cz * res * res;
    else idx = (cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u);
    if ((size & (size - 1u)) == 0u) return idx & (size - 1u);
    if (idx < size) return idx;
    if (idx - size < size) return idx - size;
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
        float x = a.x[t * 3 + d];
        if (a.x_is_world) x = unit_coord(x, a.bb_lo[d], a.bb_hi[d], a.bb_is_f64 != 0);
        const float pos = fmaf(scale, x, 0.5f);
        const float fl = floorf(pos);
        cell[d] = (uint32_t)(int)fl;
        frac[d] = pos - fl;
    }
    const int F = a.n_features;
    const long long ostr = a.out_stride ? a.out_stride : a.out_dim;
    float acc[MNE_GRID_MAX_F];
#pragma unroll
    for (int f = 0; f < MNE_GRID_MAX_F; ++f) acc[f] = BWD ? a.dout[t * ostr + level * F + (f < F ? f : 0)] : 0.0f;
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
        for (int f = 0; f < F; ++f) a.out[t * ostr + level * F + f] = acc[f];
}

// ---- fused form: the encoding of a ray batch's samples, read from / written as rows of the render tape -------------------
// hash_scatter_kernel (the plain atomic scatter, kept as the cross-check of mne_hash_scatter impl 1): one lane per
// (sample, level), d(table) += w * d(feature) with global_atomic_add_f32; rows past a ray's last backward tile were never
// written by ray_kernel and are skipped, all-zero rows (samples without gradient) issue no atomics.  The sample position
// is recomputed from the ray and z exactly as decode_tile does (render.hip), x = the OneBlob input.
__global__ __launch_bounds__(256) void hash_scatter_kernel(GridArgs a) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int level = (int)(gid & 15);
    const long long row = gid >> 4;
    if (row >= (long long)a.R * a.S || level >= a.n_levels) return;
    const int r = (int)(row / a.S), s = (int)(row % a.S);
    if (a.ray_tiles && s >= a.ray_tiles[r] * 32) return;
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
    const float2 acc = *(const float2*)(a.tape + (size_t)row * a.row_stride + a.col_d + level * 2);
    if (acc.x == 0.0f && acc.y == 0.0f) return;
    float* g = a.dparams + ((size_t)off) * 2;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float w = 1.0f;
        uint32_t cc[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if ((c >> d) & 1) { cc[d] = cell[d] + 1u; w *= frac[d]; }
            else { cc[d] = cell[d]; w *= 1.0f - frac[d]; }
        }
        const uint32_t idx = grid_index_fast(cc[0], cc[1], cc[2], res, size, dense);
        unsafeAtomicAdd(g + (size_t)idx * 2, w * acc.x);
        unsafeAtomicAdd(g + (size_t)idx * 2 + 1, w * acc.y);
    }
}

// ---- gather of a ray batch, level-major waves ----------------------------------------------------------------------------
// (Rounds 2-3 mapped a wave to 4 samples x 16 levels: every load instruction of a wave touched 64 different cache lines in
// 16 different level tables.)  Here a workgroup takes 64 CONSECUTIVE samples of one ray and a wave = those 64
// samples at ONE level (each thread: its sample at 4 levels, 32 independent 8-byte reads in flight): on the dense levels
// consecutive samples sit in the same or neighbouring cells, so the 64 lanes of a load fall into a handful of lines instead
// of 64 (the hashed levels have no locality either way).  The 16 x 2 features of a sample leave through an LDS transpose
// as full 128-byte lines of its tape row, as before.  Same arithmetic, same bits.
__global__ __launch_bounds__(256) void hash_gather_kernel(GridArgs a) {
    __shared__ float2 tile[64][17];                                  // [sample][level], padded: conflict-free both ways
    const int tid = threadIdx.x, sm = tid & 63, lq = tid >> 6;
    const int groups = (a.S + 63) / 64;
    // first pass: one workgroup per (ray, group of 64 samples); list pass (the deferred rays' remaining rows): a small grid
    // strides over (list entry, group) -- the list is empty or short in steady state
    const long long n_items = (a.ray_counts && a.ray_list) ? (long long)*a.ray_list_count * groups : (long long)a.R * groups;
    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
    int r = (int)(item / groups);
    const int s0 = (int)(item % groups) * 64;
    int s_lo = 0, s_hi = a.S;                                        // rows [s_lo, s_hi) of the ray are wanted
    if (a.ray_counts) {
        // under early termination: the first pass fills the tiles decode_kernel can reach (a-priori prefix + the resolver's
        // extension), the list pass the rest of the rays that were deferred
        if (a.ray_list) r = a.ray_list[r];
        const int ntile = (a.S + 31) / 32;
        const int need = a.ray_counts[(size_t)r * MNE_N_COUNT + MNE_C_NEED];
        int t = (need + 31) / 32;
        t = (t < 1 ? 1 : (t > ntile ? ntile : t)) + MNE_RESOLVER_MAX_EXT_FEAT;    // = prefix_tiles() of render.hip + the extension decode_kernel may take on caller-supplied features
        if (a.ray_list) s_lo = t * 32; else s_hi = t * 32 < a.S ? t * 32 : a.S;
    }
    if (s0 >= s_hi || s0 + 64 <= s_lo) continue;                      // (whole workgroup)
    const int s = s0 + sm;
    const bool in = s >= s_lo && s < s_hi;
    const long long row = (long long)r * a.S + (s < a.S ? s : a.S - 1);
    const float z = a.z_vals[row];
    float x[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float p = a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z;              // scene_rep.py:384
        x[d] = unit_coord(p, a.bb_lo[d], a.bb_hi[d], a.bb_is_f64 != 0);
    }
    float2 v[4][8];
    float w[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int level = lq + 4 * j;
        const bool live = in && level < a.n_levels;
        const int lv = level < a.n_levels ? level : 0;
        const float scale = a.scale[lv];
        const uint32_t res = a.res[lv], size = a.size[lv];
        const bool dense = (unsigned long long)res * res * res <= size;
        const float2* table = (const float2*)a.params + a.offset[lv];
        float frac[3];
        uint32_t cell[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float pos = fmaf(scale, x[d], 0.5f);
            const float fl = floorf(pos);
            cell[d] = (uint32_t)(int)fl;
            frac[d] = pos - fl;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float wc = 1.0f;
            uint32_t cc[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if ((c >> d) & 1) { cc[d] = cell[d] + 1u; wc *= frac[d]; }
                else { cc[d] = cell[d]; wc *= 1.0f - frac[d]; }
            }
            w[j][c] = wc;
            const uint32_t idx = grid_index_fast(cc[0], cc[1], cc[2], res, size, dense);
            v[j][c] = live ? table[idx] : make_float2(0.f, 0.f);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 8; ++c) { acc.x = fmaf(w[j][c], v[j][c].x, acc.x); acc.y = fmaf(w[j][c], v[j][c].y, acc.y); }   // corner order of grid_kernel
        tile[sm][lq + 4 * j] = acc;
    }
    __syncthreads();
    // 4 lanes x 32 B = the 128-byte feature line of one row
    const int rs = tid >> 2, part = tid & 3;
    const int so = s0 + rs;
    if (so >= s_lo && so < s_hi) {
        float* dst = a.tape + ((size_t)r * a.S + so) * a.row_stride + a.col_x + part * 8;
        const float2 t0 = tile[rs][part * 4], t1 = tile[rs][part * 4 + 1], t2 = tile[rs][part * 4 + 2], t3 = tile[rs][part * 4 + 3];
        *(float4*)dst = make_float4(t0.x, t0.y, t1.x, t1.y);
        *(float4*)(dst + 4) = make_float4(t2.x, t2.y, t3.x, t3.y);
    }
    __syncthreads();                                                  // the tile is reused by the next item
    }
}

// ---- ray gradients through the grid (R13 for the hash / dense grid model: pose alignment, mp_slam/mapper.py:388-408) ------
// d(total)/d(point) through the trilinear weights: per sample  sum_levels sum_corners (d(feature)_level . table[corner]) *
// d(weight_corner)/d(x) * scale_level,  x = (p - bb_lo) / (bb_hi - bb_lo)  (the floor in `cell` has no gradient: what autograd
// of the spec, oracle/hashgrid.py, computes).  One wave per RAY walks its backward rows 64 at a time (rows past the ray's last
// backward tile were never written; rows without gradient hold zeros) and ADDS the ray's sums to d_rays_o / d_rays_d, which
// already hold the OneBlob input's share (ray_kernel, MODE 3): one writer per ray, fixed order -- deterministic.
__global__ __launch_bounds__(256) void hash_raygrad_kernel(GridArgs a, float* d_rays_o, float* d_rays_d) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.R) return;
    int n_rows = a.ray_tiles[r] * 32;
    n_rows = n_rows < a.S ? n_rows : a.S;
    float go[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f};
    float inv_bb[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
        inv_bb[d] = a.bb_is_f64 ? (float)(1.0 / (a.bb_hi[d] - a.bb_lo[d])) : 1.0f / ((float)a.bb_hi[d] - (float)a.bb_lo[d]);
    for (int s0 = 0; s0 < n_rows; s0 += 64) {
        const int s = s0 + lane;
        if (s >= n_rows) continue;
        const long long row = (long long)r * a.S + s;
        const float z = a.z_vals[row];
        float x[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float p = a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z;              // scene_rep.py:384
            x[d] = unit_coord(p, a.bb_lo[d], a.bb_hi[d], a.bb_is_f64 != 0);
        }
        const float* drow = a.tape + (size_t)row * a.row_stride + a.col_d;
        float dx[3] = {0.f, 0.f, 0.f};
        for (int level = 0; level < a.n_levels; ++level) {
            const float2 g = *(const float2*)(drow + level * 2);
            if (g.x == 0.0f && g.y == 0.0f) continue;
            const float scale = a.scale[level];
            const uint32_t res = a.res[level], size = a.size[level];
            const bool dense = (unsigned long long)res * res * res <= size;
            const float2* table = (const float2*)a.params + a.offset[level];
            float frac[3];
            uint32_t cell[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float pos = fmaf(scale, x[d], 0.5f);
                const float fl = floorf(pos);
                cell[d] = (uint32_t)(int)fl;
                frac[d] = pos - fl;
            }
            float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t idx = grid_index_fast(cell[0] + (c & 1), cell[1] + ((c >> 1) & 1), cell[2] + ((c >> 2) & 1), res, size, dense);
                const float2 t = table[idx];
                const float dot = g.x * t.x + g.y * t.y;
                const float w0 = (c & 1) ? frac[0] : 1.0f - frac[0], w1 = (c & 2) ? frac[1] : 1.0f - frac[1], w2 = (c & 4) ? frac[2] : 1.0f - frac[2];
                acc[0] += dot * ((c & 1) ? 1.0f : -1.0f) * (w1 * w2);
                acc[1] += dot * ((c & 2) ? 1.0f : -1.0f) * (w0 * w2);
                acc[2] += dot * ((c & 4) ? 1.0f : -1.0f) * (w0 * w1);
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) dx[d] += acc[d] * scale;
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float dp = dx[d] * inv_bb[d];
            go[d] += dp;
            gd[d] += z * dp;
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) { go[d] = wave_sum(go[d]); gd[d] = wave_sum(gd[d]); }
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (d_rays_o) d_rays_o[r * 3 + d] += go[d];
            if (d_rays_d) d_rays_d[r * 3 + d] += gd[d];
        }
    }
}

int mne_launch_hash_raygrad(const GridArgs& a, float* d_rays_o, float* d_rays_d, hipStream_t st) {
    if (a.R <= 0) return 0;
    MNE_LAUNCH(hash_raygrad_kernel, (unsigned)((a.R + 3) / 4), 256, 0, st, a, d_rays_o, d_rays_d);
    return 0;
}

// ---- scatter with run reduction -------------------------------------------------------------------------------------
// The atomic units behind the L2 retire roughly one LINE operation per 60-100 ps chip-wide whatever the kernel does
// (profiles/r02_hash_scatter.txt: 33 M scattered float atomics = 1.87 ms; the tri-plane atomics path, 32 floats per line
// operation, lands on the same rate), and on the coarse levels thousands of samples hit the same few table entries.
// Consecutive samples of a ray walk through the grid cell by cell, so one wave takes 64 CONSECUTIVE samples of one ray
// at one level: lanes in the same cell form contiguous runs (ballot of "cell differs from my neighbour's"), the eight
// corner contributions of a run are summed with a segmented wave scan, and only the last lane of each run issues the
// sixteen atomics.  Level 0 (31 cm cells): one or two runs per wave instead of 64 lanes.
__global__ __launch_bounds__(256) void hash_scatter_runs_kernel(GridArgs a) {
    const int lane = threadIdx.x & 63;
    // blocks of 4 waves, level-minor: the 16 levels of a sample group run close together (its tape lines stay in L2)
    const int level = (int)(blockIdx.x % (unsigned)a.n_levels);
    const long long wave = (long long)(blockIdx.x / (unsigned)a.n_levels) * 4 + (threadIdx.x >> 6);
    const int chunks = (a.S + 63) / 64;
    if (wave >= (long long)a.R * chunks) return;
    const int r = (int)(wave / chunks), s0 = (int)(wave % chunks) * 64;
    int n_rows = a.ray_tiles[r] * 32;
    n_rows = n_rows < a.S ? n_rows : a.S;
    if (s0 >= n_rows) return;                                         // the whole wave leaves together
    const bool in = s0 + lane < n_rows;
    const long long row = (long long)r * a.S + (in ? s0 + lane : n_rows - 1);
    const float z = a.z_vals[row];
    const float scale = a.scale[level];
    const uint32_t res = a.res[level], size = a.size[level], off = a.offset[level];
    const bool dense = (unsigned long long)res * res * res <= size;
    float frac[3];
    uint32_t cell[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float p = a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z;
        const float x = unit_coord(p, a.bb_lo[d], a.bb_hi[d], a.bb_is_f64 != 0);
        const float pos = fmaf(scale, x, 0.5f);
        const float fl = floorf(pos);
        cell[d] = (uint32_t)(int)fl;
        frac[d] = pos - fl;
    }
    float2 g = *(const float2*)(a.tape + (size_t)row * a.row_stride + a.col_d + level * 2);
    if (!in) g = make_float2(0.f, 0.f);
    // runs of lanes in the same cell (rows past the end count as their own cell)
    const uint32_t ckey = in ? (cell[0] * 73856093u) ^ (cell[1] * 19349663u) ^ (cell[2] * 83492791u) : 0xffffffffu - (uint32_t)lane;
    const uint32_t pk = __shfl_up(ckey, 1);
    const uint32_t p0 = __shfl_up(cell[0], 1), p1 = __shfl_up(cell[1], 1), p2 = __shfl_up(cell[2], 1);
    const bool head = lane == 0 || pk != ckey || !in || p0 != cell[0] || p1 != cell[1] || p2 != cell[2];
    const unsigned long long heads = __ballot(head);
    const unsigned long long below = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    const int start = 63 - __clzll(below);                            // first lane of my run
    const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    const bool scan = heads != ~0ull;                                 // wave-uniform: some run is longer than one lane
    float* gt = a.dparams + (size_t)off * 2;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float w = 1.0f;
        uint32_t cc[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if ((c >> d) & 1) { cc[d] = cell[d] + 1u; w *= frac[d]; }
            else { cc[d] = cell[d]; w *= 1.0f - frac[d]; }
        }
        float vx = w * g.x, vy = w * g.y;
        if (scan) {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const float ux = __shfl_up(vx, d), uy = __shfl_up(vy, d);
                if (lane - d >= start) { vx += ux; vy += uy; }
            }
        }
        if (tail && in && (vx != 0.0f || vy != 0.0f)) {
            const uint32_t idx = grid_index(cc[0], cc[1], cc[2], res, size, dense);
            unsafeAtomicAdd(gt + (size_t)idx * 2, vx);
            unsafeAtomicAdd(gt + (size_t)idx * 2 + 1, vy);
        }
    }
}

// ---- table update without global atomics: slice-binned rows, exact fixed-point LDS accumulation, Adam fused ---------------
// Global float atomics retire at ~20 G/s chip-wide (previous section), and the samples of one iteration touch millions of
// distinct table entries, so no reduction in front of the atomics gets the scatter under ~0.5 ms.  The roles are turned
// around, as in tile_adam.hip: one workgroup OWNS a slice of HASH_SLICE entries of one level, sums its gradient in LDS and
// applies Adam to the slice -- no gradient buffer in HBM, the table and its moments are read and written once with plain
// coalesced accesses.  Round 4 (rounds 2-3: 16384-entry slices whose workgroups walked over ALL ~129 k backward rows of
// their level and added with ds_add_f32, 0.41 ms):
//   * BINNED ROWS.  hash_bin_kernel sorts the backward rows of a chunk of HASH_CHUNK packed rows by the slices their eight
//     corners fall into (counting sort in LDS: integer LDS atomics for the ranks, one scan, records = packed row ids written
//     slice after slice into the chunk's own segment, with the slice offsets beside it) -- no global atomics, no list
//     capacities, one pass.  A slice workgroup then reads, per chunk, exactly the rows that touch it.
//   * EXACT INTEGER ACCUMULATION.  ds_add_f32 retires 0.8 G lane-ops/s per CU, ds_add_u64 11.7 (profiles/
//     r03_lds_atomic_microbench.txt).  A level's addends w * d(feature) are bounded by its largest |d(feature)| (taken by
//     the pack kernel), so they are added as 64-bit fixed-point numbers with the power-of-two scale 2^(HASH_FIX_BITS - e),
//     max|g| < 2^e: an entry receives at most 8 * rows < 2^22 addends of magnitude < 2^39 -- no overflow -- and the sum of
//     the rounded addends is EXACT, hence independent of the order the rows arrive in: the update is bit-reproducible,
//     which the float-atomic forms were not.  Resolution: 2^-39 of the level's largest gradient (fp32 sums carry 2^-24
//     of the running sum); converted back to fp32 once per entry.
//   * A dense (coarse) level has few slices and every sample hits them: its chunks are split over `parts` workgroups per
//     slice, which add their non-zero sums into a 64-bit scratch with global integer atomics (exact, too);
//     hash_finish_kernel applies Adam to those levels.
#ifndef HASH_SLICE
#define HASH_SLICE 1024            // entries a slice holds at most: 2 x 8 B x 1024 = 16 KiB of LDS, eight workgroups per CU: a slice workgroup is a
                                   // chain of dependent round trips (offsets -> records -> rays -> Adam operands), so more, smaller ones
                                   // hide each other's latency -- 2048: 145 us stand-alone, 1024: 135, 4096: 217 (profiles/r04_hash_slice_size.txt)
#endif
#define HASH_SLICE_THREADS 256
#define HASH_CHUNK 1024            // packed rows per bin workgroup (2 per thread)
#define HASH_BIN_THREADS 512
#define HASH_RPT (HASH_CHUNK / HASH_BIN_THREADS)
#define HASH_REC_PER_ROW 8         // a row's corners fall into at most 8 slices (typically 4: the x-neighbours share a granule)
#define HASH_MAX_SLICES 4096       // per level: T <= 2^22
#define HASH_MAX_CHUNKS (2 * HASH_SLICE_THREADS)
#define HASH_FIX_BITS 39
#ifndef HASH_LEVEL_WGS
#define HASH_LEVEL_WGS 256         // workgroups a dense level gets at least (slices x parts)
#endif
// Slice geometry: contiguous slices of HASH_SLICE entries.  A hashed level has one workgroup per slice (the hash spreads the
// rows evenly).  A DENSE level's rows are concentrated -- every sample near the floor lands in the two or three slices that
// hold that z-layer of the lattice, and the coarse levels have only a handful of slices altogether -- so each of its slices
// is split into `parts` workgroups (every parts-th chunk of rows each) whose exact integer sums meet in a 64-bit scratch;
// hash_finish_kernel applies Adam to those levels.  (Measured alternatives, profiles/r04_hash_levels.txt: unbinned row walks
// for the coarse levels 180 us; interleaved slices + a segmented scan of same-cell lanes in front of the LDS atomics 2x slower
// than the contended atomics themselves.)
__host__ __device__ __forceinline__ int hash_slices_of(const GridArgs& a, int level) { return (int)((a.size[level] + HASH_SLICE - 1) / HASH_SLICE); }
__host__ __device__ __forceinline__ bool hash_level_dense(const GridArgs& a, int level) {
    return (unsigned long long)a.res[level] * a.res[level] * a.res[level] <= a.size[level];
}
__host__ __device__ __forceinline__ int hash_parts_of(const GridArgs& a, int level) {
    if (!hash_level_dense(a, level)) return 1;
    const int r = HASH_LEVEL_WGS / hash_slices_of(a, level);
    return r < 1 ? 1 : r;
}
__host__ __device__ __forceinline__ int hash_chunks_of(long long rows) { return (int)((rows + HASH_CHUNK - 1) / HASH_CHUNK); }


struct HashCorners { uint32_t idx[8]; float w[8]; };
// the eight corner entries (index within the level) and trilinear weights of grid input x at `level`: the expressions of
// grid_kernel / hash_gather_kernel (same index bits, same product order of the weights)
__device__ __forceinline__ void hash_corners(const GridArgs& a, int level, float4 x, bool dense, HashCorners& c) {
    const float scale = a.scale[level];
    const uint32_t res = a.res[level], size = a.size[level];
    float frac[3];
    uint32_t cell[3];
    const float xv[3] = {x.x, x.y, x.z};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float pos = fmaf(scale, xv[d], 0.5f);
        const float fl = floorf(pos);
        cell[d] = (uint32_t)(int)fl;
        frac[d] = pos - fl;
    }
    const float wx[2] = {1.0f - frac[0], frac[0]}, wy[2] = {1.0f - frac[1], frac[1]}, wz[2] = {1.0f - frac[2], frac[2]};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        c.idx[q] = grid_index_fast(cell[0] + (q & 1), cell[1] + ((q >> 1) & 1), cell[2] + ((q >> 2) & 1), res, size, dense);
        c.w[q] = (wx[q & 1] * wy[(q >> 1) & 1]) * wz[(q >> 2) & 1];
    }
}

// A record of the binned table update: one backward row as seen by ONE slice of one level -- its ray and z (the slice
// workgroup recomputes the grid input from the ray: 24 bytes per ray, cached), the level's d(feature) and the set of the
// row's eight corners that fall into the slice.  16 bytes = ONE store / load instruction per record (the bin kernel's
// scattered record stores are its cost: 8-byte pieces of a 24-byte record took three requests each), self-contained: the
// slice workgroup streams its records (contiguous per chunk) and never gathers from per-row arrays.
struct HashRecord { unsigned ray_mask; float z, gx, gy; };       // ray | corner mask << 24
static_assert(sizeof(HashRecord) == 16, "record layout");

// One workgroup per (level, chunk of HASH_CHUNK consecutive tape rows): counting sort of the chunk's (row, slice) records by
// slice.  Rows come straight from the iteration's buffers -- row = ray * S + sample; the sample's position is recomputed from
// the ray and z exactly as decode_tile does (render.hip), x = the OneBlob input; d(feature) of this level from the row's tape
// line; rows past a ray's last backward tile were never written by ray_kernel and are skipped, rows without gradient leave
// no record.
//   seg_off[level][chunk][0 .. n_slices]   start of every slice's records inside the chunk's segment (last = total)
//   records[level][chunk][..]              HashRecords, slice after slice
//   wgmax[level][chunk]                    largest |d(feature)| of the chunk: the slice workgroups derive the level's
//                                          fixed-point scale from these
// Ranks inside a slice come from LDS integer atomics; on a dense level ONE per run of consecutive rows with the same slice:
// consecutive rows are consecutive samples of a ray, which stay in one cell for dozens of samples -- per-row returning
// atomics on the same counter serialise (the appends of render.hip's bin_kernel know the problem).
// (Skipping the odd corners' slots on hashed levels -- x-neighbours share their slice -- is NOT valid: a sample outside the
// bounding box has wrapped cell coordinates, its x-neighbours then differ in every index bit; caught by the GPU parity test.)
__global__ __launch_bounds__(HASH_BIN_THREADS) void hash_bin_kernel(GridArgs a) {
    __shared__ unsigned hist[HASH_MAX_SLICES + 1];
    __shared__ unsigned wsum[HASH_BIN_THREADS / 64];
    __shared__ float red[HASH_BIN_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int level = blockIdx.x % a.n_levels, chunk = blockIdx.x / a.n_levels;
    const long long n_rows = (long long)a.R * a.S;
    const int ns = hash_slices_of(a, level);
    for (int i = tid; i <= ns; i += HASH_BIN_THREADS) hist[i] = 0u;
    const bool dense = hash_level_dense(a, level);
    const long long k_first = (long long)chunk * HASH_CHUNK;
    const int r_first = (int)(k_first / a.S);
    const unsigned s_first = (unsigned)(k_first - (long long)r_first * a.S);
    const float inv_S = 1.0f / (float)a.S;
    // this thread's rows, all loads first.  Row j of thread tid = chunk row j * HASH_BIN_THREADS + tid: the lanes of a wave
    // hold CONSECUTIVE rows.
    float2 g[HASH_RPT];
    float4 x[HASH_RPT];                                              // (x, y, z of the grid input, w = the sample's z along its ray)
    int ray[HASH_RPT];
    float m = 0.0f;
#pragma unroll
    for (int j = 0; j < HASH_RPT; ++j) {
        const long long k = (long long)chunk * HASH_CHUNK + j * HASH_BIN_THREADS + tid;
        g[j] = make_float2(0.f, 0.f); x[j] = make_float4(0.f, 0.f, 0.f, 0.f); ray[j] = 0;
        if (k < n_rows) {
            // (ray, sample) of row k without a 64-bit division per row: the chunk's first row is divided once (uniform),
            // the offset inside the chunk (< HASH_CHUNK + S) by a float reciprocal with one correction step each way
            const unsigned t = s_first + (unsigned)(j * HASH_BIN_THREADS + tid);
            unsigned dq = (unsigned)((float)t * inv_S);
            dq -= (dq * (unsigned)a.S > t) ? 1u : 0u;
            dq += ((dq + 1u) * (unsigned)a.S <= t) ? 1u : 0u;
            const int r = r_first + (int)dq, sidx = (int)(t - dq * (unsigned)a.S);
            ray[j] = r;
            if (sidx < a.ray_tiles[r] * 32) {
                g[j] = *(const float2*)(a.tape + (size_t)k * a.row_stride + a.col_d + level * 2);
                const float z = a.z_vals[k];
                float xv[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const float p = a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z;          // scene_rep.py:384
                    xv[d] = unit_coord(p, a.bb_lo[d], a.bb_hi[d], a.bb_is_f64 != 0);
                }
                x[j] = make_float4(xv[0], xv[1], xv[2], z);
                // NaN / Inf in EITHER component -> +Inf: poisons the level (fmaxf drops a NaN operand, so each is tested by itself)
                const float ax = fabsf(g[j].x), ay = fabsf(g[j].y);
                const float mj = (ax <= 3.0e38f && ay <= 3.0e38f) ? fmaxf(ax, ay) : __uint_as_float(0x7f800000u);
                m = fmaxf(m, mj);
            }
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
    if (lane == 0) red[wv] = m;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < HASH_BIN_THREADS / 64; ++w) m = fmaxf(m, red[w]);
        a.wgmax[(size_t)level * a.n_chunks + chunk] = m;
    }
    // slot q of a row = the slice of its corner q if no earlier corner has the same slice (then it carries the mask of all
    // corners in that slice), else empty
    static_assert(HASH_MAX_SLICES <= 4096 && HASH_CHUNK <= 4096, "slot word: slice (12 bits) | corner mask (8) | rank in the slice (12)");
    unsigned rec[HASH_RPT][HASH_REC_PER_ROW];                        // slice | corner mask << 12 | rank << 20   (mask 0: empty slot)
#pragma unroll
    for (int j = 0; j < HASH_RPT; ++j) {
        unsigned sl[8];
        const bool row_live = g[j].x != 0.0f || g[j].y != 0.0f;      // rows without gradient leave no record
        {
            HashCorners c;
            hash_corners(a, level, x[j], dense, c);
#pragma unroll
            for (int q = 0; q < 8; ++q) sl[q] = c.idx[q] / HASH_SLICE;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            bool dup = false;
            unsigned mask = 0u;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                if (p < q) dup = dup || sl[p] == sl[q];
                mask |= (sl[p] == sl[q]) ? (1u << p) : 0u;
            }
            const bool have = row_live && !dup;
            const unsigned key = have ? sl[q] : 0xffffffffu;
            rec[j][q] = have ? (sl[q] | (mask << 12)) : 0u;
            if (dense) {
                // one returning LDS atomic per RUN of consecutive lanes with the same slice in slot q
                const unsigned prev = __shfl_up(key, 1);
                const bool start = lane == 0 || key != prev;
                const unsigned long long sm = __ballot(start);
                const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
                const int leader = 63 - __clzll(sm & upto);
                const unsigned long long above = sm & ~upto;
                const int next = above ? __ffsll(above) - 1 : 64;
                unsigned base = 0u;
                if (have && leader == lane) base = atomicAdd(&hist[sl[q]], (unsigned)(next - leader));
                base = __shfl(base, leader);
                rec[j][q] |= (base + (unsigned)(lane - leader)) << 20;
            } else if (have) {
                // a hashed level scatters consecutive samples over unrelated slices: no runs to find (the search for them was
                // a third of this kernel's instructions, profiles/r04_hash_sq_counters.txt), no contention on the counters either
                rec[j][q] |= atomicAdd(&hist[sl[q]], 1u) << 20;
            }
        }
    }
    __syncthreads();
    // exclusive scan of the slice counts (ns <= 4096: up to 4 per thread), in place
    {
        unsigned v[HASH_MAX_SLICES / HASH_BIN_THREADS], sum = 0;
#pragma unroll
        for (int q = 0; q < HASH_MAX_SLICES / HASH_BIN_THREADS; ++q) {
            const int i = tid * (HASH_MAX_SLICES / HASH_BIN_THREADS) + q;
            v[q] = i < ns ? hist[i] : 0u;
            sum += v[q];
        }
        unsigned inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned u = __shfl_up(inc, d); if (lane >= d) inc += u; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < wv; ++w) base += wsum[w];
        unsigned ex = base + inc - sum;
#pragma unroll
        for (int q = 0; q < HASH_MAX_SLICES / HASH_BIN_THREADS; ++q) {
            const int i = tid * (HASH_MAX_SLICES / HASH_BIN_THREADS) + q;
            if (i < ns) hist[i] = ex;
            ex += v[q];
        }
        if (tid == HASH_BIN_THREADS - 1) hist[ns] = ex;             // total (every i >= ns contributes 0)
    }
    __syncthreads();
    unsigned* so = a.seg_off + a.seg_level[level] + (size_t)chunk * (ns + 1);
    for (int i = tid; i <= ns; i += HASH_BIN_THREADS) so[i] = hist[i];
    HashRecord* rc = (HashRecord*)a.records + ((size_t)level * a.n_chunks + chunk) * (size_t)(HASH_CHUNK * HASH_REC_PER_ROW);
#pragma unroll
    for (int j = 0; j < HASH_RPT; ++j) {
#pragma unroll
        for (int q = 0; q < HASH_REC_PER_ROW; ++q)
            if ((rec[j][q] >> 12) & 0xffu)
                *(float4*)(rc + hist[rec[j][q] & 0xfffu] + (rec[j][q] >> 20)) =
                    make_float4(__uint_as_float((unsigned)ray[j] | (((rec[j][q] >> 12) & 0xffu) << 24)), x[j].w, g[j].x, g[j].y);
    }
}

// rint(v * scale) as a two's-complement 64-bit integer (sums wrap correctly).  v * scale is exact in double (a float times
// a power of two) and below 2^40 in magnitude, so adding 2^52 + 2^51 rounds it to the nearest integer, ties to even -- what
// rint does -- and leaves that integer in the low mantissa bits: two double instructions and one 64-bit subtraction
// instead of rint + the emulated double -> int64 conversion.
__device__ __forceinline__ unsigned long long hash_fix(float v, double scale) {
    const double magic = 6755399441055744.0;                         // 2^52 + 2^51
    const double t = fma((double)v, scale, magic);
    return (unsigned long long)(__double_as_longlong(t) - __double_as_longlong(magic));
}

#define HASH_SLICE_UNROLL 4
// GROUPS: the batch has more than HASH_MAX_CHUNKS chunks (> 512 K rows: INS Indoor's 2150 x 1045, 8192 rays x 128) and the
// records come in several groups of chunks.  A template parameter, not a run-time loop bound: the loop around the record walk
// that round 5 added for such batches cost the headline shape (272 chunks, one group) 47 us of this kernel -- 164.6 -> 211.9 us,
// profiles/r04_hash_kernel_stats.txt vs r05_hash_kernel_stats.txt -- with the group loop compiled away it is the
// round-4 code again.
template <bool GROUPS>
__global__ __launch_bounds__(HASH_SLICE_THREADS) void hash_slice_adam_kernel(GridArgs a) {
    __shared__ unsigned long long acc[HASH_SLICE * 2];               // fixed-point gradient of this slice (2 x 8 B x 1024 entries = 16 KiB)
    __shared__ unsigned cstart[HASH_MAX_CHUNKS + 1];                 // prefix of the chunks' record counts (this slice, this part)
    __shared__ unsigned cbase[HASH_MAX_CHUNKS];                      //   ... and where each chunk's records of this slice begin
    __shared__ float wmaxs[HASH_SLICE_THREADS / 64];
    const int tid = threadIdx.x;
    int level = 0, k = blockIdx.x;
    while (level + 1 < a.n_levels && k >= hash_slices_of(a, level) * hash_parts_of(a, level)) {
        k -= hash_slices_of(a, level) * hash_parts_of(a, level);
        ++level;
    }
    const int n_part = hash_parts_of(a, level), part = k % n_part, slice = k / n_part;
    const int ns = hash_slices_of(a, level);
    const uint32_t size = a.size[level], off = a.offset[level];
    const uint32_t lo = (uint32_t)slice * HASH_SLICE;
    const uint32_t n_ent = size - lo < HASH_SLICE ? size - lo : HASH_SLICE;
    const bool dense = hash_level_dense(a, level);
    const int n_chunks = a.n_chunks;
    // the offsets of this slice's records in every chunk of this part (a split level: every n_part-th chunk) of the first
    // group of HASH_MAX_CHUNKS chunks, requested before anything else (one round trip); a batch of more than
    // HASH_MAX_CHUNKS * HASH_CHUNK = 512 K rows (INS Indoor's 2150 x 1045, or 8192 rays x 128) comes in further groups
    unsigned r0[HASH_MAX_CHUNKS / HASH_SLICE_THREADS], r1[HASH_MAX_CHUNKS / HASH_SLICE_THREADS];
    auto fetch_offsets = [&](int cg) {
#pragma unroll
        for (int j = 0; j < HASH_MAX_CHUNKS / HASH_SLICE_THREADS; ++j) {
            r0[j] = r1[j] = 0u;
            const int c = cg + tid + j * HASH_SLICE_THREADS;
            if (c < n_chunks && c % n_part == part) {
                const unsigned* so = a.seg_off + a.seg_level[level] + (size_t)c * (ns + 1) + slice;
                r0[j] = so[0]; r1[j] = so[1];
            }
        }
    };
    fetch_offsets(0);
    for (int i = tid; i < HASH_SLICE * 2; i += HASH_SLICE_THREADS) acc[i] = 0ull;
    // the level's fixed-point scale: max |d(feature)| over the chunks' maxima (every workgroup of the level computes the
    // same value; (slice 0, part 0) publishes it for hash_finish_kernel)
    float gm = 0.0f;
    for (int c = tid; c < n_chunks; c += HASH_SLICE_THREADS) gm = fmaxf(gm, a.wgmax[(size_t)level * n_chunks + c]);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) gm = fmaxf(gm, __shfl_xor(gm, d));
    if ((tid & 63) == 0) wmaxs[tid >> 6] = gm;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < HASH_SLICE_THREADS / 64; ++w) gm = fmaxf(gm, wmaxs[w]);
    int ge = 0;
    const bool bad = !(gm <= 3.0e38f);                                // a non-finite gradient: every entry of the level becomes NaN,
    if (gm > 0.0f && !bad) (void)frexpf(gm, &ge);                    // as a float sum would    (gm = f * 2^ge, f in [0.5, 1): gm < 2^ge)
    const double scale = bad ? 0.0 : ldexp(1.0, HASH_FIX_BITS - ge);
    const double inv = bad ? (double)__uint_as_float(0x7fc00000u) : ldexp(1.0, ge - HASH_FIX_BITS);
    if (blockIdx.x == 0 || (slice == 0 && part == 0)) { if (tid == 0) a.gscale[level] = inv; }
    const HashRecord* rec0 = (const HashRecord*)a.records + (size_t)level * a.n_chunks * (size_t)(HASH_CHUNK * HASH_REC_PER_ROW);
    for (int cg = 0; cg < (GROUPS ? n_chunks : 1); cg += HASH_MAX_CHUNKS) {
        if (GROUPS && cg > 0) {
            __syncthreads();                                         // the previous group's walk has read cstart / cbase
            fetch_offsets(cg);
        }
        const int n_here = !GROUPS ? n_chunks : (n_chunks - cg < HASH_MAX_CHUNKS ? n_chunks - cg : HASH_MAX_CHUNKS);
        // exclusive scan of the chunks' record counts, then a flat walk over this slice's records of ALL chunks of the group
        // (each chunk's are contiguous): one more round trip whatever the count
#pragma unroll
        for (int j = 0; j < HASH_MAX_CHUNKS / HASH_SLICE_THREADS; ++j) {
            cbase[tid + j * HASH_SLICE_THREADS] = r0[j];
            cstart[tid + j * HASH_SLICE_THREADS] = r1[j] - r0[j];
        }
        __syncthreads();
        if (tid < 64) {                                              // HASH_MAX_CHUNKS counts, 8 per lane of one wave
            constexpr int PER = HASH_MAX_CHUNKS / 64;
            unsigned v[PER], sum = 0;
#pragma unroll
            for (int q = 0; q < PER; ++q) { v[q] = cstart[tid * PER + q]; sum += v[q]; }
            unsigned inc = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const unsigned u = __shfl_up(inc, d); if (tid >= d) inc += u; }
            unsigned ex = inc - sum;
#pragma unroll
            for (int q = 0; q < PER; ++q) { cstart[tid * PER + q] = ex; ex += v[q]; }
            if (tid == 63) cstart[HASH_MAX_CHUNKS] = ex;
        }
        __syncthreads();
        const unsigned total = cstart[HASH_MAX_CHUNKS];
        for (unsigned i0 = 0; i0 < total; i0 += HASH_SLICE_THREADS * HASH_SLICE_UNROLL) {
            float4 rw[HASH_SLICE_UNROLL];
            bool in[HASH_SLICE_UNROLL];
#pragma unroll
            for (int q = 0; q < HASH_SLICE_UNROLL; ++q) {
                const unsigned i = i0 + q * HASH_SLICE_THREADS + tid;
                in[q] = i < total;
                rw[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (in[q]) {
                    int c = 0;                                       // chunk (within the group) of flat record i: last c with cstart[c] <= i
#pragma unroll
                    for (int st = HASH_MAX_CHUNKS / 2; st >= 1; st >>= 1)
                        if (c + st < n_here && cstart[c + st] <= i) c += st;
                    rw[q] = *(const float4*)(rec0 + (size_t)(cg + c) * (HASH_CHUNK * HASH_REC_PER_ROW) + cbase[c] + (i - cstart[c]));
                }
            }
#pragma unroll
            for (int q = 0; q < HASH_SLICE_UNROLL; ++q)
                if (in[q]) {
                    const unsigned rm = __float_as_uint(rw[q].x), r = rm & 0xffffffu, mask = rm >> 24;
                    float xv[3];
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        const float p = a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * rw[q].y;      // scene_rep.py:384 (as the bin kernel)
                        xv[d] = unit_coord(p, a.bb_lo[d], a.bb_hi[d], a.bb_is_f64 != 0);
                    }
                    HashCorners cn;
                    hash_corners(a, level, make_float4(xv[0], xv[1], xv[2], 0.0f), dense, cn);
                    const float gx = rw[q].z, gy = rw[q].w;
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if ((mask >> c) & 1u) {                      // the corners of this record: inside [lo, lo + n_ent) by construction
                            const uint32_t e = cn.idx[c] - lo;
                            atomicAdd(&acc[2 * e], hash_fix(cn.w[c] * gx, scale));
                            atomicAdd(&acc[2 * e + 1], hash_fix(cn.w[c] * gy, scale));
                        }
                }
        }
    }
    __syncthreads();
    if (n_part > 1) {                                                // a split level's partial sums -> the 64-bit scratch (exact)
        unsigned long long* gs = a.scratch64 + ((size_t)off + lo) * 2;
        for (uint32_t e = tid; e < n_ent * 2; e += HASH_SLICE_THREADS)
            if (acc[e] != 0ull) atomicAdd(gs + e, acc[e]);
        return;
    }
    // ---- Adam on the slice: entries are float2, moments have the table's layout
    const PlaneOpt o = a.opt;
    float2* P = (float2*)a.params + off + lo;
    float2* M = (float2*)o.m + off + lo;
    float2* V = (float2*)o.v + off + lo;
    for (uint32_t e = tid; e < n_ent; e += HASH_SLICE_THREADS) {
        float2 p = P[e], m = M[e], v = V[e];
        const float gx = (float)((double)(long long)acc[2 * e] * inv), gy = (float)((double)(long long)acc[2 * e + 1] * inv);
        adam_elem(p.x, gx, m.x, v.x, o);
        adam_elem(p.y, gy, m.y, v.y, o);
        P[e] = p; M[e] = m; V[e] = v;
    }
}

// Adam over the levels whose gradient went through the scratch (float2 entries [0, n_ent)); leaves the scratch zeroed
__global__ __launch_bounds__(256) void hash_finish_kernel(GridArgs a, unsigned n_ent) {
    const unsigned e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_ent) return;
    int level = 0;
    while (level + 1 < a.n_levels && e >= a.offset[level + 1]) ++level;
    const double inv = a.gscale[level];
    const PlaneOpt o = a.opt;
    unsigned long long* G = a.scratch64 + (size_t)e * 2;
    float2 p = ((float2*)a.params)[e], m = ((float2*)o.m)[e], v = ((float2*)o.v)[e];
    const float gx = (float)((double)(long long)G[0] * inv), gy = (float)((double)(long long)G[1] * inv);
    adam_elem(p.x, gx, m.x, v.x, o);
    adam_elem(p.y, gy, m.y, v.y, o);
    ((float2*)a.params)[e] = p; ((float2*)o.m)[e] = m; ((float2*)o.v)[e] = v;
    G[0] = 0ull; G[1] = 0ull;
}

int mne_hash_slice_count(const GridArgs& a) {
    int n = 0;
    for (int l = 0; l < a.n_levels; ++l) n += hash_slices_of(a, l) * hash_parts_of(a, l);
    return n;
}

// entries (float2) at the front of the table whose levels are updated through the scratch
unsigned mne_hash_scratch_entries(const GridArgs& a) {
    unsigned n = 0;
    for (int l = 0; l < a.n_levels; ++l)
        if (hash_parts_of(a, l) > 1) n = a.offset[l] + a.size[l];
    return n;
}

// workspace layout of mne_hash_slice_adam (bytes from the start; every block 256-byte aligned)
static size_t al256(size_t x) { return (x + 255) / 256 * 256; }
size_t mne_hash_layout(GridArgs& a, int R, int S, void* base) {
    const size_t rows = (size_t)R * S;
    const int n_chunks = hash_chunks_of((long long)rows);
    unsigned char* w = (unsigned char*)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { unsigned char* p = w ? w + off : nullptr; off += al256(bytes); return p; };
    a.scratch64 = (unsigned long long*)take((size_t)mne_hash_scratch_entries(a) * 2 * sizeof(unsigned long long));
    a.gscale = (double*)take((size_t)MNE_GRID_MAX_LEVELS * sizeof(double));
    a.n_chunks = n_chunks;
    a.wgmax = (float*)take((size_t)a.n_levels * n_chunks * sizeof(float));
    size_t so = 0;
    for (int l = 0; l < a.n_levels; ++l) { a.seg_level[l] = so; so += (size_t)n_chunks * (hash_slices_of(a, l) + 1); }
    a.seg_off = (unsigned*)take(so * sizeof(unsigned));
    a.records = (unsigned*)take((size_t)a.n_levels * n_chunks * HASH_CHUNK * HASH_REC_PER_ROW * sizeof(HashRecord));
    return off;
}

int mne_launch_hash_slice_adam(const GridArgs& a, hipStream_t st, void* event_after_bin) {
    if (a.R <= 0) return 0;
    for (int l = 0; l < a.n_levels; ++l)
        if (hash_slices_of(a, l) > HASH_MAX_SLICES) return -7;
    if (a.R >= (1 << 24)) return -7;                                                     // (a record carries its ray in 24 bits)
    MNE_LAUNCH(hash_bin_kernel, (unsigned)(a.n_chunks * a.n_levels), HASH_BIN_THREADS, 0, st, a);
    if (event_after_bin) (void)hipEventRecord((hipEvent_t)event_after_bin, st);
    if (a.n_chunks > HASH_MAX_CHUNKS) MNE_LAUNCH(hash_slice_adam_kernel<true>, (unsigned)mne_hash_slice_count(a), HASH_SLICE_THREADS, 0, st, a);
    else MNE_LAUNCH(hash_slice_adam_kernel<false>, (unsigned)mne_hash_slice_count(a), HASH_SLICE_THREADS, 0, st, a);
    const unsigned n_split = mne_hash_scratch_entries(a);
    if (n_split) MNE_LAUNCH(hash_finish_kernel, (n_split + 255) / 256, 256, 0, st, a, n_split);
    return 0;
}

int mne_launch_hash_rows(const GridArgs& a, int bwd, hipStream_t st) {
    if (bwd == 2) {
        const long long waves = (long long)a.R * ((a.S + 63) / 64);
        if (waves > 0) MNE_LAUNCH(hash_scatter_runs_kernel, (unsigned)((waves + 3) / 4) * a.n_levels, 256, 0, st, a);
        return 0;
    }
    const long long n = (long long)a.R * a.S * 16;
    if (n <= 0) return 0;
    if (bwd) MNE_LAUNCH(hash_scatter_kernel, (unsigned)((n + 255) / 256), 256, 0, st, a);
    else {
        long long grid = (long long)a.R * ((a.S + 63) / 64);
        if (a.ray_counts && a.ray_list && grid > 256) grid = 256;     // list pass: grid-stride over the deferred rays
        MNE_LAUNCH(hash_gather_kernel, (unsigned)grid, 256, 0, st, a);
    }
    return 0;
}

int mne_launch_grid(const GridArgs& a, int bwd, hipStream_t st) {
    const unsigned gx = (unsigned)((a.n + 255) / 256);
    if (bwd) hipLaunchOrEmu2D(grid_kernel<true>, gx, a.n_levels, 256, st, a);
    else hipLaunchOrEmu2D(grid_kernel<false>, gx, a.n_levels, 256, st, a);
    return 0;
}
