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
    long long row = gid >> 4;
    if (row >= (long long)a.R * a.S || level >= a.n_levels) return;
    int r = (int)(row / a.S);
    const int s = (int)(row % a.S);
    if (BWD && a.ray_tiles && s >= a.ray_tiles[r] * 32) return;
    if (!BWD && a.ray_counts) {
        // forward under early termination: the first pass fills the tiles decode_kernel can reach (a-priori prefix + the
        // resolver's extension), the list pass the rest of the rays that were deferred
        if (a.ray_list) {
            if (r >= *a.ray_list_count) return;
            r = a.ray_list[r];
            row = (long long)r * a.S + s;
        }
        const int ntile = (a.S + 31) / 32;
        const int need = a.ray_counts[(size_t)r * MNE_N_COUNT + MNE_C_NEED];
        int t = (need + 31) / 32;
        t = (t < 1 ? 1 : (t > ntile ? ntile : t)) + MNE_RESOLVER_MAX_EXT;         // = prefix_tiles() of render.hip + extension
        if (a.ray_list ? s < t * 32 : s >= t * 32) return;
    }
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

// ---- table update without atomics: slices of the table accumulated in LDS, Adam fused ------------------------------------
// Global float atomics retire at ~20 G/s chip-wide (previous section), and the samples of one iteration touch millions of
// distinct table entries, so no reduction in front of the atomics gets the scatter under ~0.5 ms.  The table itself,
// though, is barely larger than the chip's LDS (5.25 M entries x 8 B = 42 MB vs 256 x 160 KB), and what has to be
// examined per sample is tiny.  So the roles are turned around, as in tile_adam.hip: one workgroup OWNS a slice of
// HASH_SLICE entries of one level, keeps its gradient in LDS, walks over ALL backward rows of the iteration (their grid
// input x and this level's d(feature), packed level-major by hash_pack_kernel so that the walk is a coalesced stream
// out of L2), adds the corners that fall into its slice with ds_add_f32, and applies Adam to the slice: no gradient
// buffer in HBM, no global atomics, the table and its moments are read and written once with plain coalesced accesses.
// Walking cost: ~150 instructions per (row, workgroup) -- the kernel is VALU-bound, see HASH_SLICE below.
// slice = 16384 entries = 128 KiB of LDS, one workgroup of 1024 threads per CU: the walk is VALU-bound (~150
// instructions per row and workgroup at one wave instruction per 4 cycles and SIMD), so what counts is the number of
// (row, workgroup) visits = rows x table bytes / slice bytes.  Measured: 8192 entries x 512 threads (two per CU) 499 us,
// 16384 x 1024 383 us; unroll 1/2/4/8: 654/577/556/548 us; without the LDS atomics 481 of 556 us
// (profiles/r02_hash_slices_variants.txt).
#ifndef HASH_SLICE
#define HASH_SLICE 16384
#endif
#ifndef HASH_SLICE_THREADS
#define HASH_SLICE_THREADS 1024
#endif
#ifndef HASH_UNROLL
#define HASH_UNROLL 4
#endif

// Workgroups of one level: slices x replicas.  A dense (coarse) level has few slices and EVERY sample hits them: its rows
// are split over `replicas` workgroups (LDS float atomics retire ~0.7 G lane-ops/s per CU: one workgroup taking all
// 2 M corner updates of level 0 ran 2.9 ms), which then add their non-zero sums into a small gradient scratch with global
// atomics; hash_dense_adam_kernel finishes those levels.  Hashed levels: one workgroup per slice, Adam fused.
#ifndef HASH_LEVEL_WGS
#define HASH_LEVEL_WGS 64
#endif
#define HASH_LDS_ADD(p, v) atomicAdd((p), (v))
__host__ __device__ __forceinline__ int hash_slices_of(const GridArgs& a, int level) { return (int)((a.size[level] + HASH_SLICE - 1) / HASH_SLICE); }
__host__ __device__ __forceinline__ bool hash_level_dense(const GridArgs& a, int level) {
    return (unsigned long long)a.res[level] * a.res[level] * a.res[level] <= a.size[level];
}
__host__ __device__ __forceinline__ int hash_replicas_of(const GridArgs& a, int level) {
    if (!hash_level_dense(a, level)) return 1;
    const int r = HASH_LEVEL_WGS / hash_slices_of(a, level);
    return r < 1 ? 1 : r;
}

// first packed row of every ray: exclusive scan of min(ray_tiles[r] * 32, S), one workgroup
__global__ __launch_bounds__(1024) void hash_offsets_kernel(GridArgs a) {
    __shared__ int part[1024];
    __shared__ int carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < a.R; base += 1024) {
        const int r = base + tid;
        int n = 0;
        if (r < a.R) { n = a.ray_tiles[r] * 32; n = n < a.S ? n : a.S; }
        part[tid] = n;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int v = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (r < a.R) a.offs[r] = carry + part[tid] - n;
        __syncthreads();
        if (tid == 1023) carry += part[1023];
        __syncthreads();
    }
    if (tid == 0) a.offs[a.R] = carry;
}

// 16 consecutive backward rows of one ray per workgroup: x of each row, and the rows' d(feature) transposed to level-major
// The pack kernel leaves one 32-bit word per (row, hashed level) with the bits of the slices its eight corners fall into
// (0 for rows without gradient; inside the bounding box cell_x < 2^14, so the slice = index >> 14 depends on the (y, z)
// corner pair only: at most four bits); a slice workgroup then streams 4 bytes per row and looks at the row itself only
// when its bit is set (1 row in 8 on T = 2^19 levels).
__device__ __forceinline__ bool hash_level_masked(const GridArgs& a, int level) {
    const uint32_t size = a.size[level];
    return a.masks && !hash_level_dense(a, level) && (size & (size - 1u)) == 0u && size <= 32u * HASH_SLICE &&
           (HASH_SLICE & (HASH_SLICE - 1)) == 0;
}

__global__ __launch_bounds__(256) void hash_pack_kernel(GridArgs a) {
    __shared__ float2 tr[16][17];
    __shared__ unsigned trm[16][17];
    __shared__ float xs_l[16][4];
    const int groups = (a.S + 15) / 16;
    const int r = blockIdx.x / groups, s0 = (blockIdx.x % groups) * 16;
    int n_rows = a.ray_tiles[r] * 32;
    n_rows = n_rows < a.S ? n_rows : a.S;
    if (s0 >= n_rows) return;
    const int tid = threadIdx.x, lv = tid & 15, rr = tid >> 4;
    const bool in = s0 + rr < n_rows;
    const long long row = (long long)r * a.S + s0 + rr;
    float2 g = make_float2(0.f, 0.f);
    if (in && lv < a.n_levels) g = *(const float2*)(a.tape + (size_t)row * a.row_stride + a.col_d + lv * 2);
    tr[rr][lv] = g;
    const long long k0 = a.offs[r] + s0;
    if (lv == 0 && in) {
        const float z = a.z_vals[row];
        float x[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float p = a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z;              // scene_rep.py:384
            x[d] = unit_coord(p, a.bb_lo[d], a.bb_hi[d], a.bb_is_f64 != 0);
        }
        a.xs[k0 + rr] = make_float4(x[0], x[1], x[2], 0.0f);
        xs_l[rr][0] = x[0]; xs_l[rr][1] = x[1]; xs_l[rr][2] = x[2];
    }
    __syncthreads();
    unsigned m = 0u;
    if (in && lv < a.n_levels && (g.x != 0.0f || g.y != 0.0f)) {
        m = 0xffffffffu;
        if (hash_level_masked(a, lv)) {
            const float scale = a.scale[lv];
            const uint32_t msk = a.size[lv] - 1u;
            const uint32_t cx = (uint32_t)(int)floorf(fmaf(scale, xs_l[rr][0], 0.5f));
            const uint32_t cy = (uint32_t)(int)floorf(fmaf(scale, xs_l[rr][1], 0.5f)), cz = (uint32_t)(int)floorf(fmaf(scale, xs_l[rr][2], 0.5f));
            const uint32_t hx[2] = {cx, cx + 1u};                  // (samples outside the bounding box have wrapped cells: all 8 corners count)
            const uint32_t hy[2] = {cy * 2654435761u, (cy + 1u) * 2654435761u}, hz[2] = {cz * 805459861u, (cz + 1u) * 805459861u};
            m = 0u;
#pragma unroll
            for (int q = 0; q < 8; ++q) m |= 1u << (((hx[q & 1] ^ hy[(q >> 1) & 1] ^ hz[q >> 2]) & msk) / HASH_SLICE);
        }
    }
    trm[rr][lv] = m;
    __syncthreads();
    const int rr2 = tid & 15, lv2 = tid >> 4;                        // 16 adjacent lanes = 16 consecutive packed rows of one level
    if (s0 + rr2 < n_rows && lv2 < a.n_levels) {
        a.dfeat_lv[(size_t)lv2 * a.pack_cap + k0 + rr2] = tr[rr2][lv2];
        if (a.masks) a.masks[(size_t)lv2 * a.pack_cap + k0 + rr2] = trm[rr2][lv2];
    }
}

__global__ __launch_bounds__(HASH_SLICE_THREADS, HASH_SLICE_THREADS / 128) void hash_slice_adam_kernel(GridArgs a) {
    MNE_DYN_LDS(lds_raw);
    float* acc = (float*)lds_raw;                                    // [HASH_SLICE][2] gradient of this slice
    const int tid = threadIdx.x;
    int level = 0, k = blockIdx.x;
    while (level + 1 < a.n_levels && k >= hash_slices_of(a, level) * hash_replicas_of(a, level)) {
        k -= hash_slices_of(a, level) * hash_replicas_of(a, level);
        ++level;
    }
    const int n_rep = hash_replicas_of(a, level), rep = k % n_rep;
    k /= n_rep;
    const uint32_t res = a.res[level], size = a.size[level], off = a.offset[level];
    const uint32_t lo = (uint32_t)k * HASH_SLICE;
    const uint32_t n_ent = size - lo < HASH_SLICE ? size - lo : HASH_SLICE;
    const bool dense = (unsigned long long)res * res * res <= size;
    const bool pow2 = (size & (size - 1u)) == 0u;
    const float scale = a.scale[level];
    for (int i = tid; i < HASH_SLICE * 2 / 4; i += HASH_SLICE_THREADS) ((float4*)acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int n_live = a.offs[a.R];
    const float2* gl = a.dfeat_lv + (size_t)level * a.pack_cap;
    // The walk over this replica's share of the rows, HASH_UNROLL rows per thread in flight (the stream comes out of L2 /
    // Infinity Cache: latency, not bytes).  (Tried for the dense levels and dropped: a segmented wave scan in front of
    // the LDS atomics -- 96 dependent ds_bpermute per 64 rows, 1.7 ms for level 0's workgroup; per-thread contiguous row
    // chunks -- 2.9 ms; profiles/r02_hash_slices_*.txt.)
    const int row_lo = (int)((long long)n_live * rep / n_rep), row_hi = (int)((long long)n_live * (rep + 1) / n_rep);
    int n_iter = (row_hi - row_lo + HASH_SLICE_THREADS - 1) / HASH_SLICE_THREADS;
    if (hash_level_masked(a, level)) {
        // ---- masked walk: 4 bytes per row; the rows whose mask has this slice's bit are collected per wave (ballot +
        // prefix popcount into a 128-entry LDS ring) and processed 64 at a time with every lane busy
        const int lane = tid & 63, wv = tid >> 6;
        unsigned* ring = (unsigned*)(acc + (size_t)HASH_SLICE * 2) + wv * 128;
        const unsigned* mk = a.masks + (size_t)level * a.pack_cap;
        const uint32_t msk = size - 1u;
        int head = 0, fill = 0;                                       // wave-uniform
        auto process = [&](int count) {                               // the first `count` ring entries, one per lane
            if (lane < count) {
                const int i = (int)ring[(head + lane) & 127];
                const float2 g = gl[i];
                const float4 x = a.xs[i];
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
                const uint32_t hx[2] = {cell[0] & msk, (cell[0] + 1u) & msk};
                const uint32_t hy[2] = {(cell[1] * 2654435761u) & msk, ((cell[1] + 1u) * 2654435761u) & msk};
                const uint32_t hz[2] = {(cell[2] * 805459861u) & msk, ((cell[2] + 1u) * 805459861u) & msk};
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t e = (hx[c & 1] ^ hy[(c >> 1) & 1] ^ hz[(c >> 2) & 1]) - lo;
                    if (e < n_ent) {
                        const float w = (wx[c & 1] * wy[(c >> 1) & 1]) * wz[(c >> 2) & 1];     // product order of grid_kernel
                        HASH_LDS_ADD(acc + 2 * e, w * g.x);
                        HASH_LDS_ADD(acc + 2 * e + 1, w * g.y);
                    }
                }
            }
        };
        for (int it = 0; it < n_iter; it += HASH_UNROLL) {
            unsigned mq[HASH_UNROLL];
#pragma unroll
            for (int q = 0; q < HASH_UNROLL; ++q) {
                const int i = row_lo + (it + q) * HASH_SLICE_THREADS + tid;
                mq[q] = i < row_hi ? mk[i] : 0u;
            }
#pragma unroll
            for (int q = 0; q < HASH_UNROLL; ++q) {
                const bool hit = (mq[q] >> k) & 1u;
                const unsigned long long b = __ballot(hit);
                if (b == 0ull) continue;
                if (hit) ring[(head + fill + __popcll(b & ((1ull << lane) - 1ull))) & 127] = (unsigned)(row_lo + (it + q) * HASH_SLICE_THREADS + tid);
                fill += __popcll(b);
                MNE_WAVE_SYNC();
                if (fill >= 64) {
                    process(64);
                    head = (head + 64) & 127; fill -= 64;
                    MNE_WAVE_SYNC();
                }
            }
        }
        if (fill > 0) process(fill);
        n_iter = 0;                                                   // the generic walk below is skipped
    }
    for (int it = 0; it < n_iter; it += HASH_UNROLL) {
        float2 gq[HASH_UNROLL];
        float4 xq[HASH_UNROLL];
        bool inq[HASH_UNROLL];
#pragma unroll
        for (int q = 0; q < HASH_UNROLL; ++q) {
            const int i = row_lo + (it + q) * HASH_SLICE_THREADS + tid;
            inq[q] = i < row_hi;
            gq[q] = make_float2(0.f, 0.f); xq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (inq[q]) { gq[q] = gl[i]; xq[q] = a.xs[i]; }
        }
#pragma unroll
        for (int q = 0; q < HASH_UNROLL; ++q) {
            const float2 g = gq[q];
            const float4 x = xq[q];
            if (!(inq[q] && (g.x != 0.0f || g.y != 0.0f))) continue;
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
            if (dense) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t e = grid_index(cell[0] + (c & 1), cell[1] + ((c >> 1) & 1), cell[2] + ((c >> 2) & 1), res, size, true) - lo;
                    if (e < n_ent) {
                        const float w = (wx[c & 1] * wy[(c >> 1) & 1]) * wz[(c >> 2) & 1];     // product order of grid_kernel
                        HASH_LDS_ADD(acc + 2 * e, w * g.x);
                        HASH_LDS_ADD(acc + 2 * e + 1, w * g.y);
                    }
                }
            } else {
                // (a ^ b ^ c) & m == (a & m) ^ (b & m) ^ (c & m): mask the six components once when the level size is 2^n
                const uint32_t msk = pow2 ? size - 1u : 0xffffffffu;
                const uint32_t hx[2] = {cell[0] & msk, (cell[0] + 1u) & msk};
                const uint32_t hy[2] = {(cell[1] * 2654435761u) & msk, ((cell[1] + 1u) * 2654435761u) & msk};
                const uint32_t hz[2] = {(cell[2] * 805459861u) & msk, ((cell[2] + 1u) * 805459861u) & msk};
                const uint32_t hxy[4] = {hx[0] ^ hy[0], hx[1] ^ hy[0], hx[0] ^ hy[1], hx[1] ^ hy[1]};
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t h = hxy[c & 3] ^ hz[(c >> 2) & 1];
                    const uint32_t e = (pow2 ? h : (h % size)) - lo;
                    if (e < n_ent) {
                        const float w = (wx[c & 1] * wy[(c >> 1) & 1]) * wz[(c >> 2) & 1];
                        HASH_LDS_ADD(acc + 2 * e, w * g.x);
                        HASH_LDS_ADD(acc + 2 * e + 1, w * g.y);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (n_rep > 1) {                                                 // partial sums of a dense level: into the gradient scratch
        float* gs = a.dparams + ((size_t)off + lo) * 2;
        for (uint32_t e = tid; e < n_ent * 2; e += HASH_SLICE_THREADS)
            if (acc[e] != 0.0f) unsafeAtomicAdd(gs + e, acc[e]);
        return;
    }
    // ---- Adam on the slice: entries are float2, moments have the table's layout
    const PlaneOpt o = a.opt;
    float2* P = (float2*)a.params + off + lo;
    float2* M = (float2*)o.m + off + lo;
    float2* V = (float2*)o.v + off + lo;
    for (uint32_t e = tid; e < n_ent; e += HASH_SLICE_THREADS) {
        float2 p = P[e], m = M[e], v = V[e];
        const float2 g = *(const float2*)(acc + 2 * e);
        adam_elem(p.x, g.x, m.x, v.x, o);
        adam_elem(p.y, g.y, m.y, v.y, o);
        P[e] = p; M[e] = m; V[e] = v;
    }
}

// Adam over the levels whose gradient went through the scratch (float2 entries [0, n_ent)); leaves the scratch zeroed
__global__ __launch_bounds__(256) void hash_dense_adam_kernel(GridArgs a, unsigned n_ent) {
    const unsigned e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_ent) return;
    const PlaneOpt o = a.opt;
    float2* G = (float2*)a.dparams;
    float2 p = ((float2*)a.params)[e], m = ((float2*)o.m)[e], v = ((float2*)o.v)[e];
    const float2 g = G[e];
    adam_elem(p.x, g.x, m.x, v.x, o);
    adam_elem(p.y, g.y, m.y, v.y, o);
    ((float2*)a.params)[e] = p; ((float2*)o.m)[e] = m; ((float2*)o.v)[e] = v;
    G[e] = make_float2(0.f, 0.f);
}

int mne_hash_slice_count(const GridArgs& a) {
    int n = 0;
    for (int l = 0; l < a.n_levels; ++l) n += hash_slices_of(a, l) * hash_replicas_of(a, l);
    return n;
}

// entries (float2) at the front of the table whose levels are updated through the gradient scratch
unsigned mne_hash_scratch_entries(const GridArgs& a) {
    unsigned n = 0;
    for (int l = 0; l < a.n_levels; ++l)
        if (hash_replicas_of(a, l) > 1) n = a.offset[l] + a.size[l];
    return n;
}

int mne_launch_hash_slice_adam(const GridArgs& a, hipStream_t st) {
    if (a.R <= 0) return 0;
    MNE_LAUNCH(hash_offsets_kernel, 1, 1024, 0, st, a);
    MNE_LAUNCH(hash_pack_kernel, (unsigned)(a.R * ((a.S + 15) / 16)), 256, 0, st, a);
    const size_t lds = (size_t)HASH_SLICE * 2 * sizeof(float) + (size_t)(HASH_SLICE_THREADS / 64) * 128 * sizeof(unsigned);   // + per-wave rings
    MNE_SET_MAX_LDS(hash_slice_adam_kernel, MNE_LDS_MAX);
    MNE_LAUNCH(hash_slice_adam_kernel, (unsigned)mne_hash_slice_count(a), HASH_SLICE_THREADS, lds, st, a);
    const unsigned n_dense = mne_hash_scratch_entries(a);
    if (n_dense) MNE_LAUNCH(hash_dense_adam_kernel, (n_dense + 255) / 256, 256, 0, st, a, n_dense);
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
