// tile_adam.hip -- plane-gradient scatter WITHOUT global atomics, fused with the dense Adam step.
//
// Why: the L2 executes roughly one 4-byte atomic per clock per channel, so scattering
// (contributing samples) x 24 corner rows x 32 channels of fp32 into the plane gradients costs
// ~0.3 ms per mapping iteration however the kernel is written (profiles/r01_ablation_*).  Here the
// backward kernel only APPENDS each contributing sample (its tape row id) to the lists of the
// 16x16-cell plane tiles its bilinear footprint touches; then one workgroup per tile accumulates
// that tile's gradient in LDS (ds_add_f32, conflict-free 128-B rows) and immediately applies Adam
// to the tile's parameters.  Tiles own their rows exclusively, so parameters are written with plain
// stores, no gradient buffer exists in HBM at all, and Adam streams 24 B/param (read p,m,v; write
// p,m,v) instead of 32.
//
// Reference semantics: the sum of d(feature)*bilinear-weight over all samples is exactly what
// grid_sampler_2d_backward accumulates (model/scene_rep.py:43-47 through autograd), followed by
// torch.optim.Adam over every plane element (mneslam_mp.py:459-469); only the fp32 summation order
// differs.
#include "mne_device.h"
#include "mne_launch.h"

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const PlaneOpt& o) {
    if (o.wd != 0.0f) g = g + o.wd * p;
    m = m + (g - m) * o.omb1;
    v = v * o.b2 + o.omb2 * (g * g);
    const float denom = sqrtf(v) / o.bc2_sqrt + o.eps;
    p = p - o.step_size * (m / denom);
}

// one (tape row, plane) contribution into the LDS tile; executed by a half-wave, lane = channel
__device__ __forceinline__ void tile_accumulate(const TileAdamArgs& a, const mne_plane_t& pl, int set, int ori, int lvl,
                                                int tx0, int ty0, unsigned t, int c, float* g) {
    const float* row = a.tape + (size_t)t * a.row_stride;
    const float px = row[a.t_pn + 0], py = row[a.t_pn + 1], pz = row[a.t_pn + 2];
    float gx, gy;
    orient_coords(ori, px, py, pz, gx, gy);
    Bilin b;
    bilin_setup(gx, gy, pl.h, pl.w, b);
    const float gc = row[a.t_dfeat + set * MNE_FEAT + lvl * MNE_C + c];
    const int lx = b.ix0 - tx0 * MNE_TILE, ly = b.iy0 - ty0 * MNE_TILE;     // NW corner relative to the tile
    const bool x0in = lx >= 0 && lx < MNE_TILE, x1in = lx + 1 >= 0 && lx + 1 < MNE_TILE;
    const bool y0in = ly >= 0 && ly < MNE_TILE, y1in = ly + 1 >= 0 && ly + 1 < MNE_TILE;
    if (x0in && y0in && b.w00 != 0.0f) atomicAdd(g + ((ly * MNE_TILE + lx) * MNE_C + c), gc * b.w00);
    if (x1in && y0in && b.w01 != 0.0f) atomicAdd(g + ((ly * MNE_TILE + lx + 1) * MNE_C + c), gc * b.w01);
    if (x0in && y1in && b.w10 != 0.0f) atomicAdd(g + (((ly + 1) * MNE_TILE + lx) * MNE_C + c), gc * b.w10);
    if (x1in && y1in && b.w11 != 0.0f) atomicAdd(g + (((ly + 1) * MNE_TILE + lx + 1) * MNE_C + c), gc * b.w11);
}

__global__ __launch_bounds__(256) void tile_adam_kernel(TileAdamArgs a) {
    __shared__ __attribute__((aligned(16))) float g[MNE_TILE * MNE_TILE * MNE_C];     // 32 KiB
    const int tile = blockIdx.x, tid = threadIdx.x;
    int pidx = 0;
    while (pidx + 1 < a.n_planes && tile >= a.bins.tile_base[pidx + 1]) ++pidx;
    const int set = pidx / 6, ori = (pidx % 6) / 2, lvl = pidx % 2;           // [set][orient][level]
    const mne_plane_t& pl = a.sc.plane[set][ori][lvl];
    const int local = tile - a.bins.tile_base[pidx];
    const int tx0 = local % a.bins.ntx[pidx], ty0 = local / a.bins.ntx[pidx];
    for (int i = tid; i < MNE_TILE * MNE_TILE * MNE_C / 4; i += 256) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int cnt = a.bins.counts[tile];
    const int n = cnt < a.bins.cap ? cnt : a.bins.cap;
    const int c = tid & 31, hw = tid >> 5;                                     // 8 half-waves
    const unsigned* lst = a.bins.lists + (size_t)tile * a.bins.cap;
    for (int e = hw; e < n; e += 8) tile_accumulate(a, pl, set, ori, lvl, tx0, ty0, lst[e], c, g);
    const int ns = *a.bins.spill_count;
    if (ns > 0) {                                                              // rare: entries beyond a list's capacity
        const int nsp = ns < a.bins.spill_cap ? ns : a.bins.spill_cap;
        for (int e = hw; e < nsp; e += 8)
            if (a.bins.spill[2 * e] == (unsigned)tile) tile_accumulate(a, pl, set, ori, lvl, tx0, ty0, a.bins.spill[2 * e + 1], c, g);
    }
    __syncthreads();
    // ---- Adam on the tile: 16 rows x (16 cells x 32 ch) = 2048 float4, 8 per thread
    const PlaneOpt& o = a.opt[pidx];
    float* P = (float*)pl.data;
#pragma unroll
    for (int it = 0; it < (MNE_TILE * MNE_TILE * MNE_C / 4) / 256; ++it) {
        const int i4 = it * 256 + tid;
        const int y = i4 / (MNE_TILE * MNE_C / 4), x4 = i4 % (MNE_TILE * MNE_C / 4);
        const int cell = x4 / (MNE_C / 4), ch4 = x4 % (MNE_C / 4);
        const int gy = ty0 * MNE_TILE + y, gx = tx0 * MNE_TILE + cell;
        if (gy < pl.h && gx < pl.w) {
            const size_t off = ((size_t)gy * pl.w + gx) * MNE_C + ch4 * 4;
            float4 p = *(float4*)(P + off), m = *(float4*)(o.m + off), v = *(float4*)(o.v + off);
            const float4 gg = *(const float4*)(g + ((y * MNE_TILE + cell) * MNE_C + ch4 * 4));
            adam_elem(p.x, gg.x, m.x, v.x, o); adam_elem(p.y, gg.y, m.y, v.y, o);
            adam_elem(p.z, gg.z, m.z, v.z, o); adam_elem(p.w, gg.w, m.w, v.w, o);
            *(float4*)(P + off) = p; *(float4*)(o.m + off) = m; *(float4*)(o.v + off) = v;
        }
    }
    if (tid == 0) a.bins.counts[tile] = 0;                                     // ready for the next iteration
}

void mne_tile_geometry(const mne_scene_t& sc, TileBins& b) {
    int base = 0, k = 0;
    for (int s = 0; s < sc.n_sets; ++s)
        for (int o = 0; o < 3; ++o)
            for (int l = 0; l < 2; ++l, ++k) {
                const mne_plane_t& p = sc.plane[s][o][l];
                const int ntx = (p.w + MNE_TILE - 1) / MNE_TILE, nty = (p.h + MNE_TILE - 1) / MNE_TILE;
                b.tile_base[k] = base;
                b.ntx[k] = ntx;
                base += ntx * nty;
            }
    for (; k <= MNE_MAX_PLANES; ++k) b.tile_base[k] = base;
    b.tile_base[sc.n_sets * 6] = base;
}

int mne_launch_tile_adam(const TileAdamArgs& a, hipStream_t st) {
    const int n_tiles = a.bins.tile_base[a.n_planes];
    if (n_tiles <= 0) return 0;
    MNE_LAUNCH(tile_adam_kernel, n_tiles, 256, 0, st, a);
    return 0;
}
