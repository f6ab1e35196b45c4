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

// One list entry = MNE_ENTRY_WORDS uint32 written by the backward kernel: the tape row, the footprint's
// NW corner relative to the tile (+1, so 0 means "one cell before the tile") and the four bilinear
// weights (0 for corners outside the plane).  The tile kernel therefore does no coordinate math.
//
// LDS fp32 atomics (ds_add_f32) measured ~0.25 op/clk/CU here (profiles/r01_ablation_tile_adam.txt), so
// the accumulation avoids them: per pass of PASS_ENTRIES list entries the workgroup first copies the
// entries into LDS and files each into the queues of the (at most two) tile rows it touches (integer
// LDS counters only); then every half-wave owns one tile row (or half of one), lane = channel, and
// applies its row queue with plain read-add-write: single owner, DS operations of a wave complete in
// order.  The gradient-row loads of QB queue items are issued together.
#ifndef TILE_THREADS
#define TILE_THREADS 512
#endif
#ifndef PASS_ENTRIES
#define PASS_ENTRIES 256
#endif
#ifndef QB
#define QB 8
#endif
#define TILE_HW (TILE_THREADS / 32)           // half-waves: 16 rows x TILE_SIDES
#define TILE_SIDES (TILE_HW / MNE_TILE)
#define SIDE_CELLS (MNE_TILE / TILE_SIDES)

// Processing order: tiles bucketed by floor(log2(list length + 1)), heaviest bucket first, so the few
// very long lists (every ray of a keyframe passes through the tile holding its camera centre) start
// at once instead of forming the tail of the launch.
__global__ __launch_bounds__(1024) void tile_order_kernel(TileAdamArgs a, int n_tiles) {
    __shared__ int hist[32], start[32];
    const int tid = threadIdx.x;
    if (tid < 32) hist[tid] = 0;
    __syncthreads();
    for (int t = tid; t < n_tiles; t += 1024) atomicAdd(&hist[31 - __clz(a.bins.counts[t] + 1)], 1);
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int b = 31; b >= 0; --b) { start[b] = acc; acc += hist[b]; }
    }
    __syncthreads();
    for (int t = tid; t < n_tiles; t += 1024) a.bins.order[atomicAdd(&start[31 - __clz(a.bins.counts[t] + 1)], 1)] = t;
}

__global__ __launch_bounds__(TILE_THREADS) void tile_adam_kernel(TileAdamArgs a) {
    __shared__ __attribute__((aligned(16))) float g[MNE_TILE * MNE_TILE * MNE_C];     // 32 KiB gradient tile
    __shared__ unsigned short rowq[MNE_TILE][PASS_ENTRIES];                           // row queues of this pass
    __shared__ unsigned ents[PASS_ENTRIES][MNE_ENTRY_WORDS];                          // entries of this pass
    __shared__ int rowq_n[MNE_TILE];
    const int tid = threadIdx.x;
    const int tile = a.bins.order[blockIdx.x];
    int pidx = 0;
    while (pidx + 1 < a.n_planes && tile >= a.bins.tile_base[pidx + 1]) ++pidx;
    const int set = pidx / 6, lvl = pidx % 2;                                 // [set][orient][level]
    const mne_plane_t& pl = a.sc.plane[set][(pidx % 6) / 2][lvl];
    const int local = tile - a.bins.tile_base[pidx];
    const int tx0 = local % a.bins.ntx[pidx], ty0 = local / a.bins.ntx[pidx];
    for (int i = tid; i < MNE_TILE * MNE_TILE * MNE_C / 4; i += TILE_THREADS) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int cnt = a.bins.counts[tile];
    const int n_list = (a.dbg & 32) ? 0 : (cnt < a.bins.cap ? cnt : a.bins.cap);
    int n_spill = 0;
    if (cnt > a.bins.cap) {                     // only a tile whose list overflowed has entries in the spill area
        const int ns = *a.bins.spill_count;
        n_spill = ns < a.bins.spill_cap ? ns : a.bins.spill_cap;
    }
    const unsigned* lst = a.bins.lists + (size_t)tile * a.bins.cap * MNE_ENTRY_WORDS;
    const int c = tid & 31, hw = tid >> 5;
    const int my_row = hw / TILE_SIDES, my_lo = (hw % TILE_SIDES) * SIDE_CELLS;   // cells [my_lo, my_lo+SIDE_CELLS) of my_row
    // passes over the list, then over the spill area (spill entries of other tiles are skipped when filing)
    const int n_total = n_list + n_spill;
    for (int p0 = 0; p0 < n_total; p0 += PASS_ENTRIES) {
        if (tid < MNE_TILE) rowq_n[tid] = 0;
        __syncthreads();
        const int p1 = p0 + PASS_ENTRIES < n_total ? p0 + PASS_ENTRIES : n_total;
        for (int e = p0 + tid; e < p1; e += TILE_THREADS) {
            const unsigned* ent;
            if (e < n_list) ent = lst + (size_t)e * MNE_ENTRY_WORDS;
            else {
                const unsigned* sp = a.bins.spill + (size_t)(e - n_list) * MNE_SPILL_WORDS;
                if (sp[0] != (unsigned)tile) continue;
                ent = sp + 1;
            }
            const unsigned short item = (unsigned short)(e - p0);
#pragma unroll
            for (int w = 0; w < MNE_ENTRY_WORDS; ++w) ents[item][w] = ent[w];
            const int ly = (int)((ent[1] >> 8) & 0xff) - 1;
            if (ly >= 0) rowq[ly][atomicAdd(&rowq_n[ly], 1)] = item;
            if (ly + 1 < MNE_TILE) rowq[ly + 1][atomicAdd(&rowq_n[ly + 1], 1)] = item;
        }
        __syncthreads();
        const int nq = rowq_n[my_row];
        float* grow = g + my_row * MNE_TILE * MNE_C + c;
        for (int q0 = 0; q0 < nq; q0 += QB) {
            unsigned hdr[QB];
            float w0[QB], w1[QB], gc[QB];
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                const unsigned* ent = ents[rowq[my_row][q0 + j < nq ? q0 + j : q0]];
                hdr[j] = ent[1];
                const bool bottom = ((int)((hdr[j] >> 8) & 0xff) - 1) != my_row;     // this row is the footprint's lower row
                w0[j] = __uint_as_float(ent[bottom ? 4 : 2]);
                w1[j] = __uint_as_float(ent[bottom ? 5 : 3]);
                gc[j] = (a.dbg & 256) ? 1.0f : a.tape[(size_t)ent[0] * a.row_stride + a.t_dfeat + set * MNE_FEAT + lvl * MNE_C + c];
            }
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                if (q0 + j < nq) {
                    const int lx = (int)(hdr[j] & 0xff) - 1;
                    if (lx >= my_lo && lx < my_lo + SIDE_CELLS && w0[j] != 0.0f) grow[lx * MNE_C] += gc[j] * w0[j];
                    if (lx + 1 >= my_lo && lx + 1 < my_lo + SIDE_CELLS && w1[j] != 0.0f) grow[(lx + 1) * MNE_C] += gc[j] * w1[j];
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    // ---- Adam on the tile: 16 rows x (16 cells x 32 ch) = 2048 float4
    const PlaneOpt& o = a.opt[pidx];
    float* P = (float*)pl.data;
#pragma unroll
    for (int it = 0; it < (MNE_TILE * MNE_TILE * MNE_C / 4) / TILE_THREADS; ++it) {
        if (a.dbg & 64) break;
        const int i4 = it * TILE_THREADS + tid;
        const int y = i4 / (MNE_TILE * MNE_C / 4), x4 = i4 % (MNE_TILE * MNE_C / 4);
        const int cell = x4 / (MNE_C / 4), ch4 = x4 % (MNE_C / 4);
        const int gy = ty0 * MNE_TILE + y, gx = tx0 * MNE_TILE + cell;
        if (gy < pl.h && gx < pl.w) {
            const size_t off = ((size_t)gy * pl.w + gx) * MNE_C + ch4 * 4;
            float4 p = *(float4*)(P + off), m = *(float4*)(o.m + off), v = *(float4*)(o.v + off);
            const float4 gg = ((const float4*)g)[i4];
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
    MNE_LAUNCH(tile_order_kernel, 1, 1024, 0, st, a, n_tiles);
    MNE_LAUNCH(tile_adam_kernel, n_tiles, TILE_THREADS, 0, st, a);
    return 0;
}
