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


// One list entry = 32 bytes written by ray_kernel with two 16-byte stores: the tape row, the footprint's
// NW corner relative to the tile (+1, so 0 means "one cell before the tile"), the four bilinear
// weights (0 for corners outside the plane) and the tile id (what identifies an entry in the shared spill area).
// The tile kernel therefore does no coordinate math.
//
// Accumulation (per pass of PASS_ENTRIES list entries).  Heavy tiles are the critical path of the launch,
// and their entries are concentrated: a wall or the floor projects onto a LINE of cells of the planes
// it is perpendicular to, so a whole pass may hit one or two cell rows.  Work is therefore split by
// CONTRIBUTION, not by position:
//   A  every entry is copied to LDS and its (up to) four corner contributions are ranked per cell with
//      integer LDS counters (counting sort by cell, 256 keys);
//   B  one wave turns the counts into start offsets;  C  the contributions are written in cell order;
//   (meanwhile every entry's gradient row -- 32 floats of this plane level -- is fetched from the tape
//   into LDS exactly once, all loads of the pass in flight together: corner-level work would otherwise
//   read each row four times, and these reads compete with the Adam stream for HBM/Infinity-Cache)
//   D  the sorted array is cut into TILE_GROUPS equal ranges, one per 8-lane group (float4 per lane = the
//      32 channels of a row; the walk is VALU-issue-bound, so 8 contributions per wave instruction
//      instead of 2): each walks its range reading rows from LDS, sums runs of equal cells in registers and
//      adds a finished run to the LDS tile with a plain read-add-write: a run belongs to the group in
//      whose range it starts, so every cell has a single writer per pass and no LDS atomics are needed.
// Every group thus handles the same number of contributions whatever their spatial distribution.
#ifndef TILE_THREADS
#define TILE_THREADS 512
#endif
#ifndef PASS_ENTRIES
#define PASS_ENTRIES 256
#endif
#ifndef DB
#define DB 8
#endif
#ifndef TILE_MIN_WAVES
#define TILE_MIN_WAVES 4        // waves per SIMD the register allocation must allow: 2 workgroups of 8 waves per CU (LDS-limited)
#endif
// Cache policy of the Adam sweep's streams (bit mask: 1 = m / v stores, 2 = m / v loads, 4 = p loads, 8 = p stores nontemporal).  The moments are
// touched once per iteration and by nothing else; the parameters are read again by the next iteration's gather.  Measured
// (profiles/r05_adam_nt.txt; the kernel itself hardly changes, the kernels around it find more of their data in cache): moments
// nontemporal (3) office0 2233 -> 2267 it/s, ScanNet 1209 -> 1218, INS Indoor within noise; + parameter stores (11) the same;
// + parameter loads (15) loses it again.
#ifndef MNE_ADAM_NT
#define MNE_ADAM_NT 3
#endif
typedef float mne_f4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 nt_load4(const float* p) {
#ifndef MNE_HOST_EMU
    if constexpr (NT) {
        const mne_f4 v = __builtin_nontemporal_load((const mne_f4*)p);
        return make_float4(v.x, v.y, v.z, v.w);
    }
#endif
    return *(const float4*)p;
}
template <bool NT>
__device__ __forceinline__ void nt_store4(float* p, float4 v) {
#ifndef MNE_HOST_EMU
    if constexpr (NT) {
        mne_f4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        __builtin_nontemporal_store(t, (mne_f4*)p);
        return;
    }
#endif
    *(float4*)p = v;
}
#ifndef TILE_EMPTY_FAST
#define TILE_EMPTY_FAST 1
#endif
#define TILE_GROUPS (TILE_THREADS / 8)        // 8-lane groups (float4 per lane = one 32-channel row)
#define TILE_CELLS (MNE_TILE * MNE_TILE)
static_assert(PASS_ENTRIES <= TILE_THREADS && PASS_ENTRIES <= 256 && PASS_ENTRIES % TILE_GROUPS == 0, "one staged entry per thread; item index must fit 8 bits");
static_assert(TILE_CELLS == 256, "the prefix step assumes 4 cells per lane of one wave");

// Processing order: tiles bucketed by floor(log2(list length + 1)), heaviest bucket first, so the few
// very long lists (every ray of a keyframe passes through the tile holding its camera centre) start
// at once instead of forming the tail of the launch.
// Work items of tile_adam_kernel.  item = tile | part << 20 | parts << 26 (parts = 0: the whole list).  A list longer than
// `split` entries is cut into parts of equal length that different workgroups accumulate (each in its own LDS tile);
// their partial gradient tiles meet in the split scratch and the part that arrives last applies Adam.  Without this the
// longest lists ARE the launch: ScanNet with colour planes (12 planes, 2 M entries) ran 0.82 ms, INS Indoor (1045
// samples per ray into 750 tiles, 15 k entries per list on average) 4.2 ms, against 0.2 / 1.0 ms of balanced work.
// split = max(split_min (4096 by default), 2 * total entries / MNE_TILE_SPLIT_PARTS): at most MNE_TILE_SPLIT_PARTS split items.
// Split items come first in `order` (item index = scratch slot), then the whole tiles, heaviest bucket first.
__global__ __launch_bounds__(1024) void tile_order_kernel(TileAdamArgs a, int n_tiles) {
    __shared__ int hist[32], start[32];
    __shared__ int total, n_split;
    const int tid = threadIdx.x;
    int* n_items = a.bins.split_state ? a.bins.split_state + n_tiles : nullptr;
    if (tid < 32) hist[tid] = 0;
    if (tid == 0) { total = 0; n_split = 0; }
    __syncthreads();
    // Every list length is read from memory exactly ONCE: the kernel may run while the deferred rays' appends are still
    // being added on another stream (bins.counts grows under it), and a tile whose length changed between the classify and
    // the place pass would be counted in one bucket and placed in another.  The first ORDER_REGS * 1024 tiles
    // stay in registers for all three passes, the next ORDER_LDS in LDS; a scene with more tiles than that
    // (MNE_TILE_ORDER_SNAPSHOT) re-reads the rest, and the caller must then order AFTER the last append (FusedStep does).
    constexpr int ORDER_REGS = 8, ORDER_LDS = MNE_TILE_ORDER_SNAPSHOT - ORDER_REGS * 1024;
    __shared__ int lcache[ORDER_LDS];
    // With prev_counts a list counts as max(length now, final length of the previous launch): the kernel may run while the
    // deferred rays' appends are still to come (scenes with many deferred rays would otherwise get yesterday's heavy lists
    // unsplit: ScanNet 1.16 ms instead of 0.44 ms).
    // (Lengths are NOT clamped to the plane's list capacity: an overflowing list -- rare, a capacity is 4x the mean -- is
    // then cut into a few more parts than its in-list entries need, which is harmless; finding the plane of every tile
    // here cost 8 us of this latency-critical kernel.)
    auto length0 = [&](int t) {
        int c = 0;                                                   // the list's segments, one per XCD
#if MNE_LIST_SEGMENTS == 8
        const int4 c0 = *(const int4*)(a.bins.counts + (size_t)t * 8), c1 = *(const int4*)(a.bins.counts + (size_t)t * 8 + 4);      // two loads, not eight
        c = (c0.x + c0.y + c0.z + c0.w) + (c1.x + c1.y + c1.z + c1.w);
#else
#pragma unroll
        for (int x = 0; x < MNE_LIST_SEGMENTS; ++x) c += a.bins.counts[(size_t)t * MNE_LIST_SEGMENTS + x];
#endif
        if (!a.prev_counts) return c;
        const int p = a.prev_counts[t];
        return c > p ? c : p;
    };
    auto length = [&](int t) {                       // tiles beyond the registers: first pass fills the LDS snapshot
        const int k = t - ORDER_REGS * 1024;
        return k < ORDER_LDS ? lcache[k] : length0(t);
    };
    int cnt[ORDER_REGS];
    int mine = 0;
#pragma unroll
    for (int q = 0; q < ORDER_REGS; ++q) {
        const int t = tid + q * 1024;
        cnt[q] = t < n_tiles ? length0(t) : 0;
        mine += cnt[q];
    }
    for (int t = tid + ORDER_REGS * 1024; t < n_tiles; t += 1024) {      // (a thread reads back only what it wrote itself)
        const int c = length0(t), k = t - ORDER_REGS * 1024;
        if (k < ORDER_LDS) lcache[k] = c;
        mine += c;
    }
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d);
    if ((tid & 63) == 0 && mine) atomicAdd(&total, mine);
    __syncthreads();
    int split = 0x7fffffff;
    if (a.bins.split_scratch) {
        split = (int)((2ll * total + MNE_TILE_SPLIT_PARTS - 1) / MNE_TILE_SPLIT_PARTS);
        split = split < a.bins.split_min ? a.bins.split_min : split;
    }
    // split items first, whole tiles into their length buckets
    auto classify = [&](int t, int c0) {
        const int c = c0;
        if (c > split) {
            int np = (c + split - 1) / split;
            np = np > 63 ? 63 : np;
            const int base = atomicAdd(&n_split, np);
            for (int q = 0; q < np; ++q) a.bins.order[base + q] = (int)((unsigned)t | ((unsigned)q << 20) | ((unsigned)np << 26));
        } else {
            atomicAdd(&hist[31 - __clz(c0 + 1)], 1);
        }
    };
#pragma unroll
    for (int q = 0; q < ORDER_REGS; ++q)
        if (tid + q * 1024 < n_tiles) classify(tid + q * 1024, cnt[q]);
    for (int t = tid + ORDER_REGS * 1024; t < n_tiles; t += 1024) classify(t, length(t));
    __syncthreads();
    if (tid < 64) {                                   // exclusive scan over the 32 buckets, heaviest first, by one wave
        const int b = 31 - (tid & 31);
        int v = tid < 32 ? hist[b] : 0, inc = v;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) { const int u = __shfl_up(inc, d); if ((tid & 31) >= d) inc += u; }
        if (tid < 32) start[b] = n_split + inc - v;
        if (tid == 31 && n_items) *n_items = n_split + inc;
    }
    __syncthreads();
    auto place = [&](int t, int c0) {
        const int c = c0;
        if (c <= split) a.bins.order[atomicAdd(&start[31 - __clz(c0 + 1)], 1)] = t;
    };
#pragma unroll
    for (int q = 0; q < ORDER_REGS; ++q)
        if (tid + q * 1024 < n_tiles) place(tid + q * 1024, cnt[q]);
    for (int t = tid + ORDER_REGS * 1024; t < n_tiles; t += 1024) place(t, length(t));
}
// (A ballot-ranked counting sort without same-address atomics was measured at 14.5 us against 12 us for this one.)

// OV (EXTENSION, multi-agent: gradients of the plane cells two agents both map, mne_tile_overlap_t) --
//   0  the mapping iteration of one agent (the only form the reference has);
//   1  EXPORT: the tiles that meet a shared rectangle accumulate their lists as usual and write the gradient of the shared
//      cells into the send buffers; no Adam, list counters untouched (the same lists are walked again by form 2);
//   2  form 0 with the peers' gradients of the shared cells (recv buffers) added before the Adam update.
// The shared cells of a plane are a rectangle of NODES in this agent's indices (the agents sit on one lattice,
// mneslam_amd/dist.py::overlap_slices); buffers hold them plane after plane as [rows][cols][32].
struct NoOverlap {};
template <int OV> struct OvArg { typedef TileOverlap type; };
template <> struct OvArg<0> { typedef NoOverlap type; };

template <int OV>
__global__ __launch_bounds__(TILE_THREADS, TILE_MIN_WAVES) void tile_adam_kernel(TileAdamArgs a, typename OvArg<OV>::type ov) {
    MNE_DYN_LDS(lds_raw);
    float* g = (float*)lds_raw;                                   // [16][16][32] gradient tile, 32 KiB
    float* stage = g + TILE_CELLS * MNE_C;                        // [PASS_ENTRIES][32] gradient rows of this pass
    __shared__ unsigned erow[PASS_ENTRIES];                       // tape row of each staged entry (~0u: slot unused)
    __shared__ unsigned short skey[PASS_ENTRIES * 4];             // contributions in cell order: cell << 8 | item
    __shared__ float swt[PASS_ENTRIES * 4];                       //   ... and their bilinear weights
    __shared__ int hist[TILE_CELLS];                              // per-cell counts, then start offsets
    __shared__ int n_contrib;
    const int tid = threadIdx.x;
    if (a.bins.split_state && (int)blockIdx.x >= a.bins.split_state[a.n_tiles]) return;       // the grid covers the item CAPACITY
    const unsigned item = (unsigned)a.bins.order[blockIdx.x];
    const int tile = (int)(item & 0xfffffu), part = (int)((item >> 20) & 63u), n_parts = (item >> 26) ? (int)(item >> 26) : 1;
    int pidx = 0;
    while (pidx + 1 < a.n_planes && tile >= a.bins.tile_base[pidx + 1]) ++pidx;
    const int set = pidx / 6, lvl = pidx % 2;                                 // [set][orient][level]
    const mne_plane_t& pl = a.sc.plane[set][(pidx % 6) / 2][lvl];
    const int local = tile - a.bins.tile_base[pidx];
    const int tx0 = local % a.bins.ntx[pidx], ty0 = local / a.bins.ntx[pidx];
    // the list's MNE_LIST_SEGMENTS segments (one per XCD, appended to by that XCD's waves): cursor of each, entries it holds
    // (a cursor beyond the segment's capacity = the rest of its entries went to the spill area), start of each in the flat
    // entry numbering of the list
    const int cap = a.bins.pcap[pidx], seg_cap = cap / MNE_LIST_SEGMENTS;
    int seg_start[MNE_LIST_SEGMENTS + 1];
    int cnt = 0;
    bool overflow = false;
    {
        int run = 0;
#pragma unroll
        for (int x = 0; x < MNE_LIST_SEGMENTS; ++x) {
            const int cx = a.bins.counts[(size_t)tile * MNE_LIST_SEGMENTS + x];      // (wave-uniform: scalar loads)
            seg_start[x] = run;
            run += cx < seg_cap ? cx : seg_cap;
            cnt += cx;
            overflow = overflow || cx > seg_cap;
        }
        seg_start[MNE_LIST_SEGMENTS] = run;
    }
    bool shared_tile = false;                                  // does this tile hold cells of a shared rectangle?
    if constexpr (OV != 0) {
        for (int k = 0; k < ov.n_peers; ++k) {
            const OverlapRect& r = ov.rect[k][pidx];
            shared_tile = shared_tile || (r.x1 > r.x0 && tx0 * MNE_TILE < r.x1 && (tx0 + 1) * MNE_TILE > r.x0 &&
                                          ty0 * MNE_TILE < r.y1 && (ty0 + 1) * MNE_TILE > r.y0);
        }
        if (OV == 1 && !shared_tile) return;                   // export: only the shared tiles have anything to say
    }
    const bool empty = TILE_EMPTY_FAST && cnt == 0;            // no contribution: the sweep runs with g = 0, LDS untouched
    if (a.live) {
        // A tile that has NEVER received a gradient has m = v = 0, and Adam then leaves it exactly as it is:
        // p - step_size * (0 / (sqrt(0) / bc2 + eps)) = p, m and v stay 0 (plane groups carry no weight decay; checked).
        // Such tiles -- the margin of the extended bound, the part of the volume no keyframe has looked into yet: 30 % of
        // office0's parameters in every iteration, profiles/r04_empty_tiles.txt -- are neither read nor written.
        if (cnt == 0 && !shared_tile && a.live[tile] == 0 && a.opt[pidx].wd == 0.0f) {
            if (tid == 0 && a.prev_counts) a.prev_counts[tile] = 0;
            return;
        }
        if (cnt > 0 && tid == 0 && OV != 1) a.live[tile] = 1;
    }
    if (!empty)
        for (int i = tid; i < TILE_CELLS * MNE_C / 4; i += TILE_THREADS) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int n_list = seg_start[MNE_LIST_SEGMENTS];
    int n_spill = 0;
    if (overflow) {                      // only a tile with an overflowed segment has entries in the spill area
        const int ns = *a.bins.spill_count;
        n_spill = ns < a.bins.spill_cap ? ns : a.bins.spill_cap;
    }
    const unsigned* lst = a.bins.lists + (size_t)(a.bins.list_off[pidx] + (long long)tile * cap) * MNE_ENTRY_WORDS;
    constexpr int NIT = (TILE_CELLS * MNE_C / 4) / TILE_THREADS;
    PlaneOpt o = a.opt[pidx];
    if (a.clk.bias_table) clock_bias(a.clk, o.lr, o.step, o.step_size, o.bc2_sqrt);       // graph replay: step from device memory
    // (requesting the sweep's operands before / behind the first list pass was measured at three depths: no gain,
    // profiles/r03_prefetch_and_decode12_negative.txt)
    // lane layout of the row work: 8 lanes x float4 = the 32 channels of one gradient row
    const int sub = tid & 7, grp = tid >> 3;
    const float* dfeat = a.tape + a.t_dfeat + set * MNE_FEAT + lvl * MNE_C + sub * 4;
    // passes over the list, then over the spill area (spill entries of other tiles contribute nothing)
    const int n_total = n_list + n_spill;
    // this part's share of the list; the last part also scans the spill area
    // (every bound is clamped to the list: with a tiny split threshold the part count is capped and part * chunk can pass
    // the list's end -- the last part must then still start at the FIRST spill entry)
    const int chunk = (n_list + n_parts - 1) / n_parts;
    const int e_lo = part * chunk < n_list ? part * chunk : n_list;
    const int e_hi = part == n_parts - 1 ? n_total : ((part + 1) * chunk < n_list ? (part + 1) * chunk : n_list);
    for (int p0 = e_lo; p0 < e_hi; p0 += PASS_ENTRIES) {
        for (int i = tid; i < TILE_CELLS; i += TILE_THREADS) hist[i] = 0;
        __syncthreads();
        // ---- A: this thread's entry (kept in registers), its contributions ranked per cell
        int cellk[4] = {-1, -1, -1, -1}, rank[4] = {0, 0, 0, 0};
        float wq[4] = {0.f, 0.f, 0.f, 0.f};
        const int e = p0 + tid;
        if (tid < PASS_ENTRIES) {
            unsigned row = 0xffffffffu;
            const unsigned* ent = nullptr;
            if (e < e_hi && e < n_list) {
                int x = 0, s0 = 0;                                       // segment of flat entry e and the segment's first entry
#pragma unroll                                                          // (static register indices only)
                for (int k = 1; k < MNE_LIST_SEGMENTS; ++k) {
                    const bool ge = e >= seg_start[k];
                    x += ge ? 1 : 0;
                    s0 = ge ? seg_start[k] : s0;
                }
                ent = lst + ((size_t)x * seg_cap + (size_t)(e - s0)) * MNE_ENTRY_WORDS;
            }
            else if (e < e_hi) ent = a.bins.spill + (size_t)(e - n_list) * MNE_ENTRY_WORDS;
            if (ent) {
                const uint4 e0 = *(const uint4*)ent, e1 = *(const uint4*)(ent + 4);      // one 32-byte entry
                if (e < n_list || e1.z == (unsigned)tile) {                              // spill entries carry their tile id
                    row = e0.x;
                    const int lx = (int)(e0.y & 0xff) - 1, ly = (int)((e0.y >> 8) & 0xff) - 1;
                    const unsigned wbits[4] = {e0.z, e0.w, e1.x, e1.y};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int x = lx + (q & 1), y = ly + (q >> 1);
                        wq[q] = __uint_as_float(wbits[q]);
                        if (x >= 0 && x < MNE_TILE && y >= 0 && y < MNE_TILE && wq[q] != 0.0f) {
                            cellk[q] = y * MNE_TILE + x;
                            rank[q] = atomicAdd(&hist[cellk[q]], 1);
                        }
                    }
                }
            }
            erow[tid] = row;
        }
        __syncthreads();
        // ---- every entry's gradient row (this plane level: 32 floats) is fetched ONCE per pass, by the
        // 8-lane groups round-robin, all loads in flight together; its (up to four) corner contributions
        // then read it from LDS.  The loads overlap steps B and C.
        float4 grow[PASS_ENTRIES / TILE_GROUPS];
#pragma unroll
        for (int j = 0; j < PASS_ENTRIES / TILE_GROUPS; ++j) {
            const unsigned row = erow[j * TILE_GROUPS + grp];
            grow[j] = row != 0xffffffffu ? *(const float4*)(dfeat + (size_t)row * a.row_stride) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // ---- B: counts -> exclusive start offsets (one wave, 4 cells per lane)
        if (tid < MNE_WAVE) {
            const int v0 = hist[4 * tid], v1 = hist[4 * tid + 1], v2 = hist[4 * tid + 2], v3 = hist[4 * tid + 3];
            const int sum = v0 + v1 + v2 + v3;
            int inc = sum;
#pragma unroll
            for (int d = 1; d < MNE_WAVE; d <<= 1) {
                const int up = __shfl_up(inc, d);
                if (tid >= d) inc += up;
            }
            const int ex = inc - sum;
            hist[4 * tid] = ex; hist[4 * tid + 1] = ex + v0; hist[4 * tid + 2] = ex + v0 + v1; hist[4 * tid + 3] = ex + v0 + v1 + v2;
            if (tid == MNE_WAVE - 1) n_contrib = inc;
        }
        __syncthreads();
        // ---- C: contribution records in cell order
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (cellk[q] >= 0) {
                const int pos = hist[cellk[q]] + rank[q];
                skey[pos] = (unsigned short)((cellk[q] << 8) | tid);
                swt[pos] = wq[q];
            }
#pragma unroll
        for (int j = 0; j < PASS_ENTRIES / TILE_GROUPS; ++j) *(float4*)(stage + (j * TILE_GROUPS + grp) * MNE_C + sub * 4) = grow[j];
        __syncthreads();
        // ---- D: equal ranges of the sorted contributions, one per 8-lane group; a run of equal cells is
        // summed by the group in whose range it STARTS (that group reads on past its range end, the next
        // one skips to the end of the run using the start offsets), so the LDS tile is only ever
        // updated with plain read-add-write by a single owner.  (ds_add_f32 is very slow here: with
        // atomic boundary runs this step took 3x longer -- profiles/r01_tile_adam_phases.txt.)
        const int nC = n_contrib;
        const int chunk = (nC + TILE_GROUPS - 1) / TILE_GROUPS;
        const int b0 = grp * chunk, b1 = b0 + chunk < nC ? b0 + chunk : nC;
        int q0 = b0;
        if (b0 > 0 && b0 < nC) {
            const int pc = (int)(skey[b0 - 1] >> 8);
            q0 = pc == TILE_CELLS - 1 ? nC : hist[pc + 1];
        }
        bool done = q0 >= b1;
        int cur = -1;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (; !done; q0 += DB) {
            unsigned key[DB];
            float wt[DB];
            float4 gv[DB];
#pragma unroll
            for (int j = 0; j < DB; ++j) {
                const int q = q0 + j < nC ? q0 + j : nC - 1;
                key[j] = skey[q];
                wt[j] = swt[q];
            }
#pragma unroll
            for (int j = 0; j < DB; ++j) gv[j] = *(const float4*)(stage + (key[j] & 255u) * MNE_C + sub * 4);
#pragma unroll
            for (int j = 0; j < DB; ++j) {
                const int cell = (int)(key[j] >> 8);
                if (!done && (q0 + j >= nC || (q0 + j >= b1 && cell != cur))) done = true;
                if (!done) {
                    if (cell != cur) {
                        if (cur >= 0) {                                   // a finished run
                            float* gp = g + cur * MNE_C + sub * 4;
                            float4 t = *(float4*)gp;
                            t.x += acc.x; t.y += acc.y; t.z += acc.z; t.w += acc.w;
                            *(float4*)gp = t;
                        }
                        cur = cell;
                        acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    acc.x = fmaf(gv[j].x, wt[j], acc.x); acc.y = fmaf(gv[j].y, wt[j], acc.y);
                    acc.z = fmaf(gv[j].z, wt[j], acc.z); acc.w = fmaf(gv[j].w, wt[j], acc.w);
                }
            }
        }
        if (cur >= 0) {
            float* gp = g + cur * MNE_C + sub * 4;
            float4 t = *(float4*)gp;
            t.x += acc.x; t.y += acc.y; t.z += acc.z; t.w += acc.w;
            *(float4*)gp = t;
        }
        __syncthreads();
    }
    if (!empty) __syncthreads();
    if (n_parts > 1) {
        // ---- partial tile -> split scratch; the part that arrives last sums all of them and goes on to Adam.
        // Hand-off between workgroups on possibly different XCDs (L2s not coherent): plain stores, every wave drains
        // its stores, one lane releases at agent scope, then takes a ticket; the last arriver acquires and reads.
        float* slab = a.bins.split_scratch + (size_t)blockIdx.x * (TILE_CELLS * MNE_C);
        for (int i = tid; i < TILE_CELLS * MNE_C / 4; i += TILE_THREADS) ((float4*)slab)[i] = ((const float4*)g)[i];
        MNE_DRAIN_STORES();
        __syncthreads();
        if (tid == 0) {
            MNE_FENCE_RELEASE_AGENT();
            MNE_DRAIN_STORES();
            n_contrib = atomicAdd(a.bins.split_state + tile, 1);                 // (n_contrib: the one small LDS word, reused as the ticket)
        }
        __syncthreads();
        if (n_contrib != n_parts - 1) return;
        if (tid == 0) MNE_FENCE_ACQUIRE_AGENT();
        __syncthreads();
        const float* first = a.bins.split_scratch + (size_t)(blockIdx.x - part) * (TILE_CELLS * MNE_C);   // parts are consecutive items
        for (int q = 0; q < n_parts; ++q) {
            if (q == part) continue;
            const float4* other = (const float4*)(first + (size_t)q * (TILE_CELLS * MNE_C));
            for (int i = tid; i < TILE_CELLS * MNE_C / 4; i += TILE_THREADS) {
                float4 t = ((float4*)g)[i];
                const float4 u = other[i];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
                ((float4*)g)[i] = t;
            }
        }
        __syncthreads();
        if (tid == 0) a.bins.split_state[tile] = 0;                                  // arrival counter ready for the next call
    }
    if constexpr (OV == 1) {                                   // ---- export the shared cells' gradient, nothing else
        for (int it = 0; it < NIT; ++it) {
            const int i4 = it * TILE_THREADS + tid;
            const int y = i4 / (MNE_TILE * MNE_C / 4), x4 = i4 % (MNE_TILE * MNE_C / 4);
            const int gy = ty0 * MNE_TILE + y, gx = tx0 * MNE_TILE + x4 / (MNE_C / 4);
            const float4 gg = empty ? make_float4(0.f, 0.f, 0.f, 0.f) : ((const float4*)g)[i4];
            for (int k = 0; k < ov.n_peers; ++k) {
                const OverlapRect& r = ov.rect[k][pidx];
                if (gx >= r.x0 && gx < r.x1 && gy >= r.y0 && gy < r.y1)
                    *(float4*)(ov.send[k] + r.off + ((size_t)(gy - r.y0) * (r.x1 - r.x0) + (gx - r.x0)) * MNE_C + (x4 % (MNE_C / 4)) * 4) = gg;
            }
        }
        return;
    }
    // ---- Adam on the tile: 16 rows x (16 cells x 32 ch) = 2048 float4
    const bool f16 = a.sc.plane_f16 != 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i4 = it * TILE_THREADS + tid;
        const int y = i4 / (MNE_TILE * MNE_C / 4), x4 = i4 % (MNE_TILE * MNE_C / 4);
        const int cell = x4 / (MNE_C / 4), ch4 = x4 % (MNE_C / 4);
        const int gy = ty0 * MNE_TILE + y, gx = tx0 * MNE_TILE + cell;
        if (gy < pl.h && gx < pl.w) {
            const size_t off = ((size_t)gy * pl.w + gx) * MNE_C + ch4 * 4;
            const size_t offmv = off;    // (moments stored tile-major were measured: no DRAM-locality effect, profiles/r02_tile_adam_variants.txt)
            // parameters: fp32, or half precision (mne_scene_t.plane_f16: p32 = float(p16) -> Adam in fp32 -> round to nearest)
            float4 p = f16 ? half4_to_float4(*(const uint2*)((const _Float16*)pl.data + off)) : nt_load4<(MNE_ADAM_NT & 4) != 0>((const float*)pl.data + off);
            float4 m = nt_load4<(MNE_ADAM_NT & 2) != 0>(o.m + offmv), v = nt_load4<(MNE_ADAM_NT & 2) != 0>(o.v + offmv);
            float4 gg = empty ? make_float4(0.f, 0.f, 0.f, 0.f) : ((const float4*)g)[i4];
            if constexpr (OV == 2) {
                if (shared_tile) {
                    // A shared cell takes its OWN share from the send buffer, i.e. exactly the value the peer received, not
                    // from this launch's LDS sum (same lists, but the summation order inside a cell is not fixed): both
                    // agents then add the same two numbers and their shared cells stay bit-equal.
                    bool own_taken = false;
                    for (int k = 0; k < ov.n_peers; ++k) {
                        const OverlapRect& r = ov.rect[k][pidx];
                        if (gx >= r.x0 && gx < r.x1 && gy >= r.y0 && gy < r.y1) {
                            const size_t at = r.off + ((size_t)(gy - r.y0) * (r.x1 - r.x0) + (gx - r.x0)) * MNE_C + ch4 * 4;
                            if (!own_taken) { gg = *(const float4*)(ov.send[k] + at); own_taken = true; }
                            const float4 t = *(const float4*)(ov.recv[k] + at);
                            gg.x += t.x; gg.y += t.y; gg.z += t.z; gg.w += t.w;
                        }
                    }
                }
            }
            adam_elem(p.x, gg.x, m.x, v.x, o); adam_elem(p.y, gg.y, m.y, v.y, o);
            adam_elem(p.z, gg.z, m.z, v.z, o); adam_elem(p.w, gg.w, m.w, v.w, o);
            if (f16) *(uint2*)((_Float16*)pl.data + off) = float4_to_half4(p);
            else nt_store4<(MNE_ADAM_NT & 8) != 0>((float*)pl.data + off, p);
            nt_store4<(MNE_ADAM_NT & 1) != 0>(o.m + offmv, m); nt_store4<(MNE_ADAM_NT & 1) != 0>(o.v + offmv, v);
        }
    }
    if (tid == 0 && a.prev_counts) a.prev_counts[tile] = cnt;
    if (tid < MNE_LIST_SEGMENTS) a.bins.counts[(size_t)tile * MNE_LIST_SEGMENTS + tid] = 0;      // ready for the next iteration
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

int mne_launch_tile_order(const TileAdamArgs& a, hipStream_t st) {
    const int n_tiles = a.bins.tile_base[a.n_planes];
    if (n_tiles > 0) MNE_LAUNCH(tile_order_kernel, 1, 1024, 0, st, a, n_tiles);
    return 0;
}

int mne_launch_tile_adam(const TileAdamArgs& a, hipStream_t st, const TileOverlap* ov, int form) {
    const int n_tiles = a.bins.tile_base[a.n_planes];
    if (n_tiles <= 0) return 0;
    const size_t lds = (size_t)(TILE_CELLS + PASS_ENTRIES) * MNE_C * sizeof(float);
    if (lds > 32 * 1024) {
        MNE_SET_MAX_LDS((tile_adam_kernel<0>), lds); MNE_SET_MAX_LDS((tile_adam_kernel<1>), lds); MNE_SET_MAX_LDS((tile_adam_kernel<2>), lds);
    }        // static LDS (keys, weights, counters) comes on top
    const int grid = n_tiles + (a.bins.split_scratch ? MNE_TILE_SPLIT_PARTS : 0);          // item capacity; surplus workgroups leave at once
    if (!ov) MNE_LAUNCH((tile_adam_kernel<0>), grid, TILE_THREADS, lds, st, a, NoOverlap{});
    else if (form == 1) MNE_LAUNCH((tile_adam_kernel<1>), grid, TILE_THREADS, lds, st, a, *ov);
    else MNE_LAUNCH((tile_adam_kernel<2>), grid, TILE_THREADS, lds, st, a, *ov);
    return 0;
}
