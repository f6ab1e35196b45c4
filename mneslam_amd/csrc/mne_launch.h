// mne_launch.h -- kernel argument blocks and host-side launchers shared by the .hip translation
// units and the C ABI (capi.cpp).  Internal; the public surface is include/mneslam_hip.h.
#pragma once
#include "mne_platform.h"
#include "mneslam_hip.h"

// device-side view of mne_clock_t (all NULL / 0 = values come from the kernel arguments)
struct Clock {
    const unsigned long long* iteration;
    const int* step_offset;
    const double* bias_table;
    int n_table;
    unsigned long long z_offset_stride;
};

struct ZArgs {
    int R, S, n_a, n_b, has_d;
    float perturb;
    float e_T, e_T04, co_T, depth_trunc;
    const float* target_d;
    const float* u;
    const float* tables;     // has_d: uniform[n_a] | surface offsets[n_b] | invalid-depth[n_b];  else full[S]
    uint64_t seed, offset;
    float* z_vals;
    int* counts;
    int* ray_counts;     // [R][MNE_N_COUNT] scratch
    Clock clk;
};

#define MNE_TILE 16            // plane tile edge (cells) of the binned scatter
#define MNE_MAX_PLANES 12
#define MNE_ENTRY_WORDS 8       // list / spill entry (32 B): tape row | (lx+1)|(ly+1)<<8 | 4 bilinear weights | tile id | 0
#define MNE_SPILL_WORDS 8

// Every tile list is cut into MNE_LIST_SEGMENTS = 8 segments, one per XCD, each with its own cursor: the waves of one XCD
// append to their own segment.  Returning atomics on ONE address retire one after the other at ~12 ns each
// (profiles/r05_xcd_atomics.txt) and every ray of a keyframe passes through the few tiles around its camera centre; eight
// cursors per list divide that queue by eight (bin_kernel 68 -> 50 us on office0, 153 -> 108 on ScanNet, 268 -> 160 on INS
// Indoor in the timing experiment, profiles/r05_bin_xcd_experiment.txt).
#ifndef MNE_LIST_SEGMENTS
#define MNE_LIST_SEGMENTS 8
#endif
struct TileBins {
    unsigned* lists;          // [n_tiles][MNE_LIST_SEGMENTS][cap / MNE_LIST_SEGMENTS][MNE_ENTRY_WORDS]
    int* counts;              // [n_tiles][MNE_LIST_SEGMENTS] cursors (reset by tile_adam_kernel)
    unsigned* spill;          // [spill_cap][MNE_SPILL_WORDS] overflow entries
    int* spill_count;
    int* order;               // [n_tiles] processing order of tile_adam_kernel (heaviest lists first)
    int cap, spill_cap;       // cap: capacity of every list when the caller gives no per-plane capacities
    int* dropped;             // sticky count of entries lost to a full spill area
    float* split_scratch;     // [MNE_TILE_SPLIT_PARTS][16*16*32] partial gradient tiles of split lists (NULL: never split)
    int* split_state;         // [n_tiles + 1]: arrival counters per tile (zero between calls), [n_tiles] = number of work items
    int split_min;            // lists up to this length are never split
#define MNE_TILE_SPLIT_MIN_DEFAULT 4096   // a part costs a 32 KiB slab round trip + ~5 us of fences: 256..2048 cost office0 1-3 %,
                                          // 8192 costs Indoor 3 % (profiles/r02_tile_split_min.txt)
    int tile_base[MNE_MAX_PLANES + 1];   // first tile id of each plane ([set][orient][level] order)
    int ntx[MNE_MAX_PLANES];             // tiles per plane row
    // list capacity per plane (a coarse plane's few tiles take thousands of entries each, a fine plane's many tiles a few
    // hundred): list of tile t of plane p = lists + (list_off[p] + (long long)t * pcap[p]) entries  (t = GLOBAL tile id;
    // list_off already has tile_base[p] * pcap[p] taken off)
    int pcap[MNE_MAX_PLANES];
    long long list_off[MNE_MAX_PLANES];
};

struct GridArgs;
struct RenderArgs {
    mne_scene_t sc;
    int R, S;
    float trunc_f;        // (float)training.trunc                  (sdf / trunc)
    float win_f;          // (float)(data.sc_factor*training.trunc) (render window, Co-SLAM truncation)
    float e_T, e_T04;     // (float)model.truncation, (float)(0.4*model.truncation)
    float depth_trunc;
    const float *rays_o, *rays_d, *target_rgb, *target_d, *z_vals, *packed;
    float *rgb, *depth, *disp, *acc, *depth_var, *raw, *ray_sums;
    const float* raw_in;        // backward-only call: raw of ALL samples from the forward call (NULL otherwise)
    int lds_samples;             // ray_kernel: samples of a ray its LDS arrays hold (0 = all S); rays that need more are deferred
#if defined(MNE_ARGS_PAD) && MNE_ARGS_PAD > 0
    // LAYOUT FUZZ (tests only, -DMNE_ARGS_PAD=N builds under mneslam_amd/_fuzz/): dummy bytes in the middle of the kernel
    // argument block.  No kernel reads them; a kernel whose results change with N is miscompiled or racy.
    char args_pad[MNE_ARGS_PAD];
#endif
    int ext_feat;                // feature rows come from the caller (tape columns T_X..): no plane gather, no plane scatter
    const float* ext_rows;       // forward-only calls with caller-supplied features: [R*S][ext_stride] rows, 64 floats used
    int ext_stride;
    const int* ray_counts;      // [R][MNE_N_COUNT] from sample_z (slot MNE_C_NEED = a-priori sample count)
    int prefix_default;         // ray_counts == NULL: a-priori tiles of every ray (ntile = decode everything, 1 = on demand only)
    int frame_min_tiles;         // (host only) forward decode: rays per wave from which decode_frame_kernel is used (0: build default, < 0: always; tests)
    const float *coef, *g_rgb, *g_depth;
    float* tape;                // [R*S][ROW]; NULL = forward only
    int tape_row, tape_tx, tape_tcf;   // gather_kernel: row length and the columns of the two feature blocks
    int tape_tpn, tape_tdfeat;         // bin_kernel / scatter_kernel: columns of (pn.xyz, live flag) and of the d(feature) rows
    int plane_grads;                   // the backward leaves d(feature) rows of the samples with gradient in the tape (0: pose-only loops)
    int* tape_rows;             // total number of samples that received gradient
    unsigned* relu_mask;        // [R*S][4]: per lane half (h mask, hc mask)
    int* ray_tiles;             // [R] number of leading 32-sample tiles of each ray whose tape rows are complete
    int* dec_tiles;             // [R] leading tiles of each ray decode_kernel really decoded (a-priori prefix + its extension)
    const int* tile_need;       // [R] backward of an earlier forward call: tiles of each ray whose tape rows the backward walks
                                // (tile_need_kernel): the decode makes exactly these, tile-parallel (NULL: the a-priori prefix)
    int* defer_list;            // [R] rays the training kernel could not resolve from the decoded prefix
    int* defer_count;           // [1]
    int* long_list;             // [R] rays whose decoded prefix exceeds the first pass's LDS sample cap (long rays only)
    int* long_count;            // [1]
    int list_keeps_prefix;      // list pass over long_list: tiles decoded so far = dec_tiles[r] (not "everything")
    const int* ray_list;        // ray_kernel works through this list instead of all rays (NULL = rays 0..R-1)
    const int* ray_list_count;
    float *d_rays_o, *d_rays_d;
    int adapt_update;           // this launch is the last ray launch of the call: it rewrites adapt[0] for the next call
    int* adapt;                 // optional [4] device words: [0] = decode every sample a priori in this call, [1] = rays the a-priori
                                // prefix did / would not resolve (mne_fused_opts_t::adapt_state)
    TileBins bins;              // bins.lists != NULL: binned scatter instead of atomics into plane[].grad
    // rays with more than heavy_min backward tiles (0 = never): the training kernel lists them with their gradient constants,
    // heavy_bwd_kernel walks their tiles tile-parallel (render.hip)
    int heavy_min;              // (in: 0 = the build's default, MNE_HEAVY_TILES)
    int heavy_ntile;            // the heavy list exists for rays of more than this many tiles (0 = default MNE_HEAVY_NTILE, < 0 = never)
    int* heavy_list;            // [R]
    int* heavy_count;           // [1]
    float* heavy_rec;           // [R][8]: denom, z_lim, Aq, g_dep | g_rgb[3], decoded samples (int bits)
};

// host-side extras of one training render (never part of a kernel argument block)
struct RenderHost {
    const struct GridArgs* ext_grid = nullptr;   // ext_feat: hash grid whose rows the call gathers itself (NULL = the caller filled the tape)
    void* const* marks = nullptr;                // hipEvent_t handles recorded between the kernels (mne_fused_opts_t::timing_events)
    int n_marks = 0;
    int external_bin = 0;                        // the caller runs the list appends itself (mne_tile_bin)
    void* ev_after_decode = nullptr;             // hipEvent_t recorded once the prefix decode is enqueued
    int features_pregathered = 0;                // ext_grid: the caller already gathered the first pass's rows (mne_hash_gather with ray_counts)
};

struct LossArgs {
    int R, S;
    const float* ray_sums;
    const int* counts;
    float* losses;
    const float* grad_losses;
    float* coef;
    float e_T, co_T;
};

struct QueryArgs {
    mne_scene_t sc;
    long long n;
    const float* pts;
    const float* packed;
    float *raw, *geo, *feat_out;
    const float* ext_rows;   // optional [n][64]: the points' feature rows come from the caller (hash / dense grid) instead of the planes
    int* corner_idx;     // optional [n][3*n_sets][2][2]: (ix0, iy0) per plane
    int flags;
};

struct SampleRaysArgs {
    const float* kf_rays;        // [n_kf_rays][7]  (dir3, rgb3, depth1), keyframe-major
    long long n_kf_rays;
    int n_save;                  // rays stored per keyframe (owner keyframe = idx / n_save)
    const int* kf_pose_ids;      // optional map keyframe slot -> row of `poses` (NULL = identity)
    const float* cur_rays;       // [n_cur_rays][7] current frame
    long long n_cur_rays;
    const float* poses;          // [n_poses][4][4] c2w; the current frame's pose is the LAST row
    int n_poses, n_global, n_cur;
    const long long* idx_global; // optional explicit indices (host RNG) instead of the device permutation
    const long long* idx_cur;
    long long* out_idx;          // optional [R]: the indices used
    float *rays_o, *rays_d, *target_rgb, *target_d;
    int half_bits_kf, half_bits_cur;
    unsigned long long seed, iteration;      // keys are derived from (seed, iteration [+ clock]) inside the kernel
    Clock clk;
};

struct PlaneOpt { float* m; float* v; float omb1, b2, omb2, eps, wd, step_size, bc2_sqrt; double lr; int step; };

// Tiles the resolver wave of decode_kernel may decode beyond a ray's a-priori prefix before it leaves the rest to the
// deferred pass (render.hip); the hash gather fills exactly that many tiles ahead.
#ifndef MNE_RESOLVER_MAX_EXT
#define MNE_RESOLVER_MAX_EXT 1
#endif
// ... with colour planes: the extension tile's inline gather walks two plane sets (48 corner rows per sample, serial in the
// one wave), and leaving the unresolved rays to the tile-parallel deferred pass is cheaper (round 6, profiles/r06_resolver_ext_cp.txt:
// ScanNet scene0000 decode_kernel 121 -> 94 us, 1180 -> 1209 it/s; office0 without colour planes loses 3.5 % that way, INS Indoor +-0)
#ifndef MNE_RESOLVER_MAX_EXT_CP
#define MNE_RESOLVER_MAX_EXT_CP 0
#endif
// ... on caller-supplied features (the hash-grid iteration): every extension tile a ray MIGHT take has to be gathered up front for all rays (mne_hash_gather
// fills a-priori tiles + extension), 43 % more rows for the few rays that extend; without it the gather is 25 us and the decode 9 us shorter and the deferred pass
// 28 us longer: +0.9 % (round 6, profiles/r06_resolver_ext_cp.txt)
#ifndef MNE_RESOLVER_MAX_EXT_FEAT
#define MNE_RESOLVER_MAX_EXT_FEAT 0
#endif
template <bool CP> struct ResolverExt { static constexpr int MAX = CP ? MNE_RESOLVER_MAX_EXT_CP : MNE_RESOLVER_MAX_EXT; };
#define MNE_GRID_MAX_LEVELS 32
#define MNE_GRID_MAX_F 8
struct GridArgs {
    long long n;                 // points
    const float* x;              // [n][3] in [0,1]
    const float* params;         // flat table, level after level, F floats per entry
    float* out;                  // [n][out_stride] (n_levels*F used)
    unsigned* idx_out;           // optional [n][n_levels][8] table indices (within the level)
    const float* dout;           // backward: [n][n_levels*F]
    float* dparams;              // backward: same layout as params, accumulated into
    int n_levels, n_features, out_dim;
    int out_stride;              // row length of `out` / `dout` (0 = out_dim)
    int x_is_world;              // x holds world points: the grid input is (p - bb_lo) / (bb_hi - bb_lo), as the OneBlob input
    float scale[MNE_GRID_MAX_LEVELS];
    unsigned res[MNE_GRID_MAX_LEVELS], size[MNE_GRID_MAX_LEVELS], offset[MNE_GRID_MAX_LEVELS];
    // fused form (hash_rows_kernel): the points are the samples of a ray batch, the outputs rows of the tape
    const float* rays_o; const float* rays_d; const float* z_vals;      // [R][3], [R][3], [R][S]
    const int* ray_tiles;        // backward: only the first ray_tiles[r] tiles of ray r hold d(feature) rows
    float* tape;                 // [R*S][row_stride]; features at column col_x, d(feature) at column col_d
    int R, S, row_stride, col_x, col_d;
    double bb_lo[3], bb_hi[3];   // raw bounding box: x = (p - lo) / (hi - lo), as the OneBlob input
    int bb_is_f64;
    // slice form (hash_bin_kernel / hash_slice_adam_kernel): the backward rows' slice-binned records, Adam state of the table
    float* wgmax;                // [n_levels][n_chunks] largest |d(feature)| per level and chunk of rows
    double* gscale;              // [n_levels] inverse fixed-point scale of each level's gradient sums (for hash_finish_kernel)
    unsigned* seg_off;           // per level: [n_chunks][n_slices + 1] start of each slice's records in the chunk's segment
    size_t seg_level[MNE_GRID_MAX_LEVELS];   // first word of each level's block in seg_off
    unsigned* records;           // [n_levels][n_chunks][HASH_CHUNK * 8] 24-byte records (gridenc.hip: HashRecord), slice after slice
    int n_chunks;                // chunks of HASH_CHUNK tape rows: ceil(R*S / HASH_CHUNK)
    unsigned long long* scratch64;   // fixed-point gradient sums of the levels that are split over several workgroups per slice
    PlaneOpt opt;                // table optimizer state and step constants
    // gather restricted to the rows the exact early termination can decode (inside mne_render_fused_features)
    const int* ray_counts;       // [R][MNE_N_COUNT]: rows [0, (a-priori tiles + MNE_RESOLVER_MAX_EXT_FEAT) * 32) of every ray; NULL = all rows
    const int* ray_list;         // second pass: the remaining rows of the listed (deferred) rays
    const int* ray_list_count;
};

struct WgradArgs {
    const float* tape;        // [R*S][ROW]
    const int* ray_tiles;     // [R] leading tiles of each ray with complete tape rows (ray_kernel)
    int R, S;
    float* partials;          // [n_partials][NPARAM]
    float* grad_out;          // [NPARAM]
    int n_waves;              // number of partial results
};


// wgrad reduce + decoder Adam + loss scalars in one launch (render.hip, decoder_update_kernel)
struct DecUpdateArgs {
    mne_scene_t sc;              // decoder weights (updated in place) and dims
    const float* partials;       // [n_partials][NPARAM] from the fused weight-gradient pass
    int n_partials;
    float* grad_out;             // [NPARAM] the summed gradient (kept: tests and callers read it)
    float* m[4]; float* v[4];    // Adam moments in decoder.parameters() order: col0, col1, sdf0, sdf1
    PlaneOpt opt;
    LossArgs fin;                // loss scalars of the iteration (fin.losses NULL: not wanted)
    Clock clk;
};

struct TileAdamArgs {
    mne_scene_t sc;
    TileBins bins;
    PlaneOpt opt[MNE_MAX_PLANES];
    const float* tape;
    int row_stride, t_dfeat, t_pn;
    int n_planes, n_tiles;
    Clock clk;
    int* prev_counts;         // [n_tiles] final list lengths of the previous tile_adam launch (NULL: none) -- balance hint of tile_order
    int* live;                // [n_tiles] 1 = the tile has received a gradient at some point (its moments may be non-zero); NULL: every tile is swept
};

// EXTENSION (multi-agent): node rectangles of this agent's planes that a peer maps as well, with the exchange buffers
// (mne_tile_overlap_t of the C ABI + the float offset of each plane's rectangle inside a buffer).
struct OverlapRect { int x0, y0, x1, y1; long long off; };
struct TileOverlap {
    int n_peers;
    OverlapRect rect[MNE_MAX_OVERLAP_PEERS][MNE_MAX_PLANES];
    float* send[MNE_MAX_OVERLAP_PEERS];
    const float* recv[MNE_MAX_OVERLAP_PEERS];
};

// pose-alignment loop (csrc/pose.hip)
struct PoseArgs {
    int n;
    const float* dirs;          // [n][3] camera-frame directions
    int n_rot;                  // 3: axis-angle, 4: quaternion (w, x, y, z)
    float* rot;                 // [n_rot]
    float* trans;               // [3]
    float r_base[9];            // row-major
    float* c2w;                 // [12] row-major 3x4 of the CURRENT parameters
    float* rays_o;              // [n][3]
    float* rays_d;              // [n][3]
    // loss
    const float *rgb, *depth, *want_rgb, *want_depth;
    float w_rgb, w_depth;
    float* d_rgb;               // [n][3]
    float* d_depth;             // [n]
    float* partials;            // [n_partials] loss partial sums (one per workgroup of pose_loss_kernel)
    int n_partials;
    // update
    const float *d_rays_o, *d_rays_d;
    float* m;                   // [n_rot + 3] exp_avg      (rot, then trans)
    float* v;                   // [n_rot + 3] exp_avg_sq
    int* step;                  // [1] steps taken so far
    float* best_loss;           // [1]
    float* best_c2w;            // [12]
    float* last_loss;           // [1]
    double lr_rot, lr_trans, beta1, beta2, eps;
};
int mne_launch_pose(const PoseArgs& a, int what, hipStream_t st);

struct AdamArgs {
    mne_adam_seg_t seg[32];
    float step_size[32];      // lr / (1 - beta1^t)
    float bc2_sqrt[32];       // sqrt(1 - beta2^t)
    long long blk_start[33];  // prefix sum of blocks per segment
    int n_seg;
    int zero_grad;
    Clock clk;
};

int mne_launch_sample_z(const ZArgs& a, hipStream_t st);
int mne_launch_pack(const mne_scene_t& sc, float* pk, hipStream_t st);
// mode: 0 = forward, every sample decoded (raw complete);  1 = forward with early ray termination (maps only);
//       2 = training iteration (decode + backward);  3 = backward of an earlier forward call (raw_in given)
int mne_launch_render(const RenderArgs& a, int mode, void* workspace, const RenderHost& host, hipStream_t st);
int mne_launch_bin(RenderArgs a, int pass, void* workspace, hipStream_t st);
int mne_launch_hash_rows(const GridArgs& a, int bwd, hipStream_t st);
int mne_launch_hash_raygrad(const GridArgs& a, float* d_rays_o, float* d_rays_d, hipStream_t st);
int mne_launch_hash_slice_adam(const GridArgs& a, hipStream_t st, void* event_after_bin = nullptr);
int mne_hash_slice_count(const GridArgs& a);
unsigned mne_hash_scratch_entries(const GridArgs& a);
size_t mne_hash_layout(GridArgs& a, int R, int S, void* base);     // fills the workspace pointers (base NULL: sizes only); returns bytes
size_t mne_render_workspace(int R, int S);
int mne_launch_query(const QueryArgs& a, hipStream_t st);
int mne_launch_oneblob(long long n, int dims, const float* x, float* out, hipStream_t st);
int mne_launch_frequency(long long n, int dims, int F, const float* x, float* out, hipStream_t st);
int mne_launch_frequency_backward(long long n, int dims, int F, const float* x, const float* dout, float* dx, hipStream_t st);
int mne_launch_sh(long long n, int n_coef, const float* in, float* out, hipStream_t st);
int mne_launch_sh_backward(long long n, int n_coef, const float* in, const float* dout, float* din, hipStream_t st);
int mne_launch_identity(long long n_elems, float scale, float offset, const float* x, float* out, hipStream_t st);
int mne_launch_loss_finalize(const LossArgs& a, hipStream_t st);
int mne_launch_loss_coef(const LossArgs& a, hipStream_t st);
int mne_launch_wgrad(const mne_scene_t& sc, WgradArgs a, int impl, hipStream_t st);
int mne_wgrad_partial_count(const mne_scene_t& sc, int n_rays);
int mne_launch_decoder_update(const DecUpdateArgs& a, hipStream_t st);
int mne_launch_adam(const AdamArgs& a, hipStream_t st);
int mne_launch_grid(const GridArgs& a, int bwd, hipStream_t st);
int mne_launch_tile_order(const TileAdamArgs& a, hipStream_t st);
int mne_launch_tile_adam(const TileAdamArgs& a, hipStream_t st, const TileOverlap* ov = nullptr, int form = 0);
void mne_tile_geometry(const mne_scene_t& sc, TileBins& b);
int mne_launch_sample_rays(SampleRaysArgs a, unsigned long long seed, unsigned long long iteration, hipStream_t st);
int mne_launch_batch(SampleRaysArgs sr, unsigned long long seed, unsigned long long iteration, const ZArgs& a, const LossArgs& lc,
                     hipStream_t st);
size_t mne_dims_packed(const mne_scene_t& sc);
size_t mne_dims_tape_row(const mne_scene_t& sc);
size_t mne_dims_nparam(const mne_scene_t& sc);
size_t mne_dims_tape_dfeat(const mne_scene_t& sc);
size_t mne_dims_tape_pn(const mne_scene_t& sc);
int mne_wgrad_waves(void);
size_t mne_render_lds_bytes(const mne_scene_t& sc, int S, int bwd);
