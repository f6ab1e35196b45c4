/*
 * mneslam_hip.h -- C ABI of the MI355X (gfx950) mapping hot path of MNE-SLAM.
 *
 * Drop-in boundary: the reference has no FFI on this path -- its
 * JointEncoding / Mapper Python classes call ATen + tinycudann.  This library is what a
 * Python (ctypes) binding of those classes binds instead; every entry point lists the reference
 * code it replaces.  Plain pointers and sizes only: all device buffers are owned by the caller
 * (PyTorch tensors in the shipped host layer) and are borrowed for the duration of the call.
 * Nothing is cached between calls (planes may be re-bound wholesale, mp_slam/mapper.py:718-719).
 * Every kernel is launched on the caller's stream (`stream` = hipStream_t, may be NULL).
 *
 * Return value: 0 on success, negative on error; mne_last_error() gives the message
 * (thread-local).  No entry point aborts or throws.
 *
 * Physical layouts
 *   plane      : [H][W][C] fp32 (torch channels_last storage of the reference's [1,C,H,W] plane)
 *   rays, rgb  : [R][3] fp32        target_d, depth : [R] fp32        z_vals : [R][S] fp32
 *   raw        : [R][S][4] fp32 = (r, g, b raw logits, sdf)   (model/decoder.py:141,175)
 *   decoder    : nn.Linear layout [out][in] fp32, bias-free (model/decoder.py:51,104)
 */
#ifndef MNESLAM_HIP_H
#define MNESLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNE_ABI_VERSION 8

/* orientation index of a plane triple, as in JointEncoding.all_planes (scene_rep.py:117) */
enum { MNE_XY = 0, MNE_XZ = 1, MNE_YZ = 2 };
/* loss slots: order of the scalars JointEncoding.forward returns (scene_rep.py:597-609) */
enum { MNE_L_RGB = 0, MNE_L_DEPTH = 1, MNE_L_CO_SDF = 2, MNE_L_CO_FS = 3, MNE_L_E_FS = 4,
       MNE_L_E_CENTER = 5, MNE_L_E_TAIL = 6, MNE_L_PSNR = 7, MNE_N_LOSS = 8 };
/* mask-count slots produced by mne_sample_z */
enum { MNE_C_VALID = 0, MNE_C_E_FRONT = 1, MNE_C_E_CENTER = 2, MNE_C_E_TAIL = 3, MNE_C_CO_FS = 4,
       MNE_C_CO_SDF = 5,
       MNE_C_NEED = 6,   /* per-ray only: leading samples that can carry a loss term (z <= target depth + truncation) */
       MNE_C_TILE0 = 7,  /* per-ray only: exclusive prefix over the rays of their a-priori 32-sample tile counts (clamp(ceil(
                          * MNE_C_NEED / 32), 1, tiles per ray)): the render calls deal the decode's tile tasks evenly from it.
                          * Written for batches of at most 16384 rays (whole frames are summed by many workgroups and decoded with
                          * the fixed-stride schedule).  CONTRACT of every render call that takes `ray_counts` (mne_render_forward /
                          * _backward / _fused and their _features forms): the prefix is the one mne_sample_z / mne_sample_batch wrote
                          * for EXACTLY the call's n_rays rays (slot of ray 0 == 0).  A sub-range of a larger batch, or counts filled
                          * by the caller, fail that test on the device and are decoded with the fixed-stride schedule instead (same
                          * results, a few us slower) -- never out of bounds. */
       MNE_N_COUNT = 8 };

typedef struct mne_plane {
    const void* data;    /* [h][w][c_dim]: the parameter (what Adam updates, what the reference's tensor holds); fp32, or IEEE
                          * half precision when mne_scene_t.plane_f16 is set */
    float* grad;         /* [h][w][c_dim] fp32; accumulated into (atomics); NULL when not needed */
    int32_t h, w;
} mne_plane_t;

/* One JointEncoding's tensors (model/scene_rep.py:15-26, :85-181). */
typedef struct mne_scene {
    int32_t n_sets;        /* 1 = geometry planes only (grid.oneGrid), 2 = + colour planes */
    int32_t c_dim;         /* model.c_dim; this build supports 32 */
    int32_t hidden;        /* decoder.hidden_dim       (32 or 64), num_layers == 2 */
    int32_t hidden_color;  /* decoder.hidden_dim_color (32 or 64), num_layers_color == 2 */
    int32_t geo_feat_dim;  /* decoder.geo_feat_dim; this build supports 15 */
    int32_t n_bins;        /* pos.n_bins; this build supports 16 */
    int32_t bb_is_f64;     /* OneBlob input normalised in fp64 (bounding_box is float64, A4) */
    /* EXTENSION (BASELINE configs[4], "fp16 features + fp32 accumulate"; not reference behaviour): != 0 = every plane is
     * STORED in IEEE half precision ([h][w][c_dim] halves: 64-byte corner rows) -- there is no fp32 copy anywhere.  Lookups
     * convert to fp32 on load; interpolation, decoder, compositing, losses, gradients and their accumulation stay fp32;
     * the plane update (mne_tile_adam, mne_adam_step with p_f16) computes p32 = float(p16), applies Adam with fp32
     * moments and an fp32 gradient sum, and stores round-to-nearest-even(p32). */
    int32_t plane_f16;
    mne_plane_t plane[2][3][2];          /* [set][xy,xz,yz][coarse,fine] */
    float bound_lo[3], bound_hi[3];      /* EXTENDED bound (scene_rep.py:80-83), planes lookup */
    double bb_lo[3], bb_hi[3];           /* RAW bounding_box (scene_rep.py:292), OneBlob input */
    const float* w_sdf0;   /* [hidden][2*c_dim + 3*n_bins]          sdf_net.model.0.weight  */
    const float* w_sdf1;   /* [1+geo][hidden]                       sdf_net.model.2.weight  */
    const float* w_col0;   /* [hidden_color][3*n_bins (+2*c_dim) + geo]  color_net.model.0  */
    const float* w_col1;   /* [3][hidden_color]                     color_net.model.2       */
} mne_scene_t;

/* Sampling / compositing / loss constants (configs/Replica/replica.yaml:103-142,161). */
typedef struct mne_render_cfg {
    /* python floats of the YAML, kept in double so that derived constants such as
     * 0.4*truncation or sc_factor*trunc round to fp32 exactly as in the reference */
    double near_z, far_z;      /* cam.near, cam.far */
    double range_d;            /* training.range_d     */
    double perturb;            /* training.perturb     */
    double trunc;              /* training.trunc       (render weights, Co-SLAM losses) */
    double sc_factor;          /* data.sc_factor       */
    double truncation;         /* model.truncation     (ESLAM losses) */
    double depth_trunc;        /* cam.depth_trunc      */
    int32_t n_samples;         /* training.n_samples   (target_d == NULL path) */
    int32_t n_samples_d;       /* training.n_samples_d */
    int32_t n_range_d;         /* training.n_range_d   */
    int32_t reserved;
} mne_render_cfg_t;

/* Per-tile sample lists of the binned scatter (mne_render_fused -> mne_tile_adam).  All buffers are
 * caller-owned device memory; counts and *spill_count must be zero before the first use (each
 * mne_tile_adam call leaves counts zeroed again; mne_render_fused zeroes *spill_count itself). */
typedef struct mne_tile_bins {
    uint32_t* lists;       /* [mne_tile_list_entries()][8] = [mne_tile_count()][cap][8] with one capacity: tape row, packed local corner, 4 weights, tile id, 0 (32-byte entries) */
    int32_t* counts;       /* [mne_tile_count()][mne_tile_list_segments()]: ABI 7 -- every list is cut into mne_tile_list_segments() = 8 equal
                            * segments, one per XCD of the MI355X, each with its own cursor (the appending waves of an XCD use their own:
                            * returning atomics on one address retire one after the other).  A list's capacity is rounded down to a
                            * multiple of the segment count.  A SEGMENT that is full overflows to the spill area even while its
                            * list's other segments have room (entries appended by fewer than 8 XCDs -- a launch of fewer than 8
                            * workgroups -- use cap * k / 8 of a list): overflow stays exact (the spill area is walked by every
                            * overflowing tile: slower, not wrong), entries beyond a caller-chosen spill_cap are COUNTED in `dropped`
                            * (FusedStep.check() raises on it) -- size caps for the per-segment share, or keep the default worst-case
                            * spill area.  The segment of a wave comes from the hardware register XCC_ID (gfx94x / gfx950: this
                            * library is built for gfx950 only). */
    uint32_t* spill;       /* [spill_cap][8] overflow entries (same layout) */
    int32_t* spill_count;  /* [1] */
    int32_t* order;        /* [mne_tile_count()] scratch: tile processing order (heaviest first) */
    int32_t cap, spill_cap;
    int32_t* dropped;      /* [1] sticky: entries that fit neither their list nor the spill area (their gradient is
                            * LOST); never reset by the library.  A spill area of n_rays*n_samples*6*n_sets*4 entries
                            * cannot overflow.  Must be zero-initialised by the caller; check it after a run. */
    /* Optional (both or neither): long lists are processed by several workgroups whose partial gradient tiles are
     * combined in this scratch (load balance; results differ only in summation order).  With them, `order` must hold
     * mne_tile_count() + MNE_TILE_SPLIT_PARTS entries. */
    float* split_scratch;  /* [MNE_TILE_SPLIT_PARTS][16*16*c_dim] */
    int32_t* split_state;  /* [mne_tile_count() + 1], zero-initialised by the caller, left zeroed by mne_tile_adam */
    /* Optional: [mne_tile_count()], zero-initialised by the caller.  mne_tile_adam leaves every list's final length here
     * and mne_tile_order balances on max(length now, length of the previous call): it may then run BEFORE the lists are
     * complete (beside the deferred rays' second pass) without mis-judging scenes where many rays are deferred. */
    int32_t* prev_counts;
    /* Optional: [mne_tile_count()], 0 = "this tile's Adam moments are identically zero" (it has never received a gradient since
     * the optimizer state was created), anything else = live.  mne_tile_adam sets a tile's word when it first sees a non-empty list
     * and SKIPS tiles whose word is 0 and whose list is empty: Adam leaves them bit-for-bit unchanged (m = v = 0, no weight decay on
     * plane groups), so they are neither read nor written.  Zero-initialise only together with fresh (all-zero) moments; fill with
     * ones whenever the moments come from somewhere else (a loaded optimizer state).  NULL = every tile is swept. */
    int32_t* live;
    /* Optional per-plane list capacities in JointEncoding.all_planes order ([set][xy,xz,yz][coarse,fine]); an entry of 0
     * means `cap`.  One capacity for all tiles reserves tens of GB on scenes whose coarse planes have a few dozen tiles
     * taking 10^4-10^5 entries each next to thousands of fine tiles taking a few hundred (ScanNet with colour planes, INS
     * Indoor); per plane, 4x the mean list length costs 0.6-2.3 GB on every workload of BASELINE.json.  `lists` then holds
     * mne_tile_list_entries() entries: plane after plane, each plane's tiles back to back. */
    int32_t plane_cap[12];
} mne_tile_bins_t;
#define MNE_TILE_SPLIT_PARTS 2048
/* mne_tile_order snapshots the list lengths of up to this many tiles (it may then run while appends are still being added on
 * another stream); with more tiles call it after the last append. */
#define MNE_TILE_ORDER_SNAPSHOT 20480

/* Adam state and hyper-parameters of one plane, in JointEncoding.all_planes order
 * ([set][xy,xz,yz][coarse,fine]) for mne_tile_adam. */
typedef struct mne_plane_opt {
    float* m; float* v;                            /* exp_avg, exp_avg_sq: same layout as the plane */
    double lr, beta1, beta2, eps, weight_decay;
    int32_t step;                                  /* 1-based */
    int32_t reserved;
} mne_plane_opt_t;

/* Per-tensor view for the fused Adam step. */
typedef struct mne_adam_seg {
    void* p;               /* fp32, or half precision with p_f16 */
    float* g; float* m; float* v;
    int64_t n;
    double lr, beta1, beta2, eps, weight_decay;   /* python floats of the param group */
    int32_t step;          /* 1-based step number of this tensor's group */
    int32_t p_f16;         /* != 0: p is stored in IEEE half precision (g, m, v stay fp32); see mne_scene_t.plane_f16 */
} mne_adam_seg_t;

/* Device-resident iteration state, for callers that record one mapping iteration into a HIP graph and replay it:
 * the values that change from one iteration to the next are then read from device memory instead of kernel
 * arguments.  Entry points that take a `const mne_clock_t* clock` behave as before when it is NULL; otherwise
 *   iteration used = `iteration` argument + *clock->iteration     (ray sampling keys, jitter counter offset)
 *   Adam step used = `step` field        + *clock->step_offset    (bias corrections looked up in bias_table)
 * bias_table[t-1] = (1 - beta1^t, 1 - beta2^t) as doubles, computed by the caller for t = 1..n_table with the betas
 * of its param groups (all groups must share them); mne_clock_advance adds 1 to both counters (one tiny kernel, the
 * last node of a recorded iteration). */
typedef struct mne_clock {
    const uint64_t* iteration;     /* [1] device */
    const int32_t* step_offset;    /* [1] device */
    const double* bias_table;      /* [n_table][2] device */
    int32_t n_table;
    int32_t reserved;
    double beta1, beta2;           /* what bias_table was computed for (checked against the optimizer's) */
    uint64_t z_offset_stride;      /* mne_sample_z: counter offset used = offset + iteration used * z_offset_stride */
} mne_clock_t;

/* ---- library ------------------------------------------------------------------------- */
int mne_abi_version(void);
const char* mne_last_error(void);
size_t mne_sizeof_scene(void);
size_t mne_sizeof_render_cfg(void);
size_t mne_sizeof_adam_seg(void);
size_t mne_sizeof_tile_bins(void);
size_t mne_sizeof_plane_opt(void);
size_t mne_sizeof_clock(void);
size_t mne_sizeof_fused_opts(void);
int mne_clock_advance(uint64_t* iteration, int32_t* step_offset, void* stream);

/* Number of samples per ray: n_range_d + n_samples_d with depth guidance, n_samples without
 * (model/scene_rep.py:362-374). */
int mne_num_samples(const mne_render_cfg_t* cfg, int has_target_d);

/* ---- R1/R2: per-iteration ray batch --------------------------------------------------------- */
/* Replaces KeyFrameDatabase.sample_global_rays (model/keyframe.py:91-103) + the ray assembly of
 * Mapper.mapping_optimize (mp_slam/mapper.py:135-153) with device-resident data: n_global rows
 * sampled without replacement from kf_rays [n_kf_rays][7] (dir3,rgb3,depth1; owner keyframe =
 * index / n_save, its pose = poses[kf_pose_ids ? kf_pose_ids[owner] : owner]) plus n_cur pixels of
 * cur_rays [n_cur_rays][7] (pose = poses[n_poses-1], the reference's id -1), directions rotated
 * into the world frame.  idx_global / idx_cur (int64, device) supply explicit indices -- e.g. the
 * host RNG's draws, which makes the batch identical to the reference's; NULL = keyed Feistel
 * permutation of (seed, iteration) on the device.  Outputs are [R][3], [R][3], [R][3], [R] with
 * R = n_global + n_cur; out_idx (optional, int64 [R]) receives the indices used. */
int mne_sample_rays(const float* kf_rays, int64_t n_kf_rays, int n_save, const int32_t* kf_pose_ids,
                    const float* cur_rays, int64_t n_cur_rays, const float* poses, int n_poses,
                    int n_global, int n_cur, const int64_t* idx_global, const int64_t* idx_cur,
                    uint64_t seed, uint64_t iteration, float* rays_o, float* rays_d, float* target_rgb,
                    float* target_d, int64_t* out_idx, const mne_clock_t* clock, void* stream);

/* mne_sample_rays + mne_sample_z + mne_loss_coef for one training batch as ONE call of two launches (the per-iteration
 * batch preparation of the fused mapping step; separately they are four): same arguments and results as the three
 * calls (see below).  target_d of the z sampling is the depth of the rays just drawn; coef / grad_losses NULL = no
 * coefficients. */
int mne_sample_batch(const float* kf_rays, int64_t n_kf_rays, int n_save, const int32_t* kf_pose_ids,
                     const float* cur_rays, int64_t n_cur_rays, const float* poses, int n_poses,
                     int n_global, int n_cur, const int64_t* idx_global, const int64_t* idx_cur,
                     uint64_t seed, uint64_t iteration, float* rays_o, float* rays_d, float* target_rgb,
                     float* target_d, int64_t* out_idx, const struct mne_render_cfg* cfg, const float* u,
                     const float* lin_tables, uint64_t z_offset, float* z_vals, int32_t* counts, int32_t* ray_counts,
                     const float* grad_losses, float* coef, const mne_clock_t* clock, void* stream);

/* ---- R3: z sampling -------------------------------------------------------------------- */
/* Replaces render_rays' sampling block, model/scene_rep.py:362-381: near-surface linspace around
 * target_d (rays with d<=0 get linspace(near,far)), merged with the uniform samples, sorted, then
 * stratified jitter lower+(upper-lower)*U.  `u` [R][S] supplies U (the reference draws it with the
 * CPU generator, :381); NULL = counter-based Philox4x32-10 (seed, offset) on the device.
 * `target_d` NULL = linspace(near,far,n_samples).  `lin_tables` (device) holds the linspace
 * values themselves, computed by the host exactly as the reference computes them (torch.linspace
 * on the CPU, :363-373): with depth  [linspace(near,far,n_samples_d) | linspace(-range_d,range_d,
 * n_range_d) | linspace(near,far,n_range_d)], without  [linspace(near,far,n_samples)].  Also counts the loss masks that depend only on
 * z and target_d (scene_rep.py:489-499, :570; model/utils.py:131-145) into counts[MNE_N_COUNT]
 * (int32; overwritten by the call) via the scratch ray_counts [R][MNE_N_COUNT] (per-ray counts,
 * summed in a fixed order; both may be NULL when target_d is NULL).  `target_d` is [R]. */
int mne_sample_z(const mne_render_cfg_t* cfg, int n_rays, const float* target_d, const float* u,
                 const float* lin_tables, uint64_t seed, uint64_t offset, float* z_vals,
                 int32_t* counts, int32_t* ray_counts, const mne_clock_t* clock, void* stream);

/* ---- R8 helper: decoder weights in the kernels' packed form ------------------------------ */
size_t mne_packed_decoder_floats(const mne_scene_t* scene);
int mne_pack_decoder(const mne_scene_t* scene, float* packed, void* stream);

/* ---- R4-R9 (+R10 partial sums): forward -------------------------------------------------- */
/* Replaces JointEncoding.render_rays' body after sampling, model/scene_rep.py:384-386 ->
 * run_network (:303-317) -> query_color_sdf (:273-301: tri-plane bilinear lookup :28-53,
 * OneBlob model/encodings.py:61-71, decoder model/decoder.py:143-175) -> raw2outputs/sdf2weights
 * (:183-230).  Outputs (any may be NULL except raw): rgb [R][3], depth/disp/acc/depth_var [R],
 * raw [R][S][4].  When target_rgb/target_d are given, ray_sums [R][MNE_N_LOSS] receives each ray's
 * partial sums of the seven losses of JointEncoding.forward (:570-590).
 * flags & MNE_RENDER_EARLY_TERMINATION: exact early ray termination -- a ray's samples are decoded only up to the
 * last one that can influence its maps or losses (first SDF sign change + truncation window, loss masks); the maps
 * and ray_sums are the same, but `raw` is then scratch: entries behind a ray's last needed sample are undefined.
 * ray_counts (optional, with target_d) = the per-ray counts of mne_sample_z: lets the decode of the samples that are
 * needed whatever the decoder says run tile-parallel instead of on demand. */
#define MNE_RENDER_EARLY_TERMINATION 1
int mne_render_forward(const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples,
                       const float* rays_o, const float* rays_d, const float* target_rgb,
                       const float* target_d, const float* z_vals, const float* packed_decoder,
                       float* rgb, float* depth, float* disp, float* acc, float* depth_var,
                       float* raw, float* ray_sums, const int32_t* ray_counts, int flags, void* stream);

/* mne_render_forward for a scene encoding that is the caller's (NS-a: hash / dense grid): `features` [R*S][64] holds the
 * decoder's feature input of every sample (row = ray * n_samples + sample; e.g. filled by mne_hash_features); the plane
 * descriptors of `scene` are ignored.  Everything else as mne_render_forward. */
int mne_render_forward_features(const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples,
                                const float* rays_o, const float* rays_d, const float* target_rgb,
                                const float* target_d, const float* z_vals, const float* packed_decoder,
                                const float* features, float* rgb, float* depth, float* disp, float* acc, float* depth_var,
                                float* raw, float* ray_sums, const int32_t* ray_counts, int flags, void* stream);

/* ---- R10: loss scalars ------------------------------------------------------------------- */
/* losses[MNE_N_LOSS] = rgb, depth, co_sdf, co_fs, e_fs, e_center, e_tail, psnr with the reference's
 * normalisations (means over selections; empty selection -> NaN; Co-SLAM count weights). */
int mne_loss_finalize(int n_rays, int n_samples, const float* ray_sums, const int32_t* counts,
                      float* losses, void* stream);

/* d(total)/d(sample) coefficients from d(total)/d(loss_k) (`grad_losses`, device, 7 floats in
 * MNE_L_* order: the weights of MNESLAM.get_loss_from_ret, mneslam_mp.py:350-372, or whatever
 * autograd hands back) and the mask counts. coef is 8 floats on the device. */
int mne_loss_coef(const mne_render_cfg_t* cfg, int n_rays, int n_samples, const int32_t* counts,
                  const float* grad_losses, float* coef, void* stream);

/* ---- backward of R4-R10 (+R13) ----------------------------------------------------------- */
/* Floats per tape row (one row per SAMPLE: row = ray * n_samples + sample). */
size_t mne_tape_row_floats(const mne_scene_t* scene);
size_t mne_tape_dfeat_offset(const mne_scene_t* scene);    /* column of the d(feature) half of a row ([64 * n_sets] floats) */
/* Replaces loss.backward() through the graph built by JointEncoding.forward
 * (mp_slam/mapper.py:159): accumulates plane gradients into scene->plane[..].grad (atomic adds,
 * buffers must be zeroed by the caller / the fused Adam; ALL grad pointers NULL = no plane gradients
 * wanted, e.g. the pose-only loop of loop closure, mp_slam/mapper.py:388-408), fills the tape rows of every
 * sample that can receive gradient (and of the rest of their 32-sample tiles) for mne_decoder_wgrad, and optionally
 * writes d/d rays_o, d/d rays_d [R][3] (R13: pose optimisation in loop closure, mp_slam/mapper.py:388-408).
 * `coef` from mne_loss_coef (NULL = no loss terms); g_rgb [R][3] / g_depth [R] are optional extra upstream gradients
 * of the rendered maps (callers that build their own loss on render_rays outputs).  `raw` is the forward's output
 * for the same inputs (complete: produced without MNE_RENDER_EARLY_TERMINATION).  ray_counts: as for
 * mne_render_forward (NULL without target_d).  Outputs: *tape_rows (int32, device) = number of samples that received
 * gradient; ray_tiles [R] (int32, device) = per ray, the number of leading 32-sample tiles whose tape rows are
 * complete (input of mne_decoder_wgrad).  `workspace` is caller-owned device scratch of at least
 * mne_render_workspace_bytes(n_rays, n_samples) bytes (ReLU bit masks of the decoded samples). */
size_t mne_render_workspace_bytes(int n_rays, int n_samples);
int mne_render_backward(const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples,
                        const float* rays_o, const float* rays_d, const float* target_rgb,
                        const float* target_d, const float* z_vals, const int32_t* ray_counts,
                        const float* packed_decoder, const float* raw, const float* coef, const float* g_rgb,
                        const float* g_depth, float* tape, int64_t tape_capacity_rows, int32_t* tape_rows,
                        int32_t* ray_tiles, float* d_rays_o, float* d_rays_d, void* workspace, size_t workspace_bytes,
                        void* stream);

/* Optional per-call extras of the fused training call (NULL = none of them).  No field is kept by the library: the
 * call is stateless, whatever carries over from one iteration to the next lives in caller-owned device memory. */
typedef struct mne_fused_opts {
    /* Measurement: up to 6 hipEvent_t handles that THIS call records on its stream -- [0] before the feature gather, [1]
     * after it (scenes without colour planes gather inside the decode launch: nothing runs between [0] and [1]), [2] after the prefix decode, [3] after the ray kernel, [4] after the deferred pass, [5] after the list
     * appends (bin_kernel) -- so that bench.py can time the kernels of the call live (HIP events on the launch stream).
     * NULL entries are skipped. */
    void* const* timing_events;
    int32_t n_timing_events;
    int32_t lds_samples_cap;   /* 0 = default (256): samples of a ray the training kernel's first pass keeps in LDS */
    /* Adaptive a-priori prefix (exact in either mode, only the schedule changes): [4] int32 of caller-owned device
     * memory, zero-initialised once and handed to every call of one training run.  The exact early ray termination
     * decodes each ray's a-priori prefix tile-parallel and finishes the rays that turn out unresolved in a second
     * (deferred) pass; while the SDF is untrained most rays are unresolved and the second pass costs more than it saves.
     * word 0 = mode of THIS call (0: prefix + deferred pass, 1: decode every sample a priori), rewritten at the end of
     * the call from the fraction of rays the prefix did not (mode 0) / would not (mode 1) resolve: > 1/8 switches to mode
     * 1, < 1/16 back to mode 0.  word 1 = that count (scratch); words 2, 3 reserved.  NULL = always mode 0. */
    int32_t* adapt_state;
    /* The list appends of the binned plane update only need the decode's outputs, not the backward: with external_bin != 0
     * mne_render_fused leaves those of the rays its a-priori prefix resolves to the caller, who runs mne_tile_bin(pass 0)
     * on a SECOND stream once event_after_decode (a hipEvent_t this call records behind its prefix decode) has fired --
     * before mne_tile_adam.  The deferred rays' appends stay in this call (behind its deferred pass).  0 / NULL: the call
     * does all appends itself. */
    int32_t external_bin;
    /* mne_render_fused_features with grid_cfg + table: != 0 = the rows of the first pass (a-priori tiles + the resolver's
     * extension) were already gathered by the caller -- mne_hash_gather with the same ray_counts, e.g. on the caller's stream
     * while the decoder of the previous iteration is still being updated on another; the call then gathers only the
     * deferred rays' remaining rows. */
    int32_t features_pregathered;
    void* event_after_decode;
} mne_fused_opts_t;

/* mne_render_backward for a caller-owned encoding: the feature columns of every tape row ([64] floats at column 0 of row
 * ray * n_samples + sample) are filled by the caller before the call (mne_hash_gather); no plane gradients -- the d(feature)
 * rows of every sample of the first ray_tiles[r] tiles of ray r are left in the tape (column mne_tape_dfeat_offset) for the
 * caller's scatter (mne_hash_scatter); tape rows for mne_decoder_wgrad as usual.  d_rays_o / d_rays_d (optional, [R][3]):
 * the share of the ray gradients that goes through the OneBlob input; the caller ADDS its encoding's share
 * (mne_hash_ray_grad) -- together R13 for the hash / dense grid model. */
int mne_render_backward_features(const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples,
                                 const float* rays_o, const float* rays_d, const float* target_rgb,
                                 const float* target_d, const float* z_vals, const int32_t* ray_counts,
                                 const float* packed_decoder, const float* raw, const float* coef, const float* g_rgb,
                                 const float* g_depth, float* tape, int64_t tape_capacity_rows, int32_t* tape_rows,
                                 int32_t* ray_tiles, float* d_rays_o, float* d_rays_d, void* workspace, size_t workspace_bytes,
                                 void* stream);

/* Fused training form of the two calls above (what Mapper.mapping_optimize runs per iteration), with early ray
 * termination: decodes every ray up to the last sample it needs (tile-parallel for the samples ray_counts marks,
 * on demand for the rest), composites (rgb, depth, ray_sums like mne_render_forward) and back-propagates with the
 * loss coefficients `coef` (mne_loss_coef) from the ReLU masks saved by the decode -- no forward recompute.
 * `raw` is scratch (see MNE_RENDER_EARLY_TERMINATION).  ray_counts NULL = decode everything up front. */
int mne_render_fused(const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples,
                     const float* rays_o, const float* rays_d, const float* target_rgb,
                     const float* target_d, const float* z_vals, const int32_t* ray_counts,
                     const float* packed_decoder, const float* coef, float* rgb, float* depth, float* raw,
                     float* ray_sums, float* tape, int64_t tape_capacity_rows, int32_t* tape_rows, int32_t* ray_tiles,
                     const mne_tile_bins_t* bins, void* workspace, size_t workspace_bytes, const mne_fused_opts_t* opts,
                     void* stream);

/* The list appends of the binned plane update as a call of its own (see mne_fused_opts_t::external_bin): every sample of
 * the batch that receives gradient -- derived from the decode's outputs with the rules of the training kernel: render
 * window behind the first SDF sign change, active loss masks -- is appended to the lists of the plane tiles its bilinear
 * footprints touch.  pass 0: the rays whose a-priori prefix (+ the resolver's extension) resolves them; pass 1: the rays of
 * the deferred list, after mne_render_fused returned.  Arguments as given to that mne_render_fused call (same workspace:
 * it holds the decoded-tile counts and the deferred list); opts->adapt_state must be the same pointer. */
int mne_tile_bin(const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples, const float* rays_o,
                 const float* rays_d, const float* target_d, const float* z_vals, const int32_t* ray_counts,
                 const float* coef, const float* raw, const mne_tile_bins_t* bins, void* workspace, size_t workspace_bytes,
                 int pass, const mne_fused_opts_t* opts, void* stream);

/* Binned scatter + Adam for the planes (see csrc/tile_adam.hip): with `bins` given, mne_render_fused
 * does not touch plane[].grad; it appends every contributing sample to the lists of the 16x16-cell
 * plane tiles it touches, and mne_tile_adam then (per tile) sums the samples' d(feature) x bilinear
 * weights in LDS and applies torch.optim.Adam's update to the tile -- i.e. it replaces
 * grid_sampler_2d_backward + Adam.step() + zero_grad() for the plane groups
 * (mneslam_mp.py:459-469) with no gradient buffer and no global atomics.  opt[] has 6*n_sets entries.
 * Entries reference tape rows (ray * n_samples + sample). */
size_t mne_tile_count(const mne_scene_t* scene);
int mne_tile_list_segments(void);     /* cursors per tile list (8) */
/* Entries `bins->lists` must hold for this scene with bins->cap / bins->plane_cap (0 on invalid arguments). */
size_t mne_tile_list_entries(const mne_scene_t* scene, const mne_tile_bins_t* bins);
/* Processing order of the tiles for the next mne_tile_adam call (bins->order): longest lists first, so the
 * few very long lists do not form the tail of the launch.  Call after mne_render_fused, before mne_tile_adam. */
int mne_tile_order(const mne_scene_t* scene, const mne_tile_bins_t* bins, void* stream);
int mne_tile_adam(const mne_scene_t* scene, const mne_plane_opt_t* opt, const float* tape,
                  const mne_tile_bins_t* bins, const mne_clock_t* clock, void* stream);

/* EXTENSION (multi-agent; the reference's agents exchange maps through files only, mp_slam/mapper.py:491-509 just tests
 * which overlap box a point is in): plane gradients of the cells two agents both map, on the binned path -- no dense
 * gradient tensor on either side.  The agents' planes sit on one lattice; the cells shared with peer k are, per plane, a
 * rectangle of nodes [x0,x1) x [y0,y1) in THIS agent's indices (x1 <= x0: the plane shares nothing).  A buffer holds the
 * rectangles of all planes back to back in all_planes order, each as [y1-y0][x1-x0][c_dim] floats
 * (mne_tile_overlap_floats() in total).
 *   mne_tile_order -> mne_tile_grad_export (send[k] filled) -> exchange with the peers (one message each way per peer)
 *   -> mne_tile_adam_shared (= mne_tile_adam where a shared cell's gradient is send[k] + recv[k]: the agents add the same
 *      two numbers, so cells that start equal stay bit-equal on both).
 * ABI 8: three slots -- an agent of a chain of slabs has two neighbours, and the planes that do NOT contain the slab axis (yz for
 * slabs along x) are held as a whole by EVERY agent: with more than two agents their gradient must be the sum over ALL agents
 * (pairwise sums would give the middle agent g0 + g1 + g2 and its neighbours g0 + g1 / g1 + g2: the copies drift apart).  The caller
 * gives such planes a slot of their own (rectangle = the whole plane), all-reduces send[k] into recv[k] and zeroes send[k]
 * (mneslam_amd/dist.py::allreduce_sum_into): every agent then applies the same total. */
#define MNE_MAX_OVERLAP_PEERS 3
typedef struct mne_tile_overlap {
    int32_t n_peers, reserved;
    struct { int32_t x0, y0, x1, y1; } rect[MNE_MAX_OVERLAP_PEERS][12];
    float* send[MNE_MAX_OVERLAP_PEERS];
    const float* recv[MNE_MAX_OVERLAP_PEERS];
} mne_tile_overlap_t;
size_t mne_sizeof_tile_overlap(void);
size_t mne_tile_overlap_floats(const mne_scene_t* scene, const mne_tile_overlap_t* overlap, int peer);
int mne_tile_grad_export(const mne_scene_t* scene, const float* tape, const mne_tile_bins_t* bins,
                         const mne_tile_overlap_t* overlap, void* stream);
int mne_tile_adam_shared(const mne_scene_t* scene, const mne_plane_opt_t* opt, const float* tape,
                         const mne_tile_bins_t* bins, const mne_tile_overlap_t* overlap, const mne_clock_t* clock, void* stream);

/* N2, pose alignment of loop closure (mp_slam/mapper.py:362-412) without an autograd graph or a torch.optim step:
 *   mne_pose_rays    rot / trans -> c2w, rays_o / rays_d   (SLAM.matrix_from_tensor for rot_rep 'axis_angle' or 'quat',
 *                    optimization/utils.py:161-210, + the ray rotation of mp_slam/mapper.py:388-392)
 *   mne_sample_z, mne_render_forward(target_d = NULL), then
 *   mne_pose_loss    w_rgb * mse(rgb, want_rgb) + w_depth * mse(depth, want_depth) (mapper.py:394-396): gradients of the
 *                    maps for mne_render_backward(g_rgb, g_depth, d_rays_o, d_rays_d) + loss partial sums
 *   mne_pose_update  ray gradients -> d/d(rot, trans) (analytic Jacobian of the axis-angle map), best pose so far
 *                    (mapper.py:399-403), one torch.optim.Adam step on the six parameters (groups lr_rot / lr_trans,
 *                    mneslam_mp.py:577-584); the step count lives in device memory.
 * R = Rot(rot) * r_base: r_base = identity is the reference's parameterisation. */
typedef struct mne_pose_state {
    float* rot;            /* [n_rot]: axis-angle, or quaternion (real part first) */
    float* trans;          /* [3] */
    float* m;              /* [n_rot + 3] exp_avg (rot, trans), zero-initialised by the caller */
    float* v;              /* [n_rot + 3] exp_avg_sq */
    int32_t* step;         /* [1] Adam steps taken so far */
    float* c2w;            /* [12] row-major 3x4 of the current parameters (written by mne_pose_rays) */
    float* best_loss;      /* [1] initialise to +inf */
    float* best_c2w;       /* [12] pose at which best_loss was seen */
    float* last_loss;      /* [1] */
    float r_base[9];       /* row-major */
    int32_t n_rot;         /* 3: axis-angle (rot_rep 'axis_angle'), 4: quaternion (rot_rep 'quat',
                            * pytorch3d.transforms.quaternion_to_matrix: normalising, real part first) */
    double lr_rot, lr_trans, beta1, beta2, eps;
} mne_pose_state_t;
size_t mne_sizeof_pose_state(void);
int mne_pose_rays(const mne_pose_state_t* pose, int n_rays, const float* dirs_cam, float* rays_o, float* rays_d, void* stream);
int mne_pose_loss(int n_rays, const float* rgb, const float* depth, const float* want_rgb, const float* want_depth,
                  double w_rgb, double w_depth, float* d_rgb, float* d_depth, float* partials /* [(n_rays + 255) / 256] */,
                  void* stream);
int mne_pose_update(const mne_pose_state_t* pose, int n_rays, const float* dirs_cam, const float* d_rays_o,
                    const float* d_rays_d, const float* partials, void* stream);

/* Decoder weight gradients from the tape: dW = sum_rows outer(d_out, in) for the four matrices,
 * written as [w_col0 | w_col1 | w_sdf0 | w_sdf1] (the order of decoder.parameters(),
 * model/decoder.py:150-159) into grad_out.  Rows = for every ray r, the first ray_tiles[r] * 32 samples (as left by
 * mne_render_backward / mne_render_fused); summed ray by ray in a fixed order.  `partials` is scratch of
 * mne_wgrad_partial_floats() floats.  impl: 0 = MFMA (v_mfma_f32_32x32x2_f32), one fused pass over the tape,
 * 1 = scalar check, 2 = MFMA with one launch per matrix, 3 = the pass of 0 without the final reduction (grad_out is not
 * written: mne_decoder_update sums the partials and applies Adam in the same launch). */
size_t mne_decoder_param_floats(const mne_scene_t* scene);
size_t mne_wgrad_partial_floats(const mne_scene_t* scene);
int mne_decoder_wgrad(const mne_scene_t* scene, const float* tape, const int32_t* ray_tiles, int n_rays, int n_samples,
                      float* partials, float* grad_out, int impl, void* stream);

/* The decoder's optimizer step in ONE launch: the fixed-order sum of the partial weight gradients left by
 * mne_decoder_wgrad(impl = 3) -> grad_out, torch.optim.Adam on the four decoder tensors (in place; opt->m / opt->v in
 * decoder.parameters() order: w_col0, w_col1, w_sdf0, w_sdf1; one param group) and the loss scalars of the iteration
 * (mne_loss_finalize; losses NULL = skip).  Follow with mne_pack_decoder for the next render. */
typedef struct mne_decoder_opt {
    float* m[4]; float* v[4];
    double lr, beta1, beta2, eps, weight_decay;
    int32_t step;          /* 1-based */
    int32_t reserved;
} mne_decoder_opt_t;
size_t mne_sizeof_decoder_opt(void);
int mne_decoder_update(const mne_scene_t* scene, const float* partials, int n_rays, float* grad_out,
                       const mne_decoder_opt_t* opt, int n_samples, const float* ray_sums,
                       const int32_t* counts, float* losses, const mne_clock_t* clock, void* stream);

/* ---- R12: fused dense Adam --------------------------------------------------------------- */
/* Replaces torch.optim.Adam.step() + zero_grad() over the groups of MNESLAM.create_optimizer
 * (mneslam_mp.py:459-469): one pass, m = lerp(m,g,1-b1); v = b2 v + (1-b2) g^2;
 * p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps); weight_decay is L2 into g; g is zeroed
 * when zero_grad != 0.  segs is a HOST array of n_seg (<= 32) entries. */
int mne_adam_step(const mne_adam_seg_t* segs, int n_seg, int zero_grad, const mne_clock_t* clock, void* stream);

/* ---- point queries (forward only) -------------------------------------------------------- */
/* Replaces JointEncoding.query_color_sdf / query_sdf / run_network_flat (scene_rep.py:232-331):
 * raw [N][4] for arbitrary points [N][3]; geo (optional) [N][geo_feat_dim]; feat (optional)
 * [N][2*c_dim] = sample_plane_feature of the geometry planes (scene_rep.py:28-53).
 * corner_idx (optional) [N][3*n_sets][2][2] int32 receives the integer NW corner (ix0, iy0) of the
 * bilinear footprint of every point in every plane, in [set*3 + orientation][level] order -- the
 * "bit-exact integer indices" of the tri-plane lookup (ATen grid_sampler_2d, align_corners=True,
 * padding_mode='border'; scene_rep.py:43-47).  They are the values the gather itself uses. */
#define MNE_QUERY_PTS_NORMALISED 1   /* pts are already in [-1,1] plane coordinates (feat / corner_idx only) */
int mne_query_points(const mne_scene_t* scene, int64_t n_pts, const float* pts,
                     const float* packed_decoder, float* raw, float* geo, float* feat,
                     int32_t* corner_idx, int flags, void* stream);

/* mne_query_points for a caller-owned encoding: `features` [N][64] = the decoder's feature input of every point (e.g.
 * mne_grid_encode_box with out_stride 64 into a zero-initialised buffer). */
int mne_query_features(const mne_scene_t* scene, int64_t n_pts, const float* pts, const float* features,
                       const float* packed_decoder, float* raw, float* geo, void* stream);

/* ---- R7: OneBlob encoding as a stand-alone op -------------------------------------------- */
/* Replaces tcnn.Encoding(otype="OneBlob", n_bins=16) (model/encodings.py:61-71):
 * x [N][dims] in [0,1] -> out [N][dims*16], layout [dim0 bins | dim1 bins | ...]. */
int mne_encode_oneblob(int64_t n_pts, int dims, const float* x, float* out, void* stream);

/* ---- the remaining branches of get_encoder (model/encodings.py:48-58, 73-95; ABI 8) -----------
 * tcnn.Encoding(otype = "Frequency" | "SphericalHarmonics" | "Identity"): never reached by the reference's mapping path
 * (model/scene_rep.py:157 requests OneBlob); forward + gradient with respect to the input.  Spec and parity status:
 * oracle/encodings_misc.py (tinycudann is not in the reference tree: parity unpinned).
 *   frequency: x [N][dims] -> out [N][dims * 2 * n_frequencies], out[dim * 2F + 2f + s] = sin(2^f * pi * x[dim] + s * pi / 2)
 *   sh       : in [N][3] in [0,1] (direction = 2 in - 1) -> out [N][degree^2], degree 1..4
 *   identity : out = x * scale + offset */
int mne_encode_frequency(int64_t n_pts, int dims, int n_frequencies, const float* x, float* out, void* stream);
int mne_encode_frequency_backward(int64_t n_pts, int dims, int n_frequencies, const float* x, const float* d_out, float* d_x, void* stream);
int mne_encode_sh(int64_t n_pts, int degree, const float* in, float* out, void* stream);
int mne_encode_sh_backward(int64_t n_pts, int degree, const float* in, const float* d_out, float* d_in, void* stream);
int mne_encode_identity(int64_t n_elems, float scale, float offset, const float* x, float* out, void* stream);

/* ---- R14: multiresolution hash / dense grid encoding (tinycudann replacement surface) ------ */
/* Replaces tcnn.Encoding(otype="HashGrid"/"Grid") as configured by get_encoder (model/encodings.py:
 * 13-46; not executed by the reference's mapping path).  grid_type 0 = Hash, 1 = Dense.  Spec and
 * parity status: oracle/hashgrid.py (tinycudann is not in the reference tree: parity unpinned). */
typedef struct mne_grid_cfg {
    int32_t n_levels, n_features, base_resolution, log2_hashmap_size, grid_type, reserved;
    double per_level_scale;
} mne_grid_cfg_t;
/* per-level constants (host arrays of n_levels entries each; any may be NULL) */
int mne_grid_level_table(const mne_grid_cfg_t* cfg, float* scale, uint32_t* resolution, uint32_t* size, uint32_t* offset);
/* number of fp32 parameters of the encoding (all levels) */
size_t mne_grid_param_count(const mne_grid_cfg_t* cfg);
/* x [N][3] in [0,1] -> out [N][n_levels*n_features]; idx (optional) [N][n_levels][8] uint32 table
 * indices within each level (the "integer hash indices" of this encoding) */
int mne_grid_encode(const mne_grid_cfg_t* cfg, int64_t n_pts, const float* x, const float* params,
                    float* out, uint32_t* idx, void* stream);
/* the same for WORLD points: x = (p - bb_lo) / (bb_hi - bb_lo) with scene's raw bounding box, exactly the OneBlob input
 * of the render kernels; out rows are out_stride floats apart (>= n_levels*n_features; the rest of a row is untouched) */
int mne_grid_encode_box(const mne_grid_cfg_t* cfg, const mne_scene_t* scene, int64_t n_pts, const float* pts,
                        const float* params, float* out, int out_stride, void* stream);
/* d(params) += scatter of dout [N][n_levels*n_features] (dparams pre-zeroed by the caller) */
int mne_grid_encode_backward(const mne_grid_cfg_t* cfg, int64_t n_pts, const float* x, const float* dout,
                             float* dparams, void* stream);

/* ---- NS-a: the mapping iteration with a multiresolution hash grid as the scene encoding (Co-SLAM wiring, the call the
 * reference keeps commented out at model/scene_rep.py:160,243,250; parity unpinned like the encoding itself) ---------
 * The decoder, compositing, losses and their backward are the tri-plane path's kernels: the feature half of a tape row
 * ([64] floats at column 0 of row ray*n_samples + sample) is filled by mne_hash_gather -- the grid's n_levels*2 features,
 * the remaining columns stay zero -- instead of by the plane gather, and the 64-wide first decoder layer simply has dead
 * input columns.  One iteration:
 *   mne_hash_gather            x = (o + d*z - bb_lo) / (bb_hi - bb_lo) per sample (the OneBlob input), 8 corners x 16 levels
 *   mne_render_fused_features  = mne_render_fused without plane gather / scatter: every decode reads its features from the
 *                                tape (ray_counts as in mne_render_fused: exact early termination).  With grid_cfg + table
 *                                the call does the gather itself and only where rows can be decoded: the a-priori tiles of
 *                                every ray first (round 6: no resolver extension on caller-supplied features), the remaining rows of the deferred rays
 *                                before the second pass (then mne_hash_gather is not needed); NULL, NULL: the caller has
 *                                filled every row; the d(feature) rows of every sample of the first
 *                                ray_tiles[r] tiles of ray r are left in the tape (column mne_tape_dfeat_offset)
 *   mne_hash_scatter           grad_table += w * d(feature)  (global_atomic_add_f32; grad_table zeroed by the caller)
 *   mne_decoder_wgrad, mne_adam_step (table + decoder segments, zero_grad fused)
 *   -- or, instead of mne_hash_scatter + the table's Adam segment --
 *   mne_hash_slice_adam        the table update without float atomics and without a gradient buffer: the iteration's
 *                              backward rows are packed level-major into `workspace` and sorted by the 1024-entry slices of
 *                              the table their corners fall into; one workgroup per slice sums the slice's gradient in LDS
 *                              as 64-bit fixed-point numbers (exact, order-independent: the update is bit-reproducible) and
 *                              applies Adam (opt: moments with the table's layout, step 1-based) to the slice.  Same
 *                              arithmetic as a grid_sampler-style scatter + torch.optim.Adam up to the rounding of each
 *                              addend to 2^-39 of the level's largest gradient component.
 * scene: bounding box, decoder dims and weights are read; the plane descriptors are ignored.  cfg: n_features 2, <= 16 levels. */
size_t mne_hash_workspace_bytes(const mne_grid_cfg_t* cfg, int n_rays, int n_samples);   /* zero-fill the workspace before the FIRST call */
/* event_after_bin (optional hipEvent_t): recorded on `stream` behind the binning launch, in front of the slice launch -- a caller
 * that runs the decoder's weight-gradient chain on another stream starts it there: beside the HBM-bound slice / Adam launch
 * instead of beside the latency-bound binning. */
int mne_hash_slice_adam(const mne_grid_cfg_t* cfg, const mne_scene_t* scene, int n_rays, int n_samples, const float* rays_o,
                        const float* rays_d, const float* z_vals, const float* tape, const int32_t* ray_tiles, float* table,
                        const mne_plane_opt_t* opt, void* workspace, size_t workspace_bytes, void* event_after_bin, void* stream);
/* ray_counts (optional, the per-ray counts of mne_sample_z): only the rows the exact early termination can decode in its first
 * pass -- the a-priori tiles of every ray (what the first decode pass of mne_render_fused_features reads); NULL = every row */
int mne_hash_gather(const mne_grid_cfg_t* cfg, const mne_scene_t* scene, int n_rays, int n_samples, const float* rays_o,
                    const float* rays_d, const float* z_vals, const int32_t* ray_counts, const float* table, float* tape, void* stream);
/* the grid features of every sample as compact rows: features [R*S][64], columns [0, n_levels*2) written, the rest
 * untouched (zero-initialise once): the input of mne_render_forward_features */
int mne_hash_features(const mne_grid_cfg_t* cfg, const mne_scene_t* scene, int n_rays, int n_samples, const float* rays_o,
                      const float* rays_d, const float* z_vals, const float* table, float* features, void* stream);
int mne_render_fused_features(const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples,
                              const float* rays_o, const float* rays_d, const float* target_rgb, const float* target_d,
                              const float* z_vals, const int32_t* ray_counts, const float* packed_decoder, const float* coef,
                              float* rgb, float* depth, float* raw, float* ray_sums, float* tape, int64_t tape_capacity_rows,
                              int32_t* tape_rows, int32_t* ray_tiles, void* workspace, size_t workspace_bytes,
                              const mne_grid_cfg_t* grid_cfg, const float* table, const mne_fused_opts_t* opts, void* stream);
int mne_hash_scatter(const mne_grid_cfg_t* cfg, const mne_scene_t* scene, int n_rays, int n_samples, const float* rays_o,
                     const float* rays_d, const float* z_vals, const float* tape, const int32_t* ray_tiles,
                     float* grad_table, void* stream);
/* R13 for the grid: d(total)/d(rays_o), d(total)/d(rays_d) through the trilinear weights of every level, from the d(feature)
 * rows mne_render_backward_features left in the tape; ADDED to d_rays_o / d_rays_d [R][3] (which hold the OneBlob share). */
int mne_hash_ray_grad(const mne_grid_cfg_t* cfg, const mne_scene_t* scene, int n_rays, int n_samples, const float* rays_o,
                      const float* rays_d, const float* z_vals, const float* table, const float* tape, const int32_t* ray_tiles,
                      float* d_rays_o, float* d_rays_d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MNESLAM_HIP_H */
