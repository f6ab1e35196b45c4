// capi.hip -- extern "C" entry points of libmneslam_hip.so (declared in include/mneslam_hip.h).
// Argument validation, derivation of the fp32 constants from the YAML doubles, kernel launches on
// the caller's stream.  No device allocation, no synchronisation, no state kept between calls.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "mne_launch.h"

long long mne_adam_blocks_for(long long n);
int mne_launch_clock_advance(unsigned long long* iteration, int* step_offset, hipStream_t st);

static int fill_clock(const mne_clock_t* c, Clock& out, double beta1, double beta2, bool need_table) {
    out = Clock{};
    if (!c) return 0;
    if (need_table) {
        if (!c->step_offset || !c->bias_table || c->n_table < 1) return -1;
        if (c->beta1 != beta1 || c->beta2 != beta2) return -2;
        out.step_offset = c->step_offset; out.bias_table = c->bias_table; out.n_table = c->n_table;
    } else {
        if (!c->iteration) return -1;
        out.iteration = (const unsigned long long*)c->iteration;
        out.z_offset_stride = c->z_offset_stride;
    }
    return 0;
}

static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-10, std::string(what) + ": " + hipGetErrorString(e));
    return 0;
}

static int check_scene(const mne_scene_t* sc, bool need_grad, bool need_planes = true) {
    if (!sc) return fail(-1, "scene is NULL");
    if (sc->n_sets != 1 && sc->n_sets != 2) return fail(-1, "n_sets must be 1 (oneGrid) or 2 (with colour planes)");
    if (sc->c_dim != 32) return fail(-1, "model.c_dim must be 32 in this build");
    if (sc->n_bins != 16) return fail(-1, "pos.n_bins must be 16 in this build");
    if (sc->geo_feat_dim != 15) return fail(-1, "decoder.geo_feat_dim must be 15 in this build");
    if (!((sc->hidden == 32 && sc->hidden_color == 32) || (sc->hidden == 64 && sc->hidden_color == 64)))
        return fail(-2, "decoder hidden_dim/hidden_dim_color must both be 32 or both 64 in this build");
    for (int s = 0; need_planes && s < sc->n_sets; ++s)
        for (int o = 0; o < 3; ++o)
            for (int l = 0; l < 2; ++l) {
                const mne_plane_t& p = sc->plane[s][o][l];
                if (!p.data || p.h < 1 || p.w < 1) return fail(-1, "plane pointer/shape missing");
                if (need_grad && !p.grad) return fail(-1, "plane gradient buffer missing");
                if ((long long)p.h * p.w * sc->c_dim >= (1ll << 31)) return fail(-1, "plane too large for 32-bit offsets");
            }
    if (!sc->w_sdf0 || !sc->w_sdf1 || !sc->w_col0 || !sc->w_col1) return fail(-1, "decoder weight pointer missing");
    return 0;
}

static void fill_render_consts(RenderArgs& a, const mne_render_cfg_t* cfg) {
    a.trunc_f = (float)cfg->trunc;
    a.win_f = (float)(cfg->sc_factor * cfg->trunc);       // python: sc_factor * trunc, then fp32
    a.e_T = (float)cfg->truncation;
    a.e_T04 = (float)(0.4 * cfg->truncation);
    a.depth_trunc = (float)cfg->depth_trunc;
}


extern "C" {

int mne_abi_version(void) { return MNE_ABI_VERSION; }
const char* mne_last_error(void) { return g_err.c_str(); }
size_t mne_sizeof_scene(void) { return sizeof(mne_scene_t); }
size_t mne_sizeof_render_cfg(void) { return sizeof(mne_render_cfg_t); }
size_t mne_sizeof_adam_seg(void) { return sizeof(mne_adam_seg_t); }
size_t mne_sizeof_tile_bins(void) { return sizeof(mne_tile_bins_t); }
size_t mne_sizeof_plane_opt(void) { return sizeof(mne_plane_opt_t); }
size_t mne_sizeof_clock(void) { return sizeof(mne_clock_t); }
size_t mne_sizeof_fused_opts(void) { return sizeof(mne_fused_opts_t); }
size_t mne_sizeof_decoder_opt(void) { return sizeof(mne_decoder_opt_t); }

int mne_clock_advance(uint64_t* iteration, int32_t* step_offset, void* stream) {
    if (!iteration && !step_offset) return fail(-1, "mne_clock_advance: NULL argument");
    mne_launch_clock_advance((unsigned long long*)iteration, step_offset, (hipStream_t)stream);
    return check_launch("clock_advance");
}

int mne_num_samples(const mne_render_cfg_t* cfg, int has_target_d) {
    if (!cfg) return fail(-1, "cfg is NULL");
    return has_target_d ? cfg->n_range_d + cfg->n_samples_d : cfg->n_samples;
}

int mne_sample_z(const mne_render_cfg_t* cfg, int n_rays, const float* target_d, const float* u,
                 const float* lin_tables, uint64_t seed, uint64_t offset, float* z_vals, int32_t* counts,
                 int32_t* ray_counts, const mne_clock_t* clock, void* stream) {
    if (!cfg || !z_vals || !lin_tables) return fail(-1, "mne_sample_z: NULL argument");
    if (n_rays <= 0) return 0;
    ZArgs a;
    a.R = n_rays;
    a.has_d = target_d != nullptr;
    a.n_a = a.has_d ? cfg->n_samples_d : 0;
    a.n_b = a.has_d ? cfg->n_range_d : 0;
    a.S = a.has_d ? a.n_a + a.n_b : cfg->n_samples;
    if (a.S < 1 || a.S > 16384) return fail(-1, "mne_sample_z: samples per ray out of range [1,16384]");
    a.perturb = (float)cfg->perturb;
    a.e_T = (float)cfg->truncation;
    a.e_T04 = (float)(0.4 * cfg->truncation);
    a.co_T = (float)(cfg->trunc * cfg->sc_factor);
    a.depth_trunc = (float)cfg->depth_trunc;
    a.target_d = target_d;
    a.u = u;
    a.tables = lin_tables;
    a.seed = seed;
    a.offset = offset;
    a.z_vals = z_vals;
    a.counts = counts;
    a.ray_counts = ray_counts;
    if (fill_clock(clock, a.clk, 0, 0, false)) return fail(-1, "mne_sample_z: incomplete clock");
    hipStream_t st = (hipStream_t)stream;
    if (a.has_d && (!counts || !ray_counts)) return fail(-1, "mne_sample_z: counts and ray_counts buffers required when target_d is given");
    mne_launch_sample_z(a, st);
    return check_launch("sample_z");
}

size_t mne_packed_decoder_floats(const mne_scene_t* scene) { return scene ? mne_dims_packed(*scene) : 0; }
size_t mne_tape_row_floats(const mne_scene_t* scene) { return scene ? mne_dims_tape_row(*scene) : 0; }
size_t mne_tape_dfeat_offset(const mne_scene_t* scene) { return scene ? mne_dims_tape_dfeat(*scene) : 0; }
size_t mne_decoder_param_floats(const mne_scene_t* scene) { return scene ? mne_dims_nparam(*scene) : 0; }
size_t mne_wgrad_partial_floats(const mne_scene_t* scene) {
    return scene ? mne_dims_nparam(*scene) * (size_t)mne_wgrad_waves() : 0;
}

int mne_pack_decoder(const mne_scene_t* scene, float* packed, void* stream) {
    if (int rc = check_scene(scene, false, false)) return rc;
    if (!packed) return fail(-1, "mne_pack_decoder: packed is NULL");
    if (int rc = mne_launch_pack(*scene, packed, (hipStream_t)stream)) return fail(rc, "unsupported decoder shape");
    return check_launch("pack_decoder");
}

static int render_forward(const float* features, const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples,
                       const float* rays_o, const float* rays_d, const float* target_rgb,
                       const float* target_d, const float* z_vals, const float* packed_decoder,
                       float* rgb, float* depth, float* disp, float* acc, float* depth_var,
                       float* raw, float* ray_sums, const int32_t* ray_counts, int flags, void* stream) {
    if (int rc = check_scene(scene, false, features == nullptr)) return rc;
    if (features && scene->n_sets != 1) return fail(-2, "caller-supplied features replace ONE plane set (no colour planes)");
    if (!cfg || !rays_o || !rays_d || !z_vals || !packed_decoder || !raw) return fail(-1, "mne_render_forward: NULL argument");
    if (n_rays <= 0) return 0;
    if (n_samples < 1 || n_samples > 16384) return fail(-1, "samples per ray out of range");
    if (mne_render_lds_bytes(*scene, n_samples, 0) > 160 * 1024) return fail(-1, "samples per ray too large for the LDS staging");
    if (ray_sums && (!target_rgb || !target_d)) return fail(-1, "ray_sums needs target_rgb and target_d");
    if (ray_counts && !target_d) return fail(-1, "ray_counts belongs to a depth-guided batch (target_d)");
    RenderArgs a = {};
    a.sc = *scene;
    a.R = n_rays; a.S = n_samples;
    fill_render_consts(a, cfg);
    a.rays_o = rays_o; a.rays_d = rays_d; a.target_rgb = target_rgb; a.target_d = target_d;
    a.z_vals = z_vals; a.packed = packed_decoder;
    a.rgb = rgb; a.depth = depth; a.disp = disp; a.acc = acc; a.depth_var = depth_var; a.raw = raw;
    a.ray_sums = ray_sums;
    const bool early = (flags & MNE_RENDER_EARLY_TERMINATION) != 0;
    a.ray_counts = early ? ray_counts : nullptr;
    a.prefix_default = early ? 1 : (1 << 30);
    if (features) { a.ext_feat = 1; a.ext_rows = features; a.ext_stride = 64; }
    if (const char* c = std::getenv("MNE_FRAME_MIN_TILES")) a.frame_min_tiles = std::atoi(c);      // tests run the pipelined frame decode on small batches
    if (int rc = mne_launch_render(a, early ? 1 : 0, nullptr, RenderHost{}, (hipStream_t)stream)) return fail(rc, "unsupported scene configuration");
    return check_launch("render_forward");
}

int mne_render_forward(const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples,
                       const float* rays_o, const float* rays_d, const float* target_rgb,
                       const float* target_d, const float* z_vals, const float* packed_decoder,
                       float* rgb, float* depth, float* disp, float* acc, float* depth_var,
                       float* raw, float* ray_sums, const int32_t* ray_counts, int flags, void* stream) {
    return render_forward(nullptr, scene, cfg, n_rays, n_samples, rays_o, rays_d, target_rgb, target_d, z_vals, packed_decoder, rgb,
                          depth, disp, acc, depth_var, raw, ray_sums, ray_counts, flags, stream);
}

int mne_render_forward_features(const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples,
                                const float* rays_o, const float* rays_d, const float* target_rgb,
                                const float* target_d, const float* z_vals, const float* packed_decoder,
                                const float* features, float* rgb, float* depth, float* disp, float* acc, float* depth_var,
                                float* raw, float* ray_sums, const int32_t* ray_counts, int flags, void* stream) {
    if (!features) return fail(-1, "mne_render_forward_features: features is NULL");
    return render_forward(features, scene, cfg, n_rays, n_samples, rays_o, rays_d, target_rgb, target_d, z_vals, packed_decoder, rgb,
                          depth, disp, acc, depth_var, raw, ray_sums, ray_counts, flags, stream);
}

int mne_loss_finalize(int n_rays, int n_samples, const float* ray_sums, const int32_t* counts, float* losses,
                      void* stream) {
    if (!ray_sums || !counts || !losses) return fail(-1, "mne_loss_finalize: NULL argument");
    LossArgs a = {};
    a.R = n_rays; a.S = n_samples; a.ray_sums = ray_sums; a.counts = counts; a.losses = losses;
    mne_launch_loss_finalize(a, (hipStream_t)stream);
    return check_launch("loss_finalize");
}

int mne_loss_coef(const mne_render_cfg_t* cfg, int n_rays, int n_samples, const int32_t* counts,
                  const float* grad_losses, float* coef, void* stream) {
    if (!cfg || !counts || !grad_losses || !coef) return fail(-1, "mne_loss_coef: NULL argument");
    LossArgs a = {};
    a.R = n_rays; a.S = n_samples; a.counts = counts; a.grad_losses = grad_losses; a.coef = coef;
    a.e_T = (float)cfg->truncation;
    a.co_T = (float)(cfg->trunc * cfg->sc_factor);
    mne_launch_loss_coef(a, (hipStream_t)stream);
    return check_launch("loss_coef");
}

size_t mne_render_workspace_bytes(int n_rays, int n_samples) {
    if (n_rays <= 0 || n_samples <= 0) return 0;
    return mne_render_workspace(n_rays, n_samples);
}

// per-plane capacities and list offsets (tile_base / ntx must be filled); returns the total number of entries
static size_t list_layout(const mne_scene_t& sc, const mne_tile_bins_t* bins, TileBins& out) {
    long long off = 0;
    for (int p = 0; p < MNE_MAX_PLANES; ++p) {
        int cap = (p < sc.n_sets * 6 && bins->plane_cap[p] > 0) ? bins->plane_cap[p] : bins->cap;
        cap = cap < MNE_LIST_SEGMENTS ? MNE_LIST_SEGMENTS : cap - cap % MNE_LIST_SEGMENTS;      // MNE_LIST_SEGMENTS equal segments per list
        out.pcap[p] = cap;
        out.list_off[p] = off - (long long)out.tile_base[p] * cap;
        off += (long long)(out.tile_base[p + 1] - out.tile_base[p]) * cap;
    }
    return (size_t)off;
}

static int fill_bins(const mne_scene_t* scene, const mne_tile_bins_t* bins, TileBins& out) {
    if (!bins->lists || !bins->counts || !bins->spill || !bins->spill_count || !bins->order || !bins->dropped || bins->cap < 1 ||
        bins->spill_cap < 1)
        return fail(-1, "incomplete tile bins");
    out.lists = bins->lists; out.counts = bins->counts; out.spill = bins->spill;
    out.spill_count = bins->spill_count; out.order = bins->order; out.cap = bins->cap; out.spill_cap = bins->spill_cap;
    out.dropped = bins->dropped;
    if ((bins->split_scratch != nullptr) != (bins->split_state != nullptr)) return fail(-1, "tile bins: split_scratch and split_state go together");
    out.split_scratch = bins->split_scratch; out.split_state = bins->split_state;
    mne_tile_geometry(*scene, out);
    list_layout(*scene, bins, out);
    return 0;
}

static int render_backward(bool ext_feat, const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples,
                        const float* rays_o, const float* rays_d, const float* target_rgb,
                        const float* target_d, const float* z_vals, const int32_t* ray_counts,
                        const float* packed_decoder, const float* raw, const float* coef, const float* g_rgb,
                        const float* g_depth, float* tape, int64_t tape_capacity_rows, int32_t* tape_rows,
                        int32_t* ray_tiles, float* d_rays_o, float* d_rays_d, void* workspace, size_t workspace_bytes,
                        void* stream) {
    // plane gradients are optional as a whole: all plane[].grad NULL = ray / decoder gradients only
    if (int rc = check_scene(scene, !ext_feat && scene && scene->plane[0][0][0].grad != nullptr, !ext_feat)) return rc;
    if (ext_feat && scene->n_sets != 1) return fail(-2, "caller-supplied features replace ONE plane set (no colour planes)");
    if (!cfg || !rays_o || !rays_d || !z_vals || !packed_decoder || !raw || !tape || !tape_rows || !ray_tiles || !workspace)
        return fail(-1, "mne_render_backward: NULL argument");
    if (n_rays <= 0) return 0;
    if (workspace_bytes < mne_render_workspace(n_rays, n_samples)) return fail(-1, "mne_render_backward: workspace too small");
    if (n_samples < 1 || mne_render_lds_bytes(*scene, n_samples, 1) > 160 * 1024) return fail(-1, "samples per ray out of range");
    if (tape_capacity_rows < (int64_t)n_rays * n_samples) return fail(-1, "tape must hold n_rays*n_samples rows");
    if (coef && (!target_rgb || !target_d)) return fail(-1, "loss coefficients need target_rgb and target_d");
    if (ray_counts && !target_d) return fail(-1, "ray_counts belongs to a depth-guided batch (target_d)");
    RenderArgs a = {};
    a.sc = *scene;
    a.R = n_rays; a.S = n_samples;
    fill_render_consts(a, cfg);
    a.rays_o = rays_o; a.rays_d = rays_d; a.target_rgb = target_rgb; a.target_d = target_d;
    a.z_vals = z_vals; a.packed = packed_decoder; a.raw_in = raw;
    a.ray_counts = ray_counts;
    a.prefix_default = ray_counts ? 0 : 1;          // without depth guidance the forward half of the tape is made on demand
    a.coef = coef; a.g_rgb = g_rgb; a.g_depth = g_depth;
    a.tape = tape; a.tape_rows = tape_rows; a.ray_tiles = ray_tiles;
    a.d_rays_o = d_rays_o; a.d_rays_d = d_rays_d;
    a.ext_feat = ext_feat ? 1 : 0;
    if (int rc = mne_launch_render(a, 3, workspace, RenderHost{}, (hipStream_t)stream)) return fail(rc, "unsupported scene configuration");
    return check_launch("render_backward");
}

extern "C" {
int mne_render_backward(const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples,
                        const float* rays_o, const float* rays_d, const float* target_rgb,
                        const float* target_d, const float* z_vals, const int32_t* ray_counts,
                        const float* packed_decoder, const float* raw, const float* coef, const float* g_rgb,
                        const float* g_depth, float* tape, int64_t tape_capacity_rows, int32_t* tape_rows,
                        int32_t* ray_tiles, float* d_rays_o, float* d_rays_d, void* workspace, size_t workspace_bytes,
                        void* stream) {
    return render_backward(false, scene, cfg, n_rays, n_samples, rays_o, rays_d, target_rgb, target_d, z_vals, ray_counts, packed_decoder,
                           raw, coef, g_rgb, g_depth, tape, tape_capacity_rows, tape_rows, ray_tiles, d_rays_o, d_rays_d, workspace,
                           workspace_bytes, stream);
}

int mne_render_backward_features(const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples,
                                 const float* rays_o, const float* rays_d, const float* target_rgb,
                                 const float* target_d, const float* z_vals, const int32_t* ray_counts,
                                 const float* packed_decoder, const float* raw, const float* coef, const float* g_rgb,
                                 const float* g_depth, float* tape, int64_t tape_capacity_rows, int32_t* tape_rows,
                                 int32_t* ray_tiles, float* d_rays_o, float* d_rays_d, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    return render_backward(true, scene, cfg, n_rays, n_samples, rays_o, rays_d, target_rgb, target_d, z_vals, ray_counts, packed_decoder,
                           raw, coef, g_rgb, g_depth, tape, tape_capacity_rows, tape_rows, ray_tiles, d_rays_o, d_rays_d, workspace,
                           workspace_bytes, stream);
}
}   // extern "C"


}   // extern "C"

static int fill_hash_rows(const mne_grid_cfg_t* cfg, const mne_scene_t* scene, int n_rays, int n_samples, const float* rays_o,
                          const float* rays_d, const float* z_vals, float* tape, GridArgs& a);

static int render_fused(bool ext_feat, const GridArgs* ext_grid, const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples,
                     const float* rays_o, const float* rays_d, const float* target_rgb,
                     const float* target_d, const float* z_vals, const int32_t* ray_counts,
                     const float* packed_decoder, const float* coef, float* rgb, float* depth, float* raw,
                     float* ray_sums, float* tape, int64_t tape_capacity_rows, int32_t* tape_rows, int32_t* ray_tiles,
                     const mne_tile_bins_t* bins, void* workspace, size_t workspace_bytes, const mne_fused_opts_t* opts, void* stream) {
    if (int rc = check_scene(scene, bins == nullptr && !ext_feat, !ext_feat)) return rc;
    if (!cfg || !rays_o || !rays_d || !target_rgb || !target_d || !z_vals || !packed_decoder || !coef || !raw ||
        !tape || !tape_rows || !ray_tiles || !workspace)
        return fail(-1, "mne_render_fused: NULL argument");
    if (n_rays <= 0) return 0;
    if (workspace_bytes < mne_render_workspace(n_rays, n_samples)) return fail(-1, "mne_render_fused: workspace too small");
    if (n_samples < 1 || mne_render_lds_bytes(*scene, n_samples, 1) > 160 * 1024) return fail(-1, "samples per ray out of range");
    if (tape_capacity_rows < (int64_t)n_rays * n_samples) return fail(-1, "tape must hold n_rays*n_samples rows");
    RenderArgs a = {};
    a.sc = *scene;
    a.R = n_rays; a.S = n_samples;
    fill_render_consts(a, cfg);
    a.rays_o = rays_o; a.rays_d = rays_d; a.target_rgb = target_rgb; a.target_d = target_d;
    a.z_vals = z_vals; a.packed = packed_decoder; a.coef = coef;
    a.ray_counts = ray_counts;
    a.prefix_default = 1 << 30;                     // no counts: decode every sample up front
    a.rgb = rgb; a.depth = depth; a.raw = raw; a.ray_sums = ray_sums;
    a.tape = tape; a.tape_rows = tape_rows; a.ray_tiles = ray_tiles;
    a.ext_feat = ext_feat ? 1 : 0;
    RenderHost host;
    host.ext_grid = ext_grid;
    if (const char* c = std::getenv("MNE_HOT_LDS_SAMPLES")) a.lds_samples = std::atoi(c);      // tests force the two-pass split on small S
    if (const char* c = std::getenv("MNE_HEAVY_NTILE")) a.heavy_ntile = std::atoi(c);          // ... and the heavy-ray list (< 0: off)
    if (const char* c = std::getenv("MNE_HEAVY_TILES")) a.heavy_min = std::atoi(c);
    if (opts) {
        if (opts->n_timing_events < 0 || opts->n_timing_events > 6 || (opts->n_timing_events > 0 && !opts->timing_events))
            return fail(-1, "mne_fused_opts: timing_events holds 0..6 event handles");
        host.marks = opts->timing_events; host.n_marks = opts->n_timing_events;
        if (opts->lds_samples_cap > 0) a.lds_samples = opts->lds_samples_cap;
        a.adapt = ray_counts ? opts->adapt_state : nullptr;       // without per-ray counts everything is decoded a priori anyway
        host.external_bin = opts->external_bin; host.ev_after_decode = opts->event_after_decode;
        host.features_pregathered = opts->features_pregathered;
    }
    if (bins)
        if (int rc = fill_bins(scene, bins, a.bins)) return rc;
    if (int rc = mne_launch_render(a, 2, workspace, host, (hipStream_t)stream)) return fail(rc, "unsupported scene configuration");
    return check_launch("render_fused");
}

extern "C" {

int mne_render_fused(const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples,
                     const float* rays_o, const float* rays_d, const float* target_rgb,
                     const float* target_d, const float* z_vals, const int32_t* ray_counts,
                     const float* packed_decoder, const float* coef, float* rgb, float* depth, float* raw,
                     float* ray_sums, float* tape, int64_t tape_capacity_rows, int32_t* tape_rows, int32_t* ray_tiles,
                     const mne_tile_bins_t* bins, void* workspace, size_t workspace_bytes, const mne_fused_opts_t* opts, void* stream) {
    return render_fused(false, nullptr, scene, cfg, n_rays, n_samples, rays_o, rays_d, target_rgb, target_d, z_vals, ray_counts,
                        packed_decoder, coef, rgb, depth, raw, ray_sums, tape, tape_capacity_rows, tape_rows, ray_tiles, bins,
                        workspace, workspace_bytes, opts, stream);
}

int mne_render_fused_features(const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples,
                              const float* rays_o, const float* rays_d, const float* target_rgb, const float* target_d,
                              const float* z_vals, const int32_t* ray_counts, const float* packed_decoder, const float* coef,
                              float* rgb, float* depth, float* raw, float* ray_sums, float* tape, int64_t tape_capacity_rows,
                              int32_t* tape_rows, int32_t* ray_tiles, void* workspace, size_t workspace_bytes,
                              const mne_grid_cfg_t* grid_cfg, const float* table, const mne_fused_opts_t* opts, void* stream) {
    GridArgs g = {};
    if (grid_cfg || table) {                          // the call gathers the hash-grid rows itself, only where they can be decoded
        if (!grid_cfg || !table) return fail(-1, "mne_render_fused_features: grid_cfg and table go together");
        if (int rc = fill_hash_rows(grid_cfg, scene, n_rays, n_samples, rays_o, rays_d, z_vals, tape, g)) return rc;
        g.params = table;
    }
    return render_fused(true, grid_cfg ? &g : nullptr, scene, cfg, n_rays, n_samples, rays_o, rays_d, target_rgb, target_d, z_vals,
                        ray_counts, packed_decoder, coef, rgb, depth, raw, ray_sums, tape, tape_capacity_rows, tape_rows, ray_tiles,
                        nullptr, workspace, workspace_bytes, opts, stream);
}

int mne_tile_bin(const mne_scene_t* scene, const mne_render_cfg_t* cfg, int n_rays, int n_samples, const float* rays_o,
                 const float* rays_d, const float* target_d, const float* z_vals, const int32_t* ray_counts,
                 const float* coef, const float* raw, const mne_tile_bins_t* bins, void* workspace, size_t workspace_bytes,
                 int pass, const mne_fused_opts_t* opts, void* stream) {
    if (int rc = check_scene(scene, false)) return rc;
    if (!cfg || !rays_o || !rays_d || !target_d || !z_vals || !coef || !raw || !bins || !workspace) return fail(-1, "mne_tile_bin: NULL argument");
    if (pass != 0 && pass != 1) return fail(-1, "mne_tile_bin: pass is 0 or 1");
    if (n_rays <= 0) return 0;
    if (workspace_bytes < mne_render_workspace(n_rays, n_samples)) return fail(-1, "mne_tile_bin: workspace too small");
    RenderArgs a = {};
    a.sc = *scene;
    a.R = n_rays; a.S = n_samples;
    fill_render_consts(a, cfg);
    a.rays_o = rays_o; a.rays_d = rays_d; a.target_d = target_d; a.z_vals = z_vals; a.coef = coef; a.raw = (float*)raw;
    a.ray_counts = ray_counts;
    a.prefix_default = 1 << 30;
    a.adapt = (opts && ray_counts) ? opts->adapt_state : nullptr;
    if (int rc = fill_bins(scene, bins, a.bins)) return rc;
    mne_launch_bin(a, pass, workspace, (hipStream_t)stream);
    return check_launch("tile_bin");
}

int mne_tile_list_segments(void) { return MNE_LIST_SEGMENTS; }

size_t mne_tile_count(const mne_scene_t* scene) {
    if (!scene || (scene->n_sets != 1 && scene->n_sets != 2)) return 0;
    TileBins b = {};
    mne_tile_geometry(*scene, b);
    return (size_t)b.tile_base[scene->n_sets * 6];
}

size_t mne_tile_list_entries(const mne_scene_t* scene, const mne_tile_bins_t* bins) {
    if (!scene || !bins || bins->cap < 1 || (scene->n_sets != 1 && scene->n_sets != 2)) return 0;
    for (int p = 0; p < 12; ++p) if (bins->plane_cap[p] < 0) return 0;
    TileBins b = {};
    mne_tile_geometry(*scene, b);
    return list_layout(*scene, bins, b);
}

int mne_tile_order(const mne_scene_t* scene, const mne_tile_bins_t* bins, void* stream) {
    if (int rc = check_scene(scene, false)) return rc;
    if (!bins || !bins->counts || !bins->order) return fail(-1, "mne_tile_order: NULL argument");
    TileAdamArgs a = {};
    a.sc = *scene;
    a.n_planes = scene->n_sets * 6;
    a.bins.counts = bins->counts; a.bins.order = bins->order; a.bins.cap = bins->cap;
    a.bins.split_scratch = bins->split_scratch; a.bins.split_state = bins->split_state; a.prev_counts = bins->prev_counts;
    const char* sm = std::getenv("MNE_TILE_SPLIT_MIN");                   // tests force splitting on tiny scenes
    const int split_min = sm ? std::atoi(sm) : MNE_TILE_SPLIT_MIN_DEFAULT;
    a.bins.split_min = split_min < 1 ? 1 : split_min;
    mne_tile_geometry(*scene, a.bins);
    list_layout(*scene, bins, a.bins);
    mne_launch_tile_order(a, (hipStream_t)stream);
    return check_launch("tile_order");
}

// overlap descriptor -> kernel form (rectangles clipped to nothing are refused: the caller's geometry is wrong then)
static int fill_overlap(const mne_scene_t* scene, const mne_tile_overlap_t* ov, TileOverlap& out, bool need_send, bool need_recv) {
    if (!ov || ov->n_peers < 1 || ov->n_peers > MNE_MAX_OVERLAP_PEERS) return fail(-1, "tile overlap: 1..MNE_MAX_OVERLAP_PEERS peers");
    out.n_peers = ov->n_peers;
    for (int k = 0; k < ov->n_peers; ++k) {
        long long off = 0;
        for (int p = 0; p < MNE_MAX_PLANES; ++p) {
            OverlapRect& r = out.rect[k][p];
            r = OverlapRect{0, 0, 0, 0, 0};
            if (p >= scene->n_sets * 6) continue;
            const mne_plane_t& pl = scene->plane[p / 6][(p % 6) / 2][p % 2];
            const auto& q = ov->rect[k][p];
            if (q.x1 <= q.x0 || q.y1 <= q.y0) continue;
            if (q.x0 < 0 || q.y0 < 0 || q.x1 > pl.w || q.y1 > pl.h) return fail(-1, "tile overlap: rectangle outside its plane");
            r = OverlapRect{q.x0, q.y0, q.x1, q.y1, off};
            off += (long long)(q.x1 - q.x0) * (q.y1 - q.y0) * scene->c_dim;
        }
        if ((need_send && !ov->send[k]) || (need_recv && !ov->recv[k])) return fail(-1, "tile overlap: NULL exchange buffer");
        out.send[k] = ov->send[k]; out.recv[k] = ov->recv[k];
    }
    return 0;
}

size_t mne_sizeof_tile_overlap(void) { return sizeof(mne_tile_overlap_t); }

size_t mne_tile_overlap_floats(const mne_scene_t* scene, const mne_tile_overlap_t* ov, int peer) {
    if (!scene || !ov || peer < 0 || peer >= ov->n_peers || peer >= MNE_MAX_OVERLAP_PEERS) return 0;
    size_t n = 0;
    for (int p = 0; p < scene->n_sets * 6 && p < 12; ++p) {
        const auto& q = ov->rect[peer][p];
        if (q.x1 > q.x0 && q.y1 > q.y0) n += (size_t)(q.x1 - q.x0) * (q.y1 - q.y0) * scene->c_dim;
    }
    return n;
}

static int tile_adam_impl(const mne_scene_t* scene, const mne_plane_opt_t* opt, const float* tape,
                          const mne_tile_bins_t* bins, const mne_tile_overlap_t* overlap, int form, const mne_clock_t* clock, void* stream) {
    if (int rc = check_scene(scene, false)) return rc;
    if ((!opt && form != 1) || !tape || !bins) return fail(-1, "mne_tile_adam: NULL argument");
    TileAdamArgs a = {};
    a.sc = *scene;
    a.n_planes = scene->n_sets * 6;
    if (int rc = fill_bins(scene, bins, a.bins)) return rc;
    a.prev_counts = bins->prev_counts;
    a.live = bins->live;
    TileOverlap ov = {};
    if (form != 0)
        if (int rc = fill_overlap(scene, overlap, ov, true, form == 2)) return rc;
    for (int k = 0; form != 1 && k < a.n_planes; ++k) {
        const mne_plane_opt_t& g = opt[k];
        if (!g.m || !g.v || g.step < 1) return fail(-1, "mne_tile_adam: bad plane optimizer state");
        PlaneOpt& o = a.opt[k];
        o.m = g.m; o.v = g.v;
        o.omb1 = (float)(1.0 - g.beta1); o.b2 = (float)g.beta2; o.omb2 = (float)(1.0 - g.beta2);
        o.eps = (float)g.eps; o.wd = (float)g.weight_decay;
        o.step_size = (float)(g.lr / (1.0 - std::pow(g.beta1, (double)g.step)));
        o.bc2_sqrt = (float)std::sqrt(1.0 - std::pow(g.beta2, (double)g.step));
        o.lr = g.lr; o.step = g.step;
        if (clock && fill_clock(clock, a.clk, g.beta1, g.beta2, true))
            return fail(-1, "mne_tile_adam: clock incomplete or made for other betas");
    }
    a.tape = tape;
    a.n_tiles = a.bins.tile_base[a.n_planes];
    a.row_stride = (int)mne_dims_tape_row(*scene);
    a.t_dfeat = (int)mne_dims_tape_dfeat(*scene);
    a.t_pn = (int)mne_dims_tape_pn(*scene);
    mne_launch_tile_adam(a, (hipStream_t)stream, form ? &ov : nullptr, form);
    return check_launch("tile_adam");
}

int mne_tile_adam(const mne_scene_t* scene, const mne_plane_opt_t* opt, const float* tape,
                  const mne_tile_bins_t* bins, const mne_clock_t* clock, void* stream) {
    return tile_adam_impl(scene, opt, tape, bins, nullptr, 0, clock, stream);
}

int mne_tile_grad_export(const mne_scene_t* scene, const float* tape, const mne_tile_bins_t* bins,
                         const mne_tile_overlap_t* overlap, void* stream) {
    return tile_adam_impl(scene, nullptr, tape, bins, overlap, 1, nullptr, stream);
}

int mne_tile_adam_shared(const mne_scene_t* scene, const mne_plane_opt_t* opt, const float* tape,
                         const mne_tile_bins_t* bins, const mne_tile_overlap_t* overlap, const mne_clock_t* clock, void* stream) {
    return tile_adam_impl(scene, opt, tape, bins, overlap, 2, clock, stream);
}

int mne_sample_rays(const float* kf_rays, int64_t n_kf_rays, int n_save, const int32_t* kf_pose_ids,
                    const float* cur_rays, int64_t n_cur_rays, const float* poses, int n_poses,
                    int n_global, int n_cur, const int64_t* idx_global, const int64_t* idx_cur,
                    uint64_t seed, uint64_t iteration, float* rays_o, float* rays_d, float* target_rgb,
                    float* target_d, int64_t* out_idx, const mne_clock_t* clock, void* stream) {
    if (!poses || !rays_o || !rays_d || !target_rgb || !target_d || n_poses < 1) return fail(-1, "mne_sample_rays: NULL argument");
    if (n_global < 0 || n_cur < 0) return fail(-1, "mne_sample_rays: negative count");
    if (n_global > 0 && (!kf_rays || n_save < 1 || n_kf_rays < n_global)) return fail(-1, "mne_sample_rays: cannot draw n_global distinct keyframe rays");
    if (n_cur > 0 && (!cur_rays || n_cur_rays < n_cur)) return fail(-1, "mne_sample_rays: cannot draw n_cur distinct current-frame rays");
    if (n_global + n_cur == 0) return 0;
    SampleRaysArgs a = {};
    a.kf_rays = kf_rays; a.n_kf_rays = n_kf_rays; a.n_save = n_save; a.kf_pose_ids = kf_pose_ids;
    a.cur_rays = cur_rays; a.n_cur_rays = n_cur_rays; a.poses = poses; a.n_poses = n_poses;
    a.n_global = n_global; a.n_cur = n_cur;
    a.idx_global = (const long long*)idx_global; a.idx_cur = (const long long*)idx_cur; a.out_idx = (long long*)out_idx;
    a.rays_o = rays_o; a.rays_d = rays_d; a.target_rgb = target_rgb; a.target_d = target_d;
    if (fill_clock(clock, a.clk, 0, 0, false)) return fail(-1, "mne_sample_rays: incomplete clock");
    mne_launch_sample_rays(a, seed, iteration, (hipStream_t)stream);
    return check_launch("sample_rays");
}

static void fill_zargs(ZArgs& a, const mne_render_cfg_t* cfg, int n_rays, bool has_d) {
    a.R = n_rays;
    a.has_d = has_d;
    a.n_a = has_d ? cfg->n_samples_d : 0;
    a.n_b = has_d ? cfg->n_range_d : 0;
    a.S = has_d ? a.n_a + a.n_b : cfg->n_samples;
    a.perturb = (float)cfg->perturb;
    a.e_T = (float)cfg->truncation;
    a.e_T04 = (float)(0.4 * cfg->truncation);
    a.co_T = (float)(cfg->trunc * cfg->sc_factor);
    a.depth_trunc = (float)cfg->depth_trunc;
}

int mne_sample_batch(const float* kf_rays, int64_t n_kf_rays, int n_save, const int32_t* kf_pose_ids,
                     const float* cur_rays, int64_t n_cur_rays, const float* poses, int n_poses,
                     int n_global, int n_cur, const int64_t* idx_global, const int64_t* idx_cur,
                     uint64_t seed, uint64_t iteration, float* rays_o, float* rays_d, float* target_rgb,
                     float* target_d, int64_t* out_idx, const mne_render_cfg_t* cfg, const float* u,
                     const float* lin_tables, uint64_t z_offset, float* z_vals, int32_t* counts, int32_t* ray_counts,
                     const float* grad_losses, float* coef, const mne_clock_t* clock, void* stream) {
    if (!poses || !rays_o || !rays_d || !target_rgb || !target_d || n_poses < 1 || !cfg || !lin_tables || !z_vals || !counts ||
        !ray_counts)
        return fail(-1, "mne_sample_batch: NULL argument");
    if ((coef == nullptr) != (grad_losses == nullptr)) return fail(-1, "mne_sample_batch: coef and grad_losses go together");
    if (n_global < 0 || n_cur < 0) return fail(-1, "mne_sample_batch: negative count");
    if (n_global > 0 && (!kf_rays || n_save < 1 || n_kf_rays < n_global)) return fail(-1, "mne_sample_batch: cannot draw n_global distinct keyframe rays");
    if (n_cur > 0 && (!cur_rays || n_cur_rays < n_cur)) return fail(-1, "mne_sample_batch: cannot draw n_cur distinct current-frame rays");
    const int R = n_global + n_cur;
    if (R == 0) return 0;
    SampleRaysArgs sr = {};
    sr.kf_rays = kf_rays; sr.n_kf_rays = n_kf_rays; sr.n_save = n_save; sr.kf_pose_ids = kf_pose_ids;
    sr.cur_rays = cur_rays; sr.n_cur_rays = n_cur_rays; sr.poses = poses; sr.n_poses = n_poses;
    sr.n_global = n_global; sr.n_cur = n_cur;
    sr.idx_global = (const long long*)idx_global; sr.idx_cur = (const long long*)idx_cur; sr.out_idx = (long long*)out_idx;
    sr.rays_o = rays_o; sr.rays_d = rays_d; sr.target_rgb = target_rgb; sr.target_d = target_d;
    if (fill_clock(clock, sr.clk, 0, 0, false)) return fail(-1, "mne_sample_batch: incomplete clock");
    ZArgs a = {};
    fill_zargs(a, cfg, R, true);
    if (a.S < 1 || a.S > 16384) return fail(-1, "mne_sample_batch: samples per ray out of range [1,16384]");
    a.target_d = target_d; a.u = u; a.tables = lin_tables; a.seed = seed; a.offset = z_offset;
    a.z_vals = z_vals; a.counts = counts; a.ray_counts = ray_counts;
    a.clk = sr.clk;
    LossArgs lc = {};
    lc.R = R; lc.S = a.S; lc.counts = counts; lc.grad_losses = grad_losses; lc.coef = coef;
    lc.e_T = (float)cfg->truncation;
    lc.co_T = (float)(cfg->trunc * cfg->sc_factor);
    mne_launch_batch(sr, seed, iteration, a, lc, (hipStream_t)stream);
    return check_launch("sample_batch");
}

int mne_decoder_update(const mne_scene_t* scene, const float* partials, int n_rays, float* grad_out,
                       const mne_decoder_opt_t* opt, int n_samples, const float* ray_sums,
                       const int32_t* counts, float* losses, const mne_clock_t* clock, void* stream) {
    if (int rc = check_scene(scene, false, false)) return rc;
    if (!partials || !grad_out || !opt || n_rays < 1) return fail(-1, "mne_decoder_update: NULL argument");
    if (losses && (!ray_sums || !counts || n_samples < 1)) return fail(-1, "mne_decoder_update: the loss scalars need ray_sums and counts");
    if (opt->step < 1) return fail(-1, "mne_decoder_update: bad optimizer state");
    DecUpdateArgs a = {};
    a.sc = *scene;
    a.partials = partials; a.n_partials = mne_wgrad_partial_count(*scene, n_rays); a.grad_out = grad_out;
    for (int k = 0; k < 4; ++k) {
        if (!opt->m[k] || !opt->v[k]) return fail(-1, "mne_decoder_update: bad optimizer state");
        a.m[k] = opt->m[k]; a.v[k] = opt->v[k];
    }
    PlaneOpt& o = a.opt;
    o.omb1 = (float)(1.0 - opt->beta1); o.b2 = (float)opt->beta2; o.omb2 = (float)(1.0 - opt->beta2);
    o.eps = (float)opt->eps; o.wd = (float)opt->weight_decay;
    o.step_size = (float)(opt->lr / (1.0 - std::pow(opt->beta1, (double)opt->step)));
    o.bc2_sqrt = (float)std::sqrt(1.0 - std::pow(opt->beta2, (double)opt->step));
    o.lr = opt->lr; o.step = opt->step;
    if (clock && fill_clock(clock, a.clk, opt->beta1, opt->beta2, true))
        return fail(-1, "mne_decoder_update: clock incomplete or made for other betas");
    a.fin.R = n_rays; a.fin.S = n_samples; a.fin.ray_sums = ray_sums; a.fin.counts = counts; a.fin.losses = losses;
    if (int rc = mne_launch_decoder_update(a, (hipStream_t)stream)) return fail(rc, "unsupported decoder shape");
    return check_launch("decoder_update");
}

int mne_decoder_wgrad(const mne_scene_t* scene, const float* tape, const int32_t* ray_tiles, int n_rays, int n_samples,
                      float* partials, float* grad_out, int impl, void* stream) {
    if (int rc = check_scene(scene, false, false)) return rc;
    if (!tape || !ray_tiles || (!grad_out && impl != 3) || (impl != 1 && !partials)) return fail(-1, "mne_decoder_wgrad: NULL argument");
    if (n_rays < 1 || n_samples < 1) return fail(-1, "mne_decoder_wgrad: empty batch");
    WgradArgs a = {};
    a.tape = tape; a.ray_tiles = ray_tiles; a.R = n_rays; a.S = n_samples; a.partials = partials; a.grad_out = grad_out;
    if (int rc = mne_launch_wgrad(*scene, a, impl, (hipStream_t)stream)) return fail(rc, "unsupported decoder shape");
    return check_launch("decoder_wgrad");
}

size_t mne_sizeof_pose_state(void) { return sizeof(mne_pose_state_t); }

static int fill_pose(const mne_pose_state_t* ps, PoseArgs& a) {
    if (!ps || !ps->rot || !ps->trans || !ps->m || !ps->v || !ps->step || !ps->c2w || !ps->best_loss || !ps->best_c2w || !ps->last_loss)
        return fail(-1, "mne_pose_state: NULL field");
    if (ps->n_rot != 3 && ps->n_rot != 4) return fail(-1, "mne_pose_state: n_rot must be 3 (axis-angle) or 4 (quaternion)");
    a.n_rot = ps->n_rot;
    a.rot = ps->rot; a.trans = ps->trans; a.m = ps->m; a.v = ps->v; a.step = ps->step; a.c2w = ps->c2w;
    a.best_loss = ps->best_loss; a.best_c2w = ps->best_c2w; a.last_loss = ps->last_loss;
    for (int k = 0; k < 9; ++k) a.r_base[k] = ps->r_base[k];
    a.lr_rot = ps->lr_rot; a.lr_trans = ps->lr_trans; a.beta1 = ps->beta1; a.beta2 = ps->beta2; a.eps = ps->eps;
    return 0;
}

int mne_pose_rays(const mne_pose_state_t* pose, int n_rays, const float* dirs_cam, float* rays_o, float* rays_d, void* stream) {
    PoseArgs a = {};
    if (int rc = fill_pose(pose, a)) return rc;
    if (n_rays < 1 || !dirs_cam || !rays_o || !rays_d) return fail(-1, "mne_pose_rays: NULL argument");
    a.n = n_rays; a.dirs = dirs_cam; a.rays_o = rays_o; a.rays_d = rays_d;
    mne_launch_pose(a, 0, (hipStream_t)stream);
    return check_launch("pose_rays");
}

int mne_pose_loss(int n_rays, const float* rgb, const float* depth, const float* want_rgb, const float* want_depth,
                  double w_rgb, double w_depth, float* d_rgb, float* d_depth, float* partials, void* stream) {
    if (n_rays < 1 || !rgb || !depth || !want_rgb || !want_depth || !d_rgb || !d_depth || !partials)
        return fail(-1, "mne_pose_loss: NULL argument");
    PoseArgs a = {};
    a.n = n_rays; a.rgb = rgb; a.depth = depth; a.want_rgb = want_rgb; a.want_depth = want_depth;
    a.w_rgb = (float)w_rgb; a.w_depth = (float)w_depth; a.d_rgb = d_rgb; a.d_depth = d_depth; a.partials = partials;
    mne_launch_pose(a, 1, (hipStream_t)stream);
    return check_launch("pose_loss");
}

int mne_pose_update(const mne_pose_state_t* pose, int n_rays, const float* dirs_cam, const float* d_rays_o,
                    const float* d_rays_d, const float* partials, void* stream) {
    PoseArgs a = {};
    if (int rc = fill_pose(pose, a)) return rc;
    if (n_rays < 1 || !dirs_cam || !d_rays_o || !d_rays_d || !partials) return fail(-1, "mne_pose_update: NULL argument");
    if (!(pose->beta1 >= 0.0 && pose->beta1 < 1.0 && pose->beta2 >= 0.0 && pose->beta2 < 1.0)) return fail(-1, "mne_pose_update: betas out of range");
    a.n = n_rays; a.dirs = dirs_cam; a.d_rays_o = d_rays_o; a.d_rays_d = d_rays_d; a.partials = (float*)partials;
    a.n_partials = (n_rays + 255) / 256;
    mne_launch_pose(a, 2, (hipStream_t)stream);
    return check_launch("pose_update");
}

int mne_adam_step(const mne_adam_seg_t* segs, int n_seg, int zero_grad, const mne_clock_t* clock, void* stream) {
    if (n_seg < 0 || n_seg > 32) return fail(-1, "mne_adam_step: n_seg must be in [0,32]");
    if (n_seg == 0) return 0;
    if (!segs) return fail(-1, "mne_adam_step: segs is NULL");
    AdamArgs a = {};
    a.n_seg = n_seg;
    a.zero_grad = zero_grad;
    a.blk_start[0] = 0;
    for (int s = 0; s < n_seg; ++s) {
        const mne_adam_seg_t& g = segs[s];
        if (!g.p || !g.g || !g.m || !g.v || g.n < 0 || g.step < 1) return fail(-1, "mne_adam_step: bad segment");
        a.seg[s] = g;
        const double bc1 = 1.0 - std::pow(g.beta1, (double)g.step);
        const double bc2 = 1.0 - std::pow(g.beta2, (double)g.step);
        a.step_size[s] = (float)(g.lr / bc1);
        a.bc2_sqrt[s] = (float)std::sqrt(bc2);
        a.blk_start[s + 1] = a.blk_start[s] + mne_adam_blocks_for(g.n);
        if (clock && fill_clock(clock, a.clk, g.beta1, g.beta2, true))
            return fail(-1, "mne_adam_step: clock incomplete or made for other betas");
    }
    mne_launch_adam(a, (hipStream_t)stream);
    return check_launch("adam_step");
}

int mne_encode_oneblob(int64_t n_pts, int dims, const float* x, float* out, void* stream) {
    if (!x || !out || dims < 1) return fail(-1, "mne_encode_oneblob: bad argument");
    if (n_pts <= 0) return 0;
    mne_launch_oneblob(n_pts, dims, x, out, (hipStream_t)stream);
    return check_launch("encode_oneblob");
}

int mne_encode_frequency(int64_t n_pts, int dims, int n_frequencies, const float* x, float* out, void* stream) {
    if (!x || !out || dims < 1 || n_frequencies < 1 || n_frequencies > 32) return fail(-1, "mne_encode_frequency: bad argument");
    if (n_pts <= 0) return 0;
    mne_launch_frequency(n_pts, dims, n_frequencies, x, out, (hipStream_t)stream);
    return check_launch("encode_frequency");
}

int mne_encode_frequency_backward(int64_t n_pts, int dims, int n_frequencies, const float* x, const float* d_out, float* d_x, void* stream) {
    if (!x || !d_out || !d_x || dims < 1 || n_frequencies < 1 || n_frequencies > 32) return fail(-1, "mne_encode_frequency_backward: bad argument");
    if (n_pts <= 0) return 0;
    mne_launch_frequency_backward(n_pts, dims, n_frequencies, x, d_out, d_x, (hipStream_t)stream);
    return check_launch("encode_frequency_backward");
}

int mne_encode_sh(int64_t n_pts, int degree, const float* in, float* out, void* stream) {
    if (!in || !out || degree < 1 || degree > 4) return fail(-1, "mne_encode_sh: degree 1..4 (tinycudann goes to 8; the reference asks for 4)");
    if (n_pts <= 0) return 0;
    mne_launch_sh(n_pts, degree * degree, in, out, (hipStream_t)stream);
    return check_launch("encode_sh");
}

int mne_encode_sh_backward(int64_t n_pts, int degree, const float* in, const float* d_out, float* d_in, void* stream) {
    if (!in || !d_out || !d_in || degree < 1 || degree > 4) return fail(-1, "mne_encode_sh_backward: degree 1..4");
    if (n_pts <= 0) return 0;
    mne_launch_sh_backward(n_pts, degree * degree, in, d_out, d_in, (hipStream_t)stream);
    return check_launch("encode_sh_backward");
}

int mne_encode_identity(int64_t n_elems, float scale, float offset, const float* x, float* out, void* stream) {
    if (!x || !out) return fail(-1, "mne_encode_identity: NULL argument");
    if (n_elems <= 0) return 0;
    mne_launch_identity(n_elems, scale, offset, x, out, (hipStream_t)stream);
    return check_launch("encode_identity");
}

int mne_query_features(const mne_scene_t* scene, int64_t n_pts, const float* pts, const float* features,
                       const float* packed_decoder, float* raw, float* geo, void* stream) {
    if (int rc = check_scene(scene, false, false)) return rc;
    if (!pts || !features || !packed_decoder || (!raw && !geo)) return fail(-1, "mne_query_features: NULL argument");
    if (scene->n_sets != 1) return fail(-2, "caller-supplied features replace ONE plane set (no colour planes)");
    if (n_pts <= 0) return 0;
    QueryArgs a = {};
    a.sc = *scene; a.n = n_pts; a.pts = pts; a.packed = packed_decoder; a.raw = raw; a.geo = geo; a.ext_rows = features;
    if (int rc = mne_launch_query(a, (hipStream_t)stream)) return fail(rc, "unsupported scene configuration");
    return check_launch("query_features");
}

int mne_query_points(const mne_scene_t* scene, int64_t n_pts, const float* pts, const float* packed_decoder,
                     float* raw, float* geo, float* feat, int32_t* corner_idx, int flags, void* stream) {
    if (int rc = check_scene(scene, false)) return rc;
    if (!pts || ((raw || geo) && !packed_decoder)) return fail(-1, "mne_query_points: NULL argument");
    if (n_pts <= 0) return 0;
    QueryArgs a = {};
    a.sc = *scene; a.n = n_pts; a.pts = pts; a.packed = packed_decoder; a.raw = raw; a.geo = geo; a.feat_out = feat; a.corner_idx = corner_idx; a.flags = flags;
    if ((flags & MNE_QUERY_PTS_NORMALISED) && (raw || geo)) return fail(-1, "normalised points only support the feature output");
    if (int rc = mne_launch_query(a, (hipStream_t)stream)) return fail(rc, "unsupported scene configuration");
    return check_launch("query_points");
}

static int fill_grid(const mne_grid_cfg_t* cfg, GridArgs& a) {
    if (!cfg || cfg->n_levels < 1 || cfg->n_levels > MNE_GRID_MAX_LEVELS || cfg->n_features < 1 ||
        cfg->n_features > MNE_GRID_MAX_F || cfg->base_resolution < 1 || cfg->log2_hashmap_size < 1 ||
        cfg->log2_hashmap_size > 28 || (cfg->grid_type != 0 && cfg->grid_type != 1))
        return fail(-1, "bad grid encoding configuration");
    a.n_levels = cfg->n_levels; a.n_features = cfg->n_features; a.out_dim = cfg->n_levels * cfg->n_features;
    const float l2 = (float)std::log2(cfg->per_level_scale);
    unsigned long long off = 0;
    for (int l = 0; l < cfg->n_levels; ++l) {
        const float scale = std::exp2((float)l * l2) * (float)cfg->base_resolution - 1.0f;
        const unsigned res = (unsigned)std::ceil(scale) + 1u;
        unsigned long long n = (unsigned long long)res * res * res;
        n = (n + 7) / 8 * 8;
        if (cfg->grid_type == 0 && n > (1ull << cfg->log2_hashmap_size)) n = 1ull << cfg->log2_hashmap_size;
        if (n >= (1ull << 32) || off + n >= (1ull << 32)) return fail(-1, "grid encoding too large");
        a.scale[l] = scale; a.res[l] = res; a.size[l] = (unsigned)n; a.offset[l] = (unsigned)off;
        off += n;
    }
    return (int)0;
}

int mne_grid_level_table(const mne_grid_cfg_t* cfg, float* scale, uint32_t* resolution, uint32_t* size, uint32_t* offset) {
    GridArgs a = {};
    if (int rc = fill_grid(cfg, a)) return rc;
    for (int l = 0; l < cfg->n_levels; ++l) {
        if (scale) scale[l] = a.scale[l];
        if (resolution) resolution[l] = a.res[l];
        if (size) size[l] = a.size[l];
        if (offset) offset[l] = a.offset[l];
    }
    return 0;
}

size_t mne_grid_param_count(const mne_grid_cfg_t* cfg) {
    GridArgs a = {};
    if (fill_grid(cfg, a)) return 0;
    const int l = cfg->n_levels - 1;
    return ((size_t)a.offset[l] + a.size[l]) * cfg->n_features;
}

int mne_grid_encode(const mne_grid_cfg_t* cfg, int64_t n_pts, const float* x, const float* params,
                    float* out, uint32_t* idx, void* stream) {
    GridArgs a = {};
    if (int rc = fill_grid(cfg, a)) return rc;
    if (!x || !params || !out) return fail(-1, "mne_grid_encode: NULL argument");
    if (n_pts <= 0) return 0;
    a.n = n_pts; a.x = x; a.params = params; a.out = out; a.idx_out = idx;
    mne_launch_grid(a, 0, (hipStream_t)stream);
    return check_launch("grid_encode");
}

int mne_grid_encode_box(const mne_grid_cfg_t* cfg, const mne_scene_t* scene, int64_t n_pts, const float* pts,
                        const float* params, float* out, int out_stride, void* stream) {
    GridArgs a = {};
    if (int rc = fill_grid(cfg, a)) return rc;
    if (!scene || !pts || !params || !out || out_stride < a.out_dim) return fail(-1, "mne_grid_encode_box: bad argument");
    if (n_pts <= 0) return 0;
    a.n = n_pts; a.x = pts; a.params = params; a.out = out; a.out_stride = out_stride; a.x_is_world = 1;
    for (int k = 0; k < 3; ++k) { a.bb_lo[k] = scene->bb_lo[k]; a.bb_hi[k] = scene->bb_hi[k]; }
    a.bb_is_f64 = scene->bb_is_f64;
    mne_launch_grid(a, 0, (hipStream_t)stream);
    return check_launch("grid_encode_box");
}

int mne_grid_encode_backward(const mne_grid_cfg_t* cfg, int64_t n_pts, const float* x, const float* dout,
                             float* dparams, void* stream) {
    GridArgs a = {};
    if (int rc = fill_grid(cfg, a)) return rc;
    if (!x || !dout || !dparams) return fail(-1, "mne_grid_encode_backward: NULL argument");
    if (n_pts <= 0) return 0;
    a.n = n_pts; a.x = x; a.dout = dout; a.dparams = dparams;
    mne_launch_grid(a, 1, (hipStream_t)stream);
    return check_launch("grid_encode_backward");
}

}  // extern "C"

static int fill_hash_rows(const mne_grid_cfg_t* cfg, const mne_scene_t* scene, int n_rays, int n_samples, const float* rays_o,
                          const float* rays_d, const float* z_vals, float* tape, GridArgs& a) {
    if (int rc = fill_grid(cfg, a)) return rc;
    if (int rc = check_scene(scene, false, false)) return rc;
    if (cfg->n_features != 2 || cfg->n_levels > 16) return fail(-2, "the fused form needs 2 features per level and at most 16 levels");
    if (scene->n_sets != 1) return fail(-2, "the fused hash-grid form has one feature set (no colour planes)");
    if (!rays_o || !rays_d || !z_vals || !tape) return fail(-1, "hash rows: NULL argument");
    a.R = n_rays; a.S = n_samples;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z_vals = z_vals; a.tape = tape;
    a.row_stride = (int)mne_dims_tape_row(*scene); a.col_x = 0; a.col_d = (int)mne_dims_tape_dfeat(*scene);
    for (int k = 0; k < 3; ++k) { a.bb_lo[k] = scene->bb_lo[k]; a.bb_hi[k] = scene->bb_hi[k]; }
    a.bb_is_f64 = scene->bb_is_f64;
    return 0;
}

extern "C" {

int mne_hash_gather(const mne_grid_cfg_t* cfg, const mne_scene_t* scene, int n_rays, int n_samples, const float* rays_o,
                    const float* rays_d, const float* z_vals, const int32_t* ray_counts, const float* table, float* tape, void* stream) {
    GridArgs a = {};
    if (int rc = fill_hash_rows(cfg, scene, n_rays, n_samples, rays_o, rays_d, z_vals, tape, a)) return rc;
    if (!table) return fail(-1, "mne_hash_gather: NULL argument");
    if (n_rays <= 0) return 0;
    a.params = table;
    a.ray_counts = ray_counts;
    mne_launch_hash_rows(a, 0, (hipStream_t)stream);
    return check_launch("hash_gather");
}

int mne_hash_features(const mne_grid_cfg_t* cfg, const mne_scene_t* scene, int n_rays, int n_samples, const float* rays_o,
                      const float* rays_d, const float* z_vals, const float* table, float* features, void* stream) {
    GridArgs a = {};
    if (int rc = fill_hash_rows(cfg, scene, n_rays, n_samples, rays_o, rays_d, z_vals, features, a)) return rc;
    if (!table) return fail(-1, "mne_hash_features: NULL argument");
    if (n_rays <= 0) return 0;
    a.params = table;
    a.row_stride = 64; a.col_x = 0;            // compact [R*S][64] rows (the decoder's feature slot), not tape rows
    mne_launch_hash_rows(a, 0, (hipStream_t)stream);
    return check_launch("hash_features");
}

size_t mne_hash_workspace_bytes(const mne_grid_cfg_t* cfg, int n_rays, int n_samples) {
    GridArgs a = {};
    if (n_rays <= 0 || n_samples <= 0 || fill_grid(cfg, a)) return 0;
    return mne_hash_layout(a, n_rays, n_samples, nullptr);
}

int mne_hash_slice_adam(const mne_grid_cfg_t* cfg, const mne_scene_t* scene, int n_rays, int n_samples, const float* rays_o,
                        const float* rays_d, const float* z_vals, const float* tape, const int32_t* ray_tiles, float* table,
                        const mne_plane_opt_t* opt, void* workspace, size_t workspace_bytes, void* event_after_bin, void* stream) {
    GridArgs a = {};
    if (int rc = fill_hash_rows(cfg, scene, n_rays, n_samples, rays_o, rays_d, z_vals, (float*)tape, a)) return rc;
    if (!ray_tiles || !table || !opt || !workspace) return fail(-1, "mne_hash_slice_adam: NULL argument");
    if (!opt->m || !opt->v || opt->step < 1) return fail(-1, "mne_hash_slice_adam: bad optimizer state");
    if (n_rays <= 0) return 0;
    if (workspace_bytes < mne_hash_layout(a, n_rays, n_samples, workspace)) return fail(-1, "mne_hash_slice_adam: workspace too small");
    a.ray_tiles = ray_tiles; a.params = table;
    PlaneOpt& o = a.opt;
    o.m = opt->m; o.v = opt->v;
    o.omb1 = (float)(1.0 - opt->beta1); o.b2 = (float)opt->beta2; o.omb2 = (float)(1.0 - opt->beta2);
    o.eps = (float)opt->eps; o.wd = (float)opt->weight_decay;
    o.step_size = (float)(opt->lr / (1.0 - std::pow(opt->beta1, (double)opt->step)));
    o.bc2_sqrt = (float)std::sqrt(1.0 - std::pow(opt->beta2, (double)opt->step));
    o.lr = opt->lr; o.step = opt->step;
    if (int rc = mne_launch_hash_slice_adam(a, (hipStream_t)stream, event_after_bin)) return fail(rc, "mne_hash_slice_adam: table too large for the slice kernels");
    return check_launch("hash_slice_adam");
}

int mne_hash_ray_grad(const mne_grid_cfg_t* cfg, const mne_scene_t* scene, int n_rays, int n_samples, const float* rays_o,
                      const float* rays_d, const float* z_vals, const float* table, const float* tape, const int32_t* ray_tiles,
                      float* d_rays_o, float* d_rays_d, void* stream) {
    GridArgs a = {};
    if (int rc = fill_hash_rows(cfg, scene, n_rays, n_samples, rays_o, rays_d, z_vals, (float*)tape, a)) return rc;
    if (!table || !ray_tiles || (!d_rays_o && !d_rays_d)) return fail(-1, "mne_hash_ray_grad: NULL argument");
    if (n_rays <= 0) return 0;
    a.params = table; a.ray_tiles = ray_tiles;
    mne_launch_hash_raygrad(a, d_rays_o, d_rays_d, (hipStream_t)stream);
    return check_launch("hash_ray_grad");
}

int mne_hash_scatter(const mne_grid_cfg_t* cfg, const mne_scene_t* scene, int n_rays, int n_samples, const float* rays_o,
                     const float* rays_d, const float* z_vals, const float* tape, const int32_t* ray_tiles,
                     float* grad_table, void* stream) {
    GridArgs a = {};
    if (int rc = fill_hash_rows(cfg, scene, n_rays, n_samples, rays_o, rays_d, z_vals, (float*)tape, a)) return rc;
    if (!ray_tiles || !grad_table) return fail(-1, "mne_hash_scatter: NULL argument");
    if (n_rays <= 0) return 0;
    a.ray_tiles = ray_tiles; a.dparams = grad_table;
    mne_launch_hash_rows(a, 2, (hipStream_t)stream);                    // run-reduced global atomics
    return check_launch("hash_scatter");
}

}  // extern "C"
