"""Glue between PyTorch tensors and the C ABI for the render path: struct marshalling, scratch
allocation (through torch's caching allocator, on the caller's stream) and the autograd node that
makes ``JointEncoding.forward / render_rays`` differentiable exactly like the reference's graph
(planes, decoder weights; rays when requested).

Nothing here computes per-sample math in PyTorch: every tensor op below is allocation, a view, or a
layout conversion of a caller-provided plane.
"""
import ctypes as C

import torch

from . import _lib


# --------------------------------------------------------------------------------------------------
# marshalling
# --------------------------------------------------------------------------------------------------
def render_cfg_struct(config):
    tr, cam = config["training"], config["cam"]
    rc = _lib.RenderCfg()
    rc.near_z, rc.far_z = float(cam["near"]), float(cam["far"])
    rc.range_d = float(tr["range_d"])
    rc.perturb = float(tr["perturb"])
    rc.trunc = float(tr["trunc"])
    rc.sc_factor = float(config["data"]["sc_factor"])
    rc.truncation = float(config["model"]["truncation"])
    rc.depth_trunc = float(cam["depth_trunc"])
    rc.n_samples = int(tr.get("n_samples", 0) or 0)
    rc.n_samples_d = int(tr["n_samples_d"])
    rc.n_range_d = int(tr["n_range_d"])
    return rc


def as_channels_last(plane):
    """Physical [H][W][C] view of a logical [1,C,H,W] plane (copy only if the caller handed us a
    plane in another layout, e.g. a peer's NCHW checkpoint; the copy stays in the autograd graph)."""
    if plane.dim() != 4 or plane.shape[0] != 1:
        raise ValueError(f"plane must be [1,C,H,W], got {tuple(plane.shape)}")
    if plane.dtype not in (torch.float32, torch.float16):
        raise TypeError("planes must be float32 (or float16: the half-precision plane storage extension)")
    if plane.is_contiguous(memory_format=torch.channels_last) and plane.stride(1) == 1:
        return plane
    return plane.contiguous(memory_format=torch.channels_last)


def scene_struct(model_info, planes_cl, dec_w, grads=None):
    """Fill mne_scene_t.  ``planes_cl``: flat list in all_planes order
    (xy[coarse,fine], xz[...], yz[...], then the colour planes); ``grads``: same order or None.  Planes are fp32, or ALL
    float16 (EXTENSION, BASELINE configs[4]: half-precision plane storage, mne_scene_t.plane_f16)."""
    sc = _lib.Scene()
    n_sets = len(planes_cl) // 6
    sc.n_sets = n_sets
    sc.c_dim = model_info["c_dim"]
    sc.hidden, sc.hidden_color = model_info["hidden"], model_info["hidden_color"]
    sc.geo_feat_dim, sc.n_bins = model_info["geo_feat_dim"], model_info["n_bins"]
    sc.bb_is_f64 = 1 if model_info["bb_is_f64"] else 0
    dtypes = {p.dtype for p in planes_cl}
    if len(dtypes) > 1 or not dtypes <= {torch.float32, torch.float16}:
        raise TypeError(f"planes must all be float32 or all float16, got {sorted(map(str, dtypes))}")
    sc.plane_f16 = 1 if dtypes == {torch.float16} else 0
    for s in range(n_sets):
        for o in range(3):
            for l in range(2):
                p = planes_cl[s * 6 + o * 2 + l]
                pl = sc.plane[s][o][l]
                pl.data = p.data_ptr()
                pl.h, pl.w = p.shape[2], p.shape[3]
                pl.grad = grads[s * 6 + o * 2 + l].data_ptr() if grads is not None else None
    for k in range(3):
        sc.bound_lo[k], sc.bound_hi[k] = model_info["bound_lo"][k], model_info["bound_hi"][k]
        sc.bb_lo[k], sc.bb_hi[k] = model_info["bb_lo"][k], model_info["bb_hi"][k]
    w_sdf0, w_sdf1, w_col0, w_col1 = dec_w
    sc.w_sdf0, sc.w_sdf1 = w_sdf0.data_ptr(), w_sdf1.data_ptr()
    sc.w_col0, sc.w_col1 = w_col0.data_ptr(), w_col1.data_ptr()
    return sc


def _f32c(t, name):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def linspace_tables(config, has_depth, device):
    """The linspace values the reference computes on the CPU with torch.linspace
    (model/scene_rep.py:363-373), uploaded once per call (a few hundred bytes)."""
    tr, cam = config["training"], config["cam"]
    if has_depth:
        parts = [torch.linspace(cam["near"], cam["far"], tr["n_samples_d"]),
                 torch.linspace(-tr["range_d"], tr["range_d"], steps=tr["n_range_d"]),
                 torch.linspace(cam["near"], cam["far"], steps=tr["n_range_d"])]
        return torch.cat(parts).to(device)
    return torch.linspace(cam["near"], cam["far"], tr["n_samples"]).to(device)


# --------------------------------------------------------------------------------------------------
# the autograd node
# --------------------------------------------------------------------------------------------------
class RenderFunction(torch.autograd.Function):
    """(rays, targets, jitter, planes..., decoder weights...) ->
       rgb[R,3], depth[R], disp[R], acc[R], depth_var[R], z_vals[R,S], raw[R,S,4], losses[8].

    forward  = mne_sample_z + mne_pack_decoder + mne_render_forward (+ mne_loss_finalize)
    backward = mne_loss_coef + mne_render_backward + mne_decoder_wgrad
    z_vals, raw, disp, acc and depth_var are returned as non-differentiable (the reference's losses
    on raw/z are computed inside the node)."""

    @staticmethod
    def forward(ctx, info, tables, rays_o, rays_d, target_rgb, target_d, u, seed_offset, *params):
        lib = _lib.load()
        n_planes = info["n_planes"]
        planes = params[:n_planes]
        dec_w = params[n_planes:]
        dev = rays_o.device
        st = _lib.stream_for(rays_o)
        rc = info["render_cfg"]
        R = rays_o.shape[0]
        has_d = target_d is not None
        S = lib.mne_num_samples(C.byref(rc), 1 if has_d else 0)
        rays_o_c, rays_d_c = _f32c(rays_o.detach(), "rays_o"), _f32c(rays_d.detach(), "rays_d")
        tgt_rgb = _f32c(target_rgb, "target_rgb")
        tgt_d = _f32c(target_d.reshape(-1), "target_d") if has_d else None
        u_c = _f32c(u, "u")
        opts = dict(device=dev, dtype=torch.float32)
        z_vals = torch.empty(R, S, **opts)
        counts = torch.empty(_lib.N_COUNT, device=dev, dtype=torch.int32)
        ray_counts = torch.empty(R, _lib.N_COUNT, device=dev, dtype=torch.int32) if has_d else None
        seed, offset = seed_offset
        _lib.check(lib.mne_sample_z(C.byref(rc), R, _lib.ptr(tgt_d), _lib.ptr(u_c), _lib.ptr(tables), seed, offset,
                                    _lib.ptr(z_vals), _lib.ptr(counts), _lib.ptr(ray_counts), None, st), "mne_sample_z")
        sc = scene_struct(info, [p.detach() for p in planes], [w.detach() for w in dec_w])
        packed = torch.empty(lib.mne_packed_decoder_floats(C.byref(sc)), **opts)
        _lib.check(lib.mne_pack_decoder(C.byref(sc), _lib.ptr(packed), st), "mne_pack_decoder")
        rgb, depth = torch.empty(R, 3, **opts), torch.empty(R, **opts)
        disp, acc, var = torch.empty(R, **opts), torch.empty(R, **opts), torch.empty(R, **opts)
        raw = torch.empty(R, S, 4, **opts)
        want_losses = has_d and target_rgb is not None
        ray_sums = torch.empty(R, _lib.N_LOSS, **opts) if want_losses else None
        _lib.check(lib.mne_render_forward(C.byref(sc), C.byref(rc), R, S, _lib.ptr(rays_o_c), _lib.ptr(rays_d_c),
                                          _lib.ptr(tgt_rgb) if want_losses else None, _lib.ptr(tgt_d),
                                          _lib.ptr(z_vals), _lib.ptr(packed), _lib.ptr(rgb), _lib.ptr(depth),
                                          _lib.ptr(disp), _lib.ptr(acc), _lib.ptr(var), _lib.ptr(raw),
                                          _lib.ptr(ray_sums), None, 0, st), "mne_render_forward")
        losses = torch.zeros(_lib.N_LOSS, **opts)
        if want_losses:
            _lib.check(lib.mne_loss_finalize(R, S, _lib.ptr(ray_sums), _lib.ptr(counts), _lib.ptr(losses), st),
                       "mne_loss_finalize")
        ctx.info, ctx.S, ctx.want_losses = info, S, want_losses
        if any(p.dtype == torch.float16 for p in planes):
            # half-precision planes: backward leaves their fp32 gradient sums on the parameter objects the caller names
            ctx.plane_params = list(info.get("plane_owners") or planes)
        ctx.save_for_backward(rays_o_c, rays_d_c, tgt_rgb, tgt_d, z_vals, raw, counts, ray_counts, packed, *params)
        ctx.mark_non_differentiable(disp, acc, var, z_vals, raw)
        return rgb, depth, disp, acc, var, z_vals, raw, losses

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_disp, g_acc, g_var, g_z, g_raw, g_losses):
        lib = _lib.load()
        info, S = ctx.info, ctx.S
        rays_o, rays_d, tgt_rgb, tgt_d, z_vals, raw, counts, ray_counts, packed, *params = ctx.saved_tensors
        n_planes = info["n_planes"]
        planes, dec_w = params[:n_planes], params[n_planes:]
        dev, st = rays_o.device, _lib.stream_for(rays_o)
        rc = info["render_cfg"]
        R = rays_o.shape[0]
        opts = dict(device=dev, dtype=torch.float32)
        want_ray = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        d_o = torch.zeros(R, 3, **opts) if want_ray else None
        d_d = torch.zeros(R, 3, **opts) if want_ray else None
        # plane gradients only when some plane asks for one (the pose-alignment loop differentiates w.r.t. rays only:
        # no zero-fill of 150-300 MB, no scatter)
        first_plane = 8
        want_planes = any(ctx.needs_input_grad[first_plane:first_plane + n_planes])
        # (fp32 accumulators also for half-precision planes; autograd gets them in the planes' dtype below)
        grads = [torch.zeros_like(p, dtype=torch.float32) for p in planes] if want_planes else None     # channels_last preserved
        sc = scene_struct(info, list(planes), list(dec_w), grads)
        coef = None
        if ctx.want_losses and g_losses is not None:
            coef = torch.empty(_lib.N_LOSS, **opts)
            g_l = _f32c(g_losses, "g")                   # (named: a converted copy must outlive the launch that reads it)
            _lib.check(lib.mne_loss_coef(C.byref(rc), R, S, _lib.ptr(counts), _lib.ptr(g_l),
                                         _lib.ptr(coef), st), "mne_loss_coef")
        # incoming gradients as dense fp32: a converted / made-contiguous COPY is a temporary -- it is kept in a name until the
        # call that enqueues its reader has returned (found by the emulator's AddressSanitizer build: use after free)
        g_rgb_c, g_depth_c = _f32c(g_rgb, "g_rgb"), _f32c(g_depth, "g_depth")
        row = lib.mne_tape_row_floats(C.byref(sc))
        tape = torch.empty(R * S, row, **opts)
        tape_rows = torch.zeros(1, device=dev, dtype=torch.int32)
        ray_tiles = torch.empty(R, device=dev, dtype=torch.int32)
        ws_bytes = lib.mne_render_workspace_bytes(R, S)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        _lib.check(lib.mne_render_backward(C.byref(sc), C.byref(rc), R, S, _lib.ptr(rays_o), _lib.ptr(rays_d),
                                           _lib.ptr(tgt_rgb) if coef is not None else None,
                                           _lib.ptr(tgt_d), _lib.ptr(z_vals), _lib.ptr(ray_counts), _lib.ptr(packed),
                                           _lib.ptr(raw), _lib.ptr(coef), _lib.ptr(g_rgb_c),
                                           _lib.ptr(g_depth_c), _lib.ptr(tape), R * S,
                                           _lib.ptr(tape_rows), _lib.ptr(ray_tiles), _lib.ptr(d_o), _lib.ptr(d_d),
                                           _lib.ptr(ws), ws_bytes, st),
                   "mne_render_backward")
        nparam = lib.mne_decoder_param_floats(C.byref(sc))
        partials = torch.empty(lib.mne_wgrad_partial_floats(C.byref(sc)), **opts)
        dgrad = torch.empty(nparam, **opts)
        _lib.check(lib.mne_decoder_wgrad(C.byref(sc), _lib.ptr(tape), _lib.ptr(ray_tiles), R, S, _lib.ptr(partials),
                                         _lib.ptr(dgrad), info.get("wgrad_impl", 0), st), "mne_decoder_wgrad")
        w_sdf0, w_sdf1, w_col0, w_col1 = dec_w
        n0, n1, n2 = w_col0.numel(), w_col1.numel(), w_sdf0.numel()
        g_col0 = dgrad[:n0].view_as(w_col0)
        g_col1 = dgrad[n0:n0 + n1].view_as(w_col1)
        g_sdf0 = dgrad[n0 + n1:n0 + n1 + n2].view_as(w_sdf0)
        g_sdf1 = dgrad[n0 + n1 + n2:].view_as(w_sdf1)
        if grads is not None:
            for g, p in zip(grads, getattr(ctx, "plane_params", None) or planes):     # (the caller's own tensor objects)
                if p.dtype == torch.float16:
                    # autograd wants the parameter's dtype; the fp32 sums are kept beside it for the optimizer (FusedAdam reads
                    # ``grad32``, accumulated over backward calls like .grad, cleared by zero_grad): a cast to fp16 flushes a
                    # mean-reduced plane gradient to zero below ~3e-8
                    acc = getattr(p, "grad32", None)
                    if acc is None or acc.shape != g.shape or acc.device != g.device:
                        p.grad32 = g.clone()
                    else:
                        acc.add_(g)
        return (None, None, d_o if ctx.needs_input_grad[2] else None, d_d if ctx.needs_input_grad[3] else None,
                None, None, None, None,
                *([g if g.dtype == p.dtype else g.to(p.dtype) for g, p in zip(grads, planes)] if grads is not None else [None] * n_planes),
                g_sdf0, g_sdf1, g_col0, g_col1)


@torch.no_grad()
def render_maps(info, tables, rays_o, rays_d, target_d, u, seed_offset, planes, dec_w, early_termination=True, stats=None):
    """No-grad rendering of any number of rays in one launch sequence (no autograd node, no tape, no backward
    workspace): mne_sample_z + mne_pack_decoder + mne_render_forward, by default with exact early ray termination --
    a ray's samples are decoded only up to the last one that can influence its maps.  Returns rgb [R,3], depth, disp,
    acc, depth_var [R].  What ``render_img`` / teacher renders / visualisation use.
    ``stats`` (measurement only: a dict; costs a fill of the scratch and a host sync): ``decoded_samples`` / ``nominal_samples``
    are ADDED to it -- the samples the exact early termination really decoded (their ``raw`` entries are the ones written)."""
    lib = _lib.load()
    dev, st = rays_o.device, _lib.stream_for(rays_o)
    rc = info["render_cfg"]
    R = rays_o.shape[0]
    has_d = target_d is not None
    S = lib.mne_num_samples(C.byref(rc), 1 if has_d else 0)
    rays_o_c, rays_d_c = _f32c(rays_o.detach(), "rays_o"), _f32c(rays_d.detach(), "rays_d")
    tgt_d = _f32c(target_d.reshape(-1), "target_d") if has_d else None
    u_c = _f32c(u, "u")
    opts = dict(device=dev, dtype=torch.float32)
    z_vals = torch.empty(R, S, **opts)
    counts = torch.empty(_lib.N_COUNT, device=dev, dtype=torch.int32) if has_d else None
    ray_counts = torch.empty(R, _lib.N_COUNT, device=dev, dtype=torch.int32) if has_d else None
    seed, offset = seed_offset
    _lib.check(lib.mne_sample_z(C.byref(rc), R, _lib.ptr(tgt_d), _lib.ptr(u_c), _lib.ptr(tables), seed, offset,
                                _lib.ptr(z_vals), _lib.ptr(counts), _lib.ptr(ray_counts), None, st), "mne_sample_z")
    sc = scene_struct(info, [as_channels_last(p.detach()) for p in planes], [w.detach() for w in dec_w])
    packed = torch.empty(lib.mne_packed_decoder_floats(C.byref(sc)), **opts)
    _lib.check(lib.mne_pack_decoder(C.byref(sc), _lib.ptr(packed), st), "mne_pack_decoder")
    rgb, depth = torch.empty(R, 3, **opts), torch.empty(R, **opts)
    disp, acc, var = torch.empty(R, **opts), torch.empty(R, **opts), torch.empty(R, **opts)
    raw = torch.empty(R, S, 4, **opts)                  # scratch under early termination
    if stats is not None:
        raw.fill_(float("nan"))
    _lib.check(lib.mne_render_forward(C.byref(sc), C.byref(rc), R, S, _lib.ptr(rays_o_c), _lib.ptr(rays_d_c), None,
                                      _lib.ptr(tgt_d), _lib.ptr(z_vals), _lib.ptr(packed), _lib.ptr(rgb), _lib.ptr(depth),
                                      _lib.ptr(disp), _lib.ptr(acc), _lib.ptr(var), _lib.ptr(raw), None,
                                      _lib.ptr(ray_counts) if early_termination else None,
                                      _lib.RENDER_EARLY_TERMINATION if early_termination else 0, st), "mne_render_forward")
    if stats is not None:
        sdf = raw[..., 3]
        stats["decoded_samples"] = stats.get("decoded_samples", 0) + int((sdf == sdf).sum().item())
        stats["nominal_samples"] = stats.get("nominal_samples", 0) + R * S
    return rgb, depth, disp, acc, var


def query_points(info, planes, dec_w, pts, want_raw=True, want_geo=False, want_feat=False, normalised=False,
                 want_corner_idx=False):
    """Forward-only point query (no autograd): raw [N,4], geo [N,15], feat [N,64]
    (+ corner_idx [N, n_planes/2, 2, 2] int32 = (ix0, iy0) per plane when asked for)."""
    lib = _lib.load()
    flat = _f32c(pts.reshape(-1, 3).detach(), "pts")
    n = flat.shape[0]
    opts = dict(device=flat.device, dtype=torch.float32)
    st = _lib.stream_for(flat)
    planes_cl = [as_channels_last(p.detach()) for p in planes]
    sc = scene_struct(info, planes_cl, [w.detach() for w in dec_w])
    packed = torch.empty(lib.mne_packed_decoder_floats(C.byref(sc)), **opts)
    _lib.check(lib.mne_pack_decoder(C.byref(sc), _lib.ptr(packed), st), "mne_pack_decoder")
    raw = torch.empty(n, 4, **opts) if want_raw else None
    geo = torch.empty(n, info["geo_feat_dim"], **opts) if want_geo else None
    feat = torch.empty(n, 2 * info["c_dim"], **opts) if want_feat else None
    cidx = torch.empty(n, len(planes_cl) // 2, 2, 2, device=flat.device, dtype=torch.int32) if want_corner_idx else None
    _lib.check(lib.mne_query_points(C.byref(sc), n, _lib.ptr(flat), _lib.ptr(packed), _lib.ptr(raw), _lib.ptr(geo),
                                    _lib.ptr(feat), _lib.ptr(cidx), 1 if normalised else 0, st), "mne_query_points")
    if want_corner_idx:
        return raw, geo, feat, cidx
    return raw, geo, feat


# --------------------------------------------------------------------------------------------------
# NS-a: the same surface for a scene model whose encoding is a multiresolution hash / dense grid
# (model/scene_rep_hash.py; EXTENSION, parity unpinned).  The grid features are the caller-supplied feature rows of the
# *_features entry points; everything behind them (OneBlob, decoder, compositing, losses) is the tri-plane path's.
# --------------------------------------------------------------------------------------------------
def _hash_scene(info, dec_w):
    sc = scene_struct(info, [], [w.detach() for w in dec_w])
    sc.n_sets = 1
    return sc


def _hash_forward(lib, info, grid_cfg, tables, rays_o, rays_d, tgt_rgb, tgt_d, u, seed_offset, table, dec_w, early, want_losses):
    """sample_z + pack + grid features of every sample + forward render.  Returns everything a caller may need."""
    dev, st = rays_o.device, _lib.stream_for(rays_o)
    rc = info["render_cfg"]
    R = rays_o.shape[0]
    has_d = tgt_d is not None
    S = lib.mne_num_samples(C.byref(rc), 1 if has_d else 0)
    opts = dict(device=dev, dtype=torch.float32)
    z_vals = torch.empty(R, S, **opts)
    counts = torch.empty(_lib.N_COUNT, device=dev, dtype=torch.int32) if has_d else None
    ray_counts = torch.empty(R, _lib.N_COUNT, device=dev, dtype=torch.int32) if has_d else None
    seed, offset = seed_offset
    _lib.check(lib.mne_sample_z(C.byref(rc), R, _lib.ptr(tgt_d), _lib.ptr(u), _lib.ptr(tables), seed, offset,
                                _lib.ptr(z_vals), _lib.ptr(counts), _lib.ptr(ray_counts), None, st), "mne_sample_z")
    sc = _hash_scene(info, dec_w)
    packed = torch.empty(lib.mne_packed_decoder_floats(C.byref(sc)), **opts)
    _lib.check(lib.mne_pack_decoder(C.byref(sc), _lib.ptr(packed), st), "mne_pack_decoder")
    feats = torch.zeros(R * S, 64, **opts)               # the grid fills the first n_levels*2 columns of the 64-wide slot
    _lib.check(lib.mne_hash_features(C.byref(grid_cfg), C.byref(sc), R, S, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(z_vals),
                                     _lib.ptr(table.detach()), _lib.ptr(feats), st), "mne_hash_features")
    rgb, depth = torch.empty(R, 3, **opts), torch.empty(R, **opts)
    disp, acc, var = torch.empty(R, **opts), torch.empty(R, **opts), torch.empty(R, **opts)
    raw = torch.empty(R, S, 4, **opts)
    ray_sums = torch.empty(R, _lib.N_LOSS, **opts) if want_losses else None
    _lib.check(lib.mne_render_forward_features(C.byref(sc), C.byref(rc), R, S, _lib.ptr(rays_o), _lib.ptr(rays_d),
                                               _lib.ptr(tgt_rgb) if want_losses else None, _lib.ptr(tgt_d), _lib.ptr(z_vals),
                                               _lib.ptr(packed), _lib.ptr(feats), _lib.ptr(rgb), _lib.ptr(depth), _lib.ptr(disp),
                                               _lib.ptr(acc), _lib.ptr(var), _lib.ptr(raw), _lib.ptr(ray_sums),
                                               _lib.ptr(ray_counts) if early else None,
                                               _lib.RENDER_EARLY_TERMINATION if early else 0, st), "mne_render_forward_features")
    return dict(S=S, sc=sc, z_vals=z_vals, counts=counts, ray_counts=ray_counts, packed=packed, rgb=rgb, depth=depth, disp=disp,
                acc=acc, var=var, raw=raw, ray_sums=ray_sums)


class HashRenderFunction(torch.autograd.Function):
    """RenderFunction for a hash / dense grid scene model:
       (rays, targets, jitter, table, decoder weights...) -> rgb, depth, disp, acc, depth_var, z_vals, raw, losses[8].
    forward  = mne_sample_z + mne_pack_decoder + mne_hash_features + mne_render_forward_features (+ mne_loss_finalize)
    backward = mne_loss_coef + mne_hash_gather + mne_render_backward_features + mne_hash_scatter + mne_decoder_wgrad.
    backward (+ rays) adds mne_hash_ray_grad.  Differentiable w.r.t. the table, the decoder and the rays (R13: the pose
    loops of loop closure, mp_slam/mapper.py:388-408, run on this model through the host's own autograd loop)."""

    @staticmethod
    def forward(ctx, info, grid_cfg, tables, rays_o, rays_d, target_rgb, target_d, u, seed_offset, table, *dec_w):
        lib = _lib.load()
        rays_o_c, rays_d_c = _f32c(rays_o.detach(), "rays_o"), _f32c(rays_d.detach(), "rays_d")
        tgt_rgb = _f32c(target_rgb, "target_rgb")
        tgt_d = _f32c(target_d.reshape(-1), "target_d") if target_d is not None else None
        want_losses = tgt_d is not None and target_rgb is not None
        f = _hash_forward(lib, info, grid_cfg, tables, rays_o_c, rays_d_c, tgt_rgb, tgt_d, _f32c(u, "u"), seed_offset, table, dec_w,
                          early=False, want_losses=want_losses)
        losses = torch.zeros(_lib.N_LOSS, device=rays_o.device, dtype=torch.float32)
        if want_losses:
            _lib.check(lib.mne_loss_finalize(rays_o.shape[0], f["S"], _lib.ptr(f["ray_sums"]), _lib.ptr(f["counts"]), _lib.ptr(losses),
                                             _lib.stream_for(rays_o)), "mne_loss_finalize")
        ctx.info, ctx.grid_cfg, ctx.S, ctx.want_losses = info, grid_cfg, f["S"], want_losses
        ctx.save_for_backward(rays_o_c, rays_d_c, tgt_rgb, tgt_d, f["z_vals"], f["raw"], f["counts"], f["ray_counts"], f["packed"],
                              table, *dec_w)
        ctx.mark_non_differentiable(f["disp"], f["acc"], f["var"], f["z_vals"], f["raw"])
        return f["rgb"], f["depth"], f["disp"], f["acc"], f["var"], f["z_vals"], f["raw"], losses

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_disp, g_acc, g_var, g_z, g_raw, g_losses):
        lib = _lib.load()
        info, S, gc = ctx.info, ctx.S, ctx.grid_cfg
        want_ray = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        rays_o, rays_d, tgt_rgb, tgt_d, z_vals, raw, counts, ray_counts, packed, table, *dec_w = ctx.saved_tensors
        dev, st = rays_o.device, _lib.stream_for(rays_o)
        rc = info["render_cfg"]
        R = rays_o.shape[0]
        opts = dict(device=dev, dtype=torch.float32)
        sc = _hash_scene(info, dec_w)
        coef = None
        if ctx.want_losses and g_losses is not None:
            coef = torch.empty(_lib.N_LOSS, **opts)
            g_l = _f32c(g_losses, "g")                   # (named: a converted copy must outlive the launch that reads it)
            _lib.check(lib.mne_loss_coef(C.byref(rc), R, S, _lib.ptr(counts), _lib.ptr(g_l), _lib.ptr(coef), st),
                       "mne_loss_coef")
        g_rgb_c, g_depth_c = _f32c(g_rgb, "g_rgb"), _f32c(g_depth, "g_depth")
        row = lib.mne_tape_row_floats(C.byref(sc))
        tape = torch.empty(R * S, row, **opts)
        tape[:, :64].zero_()                               # feature columns the grid does not fill must read as zero
        _lib.check(lib.mne_hash_gather(C.byref(gc), C.byref(sc), R, S, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(z_vals), None,
                                       _lib.ptr(table.detach()), _lib.ptr(tape), st), "mne_hash_gather")
        tape_rows = torch.zeros(1, device=dev, dtype=torch.int32)
        ray_tiles = torch.empty(R, device=dev, dtype=torch.int32)
        ws_bytes = lib.mne_render_workspace_bytes(R, S)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        d_o = torch.zeros(R, 3, **opts) if want_ray else None
        d_d = torch.zeros(R, 3, **opts) if want_ray else None
        _lib.check(lib.mne_render_backward_features(C.byref(sc), C.byref(rc), R, S, _lib.ptr(rays_o), _lib.ptr(rays_d),
                                                    _lib.ptr(tgt_rgb) if coef is not None else None, _lib.ptr(tgt_d),
                                                    _lib.ptr(z_vals), _lib.ptr(ray_counts), _lib.ptr(packed), _lib.ptr(raw),
                                                    _lib.ptr(coef), _lib.ptr(g_rgb_c), _lib.ptr(g_depth_c),
                                                    _lib.ptr(tape), R * S, _lib.ptr(tape_rows), _lib.ptr(ray_tiles), _lib.ptr(d_o), _lib.ptr(d_d),
                                                    _lib.ptr(ws), ws_bytes, st), "mne_render_backward_features")
        if want_ray:                                    # + the grid's share (trilinear weights of every level)
            _lib.check(lib.mne_hash_ray_grad(C.byref(gc), C.byref(sc), R, S, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(z_vals),
                                             _lib.ptr(table.detach()), _lib.ptr(tape), _lib.ptr(ray_tiles), _lib.ptr(d_o), _lib.ptr(d_d), st),
                       "mne_hash_ray_grad")
        g_table = None
        if ctx.needs_input_grad[9]:
            g_table = torch.zeros_like(table)
            _lib.check(lib.mne_hash_scatter(C.byref(gc), C.byref(sc), R, S, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(z_vals),
                                            _lib.ptr(tape), _lib.ptr(ray_tiles), _lib.ptr(g_table), st), "mne_hash_scatter")
        partials = torch.empty(lib.mne_wgrad_partial_floats(C.byref(sc)), **opts)
        dgrad = torch.empty(lib.mne_decoder_param_floats(C.byref(sc)), **opts)
        _lib.check(lib.mne_decoder_wgrad(C.byref(sc), _lib.ptr(tape), _lib.ptr(ray_tiles), R, S, _lib.ptr(partials),
                                         _lib.ptr(dgrad), info.get("wgrad_impl", 0), st), "mne_decoder_wgrad")
        w_sdf0, w_sdf1, w_col0, w_col1 = dec_w
        n0, n1, n2 = w_col0.numel(), w_col1.numel(), w_sdf0.numel()
        return (None, None, None, d_o if ctx.needs_input_grad[3] else None, d_d if ctx.needs_input_grad[4] else None,
                None, None, None, None, g_table,
                dgrad[n0 + n1:n0 + n1 + n2].view_as(w_sdf0), dgrad[n0 + n1 + n2:].view_as(w_sdf1),
                dgrad[:n0].view_as(w_col0), dgrad[n0:n0 + n1].view_as(w_col1))


@torch.no_grad()
def hash_render_maps(info, grid_cfg, tables, rays_o, rays_d, target_d, u, seed_offset, table, dec_w):
    """No-grad rendering with exact early ray termination for the hash / dense grid model (the counterpart of render_maps)."""
    lib = _lib.load()
    tgt_d = _f32c(target_d.reshape(-1), "target_d") if target_d is not None else None
    f = _hash_forward(lib, info, grid_cfg, tables, _f32c(rays_o.detach(), "rays_o"), _f32c(rays_d.detach(), "rays_d"), None, tgt_d,
                      _f32c(u, "u"), seed_offset, table, dec_w, early=True, want_losses=False)
    return f["rgb"], f["depth"], f["disp"], f["acc"], f["var"]


@torch.no_grad()
def hash_query_points(info, grid_cfg, table, dec_w, pts, want_raw=True, want_geo=False, want_feat=False):
    """Forward-only point query of the hash / dense grid model: raw [N,4], geo [N,15], feat [N, n_levels*2]."""
    lib = _lib.load()
    flat = _f32c(pts.reshape(-1, 3).detach(), "pts")
    n = flat.shape[0]
    opts = dict(device=flat.device, dtype=torch.float32)
    st = _lib.stream_for(flat)
    sc = _hash_scene(info, dec_w)
    feats = torch.zeros(n, 64, **opts)
    _lib.check(lib.mne_grid_encode_box(C.byref(grid_cfg), C.byref(sc), n, _lib.ptr(flat), _lib.ptr(table.detach()),
                                       _lib.ptr(feats), 64, st), "mne_grid_encode_box")
    raw = geo = None
    if want_raw or want_geo:
        packed = torch.empty(lib.mne_packed_decoder_floats(C.byref(sc)), **opts)
        _lib.check(lib.mne_pack_decoder(C.byref(sc), _lib.ptr(packed), st), "mne_pack_decoder")
        raw = torch.empty(n, 4, **opts)
        geo = torch.empty(n, info["geo_feat_dim"], **opts) if want_geo else None
        _lib.check(lib.mne_query_features(C.byref(sc), n, _lib.ptr(flat), _lib.ptr(feats), _lib.ptr(packed), _lib.ptr(raw),
                                          _lib.ptr(geo), st), "mne_query_features")
    return raw, geo, (feats[:, :grid_cfg.n_levels * grid_cfg.n_features] if want_feat else None)


# --------------------------------------------------------------------------------------------------
# N2: the pose-alignment loop as device work (csrc/pose.hip)
# --------------------------------------------------------------------------------------------------
def _rodrigues_host(rot):
    """axis-angle [3] -> 3x3, the formula of optimization/utils.py:161-177 (host-side probe only)."""
    th = torch.sqrt((rot * rot).sum() + 1e-24)
    o = rot / th
    K = torch.zeros(3, 3, dtype=rot.dtype)
    K[0, 1], K[0, 2], K[1, 0], K[1, 2], K[2, 0], K[2, 1] = -o[2], o[1], o[2], -o[0], -o[1], o[0]
    return torch.eye(3, dtype=rot.dtype) + torch.sin(th) * K + (1.0 - torch.cos(th)) * (K @ K)


def _quaternion_host(q):
    """quaternion (real part first) [4] -> 3x3, pytorch3d.transforms.quaternion_to_matrix (host-side probe only)."""
    r, i, j, k = q
    s = 2.0 / (q * q).sum()
    return torch.stack([1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r),
                        s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r),
                        s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j)]).reshape(3, 3)


def probe_axis_angle(matrix_from_tensor, rot0, trans0):
    """Does the host's ``matrix_from_tensor(rot, trans)`` have the form  R = Rot(rot) @ R_base,  t = trans  with Rot =
    Rodrigues' formula (3 parameters: the reference's ``rot_rep: 'axis_angle'``) or the normalising quaternion map (4
    parameters, real part first: ``rot_rep: 'quat'``)?  Returns R_base (3x3 float32 on the CPU; identity for the
    reference) or None (anything else): the caller then keeps the host's own autograd loop.  Two evaluations of the host
    function, once per alignment (not per iteration)."""
    n = rot0.shape[-1]
    if n not in (3, 4) or trans0.shape[-1] != 3:
        return None
    rot_of = _rodrigues_host if n == 3 else _quaternion_host
    with torch.no_grad():
        r0, t0 = rot0.detach().reshape(1, n).float().cpu(), trans0.detach().reshape(1, 3).float().cpu()
        if n == 4 and float((r0 * r0).sum()) < 1e-12:
            return None
        M0 = matrix_from_tensor(r0.to(rot0.device), t0.to(rot0.device)).detach().float().cpu().reshape(4, 4)
        base = rot_of(r0[0]).T @ M0[:3, :3]
        r1 = r0 + torch.tensor([[0.011, -0.017, 0.013, 0.007][:n]])
        t1 = t0 + torch.tensor([[0.02, -0.01, 0.03]])
        M1 = matrix_from_tensor(r1.to(rot0.device), t1.to(rot0.device)).detach().float().cpu().reshape(4, 4)
        ok = (torch.allclose(rot_of(r1[0]) @ base, M1[:3, :3], atol=2e-6) and torch.allclose(M1[:3, 3], t1[0], atol=1e-7)
              and torch.allclose(base @ base.T, torch.eye(3), atol=1e-5))
    return base.contiguous() if ok else None


class PoseAlignment:
    """Scratch + state of one alignment (``n`` rays): iterate with ``step(u)``; ``best()`` returns (best c2w 4x4, loss)."""

    def __init__(self, model, dirs, want_rgb, want_depth, rot0, trans0, r_base, lr_rot, lr_trans, betas, eps, w_rgb, w_depth):
        lib = self.lib = _lib.load()
        dev = dirs.device
        self.model, self.n = model, dirs.shape[0]
        n = self.n
        f = dict(device=dev, dtype=torch.float32)
        self.dirs = _f32c(dirs, "dirs")
        self.want_rgb, self.want_depth = _f32c(want_rgb, "rgb"), _f32c(want_depth.reshape(-1), "depth")
        self.w = (float(w_rgb), float(w_depth))
        cfg = model.config
        if not cfg["training"].get("n_samples"):
            raise KeyError("n_samples")
        self.info = model._info()
        self.rc = self.info["render_cfg"]
        self.S = lib.mne_num_samples(C.byref(self.rc), 0)
        S = self.S
        self.tables = linspace_tables(cfg, False, dev)
        self.dec_w = [w.detach() for w in model.decoder.hip_weights()]
        # hash / dense grid model (HashJointEncoding): the grid's features are caller-supplied rows of the *_features entry
        # points, its share of the ray gradients comes from mne_hash_ray_grad -- the loop is nine launches instead of six,
        # still no autograd graph, no allocation, no host sync (mp_slam/mapper.py:388-408 on the hash wiring)
        self.grid = getattr(model, "embed_fn", None)
        if self.grid is not None:
            self.planes = []
            self.scene = _hash_scene(self.info, self.dec_w)
            self.table = self.grid.params.detach()
            self.feats = torch.zeros(n * S, 64, **f)           # the grid fills the first n_levels * 2 columns of the 64-wide slot
        else:
            self.planes = [p.detach() for p in model._flat_planes()]
            self.scene = scene_struct(self.info, self.planes, self.dec_w)
        self.packed = torch.empty(lib.mne_packed_decoder_floats(C.byref(self.scene)), **f)
        n_rot = rot0.shape[-1]
        self.rot, self.trans = rot0.detach().reshape(n_rot).to(**f).clone(), trans0.detach().reshape(3).to(**f).clone()
        self.m, self.v = torch.zeros(n_rot + 3, **f), torch.zeros(n_rot + 3, **f)
        self.step_count = torch.zeros(1, device=dev, dtype=torch.int32)
        self.c2w, self.best_c2w = torch.zeros(12, **f), torch.zeros(12, **f)
        self.best_loss, self.last_loss = torch.full((1,), float("inf"), **f), torch.zeros(1, **f)
        self.rays_o, self.rays_d = torch.empty(n, 3, **f), torch.empty(n, 3, **f)
        self.z, self.raw = torch.empty(n, S, **f), torch.empty(n, S, 4, **f)
        self.counts = torch.empty(_lib.N_COUNT, device=dev, dtype=torch.int32)
        self.rgb, self.depth = torch.empty(n, 3, **f), torch.empty(n, **f)
        self.aux = torch.empty(3, n, **f)
        self.d_rgb, self.d_depth = torch.empty(n, 3, **f), torch.empty(n, **f)
        self.partials = torch.empty((n + 255) // 256, **f)
        self.d_o, self.d_d = torch.empty(n, 3, **f), torch.empty(n, 3, **f)
        self.tape = torch.empty(n * S, lib.mne_tape_row_floats(C.byref(self.scene)), **f)
        if self.grid is not None:
            self.tape[:, :64].zero_()                          # feature columns the grid does not fill must read as zero
        self.tape_rows = torch.zeros(1, device=dev, dtype=torch.int32)
        self.ray_tiles = torch.empty(n, device=dev, dtype=torch.int32)
        self.ws_bytes = lib.mne_render_workspace_bytes(n, S)
        self.ws = torch.empty(self.ws_bytes, device=dev, dtype=torch.uint8)
        ps = self.ps = _lib.PoseState()
        ps.rot, ps.trans, ps.m, ps.v, ps.step = (t.data_ptr() for t in (self.rot, self.trans, self.m, self.v, self.step_count))
        ps.c2w, ps.best_loss, ps.best_c2w, ps.last_loss = (t.data_ptr() for t in (self.c2w, self.best_loss, self.best_c2w, self.last_loss))
        ps.n_rot = n_rot
        for k, val in enumerate(r_base.reshape(-1).tolist()):
            ps.r_base[k] = val
        ps.lr_rot, ps.lr_trans, (ps.beta1, ps.beta2), ps.eps = float(lr_rot), float(lr_trans), map(float, betas), float(eps)
        st = _lib.stream_for(self.dirs)
        _lib.check(lib.mne_pack_decoder(C.byref(self.scene), _lib.ptr(self.packed), st), "mne_pack_decoder")   # weights are fixed here
        # The best pose starts as the START pose (mp_slam/mapper.py:386: best_target_c2w_est = target_c2w_initial.clone()):
        # with zero iterations, or when every loss is NaN, best() then returns a proper transform, never a singular matrix.
        _lib.check(lib.mne_pose_rays(C.byref(self.ps), n, _lib.ptr(self.dirs), _lib.ptr(self.rays_o), _lib.ptr(self.rays_d), st),
                   "mne_pose_rays")
        self.best_c2w.copy_(self.c2w)

    def step(self, u=None, seed_offset=(0, 0)):
        """One iteration: rays from the current parameters, render, loss, ray gradients, Adam.  ``u`` [n, S] = the host's
        jitter draw (reference order) or None (device generator, ``seed_offset``)."""
        lib, P, n, S = self.lib, _lib.ptr, self.n, self.S
        st = _lib.stream_for(self.dirs)
        sc, rc = C.byref(self.scene), C.byref(self.rc)
        _lib.check(lib.mne_pose_rays(C.byref(self.ps), n, P(self.dirs), P(self.rays_o), P(self.rays_d), st), "mne_pose_rays")
        u_c = _f32c(u, "u") if u is not None else None
        _lib.check(lib.mne_sample_z(rc, n, None, P(u_c), P(self.tables), seed_offset[0], seed_offset[1], P(self.z), P(self.counts),
                                    None, None, st), "mne_sample_z")
        if self.grid is not None:
            gc = C.byref(self.grid.cfg)
            _lib.check(lib.mne_hash_features(gc, sc, n, S, P(self.rays_o), P(self.rays_d), P(self.z), P(self.table), P(self.feats), st),
                       "mne_hash_features")
            _lib.check(lib.mne_render_forward_features(sc, rc, n, S, P(self.rays_o), P(self.rays_d), None, None, P(self.z),
                                                       P(self.packed), P(self.feats), P(self.rgb), P(self.depth), P(self.aux[0]),
                                                       P(self.aux[1]), P(self.aux[2]), P(self.raw), None, None, 0, st),
                       "mne_render_forward_features")
        else:
            _lib.check(lib.mne_render_forward(sc, rc, n, S, P(self.rays_o), P(self.rays_d), None, None, P(self.z), P(self.packed),
                                              P(self.rgb), P(self.depth), P(self.aux[0]), P(self.aux[1]), P(self.aux[2]), P(self.raw),
                                              None, None, 0, st), "mne_render_forward")
        _lib.check(lib.mne_pose_loss(n, P(self.rgb), P(self.depth), P(self.want_rgb), P(self.want_depth), self.w[0], self.w[1],
                                     P(self.d_rgb), P(self.d_depth), P(self.partials), st), "mne_pose_loss")
        self.d_o.zero_()
        self.d_d.zero_()
        if self.grid is not None:
            _lib.check(lib.mne_hash_gather(gc, sc, n, S, P(self.rays_o), P(self.rays_d), P(self.z), None, P(self.table), P(self.tape), st),
                       "mne_hash_gather")
            _lib.check(lib.mne_render_backward_features(sc, rc, n, S, P(self.rays_o), P(self.rays_d), None, None, P(self.z), None,
                                                        P(self.packed), P(self.raw), None, P(self.d_rgb), P(self.d_depth), P(self.tape),
                                                        n * S, P(self.tape_rows), P(self.ray_tiles), P(self.d_o), P(self.d_d), P(self.ws),
                                                        self.ws_bytes, st), "mne_render_backward_features")
            _lib.check(lib.mne_hash_ray_grad(gc, sc, n, S, P(self.rays_o), P(self.rays_d), P(self.z), P(self.table), P(self.tape),
                                             P(self.ray_tiles), P(self.d_o), P(self.d_d), st), "mne_hash_ray_grad")     # + the grid's share
        else:
            _lib.check(lib.mne_render_backward(sc, rc, n, S, P(self.rays_o), P(self.rays_d), None, None, P(self.z), None,
                                               P(self.packed), P(self.raw), None, P(self.d_rgb), P(self.d_depth), P(self.tape), n * S,
                                               P(self.tape_rows), P(self.ray_tiles), P(self.d_o), P(self.d_d), P(self.ws),
                                               self.ws_bytes, st), "mne_render_backward")
        _lib.check(lib.mne_pose_update(C.byref(self.ps), n, P(self.dirs), P(self.d_o), P(self.d_d), P(self.partials), st),
                   "mne_pose_update")

    def best(self):
        T = torch.eye(4, device=self.dirs.device)
        T[:3, :4] = self.best_c2w.reshape(3, 4)
        return T, self.best_loss[0]
