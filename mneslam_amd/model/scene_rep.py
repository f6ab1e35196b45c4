"""JointEncoding -- host-side mirror of the reference scene model (model/scene_rep.py:15-611).

Same constructor, attributes (``all_planes``, ``bound``, ``bounding_box``, ``decoder``,
``embedpos_fn``, ``color_net``/``sdf_net`` aliases, ``config``, ``ray_batch_size``), state_dict keys
and method names, so the reference's tracking / keyframe / multi-agent code can hold one of these
instead.  The per-sample work -- z sampling, tri-plane lookup, OneBlob, MLP,
compositing, losses and their backward -- runs in the HIP kernels behind include/mneslam_hip.h.

Differences that are deliberate and invisible to callers:
* planes are allocated ``torch.channels_last`` (physical [H][W][C]: one bilinear corner = one 128-B
  line); their logical shape stays [1,C,H,W], they pickle/``torch.save`` like any tensor;
* ``z_vals``/``raw``/``disp_map``/``acc_map``/``depth_var`` come back without a grad_fn (the
  reference never differentiates through them outside ``forward``).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import hip_path
from .decoder import ColorSDFNet, ColorSDFNet_v2
from .encodings import get_encoder
from .utils import batchify, get_rays


class JointEncoding(nn.Module):
    def __init__(self, config, bound_box, ray_batch_size=4096):
        super().__init__()
        self.config = config
        self.bounding_box = bound_box
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.load_bound(config)
        self.get_encoding(config)
        self.get_decoder(config)
        self.ray_batch_size = ray_batch_size
        # 'torch_cpu': jitter drawn like the reference, torch.rand on the CPU generator then copied
        # (scene_rep.py:381; reproducible across devices).  'device': Philox inside the kernel.
        self.jitter_rng = "torch_cpu"
        self._philox_offset = 0
        self.wgrad_impl = 0
        if config["training"]["n_importance"] > 0:
            raise NotImplementedError("training.n_importance > 0 is dead code in every shipped config "
                                      "(SURVEY.md A22) and is not provided")
        if config["training"]["white_bkgd"]:
            raise NotImplementedError("training.white_bkgd is False in every shipped config and is not provided")

    # ------------------------------------------------------------------ construction
    def load_bound(self, cfg):
        """Extended fp32 bound, kept on the CPU like the reference (scene_rep.py:72-83)."""
        self.bound = torch.from_numpy(np.array(cfg["mapping"]["bound"]) * cfg["scale"]).float()
        bd = cfg["planes_res"]["bound_dividable"]
        self.bound[:, 1] = (((self.bound[:, 1] - self.bound[:, 0]) / bd).int() + 1) * bd + self.bound[:, 0]

    def _init_planes(self, resolutions, c_dim):
        xyz_len = self.bound[:, 1] - self.bound[:, 0]
        planes_xy, planes_xz, planes_yz = [], [], []
        # EXTENSION (multi-agent overlap exchange, mneslam_amd/dist.py): ``planes_res.lattice: true`` puts the nodes at EXACTLY
        # ``res`` apart (round(len / res) + 1 nodes per axis; the extended bound is then a whole number of cells when
        # ``bound_dividable`` is a multiple of ``res``), so that agents whose bounds start on a common lattice share its nodes.
        # The reference's int(len / res) nodes (scene_rep.py:98-100) give a spacing that depends on the agent's own extent.
        lattice = bool(self.config["planes_res"].get("lattice", False))
        for res in resolutions:
            gs = [int(round(v / res)) + 1 for v in xyz_len.tolist()] if lattice else list(map(int, (xyz_len / res).tolist()))
            gs[0], gs[2] = gs[2], gs[0]
            # same CPU draws, in the same order and logical shape, as scene_rep.py:107-109
            planes_xy.append(torch.empty([1, c_dim, *gs[1:]]).normal_(mean=0, std=0.01))
            planes_xz.append(torch.empty([1, c_dim, gs[0], gs[2]]).normal_(mean=0, std=0.01))
            planes_yz.append(torch.empty([1, c_dim, *gs[:2]]).normal_(mean=0, std=0.01))
        # EXTENSION (BASELINE configs[4], "fp16 features + fp32 accumulate"): grid.plane_dtype 'fp16' STORES the planes in half
        # precision -- the fp32 draws above rounded to nearest; no fp32 copy exists.  Lookups convert on load, everything
        # after them (interpolation, decoder, gradients, their sums, Adam's moments) stays fp32; include/mneslam_hip.h,
        # mne_scene_t.plane_f16.  Absent / 'fp32' = the reference's storage.
        dtype = {"fp32": torch.float32, "fp16": torch.float16}[self.config["grid"].get("plane_dtype", "fp32")]
        for planes in (planes_xy, planes_xz, planes_yz):
            for i, p in enumerate(planes):
                p = p.to(self.device, dtype).contiguous(memory_format=torch.channels_last)
                if p.device.type == "cpu":
                    p.share_memory_()
                planes[i] = p
        return planes_xy, planes_xz, planes_yz

    def init_all_planes(self, config):
        """reference: scene_rep.py:85-117"""
        self.coarse_planes_res = config["planes_res"]["coarse"]
        self.fine_planes_res = config["planes_res"]["fine"]
        return self._init_planes([self.coarse_planes_res, self.fine_planes_res], config["model"]["c_dim"])

    def init_all_c_planes(self, config):
        """reference: scene_rep.py:119-150"""
        self.coarse_c_planes_res = config["c_planes_res"]["coarse"]
        self.fine_c_planes_res = config["c_planes_res"]["fine"]
        return self._init_planes([self.coarse_c_planes_res, self.fine_c_planes_res], config["model"]["c_dim"])

    def get_encoding(self, config):
        """reference: scene_rep.py:152-169 (tri-planes are what is wired; the hash-grid call is
        commented out there)."""
        self.embedpos_fn, self.input_ch_pos = get_encoder(config["pos"]["enc"], n_bins=config["pos"]["n_bins"])
        self.input_ch = config["model"]["input_ch"]
        self.input_ch_pos = config["model"]["input_ch_pos"]
        self.all_planes = self.init_all_planes(config)
        if not config["grid"]["oneGrid"]:
            self.all_planes = self.all_planes + self.init_all_c_planes(config)

    def get_decoder(self, config):
        """reference: scene_rep.py:171-181"""
        cls = ColorSDFNet_v2 if config["grid"]["oneGrid"] else ColorSDFNet
        self.decoder = cls(config, input_ch=self.input_ch, input_ch_pos=self.input_ch_pos)
        self.color_net = batchify(self.decoder.color_net, None)      # == the module itself
        self.sdf_net = batchify(self.decoder.sdf_net, None)

    # ------------------------------------------------------------------ HIP plumbing
    def _info(self):
        cfg = self.config
        if cfg["model"]["c_dim"] != 32 or cfg["pos"]["n_bins"] != 16 or cfg["decoder"]["geo_feat_dim"] != 15:
            raise NotImplementedError("the HIP path is built for c_dim=32, n_bins=16, geo_feat_dim=15 "
                                      "(every shipped config)")
        bb_cpu = self._host_copy("bounding_box")
        bound = self._host_copy("bound")          # a peer's checkpoint puts it on the device (mp_slam/mapper.py:721)
        return {
            "n_planes": 6 * (1 if cfg["grid"]["oneGrid"] else 2),
            "c_dim": cfg["model"]["c_dim"], "hidden": cfg["decoder"]["hidden_dim"],
            "hidden_color": cfg["decoder"]["hidden_dim_color"], "geo_feat_dim": cfg["decoder"]["geo_feat_dim"],
            "n_bins": cfg["pos"]["n_bins"], "bb_is_f64": bb_cpu.dtype == torch.float64,
            "bound_lo": [float(v) for v in bound[:, 0]], "bound_hi": [float(v) for v in bound[:, 1]],
            "bb_lo": [float(v) for v in bb_cpu[:, 0]], "bb_hi": [float(v) for v in bb_cpu[:, 1]],
            "render_cfg": hip_path.render_cfg_struct(cfg), "wgrad_impl": self.wgrad_impl,
        }

    def _host_copy(self, attr):
        """Host copy of ``bounding_box`` / ``bound`` (device tensors in the live system): fetched once per distinct
        tensor / in-place version instead of one device synchronisation per render call."""
        t = getattr(self, attr)
        caches = self.__dict__.setdefault("_host_cache", {})
        hit = caches.get(attr)
        # the cache entry HOLDS the tensor it was made from: `is` cannot be fooled by a freed tensor's recycled id()
        # (load_foreign_model re-binds bound / bounding_box to fresh tensors on every peer checkpoint)
        if hit is None or hit[0] is not t or hit[1] != getattr(t, "_version", None):
            hit = caches[attr] = (t, getattr(t, "_version", None), torch.as_tensor(t).detach().cpu())
        return hit[2]

    def _flat_planes(self):
        planes = [p for lst in self.all_planes for p in lst]
        expect = 6 if self.config["grid"]["oneGrid"] else 12
        if len(planes) != expect:
            raise ValueError(f"all_planes must hold {expect} planes, found {len(planes)}")
        return planes

    def _render(self, rays_o, rays_d, target_rgb, target_d, u=None):
        info = self._info()
        dev = rays_o.device
        owners = self._flat_planes()
        planes = [hip_path.as_channels_last(p if p.device == dev else p.to(dev)) for p in owners]
        if any(p.dtype == torch.float16 for p in owners):
            # half-precision planes: the backward leaves the fp32 gradient sums (``grad32``) on the PARAMETERS themselves -- a plane
            # that is not channels_last or not on the render device reaches the node as a temporary copy
            info = dict(info, plane_owners=owners)
        dec_w = self.decoder.hip_weights()
        has_d = target_d is not None
        if not has_d and not self.config["training"].get("n_samples"):
            raise KeyError("n_samples")            # the reference raises the same
        tables = hip_path.linspace_tables(self.config, has_d, dev)
        seed_offset = (0, 0)
        if u is None:
            S = (self.config["training"]["n_range_d"] + self.config["training"]["n_samples_d"]) if has_d \
                else self.config["training"]["n_samples"]
            u, seed_offset = self._jitter(rays_o.shape[0], S, rays_o)
        return hip_path.RenderFunction.apply(info, tables, rays_o, rays_d, target_rgb, target_d, u, seed_offset,
                                             *planes, *dec_w)

    # ------------------------------------------------------------------ rendering API
    def render_rays(self, rays_o, rays_d, target_d=None):
        """reference: scene_rep.py:351-419"""
        rgb, depth, disp, acc, var, z_vals, raw, _ = self._render(rays_o, rays_d, None, target_d)
        return {"rgb": rgb, "depth": depth, "disp_map": disp, "acc_map": acc, "depth_var": var,
                "z_vals": z_vals, "raw": raw}

    def forward(self, rays_o, rays_d, target_rgb, target_d, global_step=0):
        """reference: scene_rep.py:549-611.  Training mode returns the loss dict (all seven losses
        and psnr are computed on every call); eval mode returns render_rays' dict."""
        if not self.training:
            return self.render_rays(rays_o, rays_d, target_d=target_d)
        rgb, depth, _, _, _, _, _, losses = self._render(rays_o, rays_d, target_rgb, target_d)
        L = hip_path._lib
        return {"rgb": rgb, "depth": depth, "rgb_loss": losses[L.L_RGB], "depth_loss": losses[L.L_DEPTH],
                "co_sdf_loss": losses[L.L_CO_SDF], "co_fs_loss": losses[L.L_CO_FS],
                "e_fs_loss": losses[L.L_E_FS], "e_center_loss": losses[L.L_E_CENTER],
                "e_tail_loss": losses[L.L_E_TAIL], "psnr": losses[L.L_PSNR:L.L_PSNR + 1]}

    def _jitter(self, R, S, like):
        """(u, seed_offset) for R x S samples: the reference's CPU draw, or the device generator's counter."""
        if self.config["training"]["perturb"] <= 0.0:
            return None, (0, 0)
        if self.jitter_rng == "torch_cpu":
            return torch.rand(R, S).to(like), (0, 0)
        so = (int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF, self._philox_offset)
        self._philox_offset += (R * S + 3) // 4
        return None, so

    def render_maps(self, rays_o, rays_d, target_d=None, u=None, stats=None):
        """No-grad fast path of ``render_rays`` for any number of rays at once: {rgb, depth, disp_map, acc_map,
        depth_var} (no z_vals / raw: with exact early ray termination the samples behind a ray's surface are never
        decoded).  Same sampling, same maps as ``render_rays``."""
        dev = rays_o.device
        has_d = target_d is not None
        if not has_d and not self.config["training"].get("n_samples"):
            raise KeyError("n_samples")            # the reference raises the same
        tr = self.config["training"]
        S = (tr["n_range_d"] + tr["n_samples_d"]) if has_d else tr["n_samples"]
        seed_offset = (0, 0)
        if u is None:
            u, seed_offset = self._jitter(rays_o.shape[0], S, rays_o)
        planes = [p if p.device == dev else p.to(dev) for p in self._flat_planes()]
        rgb, depth, disp, acc, var = hip_path.render_maps(self._info(), hip_path.linspace_tables(self.config, has_d, dev), rays_o,
                                                          rays_d, target_d, u, seed_offset, planes, self.decoder.hip_weights(),
                                                          stats=stats)
        return {"rgb": rgb, "depth": depth, "disp_map": disp, "acc_map": acc, "depth_var": var}

    def render_img(self, c2w, device, gt_depth=None, stats=None):
        """reference: scene_rep.py:422-473 (depth is returned as float64, A20).  The reference walks the image in
        ``ray_batch_size`` = 4096-ray chunks through ``render_rays``; here the whole frame (816 k rays on Replica) is
        ONE no-grad launch sequence (``render_chunk_rays`` bounds the scratch for very large frames).  With
        ``jitter_rng == "torch_cpu"`` the jitter is drawn chunk by chunk exactly like the reference does."""
        with torch.no_grad():
            cam = self.config["cam"]
            H, W = cam["H"] - 2 * cam["crop_edge"], cam["W"] - 2 * cam["crop_edge"]
            rays_o, rays_d = get_rays(H, W, cam["fx"], cam["fy"], cam["cx"], cam["cy"], c2w.to(device), device)
            rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
            if gt_depth is not None:
                gt_depth = gt_depth.reshape(-1).unsqueeze(1).to(device)
            n = rays_d.shape[0]
            tr = self.config["training"]
            S = (tr["n_range_d"] + tr["n_samples_d"]) if gt_depth is not None else tr.get("n_samples")
            # Ray order inside the launch: pixels along a Z-order curve instead of row by row, so that the 8 rays a workgroup
            # decodes at a time are a 4 x 2 pixel patch and the 256 rays of an XCD's workgroups a 16 x 16 one -- their samples
            # fall into the same 1-2 cm plane cells, whose corner rows then come from L1 / L2 instead of from HBM (the frame
            # kernels are bound by exactly those fetches: DESIGN.md 3.6).  Per-ray results do not depend on the order; the
            # device jitter is keyed by the position in the launch, so a pixel gets a different (equally distributed) draw.
            # Not with the reference's CPU draws (torch_cpu): those are consumed in image order, chunk by chunk.
            order = None
            if getattr(self, "render_patch_order", True) and not (tr["perturb"] > 0.0 and self.jitter_rng == "torch_cpu" and S):
                order = self._pixel_order(H, W, device)
                rays_o, rays_d = rays_o[order], rays_d[order]
                if gt_depth is not None:
                    gt_depth = gt_depth[order]
            depths, colors = [], []
            step = getattr(self, "render_chunk_rays", 1 << 20)
            for i in range(0, n, step):
                sl = slice(i, min(i + step, n))
                u = None
                if tr["perturb"] > 0.0 and self.jitter_rng == "torch_cpu" and S:
                    # the reference's draws: one torch.rand(chunk, S) per 4096-ray chunk, in order
                    u = torch.cat([torch.rand(min(self.ray_batch_size, sl.stop - j), S)
                                   for j in range(sl.start, sl.stop, self.ray_batch_size)], 0).to(device)
                ret = self.render_maps(rays_o[sl], rays_d[sl], None if gt_depth is None else gt_depth[sl], u=u,
                                       **({"stats": stats} if stats is not None else {}))
                depths.append(ret["depth"].double())
                colors.append(ret["rgb"])
            depth, color = torch.cat(depths, 0), torch.cat(colors, 0)
            if order is not None:
                depth, color = torch.empty_like(depth).index_copy_(0, order, depth), torch.empty_like(color).index_copy_(0, order, color)
            return depth.reshape(H, W), color.reshape(H, W, 3)

    def _pixel_order(self, H, W, device):
        """Pixel indices (row-major y * W + x) of an H x W frame sorted along the Z-order (Morton) curve: any aligned run of
        8 / 64 / 256 consecutive entries is a 4 x 2 / 8 x 8 / 16 x 16 pixel block (cut at the frame's edges).  Cached."""
        key = (H, W, str(device))
        cache = self.__dict__.setdefault("_pixel_order_cache", {})
        if key not in cache:
            y, x = torch.meshgrid(torch.arange(H, dtype=torch.int64), torch.arange(W, dtype=torch.int64), indexing="ij")
            code = torch.zeros(H, W, dtype=torch.int64)
            for b in range(16):
                code |= ((x >> b) & 1) << (2 * b)
                code |= ((y >> b) & 1) << (2 * b + 1)
            cache[key] = torch.argsort(code.reshape(-1), stable=True).to(device)
        return cache[key]

    # ------------------------------------------------------------------ point queries (forward only)
    def _query(self, pts, **kw):
        dev = pts.device
        planes = [p if p.device == dev else p.to(dev) for p in self._flat_planes()]
        return hip_path.query_points(self._info(), planes, self.decoder.hip_weights(), pts, **kw)

    def query_color_sdf(self, query_points):
        """reference: scene_rep.py:273-301 -> raw [N,4] (flattened like the reference)."""
        return self._query(query_points)[0]

    def query_sdf(self, query_points, return_geo=False, embed=False):
        """reference: scene_rep.py:232-268"""
        lead = list(query_points.shape[:-1])
        if embed:
            feat = self._query(query_points, want_raw=False, want_feat=True)[2]
            return feat.reshape(lead + [feat.shape[-1]])
        raw, geo, _ = self._query(query_points, want_geo=return_geo)
        sdf = raw[..., 3].reshape(lead)
        if not return_geo:
            return sdf
        return sdf, geo.reshape(lead + [geo.shape[-1]])

    def query_color(self, query_points):
        """reference: scene_rep.py:270-271"""
        return torch.sigmoid(self.query_color_sdf(query_points)[..., :3])

    def run_network(self, inputs):
        """reference: scene_rep.py:303-317"""
        flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
        out = self.query_color_sdf(flat)
        return torch.reshape(out, list(inputs.shape[:-1]) + [out.shape[-1]])

    def run_network_flat(self, inputs_flat):
        """reference: scene_rep.py:319-331"""
        return self.query_color_sdf(inputs_flat)

    def sample_plane_feature(self, p_nor, planes_xy, planes_xz, planes_yz):
        """reference: scene_rep.py:28-53 (forward only; p_nor already normalised to [-1,1])."""
        info = self._info()
        info["n_planes"] = 6
        planes = [p for lst in (planes_xy, planes_xz, planes_yz) for p in lst]
        return hip_path.query_points(info, planes, self.decoder.hip_weights(), p_nor, want_raw=False,
                                     want_feat=True, normalised=True)[2]

    # ------------------------------------------------------------------ compositing helpers (cold API)
    def sdf2weights(self, sdf, z_vals, args=None):
        """reference: scene_rep.py:183-203 (kept for API completeness; the hot path composites in-kernel)."""
        args = args or self.config
        tr = args["training"]["trunc"]
        w = torch.sigmoid(sdf / tr) * torch.sigmoid(-sdf / tr)
        crossing = (sdf[:, 1:] * sdf[:, :-1] < 0.0).to(sdf.dtype)
        z_min = torch.gather(z_vals, 1, torch.argmax(crossing, dim=1, keepdim=True))
        w = w * (z_vals < z_min + args["data"]["sc_factor"] * tr).to(sdf.dtype)
        return w / (w.sum(-1, keepdim=True) + 1e-8)

    def raw2outputs(self, raw, z_vals, white_bkgd=False):
        """reference: scene_rep.py:205-230"""
        rgb = torch.sigmoid(raw[..., :3])
        w = self.sdf2weights(raw[..., 3], z_vals, args=self.config)
        rgb_map = torch.sum(w[..., None] * rgb, -2)
        depth_map = torch.sum(w * z_vals, -1)
        depth_var = torch.sum(w * torch.square(z_vals - depth_map.unsqueeze(-1)), dim=-1)
        acc_map = torch.sum(w, -1)
        disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)
        if white_bkgd:
            rgb_map = rgb_map + (1.0 - acc_map[..., None])
        return rgb_map, disp_map, acc_map, w, depth_map, depth_var

    def render_surface_color(self, rays_o, normal):
        """reference: scene_rep.py:333-349"""
        trunc = self.config["training"]["trunc"]
        z_vals = torch.linspace(-trunc, trunc, steps=self.config["training"]["n_range_d"]).to(rays_o)
        z_vals = z_vals.repeat(rays_o.shape[0], 1)
        pts = rays_o[..., :] + normal[..., None, :] * z_vals[..., :, None]
        raw = self.run_network(pts)
        return self.raw2outputs(raw, z_vals, self.config["training"]["white_bkgd"])[0]
