"""CPU restatement of the reference scene model (TEST INFRASTRUCTURE ONLY).

Follows dtc111111/MNESLAM:
  model/scene_rep.py   (JointEncoding: planes, lookup, compositing, losses)
  model/decoder.py     (bias-free ReLU MLPs)
  model/utils.py       (normalisation, masks, Co-SLAM sdf loss, psnr)
Plain torch ops on CPU, differentiable through autograd so that gradients of
planes, decoder weights and rays can be used as references for the HIP path.

Written functionally (state = ``OracleScene``) rather than as an nn.Module so that
the host-side mirror under mneslam_amd/ and this checker share no code.
"""
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .oneblob import oneblob


# --------------------------------------------------------------------------------------
# configuration view
# --------------------------------------------------------------------------------------
@dataclass
class PathConfig:
    """The hot-path keys of the reference YAML ("Config / flags")."""
    near: float
    far: float
    depth_trunc: float
    n_samples: Optional[int]      # training.n_samples (absent in ScanNet configs)
    n_samples_d: int
    range_d: float
    n_range_d: int
    n_importance: int
    perturb: float
    trunc: float                  # training.trunc
    white_bkgd: bool
    sc_factor: float              # data.sc_factor
    truncation: float             # model.truncation
    c_dim: int
    one_grid: bool
    n_bins: int
    geo_feat_dim: int
    hidden_dim: int
    num_layers: int
    hidden_dim_color: int
    num_layers_color: int
    input_ch: int
    input_ch_pos: int
    coarse: float
    fine: float
    bound_dividable: float
    c_coarse: float
    c_fine: float
    scale: float
    bound: Sequence[Sequence[float]]

    @staticmethod
    def from_dict(cfg) -> "PathConfig":
        tr, cam, dec = cfg["training"], cfg["cam"], cfg["decoder"]
        return PathConfig(
            near=cam["near"], far=cam["far"], depth_trunc=cam["depth_trunc"],
            n_samples=tr.get("n_samples"), n_samples_d=tr["n_samples_d"], range_d=tr["range_d"],
            n_range_d=tr["n_range_d"], n_importance=tr["n_importance"], perturb=tr["perturb"],
            trunc=tr["trunc"], white_bkgd=tr["white_bkgd"], sc_factor=cfg["data"]["sc_factor"],
            truncation=cfg["model"]["truncation"], c_dim=cfg["model"]["c_dim"],
            one_grid=bool(cfg["grid"]["oneGrid"]), n_bins=cfg["pos"]["n_bins"],
            geo_feat_dim=dec["geo_feat_dim"], hidden_dim=dec["hidden_dim"], num_layers=dec["num_layers"],
            hidden_dim_color=dec["hidden_dim_color"], num_layers_color=dec["num_layers_color"],
            input_ch=cfg["model"]["input_ch"], input_ch_pos=cfg["model"]["input_ch_pos"],
            coarse=cfg["planes_res"]["coarse"], fine=cfg["planes_res"]["fine"],
            bound_dividable=cfg["planes_res"]["bound_dividable"],
            c_coarse=cfg["c_planes_res"]["coarse"], c_fine=cfg["c_planes_res"]["fine"],
            scale=cfg["scale"], bound=cfg["mapping"]["bound"])


# --------------------------------------------------------------------------------------
# bounds and plane geometry
# --------------------------------------------------------------------------------------
def extended_bound(pc: PathConfig) -> torch.Tensor:
    """fp32 [3,2]; upper bound rounded up to a multiple of bound_dividable.
    model/scene_rep.py:80-83 (same torch ops: float-sensitive)."""
    b = torch.from_numpy(np.array(pc.bound) * pc.scale).float()
    b[:, 1] = (((b[:, 1] - b[:, 0]) / pc.bound_dividable).int() + 1) * pc.bound_dividable + b[:, 0]
    return b


def plane_shapes(bound: torch.Tensor, res: float) -> List[Tuple[int, int]]:
    """(H, W) of the xy, xz, yz planes at one resolution.  model/scene_rep.py:96,105-109:
    grid_shape = int(len/res) per axis, then axes 0<->2 swapped, so xy=[Y,X], xz=[Z,X], yz=[Z,Y]."""
    xyz_len = bound[:, 1] - bound[:, 0]
    nx, ny, nz = [int(v) for v in (xyz_len / res).tolist()]
    return [(ny, nx), (nz, nx), (nz, ny)]


def make_planes(bound, resolutions, c_dim, generator=None):
    """Three lists (xy, xz, yz), one plane per resolution, N(0, 0.01^2) init.
    model/scene_rep.py:98-117 (draw order: per resolution xy, xz, yz)."""
    xy, xz, yz = [], [], []
    for res in resolutions:
        for lst, (h, w) in zip((xy, xz, yz), plane_shapes(bound, res)):
            lst.append(torch.empty(1, c_dim, h, w).normal_(mean=0, std=0.01, generator=generator))
    return xy, xz, yz


# --------------------------------------------------------------------------------------
# coordinates and bilinear lookup
# --------------------------------------------------------------------------------------
def normalize_points(p: torch.Tensor, bound: torch.Tensor) -> torch.Tensor:
    """[-1,1] normalisation with the EXTENDED bound.  model/utils.py:27-41."""
    p = p.reshape(-1, 3)
    cols = [((p[:, k] - bound[k, 0]) / (bound[k, 1] - bound[k, 0])) * 2 - 1.0 for k in range(3)]
    return torch.stack(cols, dim=-1)


def unnormalize_index(g: torch.Tensor, size: int) -> torch.Tensor:
    """ATen grid_sampler source index, align_corners=True, padding_mode='border'
    (aten/native/cuda/GridSampler.cuh: ((g+1)/2)*(size-1), clipped to [0,size-1]).
    The reference reaches it through F.grid_sample at model/scene_rep.py:43-47."""
    idx = ((g + 1.0) / 2.0) * float(size - 1)
    return torch.clamp(idx, 0.0, float(size - 1))


def bilinear_corners(gx, gy, h, w):
    """Integer NW corner (ix0, iy0) and the four weights (nw, ne, sw, se) -- the
    "bit-exact integer indices" of this path."""
    ix, iy = unnormalize_index(gx, w), unnormalize_index(gy, h)
    ix0, iy0 = torch.floor(ix), torch.floor(iy)
    ix1, iy1 = ix0 + 1.0, iy0 + 1.0
    nw = (ix1 - ix) * (iy1 - iy)
    ne = (ix - ix0) * (iy1 - iy)
    sw = (ix1 - ix) * (iy - iy0)
    se = (ix - ix0) * (iy - iy0)
    return ix0.long(), iy0.long(), (nw, ne, sw, se)


def sample_plane_explicit(plane, gx, gy):
    """Bilinear lookup of one [1,C,H,W] plane at normalised (gx -> W, gy -> H).
    Out-of-range corners (ix0+1 == W etc.) are skipped, as ATen does."""
    _, c, h, w = plane.shape
    ix0, iy0, (nw, ne, sw, se) = bilinear_corners(gx, gy, h, w)
    flat = plane[0].permute(1, 2, 0).reshape(h * w, c)       # [H*W, C]
    out = torch.zeros(gx.shape[0], c, dtype=plane.dtype)
    for dx, dy, wgt in ((0, 0, nw), (1, 0, ne), (0, 1, sw), (1, 1, se)):
        x, y = ix0 + dx, iy0 + dy
        ok = (x >= 0) & (x < w) & (y >= 0) & (y < h)
        lin = (y.clamp(0, h - 1) * w + x.clamp(0, w - 1))
        out = out + flat[lin] * (wgt * ok.to(wgt.dtype))[:, None]
    return out


def sample_plane_grid_sample(plane, gx, gy):
    """Same lookup through F.grid_sample exactly as model/scene_rep.py:39-48 calls it."""
    grid = torch.stack([gx, gy], dim=-1)[None, :, None, :]
    out = F.grid_sample(plane, grid, padding_mode="border", align_corners=True, mode="bilinear")
    return out[0, :, :, 0].transpose(0, 1)


def sample_plane_feature(p_nor, planes_xy, planes_xz, planes_yz, impl="explicit"):
    """Sum over the three orientations, concat over levels.  model/scene_rep.py:28-53."""
    fn = sample_plane_explicit if impl == "explicit" else sample_plane_grid_sample
    x, y, z = p_nor[:, 0], p_nor[:, 1], p_nor[:, 2]
    levels = []
    for pxy, pxz, pyz in zip(planes_xy, planes_xz, planes_yz):
        levels.append(fn(pxy, x, y) + fn(pxz, x, z) + fn(pyz, y, z))
    return torch.cat(levels, dim=-1)


# --------------------------------------------------------------------------------------
# decoder
# --------------------------------------------------------------------------------------
def make_decoder_weights(pc: PathConfig, generator=None):
    """Weight lists (sdf_net, color_net), nn.Linear default init (kaiming-uniform a=sqrt(5)
    == U(-1/sqrt(in), 1/sqrt(in))), bias-free.  model/decoder.py:39-55, :92-108, :110-160."""
    def linear(out_f, in_f):
        bound = 1.0 / np.sqrt(in_f)
        return (torch.rand(out_f, in_f, generator=generator) * 2 - 1) * bound

    in_sdf = pc.input_ch + pc.input_ch_pos
    sdf = []
    for l in range(pc.num_layers):
        i = in_sdf if l == 0 else pc.hidden_dim
        o = 1 + pc.geo_feat_dim if l == pc.num_layers - 1 else pc.hidden_dim
        sdf.append(linear(o, i))
    in_col = (pc.input_ch_pos if pc.one_grid else pc.input_ch + pc.input_ch_pos) + pc.geo_feat_dim
    col = []
    for l in range(pc.num_layers_color):
        i = in_col if l == 0 else pc.hidden_dim_color
        o = 3 if l == pc.num_layers_color - 1 else pc.hidden_dim_color
        col.append(linear(o, i))
    return sdf, col


def mlp(weights, x):
    for k, w in enumerate(weights):
        x = x @ w.t()
        if k != len(weights) - 1:
            x = torch.relu(x)
    return x


def decode(sdf_w, col_w, feat, pos, cfeat=None):
    """raw = [rgb_raw(3), sdf(1)].  model/decoder.py:128-141 (colour planes) / :161-175 (oneGrid)."""
    h = mlp(sdf_w, torch.cat([feat, pos], dim=-1))
    sdf, geo = h[..., :1], h[..., 1:]
    cin = [pos, geo] if cfeat is None else [pos, cfeat, geo]
    rgb = mlp(col_w, torch.cat(cin, dim=-1))
    return torch.cat([rgb, sdf], dim=-1)


# --------------------------------------------------------------------------------------
# the scene state
# --------------------------------------------------------------------------------------
class OracleScene:
    """State of one JointEncoding: config view, bounds, planes, decoder weights."""

    def __init__(self, cfg_dict, bounding_box, generator=None, build=True):
        self.cfg = cfg_dict
        self.pc = PathConfig.from_dict(cfg_dict)
        self.bounding_box = torch.as_tensor(bounding_box)         # raw; float64 in the live system
        self.bound = extended_bound(self.pc)
        if build:
            planes = make_planes(self.bound, [self.pc.coarse, self.pc.fine], self.pc.c_dim, generator)
            if not self.pc.one_grid:
                planes = planes + make_planes(self.bound, [self.pc.c_coarse, self.pc.c_fine],
                                              self.pc.c_dim, generator)
            self.all_planes = tuple(planes)
            self.sdf_w, self.col_w = make_decoder_weights(self.pc, generator)

    # ---- parameters --------------------------------------------------------------
    def plane_list(self):
        return [p for lst in self.all_planes for p in lst]

    def decoder_list(self):
        """Order of nn.Module.parameters() on ColorSDFNet(_v2): color_net first, then sdf_net
        (model/decoder.py:117-126 / :150-159)."""
        return list(self.col_w) + list(self.sdf_w)

    def requires_grad_(self, flag=True):
        for t in self.plane_list() + self.decoder_list():
            t.requires_grad_(flag)
        return self

    # ---- point queries -----------------------------------------------------------
    def embed_pos(self, pts):
        """OneBlob input uses the RAW bounding box (scene_rep.py:292); promoted to the box's
        dtype (float64 live) then cast to fp32 inside the encoding."""
        bb = self.bounding_box
        u = (pts - bb[:, 0]) / (bb[:, 1] - bb[:, 0])
        return oneblob(u.reshape(-1, 3), self.pc.n_bins)

    def query_color_sdf(self, pts, impl="explicit", return_parts=False):
        """model/scene_rep.py:273-301."""
        flat = pts.reshape(-1, 3)
        p_nor = normalize_points(flat, self.bound).float()
        feat = sample_plane_feature(p_nor, *self.all_planes[:3], impl=impl)
        pos = self.embed_pos(flat)
        cfeat = None
        if not self.pc.one_grid:
            cfeat = sample_plane_feature(p_nor, *self.all_planes[3:6], impl=impl)
        raw = decode(self.sdf_w, self.col_w, feat, pos, cfeat)
        if return_parts:
            return raw, dict(p_nor=p_nor, feat=feat, pos=pos, cfeat=cfeat)
        return raw

    def query_sdf(self, pts, return_geo=False, embed=False, impl="explicit"):
        """model/scene_rep.py:232-268."""
        flat = pts.reshape(-1, 3)
        p_nor = normalize_points(flat, self.bound).float()
        feat = sample_plane_feature(p_nor, *self.all_planes[:3], impl=impl)
        if embed:
            return feat.reshape(*pts.shape[:-1], feat.shape[-1])
        h = mlp(self.sdf_w, torch.cat([feat, self.embed_pos(flat)], dim=-1))
        sdf = h[..., 0].reshape(pts.shape[:-1])
        if not return_geo:
            return sdf
        return sdf, h[..., 1:].reshape(*pts.shape[:-1], h.shape[-1] - 1)

    # ---- z sampling --------------------------------------------------------------
    def sample_z(self, n_rays, target_d=None, u=None):
        """Depth-guided sampling + stratified jitter.  model/scene_rep.py:362-381.
        ``u`` [R,S] in [0,1) replaces the reference's CPU torch.rand draw (:381); when None
        it is drawn the same way (global CPU generator) so seeds reproduce the reference."""
        pc = self.pc
        if target_d is not None:
            near_surf = torch.linspace(-pc.range_d, pc.range_d, steps=pc.n_range_d).to(target_d)
            z_s = near_surf[None, :].repeat(n_rays, 1) + target_d
            invalid = target_d.reshape(-1) <= 0
            z_s[invalid] = torch.linspace(pc.near, pc.far, steps=pc.n_range_d).to(target_d)
            if pc.n_samples_d > 0:
                uni = torch.linspace(pc.near, pc.far, pc.n_samples_d)[None, :].repeat(n_rays, 1).to(target_d)
                z, _ = torch.sort(torch.cat([uni, z_s], -1), -1)
            else:
                z = z_s
        else:
            z = torch.linspace(pc.near, pc.far, pc.n_samples)[None, :].repeat(n_rays, 1)
        if pc.perturb > 0.0:
            mids = 0.5 * (z[..., 1:] + z[..., :-1])
            upper = torch.cat([mids, z[..., -1:]], -1)
            lower = torch.cat([z[..., :1], mids], -1)
            if u is None:
                u = torch.rand(z.shape)
            z = lower + (upper - lower) * u.to(z)
        return z

    # ---- compositing -------------------------------------------------------------
    def sdf2weights(self, sdf, z):
        """model/scene_rep.py:183-203."""
        tr = self.pc.trunc
        w = torch.sigmoid(sdf / tr) * torch.sigmoid(-sdf / tr)
        crossing = (sdf[:, 1:] * sdf[:, :-1] < 0.0).to(sdf.dtype)
        first = torch.argmax(crossing, dim=1, keepdim=True)          # 0 when no crossing
        z_min = torch.gather(z, 1, first)
        keep = (z < z_min + self.pc.sc_factor * tr).to(z.dtype)
        w = w * keep
        return w / (w.sum(-1, keepdim=True) + 1e-8)

    def composite(self, raw, z):
        """model/scene_rep.py:205-230 -> dict of maps."""
        rgb = torch.sigmoid(raw[..., :3])
        w = self.sdf2weights(raw[..., 3], z)
        rgb_map = (w[..., None] * rgb).sum(-2)
        depth = (w * z).sum(-1)
        depth_var = (w * (z - depth[:, None]) ** 2).sum(-1)
        acc = w.sum(-1)
        disp = 1.0 / torch.max(1e-10 * torch.ones_like(depth), depth / acc)
        if self.pc.white_bkgd:
            rgb_map = rgb_map + (1.0 - acc[..., None])
        return dict(rgb=rgb_map, depth=depth, disp_map=disp, acc_map=acc, depth_var=depth_var, weights=w)

    def render_rays(self, rays_o, rays_d, target_d=None, u=None, impl="explicit", z_vals=None):
        """model/scene_rep.py:351-419 (n_importance == 0 in every shipped config).
        ``z_vals`` (checker convenience, not reference API): use these samples instead of drawing them,
        so that a batch whose jitter came from another generator can be re-evaluated."""
        assert self.pc.n_importance == 0, "importance sampling is dead code in the reference configs"
        n = rays_o.shape[0]
        z = (self.sample_z(n, target_d, u) if z_vals is None else z_vals).to(rays_o)
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]
        raw = self.query_color_sdf(pts, impl=impl).reshape(n, z.shape[1], 4)
        out = self.composite(raw, z)
        out.pop("weights")
        out["z_vals"] = z
        out["raw"] = raw
        return out

    # ---- losses ------------------------------------------------------------------
    def eslam_sdf_losses(self, sdf, z, gt):
        """model/scene_rep.py:475-511 (means over boolean selections; empty -> NaN)."""
        t = self.pc.truncation
        g = gt[:, None]
        front = z < (g - t)
        back = z > (g + t)
        center = (z > (g - 0.4 * t)) & (z < (g + 0.4 * t))
        tail = (~front) & (~back) & (~center)
        fs = torch.mean((sdf[front] - 1.0) ** 2)
        pred = z + sdf * t
        ge = g.expand(z.shape)
        return fs, torch.mean((pred[center] - ge[center]) ** 2), torch.mean((pred[tail] - ge[tail]) ** 2)

    def coslam_sdf_losses(self, sdf, z, target_d):
        """model/utils.py:117-145 + :164-185 with truncation = trunc*sc_factor
        (model/scene_rep.py:586-587); target_d is [R,1] and broadcasts."""
        t = self.pc.trunc * self.pc.sc_factor
        front = (z < (target_d - t)).to(z.dtype)
        back = (z > (target_d + t)).to(z.dtype)
        valid = (target_d > 0.0).to(z.dtype)
        sdf_m = (1.0 - front) * (1.0 - back) * valid
        n_fs, n_sdf = torch.count_nonzero(front), torch.count_nonzero(sdf_m)
        tot = n_fs + n_sdf
        fs_w, sdf_w = 1.0 - n_fs / tot, 1.0 - n_sdf / tot
        fs = F.mse_loss(sdf * front, torch.ones_like(sdf) * front) * fs_w
        sd = F.mse_loss((z + sdf * t) * sdf_m, target_d * sdf_m) * sdf_w
        return fs, sd

    def forward(self, rays_o, rays_d, target_rgb, target_d, u=None, impl="explicit", z_vals=None):
        """Training-mode JointEncoding.forward.  model/scene_rep.py:549-611."""
        rd = self.render_rays(rays_o, rays_d, target_d, u, impl, z_vals=z_vals)
        td = target_d.squeeze()
        valid = (td > 0.0) & (td < self.pc.depth_trunc)
        rgb_loss = F.mse_loss(rd["rgb"], target_rgb)
        psnr = -10.0 * torch.log(rgb_loss) / torch.log(torch.tensor([10.0]))
        depth_loss = F.mse_loss(rd["depth"].squeeze()[valid], td[valid])
        z, sdf = rd["z_vals"], rd["raw"][..., -1]
        co_fs, co_sdf = self.coslam_sdf_losses(sdf, z, target_d)
        has_d = td > 0
        e_fs, e_center, e_tail = self.eslam_sdf_losses(sdf[has_d], z[has_d], td[has_d])
        return dict(rgb=rd["rgb"], depth=rd["depth"], rgb_loss=rgb_loss, depth_loss=depth_loss,
                    co_sdf_loss=co_sdf, co_fs_loss=co_fs, e_fs_loss=e_fs, e_center_loss=e_center,
                    e_tail_loss=e_tail, psnr=psnr, z_vals=z, raw=rd["raw"], depth_var=rd["depth_var"],
                    acc_map=rd["acc_map"], disp_map=rd["disp_map"])


# --------------------------------------------------------------------------------------
# hash-grid wiring (PARITY UNPINNED: the reference keeps this call commented out, model/scene_rep.py:160,243)
# --------------------------------------------------------------------------------------
class OracleHashScene(OracleScene):
    """OracleScene whose first decoder input is the multiresolution hash grid of oracle/hashgrid.py evaluated at the
    OneBlob input u = (p - bb_lo) / (bb_hi - bb_lo) (Co-SLAM's ``embed_fn(inputs_flat)``), placed in the first
    n_levels*F columns of the decoder's 64-wide feature slot; the other columns are zero.  Checker for the build's own
    HashFusedStep: there is no reference behaviour to pin it to."""

    def __init__(self, cfg_dict, bounding_box, table, grid, scales=None):
        super().__init__(cfg_dict, bounding_box, build=False)
        self.table, self.grid, self.scales = table, dict(grid), scales
        self.all_planes = ()

    def plane_list(self):
        return [self.table]

    def grid_features(self, flat):
        from .hashgrid import grid_encode
        bb = self.bounding_box
        u = ((flat - bb[:, 0]) / (bb[:, 1] - bb[:, 0])).float()
        f = grid_encode(u, self.table, scales=self.scales, **self.grid)
        return torch.cat([f, torch.zeros(f.shape[0], 64 - f.shape[1], dtype=f.dtype)], dim=-1)

    def query_color_sdf(self, pts, impl="explicit", return_parts=False):
        flat = pts.reshape(-1, 3)
        feat, pos = self.grid_features(flat), self.embed_pos(flat)
        raw = decode(self.sdf_w, self.col_w, feat, pos, None)
        return (raw, dict(feat=feat, pos=pos)) if return_parts else raw

    def query_sdf(self, pts, return_geo=False, embed=False, impl="explicit"):
        raise NotImplementedError
