"""CPU restatement of the mapping-iteration driver (TEST INFRASTRUCTURE ONLY).

Follows dtc111111/MNESLAM:
  mneslam_mp.py:342-348   select_samples
  mneslam_mp.py:350-372   get_loss_from_ret
  mneslam_mp.py:431-469   create_optimizer (Adam groups)  + torch.optim.Adam arithmetic
  model/keyframe.py:64-103   KeyFrameDatabase.add_keyframe / sample_global_rays
  mp_slam/mapper.py:118-162  Mapper.mapping_optimize
  mp_slam/mapper.py:52-89    Mapper.first_frame_mapping (inner loop)
"""
import math
import random
from typing import Dict, List

import torch

from .scene_rep import OracleScene


# --------------------------------------------------------------------------------------
# loss weighting
# --------------------------------------------------------------------------------------
def loss_from_ret(cfg, ret, is_co_sdf=True, rgb=True, depth=True, sdf=True):
    """mneslam_mp.py:350-372 (the ``smooth`` branch is never enabled by any caller)."""
    loss = 0
    if rgb:
        loss = loss + cfg["training"]["rgb_weight"] * ret["rgb_loss"]
    if depth:
        loss = loss + cfg["training"]["depth_weight"] * ret["depth_loss"]
    if sdf:
        if is_co_sdf:
            loss = loss + (cfg["training"]["sdf_weight"] * ret["co_sdf_loss"]
                           + cfg["training"]["fs_weight"] * ret["co_fs_loss"])
        else:
            loss = loss + (cfg["mapping"]["w_sdf_fs"] * ret["e_fs_loss"]
                           + cfg["mapping"]["w_sdf_center"] * ret["e_center_loss"]
                           + cfg["mapping"]["w_sdf_tail"] * ret["e_tail_loss"])
    return loss


# --------------------------------------------------------------------------------------
# Adam, written out (torch.optim.Adam, amsgrad=False, maximize=False)
# --------------------------------------------------------------------------------------
class AdamGroup:
    def __init__(self, params: List[torch.Tensor], lr, eps=1e-8, weight_decay=0.0, betas=(0.9, 0.99)):
        self.params, self.lr, self.eps, self.wd, self.betas = params, lr, eps, weight_decay, betas
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.t = 0


class OracleAdam:
    """Dense Adam with the param groups of mneslam_mp.py:459-469:
    decoder {lr_decoder, weight_decay 1e-6 (L2 into the gradient), eps 1e-8},
    planes {lr_embed, eps 1e-15}, colour planes {lr_embed_color, eps 1e-15}; betas (0.9, 0.99).
    Update (torch >= 2 single-tensor form): m = lerp(m, g, 1-b1); v = b2*v + (1-b2)*g*g;
    p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""

    def __init__(self, scene: OracleScene, cfg):
        m = cfg["mapping"]
        self.groups = [AdamGroup(scene.decoder_list(), m["lr_decoder"], eps=1e-8, weight_decay=1e-6)]
        geo = [p for lst in scene.all_planes[:3] for p in lst] if scene.all_planes else scene.plane_list()    # hash wiring: the table
        # reference order inside the group: xy[coarse,fine], xz[...], yz[...]  (mneslam_mp.py:453-457)
        self.groups.append(AdamGroup(geo, m["lr_embed"], eps=1e-15))
        if not scene.pc.one_grid:
            col = [p for lst in scene.all_planes[3:6] for p in lst]
            self.groups.append(AdamGroup(col, m["lr_embed_color"], eps=1e-15))

    def zero_grad(self):
        for g in self.groups:
            for p in g.params:
                p.grad = None

    @torch.no_grad()
    def step(self):
        for g in self.groups:
            g.t += 1
            b1, b2 = g.betas
            bc1 = 1.0 - b1 ** g.t
            bc2 = 1.0 - b2 ** g.t
            step_size = g.lr / bc1
            bc2_sqrt = math.sqrt(bc2)
            for p, m, v in zip(g.params, g.m, g.v):
                if p.grad is None:
                    continue
                grad = p.grad
                if g.wd != 0.0:
                    grad = grad + g.wd * p
                m.add_((grad - m) * (1.0 - b1))                 # lerp_(grad, 1-b1)
                v.mul_(b2).add_(grad * grad * (1.0 - b2))
                denom = v.sqrt() / bc2_sqrt + g.eps
                p.add_(-step_size * (m / denom))


# --------------------------------------------------------------------------------------
# keyframe ray database + sampling (host RNG = python ``random``, as the reference)
# --------------------------------------------------------------------------------------
class OracleKeyframeDB:
    """model/keyframe.py:6-19, :64-103.  rays[k] = [dir3, rgb3, depth1] per stored ray."""

    def __init__(self, H, W, num_kf, num_rays_to_save):
        self.H, self.W, self.n_save = H, W, num_rays_to_save
        self.rays = torch.zeros((num_kf, num_rays_to_save, 7))
        self.all_ids = torch.arange(0, num_kf, dtype=torch.int32)
        self.frame_ids = [0]

    def __len__(self):
        return len(self.frame_ids)

    def add_keyframe(self, batch, counter):
        rays = torch.cat([batch["direction"], batch["rgb"], batch["depth"][..., None]], dim=-1)
        rays = rays.reshape(1, -1, 7)
        idxs = random.sample(range(0, self.H * self.W), self.n_save)
        self.frame_ids = self.all_ids[:counter]
        self.rays[counter - 1] = rays[:, idxs]

    def sample_global_rays(self, bs):
        nkf = len(self.frame_ids)
        idxs = torch.tensor(random.sample(range(nkf * self.n_save), bs))
        rays = self.rays[:nkf].reshape(-1, 7)[idxs]
        ids = self.frame_ids[torch.div(idxs, self.n_save, rounding_mode="trunc")]
        return rays, ids


def select_samples(H, W, samples):
    """mneslam_mp.py:342-348."""
    return torch.tensor(random.sample(range(H * W), int(samples)))


# --------------------------------------------------------------------------------------
# the iteration driver
# --------------------------------------------------------------------------------------
def assemble_rays(rays7, ids_all, poses):
    """mp_slam/mapper.py:146-153: rotate camera-frame directions by the owning pose
    (id -1 -> poses[-1], the current frame)."""
    d_cam, tgt_rgb, tgt_d = rays7[..., :3], rays7[..., 3:6], rays7[..., 6:7]
    rays_d = torch.sum(d_cam[..., None, None, :] * poses[ids_all, None, :3, :3], -1)
    rays_o = poses[ids_all, None, :3, -1].repeat(1, rays_d.shape[1], 1).reshape(-1, 3)
    return rays_o, rays_d.reshape(-1, 3), tgt_rgb, tgt_d


def mapping_optimize(scene: OracleScene, opt: OracleAdam, cfg, kfdb: OracleKeyframeDB, batch, poses,
                     H, W, impl="grid_sample", log: List[Dict] = None, iters=None):
    """mp_slam/mapper.py:118-162.  Uses the same python ``random`` draws in the same order
    (global rays first, then current-frame pixels) and the CPU torch generator for the jitter."""
    opt.zero_grad()
    cur = torch.cat([batch["direction"], batch["rgb"], batch["depth"][..., None]], dim=-1).reshape(-1, 7)
    n_it = cfg["mapping"]["iters"] if iters is None else iters
    for _ in range(n_it):
        rays, ids = kfdb.sample_global_rays(cfg["mapping"]["sample"])
        idx_cur = random.sample(range(0, H * W),
                                max(cfg["mapping"]["sample"] // len(kfdb.frame_ids),
                                    cfg["mapping"]["min_pixels_cur"]))
        rays = torch.cat([rays, cur[idx_cur, :]], dim=0)
        ids_all = torch.cat([ids, -torch.ones((len(idx_cur)))]).to(torch.int64)
        rays_o, rays_d, tgt_rgb, tgt_d = assemble_rays(rays, ids_all, poses)
        ret = scene.forward(rays_o, rays_d, tgt_rgb, tgt_d, impl=impl)
        loss = loss_from_ret(cfg, ret, is_co_sdf=cfg["is_co_sdf"])
        loss.backward()
        opt.step()
        opt.zero_grad()
        if log is not None:
            log.append({k: float(v) for k, v in ret.items() if v.numel() == 1} | {"loss": float(loss)})


def first_frame_mapping(scene: OracleScene, opt: OracleAdam, cfg, batch, H, W, n_iters,
                        impl="grid_sample", log=None):
    """Inner loop of mp_slam/mapper.py:72-89 (NB: ``indice % H`` / ``indice // H`` as written there)."""
    c2w = batch["c2w"]
    n = cfg["mapping"]["sample"]
    for _ in range(n_iters):
        opt.zero_grad()
        ind = select_samples(H, W, n)
        ih, iw = ind % H, torch.div(ind, H, rounding_mode="trunc")
        d_cam = batch["direction"][ih, iw, :]
        tgt_rgb = batch["rgb"][ih, iw, :]
        tgt_d = batch["depth"][ih, iw].unsqueeze(-1)
        rays_o = c2w[None, :3, -1].repeat(n, 1)
        rays_d = torch.sum(d_cam[..., None, :] * c2w[:3, :3], -1)
        ret = scene.forward(rays_o, rays_d, tgt_rgb, tgt_d, impl=impl)
        loss = loss_from_ret(cfg, ret, is_co_sdf=cfg["is_co_sdf"])
        loss.backward()
        opt.step()
        if log is not None:
            log.append({k: float(v) for k, v in ret.items() if v.numel() == 1} | {"loss": float(loss)})


# --------------------------------------------------------------------------------------
# one iteration's forward + backward in ray chunks (checker convenience for batches whose autograd graph does not fit
# in host memory: INS Indoor, 2048 rays x 1045 samples)
# --------------------------------------------------------------------------------------
def forward_backward_chunked(scene: OracleScene, cfg, rays_o, rays_d, target_rgb, target_d, z_vals, is_co_sdf=None,
                             impl="grid_sample", chunk=256):
    """``scene.forward`` + ``loss_from_ret(...).backward()`` (model/scene_rep.py:549-611, mneslam_mp.py:350-372) evaluated
    chunk by chunk over the rays.  Every loss of the reference is a sum of per-ray / per-sample terms divided by a count
    that depends on z and the target depth alone (mask sizes, model/scene_rep.py:489-499, model/utils.py:131-145), so the
    counts of the WHOLE batch are taken first (no network), each chunk then contributes  weight_k * (its sum) / count_k
    to the total and its ``backward()`` accumulates into the same ``.grad`` tensors: identical to the unchunked call up to
    fp32 summation order (checked in tests/test_oracle_golden.py).  Returns the detached loss dict (+ 'loss')."""
    pc = scene.pc
    co = cfg["is_co_sdf"] if is_co_sdf is None else is_co_sdf
    tr, mp = cfg["training"], cfg["mapping"]
    R, S = z_vals.shape
    td = target_d.reshape(-1)
    z = z_vals
    with torch.no_grad():
        valid = (td > 0.0) & (td < pc.depth_trunc)
        has_d = td > 0
        g = td[:, None]
        t = pc.truncation
        front = (z < (g - t)) & has_d[:, None]
        back = (z > (g + t)) & has_d[:, None]
        center = (z > (g - 0.4 * t)) & (z < (g + 0.4 * t)) & has_d[:, None]
        tail = (~front) & (~back) & (~center) & has_d[:, None]
        n_front, n_center, n_tail = int(front.sum()), int(center.sum()), int(tail.sum())
        tc = pc.trunc * pc.sc_factor
        cf = z < (g - tc)
        cs = (~cf) & (~(z > (g + tc))) & (g > 0.0)
        n_fs, n_sdf = int(cf.sum()), int(cs.sum())
        tot = n_fs + n_sdf
        fs_w = torch.tensor(1.0) - torch.tensor(float(n_fs)) / torch.tensor(float(tot))     # fp32, as the reference's tensor ops
        sdf_w = torch.tensor(1.0) - torch.tensor(float(n_sdf)) / torch.tensor(float(tot))
        n_valid = int(valid.sum())
    nan = float("nan")
    sums = dict(rgb=0.0, depth=0.0, co_fs=0.0, co_sdf=0.0, e_fs=0.0, e_center=0.0, e_tail=0.0)
    rgb_out, depth_out, raw_out = [], [], []
    for lo in range(0, R, chunk):
        sl = slice(lo, min(lo + chunk, R))
        rd = scene.render_rays(rays_o[sl], rays_d[sl], target_d[sl], None, impl, z_vals=z[sl])
        sdf, zc, gc = rd["raw"][..., -1], z[sl], g[sl]
        pred_e, pred_c = zc + sdf * t, zc + sdf * tc
        part = dict(
            rgb=((rd["rgb"] - target_rgb[sl]) ** 2).sum(),
            depth=((rd["depth"] - td[sl]) ** 2)[valid[sl]].sum(),
            co_fs=((sdf - 1.0) ** 2)[cf[sl]].sum(),
            co_sdf=((pred_c - gc) ** 2)[cs[sl]].sum(),
            e_fs=((sdf - 1.0) ** 2)[front[sl]].sum(),
            e_center=((pred_e - gc) ** 2)[center[sl]].sum(),
            e_tail=((pred_e - gc) ** 2)[tail[sl]].sum())
        loss = tr["rgb_weight"] * part["rgb"] / (3.0 * R)
        if n_valid:
            loss = loss + tr["depth_weight"] * part["depth"] / n_valid
        if co:
            if tot:
                loss = loss + tr["sdf_weight"] * part["co_sdf"] / (R * S) * sdf_w + tr["fs_weight"] * part["co_fs"] / (R * S) * fs_w
        else:
            if n_front:
                loss = loss + mp["w_sdf_fs"] * part["e_fs"] / n_front
            if n_center:
                loss = loss + mp["w_sdf_center"] * part["e_center"] / n_center
            if n_tail:
                loss = loss + mp["w_sdf_tail"] * part["e_tail"] / n_tail
        loss.backward()
        for k in sums:
            sums[k] += float(part[k].detach())
        rgb_out.append(rd["rgb"].detach()); depth_out.append(rd["depth"].detach()); raw_out.append(rd["raw"].detach())
    T = torch.tensor
    rgb_loss = T(sums["rgb"] / (3.0 * R))
    ret = dict(rgb=torch.cat(rgb_out), depth=torch.cat(depth_out), raw=torch.cat(raw_out), z_vals=z,
               rgb_loss=rgb_loss, depth_loss=T(sums["depth"] / n_valid if n_valid else nan),
               co_sdf_loss=T(sums["co_sdf"] / (R * S)) * sdf_w, co_fs_loss=T(sums["co_fs"] / (R * S)) * fs_w,
               e_fs_loss=T(sums["e_fs"] / n_front if n_front else nan),
               e_center_loss=T(sums["e_center"] / n_center if n_center else nan),
               e_tail_loss=T(sums["e_tail"] / n_tail if n_tail else nan),
               psnr=-10.0 * torch.log(rgb_loss.reshape(1)) / torch.log(T([10.0])))
    return ret
