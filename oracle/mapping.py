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
