"""Mapper -- the mapping-thread methods that drive the hot path (reference: mp_slam/mapper.py):
``first_frame_mapping`` (:52-89, the training loop only), ``mapping_optimize`` (:118-162) and its
alias ``optimize_map`` (the name BASELINE.json uses).  Keyframe bookkeeping, image/mesh dumps, loop
closure and fusion around these calls stay with the host application.

``SLAM`` is the reference's MNESLAM-like object; the fields read here are the ones the reference's
Mapper reads for these methods: ``model``, ``map_optimizer``, ``device``, ``dataset.H/.W``,
``video.keyframe`` (KeyFrameDatabase), ``get_loss_from_ret``, ``select_samples``.
"""
import random

import torch

from ..fused import FusedStep


class Mapper():
    """``compute``: "autograd" = the reference's own sequence (model.forward -> get_loss_from_ret ->
    backward -> map_optimizer.step/zero_grad) through the autograd node; "fused" = FusedStep (same
    math, one forward+backward kernel, no graph).  ``sampler``: "host" = python ``random`` draws in
    the reference's order (seed-for-seed identical batches); "device" = keyed permutation on the GPU
    (no host work per iteration).  Defaults reproduce the reference exactly."""

    def __init__(self, config, SLAM, compute="autograd", sampler="host", scatter="binned") -> None:
        if compute not in ("autograd", "fused") or sampler not in ("host", "device"):
            raise ValueError("compute must be autograd|fused and sampler host|device")
        if sampler == "device" and compute != "fused":
            raise ValueError("the device sampler is part of the fused path")
        self.compute, self.sampler, self.scatter = compute, sampler, scatter
        self._fused = {}
        self.fused_kwargs = {}          # extra FusedStep options (e.g. tile_capacity)
        self.config = config
        self.slam = SLAM
        self.model = SLAM.model
        self.model_shared = getattr(SLAM, "model_shared", None)
        self.map_optimizer = SLAM.map_optimizer
        self.device = SLAM.device
        self.dataset = SLAM.dataset
        self.video = SLAM.video
        self.rank = getattr(SLAM, "rank", 0)
        self.world_size = getattr(SLAM, "world_size", 1)

    def first_frame_mapping(self, batch, n_iters=100):
        """Training loop of the first frame (reference: mp_slam/mapper.py:65-89): ``n_iters`` x
        {zero_grad, python-random pixel selection, forward, loss, backward, Adam step}."""
        if batch["frame_id"] != 0:
            raise ValueError("First frame mapping must be the first frame!")
        c2w = batch["c2w"].to(self.device)
        self.model.train()
        H, n = self.slam.dataset.H, self.config["mapping"]["sample"]
        if self.compute == "fused":
            return self._first_frame_fused(batch, c2w, n_iters)
        for _ in range(n_iters):
            self.map_optimizer.zero_grad()
            indice = self.slam.select_samples(self.slam.dataset.H, self.slam.dataset.W, n)
            indice_h = indice % H
            indice_w = torch.div(indice, H, rounding_mode="trunc")
            rays_d_cam = batch["direction"][indice_h, indice_w, :].to(self.device)
            target_s = batch["rgb"][indice_h, indice_w, :].to(self.device)
            target_d = batch["depth"][indice_h, indice_w].to(self.device).unsqueeze(-1)
            rays_o = c2w[None, :3, -1].repeat(n, 1)
            rays_d = torch.sum(rays_d_cam[..., None, :] * c2w[:3, :3], -1)
            ret = self.model.forward(rays_o, rays_d, target_s, target_d)
            loss = self.slam.get_loss_from_ret(ret, is_co_sdf=self.config["is_co_sdf"])
            loss.backward()
            self.map_optimizer.step()

    def mapping_optimize(self, batch, poses):
        """Global bundle adjustment over all keyframes + the current frame (reference:
        mp_slam/mapper.py:118-162).  ``poses`` [N,4,4] c2w; rows sampled from the current frame use
        ``poses[-1]`` (id -1)."""
        self.map_optimizer.zero_grad()
        current_rays = torch.cat([batch["direction"], batch["rgb"], batch["depth"][..., None]], dim=-1)
        current_rays = current_rays.reshape(-1, current_rays.shape[-1])
        n = self.config["mapping"]["sample"]
        if self.compute == "fused":
            return self._mapping_optimize_fused(current_rays, poses)
        for _ in range(self.config["mapping"]["iters"]):
            rays, ids = self.video.keyframe.sample_global_rays(n)
            idx_cur = random.sample(range(0, self.slam.dataset.H * self.slam.dataset.W),
                                    max(n // len(self.video.keyframe.frame_ids), self.config["mapping"]["min_pixels_cur"]))
            rays = torch.cat([rays, current_rays[idx_cur, :]], dim=0)
            ids_all = torch.cat([ids, -torch.ones((len(idx_cur)))]).to(torch.int64)
            rays_d_cam = rays[..., :3].to(self.device)
            target_s = rays[..., 3:6].to(self.device)
            target_d = rays[..., 6:7].to(self.device)
            rot = poses[ids_all.to(poses.device), :3, :3]
            rays_d = torch.sum(rays_d_cam[:, None, :] * rot, -1)
            rays_o = poses[ids_all.to(poses.device), :3, -1]
            ret = self.model.forward(rays_o, rays_d, target_s, target_d)
            loss = self.slam.get_loss_from_ret(ret, is_co_sdf=self.config["is_co_sdf"])
            loss.backward()
            self.map_optimizer.step()
            self.map_optimizer.zero_grad()

    optimize_map = mapping_optimize

    # ------------------------------------------------------------------ fused path
    def _fused_step(self, n_rays):
        key = (n_rays, id(self.map_optimizer))
        if key not in self._fused:
            self._fused[key] = FusedStep(self.model, self.map_optimizer, self.config, n_rays, self.device,
                                         scatter=self.scatter, **self.fused_kwargs)
        return self._fused[key]

    def _jitter(self, fs):
        if self.sampler == "host" and self.config["training"]["perturb"] > 0.0:
            return torch.rand(fs.R, fs.S).to(self.device)        # the reference's CPU draw (scene_rep.py:381)
        return None

    def _mapping_optimize_fused(self, current_rays, poses):
        kf = self.video.keyframe
        n, n_kf = self.config["mapping"]["sample"], len(kf.frame_ids)
        n_cur = max(n // n_kf, self.config["mapping"]["min_pixels_cur"])
        fs = self._fused_step(n + n_cur)
        kf_rays = kf.device_rays(self.device)
        cur = current_rays.to(self.device, torch.float32).contiguous()
        poses = poses.to(self.device, torch.float32).contiguous()
        n_pix = self.slam.dataset.H * self.slam.dataset.W
        n_it = self.config["mapping"]["iters"]
        for it in range(n_it):
            idx_g = idx_c = None
            if self.sampler == "host":                               # same draws, same order as the reference
                idx_g = torch.tensor(random.sample(range(n_kf * kf.num_rays_to_save), n)).to(self.device)
                idx_c = torch.tensor(random.sample(range(0, n_pix), n_cur)).to(self.device)
            fs.step(kf_rays, n_kf * kf.num_rays_to_save, kf.num_rays_to_save, cur, poses, n, n_cur,
                    idx_global=idx_g, idx_cur=idx_c, u=self._jitter(fs), prefetch=it + 1 < n_it)
        self.last_losses = fs.loss_dict()

    def _first_frame_fused(self, batch, c2w, n_iters):
        H, W, n = self.slam.dataset.H, self.slam.dataset.W, self.config["mapping"]["sample"]
        fs = self._fused_step(n)
        cur = torch.cat([batch["direction"], batch["rgb"], batch["depth"][..., None]], dim=-1)
        cur = cur.reshape(-1, 7).to(self.device, torch.float32).contiguous()
        poses = c2w.reshape(1, 4, 4).to(torch.float32).contiguous()
        for it in range(n_iters):
            idx_c = None
            if self.sampler == "host":
                ind = self.slam.select_samples(H, W, n)
                # the reference indexes [H,W] images with (ind % H, ind // H)  (mp_slam/mapper.py:76-77)
                idx_c = ((ind % H) * W + torch.div(ind, H, rounding_mode="trunc")).to(self.device)
            fs.step(None, 0, 1, cur, poses, 0, n, idx_cur=idx_c, u=self._jitter(fs), prefetch=it + 1 < n_iters)
        self.last_losses = fs.loss_dict()
