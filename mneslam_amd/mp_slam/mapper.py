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


class Mapper():
    def __init__(self, config, SLAM) -> None:
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
