"""Mapper -- the mapping-thread methods that drive the hot path (reference: mp_slam/mapper.py):
``first_frame_mapping`` (:52-89, the training loop only), ``mapping_optimize`` (:118-162) and its
alias ``optimize_map`` (the name BASELINE.json uses), plus the two training loops of loop closure that
run on the same kernels: the pose alignment of ``handle_loop_closure`` (:362-412) and ``distillation``
(:594-644).  Keyframe bookkeeping, image/mesh dumps, file exchange and fusion policy around these calls
stay with the host application.

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

    # ------------------------------------------------------------------ N2: loop-closure loops on R13
    def optimize_relative_pose(self, base_c2w, target_c2w_initial, model_for_base, model_for_target, n_iters=None,
                               rays_d_cam_batch=None):
        """Pose alignment loop of ``handle_loop_closure`` (reference: mp_slam/mapper.py:362-412): render
        ``mapping.sample`` random camera rays from ``base_c2w`` with the base model (teacher, no grad),
        then ``loop_iters`` Adam steps on the 6 pose parameters of the target so that the target model
        renders the same rgb/depth: ``render_rays`` (R13: gradients reach the rays) -> MSE losses weighted by
        ``training.rgb_weight / depth_weight`` -> backward -> ``pose_optimizer.step()``.  Pose
        parametrisation and optimizer are the host's (``SLAM.get_pose_param_optim``, ``SLAM.matrix_from_tensor``,
        mneslam_mp.py:577-584).  Returns (relative_transform = base_c2w @ inv(best target pose), best loss)."""
        cfg, dev = self.config, self.device
        n = cfg["mapping"]["sample"]
        n_iters = cfg["mapping"]["loop_iters"] if n_iters is None else n_iters
        base_c2w, target_c2w_initial = base_c2w.to(dev), target_c2w_initial.to(dev)
        target_rot, target_trans, pose_optimizer = self.slam.get_pose_param_optim(target_c2w_initial[None, ...], mapping=False)
        with torch.no_grad():
            if rays_d_cam_batch is None:
                rays_d_cam = self.dataset.rays_d.reshape(-1, 3)
                sample_indices = torch.randint(0, len(rays_d_cam), (n,))
                rays_d_cam_batch = rays_d_cam[sample_indices]
            rays_d_cam_batch = rays_d_cam_batch.to(dev)
            n = rays_d_cam_batch.shape[0]
            rays_o_base = base_c2w[:3, 3].unsqueeze(0).repeat(n, 1)
            rays_d_base = torch.sum(rays_d_cam_batch[..., None, :] * base_c2w[:3, :3], dim=-1)
            base_ret = model_for_base.render_rays(rays_o_base, rays_d_base, target_d=None)
            target_rgb, target_depth = base_ret["rgb"].detach(), base_ret["depth"].detach()
        best_loss, best_c2w = float("inf"), target_c2w_initial.clone()
        for _ in range(n_iters):
            pose_optimizer.zero_grad()
            c2w_est = self.slam.matrix_from_tensor(target_rot, target_trans).squeeze(0)
            rays_o = c2w_est[:3, 3].unsqueeze(0).repeat(n, 1)
            rays_d = torch.sum(rays_d_cam_batch[..., None, :] * c2w_est[:3, :3], dim=-1)
            ret = model_for_target.render_rays(rays_o, rays_d, target_d=None)
            loss_c = torch.nn.functional.mse_loss(ret["rgb"], target_rgb)
            loss_d = torch.nn.functional.mse_loss(ret["depth"], target_depth)
            loss = cfg["training"]["rgb_weight"] * loss_c + cfg["training"]["depth_weight"] * loss_d
            if loss.item() < best_loss:
                best_loss, best_c2w = loss.item(), c2w_est.detach().clone()
            loss.backward()
            pose_optimizer.step()
        return base_c2w @ torch.inverse(best_c2w), best_loss

    def distillation(self, other_rank, expanded_foreign_kfs_for_distill, num_expanded_kfs):
        """Joint distillation of a foreign agent's map (``model_shared``, the teacher) into ``model``
        (reference: mp_slam/mapper.py:594-644, the training loop; the mesh dump after it stays with the
        host): per iteration, ``sample_per_match`` random camera rays per foreign keyframe pose, teacher
        ``render_rays`` without depth guidance, student ``forward`` on the teacher's rgb/depth, the usual
        weighted loss, backward, Adam."""
        cfg = self.config
        for _ in range(cfg["mapping"]["distill_iters"]):
            all_o, all_d, all_rgb, all_depth = [], [], [], []
            per = max(cfg["mapping"]["sample"] // num_expanded_kfs, cfg["mapping"]["min_pixels_cur"]) \
                if num_expanded_kfs > 0 else cfg["mapping"]["sample"]
            for kf_data in expanded_foreign_kfs_for_distill:
                pose = kf_data["pose"].to(self.device)
                rays_d_cam = self.dataset.rays_d.reshape(-1, 3)
                idx = torch.randint(0, len(rays_d_cam), (per,))
                rays_d_cam_batch = rays_d_cam[idx].to(self.device)
                rays_o = pose[:3, 3].unsqueeze(0).repeat(per, 1)
                rays_d = torch.sum(rays_d_cam_batch[..., None, :] * pose[:3, :3], dim=-1)
                all_o.append(rays_o)
                all_d.append(rays_d)
                with torch.no_grad():
                    teacher = self.model_shared.render_rays(rays_o, rays_d, target_d=None)
                    all_rgb.append(teacher["rgb"].detach())
                    all_depth.append(teacher["depth"].detach().unsqueeze(-1))
            if not all_o:
                continue
            self.map_optimizer.zero_grad()
            ret = self.model.forward(torch.cat(all_o, 0), torch.cat(all_d, 0), torch.cat(all_rgb, 0), torch.cat(all_depth, 0))
            loss = self.slam.get_loss_from_ret(ret, is_co_sdf=cfg["is_co_sdf"])
            loss.backward()
            self.map_optimizer.step()

    # ------------------------------------------------------------------ fused path
    def _fused_step(self, n_rays):
        """ONE FusedStep per (model, optimizer), sized for the largest batch the mapping loop can ask for --
        ``sample`` global rays + max(sample // 1, min_pixels_cur) current-frame rays -- and reused for every
        smaller batch (n_cur shrinks as keyframes accumulate): its scratch (tape, tile lists, spill area) is
        allocated once, not once per distinct ray count."""
        n = self.config["mapping"]["sample"]
        cap = max(n_rays, n + max(n, self.config["mapping"]["min_pixels_cur"]))
        key = (id(self.model), id(self.map_optimizer))
        fs = self._fused.get(key)
        if fs is None or fs.R < n_rays:
            self._fused.clear()                      # drop the previous scratch before allocating the new one
            fs = self._fused[key] = FusedStep(self.model, self.map_optimizer, self.config, cap, self.device,
                                              scatter=self.scatter, **self.fused_kwargs)
        return fs

    def _jitter(self, fs):
        if self.sampler == "host" and self.config["training"]["perturb"] > 0.0:
            return torch.rand(self._n_batch, fs.S).to(self.device)   # the reference's CPU draw (scene_rep.py:381)
        return None

    def _mapping_optimize_fused(self, current_rays, poses):
        kf = self.video.keyframe
        n, n_kf = self.config["mapping"]["sample"], len(kf.frame_ids)
        n_cur = max(n // n_kf, self.config["mapping"]["min_pixels_cur"])
        fs = self._fused_step(n + n_cur)
        self._n_batch = n + n_cur
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
        fs.check()
        self.last_losses = fs.loss_dict()

    def _first_frame_fused(self, batch, c2w, n_iters):
        H, W, n = self.slam.dataset.H, self.slam.dataset.W, self.config["mapping"]["sample"]
        fs = self._fused_step(n)
        self._n_batch = n
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
        fs.check()
        self.last_losses = fs.loss_dict()
