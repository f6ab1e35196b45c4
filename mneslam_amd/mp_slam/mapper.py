"""Mapping-thread entry points of the hot path.

What a host application needs depends on the integration level:

1. **Drop-in scene model only.**  The host keeps its own ``mp_slam.mapper.Mapper`` untouched and constructs
   ``mneslam_amd.model.scene_rep.JointEncoding`` (+ ``mneslam_amd.optim.FusedAdam``) instead of its own classes: the
   host's loops call ``model.forward -> get_loss_from_ret -> loss.backward() -> map_optimizer.step()`` and every one of
   those lands in the HIP kernels.  ``tests/test_dropin_reference.py`` runs the reference's unmodified Mapper that way.

2. **Fused iteration.**  ``FusedMappingMixin`` goes in FRONT of the host's Mapper class,

       from mp_slam.mapper import Mapper as HostMapper
       Mapper = mneslam_amd.mp_slam.mapper.bind(HostMapper, sampler="host")

   and replaces exactly two methods: ``mapping_optimize`` (reference mp_slam/mapper.py:118-162 -- nothing but the
   iteration loop) and the training loop of ``first_frame_mapping`` (:72-89) -- the host's own method is then called
   with ``n_iters=0`` so that ITS bookkeeping (first keyframe, descriptor, checkpoint, image / mesh dumps, :91-116) runs
   unchanged.  ``run()``, ``final_run()``, loop closure and fusion policy stay the host's.

``Mapper`` below is the stand-alone composition used by this repository's tests and bench (no host application here):
the mixin over a minimal base that owns the reference's field names and the plain autograd iteration.
"""
import random

import torch

from ..fused import FusedStep


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def frame_ray_table(batch):
    """[H*W, 7] = dir3 | rgb3 | depth1 of one frame dict, row-major over (h, w)."""
    return torch.cat([batch["direction"], batch["rgb"], batch["depth"][..., None]], dim=-1).reshape(-1, 7)


def world_rays(rays7, c2w_of_row):
    """Camera-frame rows -> (rays_o, rays_d, target_rgb, target_d[:,None]) with one c2w per row ([N,4,4]):
    rays_d = R dir, rays_o = t  (mp_slam/mapper.py:151-153)."""
    rot, trans = c2w_of_row[:, :3, :3], c2w_of_row[:, :3, 3]
    rays_d = torch.sum(rays7[:, None, :3] * rot, -1)          # elementwise product + sum: the reference's rounding
    return trans, rays_d, rays7[:, 3:6], rays7[:, 6:7]


class FusedMappingMixin:
    """Fused replacements of the two training loops; see the module docstring.  Reads the host Mapper's fields
    ``config, slam, model, map_optimizer, device, video`` (mp_slam/mapper.py:12-24).  Options are class attributes so
    that ``bind`` can set them without touching the host's constructor."""
    compute = "fused"          # "autograd": leave the host's loops alone
    sampler = "host"           # "host": python-random draws in the reference's order;  "device": keyed permutation
    scatter = "binned"
    fused_kwargs = {}

    # ---------------------------------------------------------------- plumbing
    def _fused_step(self, n_rays):
        """ONE FusedStep per (model, optimizer), sized for the largest batch the mapping loop can ask for --
        ``sample`` global rays + max(sample, min_pixels_cur) current-frame rays -- and reused for every smaller batch
        (n_cur shrinks as keyframes accumulate): tape, tile lists and spill area are allocated once."""
        n = self.config["mapping"]["sample"]
        cap = max(n_rays, n + max(n, self.config["mapping"]["min_pixels_cur"]))
        cache = self.__dict__.setdefault("_fused", {})
        key = (id(self.model), id(self.map_optimizer))
        fs = cache.get(key)
        if fs is None or fs.R < n_rays:
            cache.clear()                            # release the previous scratch before allocating the new one
            fs = cache[key] = FusedStep(self.model, self.map_optimizer, self.config, cap, self.device,
                                        scatter=self.scatter, **self.fused_kwargs)
            # The loops below run under ``_map_guard`` (the exchange's lock, also taken by the service thread that answers a
            # peer's fetch).  A step that itself waits for the peers -- decoder all-reduce, overlap exchange -- under that lock
            # can deadlock: A sits in fetch(B), B's service waits for B's lock, B's mapper holds it inside an all-reduce that
            # waits for A.  bench.py --split drives such steps directly, without an exchange; here they are refused.
            if getattr(self.slam, "model_exchange", None) is not None and (fs.shared_decoder or fs.overlap_peers):
                raise RuntimeError("a FusedStep with per-iteration collectives (shared_decoder / overlap_peers) must not run "
                                   "under the ModelExchange lock: serve fetches from a snapshot or drop the exchange")
        return fs

    def _device_ray_db(self, store):
        """Device copy of the keyframe ray store.  This repository's KeyFrameDatabase keeps its own mirror with dirty
        tracking; a HOST database (the reference's class, which only has the CPU tensor ``rays``) is mirrored here: a new
        tensor object (``del_keyframe`` re-creates it) is uploaded whole, in-place growth uploads the slots from the last
        previously live one onwards (``add_keyframe`` rewrites only slot counter-1)."""
        if hasattr(store, "device_rays"):
            return store.device_rays(self.device)
        live = len(store.frame_ids)
        tag = (id(store.rays), store.rays._version)
        st = self.__dict__.setdefault("_kf_mirror", {"tag": None, "live": 0, "dev": None})
        if st["dev"] is None or st["tag"] is None or st["tag"][0] != tag[0] or st["dev"].shape != store.rays.shape:
            st["dev"] = store.rays.to(self.device)
        elif st["tag"][1] != tag[1] or st["live"] != live:
            lo = max(min(st["live"], live) - 1, 0)
            st["dev"][lo:live] = store.rays[lo:live].to(self.device)
        st["tag"], st["live"] = tag, live
        return st["dev"]

    def _host_jitter(self, n_rays, fs):
        if self.sampler == "host" and self.config["training"]["perturb"] > 0.0:
            return torch.rand(n_rays, fs.S).to(self.device)           # the reference's CPU draw (scene_rep.py:381)
        return None

    def _map_guard(self):
        """Held while this agent's map is being updated: a peer's fetch (dist.ModelExchange) is served between updates
        only -- at the keyframe boundaries where the reference writes the checkpoint its peers read
        (mneslam_mp.py:294-315) -- and every loop below leaves with the fused step's streams joined (``fs.check()``)."""
        exchange = getattr(self.slam, "model_exchange", None)
        return exchange.lock if exchange is not None else _NoGuard()

    # ---------------------------------------------------------------- the two replaced loops
    def mapping_optimize(self, batch, poses):
        """``mapping.iters`` iterations over `sample` global keyframe rays + the current frame's share
        (reference: mp_slam/mapper.py:118-162); ``poses`` [N,4,4] c2w, the current frame's pose last."""
        if self.compute != "fused":
            return super().mapping_optimize(batch, poses)
        with self._map_guard():
            return self._mapping_optimize_fused(batch, poses)

    def _mapping_optimize_fused(self, batch, poses):
        cfg, store = self.config["mapping"], self.video.keyframe
        self.map_optimizer.zero_grad()
        n_kf, per_kf = len(store.frame_ids), store.num_rays_to_save
        n_glob, n_cur = cfg["sample"], max(cfg["sample"] // n_kf, cfg["min_pixels_cur"])
        fs = self._fused_step(n_glob + n_cur)
        db = self._device_ray_db(store)
        cur = frame_ray_table(batch).to(self.device, torch.float32).contiguous()
        poses = poses.to(self.device, torch.float32).contiguous()
        n_pix = self.slam.dataset.H * self.slam.dataset.W
        for it in range(cfg["iters"]):
            picks = (None, None)
            if self.sampler == "host":                # same two draws, same order as the reference's iteration
                picks = (torch.tensor(random.sample(range(n_kf * per_kf), n_glob)).to(self.device),
                         torch.tensor(random.sample(range(0, n_pix), n_cur)).to(self.device))
            fs.step(db, n_kf * per_kf, per_kf, cur, poses, n_glob, n_cur, idx_global=picks[0], idx_cur=picks[1],
                    u=self._host_jitter(n_glob + n_cur, fs), prefetch=it + 1 < cfg["iters"])
        fs.check()
        self.last_losses = fs.loss_dict()

    def first_frame_mapping(self, batch, n_iters=100):
        """Fused training on the first frame, then the host's own method with zero iterations for its bookkeeping."""
        if self.compute != "fused":
            return super().first_frame_mapping(batch, n_iters)
        if batch["frame_id"] != 0:
            raise ValueError("First frame mapping must be the first frame!")
        self.model.train()
        H, W, n = self.slam.dataset.H, self.slam.dataset.W, self.config["mapping"]["sample"]
        with self._map_guard():
            fs = self._fused_step(n)
            cur = frame_ray_table(batch).to(self.device, torch.float32).contiguous()
            pose = batch["c2w"].to(self.device).reshape(1, 4, 4).to(torch.float32).contiguous()
            for it in range(n_iters):
                pick = None
                if self.sampler == "host":
                    flat = self.slam.select_samples(H, W, n)
                    # the reference turns a flat draw into (h, w) = (flat % H, flat // H)  (mp_slam/mapper.py:76-77)
                    pick = ((flat % H) * W + torch.div(flat, H, rounding_mode="trunc")).to(self.device)
                fs.step(None, 0, 1, cur, pose, 0, n, idx_cur=pick, u=self._host_jitter(n, fs), prefetch=it + 1 < n_iters)
            fs.check()
            self.last_losses = fs.loss_dict()
        return super().first_frame_mapping(batch, 0)

    # ---------------------------------------------------------------- N2: loop-closure loops on the same kernels
    def optimize_relative_pose(self, base_c2w, target_c2w_initial, model_for_base, model_for_target, n_iters=None,
                               rays_d_cam_batch=None):
        """Pose alignment of loop closure (the loop inside ``handle_loop_closure``, mp_slam/mapper.py:362-412): the base
        model renders ``mapping.sample`` camera rays from ``base_c2w`` (teacher, 256 uniform samples, no gradient); the
        6 pose parameters of the target are then fitted so that the target model renders the same rgb / depth --
        ``render_rays`` with ray gradients (R13; planes get no gradient buffers), weighted MSE, the host's pose
        optimizer (``SLAM.get_pose_param_optim`` / ``SLAM.matrix_from_tensor``, mneslam_mp.py:577-584).
        The best pose is tracked on the device: no host synchronisation inside the loop.
        Returns (base_c2w @ inv(best target pose), best loss)."""
        cfg, dev = self.config, self.device
        w_rgb, w_depth = cfg["training"]["rgb_weight"], cfg["training"]["depth_weight"]
        steps = cfg["mapping"]["loop_iters"] if n_iters is None else n_iters
        base_c2w, start = base_c2w.to(dev), target_c2w_initial.to(dev)
        rot, trans, pose_opt = self.slam.get_pose_param_optim(start[None, ...], mapping=False)
        with torch.no_grad():
            if rays_d_cam_batch is None:
                pool = self.dataset.rays_d.reshape(-1, 3)
                rays_d_cam_batch = pool[torch.randint(0, len(pool), (cfg["mapping"]["sample"],))]
            dirs = rays_d_cam_batch.to(dev)
            n = dirs.shape[0]
            o_t, d_t, _, _ = world_rays(torch.cat([dirs, dirs.new_zeros(n, 4)], -1), base_c2w.expand(n, 4, 4))
            teacher = model_for_base.render_rays(o_t, d_t, target_d=None)
            want_rgb, want_depth = teacher["rgb"].detach(), teacher["depth"].detach()
        fused = self._pose_alignment_fused(model_for_target, dirs, want_rgb, want_depth, rot, trans, pose_opt, steps, w_rgb, w_depth) \
            if self.compute == "fused" else None
        self.last_pose_loop = "device" if fused is not None else "host"
        if fused is not None:
            best_pose, best = fused
            return base_c2w @ torch.inverse(best_pose), float(best)
        best = torch.full((), float("inf"), device=dev)
        best_pose = start.clone()
        for _ in range(steps):
            pose_opt.zero_grad()
            c2w = self.slam.matrix_from_tensor(rot, trans).squeeze(0)
            rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], dim=-1)
            out = model_for_target.render_rays(c2w[:3, 3].unsqueeze(0).repeat(n, 1), rays_d, target_d=None)
            loss = w_rgb * torch.nn.functional.mse_loss(out["rgb"], want_rgb) \
                + w_depth * torch.nn.functional.mse_loss(out["depth"], want_depth)
            with torch.no_grad():
                better = loss.detach() < best
                best = torch.where(better, loss.detach(), best)
                best_pose = torch.where(better, c2w.detach(), best_pose)
            loss.backward()
            pose_opt.step()
        return base_c2w @ torch.inverse(best_pose), float(best)

    def _pose_alignment_fused(self, model, dirs, want_rgb, want_depth, rot, trans, pose_opt, steps, w_rgb, w_depth):
        """The alignment loop as six launches per iteration (csrc/pose.hip: rays from the parameters, the render forward
        and backward with ray gradients, loss, analytic axis-angle Jacobian + Adam on the six parameters, best pose tracked
        on the device): no autograd graph, no torch.optim step, no host synchronisation inside the loop.  Used when the
        host's ``matrix_from_tensor`` IS an axis-angle map (probed: the reference's ``rot_rep: 'axis_angle'``,
        optimization/utils.py:161-197, or one relative to a constant rotation) and its optimizer is plain Adam over
        (rot, trans); returns None otherwise and the caller runs the host's own loop.  ``pose_opt`` itself is NOT stepped:
        its state (step counts, moments) stays as it was; the reference builds a fresh optimizer per alignment
        (mneslam_mp.py:577-584) and drops it afterwards."""
        from .. import hip_path
        if steps < 1 or not isinstance(pose_opt, torch.optim.Adam) or len(pose_opt.param_groups) != 2:
            return None
        if not getattr(model, "all_planes", None) and getattr(model, "embed_fn", None) is None:
            return None                                       # neither the plane encoding nor the hash / dense grid
        g_rot, g_trans = pose_opt.param_groups
        same = all(g_rot[k] == g_trans[k] for k in ("betas", "eps")) and not any(
            g.get("amsgrad") or g.get("weight_decay") or g.get("maximize") or g.get("capturable")
            or not isinstance(g["lr"], (int, float))                # tensor learning rates: the host's own loop handles them
            for g in (g_rot, g_trans))
        if not same or g_rot["params"][0] is not rot or g_trans["params"][0] is not trans:
            return None
        r_base = hip_path.probe_axis_angle(self.slam.matrix_from_tensor, rot, trans)
        if r_base is None:
            return None
        pa = hip_path.PoseAlignment(model, dirs, want_rgb, want_depth, rot, trans, r_base, g_rot["lr"], g_trans["lr"],
                                    g_rot["betas"], g_rot["eps"], w_rgb, w_depth)
        host_u = self.config["training"]["perturb"] > 0.0 and getattr(model, "jitter_rng", None) == "torch_cpu"
        for _ in range(steps):
            if host_u:
                pa.step(u=torch.rand(pa.n, pa.S).to(self.device))          # the reference's CPU draw, in its order
            else:
                u, so = model._jitter(pa.n, pa.S, dirs)
                pa.step(u=u, seed_offset=so)
        with torch.no_grad():                                              # the host's parameters end where the loop ended
            rot.copy_(pa.rot.reshape(rot.shape))
            trans.copy_(pa.trans.reshape(trans.shape))
        return pa.best()

    def load_foreign_model(self, other_rank):
        """mp_slam/mapper.py:708-726 -- called by the host's ``handle_loop_closure`` / ``bound_based_fusion`` (:340-360,
        :700-706).  When the SLAM object carries a running ``model_exchange`` (mneslam_amd.dist.ModelExchange: the agents are
        ranks of one process group) the peer's map is fetched from its device memory over the group (RCCL over xGMI) into
        ``model_shared``; otherwise the peer's ``latest_checkpoint.pt`` is read like the reference does."""
        exchange = getattr(self.slam, "model_exchange", None)
        if exchange is None:
            return super().load_foreign_model(other_rank)          # the host's own method: reads the peer's file
        from .. import slam_glue
        return slam_glue.load_foreign_model(self.model_shared, self.config, other_rank, self.device, exchange=exchange)

    def distillation(self, other_rank, expanded_foreign_kfs_for_distill, num_expanded_kfs):
        """Distil a foreign agent's map (``model_shared`` = teacher) into ``model`` (the training loop of
        mp_slam/mapper.py:594-644): every iteration draws camera rays at each foreign keyframe pose, the teacher renders
        them without depth guidance (no gradient), the student trains on the teacher's rgb / depth with the usual loss."""
        cfg = self.config
        n, floor = cfg["mapping"]["sample"], cfg["mapping"]["min_pixels_cur"]
        per_kf = max(n // num_expanded_kfs, floor) if num_expanded_kfs > 0 else n
        pool = self.dataset.rays_d.reshape(-1, 3)
        if self.compute == "fused" and expanded_foreign_kfs_for_distill:
            with self._map_guard():
                return self._distillation_fused(expanded_foreign_kfs_for_distill, per_kf, pool)
        for _ in range(cfg["mapping"]["distill_iters"]):
            pieces = []
            for kf in expanded_foreign_kfs_for_distill:
                pose = kf["pose"].to(self.device)
                dirs = pool[torch.randint(0, len(pool), (per_kf,))].to(self.device)
                rays_o = pose[:3, 3].unsqueeze(0).repeat(per_kf, 1)
                rays_d = torch.sum(dirs[..., None, :] * pose[:3, :3], dim=-1)
                with torch.no_grad():
                    t = self.model_shared.render_rays(rays_o, rays_d, target_d=None)
                pieces.append((rays_o, rays_d, t["rgb"].detach(), t["depth"].detach().unsqueeze(-1)))
            if not pieces:
                continue
            self.map_optimizer.zero_grad()
            ret = self.model.forward(*[torch.cat(col, 0) for col in zip(*pieces)])
            self.slam.get_loss_from_ret(ret, is_co_sdf=cfg["is_co_sdf"]).backward()
            self.map_optimizer.step()


    def _distillation_fused(self, kfs, per_kf, pool):
        """The same loop as device work only: per iteration ONE no-grad teacher launch sequence over the rays of all foreign
        keyframes (``render_maps``) and ONE fused student iteration (FusedStep: no autograd graph, no gradient tensors,
        no torch.optim step).  The batch reaches the fused step in the keyframe-database form it already understands:
        rows [camera direction, teacher rgb, teacher depth], ``per_kf`` rows per keyframe, row k belongs to pose k // per_kf.
        Host RNG draws (ray picks, jitter with ``jitter_rng == "torch_cpu"``) keep the reference's order."""
        cfg, dev, K = self.config, self.device, len(kfs)
        R = K * per_kf
        fs = self._fused_step(R)
        poses = torch.stack([kf["pose"].to(dev, torch.float32) for kf in kfs]).contiguous()          # [K,4,4]
        pool_dev = pool.to(dev, torch.float32)
        rows = torch.arange(R, device=dev)
        teacher, S_t = self.model_shared, cfg["training"]["n_samples"]
        host_u = cfg["training"]["perturb"] > 0.0 and getattr(teacher, "jitter_rng", None) == "torch_cpu"
        host_u_student = cfg["training"]["perturb"] > 0.0 and getattr(self.model, "jitter_rng", None) == "torch_cpu"
        rot, org = poses[:, None, :3, :3], poses[:, None, :3, 3]
        for it in range(cfg["mapping"]["distill_iters"]):
            picks, us = [], []
            for _ in range(K):                          # (reference order: pick, teacher jitter, next keyframe)
                picks.append(torch.randint(0, len(pool), (per_kf,)))
                if host_u:
                    us.append(torch.rand(per_kf, S_t))
            dirs = pool_dev[torch.stack(picks).to(dev)]                                            # [K,per_kf,3]
            rays_d = torch.sum(dirs[..., None, :] * rot, dim=-1).reshape(R, 3)
            rays_o = org.expand(K, per_kf, 3).reshape(R, 3)
            with torch.no_grad():
                t = teacher.render_maps(rays_o, rays_d, target_d=None, u=torch.cat(us).to(dev) if host_u else None)
            db = torch.cat([dirs.reshape(R, 3), t["rgb"], t["depth"].reshape(R, 1)], -1).contiguous()
            fs.step(db, R, per_kf, db, poses, R, 0, idx_global=rows,
                    u=torch.rand(R, fs.S).to(dev) if host_u_student else None)
        fs.check()
        self.last_losses = fs.loss_dict()


def bind(host_mapper_cls, compute="fused", sampler="host", scatter="binned", **fused_kwargs):
    """``class Mapper(FusedMappingMixin, host_mapper_cls)`` with the given options: the host's Mapper with its two
    training loops replaced by the fused iteration (everything else inherited unchanged)."""
    if compute not in ("autograd", "fused") or sampler not in ("host", "device"):
        raise ValueError("compute must be autograd|fused and sampler host|device")
    if sampler == "device" and compute != "fused":
        raise ValueError("the device sampler is part of the fused path")
    return type("Mapper", (FusedMappingMixin, host_mapper_cls),
                dict(compute=compute, sampler=sampler, scatter=scatter, fused_kwargs=dict(fused_kwargs),
                     optimize_map=FusedMappingMixin.mapping_optimize))


class _PlainMapper:
    """Minimal stand-in for a host Mapper (this repository has no host application): the reference's field names
    (mp_slam/mapper.py:12-50) and the plain iteration -- ``forward -> get_loss_from_ret -> backward -> step`` on the
    drop-in model -- for both loops, written around two small helpers."""

    def __init__(self, config, SLAM) -> None:
        self.config, self.slam = config, SLAM
        self.model, self.model_shared = SLAM.model, getattr(SLAM, "model_shared", None)
        self.map_optimizer = SLAM.map_optimizer
        self.device, self.dataset, self.video = SLAM.device, SLAM.dataset, SLAM.video
        self.rank, self.world_size = getattr(SLAM, "rank", 0), getattr(SLAM, "world_size", 1)

    def _train_on(self, rays_o, rays_d, target_rgb, target_d):
        ret = self.model.forward(rays_o, rays_d, target_rgb, target_d)
        self.slam.get_loss_from_ret(ret, is_co_sdf=self.config["is_co_sdf"]).backward()
        self.map_optimizer.step()
        return ret

    def load_foreign_model(self, other_rank):
        """mp_slam/mapper.py:708-726: the peer's ``latest_checkpoint.pt`` into ``model_shared``."""
        from .. import slam_glue
        return slam_glue.load_foreign_model(self.model_shared, self.config, other_rank, self.device)

    def first_frame_mapping(self, batch, n_iters=100):
        if batch["frame_id"] != 0:
            raise ValueError("First frame mapping must be the first frame!")
        self.model.train()
        H, W, n = self.slam.dataset.H, self.slam.dataset.W, self.config["mapping"]["sample"]
        c2w = batch["c2w"].to(self.device)
        table = frame_ray_table(batch)
        for _ in range(n_iters):
            self.map_optimizer.zero_grad()
            flat = self.slam.select_samples(H, W, n)
            rows = (flat % H) * W + torch.div(flat, H, rounding_mode="trunc")     # (h, w) = (flat % H, flat // H), :76-77
            self._train_on(*world_rays(table[rows].to(self.device), c2w.expand(n, 4, 4)))
        self.video.keyframe.add_keyframe(batch, 1, filter_depth=self.config["mapping"]["filter_depth"])   # :92

    def mapping_optimize(self, batch, poses):
        cfg, store = self.config["mapping"], self.video.keyframe
        self.map_optimizer.zero_grad()
        table = frame_ray_table(batch)
        n_pix = self.slam.dataset.H * self.slam.dataset.W
        for _ in range(cfg["iters"]):
            glob, owner = store.sample_global_rays(cfg["sample"])
            cur_rows = random.sample(range(0, n_pix), max(cfg["sample"] // len(store.frame_ids), cfg["min_pixels_cur"]))
            rays7 = torch.cat([glob, table[cur_rows]], 0).to(self.device)
            pose_row = torch.cat([owner.to(torch.int64), torch.full((len(cur_rows),), -1, dtype=torch.int64)])
            self._train_on(*world_rays(rays7, poses[pose_row.to(poses.device)]))     # id -1 = the current frame's pose
            self.map_optimizer.zero_grad()


class Mapper(FusedMappingMixin, _PlainMapper):
    """Stand-alone mapper of this repository: ``compute`` "autograd" (the plain iteration) or "fused";
    ``sampler`` "host" (seed-for-seed the reference's batches) or "device"."""

    def __init__(self, config, SLAM, compute="autograd", sampler="host", scatter="binned") -> None:
        if compute not in ("autograd", "fused") or sampler not in ("host", "device"):
            raise ValueError("compute must be autograd|fused and sampler host|device")
        if sampler == "device" and compute != "fused":
            raise ValueError("the device sampler is part of the fused path")
        super().__init__(config, SLAM)
        self.compute, self.sampler, self.scatter = compute, sampler, scatter
        self.fused_kwargs = {}

    optimize_map = FusedMappingMixin.mapping_optimize          # the name BASELINE.json uses
