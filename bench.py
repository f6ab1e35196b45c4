#!/usr/bin/env python3
"""bench.py -- mapping iterations / second of the MNE-SLAM mapping hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: spawns its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one mapping iteration (R1-R12): sample 2048 global keyframe rays
+ the current frame's share -> z sampling -> tri-plane/OneBlob/MLP/compositing forward -> losses ->
backward -> dense Adam over planes + decoder.  Workload = BASELINE.json configs[1] in its as-wired
form: Replica office0 tri-planes (0.02/0.01 m, 38.4 M fp32 params), 2x32 MLPs, 2048 rays x 128
samples (n_range_d 32 + n_samples_d 96), synthetic 1200x680 RGB-D frames, 20 keyframes.  All inputs
(keyframe ray database, current frame, poses, parameters) are resident in HBM before timing.

Multi-GPU: one agent per GPU (the reference's own decomposition, multi_agents.py:43-52), no data-path
collective (the reference has none); value = iterations of all agents / max-over-ranks time (weak).
"""
import argparse
import json
import math
import os
import random
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from mneslam_amd import configs, slam_glue, synthetic  # noqa: E402
from mneslam_amd.model.keyframe import KeyFrameDatabase  # noqa: E402
from mneslam_amd.fused import FusedStep, HashFusedStep  # noqa: E402
from mneslam_amd.model.scene_rep import JointEncoding  # noqa: E402
from mneslam_amd.model.scene_rep_hash import HashJointEncoding  # noqa: E402

PMC_PREFIX = "r06"                      # committed counter passes the line's `traffic` figures come from (profiles/r06_pmc.sh)
PMC_JSON = PMC_PREFIX + "_pmc_traffic.json"
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--keyframes", type=int, default=20)
    ap.add_argument("--config", default="office0", choices=sorted(configs.WORKLOADS),
                    help="workload (mneslam_amd/configs.py::WORKLOADS): office0 = BASELINE configs[1] as wired (the metric's "
                         "configuration, default); apartment / scannet / indoor = the single-agent shapes of configs[2..4]")
    ap.add_argument("--hidden", type=int, default=None, choices=[32, 64], help="decoder width (default: the workload's own)")
    ap.add_argument("--path", default="fused", choices=["fused", "autograd"],
                    help="fused = FusedStep + device sampler (default); autograd = the reference's call sequence")
    ap.add_argument("--cpu-iters", type=int, default=10,
                    help="timed oracle iterations for cpu_baseline (0 = skip); the median is reported, the loop stops after ~40 s")
    ap.add_argument("--graph", default=None, choices=["two_stream", "one_stream"],
                    help="EXTENSION (BASELINE configs[4]): replay the steady-state iteration as ONE captured HIP graph "
                         "(two_stream = the eager schedule's two queues captured, one_stream = one queue); default: eager launches")
    ap.add_argument("--scatter", default="binned", choices=["binned", "atomics"],
                    help="fused path: tile-binned LDS scatter fused with Adam (default) or global atomics + streaming Adam")
    ap.add_argument("--pretrain", type=int, default=10, help="--mode render_img: mapping iterations before the timed renders (an untrained "
                    "SDF has no zero crossing: no ray terminates early)")
    ap.add_argument("--no-overlap", action="store_true", help="run the plane update and the decoder chain on ONE stream (ablation)")
    ap.add_argument("--event-every", type=int, default=None,
                    help="bracket the launches with HIP events on every N-th timed step (default: 25, or one step in the "
                         "middle of a run shorter than 50 steps): a step with its 12 event records is ~60 us longer")
    ap.add_argument("--share-decoder", action="store_true",
                    help="EXTENSION: all-reduce (mean) the decoder gradient over agents each iteration (RCCL)")
    ap.add_argument("--split", action="store_true",
                    help="EXTENSION, BASELINE configs[2..4] as worded (--gpus N > 1, --config apartment | scannet | indoor): the N "
                         "agents map ONE scene -- rank r takes the r-th slab of it (0.5 m overlap with its neighbours, one "
                         "lattice), the overlap rectangles' plane gradients are exchanged (RCCL point-to-point) and the "
                         "decoder is shared (all-reduce) every iteration")
    ap.add_argument("--no-variants", dest="variants", action="store_false",
                    help="skip the short runs of the other section-8d workloads reported in the line's `variants` object")
    ap.add_argument("--rays", type=int, default=None, help="with --small: global rays per iteration (functional runs)")
    ap.add_argument("--small", action="store_true", help="tiny planes/frames (functional check, not a benchmark)")
    ap.add_argument("--mode", default="mapping", choices=["mapping", "render_img"],
                    help="mapping = the metric (default); render_img = SURVEY 8f row N1: full-frame no-grad renders, the "
                         "two per keyframe the reference's save_imgs does (with and without depth guidance)")
    return ap.parse_args()


class Agent:
    """One mapping agent: scene model + optimizer + device-resident keyframe rays."""

    def __init__(self, cfg, device, seed, n_keyframes, small=False, path="fused", scatter="binned", share_decoder=False,
                 overlap=True, graph=None, peers_of=None, model_seed=None, overlap_group_axis=None):
        """peers_of (split scenes): callable(model) -> FusedStep ``overlap_peers`` (collective: every rank calls it once its
        model exists); model_seed: the SAME decoder initialisation on every agent of a shared decoder."""
        self.cfg, self.device, self.path = cfg, device, path
        cam = synthetic.camera_from_config(cfg)          # office0: 680x1200, fx=fy=600, cx=599, cy=339
        if small:
            cam = dict(H=68, W=120, fx=60.0, fy=60.0, cx=59.0, cy=33.0)
        self.H, self.W = cam["H"], cam["W"]
        room = synthetic.room_from_config(cfg)           # office0: [[-2.2,2.6],[-3.4,2.1],[-1.4,2.0]]
        if small and peers_of is None:
            room = [[-0.8, 0.8], [-1.0, 0.9], [-0.6, 0.7]]
        random.seed(seed)
        torch.manual_seed(seed)
        frames = synthetic.make_frames(n_keyframes + 1, self.H, self.W, cam["fx"], cam["fy"], cam["cx"], cam["cy"],
                                       room, seed=seed)
        n_save = int(self.H * self.W * cfg["mapping"]["n_pixels"])
        self.kfdb = KeyFrameDatabase(cfg, self.H, self.W, n_keyframes + 1, n_save, device)
        for k in range(n_keyframes):
            self.kfdb.add_keyframe(frames[k], k + 1)
        self.n_kf, self.n_save = n_keyframes, n_save
        self.kf_rays = self.kfdb.device_rays(device)[:n_keyframes].reshape(-1, 7).contiguous()
        cur = frames[n_keyframes]
        self.cur_rays = torch.cat([cur["direction"], cur["rgb"], cur["depth"][..., None]], -1).reshape(-1, 7).to(device)
        self.poses = torch.stack([f["c2w"] for f in frames]).to(device)          # [n_kf+1,4,4]; last = current
        bb = torch.tensor(cfg["mapping"]["bound"], dtype=torch.float64, device=device)
        self.hash = cfg.get("scene_encoding") == "hash"
        if model_seed is not None:
            torch.manual_seed(model_seed)
        self.model = (HashJointEncoding if self.hash else JointEncoding)(cfg, bb).to(device).train()
        self.model.jitter_rng = "device"
        if share_decoder:
            # ONE decoder shared by the agents starts as ONE decoder: rank 0's initialisation (collective; every rank of a
            # shared-decoder run builds its agent at the same point)
            import torch.distributed as tdist
            if tdist.is_initialized() and tdist.get_world_size() > 1:
                for w in self.model.decoder.parameters():
                    tdist.broadcast(w.data, 0)
        peers = peers_of(self.model) if peers_of is not None else None     # (collective; fills the shared cells in place)
        self.opt = slam_glue.create_optimizer(self.model, cfg)
        self.n_cur = max(cfg["mapping"]["sample"] // n_keyframes, cfg["mapping"]["min_pixels_cur"])
        self.n_plane_params = (self.model.embed_fn.params.numel() if self.hash
                               else sum(p.numel() for lst in self.model.all_planes for p in lst))
        self.n_dec_params = sum(p.numel() for p in self.model.decoder.parameters())
        self.last = None
        self.fused = None
        if self.hash:
            if path != "fused":
                raise SystemExit("the hash-grid workload runs the fused step only")
            self.fused = HashFusedStep(self.model, self.opt, cfg, cfg["mapping"]["sample"] + self.n_cur, device)
            self.fused.seed = seed
        elif path == "fused":
            self.fused = FusedStep(self.model, self.opt, cfg, cfg["mapping"]["sample"] + self.n_cur, device,
                                   scatter=scatter, shared_decoder=share_decoder, use_graph=graph, overlap_peers=peers,
                                   overlap_group_axis=overlap_group_axis if peers else None,
                                   overlap=overlap and os.environ.get("MNE_NO_OVERLAP", "0") != "1")
            self.fused.seed = seed

    def sample_rays(self):
        """R1/R2 ray assembly on the device (mp_slam/mapper.py:135-153): `sample` rows without
        replacement over all stored keyframe rays + n_cur pixels of the current frame."""
        n = self.cfg["mapping"]["sample"]
        idx = torch.randperm(self.kf_rays.shape[0], device=self.device)[:n]
        ids = torch.div(idx, self.n_save, rounding_mode="trunc")
        idc = torch.randperm(self.cur_rays.shape[0], device=self.device)[:self.n_cur]
        rays = torch.cat([self.kf_rays[idx], self.cur_rays[idc]], 0)
        ids_all = torch.cat([ids, torch.full((self.n_cur,), self.n_kf, device=self.device, dtype=ids.dtype)])
        rot = self.poses[ids_all, :3, :3]
        rays_d = torch.sum(rays[:, None, :3] * rot, -1)
        rays_o = self.poses[ids_all, :3, 3]
        return rays_o, rays_d, rays[:, 3:6], rays[:, 6:7]

    def step(self, timers=None, prefetch=False):
        if self.fused is not None:
            self.fused.events = timers
            self.fused.step(self.kf_rays, self.kf_rays.shape[0], self.n_save, self.cur_rays, self.poses,
                            self.cfg["mapping"]["sample"], self.n_cur, prefetch=prefetch)
            return
        rays_o, rays_d, tgt_rgb, tgt_d = self.sample_rays()
        ret = self.model.forward(rays_o, rays_d, tgt_rgb, tgt_d)
        loss = slam_glue.get_loss_from_ret(self.cfg, ret, is_co_sdf=self.cfg["is_co_sdf"])
        loss.backward()
        if timers is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self.opt.step(zero_grad=False)
        if timers is not None:
            e1.record()
            timers.setdefault("adam", []).append((e0, e1))
        self.opt.zero_grad(set_to_none=True)
        self.last = (ret, tgt_rgb, tgt_d)

    def quality(self):
        if self.fused is not None:
            f = self.fused
            f.check()                                   # no list entry was dropped during the run
            ret, tgt_rgb, tgt_d = f.loss_dict(), f.tgt_rgb, f.tgt_d[:, None]
        else:
            ret, tgt_rgb, tgt_d = self.last
        d = tgt_d.squeeze(-1)
        valid = (d > 0) & (d < self.cfg["cam"]["depth_trunc"])
        l1 = (ret["depth"].detach()[valid] - d[valid]).abs().mean()
        return float(ret["psnr"].detach()[0]), float(l1)


def cpu_baseline(cfg, n_keyframes, iters, seed=0, batch=None, cores=None):
    """The oracle (CPU restatement of the reference's PyTorch path; checker code, reported baseline
    only) timed on this box's host cores on the same workload shape -- with ``batch`` = (rays_o, rays_d, rgb, depth, z_vals)
    on the very batch the device drew in its last timed iteration."""
    from oracle import mapping as omap
    from oracle.scene_rep import OracleScene
    host_cores = os.cpu_count() or 1
    try:
        host_cores_usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        host_cores_usable = host_cores
    # 32 threads: more only add contention on the scatter-heavy backward (profiles/r05_cpu_threads.txt: 32 / 64 / 128 threads)
    cores = min(host_cores_usable, 32) if cores is None else int(cores)
    torch.set_num_threads(cores)
    gen = torch.Generator().manual_seed(seed)
    bb = torch.tensor(cfg["mapping"]["bound"], dtype=torch.float64)
    if cfg.get("scene_encoding") == "hash":
        from oracle import hashgrid
        from oracle.scene_rep import OracleHashScene, make_decoder_weights
        grid = dict(n_levels=16, n_features=2, base_resolution=16, log2_hashmap_size=cfg["grid"]["hash_size"],
                    per_level_scale=float(2.0 ** (math.log2(cfg["grid"].get("desired_resolution", 512) / 16) / 15)))
        table = (torch.rand(hashgrid.n_params(**grid), generator=gen) * 2 - 1) * 1e-4
        sc = OracleHashScene(cfg, bb, table, grid)
        cfg_dec = dict(cfg, model=dict(cfg["model"], input_ch=64))
        sc.sdf_w, sc.col_w = make_decoder_weights(type(sc.pc).from_dict(cfg_dec), gen)
        sc.requires_grad_(True)
    else:
        sc = OracleScene(cfg, bb, generator=gen).requires_grad_(True)
    opt = omap.OracleAdam(sc, cfg)
    n = cfg["mapping"]["sample"] + max(cfg["mapping"]["sample"] // n_keyframes, cfg["mapping"]["min_pixels_cur"])
    z_vals = None
    if batch is not None:
        rays_o, rays_d, rgb, dep, z_vals = [t.detach().to("cpu", torch.float32) for t in batch]
        dep, n = dep.reshape(-1, 1), rays_o.shape[0]
    else:
        frames = synthetic.make_frames(1, 68, 120, 60.0, 60.0, 59.0, 33.0, synthetic.OFFICE0_ROOM, seed=seed)
        fr = frames[0]
        idx = torch.randint(0, 68 * 120, (n,), generator=gen)
        d_cam = fr["direction"].reshape(-1, 3)[idx]
        rays_d = torch.sum(d_cam[:, None, :] * fr["c2w"][:3, :3], -1)
        rays_o = fr["c2w"][None, :3, 3].repeat(n, 1)
        rgb, dep = fr["rgb"].reshape(-1, 3)[idx], fr["depth"].reshape(-1, 1)[idx]
    times = []
    t_start = time.perf_counter()
    for it in range(iters + 1):
        if it >= 4 and time.perf_counter() - t_start > 40.0:     # keep the default run bounded (>= 3 timed iterations)
            break
        t0 = time.perf_counter()
        opt.zero_grad()
        ret = sc.forward(rays_o, rays_d, rgb, dep, impl="grid_sample", z_vals=z_vals) if z_vals is not None \
            else sc.forward(rays_o, rays_d, rgb, dep, impl="grid_sample")
        omap.loss_from_ret(cfg, ret, is_co_sdf=cfg["is_co_sdf"]).backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    timed = sorted(times[1:])
    t = timed[len(timed) // 2] if len(timed) % 2 else 0.5 * (timed[len(timed) // 2 - 1] + timed[len(timed) // 2])     # median
    return {"value": 1.0 / t, "unit": "it/s", "cores": cores, "kind": "port",
            "host_cores": host_cores, "host_cores_usable": host_cores_usable,
            "s_per_iter_min_median_max": [timed[0], t, timed[-1]],
            "sample": f"median of {len(timed)} mapping iterations (1 warm-up) "
                      + ("on the batch the device drew in its last timed iteration " if batch is not None else "of the same workload ")
                      + f"({n} rays x "
                      f"{cfg['training']['n_range_d'] + cfg['training']['n_samples_d']} samples, "
                      f"{sum(p.numel() for p in sc.plane_list())} plane params) with the CPU oracle, torch "
                      f"{torch.__version__}, {cores} threads"}


MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X fp32-input MFMA (v_mfma_f32_32x32x2_f32) = the fp32 vector rate (MI355X_MICROARCH.md)


def render_img_measure(agent, cfg, device, n_pairs, n_warm):
    """N1: JointEncoding.render_img on whole frames -- one no-grad launch sequence per frame with exact early ray
    termination.  One step = the reference's per-keyframe pair (mneslam_mp.py:498,516): a depth-guided render (n_range_d +
    n_samples_d samples per ray) and a free render (training.n_samples).  Returns the record without the rate's
    normalisation over ranks: (seconds for n_pairs pairs, record dict)."""
    m = agent.model
    was_training = m.training
    m.eval()
    if os.environ.get("MNE_RENDER_PATCH_ORDER") == "0":       # (A/B of the Z-order ray walk of render_img, profiles/r05_render_img.sh)
        m.render_patch_order = False
    cam = synthetic.camera_from_config(cfg)
    cfg_cam_backup = dict(cfg["cam"])
    cfg["cam"].update(cam, crop_edge=0)                  # render_img reads the camera from the config
    frames = synthetic.make_frames(2, cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"],
                                   synthetic.room_from_config(cfg), seed=99)
    c2w, gt = frames[1]["c2w"].to(device), frames[1]["depth"].to(device)
    tr = cfg["training"]
    has_free = bool(tr.get("n_samples"))
    n_rays = cam["H"] * cam["W"]
    pts = n_rays * ((tr["n_range_d"] + tr["n_samples_d"]) + (tr["n_samples"] if has_free else 0))

    def pair(stats=None):
        d1, c1 = m.render_img(c2w, device, gt_depth=gt, stats=stats)
        if has_free:
            m.render_img(c2w, device, gt_depth=None, stats=stats)
        return d1

    for _ in range(n_warm):
        pair()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_pairs):
        d1 = pair()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0
    stats = {}
    pair(stats)                                          # one more, untimed: the samples the early termination really decodes
    cfg["cam"].update(cfg_cam_backup)
    if was_training:
        m.train()
    d = cfg["decoder"]
    hid, hidc = d["hidden_dim"], d["hidden_dim_color"]
    cin = 48 + (0 if cfg["grid"]["oneGrid"] else 64) + d["geo_feat_dim"]
    flop_per_sample = 2.0 * (hid * 112 + 16 * hid + hidc * cin + 3 * hidc)       # the two bias-free 2-layer MLPs (model/decoder.py:110-175)
    flops = flop_per_sample * stats["decoded_samples"]
    rec = {"frame": f"{cam['W']}x{cam['H']}", "rays_per_frame": n_rays, "nominal_point_queries_per_pair": pts,
           "decoded_samples_per_pair": stats["decoded_samples"], "early_ray_termination": True,
           "depth_l1_vs_gt": float((d1.float() - gt)[gt > 0].abs().mean()),
           "roofline": {"kernel": "decode_frame_kernel + ray_frame_kernel (inline tri-plane gather + OneBlob + MLP forward + compositing, early termination)",
                        "bound": "mfma_f32", "flops_per_pair": flops, "flop_per_decoded_sample": flop_per_sample,
                        "achieved": flops / (own / n_pairs) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / (own / n_pairs) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                        "note": "useful MLP flops of the decoded samples only; the kernels are bound by what their waves issue (120 dependent "
                                "MFMAs + ~1500 VALU per 32-sample tile), not by HBM or the matrix pipe: DESIGN.md 3.6"}}
    return own, rec


def pose_alignment_measure(agent, cfg, device, n_iters=None):
    """Row R13 / N2 on the record: the pose-alignment loop of loop closure (mp_slam/mapper.py:362-412) as the device
    loop runs it -- ``mapping.loop_iters`` (100) iterations of rays-from-parameters -> z samples -> render forward -> loss -> render
    backward WITH RAY GRADIENTS (ray_kernel<..., 3>) -> analytic Jacobian + Adam on the six pose parameters, ``mapping.sample`` rays x
    ``training.n_samples`` uniform samples, the teacher's maps rendered by the same model from the true pose, the start pose 5 cm and
    ~1.7 degrees off.  No autograd graph, no host synchronisation inside the loop."""
    from mneslam_amd import hip_path
    model, n = agent.model, cfg["mapping"]["sample"]
    steps = cfg["mapping"]["loop_iters"] if n_iters is None else n_iters
    g = torch.Generator(device="cpu").manual_seed(5)
    dirs = agent.cur_rays[torch.randint(0, agent.cur_rays.shape[0], (n,), generator=g).to(device), :3].contiguous()
    base = agent.poses[0]
    with torch.no_grad():
        rays_d = torch.sum(dirs[:, None, :] * base[:3, :3], -1)
        teacher = model.render_rays(base[:3, 3].expand(n, 3).contiguous(), rays_d, target_d=None)
    want_rgb, want_depth = teacher["rgb"].detach(), teacher["depth"].detach()
    rot0 = torch.tensor([0.03, 0.0, 0.0], device=device)                 # axis-angle relative to the true rotation
    trans0 = base[:3, 3] + torch.tensor([0.05, 0.0, 0.0], device=device)
    pa = hip_path.PoseAlignment(model, dirs, want_rgb, want_depth, rot0, trans0, base[:3, :3].contiguous().cpu(), 1e-3, 1e-3,
                                (0.9, 0.999), 1e-8, cfg["training"]["rgb_weight"], cfg["training"]["depth_weight"])
    for k in range(3):                                                    # warm-up (first launches of these instantiations)
        pa.step(seed_offset=(7, k * ((pa.n * pa.S + 3) // 4)))
    torch.cuda.synchronize()
    first = float(pa.last_loss[0])
    t0 = time.perf_counter()
    for k in range(steps):
        pa.step(seed_offset=(7, (k + 3) * ((pa.n * pa.S + 3) // 4)))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"iterations": steps, "rays": n, "samples_per_ray": pa.S, "total_ms": 1e3 * el, "ms_per_iteration": 1e3 * el / steps,
            "loss_start": first, "loss_best": float(pa.best_loss[0]), "loss_last": float(pa.last_loss[0]),
            "launches_per_iteration": 6, "kernels": "mne_pose_rays, mne_sample_z, mne_render_forward, mne_pose_loss, "
            "mne_render_backward (ray gradients: ray_kernel<..., 3>), mne_pose_update"}


def bench_render_img(args, cfg, workload, agent, device, rank, world, barrier, mdist):
    """--mode render_img: the N1 record as the line's metric (see render_img_measure)."""
    for _ in range(args.pretrain):
        agent.step()                                     # mapping iterations in front of the renders: the SDF needs a surface
    torch.cuda.synchronize()
    steps, warm = max(args.steps // 20, 3), max(args.warmup // 10, 1)
    barrier()
    own, rec = render_img_measure(agent, cfg, device, steps, warm)
    barrier()
    elapsed = mdist.max_over_ranks(own, device)
    if rank == 0:
        roof = rec.pop("roofline")
        print(json.dumps({
            "metric": "full-frame renders (render_img pairs) / sec", "value": world * steps / elapsed, "unit": "frame pairs/s",
            "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": dict({"workload": workload + "_render_img", "pretrain_iterations": args.pretrain}, **rec),
            "nominal_Mpts_per_s": world * steps * rec["nominal_point_queries_per_pair"] / elapsed / 1e6,
            "roofline": roof}), flush=True)


def account(cfg, agent, avg_ms):
    """Algorithmic bytes per launch of the path's kernels and the dominant one among the live-measured launches."""
    S = cfg["training"]["n_range_d"] + cfg["training"]["n_samples_d"]
    R = cfg["mapping"]["sample"] + agent.n_cur
    n_par = agent.n_plane_params + agent.n_dec_params
    # Algorithmic bytes per launch = SURVEY.md section 8d's per-unit figures x the units the launch processes (both stated
    # in DESIGN.md):  G = planes x 4 corners x 32 ch x 4 B per point per gather or scatter pass; 32 B per parameter
    # per optimiser sweep (the reference's read p,g,m,v + write p,m,v + zero g).
    #   tile_adam_kernel : scatter of the P' contributing samples + the sweep over every parameter
    #   gather_kernel    : gather of the D samples the exact early termination decodes; D is counted from below as
    #                      the samples of the tiles the backward walks (ray_tiles, read back live), D <= R*S
    #   decode_kernel, ray_kernel : the MLP forward / composite+backward; no algorithmic HBM bytes (latency-bound)
    #   atomics variant  : adam_kernel = the sweep; the render call gathers and scatters (atomics)
    G = 3072.0 * (1 if cfg["grid"]["oneGrid"] else 2)
    half = cfg["grid"].get("plane_dtype", "fp32") == "fp16"
    G_gather = G / 2 if half else G           # fp16 plane storage: 64-byte corner rows
    sweep = 28.0 if half else 32.0            # ... and 2 bytes less for the parameter's read and for its write (8d: 16 + 12 + 4)
    binned = agent.fused is not None and agent.fused.bins is not None
    p_contrib = float(agent.fused.tape_rows.item()) if agent.fused is not None else float(R * S)
    decoded = float((agent.fused.ray_tiles[:R].long() * 32).clamp(max=S).sum().item()) if agent.fused is not None else float(R * S)
    # decoder weight-gradient pass: the tape columns it reads of every backward row -- x 112 | out 16 | h | hc | dh | dhc | dout 16 | dc 4
    # (+ the 64 colour-plane features); a streaming read, MFMA work hidden behind it (wgrad.hip)
    hid = cfg["decoder"]["hidden_dim"]
    wgrad_bytes = decoded * 4.0 * (112 + 16 + 4 * hid + 16 + 4 + (0 if cfg["grid"]["oneGrid"] else 64))
    if agent.hash:
        # hash grid: 16 levels x 8 corners x 2 features x 4 B = 1,024 B per point per gather or scatter pass
        Gh = agent.model.embed_fn.cfg.n_levels * 8 * agent.model.embed_fn.cfg.n_features * 4.0
        n_gather = decoded if agent.fused.early_termination else float(R * S)       # rows the gather fills (counted from below)
        alg = {"hash_gather": n_gather * Gh, "hash_scatter": p_contrib * Gh, "adam": 32.0 * n_par, "render": 0.0}
        slices = agent.fused.table_update != "atomics"
        if slices:       # the table update = scatter + Adam sweep of the table in one call
            alg["hash_scatter"] += 32.0 * agent.n_plane_params
            alg["adam"] = 32.0 * agent.n_dec_params
        kern = {"hash_gather": "hash_gather_kernel (grid gather into the tape; the next batch's rows are gathered behind the table update)",
                "hash_scatter": ("hash_bin + hash_slice_adam + hash_finish kernels (slice-binned rows, exact fixed-point LDS sums, Adam fused; one mne_hash_slice_adam call)"
                                 if slices else "hash_scatter_runs_kernel (run-reduced global atomics)"),
                "adam": "adam_kernel (decoder)" if slices else "adam_kernel (table + decoder, one launch)", "wgrad": "wgrad kernels",
                "decode_kernel": "decode_kernel", "ray_kernel": "ray_kernel (composite + loss + backward)",
                "deferred_pass": "hash gather + decode_kernel + ray_kernel over the deferred-ray list",
                "render": "whole mne_render_fused_features call (decode + ray kernels; MFMA / latency, no algorithmic HBM bytes)"}
        alg["wgrad"] = wgrad_bytes
        alg["iteration"] = alg["hash_gather"] + alg["hash_scatter"] + alg["adam"]
    elif binned:
        # parameters the plane update really sweeps: tiles that have never received a gradient keep m = v = 0, Adam leaves them
        # bit-for-bit unchanged and the kernel skips them (tile_adam.hip) -- they carry no algorithmic bytes either
        swept = float(agent.n_plane_params)
        live = getattr(agent.fused, "tile_live", None)
        if live is not None:
            lv, off_, swept = live.cpu(), 0, 0.0
            for p_ in agent.fused.planes:
                t_ = ((p_.shape[2] + 15) // 16) * ((p_.shape[3] + 15) // 16)
                swept += p_.numel() * float((lv[off_:off_ + t_] != 0).float().mean())
                off_ += t_
        alg = {"adam": p_contrib * G + sweep * swept + 32.0 * agent.n_dec_params, "_swept": swept,
               "gather_kernel": decoded * G_gather, "render": decoded * G_gather, "wgrad": wgrad_bytes}
        kern = {"adam": "tile_adam_kernel (binned scatter + Adam, one launch)", "gather_kernel": "gather_kernel",
                "decode_kernel": "decode_kernel", "ray_kernel": "ray_kernel (composite + loss + backward)",
                "deferred_pass": "decode_kernel + ray_kernel over the deferred-ray list",
                "bin_kernel": "bin_kernel (list appends of the binned plane update)",
                "render": "whole mne_render_fused call (gather + decode + ray + deferred pass + bin)"}
        # Fused gather + decode launch (scenes without colour planes since round 6, csrc/render.hip launch_render): no gather_kernel runs
        # between its two timing marks (a few us of event spacing remain) -- the gather's algorithmic bytes are the decode launch's
        if "decode_kernel" in avg_ms and avg_ms.get("gather_kernel", 1.0) < 0.02:
            avg_ms.pop("gather_kernel")
            alg["decode_kernel"] = alg.pop("gather_kernel")
            kern["decode_kernel"] = "decode_kernel (fused launch: tri-plane gather inline + OneBlob + MLP)"
            kern["render"] = "whole mne_render_fused call (fused gather + decode, ray, deferred pass, bin)"
    else:
        alg = {"adam": 32.0 * n_par, "render": (decoded + p_contrib) * G}
        kern = {"adam": "adam_kernel (planes + decoder, one launch)",
                "render": "whole mne_render_fused call (gather + decode + ray kernels, atomic scatter)"}
    # the dominant KERNEL = the longest live-measured single launch (the bracket around the whole render call is
    # reported beside it, not as a kernel, when its kernels are timed individually)
    single = {k: v for k, v in avg_ms.items() if not (k == "render" and ("gather_kernel" in avg_ms or "decode_kernel" in avg_ms or agent.hash))}
    dom = max(single, key=single.get) if single else "adam"
    dom_ms = avg_ms.get(dom, 0.0)
    achieved = alg.get(dom, 0.0) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    return {"alg": alg, "kern": kern, "dom": dom, "dom_ms": dom_ms, "achieved": achieved, "p_contrib": p_contrib,
            "decoded": decoded, "R": R, "S": S, "swept": alg.pop("_swept", None)}


# (name, workload, hidden, graph).  indoor_fp16 / indoor_fp16_graph = BASELINE configs[4] as worded (one of its agents): fp16
# feature storage + fp32 accumulate, eager and as a hipGraph-captured iteration
VARIANTS = (("office0_2x64", "office0", 64, None), ("office0_hash", "office0_hash", None, None), ("scannet", "scannet", None, None),
            ("indoor", "indoor", None, None), ("indoor_fp16", "indoor_fp16", None, None),
            ("indoor_fp16_graph", "indoor_fp16", None, "one_stream"), ("office0_fp16", "office0_fp16", None, None))


def pmc_traffic(tag, needle):
    """HBM bytes per iteration of the kernels whose name contains ``needle``, from the committed counter passes of workload
    ``tag`` (profiles/r04_pmc_traffic_<tag>.json, profiles/r04_pmc.sh; the hash grid: r04_hash_pmc_traffic.json, per launch)."""
    try:
        if tag == "office0_hash":
            per_k = json.load(open(os.path.join(REPO, "profiles", f"{PMC_PREFIX}_hash_pmc_traffic.json")))["per_kernel_hbm_bytes"]
        else:
            per_k = json.load(open(os.path.join(REPO, "profiles", f"{PMC_PREFIX}_pmc_traffic_{tag}.json")))["per_kernel_hbm_bytes_per_iteration"]
        return sum(v for k, v in per_k.items() if needle in k) or None
    except (OSError, KeyError, ValueError):
        return None


def run_variant(config, hidden, device, keyframes, budget_s=1.5, warmup=30, block=50, graph=None, name=None):
    """A short (<= ~2 s of device time) run of another workload of SURVEY.md section 8d through the same step: the other
    decoder width, the hash-grid headline encoding, the ScanNet / indoor-scale plane sets.  Reported beside the metric,
    never as it."""
    make_cfg, workload = configs.WORKLOADS[config]
    cfg = make_cfg(hidden) if hidden else make_cfg()
    agent = Agent(cfg, device, seed=0, n_keyframes=keyframes, graph=graph)
    for _ in range(warmup):
        agent.step(prefetch=graph is not None)
    timers, steps = {}, 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while True:
        for i in range(block):
            agent.step(timers if (i % 10 == 0 and graph is None) else None, prefetch=True)
        steps += block
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if elapsed > budget_s or steps >= 2000:
            break
    if graph is not None:
        # a replayed graph carries no per-kernel events: the RATE above is the replay's; the dominant kernel is timed in a few
        # eager iterations of the same agent (a step that is given timers runs its launches eagerly: same kernels, same state)
        for i in range(20):
            agent.step(timers if i % 2 == 0 else None, prefetch=True)
        torch.cuda.synchronize()
    avg_ms = {k: sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1) for k, v in timers.items()}
    acc = account(cfg, agent, avg_ms)
    out = {"workload": workload, "mlp_hidden": cfg["decoder"]["hidden_dim"], "value": steps / elapsed, "unit": "it/s",
           "plane_dtype": cfg["grid"].get("plane_dtype", "fp32"), "launch": ("hipGraph replay (" + graph + ")") if graph else "eager",
           "ms_per_step": 1e3 * elapsed / steps, "steps": steps, "warmup": warmup,
           "plane_params": agent.n_plane_params, "rays_per_iter": acc["R"], "samples_per_ray": acc["S"],
           "dominant_kernel": acc["kern"][acc["dom"]], "avg_launch_ms": acc["dom_ms"],
           "achieved_GBs": acc["achieved"], "frac": acc["achieved"] / HBM_PEAK_GBS}
    if graph is not None:
        out["kernel_timing"] = "eager iterations of the same agent (events cannot be recorded inside a replayed graph)"
    needle = {"adam": "tile_adam_kernel", "hash_scatter": "hash_slice_adam_kernel", "hash_gather": "hash_gather_kernel"}.get(acc["dom"], acc["dom"])
    tag = name[:-len("_graph")] if (graph and name and name.endswith("_graph")) else name
    out["traffic"] = pmc_traffic(tag, needle) if name else None
    out["traffic_source"] = (f"profiles/{PMC_PREFIX}_hash_pmc_traffic.json" if tag == "office0_hash" else f"profiles/{PMC_PREFIX}_pmc_traffic_{tag}.json") if out["traffic"] else None
    out["traffic_frac"] = (out["traffic"] / (acc["dom_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if (out["traffic"] and acc["dom_ms"] > 0) else None
    it_bytes = acc["alg"].get("iteration", acc["alg"].get("adam", 0.0) + acc["alg"].get("render", 0.0))
    out["iteration_algorithmic_bytes"] = it_bytes
    out["iteration_hbm_frac"] = it_bytes / (elapsed / steps) / 1e9 / HBM_PEAK_GBS
    del agent
    torch.cuda.empty_cache()
    return out


def split_peers(rank, world, device):
    """peers_of callback of a split scene: the ranks exchange their planes' geometry, the lower neighbour's values of the
    shared node rectangles are copied into the upper one's planes (shared cells start equal, and stay bit-equal: both add the
    same two gradient shares, tile_adam_kernel<2>), and the neighbours come back as FusedStep's overlap_peers."""
    import torch.distributed as dist
    from mneslam_amd import dist as mdist

    def peers_of(model):
        geo = mdist.plane_geometry(model)
        geos = [None] * world
        dist.all_gather_object(geos, geo)
        flat = [p for lst in model.all_planes for p in lst]
        for lower in range(world - 1):                       # startup only: pair after pair
            upper = lower + 1
            if rank not in (lower, upper):
                continue
            for p, (shape, bnd, axes), (pshape, pbnd, _) in zip(flat, geo, geos[upper if rank == lower else lower]):
                sl = mdist.overlap_slices(bnd, pbnd, shape, pshape, axes)
                if sl is None:
                    continue
                (ys, xs), _ = sl
                buf = p.data[:, :, ys, xs].contiguous()
                if rank == lower:
                    dist.send(buf, upper)
                else:
                    dist.recv(buf, lower)
                    p.data[:, :, ys, xs] = buf
        return [(r, geos[r]) for r in (rank - 1, rank + 1) if 0 <= r < world]
    return peers_of


# BASELINE configs[2..4] by agent count: "apartment split into 2 agents", "scene0000 split 4-way", "INS Indoor 8-agent"
AS_WORDED = {2: "apartment", 4: "scannet", 8: "indoor"}


def make_split_agent(config, rank, world, device, keyframes, small=False, rays=None):
    """Agent ``rank`` of the scene ``config`` split over ``world`` agents (configs.split_agent_config): its slab of the scene
    on the common lattice, the neighbours as overlap peers, one decoder shared by all.  Collective."""
    make_cfg, workload = configs.WORKLOADS[config]
    cfg = make_cfg()
    if small:
        cfg["mapping"]["bound"] = [[-1.0, 1.0], [-1.2, 1.1], [-0.8, 0.9]]
        cfg["planes_res"] = {"coarse": 0.1, "fine": 0.05, "bound_dividable": 0.1}
        cfg["c_planes_res"] = {"coarse": 0.2, "fine": 0.1}       # (colour planes, ScanNet: resolutions that nest with the geometry planes')
        if rays:
            cfg["mapping"]["sample"] = rays
            cfg["mapping"]["min_pixels_cur"] = min(cfg["mapping"]["min_pixels_cur"], max(rays // 4, 1))
    else:
        cfg["mapping"]["bound"] = [list(b) for b in configs.SCENE_BOUNDS[config]]       # the WHOLE scene, then this rank's slab
    cfg, axis, slabs = configs.split_agent_config(cfg, world, rank)
    # more than two agents: the planes that do not contain the slab axis are held by every agent -- their gradient is the sum over
    # ALL agents (one all-reduce), not pairwise with the neighbours (FusedStep(overlap_group_axis))
    agent = Agent(cfg, device, seed=rank, n_keyframes=keyframes, small=small, share_decoder=True,
                  peers_of=split_peers(rank, world, device), model_seed=1234, overlap_group_axis=axis if world > 2 else None)
    return agent, cfg, {"axis": "xyz"[axis], "slabs": slabs, "workload": workload + f"_scene_split{world}"}


def multi_agent_side_records(args, cfg, rank, world, device, barrier, out):
    """N > 1 side records of the metric's line (which runs N independent agents, the reference's own decomposition):
      share_decoder  the same workload with ONE decoder shared by the agents (all-reduce of its weight gradients per iteration);
      as_worded      BASELINE configs[2] / [3] / [4] as worded for this agent count -- one scene split into N overlapping slabs on
                     one lattice, overlap-rectangle plane gradients exchanged with the neighbours (RCCL point-to-point) + the
                     shared decoder.
    Collective code that has never run on more than one GPU before the driver's first multi-GPU lease: a watchdog prints the
    metric's line (``out``, rank 0) with what is there and leaves if a record does not come back -- a side record must never
    cost the line."""
    import signal
    import torch.distributed as dist
    from mneslam_amd import dist as mdist
    side = {}

    def give_up(signum, frame):
        if rank == 0:
            side.setdefault("error", "a multi-agent side record did not return within its time limit")
            out["variants"] = side
            print(json.dumps(out), flush=True)
        os._exit(0)
    signal.signal(signal.SIGALRM, give_up)
    signal.alarm(int(os.environ.get("MNE_SIDE_RECORD_LIMIT_S", "240")))

    def timed(agent, n_var, warm):
        for _ in range(warm):
            agent.step()
        barrier()
        t1 = time.perf_counter()
        for i in range(n_var):
            agent.step(None, prefetch=i + 1 < n_var)
        barrier()
        own = time.perf_counter() - t1
        return own, mdist.max_over_ranks(own, device)

    n_var, warm = max(min(args.steps, 200), 1), min(args.warmup, 10)
    try:
        shared = Agent(cfg, device, seed=rank, n_keyframes=args.keyframes, small=args.small, path=args.path, scatter=args.scatter,
                       share_decoder=True, overlap=not args.no_overlap, graph=None)
        _, el = timed(shared, n_var, warm)
        side["share_decoder"] = {"value": world * n_var / el, "unit": "it/s", "ms_per_step": 1e3 * el / n_var, "steps": n_var,
                                 "collective": "all-reduce of the decoder's weight gradients every iteration ("
                                               + str(dist.get_backend()) + "), planes private to each agent"}
        del shared
    except Exception as e:    # noqa: BLE001
        side["share_decoder"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if args.path == "fused" and args.scatter == "binned":
        name = AS_WORDED.get(world, "apartment")
        try:
            if device.type == "cuda":
                torch.cuda.empty_cache()
            agent, scfg, info = make_split_agent(name, rank, world, device, args.keyframes, small=args.small, rays=args.rays)
            own, el = timed(agent, n_var, warm)
            psnr, l1 = agent.quality()
            rows = [None] * world
            dist.all_gather_object(rows, {"rank": rank, "it_per_s": n_var / own, "plane_params": agent.n_plane_params,
                                          "overlap_exchange_bytes_per_iter": 4 * 2 * sum(t.numel() for t in agent.fused.ov_send),
                                          "psnr_last_iter": psnr, "depth_l1_last_iter": l1})
            S = scfg["training"]["n_range_d"] + scfg["training"]["n_samples_d"]
            side["as_worded"] = {
                "baseline_config": {2: "configs[2]", 4: "configs[3]", 8: "configs[4]"}.get(world, "configs[2] (its scene, this agent count)"),
                "workload": info["workload"], "value": world * n_var / el, "unit": "it/s", "ms_per_step": 1e3 * el / n_var, "steps": n_var,
                "rays_per_iter": scfg["mapping"]["sample"] + agent.n_cur, "samples_per_ray": S,
                "parallelism": f"ONE scene ({name}) split into {world} slabs along {info['axis']} (0.5 m overlap, one lattice): "
                               "overlap-rectangle plane gradients exchanged point-to-point with the neighbours"
                               + (", gradients of the planes without the slab axis all-reduced over all agents" if world > 2 else "")
                               + " + decoder-gradient all-reduce, every iteration (" + str(dist.get_backend()) + ")",
                "exchange_bytes_per_iter_all_ranks": sum(r["overlap_exchange_bytes_per_iter"] for r in rows) + world * 4 * agent.n_dec_params,
                "slab_bounds": info["slabs"], "per_rank": rows}
            del agent
        except Exception as e:    # noqa: BLE001
            side["as_worded"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    signal.alarm(0)
    return side


def spawn_ranks(n):
    """``python bench.py --gpus N`` without a launcher: start the N ranks ourselves -- one process per GPU, as the reference
    starts one process per agent (multi_agents.py:43-52) -- by re-running this command line under torch.distributed.run
    (rendezvous on 127.0.0.1, a free port).  The ranks print the ONE JSON line (rank 0); its ``n_gpus`` and
    ``config.ranks_seen`` are what the process group reported."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    # Launcher dry run (tests/test_dist_gloo.py): with MNE_EMULATED_LIBRARY = a host-emulator build of the kernels (test
    # infrastructure, tests/hostemu) and no GPU, the same code path runs over gloo so that the multi-process logic has been
    # executed before an 8-GPU node sees it.  Its line is marked as such and is not a measurement.
    dry = os.environ.get("MNE_EMULATED_LIBRARY") if not torch.cuda.is_available() else None
    if not torch.cuda.is_available() and not dry:
        raise SystemExit("bench.py needs an MI355X (the HIP library is the only backend)")
    if dry:
        from mneslam_amd import _lib
        _lib.load(dry)
        args.event_every, args.cpu_iters = 1 << 30, 0
        args.variants = args.variants and int(os.environ.get("WORLD_SIZE", "1")) > 1      # (the N > 1 side record is launcher logic too)
    from mneslam_amd import dist as mdist
    rank, world, device = mdist.init_agents()          # one process per GPU; RCCL when WORLD_SIZE > 1
    import torch.distributed as dist
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s): one rank per GPU")
    ranks_seen = dist.get_world_size() if dist.is_initialized() else 1
    make_cfg, workload = configs.WORKLOADS[args.config]
    cfg = make_cfg(args.hidden) if args.hidden else make_cfg()
    args.hidden = cfg["decoder"]["hidden_dim"]
    if args.rays:
        if not args.small:
            raise SystemExit("--rays changes the workload: functional runs (--small) only")
        cfg["mapping"]["sample"] = args.rays
        cfg["mapping"]["min_pixels_cur"] = min(cfg["mapping"]["min_pixels_cur"], max(args.rays // 4, 1))
    if args.small:
        cfg["mapping"]["bound"] = [[-1.0, 1.0], [-1.2, 1.1], [-0.8, 0.9]]
        cfg["planes_res"] = {"coarse": 0.1, "fine": 0.05, "bound_dividable": 0.1}
        cfg["c_planes_res"] = {"coarse": 0.2, "fine": 0.1}
    split = None
    if args.split:
        if world < 2 or args.config not in configs.SCENE_BOUNDS or args.path != "fused" or args.scatter != "binned" or args.graph:
            raise SystemExit("--split: --gpus N > 1, --config " + " | ".join(sorted(configs.SCENE_BOUNDS)) + ", the fused binned path")
        agent, cfg, split = make_split_agent(args.config, rank, world, device, args.keyframes, small=args.small, rays=args.rays)
        workload, args.share_decoder = split["workload"], True
    else:
        agent = Agent(cfg, device, seed=rank, n_keyframes=args.keyframes, small=args.small, path=args.path, scatter=args.scatter,
                      share_decoder=args.share_decoder, overlap=not args.no_overlap, graph=args.graph)

    def barrier():
        if world > 1:
            dist.barrier()
        if device.type == "cuda":
            torch.cuda.synchronize()

    if args.mode == "render_img":
        return bench_render_img(args, cfg, workload, agent, device, rank, world, barrier, mdist)

    for _ in range(args.warmup):
        agent.step()
    timers = {}
    every = args.event_every or (25 if args.steps >= 50 else max(args.steps, 1))
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):        # every timed iteration draws its own batch inside the timed region:
        # step i+1's batch is drawn while step i's planes update.  HIP events bracket the two dominant launches on
        # every `every`-th step only: each record is a barrier packet on the stream (~6 us of idle time).
        agent.step(timers if i % every == every // 2 else None, prefetch=i + 1 < args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    own_elapsed = elapsed
    elapsed = mdist.max_over_ranks(elapsed, device)
    avg_ms = {k: sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1) for k, v in timers.items()}
    psnr, depth_l1 = agent.quality()
    per_rank = None
    if world > 1:
        per_rank = [None] * world
        f = agent.fused
        dist.all_gather_object(per_rank, {
            "rank": rank, "it_per_s": args.steps / own_elapsed, "plane_params": agent.n_plane_params,
            "overlap_exchange_bytes_per_iter": (4 * 2 * sum(t.numel() for t in f.ov_send)) if (f is not None and f.tile_overlap is not None) else 0,
            "psnr_last_iter": psnr, "depth_l1_last_iter": depth_l1})
    agent_dec_params = agent.n_dec_params
    out = None
    if rank == 0:
        acc = account(cfg, agent, avg_ms)
        alg, kern, dom, dom_ms, achieved = acc["alg"], acc["kern"], acc["dom"], acc["dom_ms"], acc["achieved"]
        p_contrib, decoded, R, S = acc["p_contrib"], acc["decoded"], acc["R"], acc["S"]
        # HBM traffic and matrix-pipe busy cycles from committed PMC passes (rocprofv3 cannot run inside the timed loop)
        traffic, mfma_busy = None, None
        try:
            pmc = json.load(open(os.path.join(REPO, "profiles", PMC_JSON)))
            if (pmc["workload"] == workload and not args.small and args.path == "fused"
                    and pmc["scatter"] == args.scatter and args.hidden == 32):
                per_k = pmc["per_kernel_hbm_bytes_per_iteration"]
                tag = {"adam": "tile_adam_kernel"}.get(dom, dom)
                traffic = sum(v for k, v in per_k.items() if tag in k) or None
                clk = 2.4e9 * 1024                        # SIMD-cycles per second: 256 CUs x 4 SIMDs at 2.4 GHz
                # (the counters are per iteration: the first-pass launch carries practically all of a kernel's cycles)
                mfma_busy = {name: sum(cyc for k, cyc in pmc["per_kernel_mfma_busy_cycles_per_iteration"].items() if name in k)
                             / clk / (avg_ms[name] * 1e-3)
                             for name in ("decode_kernel", "ray_kernel") if avg_ms.get(name)}
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": f"mapping iters/sec ({cfg['mapping']['sample']} rays x {S} samples)", "value": world * args.steps / elapsed,
            "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (HOST-EMULATOR DRY RUN of the launcher logic: not a measurement)" if dry else "")
                                                         + (" (RANKS SHARE GPUs over gloo, MNE_SHARE_GPUS=1: functional run of the multi-agent path on the HIP library, not a scaling measurement)"
                                                            if (world > 1 and device.type == "cuda" and world > torch.cuda.device_count()) else ""),
            "config": {"workload": workload + ("_SMALL" if args.small else ""),
                       "rays_per_iter": R, "samples_per_ray": S, "plane_params": agent.n_plane_params,
                       "decoder_params": agent.n_dec_params, "mlp_hidden": args.hidden, "keyframes": args.keyframes,
                       "frame": f"{agent.W}x{agent.H}", "path": args.path, "scatter": ("hash-" + agent.fused.table_update) if agent.hash else args.scatter if args.path == "fused" else "atomics", "agents": world,
                       "ranks_seen": ranks_seen, "collective_backend": (dist.get_backend() if dist.is_initialized() else None),
                       "plane_dtype": cfg["grid"].get("plane_dtype", "fp32"), "launch": ("hipGraph replay (" + args.graph + ")") if args.graph else "eager",
                       "encoding": "hash grid (parity unpinned: tinycudann is not in the reference tree)" if agent.hash else "tri-planes (as wired)",
                       "parallelism": (f"ONE scene ({args.config}) split into {world} slabs along {split['axis']} (0.5 m overlap, one lattice), "
                                       f"agent-per-gpu x{world}: overlap-rectangle plane gradients exchanged point-to-point with the "
                                       "neighbours" + (", gradients of the planes without the slab axis all-reduced over all agents"
                                                       if world > 2 else "")
                                       + " + decoder-gradient all-reduce, every iteration (extension)") if split else
                                      f"agent-per-gpu x{world}, " + ("decoder-gradient all-reduce (extension)" if args.share_decoder
                                                                     else "no data-path collective")},
            "psnr_last_iter": psnr, "depth_l1_last_iter": depth_l1,
            "roofline": {"kernel": kern[dom], "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": ("profiles/" + PMC_JSON + " (committed rocprofv3 --pmc passes of this workload; a profiler "
                                            "cannot sit inside the timed loop)") if traffic else None,
                         "algorithmic_bytes_per_launch": alg.get(dom, 0.0), "avg_launch_ms": dom_ms,
                         "traffic_frac": (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and dom_ms > 0) else None,
                         "mfma_busy": mfma_busy,
                         "plane_params_swept": acc["swept"],
                         "contributing_samples_last_iter": p_contrib, "gathered_samples_last_iter_lower_bound": decoded,
                         "nominal_samples": float(R * S),
                         "other_kernels_avg_ms": {kern[k]: v for k, v in avg_ms.items() if k != dom},
                         "other_kernels_algorithmic_bytes": {kern[k]: alg[k] for k in alg if k != dom and k not in ("render", "iteration") and k in kern},
                         "iteration_algorithmic_bytes": alg.get("iteration", alg["adam"] + alg["render"]),
                         "iteration_hbm_frac": alg.get("iteration", alg["adam"] + alg["render"]) / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and args.cpu_iters > 0:
            batch = None
            if agent.fused is not None and not agent.hash:       # the batch the device drew in its last timed iteration
                f = agent.fused
                batch = (f.rays_o[:R], f.rays_d[:R], f.tgt_rgb[:R], f.tgt_d[:R], f.z_vals[:R])
            out["cpu_baseline"] = cpu_baseline(cfg, args.keyframes, args.cpu_iters, batch=batch)
        if world == 1 and args.variants and not args.small:
            del agent
            torch.cuda.empty_cache()
            out["variants"] = {}
            for name, c, h, gr in VARIANTS:
                if c == args.config and (h or args.hidden) == args.hidden and gr == args.graph:
                    continue
                try:                      # a side record must never cost the metric's line
                    out["variants"][name] = run_variant(c, h, device, args.keyframes, graph=gr, name=name)
                except Exception as e:    # noqa: BLE001
                    out["variants"][name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            # (render_img comes LAST: the full-frame renders allocate and free several GB of scratch, and workloads measured after them
            # in the same process ran 20-25 % slower than stand-alone -- INS Indoor 830 vs 1110 it/s, office0 with fp16 storage 1716 vs
            # 1964, first_frame_mapping 1580 vs 1950 -- in positions that changed with the order of the records)
            try:        # the reference's first_frame_mapping: mapping.first_iters (500) iterations on ONE frame, from a fresh map
                fagent = Agent(cfg, device, seed=1, n_keyframes=1)
                n_first, n_ray = cfg["mapping"]["first_iters"], cfg["mapping"]["sample"]
                pose = fagent.poses[-1:].contiguous()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(n_first):
                    fagent.fused.step(None, 0, 1, fagent.cur_rays, pose, 0, n_ray, prefetch=i + 1 < n_first)
                torch.cuda.synchronize()
                el = time.perf_counter() - t1
                fagent.fused.check()
                out["variants"]["first_frame_mapping"] = {
                    "workload": workload + "_first_frame", "iterations": n_first, "rays_per_iter": n_ray, "value": n_first / el, "unit": "it/s",
                    "total_ms": 1e3 * el, "psnr_last_iter": float(fagent.fused.losses[7].item()),
                    "note": "mp_slam/mapper.py:52-89: every iteration on the first frame's rays, untrained map at the start (the adaptive "
                            "schedule decodes every sample a priori until the SDF has sign changes)"}
                del fagent
                torch.cuda.empty_cache()
            except Exception as e:    # noqa: BLE001
                out["variants"]["first_frame_mapping"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            try:                          # N1 on the record: the two full-frame renders of a keyframe, on a map of 100 iterations
                ragent = Agent(cfg, device, seed=0, n_keyframes=args.keyframes)
                for _ in range(100):
                    ragent.step()
                torch.cuda.synchronize()
                try:                      # R13 / N2: loop closure's pose alignment on the same (trained) map, before the frames' scratch
                    out["variants"]["pose_alignment"] = dict({"workload": workload + "_pose_alignment"},
                                                             **pose_alignment_measure(ragent, cfg, device))
                except Exception as e:    # noqa: BLE001
                    out["variants"]["pose_alignment"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                own, rec = render_img_measure(ragent, cfg, device, n_pairs=5, n_warm=1)
                out["variants"]["render_img"] = dict({"workload": workload + "_render_img", "pretrain_iterations": 100,
                                                      "value": 5 / own, "unit": "frame pairs/s", "ms_per_pair": 1e3 * own / 5}, **rec)
                del ragent
                torch.cuda.empty_cache()
            except Exception as e:    # noqa: BLE001
                out["variants"]["render_img"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if per_rank:
            out["per_rank"] = per_rank
            if split:
                out["config"]["slab_bounds"] = split["slabs"]
                out["config"]["exchange_bytes_per_iter_all_ranks"] = (sum(r["overlap_exchange_bytes_per_iter"] for r in per_rank)
                                                                      + world * 4 * agent_dec_params)
    if world > 1 and args.variants and not args.share_decoder and not split and args.mode != "render_img":
        side = multi_agent_side_records(args, cfg, rank, world, device, barrier, out if rank == 0 else None)
        if rank == 0:
            out["variants"] = side
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
