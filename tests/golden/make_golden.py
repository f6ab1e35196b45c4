#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (CPU) in the build container.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference (/root/reference) is imported through tests/golden/ref_harness.py (stubs for
absent third-party modules; tinycudann.OneBlob backed by oracle/oneblob.py).  Only inputs and
outputs are stored -- no reference source travels.  The fixtures pin, per SURVEY.md section 8(c):
z_vals, normalised points, plane features, raw decoder output, rendered maps, all seven losses,
total loss for both ``is_co_sdf`` settings, gradients of decoder / planes / rays, and parameters
after three ``Mapper.mapping_optimize`` iterations with the reference's own Adam groups.
"""
import copy
import os
import random
import sys
import threading
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402

ref_harness.install()
os.chdir(ref_harness.REF)                         # config inherit_from paths are relative
import mneslam_mp as ref_slam                     # noqa: E402
from model.scene_rep import JointEncoding         # noqa: E402
from model.keyframe import KeyFrameDatabase       # noqa: E402
from model.utils import normalize_3d_coordinate   # noqa: E402
from mp_slam.mapper import Mapper                 # noqa: E402
os.chdir(REPO)

from mneslam_amd import synthetic                 # noqa: E402  (inputs only)

SMALL_BOUND = [[-1.0, 1.0], [-1.2, 1.1], [-0.8, 0.9]]
SMALL_ROOM = [[-0.8, 0.8], [-1.0, 0.9], [-0.6, 0.7]]


def small_config(one_grid=True, is_co_sdf=False, n_samples_d=32, n_range_d=11, depth_trunc=100.0):
    cfg = copy.deepcopy(ref_harness.load_config("configs/Replica/office0.yaml"))
    cfg["mapping"]["bound"] = SMALL_BOUND
    cfg["mapping"]["marching_cubes_bound"] = SMALL_ROOM
    cfg["planes_res"] = {"coarse": 0.2, "fine": 0.1, "bound_dividable": 0.2}
    cfg["c_planes_res"] = {"coarse": 0.4, "fine": 0.2}
    cfg["grid"]["oneGrid"] = one_grid
    cfg["is_co_sdf"] = is_co_sdf
    cfg["cam"]["far"] = 4.0
    cfg["cam"]["depth_trunc"] = depth_trunc
    cfg["training"]["n_samples_d"] = n_samples_d
    cfg["training"]["n_range_d"] = n_range_d
    cfg["training"]["n_samples"] = 48
    cfg["enable_loop_detect"] = False
    return cfg


def t2n(t):
    return t.detach().cpu().numpy().copy()


class FakeSLAM:
    """Minimal stand-in for MNESLAM exposing what create_optimizer / get_loss_from_ret /
    Mapper.__init__ read (mp_slam/mapper.py:12-50).  The three reference methods are
    bound unmodified from mneslam_mp.MNESLAM."""
    get_loss_from_ret = ref_slam.MNESLAM.get_loss_from_ret
    create_optimizer = ref_slam.MNESLAM.create_optimizer
    select_samples = ref_slam.MNESLAM.select_samples

    def __init__(self, cfg, model):
        self.config = cfg
        self.model = model
        self.model_shared = model
        self.device = torch.device("cpu")


def make_rays(cfg, n, seed, with_invalid=True):
    """Rays from a synthetic frame inside the small room (+ a few invalid / far depths)."""
    H, W = 48, 64
    frames = synthetic.make_frames(2, H, W, 60.0, 60.0, 31.0, 23.0, SMALL_ROOM, seed=seed, invalid_frac=0.0)
    g = torch.Generator().manual_seed(seed)
    fr = frames[1]
    idx = torch.randperm(H * W, generator=g)[:n]
    d_cam = fr["direction"].reshape(-1, 3)[idx]
    rgb = fr["rgb"].reshape(-1, 3)[idx].clone()
    dep = fr["depth"].reshape(-1)[idx].clone()
    if with_invalid:
        dep[::7] = 0.0                              # invalid depth rows (d <= 0)
        dep[3] = -0.25
        dep[5::29] = cfg["cam"]["depth_trunc"] + 1.0  # beyond depth_trunc (still > 0)
    c2w = fr["c2w"]
    rays_d = torch.sum(d_cam[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[None, :3, -1].repeat(n, 1)
    return rays_o.contiguous(), rays_d.contiguous(), rgb.contiguous(), dep[:, None].contiguous()


def plane_arrays(model, prefix, out):
    for si, lst in enumerate(model.all_planes):
        for li, p in enumerate(lst):
            out[f"{prefix}plane_{si}_{li}"] = t2n(p)


def decoder_arrays(model, prefix, out):
    for name, p in model.decoder.named_parameters():
        out[f"{prefix}dec.{name}"] = t2n(p)


def build_model(cfg, seed):
    torch.manual_seed(seed)
    bb = torch.from_numpy(np.array(cfg["mapping"]["bound"]))       # float64, mneslam_mp.py:223
    model = JointEncoding(cfg, bb)
    # larger plane values than N(0, 0.01) so every path carries signal in fp32 comparisons
    for lst in model.all_planes:
        for i, p in enumerate(lst):
            lst[i] = p * 20.0
    slam = FakeSLAM(cfg, model)
    slam.create_optimizer()            # wraps planes in nn.Parameter in place, builds Adam
    return model, slam, bb


def case_forward(name, one_grid, n_rays, seed, depth_trunc=100.0, all_invalid=False):
    cfg = small_config(one_grid=one_grid, depth_trunc=depth_trunc)
    model, slam, bb = build_model(cfg, seed)
    model.train()
    rays_o, rays_d, tgt_rgb, tgt_d = make_rays(cfg, n_rays, seed)
    if all_invalid:
        tgt_d = -torch.ones_like(tgt_d)
    rays_o.requires_grad_(True)
    rays_d.requires_grad_(True)
    S = cfg["training"]["n_samples_d"] + cfg["training"]["n_range_d"]
    torch.manual_seed(seed + 100)
    U = torch.rand(n_rays, S)
    torch.manual_seed(seed + 100)      # the reference draws the jitter itself (scene_rep.py:381)
    ret = model.forward(rays_o, rays_d, tgt_rgb, tgt_d)
    out = {"bounding_box": t2n(bb), "bound_ext": t2n(model.bound), "rays_o": t2n(rays_o),
           "rays_d": t2n(rays_d), "target_rgb": t2n(tgt_rgb), "target_d": t2n(tgt_d), "U": t2n(U),
           "one_grid": np.array(one_grid)}
    plane_arrays(model, "", out)
    decoder_arrays(model, "", out)
    for k, v in ret.items():
        out[f"ret.{k}"] = t2n(v)
    for co in (False, True):
        model.zero_grad()
        for lst in model.all_planes:
            for p in lst:
                p.grad = None
        rays_o.grad = rays_d.grad = None
        torch.manual_seed(seed + 100)
        ret = model.forward(rays_o, rays_d, tgt_rgb, tgt_d)
        loss = slam.get_loss_from_ret(ret, is_co_sdf=co)
        out[f"loss.co{int(co)}"] = t2n(loss)
        if all_invalid:
            continue                   # loss is NaN by construction; gradients undefined
        loss.backward()
        tag = f"grad.co{int(co)}."
        for si, lst in enumerate(model.all_planes):
            for li, p in enumerate(lst):
                out[f"{tag}plane_{si}_{li}"] = t2n(p.grad)
        for nme, p in model.decoder.named_parameters():
            out[f"{tag}dec.{nme}"] = t2n(p.grad)
        out[f"{tag}rays_o"] = t2n(rays_o.grad)
        out[f"{tag}rays_d"] = t2n(rays_d.grad)
    # intermediates through the reference's own functions (eval of the same points)
    with torch.no_grad():
        torch.manual_seed(seed + 100)
        rr = model.render_rays(rays_o, rays_d, target_d=tgt_d)
        out["rr.z_vals"], out["rr.raw"] = t2n(rr["z_vals"]), t2n(rr["raw"])
        for k in ("rgb", "depth", "disp_map", "acc_map", "depth_var"):
            out[f"rr.{k}"] = t2n(rr[k])
        pts = rays_o[:, None, :] + rays_d[:, None, :] * rr["z_vals"][:, :, None]
        flat = pts.reshape(-1, 3)
        p_nor = normalize_3d_coordinate(flat.clone(), model.bound).float()
        out["mid.p_nor"] = t2n(p_nor)
        out["mid.feat"] = t2n(model.sample_plane_feature(p_nor, *model.all_planes[:3]))
        if not one_grid:
            out["mid.cfeat"] = t2n(model.sample_plane_feature(p_nor, *model.all_planes[3:]))
        u = (flat - model.bounding_box[:, 0]) / (model.bounding_box[:, 1] - model.bounding_box[:, 0])
        out["mid.pos"] = t2n(model.embedpos_fn(u))
        w = model.sdf2weights(rr["raw"][..., 3], rr["z_vals"], args=cfg)
        out["mid.weights"] = t2n(w)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(out.items())[:4]},
          "loss", out["loss.co0"], out["loss.co1"])


def case_render_nodepth(name, seed):
    """render_rays(target_d=None) as loop closure uses it (mp_slam/mapper.py:380-408):
    pose-only gradients through rgb/depth MSE."""
    cfg = small_config(one_grid=True)
    model, slam, bb = build_model(cfg, seed)
    model.eval()
    n = 40
    rays_o, rays_d, tgt_rgb, tgt_d = make_rays(cfg, n, seed, with_invalid=False)
    rays_o.requires_grad_(True)
    rays_d.requires_grad_(True)
    S = cfg["training"]["n_samples"]
    torch.manual_seed(seed + 5)
    U = torch.rand(n, S)
    torch.manual_seed(seed + 5)
    rr = model.render_rays(rays_o, rays_d, target_d=None)
    loss = (cfg["training"]["rgb_weight"] * torch.nn.functional.mse_loss(rr["rgb"], tgt_rgb)
            + cfg["training"]["depth_weight"] * torch.nn.functional.mse_loss(rr["depth"], tgt_d.squeeze()))
    loss.backward()
    out = {"bounding_box": t2n(bb), "rays_o": t2n(rays_o), "rays_d": t2n(rays_d), "target_rgb": t2n(tgt_rgb),
           "target_d": t2n(tgt_d), "U": t2n(U), "loss": t2n(loss), "grad.rays_o": t2n(rays_o.grad),
           "grad.rays_d": t2n(rays_d.grad)}
    for k in ("rgb", "depth", "disp_map", "acc_map", "depth_var", "z_vals", "raw"):
        out[f"rr.{k}"] = t2n(rr[k])
    plane_arrays(model, "", out)
    decoder_arrays(model, "", out)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(name, "loss", out["loss"])


def case_mapping(name, one_grid, is_co_sdf, seed):
    """Three iterations of the reference Mapper.mapping_optimize on a 3-keyframe database,
    with the reference's KeyFrameDatabase, create_optimizer and get_loss_from_ret."""
    cfg = small_config(one_grid=one_grid, is_co_sdf=is_co_sdf)
    cfg["mapping"]["sample"] = 64
    cfg["mapping"]["min_pixels_cur"] = 10
    cfg["mapping"]["iters"] = 3
    cfg["mapping"]["n_pixels"] = 0.25
    H, W = 24, 32
    model, slam, bb = build_model(cfg, seed)
    frames = synthetic.make_frames(4, H, W, 30.0, 30.0, 15.0, 11.0, SMALL_ROOM, seed=seed, invalid_frac=0.03)
    n_save = int(H * W * cfg["mapping"]["n_pixels"])
    out = {"bounding_box": t2n(bb), "H": np.array(H), "W": np.array(W), "n_save": np.array(n_save),
           "one_grid": np.array(one_grid), "is_co_sdf": np.array(is_co_sdf)}
    plane_arrays(model, "init.", out)
    decoder_arrays(model, "init.", out)
    for k, fr in enumerate(frames):
        out[f"frame{k}.c2w"], out[f"frame{k}.rgb"], out[f"frame{k}.depth"] = t2n(fr["c2w"]), t2n(fr["rgb"]), t2n(fr["depth"])
    out["direction"] = t2n(frames[0]["direction"])

    random.seed(seed)
    torch.manual_seed(seed)
    kfdb = KeyFrameDatabase(cfg, H, W, 8, n_save, torch.device("cpu"))
    for k in range(3):
        kfdb.add_keyframe(frames[k], k + 1)
    out["kf.rays"] = t2n(kfdb.rays[:3])

    slam.dataset = types.SimpleNamespace(H=H, W=W, fx=30.0, fy=30.0, cx=15.0, cy=11.0, rays_d=frames[0]["direction"])
    slam.video = types.SimpleNamespace(keyframe=kfdb)
    for attr in ("tracking_idx", "mapping_idx", "mapping_first_frame", "keyframe_dict", "mesher",
                 "all_agent_bounds"):
        setattr(slam, attr, None)
    slam.keyframe_dict_lock = threading.Lock()
    slam.descriptor_db_lock = threading.Lock()
    slam.rank, slam.world_size = 0, 1
    mapper = Mapper(cfg, slam)
    poses = torch.stack([f["c2w"] for f in frames])      # [4,4,4]; poses[-1] = current frame
    model.train()
    random.seed(seed + 1)
    torch.manual_seed(seed + 1)
    mapper.mapping_optimize(frames[3], poses)
    plane_arrays(model, "final.", out)
    decoder_arrays(model, "final.", out)
    st = slam.map_optimizer.state
    for gi, grp in enumerate(slam.map_optimizer.param_groups):
        for pi, p in enumerate(grp["params"]):
            out[f"adam.g{gi}.p{pi}.m"] = t2n(st[p]["exp_avg"])
            out[f"adam.g{gi}.p{pi}.v"] = t2n(st[p]["exp_avg_sq"])
            out[f"adam.g{gi}.p{pi}.step"] = np.array(float(st[p]["step"]))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(name, "done; dec w0 delta",
          float(np.abs(out["final.dec.sdf_net.model.0.weight"] - out["init.dec.sdf_net.model.0.weight"]).max()))


if __name__ == "__main__":
    torch.set_num_threads(4)
    case_forward("fwd_onegrid", one_grid=True, n_rays=48, seed=11)
    case_forward("fwd_colorplanes", one_grid=False, n_rays=32, seed=12, depth_trunc=3.0)
    case_forward("fwd_all_invalid", one_grid=True, n_rays=16, seed=13, all_invalid=True)
    case_render_nodepth("render_nodepth", seed=14)
    case_mapping("mapping3_onegrid_esdf", one_grid=True, is_co_sdf=False, seed=21)
    case_mapping("mapping3_colorplanes_cosdf", one_grid=False, is_co_sdf=True, seed=22)
