"""Synthetic RGB-D frames for the mapping hot path (no dataset ships with the repo).

The layout of a frame dict is the reference's input contract (datasets/dataset.py:121-128,
datasets/utils.py:24-57): ``{frame_id, c2w[4,4], rgb[H,W,3], depth[H,W], direction[H,W,3]}``
with OpenGL camera rays ``((i-cx)/fx, -(j-cy)/fy, -1)`` (NOT normalised, so ``depth`` is z-depth,
SURVEY.md A1).  Depth is the analytic exit distance of each ray from an axis-aligned room, colour a
smooth procedural texture of the hit point, poses a seeded smooth trajectory inside the room
(SURVEY.md section 8d).  Pure torch on CPU; deterministic for a given seed.
"""
import math
from typing import Dict, List, Sequence

import torch


def camera_rays(H, W, fx, fy, cx, cy) -> torch.Tensor:
    """OpenGL camera-frame ray directions [H,W,3] (reference: datasets/utils.py:24-57)."""
    i, j = torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32),
                          indexing="xy")
    return torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)


def look_at_c2w(eye: torch.Tensor, target: torch.Tensor, up=(0.0, 0.0, 1.0)) -> torch.Tensor:
    """OpenGL camera-to-world: camera looks down its -z axis towards ``target``."""
    f = target - eye
    f = f / f.norm()
    upv = torch.tensor(up, dtype=torch.float32)
    r = torch.linalg.cross(f, upv)
    r = r / r.norm()
    u = torch.linalg.cross(r, f)
    c2w = torch.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = r, u, -f, eye
    return c2w


def trajectory(n_frames: int, room: Sequence[Sequence[float]], seed: int = 0) -> torch.Tensor:
    """Smooth closed trajectory inside ``room`` ([3,2] lo/hi), camera looking at a moving target."""
    g = torch.Generator().manual_seed(seed)
    room_t = torch.tensor(room, dtype=torch.float32)
    ctr, half = room_t.mean(1), 0.5 * (room_t[:, 1] - room_t[:, 0])
    ph = torch.rand(4, generator=g) * 2 * math.pi
    poses = []
    for k in range(n_frames):
        t = 2 * math.pi * k / max(n_frames, 1)
        eye = ctr + half * torch.tensor([0.45 * math.cos(t + ph[0]), 0.45 * math.sin(t + ph[1]),
                                         0.25 * math.sin(2 * t + ph[2])])
        tgt = ctr + half * torch.tensor([0.9 * math.cos(t + ph[0] + 2.2), 0.9 * math.sin(t + ph[1] + 2.2),
                                         0.5 * math.cos(t + ph[3])])
        poses.append(look_at_c2w(eye, tgt))
    return torch.stack(poses)


def room_depth(c2w: torch.Tensor, dirs: torch.Tensor, room: torch.Tensor):
    """z-depth at which each camera ray leaves the room, and the hit point."""
    o = c2w[:3, 3]
    d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)                 # world dirs [H,W,3]
    safe = torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
    t_lo = (room[:, 0] - o) / safe
    t_hi = (room[:, 1] - o) / safe
    t_exit = torch.maximum(t_lo, t_hi).min(-1).values                      # camera is inside the room
    hit = o + d * t_exit[..., None]
    return t_exit, hit


def texture(hit: torch.Tensor) -> torch.Tensor:
    """Smooth colour in [0,1] as a function of the 3-D hit point."""
    x, y, z = hit[..., 0], hit[..., 1], hit[..., 2]
    r = 0.5 + 0.5 * torch.sin(1.7 * x + 0.6 * y)
    g = 0.5 + 0.5 * torch.sin(1.3 * y - 0.8 * z + 1.0)
    b = 0.5 + 0.5 * torch.sin(2.1 * z + 0.5 * x + 2.0)
    return torch.stack([r, g, b], -1)


def make_frames(n_frames: int, H: int, W: int, fx: float, fy: float, cx: float, cy: float,
                room: Sequence[Sequence[float]], seed: int = 0, invalid_frac: float = 0.02) -> List[Dict]:
    """List of frame dicts; ``invalid_frac`` of the depth pixels are zeroed (invalid depth)."""
    g = torch.Generator().manual_seed(seed + 1)
    dirs = camera_rays(H, W, fx, fy, cx, cy)
    room_t = torch.tensor(room, dtype=torch.float32)
    poses = trajectory(n_frames, room, seed)
    frames = []
    for k in range(n_frames):
        depth, hit = room_depth(poses[k], dirs, room_t)
        rgb = texture(hit)
        if invalid_frac > 0:
            drop = torch.rand(H, W, generator=g) < invalid_frac
            depth = torch.where(drop, torch.zeros_like(depth), depth)
        frames.append({"frame_id": k, "c2w": poses[k], "rgb": rgb.float().contiguous(),
                       "depth": depth.float().contiguous(), "direction": dirs})
    return frames


# Replica office0 camera after the reference's floor-division of intrinsics
# (datasets/dataset.py:41-44: cx 599.5 -> 599.0, cy 339.5 -> 339.0) and its room.
REPLICA_CAM = dict(H=680, W=1200, fx=600.0, fy=600.0, cx=599.0, cy=339.0)
OFFICE0_ROOM = [[-2.2, 2.6], [-3.4, 2.1], [-1.4, 2.0]]       # configs/Replica/office0.yaml:4
OFFICE0_BOUND = [[-3.0, 3.0], [-4.0, 2.5], [-2.0, 2.5]]      # configs/Replica/office0.yaml:3


def camera_from_config(cfg):
    """Mapping-side camera of a reference config, restating datasets/dataset.py:38-44 (H, W, fx, fy, cx, cy
    floor-divided by ``data.downsample`` -- also for the float intrinsics: cx 599.5 -> 599.0) and :174-178
    (``cam.crop_edge`` removes a border and shifts the principal point)."""
    cam, ds = cfg["cam"], cfg["data"]["downsample"]
    H, W = cam["H"] // ds, cam["W"] // ds
    fx, fy, cx, cy = cam["fx"] // ds, cam["fy"] // ds, cam["cx"] // ds, cam["cy"] // ds
    edge = cam.get("crop_edge", 0)
    if edge > 0:
        H, W, cx, cy = H - 2 * edge, W - 2 * edge, cx - edge, cy - edge
    return dict(H=int(H), W=int(W), fx=float(fx), fy=float(fy), cx=float(cx), cy=float(cy))


def room_from_config(cfg, shrink=0.0):
    """The synthetic room of a config = its ``mapping.marching_cubes_bound`` (the region the scene fills)."""
    return [[lo + shrink, hi - shrink] for lo, hi in cfg["mapping"]["marching_cubes_bound"]]
