"""Host-side helpers of the scene model: ray generation for whole images and the two normalisations of a point.
Names follow the reference's model/utils.py because callers of the scene model import them from there; the per-sample
math of the hot path does not live here (it is in the HIP kernels)."""
import torch


def pixel_directions(H, W, fx, fy, cx, cy, device="cpu"):
    """Camera-frame direction of every pixel, [H, W, 3], OpenGL convention: x right, y up, looking down -z; NOT
    normalised, so that a sample at parameter z lies at image depth z."""
    u = torch.arange(W, dtype=torch.float32, device=device).expand(H, W)
    v = torch.arange(H, dtype=torch.float32, device=device).unsqueeze(1).expand(H, W)
    return torch.stack([(u - cx) / fx, -(v - cy) / fy, torch.full((H, W), -1.0, device=device)], dim=-1)


def get_rays(H, W, fx, fy, cx, cy, c2w, device):
    """Origins and world-frame directions of all H x W pixel rays of a camera with pose ``c2w`` (tensor or ndarray):
    two [H, W, 3] tensors (reference: model/utils.py:7-25)."""
    c2w = torch.as_tensor(c2w).to(device=device, dtype=torch.float32)
    dirs = pixel_directions(H, W, fx, fy, cx, cy, device)
    rays_d = (dirs.unsqueeze(-2) * c2w[:3, :3]).sum(-1)          # R @ d, accumulated in the reference's order
    return c2w[:3, 3].expand_as(rays_d), rays_d


def normalize_3d_coordinate(p, bound):
    """Points -> [-1, 1]^3 of the (extended) bound, the plane-lookup coordinates (reference: model/utils.py:27-41).
    Returns a new [N, 3] tensor."""
    p = p.reshape(-1, 3)
    lo, hi = bound[:, 0].to(p), bound[:, 1].to(p)
    return ((p - lo) / (hi - lo)) * 2 - 1.0


def batchify(fn, chunk=1024 * 64):
    """``fn`` applied in chunks of ``chunk`` rows and concatenated; ``chunk=None`` hands ``fn`` back unchanged (which is
    how the decoder's sub-networks end up registered twice in the scene model's state_dict)."""
    if chunk is None:
        return fn

    def chunked(*tensors):
        n = tensors[0].shape[0]
        parts = [fn(*[t[i:i + chunk] for t in tensors if t is not None]) for i in range(0, n, chunk)]
        return torch.cat(parts, 0)
    return chunked
