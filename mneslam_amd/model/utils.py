"""Small host utilities with the reference's names (model/utils.py).  Tensor plumbing only; the
per-sample math of the hot path lives in the HIP kernels."""
import numpy as np
import torch
import torch.nn.functional as F


def get_rays(H, W, fx, fy, cx, cy, c2w, device):
    """Whole-image rays (reference: model/utils.py:7-25).  Returns rays_o, rays_d [H,W,3]."""
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w)
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="ij")
    i, j = i.t(), j.t()
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1).to(device)
    rays_d = torch.sum(dirs.reshape(H, W, 1, 3) * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def normalize_3d_coordinate(p, bound):
    """[-1,1] normalisation with the extended bound (reference: model/utils.py:27-41)."""
    p = p.reshape(-1, 3)
    out = torch.empty_like(p)
    for k in range(3):
        out[:, k] = ((p[:, k] - bound[k, 0]) / (bound[k, 1] - bound[k, 0])) * 2 - 1.0
    return out


def mse2psnr(x):
    """reference: model/utils.py:43-47"""
    return -10.0 * torch.log(x) / torch.log(torch.Tensor([10.0])).to(x)


def batchify(fn, chunk=1024 * 64):
    """reference: model/utils.py:106-115 (``chunk=None`` returns ``fn`` itself)."""
    if chunk is None:
        return fn

    def ret(inputs, inputs_dir=None):
        if inputs_dir is not None:
            return torch.cat([fn(inputs[i:i + chunk], inputs_dir[i:i + chunk])
                              for i in range(0, inputs.shape[0], chunk)], 0)
        return torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)
    return ret


def compute_loss(prediction, target, loss_type="l2"):
    """reference: model/utils.py:147-162"""
    if loss_type == "l2":
        return F.mse_loss(prediction, target)
    if loss_type == "l1":
        return F.l1_loss(prediction, target)
    raise Exception("Unsupported loss type")
