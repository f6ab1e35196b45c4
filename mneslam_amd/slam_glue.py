"""The three MNESLAM methods that sit on the mapping hot path (mneslam_mp.py), as free functions
taking the SLAM object / config, with the reference's names and argument meaning:

  select_samples      mneslam_mp.py:342-348
  get_loss_from_ret   mneslam_mp.py:350-372
  create_optimizer    mneslam_mp.py:431-469

plus the FILE formats through which agents hand maps and keyframe poses to each other (row N4 of SURVEY.md 8f;
the xGMI form of the same hand-off is mneslam_amd/dist.py):

  save_latest_checkpoint   mneslam_mp.py:294-315      agent_<rank>/latest_checkpoint.pt
  load_foreign_model       mp_slam/mapper.py:708-726  (reader of the above)
  save_keyframe_poses / load_keyframe_poses   mp_slam/mapper.py:565-592, :344-356   key_est_poses.npy, key_timestamps.npy

The rest of mneslam_mp.py (dataset, DROID tracker, threads, checkpoints, image dumps) is
orchestration that stays with the host application.
"""
import os
import random

import numpy as np
import torch
import torch.nn as nn

from .optim import FusedAdam


def select_samples(H, W, samples):
    """Randomly select pixel indices with python ``random`` (reference RNG)."""
    return torch.tensor(random.sample(range(H * W), int(samples)))


def get_loss_from_ret(config, ret, rgb=True, sdf=True, is_co_sdf=True, depth=True, smooth=False):
    """Weighted sum of the loss dict returned by JointEncoding.forward."""
    if smooth:
        raise NotImplementedError("the smoothness term is never enabled by any caller in the reference "
                                  "(SURVEY.md A22) and is not provided")
    loss = 0
    if rgb:
        loss = loss + config["training"]["rgb_weight"] * ret["rgb_loss"]
    if depth:
        loss = loss + config["training"]["depth_weight"] * ret["depth_loss"]
    if sdf:
        if is_co_sdf:
            loss = loss + (config["training"]["sdf_weight"] * ret["co_sdf_loss"]
                           + config["training"]["fs_weight"] * ret["co_fs_loss"])
        else:
            loss = loss + (config["mapping"]["w_sdf_fs"] * ret["e_fs_loss"]
                           + config["mapping"]["w_sdf_center"] * ret["e_center_loss"]
                           + config["mapping"]["w_sdf_tail"] * ret["e_tail_loss"])
    return loss


def loss_weight_vector(config, is_co_sdf, rgb=True, sdf=True, depth=True):
    """d(total)/d(loss_k) in MNE_L_* order (rgb, depth, co_sdf, co_fs, e_fs, e_center, e_tail) --
    what autograd would hand back for get_loss_from_ret; used by the fused training step."""
    w = [0.0] * 7
    if rgb:
        w[0] = float(config["training"]["rgb_weight"])
    if depth:
        w[1] = float(config["training"]["depth_weight"])
    if sdf:
        if is_co_sdf:
            w[2], w[3] = float(config["training"]["sdf_weight"]), float(config["training"]["fs_weight"])
        else:
            w[4], w[5], w[6] = (float(config["mapping"]["w_sdf_fs"]), float(config["mapping"]["w_sdf_center"]),
                                float(config["mapping"]["w_sdf_tail"]))
    return w


def create_optimizer(model, config, optimizer_cls=FusedAdam):
    """Wrap every plane in ``nn.Parameter`` IN PLACE in the model's lists and build Adam with the
    reference's groups: decoder {lr_decoder, weight_decay 1e-6}, planes {lr_embed, eps 1e-15},
    colour planes {lr_embed_color, eps 1e-15}, betas (0.9, 0.99)."""
    if getattr(model, "embed_fn", None) is not None:       # hash-grid wiring (model/scene_rep_hash.py): Co-SLAM's two groups
        return optimizer_cls([{"params": list(model.decoder.parameters()), "weight_decay": 1e-6, "lr": config["mapping"]["lr_decoder"]},
                              {"params": list(model.embed_fn.parameters()), "eps": 1e-15, "lr": config["mapping"]["lr_embed"]}],
                             betas=(0.9, 0.99))
    one_grid = config["grid"]["oneGrid"]
    sets = model.all_planes
    planes_para, c_planes_para = [], []
    if not one_grid:
        for c_planes in sets[3:6]:
            for i, p in enumerate(c_planes):
                p = nn.Parameter(p)
                c_planes_para.append(p)
                c_planes[i] = p
    for planes in sets[0:3]:
        for i, p in enumerate(planes):
            p = nn.Parameter(p)
            planes_para.append(p)
            planes[i] = p
    groups = [{"params": list(model.decoder.parameters()), "weight_decay": 1e-6, "lr": config["mapping"]["lr_decoder"]},
              {"params": planes_para, "eps": 1e-15, "lr": config["mapping"]["lr_embed"]}]
    if not one_grid:
        groups.append({"params": c_planes_para, "eps": 1e-15, "lr": config["mapping"]["lr_embed_color"]})
    return optimizer_cls(groups, betas=(0.9, 0.99))


# --------------------------------------------------------------------------------------------------------------
# inter-agent files (N4)
# --------------------------------------------------------------------------------------------------------------
def agent_dir(config, rank):
    """<data.output>/<data.exp_name>/agent_<rank>: where an agent publishes its state for its peers."""
    return os.path.join(config["data"]["output"], config["data"]["exp_name"], f"agent_{rank}")


def _publish(path, writer):
    """Write next to ``path`` and rename: a peer polling the file never sees a partial one."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tmp = path + ".tmp"
    writer(tmp)
    os.replace(tmp, path)


def save_latest_checkpoint(model, config, rank):
    """``latest_checkpoint.pt`` in the reference's layout: ``{'model': state_dict, 'all_planes': tuple of lists of
    [1,C,H,W] tensors, 'bound', 'bounding_box'}`` (host copies of the two boxes).  Planes are pickled as they are --
    torch.save keeps their channels_last strides, and any consumer (ATen grid_sample included) reads them through the
    logical NCHW shape."""
    path = os.path.join(agent_dir(config, rank), "latest_checkpoint.pt")
    payload = {"model": model.state_dict(), "all_planes": model.all_planes,
               "bound": torch.as_tensor(model.bound).cpu(), "bounding_box": torch.as_tensor(model.bounding_box).cpu()}
    _publish(path, lambda tmp: torch.save(payload, tmp))
    return path


def load_foreign_model(model_shared, config, other_rank, device, exchange=None):
    """A peer's map into ``model_shared`` (decoder weights, planes, both boxes), which is put in eval mode.
    With ``exchange`` (mneslam_amd.dist.ModelExchange, running on both agents) the map comes straight from the peer's
    device memory over the process group -- RCCL point-to-point over xGMI between GPUs -- and the returned dict has the
    checkpoint's keys; otherwise (and whenever no process group is up) the peer's ``latest_checkpoint.pt`` is read, the
    reference's own path (mp_slam/mapper.py:708-726), which stays format-compatible with a reference agent's files.
    Planes written by a reference agent are NCHW-contiguous: they are re-laid channels_last HERE, once, so that rendering
    never converts per call."""
    if exchange is not None:
        exchange.fetch(model_shared, other_rank)
        return {"model": model_shared.state_dict(), "all_planes": model_shared.all_planes,
                "bound": model_shared.bound, "bounding_box": model_shared.bounding_box, "source": f"exchange:{other_rank}"}
    path = os.path.join(agent_dir(config, other_rank), "latest_checkpoint.pt")
    ckpt = torch.load(path, map_location=device, weights_only=False)
    model_shared.load_state_dict(ckpt["model"])
    if "all_planes" in ckpt:
        model_shared.all_planes = tuple([p.detach().to(device).contiguous(memory_format=torch.channels_last) for p in lst]
                                        for lst in ckpt["all_planes"])
    if "bound" in ckpt:
        model_shared.bound = ckpt["bound"].to(device)
    if "bounding_box" in ckpt:
        model_shared.bounding_box = ckpt["bounding_box"].to(device)
    model_shared.eval()
    return ckpt


def save_keyframe_poses(config, rank, poses_c2w, timestamps):
    """``key_est_poses.npy`` [K,4,4] and ``key_timestamps.npy`` [K] of this agent, each published atomically."""
    d = agent_dir(config, rank)
    # np.save appends ".npy" to names without it: keep the temporary name's suffix
    for name, arr in (("key_est_poses", poses_c2w), ("key_timestamps", timestamps)):
        final = os.path.join(d, name + ".npy")
        os.makedirs(d, exist_ok=True)
        tmp = os.path.join(d, name + "_tmp.npy")
        np.save(tmp, torch.as_tensor(arr).detach().cpu().numpy())
        os.replace(tmp, final)
    return d


def load_keyframe_poses(config, other_rank):
    """(poses [K,4,4], timestamps [K]) published by agent ``other_rank``."""
    d = agent_dir(config, other_rank)
    return np.load(os.path.join(d, "key_est_poses.npy")), np.load(os.path.join(d, "key_timestamps.npy"))
