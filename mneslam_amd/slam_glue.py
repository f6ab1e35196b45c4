"""The three MNESLAM methods that sit on the mapping hot path (mneslam_mp.py), as free functions
taking the SLAM object / config, with the reference's names and argument meaning:

  select_samples      mneslam_mp.py:342-348
  get_loss_from_ret   mneslam_mp.py:350-372
  create_optimizer    mneslam_mp.py:431-469

The rest of mneslam_mp.py (dataset, DROID tracker, threads, checkpoints, image dumps) is
orchestration that stays with the host application.
"""
import random

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
