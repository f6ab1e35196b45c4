"""Pin the CPU oracle (oracle/) against golden vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import random

import numpy as np
import pytest
import torch

from mneslam_amd import configs
from oracle import mapping as omap
from oracle.scene_rep import OracleScene, normalize_points, sample_plane_feature, bilinear_corners

from helpers import (DEC_KEYS, assert_close, fixture_inputs, load_golden, n_plane_sets,
                     oracle_scene_from_golden)

FWD_CASES = [("fwd_onegrid", dict(one_grid=True)),
             ("fwd_colorplanes", dict(one_grid=False, depth_trunc=3.0))]


@pytest.mark.parametrize("name,kw", FWD_CASES)
@pytest.mark.parametrize("impl", ["explicit", "grid_sample"])
def test_forward_matches_reference(name, kw, impl):
    g = load_golden(name)
    cfg = configs.small_test_config(**kw)
    sc = oracle_scene_from_golden(g, cfg)
    assert_close(sc.bound.numpy(), g["bound_ext"], rtol=0, atol=0, what="extended bound")
    rays_o, rays_d, rgb, d, U = fixture_inputs(g)
    ret = sc.forward(rays_o, rays_d, rgb, d, u=U, impl=impl)
    assert_close(ret["z_vals"], g["rr.z_vals"], rtol=0, atol=0, what="z_vals (bit-exact)")
    assert_close(ret["raw"].detach(), g["rr.raw"], rtol=2e-5, atol=2e-6, what="raw")
    for k in ("rgb", "depth"):
        assert_close(ret[k].detach(), g[f"ret.{k}"], rtol=2e-5, atol=2e-6, what=k)
    for k in ("depth_var", "acc_map", "disp_map"):
        assert_close(ret[k].detach(), g[f"rr.{k}"], rtol=1e-4, atol=1e-6, what=k)
    for k in ("rgb_loss", "depth_loss", "co_sdf_loss", "co_fs_loss", "e_fs_loss", "e_center_loss",
              "e_tail_loss", "psnr"):
        assert_close(ret[k].detach(), g[f"ret.{k}"], rtol=2e-5, atol=1e-7, what=k)
    for co in (False, True):
        loss = omap.loss_from_ret(cfg, ret, is_co_sdf=co)
        assert_close(loss.detach(), g[f"loss.co{int(co)}"], rtol=2e-5, what=f"total loss co={co}")


@pytest.mark.parametrize("name,kw", FWD_CASES)
def test_intermediates_and_indices(name, kw):
    g = load_golden(name)
    cfg = configs.small_test_config(**kw)
    sc = oracle_scene_from_golden(g, cfg)
    rays_o, rays_d, rgb, d, U = fixture_inputs(g)
    z = torch.from_numpy(g["rr.z_vals"])
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]).reshape(-1, 3)
    p_nor = normalize_points(pts, sc.bound).float()
    assert_close(p_nor, g["mid.p_nor"], rtol=0, atol=0, what="p_nor (bit-exact)")
    feat = sample_plane_feature(p_nor, *sc.all_planes[:3], impl="explicit")
    assert_close(feat, g["mid.feat"], rtol=1e-5, atol=2e-6, what="feat")
    assert_close(sc.embed_pos(pts), g["mid.pos"], rtol=0, atol=0, what="OneBlob (own spec, bit-exact)")
    w = sc.sdf2weights(torch.from_numpy(g["rr.raw"][..., 3]), z)
    assert_close(w, g["mid.weights"], rtol=1e-6, atol=1e-8, what="weights")
    # integer corner indices: the CUDA-form index ((g+1)/2)*(size-1) (what the HIP path uses)
    # against ATen-CPU's (g+1)*((size-1)/2); they may differ only within 1 ulp of a cell edge.
    for plane, (a, b) in zip([sc.all_planes[0][1], sc.all_planes[1][1], sc.all_planes[2][1]],
                             [(0, 1), (0, 2), (1, 2)]):
        _, _, h, wd = plane.shape
        ix0, iy0, _ = bilinear_corners(p_nor[:, a], p_nor[:, b], h, wd)
        ix_cpu = torch.floor(torch.clamp((p_nor[:, a] + 1) * ((wd - 1) / 2), 0, wd - 1)).long()
        iy_cpu = torch.floor(torch.clamp((p_nor[:, b] + 1) * ((h - 1) / 2), 0, h - 1)).long()
        assert (ix0 != ix_cpu).float().mean() < 1e-3 and (iy0 != iy_cpu).float().mean() < 1e-3
        assert int(ix0.min()) >= 0 and int(ix0.max()) <= wd - 1


@pytest.mark.parametrize("name,kw", FWD_CASES)
@pytest.mark.parametrize("co", [False, True])
def test_gradients_match_reference(name, kw, co):
    g = load_golden(name)
    cfg = configs.small_test_config(**kw)
    sc = oracle_scene_from_golden(g, cfg).requires_grad_(True)
    rays_o, rays_d, rgb, d, U = fixture_inputs(g, requires_grad=True)
    ret = sc.forward(rays_o, rays_d, rgb, d, u=U, impl="explicit")
    omap.loss_from_ret(cfg, ret, is_co_sdf=co).backward()
    tag = f"grad.co{int(co)}."
    for s in range(n_plane_sets(g)):
        for l in range(2):
            ref = g[f"{tag}plane_{s}_{l}"]
            assert_close(sc.all_planes[s][l].grad, ref, rtol=2e-4, atol=1e-6 * max(1.0, np.abs(ref).max()),
                         what=f"plane grad {s},{l}")
    got = dict(zip(DEC_KEYS, sc.col_w + sc.sdf_w))
    for k in DEC_KEYS:
        ref = g[f"{tag}dec.{k}"]
        assert_close(got[k].grad, ref, rtol=2e-4, atol=2e-6 * max(1.0, np.abs(ref).max()), what=f"decoder grad {k}")
    for k, t in (("rays_o", rays_o), ("rays_d", rays_d)):
        ref = g[f"{tag}{k}"]
        assert_close(t.grad, ref, rtol=5e-4, atol=2e-5 * max(1.0, np.abs(ref).max()), what=f"grad {k}")


@pytest.mark.parametrize("name,kw", FWD_CASES)
@pytest.mark.parametrize("co", [False, True])
def test_chunked_forward_backward_matches_reference(name, kw, co):
    """oracle.mapping.forward_backward_chunked (the checker's way to evaluate batches whose autograd graph does not fit in
    host memory, INS Indoor 2048 x 1045) against the REFERENCE's losses and gradients of the golden fixtures, in chunks that
    do not divide the batch."""
    g = load_golden(name)
    cfg = configs.small_test_config(**kw)
    sc = oracle_scene_from_golden(g, cfg).requires_grad_(True)
    rays_o, rays_d, rgb, d, U = fixture_inputs(g)
    z = sc.sample_z(rays_o.shape[0], d, U)
    ret = omap.forward_backward_chunked(sc, cfg, rays_o, rays_d, rgb, d, z, is_co_sdf=co, impl="explicit", chunk=23)
    for k in ("rgb_loss", "depth_loss", "co_sdf_loss", "co_fs_loss", "e_fs_loss", "e_center_loss", "e_tail_loss", "psnr"):
        assert_close(ret[k].reshape(-1), np.asarray(g[f"ret.{k}"]).reshape(-1), rtol=2e-5, atol=1e-7, what=k)
    assert_close(ret["rgb"], g["ret.rgb"], rtol=2e-5, atol=2e-6, what="rgb")
    tag = f"grad.co{int(co)}."
    for s_ in range(n_plane_sets(g)):
        for l in range(2):
            ref = g[f"{tag}plane_{s_}_{l}"]
            assert_close(sc.all_planes[s_][l].grad, ref, rtol=2e-4, atol=1e-6 * max(1.0, np.abs(ref).max()), what=f"plane grad {s_},{l}")
    got = dict(zip(DEC_KEYS, sc.col_w + sc.sdf_w))
    for k in DEC_KEYS:
        ref = g[f"{tag}dec.{k}"]
        assert_close(got[k].grad, ref, rtol=2e-4, atol=2e-6 * max(1.0, np.abs(ref).max()), what=f"decoder grad {k}")


def test_all_invalid_depth_gives_nan_losses():
    g = load_golden("fwd_all_invalid")
    cfg = configs.small_test_config()
    sc = oracle_scene_from_golden(g, cfg)
    rays_o, rays_d, rgb, d, U = fixture_inputs(g)
    ret = sc.forward(rays_o, rays_d, rgb, d, u=U)
    for k in ("rgb_loss", "depth_loss", "co_sdf_loss", "co_fs_loss", "e_fs_loss", "e_center_loss", "e_tail_loss"):
        assert_close(ret[k].detach(), g[f"ret.{k}"], rtol=2e-5, atol=1e-7, what=k)
    assert np.isnan(g["ret.depth_loss"]) and np.isnan(g["ret.e_fs_loss"])
    assert_close(ret["rgb"].detach(), g["ret.rgb"], rtol=2e-5, atol=2e-6, what="rgb")


def test_render_without_depth_and_pose_gradients():
    g = load_golden("render_nodepth")
    cfg = configs.small_test_config()
    sc = oracle_scene_from_golden(g, cfg)
    rays_o, rays_d, rgb, d, U = fixture_inputs(g, requires_grad=True)
    rr = sc.render_rays(rays_o, rays_d, target_d=None, u=U)
    assert_close(rr["z_vals"], g["rr.z_vals"], rtol=0, atol=0, what="z_vals")
    for k in ("rgb", "depth", "raw", "acc_map", "depth_var"):
        assert_close(rr[k].detach(), g[f"rr.{k}"], rtol=1e-4, atol=2e-6, what=k)
    loss = (cfg["training"]["rgb_weight"] * torch.nn.functional.mse_loss(rr["rgb"], rgb)
            + cfg["training"]["depth_weight"] * torch.nn.functional.mse_loss(rr["depth"], d.squeeze()))
    assert_close(loss.detach(), g["loss"], rtol=2e-5, what="loss")
    loss.backward()
    for k, t in (("rays_o", rays_o), ("rays_d", rays_d)):
        ref = g[f"grad.{k}"]
        assert_close(t.grad, ref, rtol=5e-4, atol=2e-5 * max(1.0, np.abs(ref).max()), what=f"grad {k}")


@pytest.mark.parametrize("name,one_grid,co,seed", [("mapping3_onegrid_esdf", True, False, 21),
                                                   ("mapping3_colorplanes_cosdf", False, True, 22)])
def test_three_mapping_iterations_match_reference(name, one_grid, co, seed):
    """End-to-end R1-R12: python-random ray sampling, forward, loss weighting, backward, Adam."""
    g = load_golden(name)
    cfg = configs.small_test_config(one_grid=one_grid, is_co_sdf=co)
    cfg["mapping"].update(sample=64, min_pixels_cur=10, iters=3, n_pixels=0.25)
    H, W, n_save = int(g["H"]), int(g["W"]), int(g["n_save"])
    sc = oracle_scene_from_golden(g, cfg, prefix="init.").requires_grad_(True)
    opt = omap.OracleAdam(sc, cfg)
    direction = torch.from_numpy(g["direction"])
    frames = [dict(frame_id=k, c2w=torch.from_numpy(g[f"frame{k}.c2w"]), rgb=torch.from_numpy(g[f"frame{k}.rgb"]),
                   depth=torch.from_numpy(g[f"frame{k}.depth"]), direction=direction) for k in range(4)]
    random.seed(seed)
    torch.manual_seed(seed)
    kfdb = omap.OracleKeyframeDB(H, W, 8, n_save)
    for k in range(3):
        kfdb.add_keyframe(frames[k], k + 1)
    assert_close(kfdb.rays[:3], g["kf.rays"], rtol=0, atol=0, what="keyframe ray DB (bit-exact)")
    poses = torch.stack([f["c2w"] for f in frames])
    random.seed(seed + 1)
    torch.manual_seed(seed + 1)
    omap.mapping_optimize(sc, opt, cfg, kfdb, frames[3], poses, H, W, impl="explicit")
    for s in range(n_plane_sets(g, "init.")):
        for l in range(2):
            assert_close(sc.all_planes[s][l].detach(), g[f"final.plane_{s}_{l}"], rtol=1e-4, atol=2e-5,
                         what=f"plane {s},{l} after 3 iters")
    got = dict(zip(DEC_KEYS, sc.col_w + sc.sdf_w))
    for k in DEC_KEYS:
        assert_close(got[k].detach(), g[f"final.dec.{k}"], rtol=1e-4, atol=2e-5, what=f"decoder {k} after 3 iters")
    # Adam state of the decoder group (group 0, reference parameter order)
    for pi, k in enumerate(DEC_KEYS):
        assert_close(opt.groups[0].m[pi], g[f"adam.g0.p{pi}.m"], rtol=1e-3, atol=1e-7, what=f"adam m {k}")
        assert float(g[f"adam.g0.p{pi}.step"]) == 3.0
