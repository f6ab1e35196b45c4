"""Parity checks of the HIP path against the oracle and the golden fixtures, written once and run on
two backends: the real library on an MI355X (tests/test_hip_parity_gpu.py, ``-m gpu``) and the
test-only host emulation of the same kernel sources (tests/test_kernels_hostemu.py, CPU).
Every function takes the torch device to run on."""
import contextlib
import copy
import os
import random
import types

import numpy as np
import pytest
import torch

from mneslam_amd import configs, slam_glue
from mneslam_amd.model.keyframe import KeyFrameDatabase
from mneslam_amd.model.scene_rep import JointEncoding
from mneslam_amd.mp_slam.mapper import Mapper
from mneslam_amd.optim import FusedAdam
from oracle import mapping as omap
from oracle.oneblob import oneblob
from oracle.scene_rep import OracleScene

from helpers import (DEC_KEYS, assert_close, fixture_inputs, load_golden, n_plane_sets,
                     oracle_scene_from_golden)

FWD_CASES = {"fwd_onegrid": dict(one_grid=True), "fwd_colorplanes": dict(one_grid=False, depth_trunc=3.0)}
LOSS_KEYS = ("rgb_loss", "depth_loss", "co_sdf_loss", "co_fs_loss", "e_fs_loss", "e_center_loss", "e_tail_loss", "psnr")


def model_from_golden(g, cfg, device, prefix=""):
    """JointEncoding holding the fixture's planes/decoder (planes channels_last on ``device``)."""
    torch.manual_seed(0)
    m = JointEncoding(cfg, torch.from_numpy(g["bounding_box"]).to(device))
    m.device = torch.device(device)
    for s in range(n_plane_sets(g, prefix)):
        for l in range(2):
            t = torch.from_numpy(g[f"{prefix}plane_{s}_{l}"]).to(device)
            m.all_planes[s][l] = t.contiguous(memory_format=torch.channels_last)
    sd = {k: torch.from_numpy(g[f"{prefix}dec.{k}"]) for k in DEC_KEYS}
    m.decoder.load_state_dict(sd)
    return m.to(device)


def to_dev(ts, device):
    return [t.to(device) if t is not None else None for t in ts]


def check_forward(name, device):
    g = load_golden(name)
    cfg = configs.small_test_config(**FWD_CASES[name])
    m = model_from_golden(g, cfg, device).train()
    rays_o, rays_d, rgb, d, U = to_dev(fixture_inputs(g), device)
    out = m._render(rays_o, rays_d, rgb, d, u=U)
    o_rgb, o_depth, o_disp, o_acc, o_var, o_z, o_raw, o_losses = [t.detach().cpu() for t in out]
    assert_close(o_z, g["rr.z_vals"], rtol=0, atol=0, what="z_vals (bit-exact)")
    assert_close(o_raw, g["rr.raw"], rtol=1e-4, atol=1e-5, what="raw")
    assert_close(o_rgb, g["ret.rgb"], rtol=1e-4, atol=1e-5, what="rgb")
    assert_close(o_depth, g["ret.depth"], rtol=1e-4, atol=1e-5, what="depth")
    assert_close(o_var, g["rr.depth_var"], rtol=1e-3, atol=1e-5, what="depth_var")
    assert_close(o_acc, g["rr.acc_map"], rtol=1e-4, atol=1e-5, what="acc")
    assert_close(o_disp, g["rr.disp_map"], rtol=1e-4, atol=1e-5, what="disp")
    for k, key in enumerate(LOSS_KEYS):
        assert_close(o_losses[k], g[f"ret.{key}"].reshape(()), rtol=1e-4, atol=1e-6, what=key)
    # north-star tolerance: depth/colour L1 within 1e-4
    assert float(np.abs(o_rgb.numpy() - g["ret.rgb"]).mean()) < 1e-4
    assert float(np.abs(o_depth.numpy() - g["ret.depth"]).mean()) < 1e-4
    # forward() dict surface
    ret = m.forward(rays_o, rays_d, rgb, d)
    assert set(ret) == {"rgb", "depth", "rgb_loss", "depth_loss", "co_sdf_loss", "co_fs_loss", "e_fs_loss",
                        "e_center_loss", "e_tail_loss", "psnr"}
    assert ret["psnr"].shape == (1,) and ret["rgb_loss"].dim() == 0


def check_backward(name, co, device, wgrad_impl=0):
    g = load_golden(name)
    cfg = configs.small_test_config(**FWD_CASES[name])
    m = model_from_golden(g, cfg, device).train()
    m.wgrad_impl = wgrad_impl
    opt = slam_glue.create_optimizer(m, cfg)
    rays_o, rays_d, rgb, d, U = to_dev(fixture_inputs(g), device)
    out = m._render(rays_o, rays_d, rgb, d, u=U)
    losses = out[-1]
    L = {k: losses[i] for i, k in enumerate(LOSS_KEYS)}
    loss = slam_glue.get_loss_from_ret(cfg, L, is_co_sdf=co)
    assert_close(loss.detach().cpu(), g[f"loss.co{int(co)}"], rtol=1e-4, what="total loss")
    loss.backward()
    tag = f"grad.co{int(co)}."
    for s in range(n_plane_sets(g)):
        for l in range(2):
            ref = g[f"{tag}plane_{s}_{l}"]
            got = m.all_planes[s][l].grad
            assert got is not None, "plane got no gradient"
            assert_close(got.cpu(), ref, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(ref).max()), what=f"plane grad {s},{l}")
    sd = dict(m.decoder.named_parameters())
    for k in DEC_KEYS:
        ref = g[f"{tag}dec.{k}"]
        assert_close(sd[k].grad.cpu(), ref, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(ref).max()), what=f"decoder grad {k}")
    return opt


def check_ray_gradients(name, co, device):
    """R13: d(total)/d rays_o, d rays_d (pose optimisation path) against the reference's autograd."""
    g = load_golden(name)
    cfg = configs.small_test_config(**FWD_CASES[name])
    m = model_from_golden(g, cfg, device).train()
    slam_glue.create_optimizer(m, cfg)
    rays_o, rays_d, rgb, d, U = to_dev(fixture_inputs(g), device)
    rays_o.requires_grad_(True)
    rays_d.requires_grad_(True)
    out = m._render(rays_o, rays_d, rgb, d, u=U)
    L = {k: out[-1][i] for i, k in enumerate(LOSS_KEYS)}
    slam_glue.get_loss_from_ret(cfg, L, is_co_sdf=co).backward()
    tag = f"grad.co{int(co)}."
    for k, t in (("rays_o", rays_o), ("rays_d", rays_d)):
        ref = g[f"{tag}{k}"]
        assert_close(t.grad.cpu(), ref, rtol=2e-3, atol=5e-5 * max(1.0, np.abs(ref).max()), what=f"grad {k}")


def check_render_nodepth_pose_gradients(device):
    """Loop-closure usage (mp_slam/mapper.py:388-408): rgb/depth MSE on render_rays(target_d=None),
    gradients w.r.t. the rays only."""
    g = load_golden("render_nodepth")
    cfg = configs.small_test_config()
    m = model_from_golden(g, cfg, device).eval()
    rays_o, rays_d, rgb, d, U = to_dev(fixture_inputs(g), device)
    rays_o.requires_grad_(True)
    rays_d.requires_grad_(True)
    out = m._render(rays_o, rays_d, None, None, u=U)
    loss = (cfg["training"]["rgb_weight"] * torch.nn.functional.mse_loss(out[0], rgb)
            + cfg["training"]["depth_weight"] * torch.nn.functional.mse_loss(out[1], d.squeeze()))
    assert_close(loss.detach().cpu(), g["loss"], rtol=1e-4, what="loss")
    loss.backward()
    for k, t in (("rays_o", rays_o), ("rays_d", rays_d)):
        ref = g[f"grad.{k}"]
        assert_close(t.grad.cpu(), ref, rtol=2e-3, atol=5e-5 * max(1.0, np.abs(ref).max()), what=f"grad {k}")


def check_all_invalid(device):
    g = load_golden("fwd_all_invalid")
    cfg = configs.small_test_config()
    m = model_from_golden(g, cfg, device).train()
    rays_o, rays_d, rgb, d, U = to_dev(fixture_inputs(g), device)
    losses = m._render(rays_o, rays_d, rgb, d, u=U)[-1].detach().cpu()
    for k, key in enumerate(LOSS_KEYS):
        assert_close(losses[k], g[f"ret.{key}"].reshape(()), rtol=1e-4, atol=1e-6, what=key)
    assert torch.isnan(losses[1]) and torch.isnan(losses[4])


def check_render_nodepth(device):
    g = load_golden("render_nodepth")
    cfg = configs.small_test_config()
    m = model_from_golden(g, cfg, device).eval()
    rays_o, rays_d, rgb, d, U = to_dev(fixture_inputs(g), device)
    out = m._render(rays_o, rays_d, None, None, u=U)
    assert_close(out[5].cpu(), g["rr.z_vals"], rtol=0, atol=0, what="z_vals")
    assert_close(out[0].detach().cpu(), g["rr.rgb"], rtol=1e-4, atol=1e-5, what="rgb")
    assert_close(out[1].detach().cpu(), g["rr.depth"], rtol=1e-4, atol=1e-5, what="depth")
    assert_close(out[6].cpu(), g["rr.raw"], rtol=1e-4, atol=1e-5, what="raw")
    rr = m.render_rays(rays_o, rays_d, target_d=None)
    assert set(rr) == {"rgb", "depth", "disp_map", "acc_map", "depth_var", "z_vals", "raw"}


def check_mapping3(name, one_grid, co, seed, device, compute="autograd", scatter="binned", **fused_kwargs):
    """Three drop-in Mapper.mapping_optimize iterations (host RNG sampling, autograd path, FusedAdam)
    against the parameters the REFERENCE reached from the same state and seeds."""
    g = load_golden(name)
    cfg = configs.small_test_config(one_grid=one_grid, is_co_sdf=co)
    cfg["mapping"].update(sample=64, min_pixels_cur=10, iters=3, n_pixels=0.25)
    H, W, n_save = int(g["H"]), int(g["W"]), int(g["n_save"])
    m = model_from_golden(g, cfg, device, prefix="init.").train()
    opt = slam_glue.create_optimizer(m, cfg)
    direction = torch.from_numpy(g["direction"])
    frames = [dict(frame_id=k, c2w=torch.from_numpy(g[f"frame{k}.c2w"]), rgb=torch.from_numpy(g[f"frame{k}.rgb"]),
                   depth=torch.from_numpy(g[f"frame{k}.depth"]), direction=direction) for k in range(4)]
    random.seed(seed)
    torch.manual_seed(seed)
    kfdb = KeyFrameDatabase(cfg, H, W, 8, n_save, device)
    for k in range(3):
        kfdb.add_keyframe(frames[k], k + 1)
    assert_close(kfdb.rays[:3], g["kf.rays"], rtol=0, atol=0, what="keyframe ray DB")
    slam = types.SimpleNamespace(
        config=cfg, model=m, map_optimizer=opt, device=torch.device(device),
        dataset=types.SimpleNamespace(H=H, W=W), video=types.SimpleNamespace(keyframe=kfdb),
        get_loss_from_ret=lambda ret, **kw: slam_glue.get_loss_from_ret(cfg, ret, **kw),
        select_samples=slam_glue.select_samples)
    mapper = Mapper(cfg, slam, compute=compute, sampler="host", scatter=scatter)
    mapper.fused_kwargs = fused_kwargs
    poses = torch.stack([f["c2w"] for f in frames]).to(device)
    random.seed(seed + 1)
    torch.manual_seed(seed + 1)
    if compute == "fused":
        # the step is sized for the largest batch the loop can ask for, the batches here are smaller: whatever the decoder
        # update sums must have been written by the iteration itself (zero-filled allocations hide stale slots)
        fs0 = mapper._fused_step(cfg["mapping"]["sample"] + cfg["mapping"]["min_pixels_cur"])
        fs0.partials.fill_(float("nan"))
        fs0.tape.fill_(float("nan"))
    mapper.optimize_map(frames[3], poses)
    for s in range(n_plane_sets(g, "init.")):
        for l in range(2):
            assert_close(m.all_planes[s][l].detach().cpu(), g[f"final.plane_{s}_{l}"], rtol=1e-3, atol=1e-4,
                         what=f"plane {s},{l} after 3 iterations")
    sd = dict(m.decoder.named_parameters())
    for k in DEC_KEYS:
        assert_close(sd[k].detach().cpu(), g[f"final.dec.{k}"], rtol=1e-3, atol=1e-4, what=f"decoder {k} after 3 iterations")


def check_device_sampler(device):
    """mne_sample_rays: explicit indices reproduce the reference's ray assembly bit-exactly; the
    device permutation draws distinct, in-range, well-spread indices, reproducibly per (seed, it)."""
    import ctypes as C
    from mneslam_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(4)
    n_kf, n_save, HW, n_g, n_c = 5, 300, 777, 256, 40
    kf_rays = torch.randn(n_kf * n_save, 7, generator=gen)
    cur = torch.randn(HW, 7, generator=gen)
    poses = torch.randn(n_kf + 1, 4, 4, generator=gen)
    idx_g = torch.randperm(n_kf * n_save, generator=gen)[:n_g]
    idx_c = torch.randperm(HW, generator=gen)[:n_c]
    rays7 = torch.cat([kf_rays[idx_g], cur[idx_c]], 0)
    ids = torch.cat([torch.div(idx_g, n_save, rounding_mode="trunc"), -torch.ones(n_c, dtype=torch.int64)])
    ref_o, ref_d, ref_rgb, ref_dep = omap.assemble_rays(rays7, ids, poses)
    R = n_g + n_c
    d = lambda t: t.to(device).contiguous()
    out = [torch.empty(R, 3, device=device), torch.empty(R, 3, device=device), torch.empty(R, 3, device=device),
           torch.empty(R, device=device)]
    oidx = torch.empty(R, dtype=torch.int64, device=device)
    a = [d(kf_rays), d(cur), d(poses), d(idx_g), d(idx_c)]
    P = _lib.ptr
    _lib.check(lib.mne_sample_rays(P(a[0]), n_kf * n_save, n_save, None, P(a[1]), HW, P(a[2]), n_kf + 1, n_g, n_c,
                                   P(a[3]), P(a[4]), 0, 0, P(out[0]), P(out[1]), P(out[2]), P(out[3]), P(oidx),
                                   None, _lib.stream_for(out[0])))
    assert_close(out[0].cpu(), ref_o, rtol=0, atol=0, what="rays_o (bit-exact)")
    assert_close(out[1].cpu(), ref_d, rtol=0, atol=0, what="rays_d (bit-exact)")
    assert_close(out[2].cpu(), ref_rgb, rtol=0, atol=0, what="target rgb")
    assert_close(out[3].cpu(), ref_dep[:, 0], rtol=0, atol=0, what="target depth")
    seen = []
    for it in range(3):
        _lib.check(lib.mne_sample_rays(P(a[0]), n_kf * n_save, n_save, None, P(a[1]), HW, P(a[2]), n_kf + 1, n_g, n_c,
                                       None, None, 1234, it, P(out[0]), P(out[1]), P(out[2]), P(out[3]), P(oidx),
                                       None, _lib.stream_for(out[0])))
        ii = oidx.cpu()
        g, c = ii[:n_g], ii[n_g:]
        assert g.unique().numel() == n_g and int(g.min()) >= 0 and int(g.max()) < n_kf * n_save
        assert c.unique().numel() == n_c and int(c.min()) >= 0 and int(c.max()) < HW
        assert_close(out[3].cpu(), torch.cat([kf_rays[g, 6], cur[c, 6]]), rtol=0, atol=0, what="gathered depth")
        seen.append(ii.clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    _lib.check(lib.mne_sample_rays(P(a[0]), n_kf * n_save, n_save, None, P(a[1]), HW, P(a[2]), n_kf + 1, n_g, n_c,
                                   None, None, 1234, 0, P(out[0]), P(out[1]), P(out[2]), P(out[3]), P(oidx),
                                   None, _lib.stream_for(out[0])))
    assert torch.equal(oidx.cpu(), seen[0]), "same (seed, iteration) must give the same batch"
    # spread: owners of the global rows cover every keyframe roughly evenly
    own = torch.bincount(torch.div(torch.cat(seen)[: 3 * n_g].reshape(3, -1)[:, :n_g].reshape(-1), n_save, rounding_mode="trunc"),
                         minlength=n_kf).float()
    assert own.min() > 0.5 * own.mean()


def check_adam(device):
    """mne_adam_step against the written-out Adam of the oracle, incl. weight decay, odd sizes,
    channels_last tensors and gradient zeroing."""
    torch.manual_seed(3)
    shapes = [(1, 32, 9, 7), (16, 32), (3, 33), (5,)]
    ps = [torch.randn(s) for s in shapes]
    ps[0] = ps[0].contiguous(memory_format=torch.channels_last)
    ref_p = [p.clone() for p in ps]
    dev_p = [torch.nn.Parameter(p.clone().to(device)) for p in ps]
    opt = FusedAdam([{"params": dev_p[:2], "lr": 0.005, "eps": 1e-15},
                     {"params": dev_p[2:], "lr": 0.01, "weight_decay": 1e-6}], betas=(0.9, 0.99))
    grp = [omap.AdamGroup(ref_p[:2], 0.005, eps=1e-15), omap.AdamGroup(ref_p[2:], 0.01, eps=1e-8, weight_decay=1e-6)]
    for it in range(4):
        gs = [torch.randn(s) * (0.0 if (it == 2 and k == 1) else 1.0) for k, s in enumerate(shapes)]
        gs[0] = gs[0].contiguous(memory_format=torch.channels_last)
        for p, gr in zip(dev_p, gs):
            p.grad = gr.clone().to(device)
        opt.step(zero_grad=(it == 3))
        for gq in grp:
            gq.t += 1
        for k, (p, gr) in enumerate(zip(ref_p, gs)):
            q = grp[0] if k < 2 else grp[1]
            j = k if k < 2 else k - 2
            b1, b2 = q.betas
            gg = gr + q.wd * p if q.wd else gr
            q.m[j].add_((gg - q.m[j]) * (1 - b1))
            q.v[j].mul_(b2).add_(gg * gg * (1 - b2))
            den = q.v[j].sqrt() / np.sqrt(1 - b2 ** q.t) + q.eps
            p.add_(-(q.lr / (1 - b1 ** q.t)) * (q.m[j] / den))
    for p, r in zip(dev_p, ref_p):
        assert_close(p.detach().cpu(), r, rtol=1e-5, atol=1e-6, what="adam param")
        assert float(p.grad.abs().max()) == 0.0, "zero_grad=True must clear the gradient buffer"


def check_oneblob(device):
    from mneslam_amd.model.encodings import get_encoder
    enc, dim = get_encoder("OneBlob", n_bins=16)
    assert dim == 48 and enc.params.numel() == 0
    torch.manual_seed(0)
    x = torch.cat([torch.rand(200, 3), torch.tensor([[0.0, 1.0, 0.5], [-0.3, 1.7, 0.03125], [0.999, 0.001, 0.0625]])])
    got = enc(x.double().to(device)).cpu()          # double input is cast to fp32 like tinycudann
    assert_close(got, oneblob(x, 16), rtol=0, atol=0, what="OneBlob (bit-exact vs spec)")
    # inputs strictly inside the unit interval, including the bin edges next to its ends
    xi = torch.rand(256, 3) * (254.0 / 256.0) + 1.0 / 256.0
    xi = xi.clamp(1.0 / 256.0, 255.0 / 256.0)
    xi[:4] = torch.tensor([[1 / 256, 255 / 256, 0.5], [255 / 256, 1 / 256, 15 / 16], [1 / 16, 0.9375 + 1e-4, 0.06], [0.004, 0.996, 0.9]])
    got = enc(xi.double().to(device)).cpu()
    assert_close(got, oneblob(xi, 16), rtol=0, atol=0, what="OneBlob inside the box (bit-exact vs spec)")


def check_misc_encodings(device):
    """The remaining branches of the reference's encoder factory (model/encodings.py:48-58, 73-95): spherical harmonics,
    frequency, identity -- forward and input gradient against the frozen spec (oracle/encodings_misc.py: parity unpinned)."""
    from mneslam_amd.model.encodings import get_encoder
    from oracle import encodings_misc as em
    torch.manual_seed(3)
    x = torch.cat([torch.rand(300, 3), torch.tensor([[0.0, 1.0, 0.5], [0.25, 0.75, 1.0]])])
    # ---- frequency (factory default: 12 frequencies)
    enc, dim = get_encoder("Frequency", n_frequencies=12)
    assert dim == 72 and enc.n_output_dims == 72 and enc.params.numel() == 0
    xr = x.clone().to(device).requires_grad_(True)
    got = enc(xr)
    xo = x.clone().requires_grad_(True)
    ref = em.frequency(xo, 12)
    # sin of arguments up to 2^11 pi: one fp32 ulp of the ARGUMENT is 2.4e-4 there, the device's sinf and torch's agree to ~1e-6 on the same argument
    assert_close(got.detach().cpu(), ref.detach(), rtol=0, atol=2e-6, what="frequency encoding")
    g = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1))
    got.backward(g.to(device))
    ref.backward(g)
    assert_close(xr.grad.cpu(), xo.grad, rtol=1e-4, atol=1e-2, what="frequency input gradient")       # (gradients scale with 2^11 pi)
    enc4, dim4 = get_encoder("frequency", input_dim=2, n_frequencies=4)
    assert dim4 == 16
    assert_close(enc4(x[:, :2].contiguous().to(device)).cpu(), em.frequency(x[:, :2], 4), rtol=0, atol=1e-6, what="frequency 2-D")
    # ---- spherical harmonics (factory default: degree 4)
    for degree in (4, 2, 1):
        enc, dim = get_encoder("SphericalHarmonics", degree=degree)
        assert dim == degree * degree and enc.params.numel() == 0
        xr = x.clone().to(device).requires_grad_(True)
        got = enc(xr)
        xo = x.clone().requires_grad_(True)
        ref = em.spherical_harmonics(xo, degree)
        assert_close(got.detach().cpu(), ref.detach(), rtol=1e-6, atol=1e-6, what=f"SH degree {degree}")
        g = torch.randn(ref.shape, generator=torch.Generator().manual_seed(2))
        got.backward(g.to(device))
        if degree == 1:                                   # a constant: no gradient
            assert float(xr.grad.abs().max()) == 0.0
            continue
        ref.backward(g)
        assert_close(xr.grad.cpu(), xo.grad, rtol=1e-5, atol=1e-5, what=f"SH degree {degree} input gradient")
    # ---- identity
    enc, dim = get_encoder("Identity")
    assert dim == 3 and enc.params.numel() == 0
    xr = x.clone().to(device).requires_grad_(True)
    got = enc(xr)
    assert torch.equal(got.detach().cpu(), em.identity(x))
    got.sum().backward()
    assert torch.equal(xr.grad.cpu(), torch.ones_like(x))
    with pytest.raises(ValueError):
        get_encoder("no-such-encoding")


def check_grid_encoding(device, kind="hash"):
    """R14 surface: get_encoder('HashGrid'/'dense') vs the frozen spec (oracle/hashgrid.py): uint32 table
    indices bit-exact, features and parameter gradients to fp32 rounding."""
    from mneslam_amd.model.encodings import get_encoder
    from oracle import hashgrid as og
    torch.manual_seed(7)
    if kind == "hash":
        enc, dim = get_encoder("HashGrid", n_levels=8, level_dim=2, base_resolution=16, log2_hashmap_size=12,
                               desired_resolution=256)
        kw = dict(n_levels=8, n_features=2, base_resolution=16, per_level_scale=float(np.exp2(np.log2(256 / 16) / 7)),
                  log2_hashmap_size=12, grid_type="hash")
    else:
        enc, dim = get_encoder("dense", level_dim=2, base_resolution=16, desired_resolution=16)
        kw = dict(n_levels=4, n_features=2, base_resolution=16, per_level_scale=1.0, log2_hashmap_size=19, grid_type="dense")
    assert dim == kw["n_levels"] * 2
    assert enc.params.numel() == og.n_params(**kw)
    with torch.no_grad():
        enc.params.copy_(torch.randn(enc.params.numel()) * 0.3)
    enc = enc.to(device)
    x = torch.rand(300, 3)
    x[:4] = torch.tensor([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.5, 0.25, 0.75], [0.999999, 0.0, 1.0]])
    p_ref = enc.params.detach().cpu().clone().requires_grad_(True)
    lib_scales, lib_res, lib_sizes, lib_off = enc.level_table()
    o_scales, o_res, o_sizes, o_off = og.level_table(kw["n_levels"], kw["base_resolution"], kw["per_level_scale"],
                                                     kw["log2_hashmap_size"], kw["grid_type"])
    assert lib_res == o_res and lib_sizes == o_sizes and lib_off == o_off[:-1]
    assert np.allclose(lib_scales, o_scales, rtol=3e-7, atol=0)          # exp2f of two libms: <= 1 ulp apart
    ref, ref_idx = og.grid_encode(x, p_ref, return_indices=True, scales=lib_scales, **kw)
    got = enc(x.to(device))
    idx = enc.indices(x.to(device)).cpu()
    assert torch.equal(idx, ref_idx), "integer grid/hash indices must be bit-exact"
    assert_close(got.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6, what="grid features")
    wts = torch.randn(300, dim)
    (ref * wts).sum().backward()
    (got * wts.to(device)).sum().backward()
    assert_close(enc.params.grad.cpu(), p_ref.grad, rtol=1e-4, atol=1e-5, what="grid parameter gradients")
    # the BASELINE configuration's shape: 16 levels, T = 2^19, F = 2 -> 10,492,048 parameters, 32 features
    big, bdim = get_encoder("HashGrid")
    assert bdim == 32 and big.params.numel() == 10492048


def check_queries(device):
    g = load_golden("fwd_colorplanes")
    cfg = configs.small_test_config(one_grid=False, depth_trunc=3.0)
    m = model_from_golden(g, cfg, device).eval()
    sc = oracle_scene_from_golden(g, cfg)
    torch.manual_seed(1)
    pts = (torch.rand(150, 3) * 2.4 - 1.2)          # some points outside the bound (border clamp)
    raw = m.query_color_sdf(pts.to(device)).cpu()
    ref, parts = sc.query_color_sdf(pts, return_parts=True)
    assert_close(raw, ref, rtol=1e-4, atol=1e-5, what="query_color_sdf")
    sdf, geo = m.query_sdf(pts.reshape(10, 15, 3).to(device), return_geo=True)
    rs, rg = sc.query_sdf(pts.reshape(10, 15, 3), return_geo=True)
    assert_close(sdf.cpu(), rs, rtol=1e-4, atol=1e-5, what="query_sdf")
    assert_close(geo.cpu(), rg, rtol=1e-4, atol=1e-5, what="query_sdf geo")
    emb = m.query_sdf(pts.to(device), embed=True).cpu()
    assert_close(emb, parts["feat"], rtol=1e-5, atol=1e-6, what="embed")
    feat = m.sample_plane_feature(parts["p_nor"].to(device), *m.all_planes[:3]).cpu()
    assert_close(feat, parts["feat"], rtol=1e-5, atol=1e-6, what="sample_plane_feature")
    assert_close(m.query_color(pts.to(device)).cpu(), torch.sigmoid(ref[..., :3]), rtol=1e-4, atol=1e-5, what="query_color")


def check_corner_indices(device, name="fwd_colorplanes"):
    """R6 headline: the integer NW corner (ix0, iy0) of every bilinear footprint, exported by the HIP gather
    itself (mne_query_points corner_idx), must EQUAL oracle.scene_rep.bilinear_corners for every plane and
    level -- on the reference's own normalised points (golden mid.p_nor) and on adversarial points sitting
    exactly on, and one ulp either side of, every cell edge of every plane, plus border / out-of-range values."""
    from oracle.scene_rep import bilinear_corners
    g = load_golden(name)
    cfg = configs.small_test_config(**FWD_CASES[name])
    m = model_from_golden(g, cfg, device).eval()
    planes = [p for lst in m.all_planes for p in lst]
    p_nor = [torch.from_numpy(g["mid.p_nor"])]
    # cell edges: g = 2 k / (size - 1) - 1 and its fp32 neighbours, for every distinct plane extent
    sizes = sorted({int(s) for p in planes for s in p.shape[2:]})
    edge = []
    for n in sizes:
        k = torch.arange(0, n, dtype=torch.float32)
        e = (k * 2.0) / float(n - 1) - 1.0
        edge += [e, torch.nextafter(e, torch.full_like(e, 2.0)), torch.nextafter(e, torch.full_like(e, -2.0))]
    edge = torch.cat(edge + [torch.tensor([-1.5, -1.0, 1.0, 1.5, 0.0, -0.0, 1.0000001, -1.0000001, 0.99999994])])
    gen = torch.Generator().manual_seed(11)
    for _ in range(3):      # every edge value on every axis, paired with shuffled edge values on the other two
        cols = [edge[torch.randperm(edge.numel(), generator=gen)] for _ in range(3)]
        p_nor.append(torch.stack(cols, -1))
    p_nor = torch.cat(p_nor, 0).contiguous()
    info = m._info()
    out = __import__("mneslam_amd.hip_path", fromlist=["x"]).query_points(
        info, [p.to(device) for p in planes], m.decoder.hip_weights(), p_nor.to(device), want_raw=False,
        want_feat=True, normalised=True, want_corner_idx=True)
    cidx = out[3].cpu()
    n_sets = len(planes) // 6
    assert cidx.shape == (p_nor.shape[0], 3 * n_sets, 2, 2) and cidx.dtype == torch.int32
    axes = [(0, 1), (0, 2), (1, 2)]
    checked = 0
    for s in range(n_sets):
        for ori in range(3):
            for lvl in range(2):
                pl = m.all_planes[s * 3 + ori][lvl]
                h, w = pl.shape[2], pl.shape[3]
                a, b = axes[ori]
                ix0, iy0, _ = bilinear_corners(p_nor[:, a], p_nor[:, b], h, w)
                assert torch.equal(cidx[:, s * 3 + ori, lvl, 0].long(), ix0), f"ix0 differs: set {s} ori {ori} lvl {lvl}"
                assert torch.equal(cidx[:, s * 3 + ori, lvl, 1].long(), iy0), f"iy0 differs: set {s} ori {ori} lvl {lvl}"
                checked += ix0.numel() * 2
    assert checked > 0
    return checked


def check_oracle_random_scene(device, hidden=32, one_grid=True, n_rays=24, S_d=20, S_r=9, seed=5, invalid_every=5):
    """Seeded random scene at a configuration without fixture (e.g. hidden 64): HIP vs oracle,
    forward and gradients."""
    cfg = configs.small_test_config(one_grid=one_grid, n_samples_d=S_d, n_range_d=S_r)
    cfg["decoder"]["hidden_dim"] = cfg["decoder"]["hidden_dim_color"] = hidden
    gen = torch.Generator().manual_seed(seed)
    bb = torch.tensor(cfg["mapping"]["bound"], dtype=torch.float64)
    sc = OracleScene(cfg, bb, generator=gen)
    for lst in sc.all_planes:
        for i in range(len(lst)):
            lst[i] = lst[i] * 20.0
    sc.requires_grad_(True)
    torch.manual_seed(seed)
    m = JointEncoding(cfg, bb.to(device))
    m.device = torch.device(device)
    for s, lst in enumerate(sc.all_planes):
        for l, p in enumerate(lst):
            m.all_planes[s][l] = p.detach().clone().to(device).contiguous(memory_format=torch.channels_last)
    m.decoder.load_state_dict(dict(zip(DEC_KEYS, [w.detach().clone() for w in sc.col_w + sc.sdf_w])))
    m = m.to(device).train()
    slam_glue.create_optimizer(m, cfg)
    rays_o = (torch.rand(n_rays, 3, generator=gen) - 0.5) * 0.6
    rays_d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=gen), dim=-1) * 0.7
    rgb = torch.rand(n_rays, 3, generator=gen)
    d = torch.rand(n_rays, 1, generator=gen) * 2.0 + 0.2
    if invalid_every:
        d[::invalid_every] = 0.0
    U = torch.rand(n_rays, S_d + S_r, generator=gen)
    ref = sc.forward(rays_o, rays_d, rgb, d, u=U)
    out = m._render(*to_dev([rays_o, rays_d, rgb, d], device), u=U.to(device))
    assert_close(out[5].cpu(), ref["z_vals"], rtol=0, atol=0, what="z_vals")
    assert_close(out[6].cpu(), ref["raw"].detach(), rtol=1e-4, atol=1e-5, what="raw")
    assert_close(out[0].detach().cpu(), ref["rgb"].detach(), rtol=1e-4, atol=1e-5, what="rgb")
    assert_close(out[1].detach().cpu(), ref["depth"].detach(), rtol=1e-4, atol=1e-5, what="depth")
    for co in (False, True):
        for t in sc.plane_list() + sc.decoder_list():
            t.grad = None
        m.zero_grad()
        for lst in m.all_planes:
            for p in lst:
                p.grad = None
        ref = sc.forward(rays_o, rays_d, rgb, d, u=U)
        omap.loss_from_ret(cfg, ref, is_co_sdf=co).backward()
        out = m._render(*to_dev([rays_o, rays_d, rgb, d], device), u=U.to(device))
        L = {k: out[-1][i] for i, k in enumerate(LOSS_KEYS)}
        slam_glue.get_loss_from_ret(cfg, L, is_co_sdf=co).backward()
        for s, lst in enumerate(sc.all_planes):
            for l, p in enumerate(lst):
                r = p.grad.numpy()
                assert_close(m.all_planes[s][l].grad.cpu(), r, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(r).max()),
                             what=f"plane grad {s},{l} co={co}")
        got = dict(m.decoder.named_parameters())
        for k, w in zip(DEC_KEYS, sc.col_w + sc.sdf_w):
            r = w.grad.numpy()
            assert_close(got[k].grad.cpu(), r, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(r).max()), what=f"dec grad {k} co={co}")


# ---------------------------------------------------------------- N2: loop-closure loops
def _rodrigues(rot):
    """axis-angle [B,3] -> rotation matrices [B,3,3] (differentiable; test stand-in for the host's pose code)."""
    th = torch.sqrt((rot * rot).sum(-1) + 1e-12)[:, None, None]
    k = rot / th[:, :, 0]
    K = torch.zeros(rot.shape[0], 3, 3, dtype=rot.dtype, device=rot.device)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0] = -k[:, 2], k[:, 1], k[:, 2]
    K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 0], -k[:, 1], k[:, 0]
    eye = torch.eye(3, dtype=rot.dtype, device=rot.device)[None]
    return eye + torch.sin(th) * K + (1.0 - torch.cos(th)) * (K @ K)


class _PoseSLAM:
    """The two MNESLAM methods the alignment loop calls (mneslam_mp.py:577-584), axis-angle about the
    initial rotation so that no matrix logarithm is needed in the test."""

    def __init__(self, lr_rot=1e-3, lr_trans=1e-3):
        self.lr_rot, self.lr_trans, self.R0 = lr_rot, lr_trans, None

    def get_pose_param_optim(self, poses, mapping=True):
        self.R0 = poses[:, :3, :3].detach().clone()
        cur_trans = torch.nn.Parameter(poses[:, :3, 3].detach().clone())
        cur_rot = torch.nn.Parameter(torch.zeros(poses.shape[0], 3, dtype=poses.dtype, device=poses.device))
        opt = torch.optim.Adam([{"params": cur_rot, "lr": self.lr_rot}, {"params": cur_trans, "lr": self.lr_trans}])
        return cur_rot, cur_trans, opt

    def matrix_from_tensor(self, rot, trans):
        T = torch.eye(4, dtype=rot.dtype, device=rot.device)[None].repeat(rot.shape[0], 1, 1)
        T[:, :3, :3] = _rodrigues(rot) @ self.R0
        T[:, :3, 3] = trans
        return T


class _OracleModel:
    """render_rays of the CPU oracle behind the JointEncoding call signature."""

    def __init__(self, scene):
        self.scene = scene

    def render_rays(self, rays_o, rays_d, target_d=None):
        return self.scene.render_rays(rays_o, rays_d, target_d=target_d)


class _PoseSLAMAbsolute(_PoseSLAM):
    """The reference's own parameterisation (rot_rep 'axis_angle', mneslam_mp.py:199-201, :577-584): the rotation
    vector of the pose itself, c2w = [Rodrigues(rot) | trans]."""

    def get_pose_param_optim(self, poses, mapping=True):
        R = poses[0, :3, :3].detach().double()
        ang = torch.acos(torch.clamp((torch.trace(R) - 1.0) / 2.0, -1.0, 1.0))
        axis = torch.stack([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2.0 * torch.sin(ang))
        cur_rot = torch.nn.Parameter((axis * ang).to(poses.dtype)[None].clone())
        cur_trans = torch.nn.Parameter(poses[:, :3, 3].detach().clone())
        opt = torch.optim.Adam([{"params": cur_rot, "lr": self.lr_rot}, {"params": cur_trans, "lr": self.lr_trans}])
        return cur_rot, cur_trans, opt

    def matrix_from_tensor(self, rot, trans):
        T = torch.eye(4, dtype=rot.dtype, device=rot.device)[None].repeat(rot.shape[0], 1, 1)
        T[:, :3, :3] = _rodrigues(rot)
        T[:, :3, 3] = trans
        return T


class _PoseSLAMQuat(_PoseSLAM):
    """rot_rep 'quat' (mneslam_mp.py:203-206): the pose's quaternion (real part first), c2w = [R(q) | trans] with
    pytorch3d's normalising quaternion_to_matrix (optimization/utils.py:199-210)."""

    def get_pose_param_optim(self, poses, mapping=True):
        R = poses[0, :3, :3].detach().double()
        w = torch.sqrt(torch.clamp(1.0 + torch.trace(R), min=1e-12)) / 2.0          # (small rotations: the trace branch)
        q = torch.stack([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])
        cur_rot = torch.nn.Parameter(q.to(poses.dtype)[None].clone())
        cur_trans = torch.nn.Parameter(poses[:, :3, 3].detach().clone())
        opt = torch.optim.Adam([{"params": cur_rot, "lr": self.lr_rot}, {"params": cur_trans, "lr": self.lr_trans}])
        return cur_rot, cur_trans, opt

    def matrix_from_tensor(self, rot, trans):
        r, i, j, k = torch.unbind(rot, -1)
        s = 2.0 / (rot * rot).sum(-1)
        R = torch.stack([1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r),
                         s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r),
                         s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j)], -1).reshape(-1, 3, 3)
        T = torch.eye(4, dtype=rot.dtype, device=rot.device)[None].repeat(rot.shape[0], 1, 1)
        T[:, :3, :3] = R
        T[:, :3, 3] = trans
        return T


def _alignment_setup(g, device):
    rays_o, rays_d, *_ = fixture_inputs(g)
    cam_dirs = rays_d[:48].clone()                                   # any fixed bundle of camera-frame directions
    base = torch.eye(4)
    base[:3, 3] = rays_o[0]
    target0 = base.clone()
    target0[:3, :3] = _rodrigues(torch.tensor([[0.02, -0.015, 0.01]]))[0]
    target0[:3, 3] += torch.tensor([0.03, -0.02, 0.025])
    return cam_dirs, base, target0


def check_pose_alignment(device, compute="autograd", absolute=False):
    """Mapper.optimize_relative_pose (reference loop: mp_slam/mapper.py:362-412) on the HIP path vs the same loop
    driven by the CPU oracle's autograd: same seeds -> same jitter draws -> same pose trajectory.
    compute="fused": the device loop (csrc/pose.hip: no autograd graph, no torch.optim step); ``absolute``: the reference's
    own axis-angle parameterisation instead of one relative to the initial rotation."""
    g = load_golden("render_nodepth")
    cfg = configs.small_test_config()
    cfg["mapping"]["loop_iters"] = 4
    cam_dirs, base, target0 = _alignment_setup(g, device)
    results = []
    for kind in ("hip", "oracle"):
        if kind == "hip":
            model, dev = model_from_golden(g, cfg, device).eval(), device
        else:
            model, dev = _OracleModel(oracle_scene_from_golden(g, cfg)), "cpu"
        slam = types.SimpleNamespace(model=model, model_shared=model, map_optimizer=None, device=torch.device(dev),
                                     dataset=None, video=None, get_pose_param_optim=None, matrix_from_tensor=None)
        pose = _PoseSLAMQuat() if absolute == "quat" else _PoseSLAMAbsolute() if absolute else _PoseSLAM()
        slam.get_pose_param_optim, slam.matrix_from_tensor = pose.get_pose_param_optim, pose.matrix_from_tensor
        mp = Mapper(cfg, slam, compute=compute if kind == "hip" else "autograd")
        torch.manual_seed(11)
        rel, best = mp.optimize_relative_pose(base.clone(), target0.clone(), model, model, rays_d_cam_batch=cam_dirs.clone())
        if kind == "hip":
            assert mp.last_pose_loop == ("device" if compute == "fused" else "host")
        results.append((rel.detach().cpu(), best))
    (rel_h, best_h), (rel_o, best_o) = results
    assert best_h == best_h and abs(best_h - best_o) <= 1e-3 * abs(best_o) + 1e-7, (best_h, best_o)
    assert_close(rel_h, rel_o, rtol=1e-4, atol=2e-5, what="relative transform")
    assert not torch.allclose(rel_h, base @ torch.inverse(target0))     # the pose moved


def check_pose_alignment_hash(device, compute="fused"):
    """The same loop on the hash-grid model (R13 through the grid: OneBlob share from the render backward + trilinear-weight
    share from mne_hash_ray_grad): Mapper.optimize_relative_pose against the loop driven by the oracle's autograd --
    compute "fused": the DEVICE loop (hip_path.PoseAlignment on caller-supplied grid features: nine launches per iteration,
    round 5), "autograd": the host's own autograd loop through HashRenderFunction."""
    cfg = hash_test_config(hash_size=12, hidden=32, desired_resolution=128)
    cfg["mapping"]["loop_iters"] = 3
    cfg["training"]["n_samples"] = 40
    m, sc = _hash_model_and_oracle(device, cfg)
    rays_o, rays_d, _, _, fr = _synthetic_rays(cfg, 40, img=(12, 20))
    cam_dirs = fr["direction"].reshape(-1, 3)[:40].clone()
    base = fr["c2w"].clone()
    target0 = base.clone()
    target0[:3, :3] = _rodrigues(torch.tensor([[0.02, -0.015, 0.01]]))[0] @ base[:3, :3]
    target0[:3, 3] += torch.tensor([0.03, -0.02, 0.025])
    results = []
    for kind in ("hip", "oracle"):
        model, dev = (m.eval(), device) if kind == "hip" else (_OracleModel(sc), "cpu")
        slam = types.SimpleNamespace(model=model, model_shared=model, map_optimizer=None, device=torch.device(dev),
                                     dataset=None, video=None, get_pose_param_optim=None, matrix_from_tensor=None)
        pose = _PoseSLAM()
        slam.get_pose_param_optim, slam.matrix_from_tensor = pose.get_pose_param_optim, pose.matrix_from_tensor
        mp = Mapper(cfg, slam, compute=compute if kind == "hip" else "autograd")
        torch.manual_seed(11)
        rel, best = mp.optimize_relative_pose(base.clone(), target0.clone(), model, model, rays_d_cam_batch=cam_dirs.clone())
        if kind == "hip":
            assert mp.last_pose_loop == ("device" if compute == "fused" else "host")
        results.append((rel.detach().cpu(), best))
    (rel_h, best_h), (rel_o, best_o) = results
    assert best_h == best_h and abs(best_h - best_o) <= 1e-3 * abs(best_o) + 1e-7, (best_h, best_o)
    assert_close(rel_h, rel_o, rtol=1e-4, atol=5e-5, what="relative transform (hash model)")
    assert not torch.allclose(rel_h, base @ torch.inverse(target0))     # the pose moved


def check_distillation(device, compute="autograd"):
    """Mapper.distillation (reference loop: mp_slam/mapper.py:598-640): teacher = model_shared, student = model; three
    iterations on the HIP path vs the same loop on the CPU oracle (oracle forward/backward + OracleAdam).
    compute="fused": the device loop (one teacher render_maps + one FusedStep iteration, no autograd, no torch.optim)."""
    g = load_golden("mapping3_onegrid_esdf")
    cfg = configs.small_test_config(one_grid=True, is_co_sdf=False)
    cfg["mapping"].update(sample=64, min_pixels_cur=10, distill_iters=3)
    H, W = int(g["H"]), int(g["W"])
    direction = torch.from_numpy(g["direction"])
    kfs = [{"pose": torch.from_numpy(g[f"frame{k}.c2w"]).clone()} for k in (0, 2)]
    dataset = types.SimpleNamespace(H=H, W=W, rays_d=direction)
    # HIP: teacher holds the fixture's initial map, the student starts from a perturbed copy
    teacher = model_from_golden(g, cfg, device, prefix="init.").eval()
    student = model_from_golden(g, cfg, device, prefix="init.").train()
    for lst in student.all_planes:
        for l in range(2):
            lst[l] = (lst[l] * 0.5).contiguous(memory_format=torch.channels_last)
    opt = slam_glue.create_optimizer(student, cfg)
    slam = types.SimpleNamespace(model=student, model_shared=teacher, map_optimizer=opt, device=torch.device(device),
                                 dataset=dataset, video=None,
                                 get_loss_from_ret=lambda ret, is_co_sdf=True: slam_glue.get_loss_from_ret(cfg, ret, is_co_sdf=is_co_sdf))
    mp = Mapper(cfg, slam, compute=compute)
    torch.manual_seed(5)
    mp.distillation(1, kfs, len(kfs))
    if compute == "fused":
        assert all(p.grad is None for lst in student.all_planes for p in lst) and bool(torch.isfinite(mp.last_losses["rgb_loss"]))
    # oracle: the same loop with the CPU restatement
    t_sc = oracle_scene_from_golden(g, cfg, prefix="init.")
    s_sc = oracle_scene_from_golden(g, cfg, prefix="init.")
    s_sc.all_planes = tuple([(p * 0.5).clone() for p in lst] for lst in s_sc.all_planes)
    s_sc.requires_grad_(True)
    o_opt = omap.OracleAdam(s_sc, cfg)
    torch.manual_seed(5)
    per = max(cfg["mapping"]["sample"] // len(kfs), cfg["mapping"]["min_pixels_cur"])
    for _ in range(cfg["mapping"]["distill_iters"]):
        ro, rd, tr, td = [], [], [], []
        for kf in kfs:
            pose = kf["pose"]
            idx = torch.randint(0, H * W, (per,))
            dcam = direction.reshape(-1, 3)[idx]
            o = pose[:3, 3].unsqueeze(0).repeat(per, 1)
            d = torch.sum(dcam[..., None, :] * pose[:3, :3], dim=-1)
            with torch.no_grad():
                t = t_sc.render_rays(o, d, target_d=None)
            ro.append(o); rd.append(d); tr.append(t["rgb"]); td.append(t["depth"].unsqueeze(-1))
        o_opt.zero_grad()
        ret = s_sc.forward(torch.cat(ro), torch.cat(rd), torch.cat(tr), torch.cat(td))
        omap.loss_from_ret(cfg, ret, is_co_sdf=False).backward()
        o_opt.step()
    k = 0
    for s, lst in enumerate(student.all_planes):
        for l in range(2):
            assert_close(lst[l].detach().cpu(), s_sc.all_planes[s][l].detach(), rtol=1e-3, atol=1e-4, what=f"plane {s}/{l}")
            k += 1
    for (n, p), q in zip(student.decoder.named_parameters(), [s_sc.col_w[0], s_sc.col_w[1], s_sc.sdf_w[0], s_sc.sdf_w[1]]):
        assert_close(p.detach().cpu(), q.detach(), rtol=1e-3, atol=1e-4, what=n)


def check_fused_vs_autograd(device, hidden=64, one_grid=False, co=True, seed=31, iters=3, plane_dtype="fp32"):
    """Decoder shapes without golden fixtures (2x64, the class defaults BASELINE.json quotes): the fused step must
    reach the parameters of the drop-in autograd path (itself pinned against the oracle by
    check_oracle_random_scene) from the same state with the same host-drawn batches."""
    g = load_golden("mapping3_colorplanes_cosdf" if not one_grid else "mapping3_onegrid_esdf")
    H, W, n_save = int(g["H"]), int(g["W"]), int(g["n_save"])
    direction = torch.from_numpy(g["direction"])
    frames = [dict(frame_id=k, c2w=torch.from_numpy(g[f"frame{k}.c2w"]), rgb=torch.from_numpy(g[f"frame{k}.rgb"]),
                   depth=torch.from_numpy(g[f"frame{k}.depth"]), direction=direction) for k in range(4)]
    finals = []
    for compute in ("autograd", "fused"):
        cfg = configs.small_test_config(one_grid=one_grid, is_co_sdf=co)
        cfg["decoder"]["hidden_dim"] = cfg["decoder"]["hidden_dim_color"] = hidden
        cfg["mapping"].update(sample=64, min_pixels_cur=10, iters=iters, n_pixels=0.25)
        # plane_dtype "fp16" (EXTENSION): the autograd path hands autograd fp16 gradients but the optimizer consumes the fp32
        # sums the render node leaves beside them (``grad32``) -- mean-reduced plane gradients are below fp16's range, a cast
        # would lose those updates (ADAM is scale-free) and the two paths would part
        cfg["grid"]["plane_dtype"] = plane_dtype
        torch.manual_seed(seed)
        m = JointEncoding(cfg, torch.from_numpy(g["bounding_box"]).to(device))
        m.device = torch.device(device)
        m = m.to(device).train()
        for lst in m.all_planes:
            for l in range(2):
                lst[l] = (lst[l] * 20.0).contiguous(memory_format=torch.channels_last)
        opt = slam_glue.create_optimizer(m, cfg)
        random.seed(seed)
        torch.manual_seed(seed)
        kfdb = KeyFrameDatabase(cfg, H, W, 8, n_save, device)
        for k in range(3):
            kfdb.add_keyframe(frames[k], k + 1)
        slam = types.SimpleNamespace(
            config=cfg, model=m, map_optimizer=opt, device=torch.device(device),
            dataset=types.SimpleNamespace(H=H, W=W), video=types.SimpleNamespace(keyframe=kfdb),
            get_loss_from_ret=lambda ret, cfg=cfg, **kw: slam_glue.get_loss_from_ret(cfg, ret, **kw),
            select_samples=slam_glue.select_samples)
        mapper = Mapper(cfg, slam, compute=compute, sampler="host")
        poses = torch.stack([f["c2w"] for f in frames]).to(device)
        random.seed(seed + 1)
        torch.manual_seed(seed + 1)
        start = [p.detach().cpu().clone() for lst in m.all_planes for p in lst]
        mapper.optimize_map(frames[3], poses)
        finals.append([p.detach().cpu().float().clone() for lst in m.all_planes for p in lst] +
                      [p.detach().cpu().clone() for p in m.decoder.parameters()])
        if plane_dtype == "fp16":
            assert all(p.dtype == torch.float16 for lst in m.all_planes for p in lst)
            moved = sum(int((a.float() != b.float()).sum()) for a, b in zip(start, finals[-1]))
            assert moved > 1000, "half-precision planes did not train"
    for k, (a, b) in enumerate(zip(*finals)):
        assert torch.isfinite(a).all()
        assert_close(b, a, rtol=1e-3, atol=1e-4 if plane_dtype == "fp32" else 3e-4, what=f"tensor {k}: fused vs autograd path")
    assert not torch.equal(finals[0][-1], torch.zeros_like(finals[0][-1]))


def out_contrib(fs):
    return int(fs.tape_rows.item())


# Post-Adam agreement bars.  eps = 1e-15 makes the plane / table groups' Adam scale-free: where a cell's gradient is fp32
# summation noise the step is +-lr whatever its size, so a few cells differ by O(lr) between two summation orders; all others
# agree to rounding.  MEASURED on MI355X over every fused-step case of the GPU suite and on the emulator cases
# (profiles/r04_adam_parity_stats.txt; MNE_PARITY_STATS=<file> appends each case's numbers): planes  mean |d| / lr <= 8.4e-7,
# cells off by more than 0.05 lr <= 4.2e-7 of a plane;  decoder  mean |d| / lr <= 3.8e-7, no weight off by more than 0.05 lr;
# hash table  see ADAM_BARS["table"].  The bars are those maxima times ~2.4, with an absolute floor of two cells for small tensors.
ADAM_BARS = {"plane": (2e-6, 1e-6, 2), "decoder": (1e-6, 0.0, 1), "table": (2e-6, 1e-6, 2)}


def adam_agreement(p_hip, p_ref, lr, kind, what):
    """(mean |d| / lr, fraction of elements off by more than 0.05 lr) of two post-Adam tensors, asserted against ADAM_BARS."""
    d = (p_hip.float() - p_ref).abs()
    mean_bar, frac_bar, floor = ADAM_BARS[kind]
    n_out = int((d > 0.05 * lr).sum())
    mean, allowed = float(d.mean()) / lr, max(floor, int(frac_bar * d.numel()))
    # an outlier is off by up to ~2 lr: the mean bar applies to the rest
    assert mean <= mean_bar + 2.0 * n_out / d.numel() and n_out <= allowed, \
        f"{what} after Adam: mean |d| / lr {mean:.3e} (bar {mean_bar:.1e}), {n_out} elements off by > 0.05 lr (allowed {allowed})"
    return mean, n_out / d.numel()


def _record_stats(what, d):
    path = os.environ.get("MNE_PARITY_STATS")
    if path:
        import json
        with open(path, "a") as f:
            f.write(json.dumps(dict(d, case=what, test=os.environ.get("PYTEST_CURRENT_TEST", ""))) + "\n")


def check_fused_step_vs_oracle(device, cfg, n_keyframes=4, seed=3, warm_steps=0, small=False, impl="grid_sample",
                               scatter="binned", poison_tape=True, oracle_chunk=None):
    """The BENCH path -- bench.Agent: device Feistel ray sampler, Philox jitter, FusedStep on two streams --
    against ONE oracle iteration on the SAME device-drawn batch: the batch (ray indices, rays, targets, z samples)
    is copied back from the device, the oracle (CPU autograd) evaluates forward, the seven losses, backward and
    Adam from the same parameters, and everything the iteration produces is compared:
      rays / targets   bit-exact re-assembly from the indices the device sampler drew
      z samples        inside the stratified intervals [lower, upper] the oracle computes from target depth
      rgb / depth      mean L1 < 1e-4 (north-star bar), elementwise rtol 1e-4
      7 losses + psnr  rtol 1e-4
      decoder grads    rtol 2e-3        plane grads (= exp_avg / (1 - beta1) after the first step) rtol 2e-3
      post-Adam parameters and moments.
    Works at any size (full office0: 38.4 M parameters, 2150 x 128 samples; ~2 s of oracle time)."""
    import bench
    dev = torch.device(device)
    ag = bench.Agent(cfg, dev, seed=seed, n_keyframes=n_keyframes, small=small, path="fused", scatter=scatter)
    fs, m = ag.fused, ag.model
    # EXTENSION (BASELINE configs[4]): grid.plane_dtype 'fp16' -- the planes are STORED in half precision; the oracle holds the
    # same values in fp32 tensors, computes everything in fp32 and rounds the parameters to fp16 after its Adam step
    half = cfg["grid"].get("plane_dtype", "fp32") == "fp16"
    assert all((p.dtype == torch.float16) == half for lst in m.all_planes for p in lst)
    for _ in range(warm_steps):
        ag.step()
    fs.synchronize()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    cpu = lambda t: t.detach().to("cpu", copy=True)
    planes0 = [[cpu(p).float().contiguous() for p in lst] for lst in m.all_planes]
    dec0 = {k: cpu(v) for k, v in m.decoder.state_dict().items()}
    opt_state0 = [{k: (cpu(v).contiguous() if torch.is_tensor(v) else v) for k, v in ag.opt._state(p).items()}
                  for lst in m.all_planes for p in lst]
    dec_params = list(m.decoder.parameters())
    dec_state0 = [{k: (cpu(v) if torch.is_tensor(v) else v) for k, v in ag.opt._state(p).items()} for p in dec_params]
    if poison_tape:
        # every tape row the weight-gradient pass, the plane update or the hash scatter reads must have been written by
        # THIS iteration: a row left over from an earlier one (or from a zero-filled allocation, which is what the host
        # emulator sees) would go unnoticed otherwise -- the failure mode of the round-2 layout-sensitive kernel
        # (DESIGN.md section 9).  NaN x 0 = NaN: one stale row poisons a whole gradient matrix.
        fs.tape.fill_(float("nan"))
        fs.partials.fill_(float("nan"))                  # likewise every partial weight gradient the decoder update sums
    ag.step()                                            # the iteration under test
    fs.synchronize()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    fs.check()
    R, S = fs.R, fs.S
    rays_o, rays_d, tgt_rgb, tgt_d = cpu(fs.rays_o), cpu(fs.rays_d), cpu(fs.tgt_rgb), cpu(fs.tgt_d)
    z = cpu(fs.z_vals)
    # ---- the batch: indices drawn on the device -> the reference's own assembly on the CPU
    idx = cpu(fs.idx)
    n = cfg["mapping"]["sample"]
    kf, cur, poses = cpu(ag.kf_rays), cpu(ag.cur_rays), cpu(ag.poses)
    ig, ic = idx[:n], idx[n:]
    assert ig.unique().numel() == n and ic.unique().numel() == ic.numel(), "sampling without replacement violated"
    rays7 = torch.cat([kf[ig], cur[ic]], 0)
    ids = torch.cat([torch.div(ig, ag.n_save, rounding_mode="trunc"), -torch.ones(ic.numel(), dtype=torch.int64)])
    ref_o, ref_d, ref_rgb, ref_dep = omap.assemble_rays(rays7, ids, poses)
    assert_close(rays_o, ref_o, rtol=0, atol=0, what="rays_o (bit-exact)")
    assert_close(rays_d, ref_d, rtol=0, atol=0, what="rays_d (bit-exact)")
    assert_close(tgt_rgb, ref_rgb, rtol=0, atol=0, what="target rgb")
    assert_close(tgt_d, ref_dep[:, 0], rtol=0, atol=0, what="target depth")
    # ---- oracle scene with the pre-step parameters
    sc = OracleScene(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float64), build=False)
    sc.all_planes = tuple([p.contiguous() for p in lst] for lst in planes0)
    sc.col_w = [dec0["color_net.model.0.weight"], dec0["color_net.model.2.weight"]]
    sc.sdf_w = [dec0["sdf_net.model.0.weight"], dec0["sdf_net.model.2.weight"]]
    sc.requires_grad_(True)
    lo = sc.sample_z(R, tgt_d[:, None], u=torch.zeros(R, S))
    hi = sc.sample_z(R, tgt_d[:, None], u=torch.ones(R, S))
    assert bool(((z >= lo) & (z <= hi)).all()), "device z samples leave their stratified intervals"
    assert float((z - lo).abs().max()) > 0, "no jitter was applied"
    opt = omap.OracleAdam(sc, cfg)
    for g_, states in zip(opt.groups, [dec_state0, opt_state0[:6], opt_state0[6:]]):
        if states and states[0].get("step", 0):
            g_.t = int(states[0]["step"])
            g_.m = [st["exp_avg"].clone() for st in states]
            g_.v = [st["exp_avg_sq"].clone() for st in states]
    if oracle_chunk:         # batches whose autograd graph does not fit in host memory (INS Indoor: 2048 x 1045): chunks of rays
        ret = omap.forward_backward_chunked(sc, cfg, rays_o, rays_d, tgt_rgb, tgt_d[:, None], z, impl=impl, chunk=oracle_chunk)
    else:
        ret = sc.forward(rays_o, rays_d, tgt_rgb, tgt_d[:, None], impl=impl, z_vals=z)
        loss = omap.loss_from_ret(cfg, ret, is_co_sdf=cfg["is_co_sdf"])
        loss.backward()
    # ---- forward
    rgb, depth = cpu(fs.rgb), cpu(fs.depth)
    assert float((rgb - ret["rgb"].detach()).abs().mean()) < 1e-4 and float((depth - ret["depth"].detach()).abs().mean()) < 1e-4
    assert_close(rgb, ret["rgb"].detach(), rtol=1e-4, atol=2e-5, what="rgb")
    assert_close(depth, ret["depth"].detach(), rtol=1e-4, atol=2e-5, what="depth")
    # raw is scratch under early ray termination: defined for the tiles the backward walked (ray_tiles), which hold
    # every sample that influences the maps or the losses
    known = (torch.arange(S)[None, :] < cpu(fs.ray_tiles[:R]).long()[:, None] * 32)
    assert int(known.sum()) >= out_contrib(fs)
    assert_close(cpu(fs.raw)[known], ret["raw"].detach()[known], rtol=1e-4, atol=2e-5, what="raw (decoded samples)")
    L = cpu(fs.losses)
    for k, key in enumerate(LOSS_KEYS):
        assert_close(L[k], ret[key].detach().reshape(()), rtol=1e-4, atol=1e-7, what=key)
    # ---- gradients
    n0, n1, n2 = [w.numel() for w in (sc.col_w[0], sc.col_w[1], sc.sdf_w[0])]
    dg = cpu(fs.dec_grad)
    got = [dg[:n0], dg[n0:n0 + n1], dg[n0 + n1:n0 + n1 + n2], dg[n0 + n1 + n2:]]
    for gk, w, nm in zip(got, sc.decoder_list(), DEC_KEYS):
        ref = w.grad
        if scatter != "atomics":               # (that schedule's one Adam launch zeroes every gradient buffer it consumed)
            assert_close(gk.reshape(ref.shape), ref, rtol=2e-3, atol=2e-5 * max(1.0, float(ref.abs().max())), what=f"decoder grad {nm}")
    flat_planes = [p for lst in m.all_planes for p in lst]
    first_step = not (opt_state0[0].get("step", 0))
    if first_step:
        b1 = ag.opt.param_groups[1]["betas"][0]
        for k, (p, ref_p) in enumerate(zip(flat_planes, sc.plane_list())):
            g_hip = cpu(ag.opt._state(p)["exp_avg"]) / (1.0 - b1)           # m1 = (1 - b1) g  after the first step
            ref = ref_p.grad
            assert_close(g_hip, ref, rtol=2e-3, atol=2e-5 * max(1e-6, float(ref.abs().max())), what=f"plane grad {k}")
    # ---- Adam
    opt.step()
    if half:                                   # p16' = round_to_nearest_even(Adam(float(p16), g, m, v)): no fp32 master anywhere
        with torch.no_grad():
            for ref_p in sc.plane_list():
                ref_p.copy_(ref_p.half().float())
    stats = {"plane_mean_over_lr": 0.0, "plane_outliers": 0.0, "dec_mean_over_lr": 0.0, "dec_outliers": 0.0}
    for k, (p, ref_p, g_m) in enumerate(zip(flat_planes, sc.plane_list(), opt.groups[1].m + (opt.groups[2].m if len(opt.groups) > 2 else []))):
        st = ag.opt._state(p)
        assert_close(cpu(st["exp_avg"]), g_m, rtol=2e-3, atol=2e-6 * max(1e-6, float(g_m.abs().max())), what=f"exp_avg {k}")
        mean, frac = adam_agreement(cpu(p), ref_p.detach(), ag.opt.param_groups[1]["lr"], "plane", f"plane {k}")
        stats["plane_mean_over_lr"], stats["plane_outliers"] = max(stats["plane_mean_over_lr"], mean), max(stats["plane_outliers"], frac)
    for w_hip, w_ref, nm in zip(dec_params, sc.decoder_list(), DEC_KEYS):
        mean, frac = adam_agreement(cpu(w_hip), w_ref.detach(), ag.opt.param_groups[0]["lr"], "decoder", f"decoder {nm}")
        stats["dec_mean_over_lr"], stats["dec_outliers"] = max(stats["dec_mean_over_lr"], mean), max(stats["dec_outliers"], frac)
    # tiles the plane update skipped (never received a gradient: m = v = 0, Adam is the identity there): parameters bit-equal to
    # what they were, moments still zero -- and the oracle's DENSE Adam agrees (the comparison above covered them)
    skipped = 0
    if getattr(fs, "tile_live", None) is not None:
        live, off = cpu(fs.tile_live), 0
        for p, p0 in zip(flat_planes, [q for lst in planes0 for q in lst]):
            ty, tx = (p.shape[2] + 15) // 16, (p.shape[3] + 15) // 16
            dead = (live[off:off + ty * tx] == 0).reshape(ty, tx).repeat_interleave(16, 0).repeat_interleave(16, 1)[:p.shape[2], :p.shape[3]]
            off += ty * tx
            if bool(dead.any()):
                st = ag.opt._state(p)
                assert torch.equal(cpu(p).float()[0][:, dead], p0[0][:, dead]), "a skipped tile's parameters changed"
                assert float(cpu(st["exp_avg"])[0][:, dead].abs().max()) == 0.0 and float(cpu(st["exp_avg_sq"])[0][:, dead].abs().max()) == 0.0
                skipped += int(dead.sum()) * p.shape[1]
    stats["skipped_params"] = skipped
    _record_stats("fused_step_vs_oracle", dict(stats, R=R, S=S, half=half, scatter=scatter, warm=warm_steps,
                                               planes=sum(p.numel() for p in flat_planes)))
    n_defer = -1
    if fs.bins is not None:                # (the deferred-list length of the call sits in the render workspace, behind the lists)
        import struct
        a16 = lambda x: (x + 15) & ~15
        off_cnt = a16(R * S * 16) + 4 * a16(R * 4) + a16(R * 32)      # masks | deferred, decoded-tile, long, heavy lists | heavy records
        n_defer = struct.unpack("i", bytes(cpu(fs.ws)[off_cnt:off_cnt + 4].tolist()))[0]
    return {"R": R, "S": S, "contributing": int(fs.tape_rows.item()), "adam_stats": stats, "deferred_rays": n_defer,
            "rgb_l1": float((rgb - ret["rgb"].detach()).abs().mean()), "depth_l1": float((depth - ret["depth"].detach()).abs().mean())}


def hash_test_config(hash_size=12, hidden=32, desired_resolution=128):
    """A reduced hash-grid workload: small table (collisions on the fine levels), office0 bound, 11 + 32 samples."""
    cfg = configs.bench_office0_hash(hidden=hidden, hash_size=hash_size, desired_resolution=desired_resolution)
    cfg["training"]["n_range_d"], cfg["training"]["n_samples_d"] = 11, 32
    cfg["mapping"]["sample"] = 256
    return cfg


def check_hash_update_bit_reproducible(device, cfg, n_keyframes=3, seed=5, warm_steps=2, small=True, repeats=2):
    """The hash-grid table update sums 64-bit fixed-point addends with LDS / global INTEGER atomics: whatever order the rows
    arrive in, the sums -- and therefore the table and its moments after Adam -- are the same bits.  Checked the way the
    domain offers it: the update of a real iteration (run inside the pipeline, beside the weight-gradient kernel) is
    repeated stand-alone on copies of the pre-step state, each time on a fresh workspace; every result must equal the
    pipeline's bit for bit."""
    import ctypes as C
    import bench
    from mneslam_amd import _lib
    dev = torch.device(device)
    ag = bench.Agent(cfg, dev, seed=seed, n_keyframes=n_keyframes, small=small, path="fused")
    fs, m = ag.fused, ag.model
    for _ in range(warm_steps):
        ag.step()
    sync = (lambda: torch.cuda.synchronize()) if dev.type == "cuda" else (lambda: None)
    sync()
    stt = ag.opt._state(m.embed_fn.params)
    table0, m0, v0, t0 = m.embed_fn.params.detach().clone(), stt["exp_avg"].clone(), stt["exp_avg_sq"].clone(), int(stt["step"])
    ag.step()
    sync()
    assert int(stt["step"]) == t0 + 1
    want = (m.embed_fn.params.detach().clone(), stt["exp_avg"].clone(), stt["exp_avg_sq"].clone())
    assert not torch.equal(want[0], table0), "the step did not move the table"
    lib, P = fs.lib, _lib.ptr
    R, S, tape = fs.n_active, fs.S, fs.tape
    for k in range(repeats):
        tb, mm, vv = table0.clone(), m0.clone(), v0.clone()
        ws = torch.zeros(fs.hash_ws_bytes, device=dev, dtype=torch.uint8)
        o = _lib.PlaneOpt()
        for f in ("lr", "beta1", "beta2", "eps", "weight_decay"):
            setattr(o, f, getattr(fs.table_opt, f))
        o.m, o.v, o.step = mm.data_ptr(), vv.data_ptr(), t0 + 1
        _lib.check(lib.mne_hash_slice_adam(C.byref(fs.grid_cfg), C.byref(fs.scene), R, S, P(fs.rays_o), P(fs.rays_d), P(fs.z_vals),
                                           P(tape), P(fs.ray_tiles), P(tb), C.byref(o), P(ws), fs.hash_ws_bytes, None,
                                           _lib.stream_for(fs.rays_o)), "mne_hash_slice_adam")
        sync()
        for name, a, b in zip(("table", "exp_avg", "exp_avg_sq"), (tb, mm, vv), want):
            assert torch.equal(a, b), f"repeat {k}: {name} differs from the pipeline's result in {int((a != b).sum())} entries"
    return {"moved": int((want[0] != table0).sum()), "R": R, "S": S}


def check_hash_large_batch(device, cfg, n_keyframes=4, seed=7, steps=2):
    """A batch of more than 512 K tape rows (HASH_MAX_CHUNKS chunks of 1024: the slice kernel then walks its records in
    several chunk groups; round 5 -- INS Indoor's 2150 x 1045 or an 8192-ray batch did not run on the hash encoding before).
    Size-independent check: the table update by slice-binned exact LDS sums against the other implementation of the same
    update -- run-reduced global float atomics into a gradient buffer + the streaming Adam kernel (MNE_HASH_UPDATE=atomics) --
    on the same device-drawn batches from the same state."""
    import bench
    dev = torch.device(device)
    tables = []
    for update in ("slices", "atomics"):
        old = os.environ.get("MNE_HASH_UPDATE")
        os.environ["MNE_HASH_UPDATE"] = update
        try:
            ag = bench.Agent(cfg, dev, seed=seed, n_keyframes=n_keyframes, path="fused")
        finally:
            if old is None:
                os.environ.pop("MNE_HASH_UPDATE", None)
            else:
                os.environ["MNE_HASH_UPDATE"] = old
        assert ag.fused.table_update == update
        rows = ag.fused.R * ag.fused.S
        for _ in range(steps):
            ag.step()
        ag.fused.synchronize()
        torch.cuda.synchronize() if dev.type == "cuda" else None
        tables.append((ag.model.embed_fn.params.detach().cpu().clone(), ag.opt.param_groups, float(ag.fused.losses[7])))
        del ag
        if dev.type == "cuda":
            torch.cuda.empty_cache()
    (t_s, groups, psnr_s), (t_a, _, psnr_a) = tables
    assert torch.isfinite(t_s).all() and abs(psnr_s - psnr_a) < 1e-3 * abs(psnr_a) + 1e-4
    lr = next(g["lr"] for g in groups if any(p.numel() == t_s.numel() for p in g["params"]))
    mean, frac = adam_agreement(t_s, t_a, lr, "table", "slice update vs atomics update")
    return {"rows": rows, "mean_over_lr": mean, "outliers": frac}


def check_hash_fused_step_vs_oracle(device, cfg, n_keyframes=3, seed=5, warm_steps=0, small=True):
    """HashFusedStep (hash-grid wiring, EXTENSION, parity unpinned) against one iteration of the build's own CPU oracle
    (oracle.scene_rep.OracleHashScene + oracle.hashgrid) on the same device-drawn batch and parameters:
      table indices     uint32, bit-exact (HIP grid kernel vs oracle/hashgrid.py) on the batch's sample positions
      rgb / depth       mean L1 < 1e-4, elementwise rtol 1e-4;   raw on every sample;   7 losses + psnr rtol 1e-4
      decoder grads     rtol 2e-3;   table grad (= exp_avg / (1 - beta1) after the first step) rtol 2e-3
      post-Adam table and decoder."""
    import bench
    from oracle.scene_rep import OracleHashScene
    dev = torch.device(device)
    ag = bench.Agent(cfg, dev, seed=seed, n_keyframes=n_keyframes, small=small, path="fused")
    fs, m = ag.fused, ag.model
    for _ in range(warm_steps):
        ag.step()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    cpu = lambda t: t.detach().to("cpu", copy=True)
    table0 = cpu(m.embed_fn.params)
    dec0 = {k: cpu(v) for k, v in m.decoder.state_dict().items()}
    dec_params = list(m.decoder.parameters())
    st_table0 = {k: (cpu(v) if torch.is_tensor(v) else v) for k, v in ag.opt._state(m.embed_fn.params).items()}
    dec_state0 = [{k: (cpu(v) if torch.is_tensor(v) else v) for k, v in ag.opt._state(p).items()} for p in dec_params]
    ag.step()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    R, S = fs.R, fs.S
    rays_o, rays_d, tgt_rgb, tgt_d, z = cpu(fs.rays_o), cpu(fs.rays_d), cpu(fs.tgt_rgb), cpu(fs.tgt_d), cpu(fs.z_vals)
    gc = m.embed_fn.cfg
    scales, ress, sizes, offsets = m.embed_fn.level_table()
    grid = dict(n_levels=gc.n_levels, n_features=gc.n_features, base_resolution=gc.base_resolution,
                per_level_scale=gc.per_level_scale, log2_hashmap_size=gc.log2_hashmap_size)
    bb = torch.tensor(cfg["mapping"]["bound"], dtype=torch.float64)
    sc = OracleHashScene(cfg, bb, table0.clone(), grid, scales=scales)
    sc.col_w = [dec0["color_net.model.0.weight"], dec0["color_net.model.2.weight"]]
    sc.sdf_w = [dec0["sdf_net.model.0.weight"], dec0["sdf_net.model.2.weight"]]
    assert float(sc.sdf_w[0][:, gc.n_levels * gc.n_features:64].abs().max()) == 0.0, "dead feature columns must stay zero"
    sc.requires_grad_(True)
    # ---- integer table indices on this batch's positions: HIP stand-alone kernel vs oracle, bit-exact
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]).reshape(-1, 3)
    u = ((pts - bb[:, 0]) / (bb[:, 1] - bb[:, 0])).float()
    sub = torch.randperm(u.shape[0], generator=torch.Generator().manual_seed(1))[:4096]
    idx_hip = m.embed_fn.indices(u[sub].to(dev)).cpu()
    from oracle import hashgrid
    for l in range(gc.n_levels):
        idx_ref, _ = hashgrid.grid_indices(u[sub], scales[l], ress[l], sizes[l])
        assert torch.equal(idx_hip[:, l], idx_ref), f"hash indices differ at level {l}"
    opt = omap.OracleAdam(sc, cfg)
    for g_, states in zip(opt.groups, [dec_state0, [st_table0]]):
        if states and states[0].get("step", 0):
            g_.t = int(states[0]["step"])
            g_.m = [st["exp_avg"].clone() for st in states]
            g_.v = [st["exp_avg_sq"].clone() for st in states]
    ret = sc.forward(rays_o, rays_d, tgt_rgb, tgt_d[:, None], z_vals=z)
    omap.loss_from_ret(cfg, ret, is_co_sdf=cfg["is_co_sdf"]).backward()
    rgb, depth = cpu(fs.rgb), cpu(fs.depth)
    assert float((rgb - ret["rgb"].detach()).abs().mean()) < 1e-4 and float((depth - ret["depth"].detach()).abs().mean()) < 1e-4
    assert_close(rgb, ret["rgb"].detach(), rtol=1e-4, atol=2e-5, what="rgb")
    assert_close(depth, ret["depth"].detach(), rtol=1e-4, atol=2e-5, what="depth")
    known = (torch.arange(S)[None, :] < cpu(fs.ray_tiles[:R]).long()[:, None] * 32) if fs.early_termination else torch.ones(R, S, dtype=torch.bool)
    assert_close(cpu(fs.raw)[known], ret["raw"].detach()[known], rtol=1e-4, atol=2e-5, what="raw (decoded samples)")
    L = cpu(fs.losses)
    for k, key in enumerate(LOSS_KEYS):
        assert_close(L[k], ret[key].detach().reshape(()), rtol=1e-4, atol=1e-7, what=key)
    # the single Adam launch of this path zeroes every gradient accumulator (decoder included): on a first step the
    # gradients are read back from the first moments, m1 = (1 - b1) (g + wd p)
    if not st_table0.get("step", 0):
        grp = ag.opt.param_groups[0]
        for w_hip, w0, w, nm in zip(dec_params, [dec0[k] for k in DEC_KEYS], sc.decoder_list(), DEC_KEYS):
            ref = w.grad + grp["weight_decay"] * w0.detach()
            got = cpu(ag.opt._state(w_hip)["exp_avg"]) / (1.0 - grp["betas"][0])
            assert_close(got, ref, rtol=2e-3, atol=2e-5 * max(1.0, float(ref.abs().max())), what=f"decoder grad {nm}")
    st_t = ag.opt._state(m.embed_fn.params)
    ref = sc.table.grad
    assert float(ref.abs().max()) > 0
    if not st_table0.get("step", 0):
        b1 = ag.opt.param_groups[1]["betas"][0]
        g_hip = cpu(st_t["exp_avg"]) / (1.0 - b1)
        assert_close(g_hip, ref, rtol=2e-3, atol=2e-5 * float(ref.abs().max()), what="table grad")
    assert float(cpu(fs.table_grad).abs().max()) == 0.0, "the Adam kernel leaves the gradient accumulator zeroed"
    opt.step()
    assert_close(cpu(st_t["exp_avg"]), opt.groups[1].m[0], rtol=2e-3, atol=2e-6 * float(opt.groups[1].m[0].abs().max()), what="table exp_avg")
    stats = {}
    stats["table_mean_over_lr"], stats["table_outliers"] = adam_agreement(cpu(m.embed_fn.params), sc.table.detach(),
                                                                          ag.opt.param_groups[1]["lr"], "table", "table")
    touched = ref != 0
    assert float(touched.float().mean()) > 0.001
    dm = do = 0.0
    for w_hip, w_ref, nm in zip(dec_params, sc.decoder_list(), DEC_KEYS):
        mean, frac = adam_agreement(cpu(w_hip), w_ref.detach(), ag.opt.param_groups[0]["lr"], "decoder", f"decoder {nm}")
        dm, do = max(dm, mean), max(do, frac)
    _record_stats("hash_fused_step_vs_oracle", dict(stats, dec_mean_over_lr=dm, dec_outliers=do, R=R, S=S, table=int(ref.numel())))
    return {"R": R, "S": S, "touched_entries": int(touched.sum()), "rgb_l1": float((rgb - ret["rgb"].detach()).abs().mean())}


def check_device_clock(device):
    """mne_clock_t: iteration / Adam step read from device memory (graph replay) give bit-identical results to the same
    values passed as arguments -- ray sampling keys, the jitter counter offset, the bias corrections of both Adam kernels."""
    import ctypes as C
    from mneslam_amd import _lib
    lib, P = _lib.load(), _lib.ptr
    dev = torch.device(device)
    gen = torch.Generator().manual_seed(8)
    n_kf, n_save, HW, n_g, n_c = 4, 200, 500, 128, 32
    kf = torch.randn(n_kf * n_save, 7, generator=gen).to(dev)
    cur = torch.randn(HW, 7, generator=gen).to(dev)
    poses = torch.randn(n_kf + 1, 4, 4, generator=gen).to(dev)
    R = n_g + n_c
    b1, b2, n_table = 0.9, 0.99, 64
    table = torch.tensor([(1.0 - b1 ** k, 1.0 - b2 ** k) for k in range(1, n_table + 1)], dtype=torch.float64, device=dev)
    clk_iter = torch.zeros(1, dtype=torch.int64, device=dev)
    clk_step = torch.zeros(1, dtype=torch.int32, device=dev)
    ck = _lib.Clock()
    ck.iteration, ck.step_offset, ck.bias_table = clk_iter.data_ptr(), clk_step.data_ptr(), table.data_ptr()
    ck.n_table, ck.beta1, ck.beta2 = n_table, b1, b2
    cfg = configs.small_test_config()
    from mneslam_amd import hip_path
    rc = hip_path.render_cfg_struct(cfg)
    S = lib.mne_num_samples(C.byref(rc), 1)
    ck.z_offset_stride = (R * S + 3) // 4
    tables = hip_path.linspace_tables(cfg, True, dev)
    st = _lib.stream_for(kf)

    def batch(iteration, clock):
        o = [torch.empty(R, 3, device=dev), torch.empty(R, 3, device=dev), torch.empty(R, 3, device=dev), torch.empty(R, device=dev)]
        idx = torch.empty(R, dtype=torch.int64, device=dev)
        z = torch.empty(R, S, device=dev)
        cnt, rcnt = torch.empty(8, dtype=torch.int32, device=dev), torch.empty(R, 8, dtype=torch.int32, device=dev)
        _lib.check(lib.mne_sample_rays(P(kf), n_kf * n_save, n_save, None, P(cur), HW, P(poses), n_kf + 1, n_g, n_c, None, None,
                                       77, iteration, P(o[0]), P(o[1]), P(o[2]), P(o[3]), P(idx),
                                       C.byref(clock) if clock else None, st))
        d = o[3].abs() + 0.5
        _lib.check(lib.mne_sample_z(C.byref(rc), R, P(d), None, P(tables), 77, iteration * ck.z_offset_stride if not clock else 0,
                                    P(z), P(cnt), P(rcnt), C.byref(clock) if clock else None, st))
        return idx.cpu(), z.cpu()

    for it in (0, 3, 11):
        ref = batch(it, None)
        clk_iter.fill_(it)
        got = batch(0, ck)
        assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]), f"iteration {it}: clock and argument disagree"
    assert lib.mne_clock_advance(P(clk_iter), P(clk_step), st) == 0
    assert int(clk_iter.item()) == 12 and int(clk_step.item()) == 1
    # Adam: step t as argument vs step 1 + device offset t - 1
    for t in (1, 2, 17):
        res = []
        for use_clock in (False, True):
            g0 = torch.Generator().manual_seed(t)
            p = torch.randn(1000, generator=g0).to(dev); g = torch.randn(1000, generator=g0).to(dev)
            m = (0.1 * torch.randn(1000, generator=g0)).to(dev); v = (0.01 * torch.rand(1000, generator=g0)).to(dev)
            seg = _lib.AdamSeg()
            seg.p, seg.g, seg.m, seg.v, seg.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), 1000
            seg.lr, seg.beta1, seg.beta2, seg.eps, seg.weight_decay = 0.005, b1, b2, 1e-15, 1e-6
            seg.step = 1 if use_clock else t
            clk_step.fill_(t - 1)
            arr = (_lib.AdamSeg * 1)(seg)
            _lib.check(lib.mne_adam_step(arr, 1, 0, C.byref(ck) if use_clock else None, st))
            res.append((p.cpu(), m.cpu(), v.cpu()))
        for a, b in zip(*res):
            assert torch.equal(a, b), f"Adam step {t}: clock and argument disagree"
    bad = _lib.Clock(); bad.iteration = clk_iter.data_ptr(); bad.step_offset = clk_step.data_ptr(); bad.bias_table = table.data_ptr()
    bad.n_table, bad.beta1, bad.beta2 = n_table, 0.8, b2
    assert lib.mne_adam_step(arr, 1, 0, C.byref(bad), st) < 0 and b"betas" in lib.mne_last_error()


def U5f(ro, cfg, device):
    return torch.rand(ro.shape[0], cfg["training"]["n_samples"], generator=torch.Generator().manual_seed(8)).to(device)


def check_render_maps_fast_path(device):
    """N1: the no-grad fast path (one launch sequence, exact early ray termination) gives the maps of render_rays --
    without depth guidance against the reference's golden render, with depth guidance against the full render of the
    same model; render_img walks a whole frame through it (chunked and unchunked runs must agree bit for bit)."""
    g = load_golden("render_nodepth")
    cfg = configs.small_test_config()
    m = model_from_golden(g, cfg, device).eval()
    rays_o, rays_d, rgb, d, U = to_dev(fixture_inputs(g), device)
    out = m.render_maps(rays_o, rays_d, target_d=None, u=U)
    assert_close(out["rgb"].cpu(), g["rr.rgb"], rtol=1e-4, atol=1e-5, what="rgb (early termination)")
    assert_close(out["depth"].cpu(), g["rr.depth"], rtol=1e-4, atol=1e-5, what="depth (early termination)")
    assert_close(out["acc_map"].cpu(), g["rr.acc_map"], rtol=1e-4, atol=1e-5, what="acc")
    assert_close(out["depth_var"].cpu(), g["rr.depth_var"], rtol=1e-3, atol=1e-5, what="depth_var")
    assert_close(out["disp_map"].cpu(), g["rr.disp_map"], rtol=1e-4, atol=1e-5, what="disp")
    # depth-guided: fast path == full render of the same model, bit for bit (same kernels, fewer samples decoded)
    g2 = load_golden("fwd_onegrid")
    m2 = model_from_golden(g2, cfg, device).eval()
    ro, rd, _, dd, U2 = to_dev(fixture_inputs(g2), device)
    full = m2._render(ro, rd, None, dd, u=U2)
    fast = m2.render_maps(ro, rd, target_d=dd, u=U2)
    assert torch.equal(full[0].detach(), fast["rgb"]) and torch.equal(full[1].detach(), fast["depth"])
    assert torch.equal(full[3].detach(), fast["acc_map"])
    # colour planes: the forward-only kernels keep ONE set of feature rows in LDS and gather the two plane sets in turn
    # (decode_tile<..., SEQF>) -- same bits as the training-form decode behind _render, with and without depth guidance
    g4 = load_golden("fwd_colorplanes")
    cfg4 = configs.small_test_config(**FWD_CASES["fwd_colorplanes"])
    m4 = model_from_golden(g4, cfg4, device).eval()
    ro4, rd4, _, dd4, U4 = to_dev(fixture_inputs(g4), device)
    full4 = m4._render(ro4, rd4, None, dd4, u=U4)
    fast4 = m4.render_maps(ro4, rd4, target_d=dd4, u=U4)
    assert torch.equal(full4[0].detach(), fast4["rgb"]) and torch.equal(full4[1].detach(), fast4["depth"])
    assert torch.equal(full4[3].detach(), fast4["acc_map"])
    S_free = cfg4["training"]["n_samples"]
    U5 = torch.rand(ro4.shape[0], S_free, generator=torch.Generator().manual_seed(8)).to(device)
    full5 = m4._render(ro4, rd4, None, None, u=U5)
    fast5 = m4.render_maps(ro4, rd4, target_d=None, u=U5)
    assert torch.equal(full5[0].detach(), fast5["rgb"]) and torch.equal(full5[1].detach(), fast5["depth"])
    # the frame decode kernel (decode_frame_kernel: one plane set, 2x32 decoders, whole frames only) forced onto
    # these small batches: same bits as decode_kernel, with and without depth guidance, fp32 and half-precision planes
    import os
    for f16 in (False, True):
        m6 = model_from_golden(g2, cfg, device).eval()
        if f16:
            for lst in m6.all_planes:
                for k in range(len(lst)):
                    lst[k] = lst[k].half()
        outs = {}
        for mode in ("1000000", "-1"):
            os.environ["MNE_FRAME_MIN_TILES"] = mode
            try:
                outs[mode] = (m6.render_maps(ro, rd, target_d=dd, u=U2), m6.render_maps(ro, rd, target_d=None, u=U5f(ro, cfg, device)))
            finally:
                del os.environ["MNE_FRAME_MIN_TILES"]
        for a_, b_ in zip(outs["1000000"], outs["-1"]):
            for k in ("rgb", "depth", "acc_map", "depth_var", "disp_map"):
                assert torch.equal(a_[k], b_[k]), f"frame decode kernel differs ({k}, f16={f16})"
    # ... and on rays of several tiles whose sample count is no multiple of 32 (150 free samples: five tiles, the on-demand kernel
    # walks up to four of them per ray; the raw rows of a tile then share cache lines with the next tile's)
    cfg7 = configs.small_test_config()
    cfg7["training"]["n_samples"] = 150
    m7 = model_from_golden(g2, cfg7, device).eval()
    U7 = torch.rand(ro.shape[0], 150, generator=torch.Generator().manual_seed(9)).to(device)
    outs7 = {}
    for mode in ("1000000", "-1"):
        os.environ["MNE_FRAME_MIN_TILES"] = mode
        try:
            outs7[mode] = m7.render_maps(ro, rd, target_d=None, u=U7)
        finally:
            del os.environ["MNE_FRAME_MIN_TILES"]
    full7 = m7._render(ro, rd, None, None, u=U7)
    for k in ("rgb", "depth", "acc_map", "depth_var", "disp_map"):
        assert torch.equal(outs7["1000000"][k], outs7["-1"][k]), f"frame kernels differ on 150-sample rays ({k})"
    assert torch.equal(full7[0].detach(), outs7["-1"]["rgb"]) and torch.equal(full7[1].detach(), outs7["-1"]["depth"])
    # whole frame through render_img: chunked like the reference vs one launch sequence
    cfg3 = configs.small_test_config()
    cfg3["cam"].update(H=12, W=16, fx=16.0, fy=16.0, cx=8.0, cy=6.0, crop_edge=0)
    m3 = model_from_golden(g, cfg3, device).eval()
    c2w = torch.eye(4)
    c2w[:3, 3] = torch.tensor([0.1, -0.1, 0.0])
    m3.ray_batch_size = 50
    torch.manual_seed(3)
    d_a, c_a = m3.render_img(c2w.to(device), device, gt_depth=None)
    m3.render_chunk_rays = 64
    torch.manual_seed(3)
    d_b, c_b = m3.render_img(c2w.to(device), device, gt_depth=None)
    assert d_a.dtype == torch.float64 and d_a.shape == (12, 16) and c_a.shape == (12, 16, 3)
    assert torch.equal(d_a, d_b) and torch.equal(c_a, c_b)
    # production form (device jitter): the frame's rays are walked along a Z-order curve (4 x 2 pixel patches per workgroup);
    # per-pixel results do not depend on the order -- without jitter the image is the same bit for bit, chunked or not
    cfg5 = configs.small_test_config()
    cfg5["cam"].update(H=13, W=18, fx=16.0, fy=16.0, cx=9.0, cy=6.5, crop_edge=0)
    cfg5["training"]["perturb"] = 0
    m5 = model_from_golden(g, cfg5, device).eval()
    m5.jitter_rng = "device"
    gt5 = (1.0 + 0.05 * torch.arange(13 * 18, dtype=torch.float32).reshape(13, 18) % 1.7).to(device)
    for gtd in (None, gt5):
        d_on, c_on = m5.render_img(c2w.to(device), device, gt_depth=gtd)
        m5.render_chunk_rays = 100
        d_ch, c_ch = m5.render_img(c2w.to(device), device, gt_depth=gtd)
        m5.render_patch_order = False
        d_off, c_off = m5.render_img(c2w.to(device), device, gt_depth=gtd)
        m5.render_patch_order, m5.render_chunk_rays = True, 1 << 20
        assert torch.equal(d_on, d_off) and torch.equal(c_on, c_off) and torch.equal(d_on, d_ch) and torch.equal(c_on, c_ch)
    order = m5._pixel_order(13, 18, "cpu")
    assert sorted(order.tolist()) == list(range(13 * 18))
    assert sorted(order[:8].tolist()) == [0, 1, 2, 3, 18, 19, 20, 21]          # a 4 x 2 patch
    # and equal to the reference's own chunk loop over render_rays (same CPU jitter draws, same order)
    from mneslam_amd.model.utils import get_rays
    ro3, rd3 = get_rays(12, 16, 16.0, 16.0, 8.0, 6.0, c2w.to(device), device)
    ro3, rd3 = ro3.reshape(-1, 3), rd3.reshape(-1, 3)
    torch.manual_seed(3)
    ref_d = torch.cat([m3.render_rays(ro3[i:i + 50], rd3[i:i + 50], target_d=None)["depth"] for i in range(0, 192, 50)])
    assert_close(d_a.reshape(-1).float().cpu(), ref_d.detach().cpu(), rtol=1e-5, atol=1e-6, what="render_img vs chunked render_rays")


def quality_trajectory(device, cfg, n_iters=40, n_keyframes=4, seed=5, small=True):
    """Matched-quality evidence: the fused path and the oracle start from the same parameters and are
    trained on IDENTICAL batches -- every iteration the device draws the rays and the jittered z samples, the oracle
    (CPU autograd + written-out Adam) is fed the very same rays / targets / samples.  Returns per-iteration PSNR and
    depth-L1 (mean |depth - d| over rays with 0 < d < depth_trunc) of both."""
    import bench
    dev = torch.device(device)
    ag = bench.Agent(cfg, dev, seed=seed, n_keyframes=n_keyframes, small=small, path="fused", scatter="binned")
    fs, m = ag.fused, ag.model
    cpu = lambda t: t.detach().to("cpu", copy=True)
    sc = OracleScene(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float64), build=False)
    sc.all_planes = tuple([cpu(p).contiguous() for p in lst] for lst in m.all_planes)
    sd = {k: cpu(v) for k, v in m.decoder.state_dict().items()}
    sc.col_w = [sd["color_net.model.0.weight"], sd["color_net.model.2.weight"]]
    sc.sdf_w = [sd["sdf_net.model.0.weight"], sd["sdf_net.model.2.weight"]]
    sc.requires_grad_(True)
    opt = omap.OracleAdam(sc, cfg)
    rows = []
    for it in range(n_iters):
        ag.step()
        fs.synchronize()
        if dev.type == "cuda":
            torch.cuda.synchronize()
        rays_o, rays_d, tgt_rgb, tgt_d, z = cpu(fs.rays_o), cpu(fs.rays_d), cpu(fs.tgt_rgb), cpu(fs.tgt_d), cpu(fs.z_vals)
        valid = (tgt_d > 0) & (tgt_d < cfg["cam"]["depth_trunc"])
        L = cpu(fs.losses)
        hip = (float(L[7]), float((cpu(fs.depth) - tgt_d)[valid].abs().mean()))
        opt.zero_grad()
        ret = sc.forward(rays_o, rays_d, tgt_rgb, tgt_d[:, None], impl="grid_sample", z_vals=z)
        omap.loss_from_ret(cfg, ret, is_co_sdf=cfg["is_co_sdf"]).backward()
        opt.step()
        ora = (float(ret["psnr"].detach()), float((ret["depth"].detach() - tgt_d)[valid].abs().mean()))
        rows.append((it, hip[0], ora[0], hip[1], ora[1]))
    return rows


def check_quality_trajectory(device, n_iters=30):
    cfg = configs.bench_office0(n_range_d=9, n_samples_d=20)
    cfg["mapping"]["bound"] = [[-1.0, 1.0], [-1.2, 1.1], [-0.8, 0.9]]
    cfg["planes_res"] = {"coarse": 0.1, "fine": 0.05, "bound_dividable": 0.1}
    cfg["mapping"]["sample"], cfg["mapping"]["min_pixels_cur"] = 256, 32
    cfg["cam"]["far"] = 4.0
    rows = quality_trajectory(device, cfg, n_iters=n_iters)
    # same batches, same start: the two trajectories coincide to rounding for the first iterations and stay within a
    # small band afterwards (Adam with eps = 1e-15 amplifies fp32 summation-order noise on cells with tiny gradients)
    for it, p_h, p_o, d_h, d_o in rows[:3]:
        assert abs(p_h - p_o) < 1e-3 and abs(d_h - d_o) < 1e-4 * max(1.0, d_o), (it, p_h, p_o, d_h, d_o)
    for it, p_h, p_o, d_h, d_o in rows:
        assert abs(p_h - p_o) < 0.05 + 0.01 * abs(p_o) and abs(d_h - d_o) < 0.02 * max(d_o, 0.05), (it, p_h, p_o, d_h, d_o)
    assert rows[-1][1] > rows[0][1] + 1.0, "PSNR did not improve"
    return rows


# ---------------------------------------------------------------------------------------------------------------------
# NS-a: the full JointEncoding surface of the hash / dense grid model against oracle.scene_rep.OracleHashScene
# ---------------------------------------------------------------------------------------------------------------------
def _hash_model_and_oracle(device, cfg, seed=4, table_scale=200.0):
    from oracle.scene_rep import OracleHashScene
    from mneslam_amd.model.scene_rep_hash import HashJointEncoding
    torch.manual_seed(seed)
    bb = torch.tensor(cfg["mapping"]["bound"], dtype=torch.float64)
    m = HashJointEncoding(cfg, bb.to(device)).to(device)
    with torch.no_grad():
        m.embed_fn.params.mul_(table_scale)                 # U(-1e-4, 1e-4) features would leave the decoder blind
    gc = m.embed_fn.cfg
    scales, _, _, _ = m.embed_fn.level_table()
    grid = dict(n_levels=gc.n_levels, n_features=gc.n_features, base_resolution=gc.base_resolution,
                per_level_scale=gc.per_level_scale, log2_hashmap_size=gc.log2_hashmap_size)
    if gc.grid_type == 1:
        grid["grid_type"] = "dense"
    sc = OracleHashScene(cfg, bb, m.embed_fn.params.detach().cpu().clone(), grid, scales=scales)
    sd = {k: v.detach().cpu().clone() for k, v in m.decoder.state_dict().items()}
    sc.col_w = [sd["color_net.model.0.weight"], sd["color_net.model.2.weight"]]
    sc.sdf_w = [sd["sdf_net.model.0.weight"], sd["sdf_net.model.2.weight"]]
    return m, sc


def _synthetic_rays(cfg, n_rays, seed=0, img=(34, 60)):
    """Rays of a synthetic frame inside the config's bound (camera rays, rgb, depth with a few invalid pixels)."""
    from mneslam_amd import synthetic
    H, W = img
    fr = synthetic.make_frames(1, H, W, W / 2.0, W / 2.0, (W - 1) / 2.0, (H - 1) / 2.0, synthetic.room_from_config(cfg), seed=seed)[0]
    g = torch.Generator().manual_seed(seed + 1)
    idx = torch.randperm(H * W, generator=g)[:n_rays]
    d_cam = fr["direction"].reshape(-1, 3)[idx]
    rays_d = torch.sum(d_cam[:, None, :] * fr["c2w"][:3, :3], -1)
    rays_o = fr["c2w"][None, :3, 3].repeat(n_rays, 1)
    return rays_o, rays_d, fr["rgb"].reshape(-1, 3)[idx], fr["depth"].reshape(-1, 1)[idx], fr


def check_hash_scene_api(device, cfg, n_rays=48, co=False, img=(34, 60)):
    """HashJointEncoding.render_rays / forward (+ backward) / render_maps / render_img / query_* against the oracle."""
    cfg = copy.deepcopy(cfg)
    cfg["is_co_sdf"] = co
    m, sc = _hash_model_and_oracle(device, cfg)
    m.train()
    H, W = img
    cam = dict(H=H, W=W, fx=W / 2.0, fy=W / 2.0, cx=(W - 1) / 2.0, cy=(H - 1) / 2.0)
    rays_o, rays_d, rgb, dep, fr = _synthetic_rays(cfg, n_rays, img=img)
    dev = torch.device(device)
    ro, rd, tr_, td = rays_o.to(dev), rays_d.to(dev), rgb.to(dev), dep.to(dev)
    tr = cfg["training"]
    S = tr["n_range_d"] + tr["n_samples_d"]
    U = torch.rand(n_rays, S, generator=torch.Generator().manual_seed(3))
    # ---- render_rays with depth guidance: maps and raw of EVERY sample
    out = m._render(ro, rd, None, td, u=U.to(dev))
    z = out[5].cpu()
    ref = sc.render_rays(rays_o, rays_d, target_d=dep, z_vals=z)
    assert_close(out[0].cpu(), ref["rgb"], rtol=1e-4, atol=2e-5, what="hash render_rays rgb")
    assert_close(out[1].cpu(), ref["depth"], rtol=1e-4, atol=2e-5, what="hash render_rays depth")
    assert_close(out[6].cpu(), ref["raw"], rtol=1e-4, atol=2e-5, what="hash render_rays raw")
    assert float((out[0].detach().cpu() - ref["rgb"]).abs().mean()) < 1e-4 and float((out[1].detach().cpu() - ref["depth"]).abs().mean()) < 1e-4
    # ---- forward: seven losses + psnr, then the gradients of the table and the decoder through autograd
    sc.requires_grad_(True)
    # (forward() draws its own jitter; the comparison needs a fixed U: _render is what forward() calls)
    res = m._render(ro, rd, tr_, td, u=U.to(dev))
    L = res[7]
    ref = sc.forward(rays_o, rays_d, rgb, dep, z_vals=z)
    for k, key in enumerate(LOSS_KEYS):
        assert_close(L[k].detach().cpu(), ref[key].detach().reshape(()), rtol=1e-4, atol=1e-7, what="hash " + key)
    ret_hip = {"rgb_loss": L[0], "depth_loss": L[1], "co_sdf_loss": L[2], "co_fs_loss": L[3], "e_fs_loss": L[4],
               "e_center_loss": L[5], "e_tail_loss": L[6]}
    slam_glue.get_loss_from_ret(cfg, ret_hip, is_co_sdf=co).backward()
    omap.loss_from_ret(cfg, ref, is_co_sdf=co).backward()
    g_t = m.embed_fn.params.grad.cpu()
    assert float(sc.table.grad.abs().max()) > 0
    assert_close(g_t, sc.table.grad, rtol=2e-3, atol=2e-5 * float(sc.table.grad.abs().max()), what="hash table grad (autograd path)")
    for w, w_ref, nm in zip([m.decoder.color_net.model[0].weight, m.decoder.color_net.model[2].weight,
                             m.decoder.sdf_net.model[0].weight, m.decoder.sdf_net.model[2].weight], sc.decoder_list(), DEC_KEYS):
        assert_close(w.grad.cpu(), w_ref.grad, rtol=2e-3, atol=2e-5 * max(1.0, float(w_ref.grad.abs().max())), what=f"hash decoder grad {nm}")
    # ---- R13 on the grid model: gradients of a loss on the rendered maps w.r.t. the RAYS (what the pose loops of loop closure
    # differentiate, mp_slam/mapper.py:388-408) -- OneBlob share + trilinear-weight share -- against autograd through the spec
    if tr.get("n_samples"):
        U3 = torch.rand(n_rays, tr["n_samples"], generator=torch.Generator().manual_seed(11))
        ro_g, rd_g = ro.clone().requires_grad_(True), rd.clone().requires_grad_(True)
        o3 = m._render(ro_g, rd_g, None, None, u=U3.to(dev))
        wr = torch.rand(n_rays, 3, generator=torch.Generator().manual_seed(12))
        wd_ = torch.rand(n_rays, generator=torch.Generator().manual_seed(13))
        ((o3[0] * wr.to(dev)).sum() + 0.1 * (o3[1] * wd_.to(dev)).sum()).backward()
        ro_r, rd_r = rays_o.clone().requires_grad_(True), rays_d.clone().requires_grad_(True)
        ref3 = sc.render_rays(ro_r, rd_r, target_d=None, z_vals=o3[5].detach().cpu())
        ((ref3["rgb"] * wr).sum() + 0.1 * (ref3["depth"] * wd_).sum()).backward()
        assert float(ro_r.grad.abs().max()) > 0 and float(rd_r.grad.abs().max()) > 0
        assert_close(ro_g.grad.cpu(), ro_r.grad, rtol=2e-3, atol=2e-4 * float(ro_r.grad.abs().max()), what="hash d/d rays_o")
        assert_close(rd_g.grad.cpu(), rd_r.grad, rtol=2e-3, atol=2e-4 * float(rd_r.grad.abs().max()), what="hash d/d rays_d")
        m.zero_grad(set_to_none=True)
    # ---- the public entry points with their own jitter: shapes / finiteness, eval-mode forward == render_rays dict
    d1 = m.render_rays(ro, rd, target_d=td)
    assert set(d1) == {"rgb", "depth", "disp_map", "acc_map", "depth_var", "z_vals", "raw"} and d1["raw"].shape == (n_rays, S, 4)
    d2 = m.forward(ro, rd, tr_, td)
    assert set(d2) >= {"rgb", "depth", "rgb_loss", "psnr"} and torch.isfinite(d2["rgb_loss"])
    # ---- without depth guidance (training.n_samples uniform samples)
    if tr.get("n_samples"):
        U2 = torch.rand(n_rays, tr["n_samples"], generator=torch.Generator().manual_seed(5))
        o2 = m._render(ro, rd, None, None, u=U2.to(dev))
        ref2 = sc.render_rays(rays_o, rays_d, target_d=None, z_vals=o2[5].cpu())
        assert_close(o2[0].cpu(), ref2["rgb"].detach(), rtol=1e-4, atol=2e-5, what="hash render_rays rgb (no depth)")
        assert_close(o2[1].cpu(), ref2["depth"].detach(), rtol=1e-4, atol=2e-5, what="hash render_rays depth (no depth)")
    # ---- render_maps (no-grad, exact early termination) == render_rays on the same jitter
    mp = m.render_maps(ro, rd, target_d=td, u=U.to(dev))
    assert_close(mp["rgb"].cpu(), out[0].detach().cpu(), rtol=1e-5, atol=1e-6, what="hash render_maps rgb")
    assert_close(mp["depth"].cpu(), out[1].detach().cpu(), rtol=1e-5, atol=1e-6, what="hash render_maps depth")
    # ---- render_img: whole frame == render_maps over its rays, chunked or not
    m.eval()
    m.config["training"]["perturb"] = 0.0
    cam_backup = dict(cfg["cam"])
    m.config["cam"].update(cam, crop_edge=0)
    depth_img, color_img = m.render_img(fr["c2w"], dev, gt_depth=fr["depth"])
    m.render_chunk_rays = max(H * W // 3, 1)
    depth_img2, color_img2 = m.render_img(fr["c2w"], dev, gt_depth=fr["depth"])
    del m.render_chunk_rays
    assert depth_img.shape == (H, W) and depth_img.dtype == torch.float64 and color_img.shape == (H, W, 3)
    assert_close(depth_img2.cpu(), depth_img.cpu(), rtol=1e-6, atol=1e-7, what="hash render_img depth (chunked)")
    assert_close(color_img2.cpu(), color_img.cpu(), rtol=1e-6, atol=1e-7, what="hash render_img colour (chunked)")
    sc.requires_grad_(False)
    from mneslam_amd.model.utils import get_rays
    io, id_ = get_rays(H, W, cam["fx"], cam["fy"], cam["cx"], cam["cy"], fr["c2w"], "cpu")
    zi = m.render_rays(io.reshape(-1, 3).to(dev), id_.reshape(-1, 3).to(dev), target_d=fr["depth"].reshape(-1, 1).to(dev))["z_vals"].cpu()
    refi = sc.render_rays(io.reshape(-1, 3), id_.reshape(-1, 3), target_d=fr["depth"].reshape(-1, 1), z_vals=zi)    # perturb = 0: no jitter
    assert_close(depth_img.float().cpu().reshape(-1), refi["depth"], rtol=1e-4, atol=2e-5, what="hash render_img depth vs oracle")
    assert_close(color_img.cpu().reshape(-1, 3), refi["rgb"], rtol=1e-4, atol=2e-5, what="hash render_img colour vs oracle")
    m.config["cam"].update(cam_backup)
    # ---- point queries
    g = torch.Generator().manual_seed(9)
    lo, hi = sc.bounding_box[:, 0].float(), sc.bounding_box[:, 1].float()
    pts = lo + (hi - lo) * torch.rand(5, 37, 3, generator=g)
    raw_ref = sc.query_color_sdf(pts).reshape(5, 37, 4)
    assert_close(m.query_color_sdf(pts.to(dev)).cpu().reshape(5, 37, 4), raw_ref, rtol=1e-4, atol=2e-5, what="hash query_color_sdf")
    assert_close(m.query_sdf(pts.to(dev)).cpu(), raw_ref[..., 3], rtol=1e-4, atol=2e-5, what="hash query_sdf")
    sdf, geo = m.query_sdf(pts.to(dev), return_geo=True)
    assert geo.shape == (5, 37, 15)
    assert_close(m.query_color(pts.to(dev)).cpu().reshape(5, 37, 3), torch.sigmoid(raw_ref[..., :3]), rtol=1e-4, atol=2e-5, what="hash query_color")
    emb = m.query_sdf(pts.to(dev), embed=True)
    feat_ref = sc.grid_features(pts.reshape(-1, 3))[:, :emb.shape[-1]].reshape(5, 37, -1)
    assert_close(emb.cpu(), feat_ref, rtol=1e-5, atol=1e-7, what="hash query_sdf(embed=True)")
    assert_close(m.run_network(pts.to(dev)).cpu(), raw_ref, rtol=1e-4, atol=2e-5, what="hash run_network")
    # ray gradients through the public method, with depth guidance as well: finite and non-zero
    ro_g = ro.clone().requires_grad_(True)
    m.train()
    m.render_rays(ro_g, rd, target_d=td)["depth"].sum().backward()
    assert torch.isfinite(ro_g.grad).all() and float(ro_g.grad.abs().max()) > 0
    return {"S": S}


def dense_grid_config():
    """BASELINE.json configs[0] in its as-north-star form: 16^3 dense grid (get_encoder('dense',
    base_resolution=16, desired_resolution=16): 4 levels x 16^3 x 2 features) + 2x32 MLPs, 512 rays x 64 samples
    (n_range_d 21 + n_samples_d 43), office0 bound."""
    cfg = configs.bench_office0_hash(hidden=32, hash_size=19, desired_resolution=16)
    cfg["grid"]["enc"] = "dense"
    cfg["training"]["n_range_d"], cfg["training"]["n_samples_d"] = 21, 43
    cfg["mapping"]["sample"] = 512
    return cfg


def check_checkpoint_handoff(device, tmp_dir):
    """N4 on the device: an agent publishes its map from a device-resident model (slam_glue.save_latest_checkpoint), a peer
    reads it into its device-resident ``model_shared`` (slam_glue.load_foreign_model; reference: mneslam_mp.py:294-315,
    mp_slam/mapper.py:708-726) and uses it as the distillation teacher: ``render_maps`` without depth guidance
    (mp_slam/mapper.py:617-622) against the oracle holding the publisher's parameters."""
    g = load_golden("fwd_onegrid")
    cfg = configs.small_test_config(one_grid=True)
    cfg["data"].update(output=str(tmp_dir), exp_name="handoff")
    publisher = model_from_golden(g, cfg, device).train()
    path = slam_glue.save_latest_checkpoint(publisher, cfg, rank=1)
    # the reader starts from another map of ANOTHER shape (its own bound) and is replaced wholesale
    cfg2 = copy.deepcopy(cfg)
    cfg2["mapping"]["bound"] = [[-0.5, 0.7], [-0.6, 0.6], [-0.4, 0.5]]
    shared = JointEncoding(cfg2, torch.tensor(cfg2["mapping"]["bound"], dtype=torch.float64)).to(device)
    ckpt = slam_glue.load_foreign_model(shared, cfg, 1, torch.device(device))
    assert set(ckpt) == {"model", "all_planes", "bound", "bounding_box"} and os.path.dirname(path) == slam_glue.agent_dir(cfg, 1)
    assert not shared.training
    for lst_s, lst_p in zip(shared.all_planes, publisher.all_planes):
        for p, q in zip(lst_s, lst_p):
            assert p.device.type == torch.device(device).type and p.is_contiguous(memory_format=torch.channels_last)
            assert torch.equal(p, q.detach())
    rays_o, rays_d, *_ = fixture_inputs(g)
    n = rays_o.shape[0]
    u = torch.rand(n, cfg["training"]["n_samples"], generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        got = shared.render_maps(rays_o.to(device), rays_d.to(device), target_d=None, u=u.to(device))
    sc = oracle_scene_from_golden(g, cfg)
    with torch.no_grad():
        want = sc.render_rays(rays_o, rays_d, target_d=None, u=u)
    assert_close(got["rgb"].cpu(), want["rgb"], rtol=1e-4, atol=2e-5, what="teacher rgb after the hand-off")
    assert_close(got["depth"].cpu(), want["depth"], rtol=1e-4, atol=2e-5, what="teacher depth after the hand-off")


# --------------------------------------------------------------------------------------------------------------
# EXTENSION: two agents on one lattice, binned plane update, shared cells + shared decoder
# --------------------------------------------------------------------------------------------------------------
def lattice_config(rank):
    """Slabs along x whose planes sit on ONE lattice on both levels: extended x length 2.4 m with 13 / 25 nodes
    (spacing 0.2 / 0.1 m), agent r shifted by r * 1.4 m = 7 coarse / 14 fine nodes (neighbours share 1.0 m, agents two apart
    nothing); y and z extents are the agents' common ones."""
    cfg = configs.small_test_config(one_grid=True, is_co_sdf=False, n_samples_d=21, n_range_d=11)
    x0 = -1.0 + 1.4 * rank
    cfg["mapping"]["bound"] = [[x0, x0 + 2.3], [-1.2, 1.1], [-0.8, 0.9]]             # extended by bound_dividable to 2.4 / 2.4 / 1.8
    cfg["planes_res"] = {"coarse": 0.181, "fine": 0.095, "bound_dividable": 0.2}
    room = [[x0 + 0.2, x0 + 2.2], [-1.0, 0.9], [-0.6, 0.7]]
    return cfg, room


def run_overlap_agent(rank, device, comm, geometry="lattice", world=2):
    """One of ``world`` agents in a chain of overlapping slabs on one lattice: FusedStep (binned plane update, overlap_peers =
    its one or two neighbours + shared_decoder) against an oracle agent with the exchange written out in tensor ops -- plane
    gradients summed over the node rectangles neighbours both hold, decoder gradient averaged over all agents, then Adam.
    Equal by construction to ONE model over the union lattice trained on the union batch wherever cells are shared.  Checks,
    over two iterations: every plane and decoder parameter against the oracle agent, and the shared cells and the decoder
    bit-equal between the HIP agents.  ``comm.all_gather(obj)`` -> the agents' objects by rank (gloo processes on the CPU,
    threads of one process on the GPU).  An INTERIOR agent (world >= 3) runs mne_tile_grad_export / mne_tile_adam_shared with
    two rectangles per plane and exchanges with both neighbours in one step (mp_slam/mapper.py:491-509: an agent's bound
    intersects those of the agents on either side; configs/Indoor/indoor.yaml:169-173)."""
    from mneslam_amd import dist as mdist, synthetic
    from mneslam_amd.fused import FusedStep
    if geometry in ("apartment", "scannet"):
        # BASELINE configs[2] / configs[3] as worded, at full size: the Replica apartment scene split into two / ScanNet
        # scene0000 (colour planes) split into four overlapping slabs on one lattice (configs.split_agent_config: what
        # bench.py --split / its as_worded side record run)
        base = configs.WORKLOADS[geometry][0]()
        base["mapping"]["bound"] = [list(b) for b in configs.SCENE_BOUNDS[geometry]]
        cfg, axis, _ = configs.split_agent_config(base, world, rank)
        room = cfg["mapping"]["marching_cubes_bound"]
    else:
        cfg, room = lattice_config(rank)
    bb = torch.tensor(cfg["mapping"]["bound"], dtype=torch.float64)
    with getattr(comm, "lock", contextlib.nullcontext()):        # (threads of one process share the global generator)
        torch.manual_seed(11)                                    # the same decoder on all agents
        model = JointEncoding(cfg, bb).to(device).train()
    # one decoder for all agents: the planes' draws come first in the constructor and the slabs need not have the same
    # number of nodes, so the seed alone does not give equal decoders -- agent 0's initialisation is the common one
    dec0 = comm.all_gather({k: v.detach().cpu().clone() for k, v in model.decoder.state_dict().items()})[0]
    model.decoder.load_state_dict({k: v.to(device) for k, v in dec0.items()})
    geo = mdist.plane_geometry(model)
    all_geo = comm.all_gather(geo)
    nbs = [r for r in (rank - 1, rank + 1) if 0 <= r < world]
    # planes = windows of one field over the union lattice (same seed on all ranks), so shared cells start equal.  Node
    # offset of every agent on the union lattice, from the overlap of consecutive agents: off[a+1] = off[a] + start(a) - start(a+1)
    flat = [p for lst in model.all_planes for p in lst]
    rects = {nb: [] for nb in nbs}
    for k, p in enumerate(flat):
        off = [(0, 0)]
        for a in range(world - 1):
            (sa, ba, axa), (sb, bb_, _) = all_geo[a][k], all_geo[a + 1][k]
            (ys, xs), (pys, pxs) = mdist.overlap_slices(ba, bb_, sa, sb, axa)
            off.append((off[a][0] + ys.start - pys.start, off[a][1] + xs.start - pxs.start))
        hu = max(o[0] + all_geo[a][k][0][0] for a, o in enumerate(off))
        wu = max(o[1] + all_geo[a][k][0][1] for a, o in enumerate(off))
        shape = geo[k][0]
        field = 0.05 * torch.randn(1, p.shape[1], hu, wu, generator=torch.Generator().manual_seed(100 + k))
        oy, ox = off[rank]
        with torch.no_grad():
            p.copy_(field[..., oy:oy + shape[0], ox:ox + shape[1]].to(device))
        del field
        for nb in nbs:
            (sh, bnd, axes), (psh, pbnd, _) = geo[k], all_geo[nb][k]
            rects[nb].append(mdist.overlap_slices(bnd, pbnd, sh, psh, axes))
    if world > 2 and 0 < rank < world - 1:                        # the two rectangles of an interior agent do not meet
        for (sh, _, axes), ra, rb in zip(geo, rects[nbs[0]], rects[nbs[1]]):
            (ya, xa), (yb, xb) = ra[0], rb[0]
            whole = ya == yb == slice(0, sh[0]) and xa == xb == slice(0, sh[1])        # (a plane without the slab axis: every agent's)
            assert whole or ya.stop <= yb.start or yb.stop <= ya.start or xa.stop <= xb.start or xb.stop <= xa.start
    opt = slam_glue.create_optimizer(model, cfg)
    flat = [p for lst in model.all_planes for p in lst]          # (now the nn.Parameters the optimizer's groups hold)
    n_rays = 40
    # more than two agents: the planes without the slab axis are held by EVERY agent and take the sum over all agents
    gax = None
    if world > 2:
        gax = axis if geometry in ("apartment", "scannet") else 0
    group = [gax is not None and gax not in g[2] for g in geo]
    fs = FusedStep(model, opt, cfg, n_rays, device, scatter="binned", shared_decoder=True,
                   overlap_peers=[(nb, all_geo[nb]) for nb in nbs], overlap_group_axis=gax)
    assert fs.tile_overlap is not None and fs.tile_overlap.n_peers == len(nbs) + (gax is not None)
    assert all(s.numel() > 32 * 25 for s in fs.ov_send)
    H, W = 34, 60
    frames = synthetic.make_frames(2, H, W, 30.0, 30.0, 29.5, 16.5, room, seed=3 + rank)
    fr = frames[1]
    cur = torch.cat([fr["direction"], fr["rgb"], fr["depth"][..., None]], -1).reshape(-1, 7).contiguous().to(device)
    poses = fr["c2w"].reshape(1, 4, 4).contiguous().to(device)
    # ---- oracle agent
    cpu = lambda t: t.detach().cpu().clone()
    sc = OracleScene(cfg, bb, build=False)
    sc.all_planes = tuple([cpu(p).contiguous() for p in lst] for lst in model.all_planes)
    sd = model.decoder.state_dict()
    sc.col_w = [cpu(sd["color_net.model.0.weight"]), cpu(sd["color_net.model.2.weight"])]
    sc.sdf_w = [cpu(sd["sdf_net.model.0.weight"]), cpu(sd["sdf_net.model.2.weight"])]
    sc.requires_grad_(True)
    oopt = omap.OracleAdam(sc, cfg)
    gen = torch.Generator().manual_seed(7 + rank)
    for it in range(2):
        idx = torch.randperm(H * W, generator=gen)[:n_rays]
        fs.step(None, 0, 1, cur, poses, 0, n_rays, idx_cur=idx.to(device), u=torch.rand(n_rays, fs.S, generator=gen).to(device))
        fs.synchronize()
        fs.check()
        oopt.zero_grad()
        r = sc.forward(cpu(fs.rays_o), cpu(fs.rays_d), cpu(fs.tgt_rgb), cpu(fs.tgt_d)[:, None], impl="grid_sample",
                       z_vals=cpu(fs.z_vals))
        omap.loss_from_ret(cfg, r, is_co_sdf=cfg["is_co_sdf"]).backward()
        # what each neighbour needs of this agent's plane gradients: its window of the shared rectangle (not whole planes)
        mine = {nb: [None if grp else p.grad[:, :, ys, xs].clone() for p, ((ys, xs), _), grp in zip(sc.plane_list(), rects[nb], group)]
                for nb in nbs}
        mine["all"] = [p.grad.clone() if grp else None for p, grp in zip(sc.plane_list(), group)]
        theirs = comm.all_gather(mine)
        with torch.no_grad():
            for nb in nbs:
                for p, g_peer, ((ys, xs), _), grp in zip(sc.plane_list(), theirs[nb][rank], rects[nb], group):
                    if not grp:
                        p.grad[:, :, ys, xs] += g_peer
            for k, (p, grp) in enumerate(zip(sc.plane_list(), group)):
                if grp:
                    p.grad.copy_(sum((t["all"][k] for t in theirs[1:]), theirs[0]["all"][k].clone()))
            for w in sc.decoder_list():
                gs = comm.all_gather(w.grad.clone())
                w.grad.copy_(sum(gs[1:], gs[0]) / world)
        oopt.step()
        for k, (p, ref) in enumerate(zip(flat, sc.plane_list())):
            lr = fs.group_of[p]["lr"]
            adam_agreement(cpu(p), ref.detach(), lr, "plane", f"agent {rank} iteration {it} plane {k}")
        for w_hip, w_ref in zip(model.decoder.parameters(), sc.decoder_list()):
            adam_agreement(cpu(w_hip), w_ref.detach(), opt.param_groups[0]["lr"], "decoder", f"agent {rank} iteration {it}: decoder")
    # the exchange carried something: the shared cells' first moments hold the peers' share too
    assert all(float(ex.abs().max()) > 0 for ex in fs.ov_recv)
    # shared cells are bit-equal on neighbouring agents (a + b == b + a; same moments, same step), the rest is not
    win = comm.all_gather({nb: [cpu(p)[:, :, ys, xs].clone() for p, ((ys, xs), _) in zip(flat, rects[nb])] for nb in nbs})
    for nb in nbs:
        for k, (a, b) in enumerate(zip(win[rank][nb], win[nb][rank])):
            assert torch.equal(a, b), f"plane {k}: cells shared by agents {rank} and {nb} differ"
    if gax is not None:                                           # planes every agent holds: one copy, bit for bit, on all of them
        every = comm.all_gather([cpu(p) if grp else None for p, grp in zip(flat, group)])
        for k, grp in enumerate(group):
            assert not grp or all(torch.equal(every[0][k], e[k]) for e in every[1:]), f"plane {k} (held by every agent) drifted apart"
    dec = torch.cat([cpu(p).reshape(-1) for p in model.decoder.parameters()])
    both = comm.all_gather(dec)
    assert all(torch.equal(both[0], d) for d in both[1:])


def check_sample_z_frame_counts(device, R=20011):
    """mne_sample_z on a batch larger than MNE_BALANCED_MAX_RAYS (whole frames): the mask counts are summed by many workgroups;
    they must equal the sum of the per-ray counts."""
    import ctypes as C
    from mneslam_amd import hip_path, _lib
    lib = _lib.load()
    dev = torch.device(device)
    cfg = configs.small_test_config()
    rc = hip_path.render_cfg_struct(cfg)
    S = lib.mne_num_samples(C.byref(rc), 1)
    tables = hip_path.linspace_tables(cfg, True, dev)
    g = torch.Generator().manual_seed(4)
    d = (torch.rand(R, generator=g) * 4.0).to(dev)
    d[::7] = 0.0                                           # rays without depth
    P = _lib.ptr

    def run(lo, hi):
        n = hi - lo
        z = torch.empty(n, S, device=dev)
        cnt = torch.full((8,), -1, dtype=torch.int32, device=dev)
        rcnt = torch.zeros(n, 8, dtype=torch.int32, device=dev)
        dd = d[lo:hi].contiguous()
        _lib.check(lib.mne_sample_z(C.byref(rc), n, P(dd), None, P(tables), 5, lo * S // 4 * 4, P(z), P(cnt), P(rcnt), None,
                                    _lib.stream_for(z)), "mne_sample_z")
        return cnt.cpu().long(), rcnt.cpu().long()

    cnt, rcnt = run(0, R)
    keep = list(range(7))                                  # slot 7 (MNE_C_TILE0) is per-ray scratch of the balanced decode schedule
    assert torch.equal(cnt[keep], rcnt.sum(0)[keep]), (cnt, rcnt.sum(0))
    cnt1, rcnt1 = run(0, 4096)                             # a batch below the bound: the one-workgroup form, same contract
    assert torch.equal(cnt1[keep], rcnt1.sum(0)[keep]) and torch.equal(rcnt1[:, keep], rcnt[:4096, keep])
