"""Shared helpers for the tests: build an oracle scene from a golden fixture."""
import os

import numpy as np
import torch

from mneslam_amd import configs
from oracle.scene_rep import OracleScene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

DEC_KEYS = ["color_net.model.0.weight", "color_net.model.2.weight",
            "sdf_net.model.0.weight", "sdf_net.model.2.weight"]


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def n_plane_sets(g, prefix=""):
    return 6 if f"{prefix}plane_3_0" in g else 3


def oracle_scene_from_golden(g, cfg, prefix=""):
    """OracleScene with the fixture's planes / decoder weights / bounding box (float64)."""
    sc = OracleScene(cfg, torch.from_numpy(g["bounding_box"]), build=False)
    ns = n_plane_sets(g, prefix)
    sc.all_planes = tuple([torch.from_numpy(g[f"{prefix}plane_{s}_{l}"]).clone() for l in range(2)]
                          for s in range(ns))
    sc.col_w = [torch.from_numpy(g[f"{prefix}dec.color_net.model.0.weight"]).clone(),
                torch.from_numpy(g[f"{prefix}dec.color_net.model.2.weight"]).clone()]
    sc.sdf_w = [torch.from_numpy(g[f"{prefix}dec.sdf_net.model.0.weight"]).clone(),
                torch.from_numpy(g[f"{prefix}dec.sdf_net.model.2.weight"]).clone()]
    return sc


def fixture_inputs(g, requires_grad=False):
    rays_o = torch.from_numpy(g["rays_o"]).clone().requires_grad_(requires_grad)
    rays_d = torch.from_numpy(g["rays_d"]).clone().requires_grad_(requires_grad)
    return rays_o, rays_d, torch.from_numpy(g["target_rgb"]), torch.from_numpy(g["target_d"]), torch.from_numpy(g["U"])


def assert_close(a, b, rtol=1e-5, atol=1e-6, what=""):
    a, b = [t.detach().cpu() if torch.is_tensor(t) else t for t in (a, b)]
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if np.isnan(b).any() or np.isnan(a).any():
        assert np.array_equal(np.isnan(a), np.isnan(b)), f"{what}: NaN pattern differs"
        a, b = np.nan_to_num(a), np.nan_to_num(b)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not bad.any(), (f"{what}: {bad.sum()}/{bad.size} mismatches, max abs err {err.max():.3e} "
                           f"(ref scale {np.abs(b).max():.3e})")
