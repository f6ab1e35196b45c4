"""HashJointEncoding (NS-a, EXTENSION): host-side construction -- no kernel runs here (the iteration and the full
render / query surface are covered by tests/test_kernels_hostemu.py and the -m gpu tests)."""
import pytest
import torch

from mneslam_amd import configs, slam_glue
from mneslam_amd.model.scene_rep_hash import HashJointEncoding
from oracle import hashgrid


def _cfg(hash_size=12, hidden=32):
    cfg = configs.bench_office0_hash(hidden=hidden, hash_size=hash_size, desired_resolution=128)
    return cfg


def test_construction_matches_the_factory_defaults_and_the_spec():
    cfg = _cfg()
    m = HashJointEncoding(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float64))
    g = m.embed_fn.cfg
    assert (g.n_levels, g.n_features, g.base_resolution, g.log2_hashmap_size) == (16, 2, 16, 12)     # model/encodings.py:6-10
    n_spec = hashgrid.n_params(n_levels=16, n_features=2, base_resolution=16, per_level_scale=g.per_level_scale,
                               log2_hashmap_size=12)
    assert m.embed_fn.params.numel() == n_spec
    assert float(m.embed_fn.params.abs().max()) <= 1e-4                                               # U(-1e-4, 1e-4)
    # decoder: 64-wide feature slot, dead columns zero; Co-SLAM's shapes otherwise
    w0 = m.decoder.sdf_net.model[0].weight
    assert tuple(w0.shape) == (32, 64 + 48) and float(w0[:, 32:64].abs().max()) == 0.0 and float(w0[:, :32].abs().max()) > 0
    assert m.all_planes == ()
    keys = set(m.state_dict().keys())
    assert {"embed_fn.params", "embedpos_fn.params", "decoder.sdf_net.model.0.weight", "decoder.color_net.model.2.weight"} <= keys
    with pytest.raises(NotImplementedError):
        m.sample_plane_feature(torch.zeros(1, 3), [], [], [])


def test_headline_table_size_and_optimizer_groups():
    cfg = configs.bench_office0_hash()                      # T = 2^19, finest 512: SURVEY R14's 10,492,048 floats
    m = HashJointEncoding(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float64))
    assert m.embed_fn.params.numel() == 10492048 and cfg["decoder"]["hidden_dim"] == 64
    opt = slam_glue.create_optimizer(m, cfg)
    g_dec, g_tab = opt.param_groups
    assert g_dec["weight_decay"] == 1e-6 and g_dec["lr"] == cfg["mapping"]["lr_decoder"] and len(g_dec["params"]) == 4
    assert g_tab["eps"] == 1e-15 and g_tab["lr"] == cfg["mapping"]["lr_embed"] and g_tab["params"][0] is m.embed_fn.params
    assert tuple(g_tab["betas"]) == (0.9, 0.99)
