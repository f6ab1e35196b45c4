"""The C-ABI library builds, loads and exports every symbol include/mneslam_hip.h declares.
No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from mneslam_amd import _lib, build

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "mneslam_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mne_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    return build.build()          # hipcc cross-compiles gfx950 without a GPU; no-op when up to date


def test_header_and_binding_agree():
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} missing from libmneslam_hip.so"


def test_struct_layouts_and_version(lib_path):
    _lib.unload()
    lib = _lib.load(lib_path)       # checks ABI version and sizeof(struct) against ctypes
    assert lib.mne_abi_version() == _lib.ABI_VERSION
    _lib.unload()


def test_argument_validation_without_gpu(lib_path):
    _lib.unload()
    lib = _lib.load(lib_path)
    assert lib.mne_adam_step(None, 33, 0, None, None) < 0
    assert b"n_seg" in lib.mne_last_error()
    rc = _lib.RenderCfg()
    rc.n_samples_d, rc.n_range_d, rc.n_samples = 32, 11, 256
    assert lib.mne_num_samples(ctypes.byref(rc), 1) == 43
    assert lib.mne_num_samples(ctypes.byref(rc), 0) == 256
    sc = _lib.Scene()
    sc.n_sets, sc.c_dim = 1, 16
    assert lib.mne_pack_decoder(ctypes.byref(sc), None, None) < 0
    assert b"c_dim" in lib.mne_last_error()
    # workspace sizing of the backward: monotone in both arguments, zero for empty batches
    assert lib.mne_render_workspace_bytes(0, 128) == 0 and lib.mne_render_workspace_bytes(16, 0) == 0
    w1, w2, w3 = (lib.mne_render_workspace_bytes(*a) for a in ((64, 43), (64, 128), (2150, 128)))
    assert 0 < w1 < w2 < w3 and w3 >= 2150 * 128 * 16          # ReLU bit masks: 16 B per sample
    # a well-formed scene with NULL buffers: every entry point of the path rejects it before touching the device
    sc = _lib.Scene()
    sc.n_sets, sc.c_dim, sc.hidden, sc.hidden_color, sc.geo_feat_dim, sc.n_bins = 1, 32, 32, 32, 15, 16
    assert lib.mne_render_forward(ctypes.byref(sc), ctypes.byref(rc), 8, 43, *([None] * 14), 0, None) < 0
    assert b"plane" in lib.mne_last_error()
    assert lib.mne_tile_order(ctypes.byref(sc), None, None) < 0
    assert lib.mne_tile_adam(ctypes.byref(sc), None, None, None, None, None) < 0
    assert lib.mne_loss_finalize(8, 43, None, None, None, None) < 0 and b"NULL" in lib.mne_last_error()
    assert lib.mne_sample_z(ctypes.byref(rc), 8, None, None, None, 0, 0, None, None, None, None, None) < 0
    assert lib.mne_decoder_wgrad(ctypes.byref(sc), None, None, 8, 43, None, None, 0, None) < 0
    assert lib.mne_tile_count(None) == 0 and lib.mne_tape_row_floats(None) == 0
    # round-3 entry points: list sizing with per-plane capacities, overlap rectangles, pose loop, batch / decoder update
    sc.plane[0][0][0].h, sc.plane[0][0][0].w = 40, 33                    # 3 x 3 tiles; the other planes are empty
    bins = _lib.TileBins()
    bins.cap = 100
    # (ABI 7: a list = mne_tile_list_segments() equal segments, one per XCD: capacities are rounded down to a multiple of it, at least one entry each)
    assert lib.mne_tile_list_segments() == _lib.LIST_SEGMENTS == 8
    assert lib.mne_tile_list_entries(ctypes.byref(sc), ctypes.byref(bins)) == 9 * 96
    bins.plane_cap[0] = 7
    assert lib.mne_tile_list_entries(ctypes.byref(sc), ctypes.byref(bins)) == 9 * 8
    bins.plane_cap[0] = 50
    assert lib.mne_tile_list_entries(ctypes.byref(sc), ctypes.byref(bins)) == 9 * 48
    bins.plane_cap[0] = -1
    assert lib.mne_tile_list_entries(ctypes.byref(sc), ctypes.byref(bins)) == 0
    ov = _lib.TileOverlap()
    ov.n_peers = 1
    ov.rect[0][0].x0, ov.rect[0][0].x1, ov.rect[0][0].y0, ov.rect[0][0].y1 = 2, 10, 5, 9
    assert lib.mne_tile_overlap_floats(ctypes.byref(sc), ctypes.byref(ov), 0) == 8 * 4 * 32
    assert lib.mne_tile_overlap_floats(ctypes.byref(sc), ctypes.byref(ov), 1) == 0
    assert lib.mne_tile_grad_export(ctypes.byref(sc), None, None, ctypes.byref(ov), None) < 0
    assert lib.mne_tile_adam_shared(ctypes.byref(sc), None, None, None, ctypes.byref(ov), None, None) < 0
    ps = _lib.PoseState()
    assert lib.mne_pose_rays(ctypes.byref(ps), 8, None, None, None, None) < 0 and b"NULL" in lib.mne_last_error()
    assert lib.mne_pose_loss(8, *([None] * 4), 1.0, 1.0, None, None, None, None) < 0
    assert lib.mne_pose_update(ctypes.byref(ps), 8, None, None, None, None, None) < 0
    assert lib.mne_sample_batch(*([None] * 1), 0, 1, None, None, 0, None, 0, 0, 0, None, None, 0, 0, *([None] * 5),
                                ctypes.byref(rc), *([None] * 2), 0, *([None] * 5), None, None) < 0
    assert lib.mne_decoder_update(ctypes.byref(sc), None, 8, None, None, 43, None, None, None, None, None) < 0
    _lib.unload()


def test_missing_library_fails_loudly(tmp_path):
    _lib.unload()
    with pytest.raises(RuntimeError, match="only backend"):
        _lib.load(str(tmp_path / "nope.so"))
    _lib.unload()
