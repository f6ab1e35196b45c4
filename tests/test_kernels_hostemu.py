"""CPU runs of the HIP kernel SOURCES through the test-only host emulator (tests/hostemu): the same
.hip files compiled with clang++ -DMNE_HOST_EMU, one OS thread per work-item.  This is how kernel
logic is debugged without a GPU; the authoritative parity run is tests/test_hip_parity_gpu.py."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostemu"))

import parity_cases as pc  # noqa: E402
from mneslam_amd import _lib  # noqa: E402

DEV = "cpu"
# The emulator runs a wave as one OS thread with its 64 work-items as fibers (tests/hostemu/hip_emu.h): every case of this
# file, end-to-end iterations included, takes seconds, so all of them are in the default CPU run.


@pytest.fixture(scope="module", autouse=True)
def emulator_library():
    import build_emu
    path = build_emu.build()
    _lib.unload()
    _lib.load(path)
    torch.set_num_threads(2)
    yield
    _lib.unload()


def test_abi_structs_match():
    assert _lib.load().mne_abi_version() == _lib.ABI_VERSION


def test_oneblob():
    pc.check_oneblob(DEV)


def test_spherical_frequency_identity_encodings():
    """get_encoder('SphericalHarmonics' | 'Frequency' | 'Identity'), model/encodings.py:48-58, 73-95"""
    pc.check_misc_encodings(DEV)


def test_adam():
    pc.check_adam(DEV)


def test_forward_onegrid():
    pc.check_forward("fwd_onegrid", DEV)


def test_forward_colorplanes():
    pc.check_forward("fwd_colorplanes", DEV)


def test_backward_onegrid_esdf():
    pc.check_backward("fwd_onegrid", False, DEV)


@pytest.mark.parametrize("name,co", [("fwd_onegrid", True), ("fwd_colorplanes", False), ("fwd_colorplanes", True)])
def test_backward(name, co):
    pc.check_backward(name, co, DEV)


def test_backward_scalar_wgrad_crosscheck():
    pc.check_backward("fwd_onegrid", False, DEV, wgrad_impl=1)


def test_all_invalid():
    pc.check_all_invalid(DEV)


def test_render_nodepth():
    pc.check_render_nodepth(DEV)


def test_queries():
    pc.check_queries(DEV)


def test_device_clock():
    pc.check_device_clock(DEV)


def test_render_maps_fast_path_and_render_img():
    pc.check_render_maps_fast_path(DEV)


def test_corner_indices_bit_exact():
    pc.check_corner_indices(DEV, "fwd_onegrid")


def test_mapping3_onegrid():
    pc.check_mapping3("mapping3_onegrid_esdf", True, False, 21, DEV)


def test_device_sampler():
    pc.check_device_sampler(DEV)


@pytest.mark.parametrize("scatter", ["binned", "atomics"])
def test_mapping3_fused_path(scatter):
    pc.check_mapping3("mapping3_onegrid_esdf", True, False, 21, DEV, compute="fused", scatter=scatter)


def test_mapping3_fused_binned_with_list_overflow():
    """tile lists of 8 entries: most contributions go through the spill area"""
    pc.check_mapping3("mapping3_onegrid_esdf", True, False, 21, DEV, compute="fused", scatter="binned", tile_capacity=8)


def test_mapping3_fused_binned_colorplanes():
    pc.check_mapping3("mapping3_colorplanes_cosdf", False, True, 22, DEV, compute="fused", scatter="binned")


def test_ray_gradients_onegrid():
    pc.check_ray_gradients("fwd_onegrid", False, DEV)


def test_ray_gradients_colorplanes_cosdf():
    pc.check_ray_gradients("fwd_colorplanes", True, DEV)


def test_render_nodepth_pose_gradients():
    pc.check_render_nodepth_pose_gradients(DEV)


@pytest.mark.parametrize("kind", ["hash", "dense"])
def test_grid_encoding(kind):
    pc.check_grid_encoding(DEV, kind)


@pytest.mark.parametrize("compute,absolute", [("autograd", False), ("fused", False), ("fused", True), ("fused", "quat")])
def test_loop_closure_pose_alignment(compute, absolute):
    pc.check_pose_alignment(DEV, compute, absolute)


@pytest.mark.parametrize("compute", ["fused", "autograd"])
def test_loop_closure_pose_alignment_on_the_hash_model(compute):
    pc.check_pose_alignment_hash(DEV, compute)


def test_pose_alignment_falls_back_for_other_parameterisations():
    """A host whose matrix_from_tensor is neither the axis-angle nor the quaternion map keeps its own loop."""
    from mneslam_amd import hip_path
    other = lambda rot, trans: torch.eye(4)[None].repeat(rot.shape[0], 1, 1)
    assert hip_path.probe_axis_angle(other, torch.zeros(1, 4), torch.zeros(1, 3)) is None
    assert hip_path.probe_axis_angle(other, torch.tensor([[0.9, 0.1, 0.2, 0.3]]), torch.zeros(1, 3)) is None
    assert hip_path.probe_axis_angle(other, torch.tensor([[0.1, 0.2, 0.3]]), torch.zeros(1, 3)) is None
    assert hip_path.probe_axis_angle(other, torch.zeros(1, 6), torch.zeros(1, 3)) is None          # e.g. a 6-D rotation


def test_checkpoint_handoff_then_teacher_render(tmp_path):
    pc.check_checkpoint_handoff(DEV, tmp_path)


@pytest.mark.parametrize("compute", ["autograd", "fused"])
def test_loop_closure_distillation(compute):
    pc.check_distillation(DEV, compute)


@pytest.mark.parametrize("n_rays,S_d,S_r", [(1, 4, 3), (5, 20, 13), (3, 1, 1)])
def test_ragged_sizes_vs_oracle(n_rays, S_d, S_r):
    """a single ray; S = 33 (one sample past a 32-sample tile); S = 2 -- forward and gradients vs the oracle"""
    pc.check_oracle_random_scene(DEV, n_rays=n_rays, S_d=S_d, S_r=S_r, invalid_every=0)


def test_fused_step_matches_autograd_path_2x64_colorplanes():
    pc.check_fused_vs_autograd(DEV, hidden=64, one_grid=False, co=True, iters=2)


def test_fp16_planes_autograd_path_matches_fused_path():
    """Half-precision plane storage on the drop-in (autograd) path: fp32 gradient sums reach the optimizer (``grad32``)."""
    pc.check_fused_vs_autograd(DEV, hidden=32, one_grid=True, co=False, plane_dtype="fp16")


@pytest.mark.parametrize("one_grid", [True, False])
def test_random_scene_2x64_vs_oracle(one_grid):
    """2x64 decoders (ALDS / global A tables, fused 2x64 weight-gradient kernel) against the oracle's autograd"""
    pc.check_oracle_random_scene(DEV, hidden=64, one_grid=one_grid, n_rays=12, S_d=20, S_r=9)


def _tiny_bench_cfg(one_grid=True, hidden=32):
    from mneslam_amd import configs
    cfg = configs.bench_office0(n_range_d=9, n_samples_d=20, hidden=hidden)
    cfg["mapping"]["bound"] = [[-1.0, 1.0], [-1.2, 1.1], [-0.8, 0.9]]
    cfg["planes_res"] = {"coarse": 0.2, "fine": 0.1, "bound_dividable": 0.2}
    cfg["c_planes_res"] = {"coarse": 0.4, "fine": 0.2}
    cfg["grid"]["oneGrid"] = one_grid
    cfg["mapping"]["sample"], cfg["mapping"]["min_pixels_cur"] = 40, 8
    cfg["cam"]["far"] = 4.0
    return cfg


def test_bench_path_step_vs_oracle():
    """The bench path (device sampler + Philox jitter + FusedStep) vs one oracle iteration on the same batch."""
    out = pc.check_fused_step_vs_oracle(DEV, _tiny_bench_cfg(), n_keyframes=3, seed=2, small=True, impl="explicit")
    assert out["contributing"] > 0


def test_hash_grid_fused_step_vs_oracle():
    """NS-a: the hash-grid mapping iteration (hash gather -> external-feature render -> atomic scatter -> Adam over the table)
    against the build's own oracle; bit-exact table indices; tiny sizes for the emulator."""
    cfg = pc.hash_test_config(hash_size=9, hidden=32, desired_resolution=64)
    cfg["training"]["n_range_d"], cfg["training"]["n_samples_d"] = 9, 20
    cfg["mapping"]["bound"] = [[-1.0, 1.0], [-1.2, 1.1], [-0.8, 0.9]]
    cfg["mapping"]["sample"], cfg["mapping"]["min_pixels_cur"] = 24, 8
    cfg["cam"]["far"] = 4.0
    out = pc.check_hash_fused_step_vs_oracle(DEV, cfg, n_keyframes=3, seed=2, small=True)
    assert out["touched_entries"] > 0


def test_hash_table_update_is_bit_reproducible():
    """NS-a: exact integer sums -> the table update gives the same bits whatever the arrival order of its rows"""
    cfg = pc.hash_test_config(hash_size=9, hidden=32, desired_resolution=64)
    cfg["training"]["n_range_d"], cfg["training"]["n_samples_d"] = 9, 20
    cfg["mapping"]["bound"] = [[-1.0, 1.0], [-1.2, 1.1], [-0.8, 0.9]]
    cfg["mapping"]["sample"], cfg["mapping"]["min_pixels_cur"] = 24, 8
    cfg["cam"]["far"] = 4.0
    out = pc.check_hash_update_bit_reproducible(DEV, cfg, n_keyframes=3, seed=2, warm_steps=1, small=True)
    assert out["moved"] > 0


def test_hash_scene_api_vs_oracle():
    """NS-a: render_rays / forward + backward / render_maps / render_img / query_* of the hash-grid scene model"""
    cfg = pc.hash_test_config(hash_size=9, hidden=32, desired_resolution=64)
    cfg["training"]["n_range_d"], cfg["training"]["n_samples_d"], cfg["training"]["n_samples"] = 9, 20, 24
    cfg["mapping"]["bound"] = [[-1.0, 1.0], [-1.2, 1.1], [-0.8, 0.9]]
    cfg["cam"]["far"] = 4.0
    pc.check_hash_scene_api(DEV, cfg, n_rays=12, img=(6, 10))


def test_dense_grid_fused_step_vs_oracle():
    """BASELINE configs[0] as north-star: 16^3 dense grid + 2x32 (tiny batch for the emulator)"""
    cfg = pc.dense_grid_config()
    cfg["training"]["n_range_d"], cfg["training"]["n_samples_d"] = 9, 20
    cfg["mapping"]["bound"] = [[-1.0, 1.0], [-1.2, 1.1], [-0.8, 0.9]]
    cfg["mapping"]["sample"], cfg["mapping"]["min_pixels_cur"] = 24, 8
    cfg["cam"]["far"] = 4.0
    out = pc.check_hash_fused_step_vs_oracle(DEV, cfg, n_keyframes=3, seed=2, small=True)
    assert out["touched_entries"] > 0


@pytest.mark.parametrize("one_grid,warm", [(True, 0), (False, 2)])
def test_bench_path_step_fp16_plane_storage_vs_oracle(one_grid, warm):
    """BASELINE configs[4] "fp16 features + fp32 accumulate" (EXTENSION): grid.plane_dtype 'fp16' stores the planes ONLY in
    half precision; gather / inline gather convert on load, the plane update computes Adam on float(p16) with fp32 moments
    and an fp32 gradient sum and stores the rounded result.  Oracle: the same values in fp32 tensors, parameters rounded to
    fp16 after its Adam step."""
    cfg = _tiny_bench_cfg(one_grid=one_grid)
    cfg["grid"]["plane_dtype"] = "fp16"
    out = pc.check_fused_step_vs_oracle(DEV, cfg, n_keyframes=3, seed=2, small=True, impl="explicit", warm_steps=warm)
    assert out["contributing"] > 0


def test_bench_path_step_fp16_planes_atomics_scatter_vs_oracle():
    """The same storage on the atomics schedule: fp32 gradient buffers, mne_adam_step with p_f16 segments."""
    cfg = _tiny_bench_cfg()
    cfg["grid"]["plane_dtype"] = "fp16"
    out = pc.check_fused_step_vs_oracle(DEV, cfg, n_keyframes=3, seed=2, small=True, impl="explicit", scatter="atomics")
    assert out["contributing"] > 0


def test_bench_path_step_external_bin_with_many_deferred_rays(monkeypatch):
    """the resolved rays' appends as a call of their own (mne_tile_bin pass 0, mne_fused_opts_t.external_bin) on an
    UNTRAINED map pinned to the prefix schedule (MNE_NO_ADAPT): most rays are unresolved and go through the deferred pass; pass
    0 -- here run after the whole render call, i.e. after the deferred decode and its appends -- must append exactly the rays
    the a-priori prefix resolved: a ray appended twice or not at all shows up in the plane gradients vs the oracle."""
    monkeypatch.setenv("MNE_FORCE_EXTERNAL_BIN", "1")
    monkeypatch.setenv("MNE_NO_ADAPT", "1")
    cfg = _tiny_bench_cfg()
    cfg["training"]["n_samples_d"] = 88            # 97 samples = 4 tiles per ray: the a-priori prefix is shorter than the ray
    out = pc.check_fused_step_vs_oracle(DEV, cfg, n_keyframes=3, seed=2, small=True, impl="explicit")
    assert out["contributing"] > 0 and out["deferred_rays"] > 0


@pytest.mark.parametrize("hw", [(1500, 1600), (2300, 2400)])      # 6 planes x ~9.4 k tiles (LDS snapshot) / ~21.6 k tiles each > 20480 (re-read)
def test_tile_order_with_more_tiles_than_registers(hw, monkeypatch):
    """tile_order_kernel keeps the first 8192 list lengths in registers and the next 12288 in LDS (every length is
    read from memory once); beyond MNE_TILE_ORDER_SNAPSHOT tiles it re-reads.  Synthetic lengths with split lists forced
    (MNE_TILE_SPLIT_MIN): the order must hold every split list's parts (consecutive, first) and every other tile exactly
    once, heaviest length bucket first."""
    import ctypes as C
    monkeypatch.setenv("MNE_TILE_SPLIT_MIN", "64")
    lib = _lib.load()
    h, w = hw
    sc = _lib.Scene()
    sc.n_sets, sc.c_dim, sc.hidden, sc.hidden_color, sc.geo_feat_dim, sc.n_bins = 1, 32, 32, 32, 15, 16
    dummy = torch.zeros(16)
    for o in range(3):
        for l in range(2):
            pl = sc.plane[0][o][l]
            pl.data, pl.h, pl.w = dummy.data_ptr(), (h if l else 40), (w if l else 33)
    sc.w_sdf0 = sc.w_sdf1 = sc.w_col0 = sc.w_col1 = dummy.data_ptr()
    n_tiles = lib.mne_tile_count(C.byref(sc))
    assert n_tiles > 8192
    gen = torch.Generator().manual_seed(h)
    counts = torch.randint(0, 40, (n_tiles,), generator=gen, dtype=torch.int32)
    heavy = torch.randperm(n_tiles, generator=gen)[:37]
    counts[heavy] = torch.randint(200, 3000, (37,), generator=gen, dtype=torch.int32)
    counts[n_tiles - 1] = 2500                                   # a heavy list among the tiles beyond the registers / the snapshot
    order = torch.full((n_tiles + _lib.TILE_SPLIT_PARTS,), -1, dtype=torch.int32)
    split_scratch = torch.zeros(4)                               # (only its presence matters to tile_order)
    split_state = torch.zeros(n_tiles + 1, dtype=torch.int32)
    prev = torch.zeros(n_tiles, dtype=torch.int32)
    # a list's length is the sum of its segments' cursors (one per XCD): spread every length unevenly over the eight
    seg = torch.zeros(n_tiles, _lib.LIST_SEGMENTS, dtype=torch.int32)
    rest = counts.clone()
    for x in range(_lib.LIST_SEGMENTS - 1):
        part = (rest.float() * torch.rand(n_tiles, generator=gen) * 0.4).to(torch.int32)
        seg[:, x], rest = part, rest - part
    seg[:, -1] = rest
    assert torch.equal(seg.sum(1).to(torch.int32), counts)
    seg = seg.contiguous()
    b = _lib.TileBins()
    b.counts, b.order, b.cap, b.spill_cap = seg.data_ptr(), order.data_ptr(), 4096, 0
    b.split_scratch, b.split_state, b.prev_counts = split_scratch.data_ptr(), split_state.data_ptr(), prev.data_ptr()
    _lib.check(lib.mne_tile_order(C.byref(sc), C.byref(b), None), "mne_tile_order")
    n_items = int(split_state[n_tiles])
    total = int(counts.sum())
    split = max(64, (2 * total + _lib.TILE_SPLIT_PARTS - 1) // _lib.TILE_SPLIT_PARTS)
    items = order[:n_items].tolist()
    seen, k = {}, 0
    while k < n_items and (items[k] >> 26) & 63:                 # split items first: tile | part << 20 | parts << 26
        t, part, parts = items[k] & 0xfffff, (items[k] >> 20) & 63, (items[k] >> 26) & 63
        assert part == 0 and int(counts[t]) > split and parts == min(63, -(-int(counts[t]) // split))
        for q in range(parts):
            assert items[k + q] == (t | (q << 20) | (parts << 26))
        seen[t] = parts
        k += parts
    whole = items[k:]
    assert all((x >> 20) == 0 for x in whole) and len(set(whole)) == len(whole) and not (set(whole) & set(seen))
    assert len(whole) + len(seen) == n_tiles
    bucket = [(int(counts[t]) + 1).bit_length() - 1 for t in whole]
    assert bucket == sorted(bucket, reverse=True), "heaviest length bucket first"
    assert all(int(counts[t]) <= split for t in whole) and sorted(seen) == sorted(t for t in range(n_tiles) if int(counts[t]) > split)


def test_bench_path_step_with_split_tile_lists(monkeypatch):
    """Long tile lists are cut into parts accumulated by several workgroups and combined by the last arriver
    (tile_adam.hip); forced here on a tiny scene with MNE_TILE_SPLIT_MIN."""
    monkeypatch.setenv("MNE_TILE_SPLIT_MIN", "4")
    out = pc.check_fused_step_vs_oracle(DEV, _tiny_bench_cfg(), n_keyframes=3, seed=2, small=True, impl="explicit")
    assert out["contributing"] > 0


def test_bench_path_step_with_capped_ray_lds(monkeypatch):
    """The training ray kernel's first pass keeps only MNE_HOT_LDS_SAMPLES samples of a ray in LDS (more waves per CU on
    long rays, INS Indoor: S = 1045); rays whose decoded prefix is longer are finished by the second pass.  Forced here
    on S = 29 with a cap of 16."""
    monkeypatch.setenv("MNE_HOT_LDS_SAMPLES", "16")
    out = pc.check_fused_step_vs_oracle(DEV, _tiny_bench_cfg(), n_keyframes=3, seed=2, small=True, impl="explicit")
    assert out["contributing"] > 0


@pytest.mark.parametrize("capped", [False, True])
def test_bench_path_step_with_heavy_ray_list(monkeypatch, capped):
    """Rays with many backward tiles leave them to heavy_bwd_kernel (tile-parallel; INS Indoor: 33 tiles per ray).  Forced here
    on S = 111 (4 tiles): every ray with more than ONE backward tile goes to the heavy list -- from the first ray pass, from the
    long-ray pass (capped: 48 samples of a ray in LDS) and from the deferred pass alike."""
    monkeypatch.setenv("MNE_HEAVY_NTILE", "1")
    monkeypatch.setenv("MNE_HEAVY_TILES", "1")
    if capped:
        monkeypatch.setenv("MNE_HOT_LDS_SAMPLES", "48")
    cfg = _tiny_bench_cfg()
    cfg["training"]["n_samples_d"], cfg["training"]["n_range_d"] = 100, 11
    out = pc.check_fused_step_vs_oracle(DEV, cfg, n_keyframes=3, seed=2, small=True, impl="explicit")
    assert out["contributing"] > 0


def test_sample_z_frame_sized_batch_counts():
    """whole-frame batches: the striped counts reduction of mne_sample_z"""
    pc.check_sample_z_frame_counts(DEV)
