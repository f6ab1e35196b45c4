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


def test_loop_closure_pose_alignment_on_the_hash_model():
    pc.check_pose_alignment_hash(DEV)


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
