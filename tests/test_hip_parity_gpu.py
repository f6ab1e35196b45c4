"""Parity of the HIP path on a real MI355X (through the C ABI of libmneslam_hip.so) against the
golden vectors captured from the reference and against the CPU oracle.  Run by the driver with
``-m gpu``; bodies shared with the host-emulator run live in tests/parity_cases.py."""
import os

import pytest
import torch

import parity_cases as pc
from mneslam_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def real_library():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    _lib.unload()
    lib = _lib.load()                       # in-tree libmneslam_hip.so only; raises if missing
    assert os.path.samefile(lib._name, _lib.LIB_PATH)
    yield
    torch.cuda.synchronize()


def test_oneblob():
    pc.check_oneblob(DEV)


def test_adam():
    pc.check_adam(DEV)


@pytest.mark.parametrize("name", list(pc.FWD_CASES))
def test_forward_matches_reference(name):
    pc.check_forward(name, DEV)


@pytest.mark.parametrize("name", list(pc.FWD_CASES))
@pytest.mark.parametrize("co", [False, True])
def test_gradients_match_reference(name, co):
    pc.check_backward(name, co, DEV)


def test_mfma_wgrad_matches_scalar_crosscheck():
    pc.check_backward("fwd_onegrid", False, DEV, wgrad_impl=1)


def test_all_invalid_depth_nan_losses():
    pc.check_all_invalid(DEV)


def test_render_without_depth():
    pc.check_render_nodepth(DEV)


def test_point_queries():
    pc.check_queries(DEV)


@pytest.mark.parametrize("name,one_grid,co,seed", [("mapping3_onegrid_esdf", True, False, 21),
                                                   ("mapping3_colorplanes_cosdf", False, True, 22)])
def test_three_mapping_iterations_match_reference(name, one_grid, co, seed):
    pc.check_mapping3(name, one_grid, co, seed, DEV)


@pytest.mark.parametrize("name,co", [("fwd_onegrid", False), ("fwd_onegrid", True), ("fwd_colorplanes", False), ("fwd_colorplanes", True)])
def test_ray_gradients_match_reference(name, co):
    pc.check_ray_gradients(name, co, DEV)


def test_render_without_depth_pose_gradients():
    pc.check_render_nodepth_pose_gradients(DEV)


@pytest.mark.parametrize("kind", ["hash", "dense"])
def test_grid_encoding_surface(kind):
    pc.check_grid_encoding(DEV, kind)


def test_device_sampler():
    pc.check_device_sampler(DEV)


@pytest.mark.parametrize("name,one_grid,co,seed", [("mapping3_onegrid_esdf", True, False, 21),
                                                   ("mapping3_colorplanes_cosdf", False, True, 22)])
@pytest.mark.parametrize("scatter", ["binned", "atomics"])
def test_three_fused_mapping_iterations_match_reference(name, one_grid, co, seed, scatter):
    pc.check_mapping3(name, one_grid, co, seed, DEV, compute="fused", scatter=scatter)


def test_binned_scatter_with_list_overflow():
    pc.check_mapping3("mapping3_onegrid_esdf", True, False, 21, DEV, compute="fused", scatter="binned", tile_capacity=8)


@pytest.mark.parametrize("hidden,one_grid", [(64, True), (64, False), (32, True)])
def test_random_scene_vs_oracle(hidden, one_grid):
    pc.check_oracle_random_scene(DEV, hidden=hidden, one_grid=one_grid, n_rays=96, S_d=96, S_r=32)


def test_loop_closure_pose_alignment():
    pc.check_pose_alignment(DEV)


def test_loop_closure_distillation():
    pc.check_distillation(DEV)


def test_full_size_paths_agree_and_learn():
    """BASELINE-size workload (office0 planes 38.4 M params, 2150 rays x 128 samples): the fused path with
    binned scatter, the fused path with global atomics and the drop-in autograd path, driven with the
    SAME device-sampled batches and Philox jitter, must reach the same parameters after a few
    iterations (size-independent property: the three are different schedules of the same math), the
    loss must fall, and nothing may be NaN."""
    import bench
    from mneslam_amd import configs
    cfg = configs.bench_office0()
    dev = torch.device("cuda")
    finals, losses = {}, {}
    for mode in ("binned", "atomics", "binned+prefetch"):
        ag = bench.Agent(cfg, dev, seed=3, n_keyframes=4, path="fused", scatter=mode.split("+")[0])
        hist = []
        for it in range(6):
            # "+prefetch": the two HIP streams swap roles every iteration and the next batch is drawn early
            ag.step(prefetch=mode.endswith("prefetch") and it < 5)
            ag.fused.synchronize()
            hist.append(float(ag.fused.losses[0] + ag.fused.losses[1]))
        finals[mode] = [p.detach().clone() for lst in ag.model.all_planes for p in lst] + \
                       [p.detach().clone() for p in ag.model.decoder.parameters()]
        losses[mode] = hist
        del ag
        torch.cuda.empty_cache()
    # Adam with eps=1e-15 is scale-free: a cell whose gradient is at fp32-noise level takes a full lr-sized
    # step whose sign depends on the summation order, so a handful of elements may differ by O(lr);
    # everything else must agree to rounding.
    for a, b in zip(finals["binned"], finals["atomics"]):
        assert torch.isfinite(a).all()
        d = (a - b).abs()
        assert float(d.mean()) < 1e-7, "binned and atomic scatter disagree"
        assert float((d > 1e-4).float().mean()) < 1e-5 and float(d.max()) < 0.05
    for a, b in zip(finals["binned"], finals["binned+prefetch"]):
        d = (a - b).abs()
        assert float(d.mean()) < 1e-7 and float((d > 1e-4).float().mean()) < 1e-5, "prefetching / stream alternation changes the result"
    for x, y, z in zip(losses["binned"], losses["atomics"], losses["binned+prefetch"]):
        assert x == x and abs(x - y) <= 1e-3 * abs(y) and abs(x - z) <= 1e-3 * abs(x), "loss histories of the schedules diverge"
    # the mean absolute update is non-trivial (dense Adam moved the touched cells)
    assert float((finals["binned"][1] != 0).float().mean()) > 0.5
