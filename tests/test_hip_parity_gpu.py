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
