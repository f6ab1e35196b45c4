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
    assert _lib.load().mne_abi_version() == 1


def test_oneblob():
    pc.check_oneblob(DEV)


def test_adam():
    pc.check_adam(DEV)


@pytest.mark.parametrize("name", list(pc.FWD_CASES))
def test_forward(name):
    pc.check_forward(name, DEV)


@pytest.mark.parametrize("name", list(pc.FWD_CASES))
@pytest.mark.parametrize("co", [False, True])
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
