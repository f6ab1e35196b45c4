"""The ISA checks the build runs on every kernel (mneslam_amd/isa_check.py): unit tests of the scanner on
hand-written assembly, and the report of the shipped build -- no kernel may carry the spill-in-front-of-exec-restore
defect, and the kernels of the mapping iteration must not use scratch memory at all."""
import json
import os

import pytest

from mneslam_amd import build, isa_check

BAD = """
_Z6kernelv:
	s_and_saveexec_b64 s[8:9], s[16:17]
	s_cbranch_execz .LBB0_2
; %bb.1:
	global_store_dword v[0:1], v2, off
.LBB0_2:
	scratch_store_dword off, v36, off offset:264 ; 4-byte Folded Spill
	s_or_b64 exec, exec, s[8:9]
	s_endpgm
"""
GOOD = BAD.replace("""	scratch_store_dword off, v36, off offset:264 ; 4-byte Folded Spill
	s_or_b64 exec, exec, s[8:9]""", """	s_or_b64 exec, exec, s[8:9]
	scratch_store_dword off, v36, off offset:264 ; 4-byte Folded Spill""")
NARROWING = """
_Z6kernelv:
.LBB0_1:
	scratch_store_dword off, v3, off offset:8 ; 4-byte Folded Spill
	s_and_b64 s[2:3], s[6:7], s[2:3]
	s_mov_b64 exec, s[2:3]
	s_cbranch_execz .LBB0_2
.LBB0_2:
	s_endpgm
"""


def test_scanner_flags_spill_before_exec_restore():
    hz = isa_check.definite_hazards(BAD)
    assert len(hz) == 1 and hz[0][0] == "_Z6kernelv" and hz[0][1][1] == ".LBB0_2"


def test_scanner_accepts_spill_after_exec_restore_and_before_narrowing():
    assert isa_check.definite_hazards(GOOD) == []
    findings, spills = isa_check.scan_asm(NARROWING)
    assert findings == {} and spills == {"_Z6kernelv": 1}


@pytest.mark.parametrize("variant", [None] + sorted(build.FUZZ_VARIANTS))
def test_shipped_build_is_clean(variant):
    lib = build.LIB if variant is None else build.variant_path(variant)
    path = build.isa_report_path(lib)
    if not os.path.exists(path):
        pytest.skip(f"{path}: library not built here (python -m mneslam_amd.build --fuzz)")
    rep = json.load(open(path))
    assert rep["hazards"] == [], rep["hazards"]
    names = {k["kernel"]: k for k in rep["kernels"]}
    assert len(names) > 60
    hot = [k for n, k in names.items() if any(t in n for t in ("gather_kernel", "decode_kernel", "bin_kernel", "tile_adam_kernel",
                                                                "tile_order_kernel", "wgrad_fused", "adam_kernel", "sample_z_kernel", "heavy_bwd_kernel",
                                                                "sample_rays", "decode_frame_kernel", "ray_frame_kernel"))
           or ("ray_kernel" in n and n.endswith("ELi4EEv10RenderArgs"))]
    assert len(hot) >= 21
    for k in hot:
        assert k["scratch"] == 0 and k.get("vgpr_spill", 0) == 0, f"{k['kernel']} uses {k['scratch']} B of scratch per lane"
    # the weight-gradient kernels share CUs with the plane update (tile_adam_kernel: 2 workgroups of 8 waves per CU): at
    # most 168 registers per lane and no AGPRs, or they push one of its workgroups out (330 vs 265 us)
    for n, k in names.items():
        if "wgrad_fused" in n:
            assert k.get("vgpr", 0) <= 168 and k.get("agpr", 0) == 0, f"{n}: {k.get('vgpr')} VGPRs + {k.get('agpr')} AGPRs"
    # EVERY other instantiated kernel: no scratch either, except the ones named here with the bytes per lane they are allowed
    # (the rule used to cover only the kernels named above).  Round 6 took the ray-gradient and autograd-backward
    # kernels (ray_kernel<..., 2 | 3>: 20-240 B per lane in round 5) off this list -- they no longer decode inside the backward
    # loop (tile_need_kernel + the decode launch make every tape row they walk).  What is left and why:
    #   ray_kernel<64,64,colour planes,.,0>  forward with on-demand decode of the largest decoder (render_maps of such a model)
    #   scatter_kernel          scatter="atomics" (cross-check schedule of the plane update), 16 B
    # Neither is launched by the fused mapping iteration or by render_img of the shipped configs; a spill in one of them is
    # still guarded against the compiler defect by the hazard scan above.
    allowed = {"_Z10ray_kernelILi64ELi64ELb1ELb0ELi0EEv10RenderArgs": 32,
               "_Z14scatter_kernelILb1EEv10RenderArgsi": 16, "_Z14scatter_kernelILb0EEv10RenderArgsi": 16}
    if variant is None:
        assert set(allowed) <= set(names), sorted(set(allowed) - set(names))          # (a stale list would hide nothing, but say so)
    for k in rep["kernels"]:
        cap = allowed.get(k["kernel"], 0) if variant is None else 64       # (the layout-fuzz builds reshuffle registers: a looser bound)
        assert k["scratch"] <= cap, f"{k['kernel']}: {k['scratch']} B of scratch per lane (allowed {cap})"
