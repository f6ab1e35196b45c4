"""ctypes binding of libmneslam_hip.so (C ABI: include/mneslam_hip.h).

The library is the only compute backend of this package.  If it is missing the import of any hot
path raises -- there is deliberately no CPU/PyTorch fallback.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmneslam_hip.so")

ABI_VERSION = 8
N_LOSS = 8
N_COUNT = 8
C_NEED = 6
RENDER_EARLY_TERMINATION = 1
L_RGB, L_DEPTH, L_CO_SDF, L_CO_FS, L_E_FS, L_E_CENTER, L_E_TAIL, L_PSNR = range(8)


class Plane(C.Structure):
    _fields_ = [("data", C.c_void_p), ("grad", C.c_void_p), ("h", C.c_int32), ("w", C.c_int32)]


class Scene(C.Structure):
    _fields_ = [("n_sets", C.c_int32), ("c_dim", C.c_int32), ("hidden", C.c_int32), ("hidden_color", C.c_int32),
                ("geo_feat_dim", C.c_int32), ("n_bins", C.c_int32), ("bb_is_f64", C.c_int32), ("plane_f16", C.c_int32),
                ("plane", Plane * 2 * 3 * 2),          # [set][orient][level]
                ("bound_lo", C.c_float * 3), ("bound_hi", C.c_float * 3),
                ("bb_lo", C.c_double * 3), ("bb_hi", C.c_double * 3),
                ("w_sdf0", C.c_void_p), ("w_sdf1", C.c_void_p), ("w_col0", C.c_void_p), ("w_col1", C.c_void_p)]


class RenderCfg(C.Structure):
    _fields_ = [("near_z", C.c_double), ("far_z", C.c_double), ("range_d", C.c_double), ("perturb", C.c_double),
                ("trunc", C.c_double), ("sc_factor", C.c_double), ("truncation", C.c_double),
                ("depth_trunc", C.c_double), ("n_samples", C.c_int32), ("n_samples_d", C.c_int32),
                ("n_range_d", C.c_int32), ("reserved", C.c_int32)]


class AdamSeg(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("weight_decay", C.c_double), ("step", C.c_int32), ("p_f16", C.c_int32)]


class TileBins(C.Structure):
    _fields_ = [("lists", C.c_void_p), ("counts", C.c_void_p), ("spill", C.c_void_p), ("spill_count", C.c_void_p),
                ("order", C.c_void_p), ("cap", C.c_int32), ("spill_cap", C.c_int32), ("dropped", C.c_void_p),
                ("split_scratch", C.c_void_p), ("split_state", C.c_void_p), ("prev_counts", C.c_void_p), ("live", C.c_void_p),
                ("plane_cap", C.c_int32 * 12)]


LIST_SEGMENTS = 8              # cursors per tile list (re-read from mne_tile_list_segments() at load time)
TILE_SPLIT_PARTS = 2048
TILE_ORDER_SNAPSHOT = 20480
MAX_OVERLAP_PEERS = 3


class OverlapRect(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32)]


class PoseState(C.Structure):
    _fields_ = [("rot", C.c_void_p), ("trans", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("step", C.c_void_p),
                ("c2w", C.c_void_p), ("best_loss", C.c_void_p), ("best_c2w", C.c_void_p), ("last_loss", C.c_void_p),
                ("r_base", C.c_float * 9), ("n_rot", C.c_int32),
                ("lr_rot", C.c_double), ("lr_trans", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double)]


class TileOverlap(C.Structure):
    _fields_ = [("n_peers", C.c_int32), ("reserved", C.c_int32), ("rect", (OverlapRect * 12) * MAX_OVERLAP_PEERS),
                ("send", C.c_void_p * MAX_OVERLAP_PEERS), ("recv", C.c_void_p * MAX_OVERLAP_PEERS)]


class PlaneOpt(C.Structure):
    _fields_ = [("m", C.c_void_p), ("v", C.c_void_p), ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("eps", C.c_double), ("weight_decay", C.c_double), ("step", C.c_int32), ("reserved", C.c_int32)]


class Clock(C.Structure):
    _fields_ = [("iteration", C.c_void_p), ("step_offset", C.c_void_p), ("bias_table", C.c_void_p), ("n_table", C.c_int32),
                ("reserved", C.c_int32), ("beta1", C.c_double), ("beta2", C.c_double), ("z_offset_stride", C.c_uint64)]


class FusedOpts(C.Structure):
    """mne_fused_opts_t: per-call extras of mne_render_fused / mne_render_fused_features."""
    _fields_ = [("timing_events", C.POINTER(C.c_void_p)), ("n_timing_events", C.c_int32), ("lds_samples_cap", C.c_int32),
                ("adapt_state", C.c_void_p), ("external_bin", C.c_int32), ("features_pregathered", C.c_int32),
                ("event_after_decode", C.c_void_p)]


class DecoderOpt(C.Structure):
    _fields_ = [("m", C.c_void_p * 4), ("v", C.c_void_p * 4), ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("eps", C.c_double), ("weight_decay", C.c_double), ("step", C.c_int32), ("reserved", C.c_int32)]


class GridCfg(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("n_features", C.c_int32), ("base_resolution", C.c_int32),
                ("log2_hashmap_size", C.c_int32), ("grid_type", C.c_int32), ("reserved", C.c_int32),
                ("per_level_scale", C.c_double)]


# Plane * 2 * 3 * 2 builds [2][3][2] read right-to-left: ((Plane*2)*3)*2 == plane[2][3][2]  (set, orient, level)

_PROTOS = {
    "mne_abi_version": (C.c_int, []),
    "mne_last_error": (C.c_char_p, []),
    "mne_sizeof_scene": (C.c_size_t, []),
    "mne_sizeof_render_cfg": (C.c_size_t, []),
    "mne_sizeof_adam_seg": (C.c_size_t, []),
    "mne_sizeof_tile_bins": (C.c_size_t, []),
    "mne_sizeof_plane_opt": (C.c_size_t, []),
    "mne_sizeof_clock": (C.c_size_t, []),
    "mne_sizeof_fused_opts": (C.c_size_t, []),
    "mne_sizeof_decoder_opt": (C.c_size_t, []),
    "mne_sample_batch": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64] + [C.c_void_p] * 5
                         + [C.POINTER(RenderCfg), C.c_void_p, C.c_void_p, C.c_uint64] + [C.c_void_p] * 5 + [C.POINTER(Clock), C.c_void_p]),
    "mne_decoder_update": (C.c_int, [C.POINTER(Scene), C.c_void_p, C.c_int, C.c_void_p, C.POINTER(DecoderOpt), C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Clock), C.c_void_p]),
    "mne_clock_advance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mne_num_samples": (C.c_int, [C.POINTER(RenderCfg), C.c_int]),
    "mne_sample_z": (C.c_int, [C.POINTER(RenderCfg), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                               C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Clock), C.c_void_p]),
    "mne_packed_decoder_floats": (C.c_size_t, [C.POINTER(Scene)]),
    "mne_pack_decoder": (C.c_int, [C.POINTER(Scene), C.c_void_p, C.c_void_p]),
    "mne_render_forward": (C.c_int, [C.POINTER(Scene), C.POINTER(RenderCfg), C.c_int, C.c_int] + [C.c_void_p] * 14
                           + [C.c_int, C.c_void_p]),
    "mne_render_forward_features": (C.c_int, [C.POINTER(Scene), C.POINTER(RenderCfg), C.c_int, C.c_int] + [C.c_void_p] * 15
                                    + [C.c_int, C.c_void_p]),
    "mne_render_backward_features": (C.c_int, [C.POINTER(Scene), C.POINTER(RenderCfg), C.c_int, C.c_int] + [C.c_void_p] * 12
                                     + [C.c_int64] + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]),
    "mne_hash_ray_grad": (C.c_int, [C.POINTER(GridCfg), C.POINTER(Scene), C.c_int, C.c_int] + [C.c_void_p] * 9),
    "mne_query_features": (C.c_int, [C.POINTER(Scene), C.c_int64] + [C.c_void_p] * 6),
    "mne_grid_encode_box": (C.c_int, [C.POINTER(GridCfg), C.POINTER(Scene), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p]),
    "mne_hash_features": (C.c_int, [C.POINTER(GridCfg), C.POINTER(Scene), C.c_int, C.c_int] + [C.c_void_p] * 6),
    "mne_loss_finalize": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mne_loss_coef": (C.c_int, [C.POINTER(RenderCfg), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mne_tape_row_floats": (C.c_size_t, [C.POINTER(Scene)]),
    "mne_tape_dfeat_offset": (C.c_size_t, [C.POINTER(Scene)]),
    "mne_render_backward": (C.c_int, [C.POINTER(Scene), C.POINTER(RenderCfg), C.c_int, C.c_int] + [C.c_void_p] * 12
                            + [C.c_int64] + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]),
    "mne_render_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "mne_render_fused": (C.c_int, [C.POINTER(Scene), C.POINTER(RenderCfg), C.c_int, C.c_int] + [C.c_void_p] * 13
                         + [C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(TileBins), C.c_void_p, C.c_size_t, C.POINTER(FusedOpts),
                            C.c_void_p]),
    "mne_tile_bin": (C.c_int, [C.POINTER(Scene), C.POINTER(RenderCfg), C.c_int, C.c_int] + [C.c_void_p] * 7
                     + [C.POINTER(TileBins), C.c_void_p, C.c_size_t, C.c_int, C.POINTER(FusedOpts), C.c_void_p]),
    "mne_tile_count": (C.c_size_t, [C.POINTER(Scene)]),
    "mne_tile_list_segments": (C.c_int, []),
    "mne_tile_list_entries": (C.c_size_t, [C.POINTER(Scene), C.POINTER(TileBins)]),
    "mne_sizeof_tile_overlap": (C.c_size_t, []),
    "mne_sizeof_pose_state": (C.c_size_t, []),
    "mne_pose_rays": (C.c_int, [C.POINTER(PoseState), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mne_pose_loss": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mne_pose_update": (C.c_int, [C.POINTER(PoseState), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mne_tile_overlap_floats": (C.c_size_t, [C.POINTER(Scene), C.POINTER(TileOverlap), C.c_int]),
    "mne_tile_grad_export": (C.c_int, [C.POINTER(Scene), C.c_void_p, C.POINTER(TileBins), C.POINTER(TileOverlap), C.c_void_p]),
    "mne_tile_adam_shared": (C.c_int, [C.POINTER(Scene), C.POINTER(PlaneOpt), C.c_void_p, C.POINTER(TileBins),
                                       C.POINTER(TileOverlap), C.POINTER(Clock), C.c_void_p]),
    "mne_tile_order": (C.c_int, [C.POINTER(Scene), C.POINTER(TileBins), C.c_void_p]),
    "mne_tile_adam": (C.c_int, [C.POINTER(Scene), C.POINTER(PlaneOpt), C.c_void_p, C.POINTER(TileBins), C.POINTER(Clock),
                                C.c_void_p]),
    "mne_sample_rays": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int,
                                  C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64] + [C.c_void_p] * 5
                        + [C.POINTER(Clock), C.c_void_p]),
    "mne_decoder_param_floats": (C.c_size_t, [C.POINTER(Scene)]),
    "mne_wgrad_partial_floats": (C.c_size_t, [C.POINTER(Scene)]),
    "mne_decoder_wgrad": (C.c_int, [C.POINTER(Scene), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_void_p]),
    "mne_adam_step": (C.c_int, [C.POINTER(AdamSeg), C.c_int, C.c_int, C.POINTER(Clock), C.c_void_p]),
    "mne_query_points": (C.c_int, [C.POINTER(Scene), C.c_int64] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p]),
    "mne_grid_level_table": (C.c_int, [C.POINTER(GridCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mne_grid_param_count": (C.c_size_t, [C.POINTER(GridCfg)]),
    "mne_grid_encode": (C.c_int, [C.POINTER(GridCfg), C.c_int64] + [C.c_void_p] * 5),
    "mne_grid_encode_backward": (C.c_int, [C.POINTER(GridCfg), C.c_int64] + [C.c_void_p] * 4),
    "mne_encode_oneblob": (C.c_int, [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mne_encode_frequency": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mne_encode_frequency_backward": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mne_encode_sh": (C.c_int, [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mne_encode_sh_backward": (C.c_int, [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mne_encode_identity": (C.c_int, [C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mne_hash_gather": (C.c_int, [C.POINTER(GridCfg), C.POINTER(Scene), C.c_int, C.c_int] + [C.c_void_p] * 7),
    "mne_render_fused_features": (C.c_int, [C.POINTER(Scene), C.POINTER(RenderCfg), C.c_int, C.c_int] + [C.c_void_p] * 13
                                  + [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(GridCfg), C.c_void_p,
                                     C.POINTER(FusedOpts), C.c_void_p]),
    "mne_hash_workspace_bytes": (C.c_size_t, [C.POINTER(GridCfg), C.c_int, C.c_int]),
    "mne_hash_slice_adam": (C.c_int, [C.POINTER(GridCfg), C.POINTER(Scene), C.c_int, C.c_int] + [C.c_void_p] * 6
                            + [C.POINTER(PlaneOpt), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "mne_hash_scatter": (C.c_int, [C.POINTER(GridCfg), C.POINTER(Scene), C.c_int, C.c_int] + [C.c_void_p] * 7),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)

_lock = threading.Lock()
_lib = None


def load(path=None):
    """Load (once) and return the library.  ``path`` overrides the in-tree location; the test-suite
    uses it to inject its host-emulation build -- the package itself never looks anywhere else."""
    global _lib
    with _lock:
        if _lib is not None and path is None:
            return _lib
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise RuntimeError(
                f"{p} not found: the HIP library is the only backend of mneslam_amd. "
                "Build it with `python -m mneslam_amd.build` (needs hipcc; cross-compiles gfx950 without a GPU).")
        lib = C.CDLL(p)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(lib, name)          # AttributeError = the library does not export the ABI
            fn.restype, fn.argtypes = res, args
        if lib.mne_abi_version() != ABI_VERSION:
            raise RuntimeError("libmneslam_hip ABI version mismatch")
        global LIST_SEGMENTS
        LIST_SEGMENTS = int(lib.mne_tile_list_segments())      # (8 in the shipped build; experiment builds may differ)
        for fn, st in ((lib.mne_sizeof_scene, Scene), (lib.mne_sizeof_render_cfg, RenderCfg),
                       (lib.mne_sizeof_adam_seg, AdamSeg), (lib.mne_sizeof_tile_bins, TileBins),
                       (lib.mne_sizeof_plane_opt, PlaneOpt), (lib.mne_sizeof_clock, Clock),
                       (lib.mne_sizeof_fused_opts, FusedOpts), (lib.mne_sizeof_decoder_opt, DecoderOpt),
                       (lib.mne_sizeof_tile_overlap, TileOverlap), (lib.mne_sizeof_pose_state, PoseState)):
            if fn() != C.sizeof(st):
                raise RuntimeError(f"struct layout mismatch for {st.__name__}: C {fn()} vs ctypes {C.sizeof(st)}")
        _lib = lib
        return lib


def unload():
    global _lib
    with _lock:
        _lib = None


def check(rc, what=""):
    if rc != 0:
        msg = load().mne_last_error()
        raise RuntimeError(f"libmneslam_hip {what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device (or host, under the test emulator) address of a tensor, or None."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_for(t):
    """The caller's current HIP stream (kernels must run on the mapping
    thread's current stream)."""
    if t.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None
