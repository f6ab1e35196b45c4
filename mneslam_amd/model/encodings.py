"""``get_encoder`` -- the reference's encoder factory (model/encodings.py:6-97) without tinycudann.

Only OneBlob is reachable in the reference (model/scene_rep.py:157; the hash-grid call at :160 is
commented out).  The module returned here has tinycudann's surface (``n_output_dims``, a zero-size
``params`` Parameter -> state_dict key ``embedpos_fn.params``) and runs the stand-alone HIP kernel.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import _lib


class OneBlobEncoding(nn.Module):
    """tcnn.Encoding(otype="OneBlob") replacement; spec in oracle/oneblob.py (parity unpinned:
    tinycudann is not part of the reference tree)."""

    def __init__(self, n_input_dims=3, n_bins=16):
        super().__init__()
        if n_bins != 16:
            raise NotImplementedError("the HIP OneBlob kernel is built for pos.n_bins == 16 (every shipped config)")
        self.n_input_dims, self.n_bins = n_input_dims, n_bins
        self.n_output_dims = n_input_dims * n_bins
        self.params = nn.Parameter(torch.zeros(0, dtype=torch.float32))

    def forward(self, x):
        lib = _lib.load()
        x = x.detach().to(torch.float32).contiguous()          # tinycudann casts its input to fp32
        n, d = x.shape
        out = torch.empty(n, d * self.n_bins, device=x.device, dtype=torch.float32)
        _lib.check(lib.mne_encode_oneblob(n, d, _lib.ptr(x), _lib.ptr(out), _lib.stream_for(x)), "mne_encode_oneblob")
        return out


class _GridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, cfg):
        lib = _lib.load()
        x = x.detach().to(torch.float32).contiguous()
        n = x.shape[0]
        out = torch.empty(n, cfg.n_levels * cfg.n_features, device=x.device, dtype=torch.float32)
        _lib.check(lib.mne_grid_encode(C.byref(cfg), n, _lib.ptr(x), _lib.ptr(params.detach()), _lib.ptr(out), None,
                                       _lib.stream_for(x)), "mne_grid_encode")
        ctx.save_for_backward(x)
        ctx.cfg, ctx.n_params = cfg, params.numel()
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        dparams = torch.zeros(ctx.n_params, device=x.device, dtype=torch.float32)
        dout = dout.to(torch.float32).contiguous()
        _lib.check(lib.mne_grid_encode_backward(C.byref(ctx.cfg), x.shape[0], _lib.ptr(x), _lib.ptr(dout),
                                                _lib.ptr(dparams), _lib.stream_for(x)), "mne_grid_encode_backward")
        return None, dparams, None


class GridEncoding(nn.Module):
    """tcnn.Encoding(otype="HashGrid" | "Grid"/"Dense") replacement: one flat fp32 ``params`` vector
    (all levels), U(-1e-4, 1e-4) init, trilinear interpolation; spec in oracle/hashgrid.py."""

    def __init__(self, n_input_dims=3, n_levels=16, n_features_per_level=2, base_resolution=16,
                 per_level_scale=2.0, log2_hashmap_size=19, grid_type="hash"):
        super().__init__()
        if n_input_dims != 3:
            raise NotImplementedError("the grid encoding is built for 3-D inputs")
        cfg = _lib.GridCfg()
        cfg.n_levels, cfg.n_features, cfg.base_resolution = n_levels, n_features_per_level, base_resolution
        cfg.log2_hashmap_size, cfg.grid_type = log2_hashmap_size, 0 if grid_type == "hash" else 1
        cfg.per_level_scale = float(per_level_scale)
        self.cfg = cfg
        self.n_input_dims, self.n_output_dims = 3, n_levels * n_features_per_level
        n = _lib.load().mne_grid_param_count(C.byref(cfg))
        if n == 0:
            raise ValueError("bad grid encoding configuration")
        self.params = nn.Parameter((torch.rand(n) * 2 - 1) * 1e-4)

    def level_table(self):
        """(scales fp32, resolutions, sizes, offsets) per level, as the kernels use them."""
        L = self.cfg.n_levels
        sc, rs, sz, of = (C.c_float * L)(), (C.c_uint32 * L)(), (C.c_uint32 * L)(), (C.c_uint32 * L)()
        _lib.check(_lib.load().mne_grid_level_table(C.byref(self.cfg), sc, rs, sz, of), "mne_grid_level_table")
        return list(sc), list(rs), list(sz), list(of)

    def indices(self, x):
        """[N, n_levels, 8] uint32 (as int64) table indices within each level."""
        lib = _lib.load()
        x = x.detach().to(torch.float32).contiguous()
        n = x.shape[0]
        out = torch.empty(n, self.n_output_dims, device=x.device, dtype=torch.float32)
        idx = torch.empty(n, self.cfg.n_levels, 8, device=x.device, dtype=torch.int32)
        _lib.check(lib.mne_grid_encode(C.byref(self.cfg), n, _lib.ptr(x), _lib.ptr(self.params.detach()), _lib.ptr(out),
                                       _lib.ptr(idx), _lib.stream_for(x)), "mne_grid_encode")
        return idx.to(torch.int64) & 0xFFFFFFFF

    def forward(self, x):
        return _GridFn.apply(x, self.params, self.cfg)


def get_encoder(encoding, input_dim=3, degree=4, n_bins=16, n_frequencies=12, n_levels=16, level_dim=2,
                base_resolution=16, log2_hashmap_size=19, desired_resolution=512):
    """Same signature and return value ``(module, out_dim)`` as the reference factory."""
    name = encoding.lower()
    if "dense" in name:                                   # model/encodings.py:13-28 (n_levels forced to 4)
        n_levels = 4
        pls = float(np.exp2(np.log2(desired_resolution / base_resolution) / (n_levels - 1)))
        embed = GridEncoding(input_dim, n_levels, level_dim, base_resolution, pls, log2_hashmap_size, "dense")
        return embed, embed.n_output_dims
    if "hash" in name or "tiled" in name:                 # model/encodings.py:31-46
        pls = float(np.exp2(np.log2(desired_resolution / base_resolution) / (n_levels - 1)))
        embed = GridEncoding(input_dim, n_levels, level_dim, base_resolution, pls, log2_hashmap_size, "hash")
        return embed, embed.n_output_dims
    if "blob" in name:                                    # model/encodings.py:61-71
        embed = OneBlobEncoding(input_dim, n_bins)
        return embed, embed.n_output_dims
    raise NotImplementedError(
        f"encoding '{encoding}': the spherical-harmonics / frequency / identity branches of the reference factory "
        "are never reached by the mapping path (model/scene_rep.py:157 requests OneBlob) and are not provided")
