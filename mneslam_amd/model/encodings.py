"""``get_encoder`` -- the reference's encoder factory (model/encodings.py:6-97) without tinycudann.

Only OneBlob is reachable in the reference (model/scene_rep.py:157; the hash-grid call at :160 is
commented out); every branch of the factory is provided (dense / hash grid, OneBlob, and -- round 6 -- spherical
harmonics, frequency, identity).  The module returned here has tinycudann's surface (``n_output_dims``, a zero-size
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


class _FrequencyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_frequencies):
        lib = _lib.load()
        xc = x.detach().to(torch.float32).contiguous()
        n, d = xc.shape
        out = torch.empty(n, d * 2 * n_frequencies, device=xc.device, dtype=torch.float32)
        _lib.check(lib.mne_encode_frequency(n, d, n_frequencies, _lib.ptr(xc), _lib.ptr(out), _lib.stream_for(xc)), "mne_encode_frequency")
        ctx.save_for_backward(xc)
        ctx.F, ctx.dtype = n_frequencies, x.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        (xc,) = ctx.saved_tensors
        dout = dout.to(torch.float32).contiguous()
        dx = torch.empty_like(xc)
        _lib.check(_lib.load().mne_encode_frequency_backward(xc.shape[0], xc.shape[1], ctx.F, _lib.ptr(xc), _lib.ptr(dout),
                                                             _lib.ptr(dx), _lib.stream_for(xc)), "mne_encode_frequency_backward")
        return dx.to(ctx.dtype), None


class FrequencyEncoding(nn.Module):
    """tcnn.Encoding(otype="Frequency", n_frequencies=F) replacement (model/encodings.py:73-84); differentiable with respect to
    its input; spec in oracle/encodings_misc.py (parity unpinned: tinycudann is not part of the reference tree)."""

    def __init__(self, n_input_dims=3, n_frequencies=12):
        super().__init__()
        self.n_input_dims, self.n_frequencies = n_input_dims, n_frequencies
        self.n_output_dims = n_input_dims * 2 * n_frequencies
        self.params = nn.Parameter(torch.zeros(0, dtype=torch.float32))

    def forward(self, x):
        return _FrequencyFn.apply(x, self.n_frequencies)


class _ShFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, degree):
        lib = _lib.load()
        xc = x.detach().to(torch.float32).contiguous()
        n = xc.shape[0]
        out = torch.empty(n, degree * degree, device=xc.device, dtype=torch.float32)
        _lib.check(lib.mne_encode_sh(n, degree, _lib.ptr(xc), _lib.ptr(out), _lib.stream_for(xc)), "mne_encode_sh")
        ctx.save_for_backward(xc)
        ctx.degree, ctx.dtype = degree, x.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        (xc,) = ctx.saved_tensors
        dout = dout.to(torch.float32).contiguous()
        dx = torch.empty_like(xc)
        _lib.check(_lib.load().mne_encode_sh_backward(xc.shape[0], ctx.degree, _lib.ptr(xc), _lib.ptr(dout), _lib.ptr(dx),
                                                      _lib.stream_for(xc)), "mne_encode_sh_backward")
        return dx.to(ctx.dtype), None


class SphericalHarmonicsEncoding(nn.Module):
    """tcnn.Encoding(otype="SphericalHarmonics", degree=d) replacement (model/encodings.py:48-58): inputs in [0, 1]^3 are
    directions 2 x - 1, outputs the d^2 real SH coefficients; degrees 1..4 (the reference's factory default is 4)."""

    def __init__(self, n_input_dims=3, degree=4):
        super().__init__()
        if n_input_dims != 3:
            raise ValueError("spherical harmonics encode 3-D directions")
        if not 1 <= degree <= 4:
            raise NotImplementedError("the HIP spherical-harmonics kernel covers degrees 1..4 (tinycudann: up to 8)")
        self.n_input_dims, self.degree, self.n_output_dims = 3, degree, degree * degree
        self.params = nn.Parameter(torch.zeros(0, dtype=torch.float32))

    def forward(self, x):
        return _ShFn.apply(x, self.degree)


class _IdentityFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, offset):
        lib = _lib.load()
        xc = x.detach().to(torch.float32).contiguous()
        out = torch.empty_like(xc)
        _lib.check(lib.mne_encode_identity(xc.numel(), scale, offset, _lib.ptr(xc), _lib.ptr(out), _lib.stream_for(xc)), "mne_encode_identity")
        ctx.scale, ctx.dtype = scale, x.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        return (dout * ctx.scale).to(ctx.dtype), None, None


class IdentityEncoding(nn.Module):
    """tcnn.Encoding(otype="Identity") replacement (model/encodings.py:86-95): out = x * scale + offset (defaults 1, 0)."""

    def __init__(self, n_input_dims=3, scale=1.0, offset=0.0):
        super().__init__()
        self.n_input_dims = self.n_output_dims = n_input_dims
        self.scale, self.offset = float(scale), float(offset)
        self.params = nn.Parameter(torch.zeros(0, dtype=torch.float32))

    def forward(self, x):
        return _IdentityFn.apply(x, self.scale, self.offset)


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
    # the branches below are never reached by the reference's mapping path (model/scene_rep.py:157 requests OneBlob); they
    # complete the factory's surface
    if "spherical" in name:                               # model/encodings.py:48-58
        embed = SphericalHarmonicsEncoding(input_dim, degree)
        return embed, embed.n_output_dims
    if "freq" in name:                                    # model/encodings.py:73-84
        embed = FrequencyEncoding(input_dim, n_frequencies)
        return embed, embed.n_output_dims
    if "identity" in name:                                # model/encodings.py:86-95
        embed = IdentityEncoding(input_dim)
        return embed, embed.n_output_dims
    raise ValueError(f"encoding '{encoding}': not one of the reference factory's names (model/encodings.py:6-97)")
