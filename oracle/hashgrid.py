"""Multiresolution hash / dense grid encoding -- frozen spec (PARITY UNPINNED, see oracle/__init__).

Reference call sites: model/encodings.py:13-46 (``tcnn.Encoding(otype="Grid"/"HashGrid")``); never
executed by the reference (model/scene_rep.py:160 is commented out).  The arithmetic is tinycudann's
(not in the reference tree).  Restated from its published grid encoding:

    scale_l = 2^(l*log2(per_level_scale)) * base_resolution - 1        (fp32)
    res_l   = ceil(scale_l) + 1
    size_l  = min(next_multiple(res_l^3, 8), 2^log2_hashmap_size)      (Hash)  |  next_multiple(res_l^3, 8) (Dense)
    pos     = fma(scale_l, x, 0.5);  cell = floor(pos);  frac = pos - cell
    index(c)= (c.x + c.y*res + c.z*res^2) if the level is dense (res^3 <= size_l)
              else (c.x*1) ^ (c.y*2654435761) ^ (c.z*805459861)   [uint32],   then  % size_l
    out[l*F + f] = sum over the 8 corners of  prod_d (frac_d or 1-frac_d) * table[offset_l + index][f]
Parameters: one flat fp32 vector, level after level, F features per entry, init U(-1e-4, 1e-4).

Where each line comes from in tiny-cuda-nn (NVlabs/tiny-cuda-nn; unpinned in the reference, requirements.txt:120; fallback
commit 91ee479d275d322a65726435040fc20b56b9c991, README.md:99).  Recorded from the published source as known to the author of
this file -- tinycudann is NOT available in the build container, so a maintainer with a checkout must confirm it; the
helpers below sit in ``include/tiny-cuda-nn/encodings/grid.h`` in older commits and in
``include/tiny-cuda-nn/common_device.h`` in newer ones, their NAMES are stable:

    this file                            tiny-cuda-nn
    ----------------------------------   ---------------------------------------------------------------------------
    scale_l (level_table)                ``grid_scale(level, log2_per_level_scale, base_resolution)``:
                                           exp2f(level * log2_per_level_scale) * base_resolution - 1.0f
                                         (log2_per_level_scale = std::log2(per_level_scale), GridEncodingTemplated ctor)
    res_l                                ``grid_resolution(scale)``: (uint32_t)ceilf(scale) + 1
    size_l, offsets (level_table)        ``GridEncodingTemplated`` constructor: params_in_level = res^3 (capped against
                                           overflow); ``next_multiple(params_in_level, 8u)``; for GridType::Hash
                                           ``std::min(params_in_level, 1u << log2_hashmap_size)``; offsets = running sum
    pos / cell / frac (grid_indices)     ``pos_fract(input, &pos, &pos_derivative, &pos_grid, scale, identity_fun)``:
                                           *pos = fmaf(scale, input, 0.5f);  tmp = floorf(*pos);  *pos_grid = (uint32_t)(int)tmp;
                                           *pos -= tmp        (InterpolationType::Linear: weights frac / 1 - frac)
    dense index  c.x + c.y res + c.z     ``grid_index<N_POS_DIMS, HASH_TYPE>(grid_type, hashmap_size, grid_resolution, pos_grid)``:
      res^2,  chosen when res^3 <= size    stride loop ``index += pos_grid[dim] * stride; stride *= grid_resolution`` while
                                           ``stride <= hashmap_size``; ``if (grid_type == GridType::Hash && hashmap_size < stride)
                                           index = grid_hash<N_POS_DIMS, HASH_TYPE>(pos_grid)``
    hash  x*1 ^ y*2654435761 ^           ``coherent_prime_hash(pos_grid)`` (HashType::CoherentPrime, the default): primes
      z*805459861  (uint32)                {1, 2654435761, 805459861, 3674653429, ...}, ``result ^= pos_grid[i] * primes[i]``
    % size_l                             ``return index % hashmap_size``
    out[l*F + f] = sum_c w_c table[..]   ``kernel_grid``: loop ``for (idx = 0; idx < (1 << N_POS_DIMS); ++idx)``, weight *= pos[dim] or
                                           (1 - pos[dim]) by bit ``idx & (1 << dim)``, ``result[f] = fmaf(weight, val[f], result[f])``
    d/d table (autograd here)            ``kernel_grid_backward``: atomicAdd of weight * dL_dy per corner
    d/d x (autograd here; csrc/          ``kernel_grid`` with ``dy_dx`` / ``kernel_grid_backward_input``: d weight / d pos_d * scale
      gridenc.hip hash_raygrad_kernel)
    init U(-1e-4, 1e-4)                  ``GridEncodingTemplated::initialize_params``: generate_random_uniform(..., -1e-4f, 1e-4f)
"""
import math

import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861)


def level_table(n_levels, base_resolution, per_level_scale, log2_hashmap_size, grid_type="hash"):
    scales, ress, sizes, offsets = [], [], [], [0]
    l2 = np.float32(math.log2(per_level_scale))
    for l in range(n_levels):
        scale = np.float32(np.exp2(np.float32(l) * l2) * np.float32(base_resolution) - np.float32(1.0))
        res = int(math.ceil(float(scale))) + 1
        n = res ** 3
        n = (n + 7) // 8 * 8
        if grid_type == "hash":
            n = min(n, 1 << log2_hashmap_size)
        scales.append(float(scale)); ress.append(res); sizes.append(n); offsets.append(offsets[-1] + n)
    return scales, ress, sizes, offsets


def grid_indices(x, scale, res, size):
    """x [N,3] fp32 -> (idx [N,8] int64 in [0,size), w [N,8] fp32); corner c: bit d set = +1 along dim d."""
    pos = (x.double() * float(np.float32(scale)) + 0.5).float()              # fma emulated in fp64
    cell_f = torch.floor(pos)
    frac = pos - cell_f
    cell = cell_f.to(torch.int64) & 0xFFFFFFFF                               # uint32 wrap of (uint32_t)(int)
    idx, w = [], []
    dense = res ** 3 <= size
    for c in range(8):
        cc, ww = [], torch.ones(x.shape[0])
        for d in range(3):
            if (c >> d) & 1:
                cc.append((cell[:, d] + 1) & 0xFFFFFFFF); ww = ww * frac[:, d]
            else:
                cc.append(cell[:, d]); ww = ww * (1.0 - frac[:, d])
        if dense:
            i = (cc[0] + cc[1] * res + cc[2] * res * res) & 0xFFFFFFFF
        else:
            i = ((cc[0] * PRIMES[0]) & 0xFFFFFFFF) ^ ((cc[1] * PRIMES[1]) & 0xFFFFFFFF) ^ ((cc[2] * PRIMES[2]) & 0xFFFFFFFF)
        idx.append(i % size); w.append(ww)
    return torch.stack(idx, 1), torch.stack(w, 1)


def grid_encode(x, params, n_levels, n_features, base_resolution, per_level_scale, log2_hashmap_size, grid_type="hash",
                return_indices=False, scales=None):
    """x [N,3] in [0,1], params flat [total*F] -> [N, n_levels*F]; differentiable w.r.t. params.
    ``scales`` overrides the per-level fp32 scale constants (two libm exp2f may differ by one ulp)."""
    own_scales, ress, sizes, offsets = level_table(n_levels, base_resolution, per_level_scale, log2_hashmap_size, grid_type)
    scales = own_scales if scales is None else scales
    table = params.reshape(-1, n_features)
    outs, all_idx = [], []
    for l in range(n_levels):
        idx, w = grid_indices(x.float(), scales[l], ress[l], sizes[l])
        vals = table[offsets[l] + idx]                                       # [N,8,F]
        outs.append((vals * w[:, :, None]).sum(1))
        all_idx.append(idx)
    out = torch.cat(outs, -1)
    return (out, torch.stack(all_idx, 1)) if return_indices else out


def n_params(n_levels, n_features, base_resolution, per_level_scale, log2_hashmap_size, grid_type="hash"):
    return level_table(n_levels, base_resolution, per_level_scale, log2_hashmap_size, grid_type)[3][-1] * n_features
