"""OneBlob positional encoding -- frozen spec (PARITY UNPINNED, see oracle/__init__).

Reference call sites: model/encodings.py:61-71 (``tcnn.Encoding(otype="OneBlob",
n_bins=16)``), used at model/scene_rep.py:157 and :292-294.  The arithmetic is
tinycudann's (not in the reference tree; requirements.txt:120).  This file
restates tinycudann's published OneBlob: per input dim x and bin b in [0, n):

    out[dim*n + b] = cdf((b+1)/n - x) - cdf(b/n - x)
    cdf(t)         = K(t) + K(t-1) + K(t+1)                (periodic wrap)
    K(t)           = clamp(15/16*u*(1 - 2/3*u^2 + 1/5*u^4) + 1/2, 0, 1),  u = t*n

with the right boundary of the last bin taken as cdf(0/n - x) + 1 (wrap).  All
arithmetic is fp32 in the op order written below (no FMA contraction); the
input is cast to fp32 first, as tinycudann does.

Where each line comes from in tiny-cuda-nn (NVlabs/tiny-cuda-nn; the reference pins nothing, requirements.txt:120, and
names commit 91ee479d275d322a65726435040fc20b56b9c991 as its fallback, README.md:99).  Recorded from the published source
as known to the author of this file -- tinycudann is NOT available in the build container (no network), so a maintainer
with a checkout must confirm the line-for-line correspondence; file placement of the helpers moved between
``encodings/oneblob.h`` and ``common_device.h`` across commits, the function NAMES below are stable:

    this file                            tiny-cuda-nn
    ----------------------------------   ---------------------------------------------------------------------------
    quartic_cdf(t, inv_radius)           ``quartic_cdf(const float x, const float inv_radius)``
                                         (include/tiny-cuda-nn/common_device.h; the kernel family ``quartic`` /
                                         ``quartic_cdf`` / ``quartic_cdf_deriv``):  u = x * inv_radius;  u2 = u*u;  u4 = u2*u2;
                                         fmaxf(0, fminf(1, (15/16) * u * (1 - (2/3) u2 + (1/5) u4) + 0.5))
    oneblob(): left boundary b / n,      ``one_blob_subwarp_aligned(kernel, data_in, elem_index, encoded_index,
    left_cdf = K(l-x) + K(l-x-1)           num_bins_log2)`` (include/tiny-cuda-nn/encodings/oneblob.h):
               + K(l-x+1)                  left_boundary = scalbnf(bin_index, -num_bins_log2);  left_cdf = kernel(left_boundary
                                           - x, n_bins) + kernel(left_boundary - x - 1, n_bins) + kernel(left_boundary - x + 1, n_bins)
    right_cdf = roll(left_cdf, -1)       ``right_cdf = __shfl_sync(0xffffffff, left_cdf, bin_index + 1, n_bins)``  (the right
                                           boundary of bin b is the left boundary of bin b + 1, taken from the neighbouring lane)
    wrap[-1] = 1                         ``if (bin_index == n_bins - 1) right_cdf += 1``  (the wrapped CDF lost one saturated term)
    out = right_cdf - left_cdf           ``return right_cdf - left_cdf``; kernel ``kernel_one_blob`` writes it at
                                           [dim * n_bins + bin] of the encoded row
    derivative 15/16 (1 - u^2)^2 n       ``quartic_cdf_deriv`` / ``kernel_one_blob_backward`` (used by csrc/render.hip's
                                           oneblob_half_backward for the ray gradients, R13)
    n_bins power of two                  ``OneBlobEncoding`` constructor: "Number of bins must be a power of 2"
"""
import torch

_C0 = 15.0 / 16.0
_C1 = 2.0 / 3.0
_C2 = 1.0 / 5.0


def quartic_cdf(t: torch.Tensor, inv_radius: float) -> torch.Tensor:
    u = t * inv_radius
    u2 = u * u
    u4 = u2 * u2
    poly = (_C0 * u) * ((1.0 - _C1 * u2) + _C2 * u4) + 0.5
    return torch.clamp(poly, 0.0, 1.0)


def oneblob(x: torch.Tensor, n_bins: int = 16) -> torch.Tensor:
    """x: [N, D] (any float dtype; cast to fp32) -> [N, D*n_bins] fp32."""
    assert n_bins & (n_bins - 1) == 0, "tinycudann OneBlob needs a power-of-two bin count"
    x = x.to(torch.float32)
    n, d = x.shape
    left = torch.arange(n_bins, dtype=torch.float32, device=x.device) / n_bins  # exact
    t = left[None, None, :] - x[:, :, None]                                     # [N, D, n]
    inv_r = float(n_bins)
    left_cdf = quartic_cdf(t, inv_r) + quartic_cdf(t - 1.0, inv_r) + quartic_cdf(t + 1.0, inv_r)
    right_cdf = torch.roll(left_cdf, shifts=-1, dims=-1)
    wrap = torch.zeros(n_bins, dtype=torch.float32, device=x.device)
    wrap[-1] = 1.0
    right_cdf = right_cdf + wrap
    return (right_cdf - left_cdf).reshape(n, d * n_bins)


class OneBlobEncoding(torch.nn.Module):
    """Stand-in with tinycudann's module surface (n_output_dims, zero-size ``params``)."""

    def __init__(self, n_input_dims=3, n_bins=16):
        super().__init__()
        self.n_input_dims = n_input_dims
        self.n_bins = n_bins
        self.n_output_dims = n_input_dims * n_bins
        self.params = torch.nn.Parameter(torch.zeros(0, dtype=torch.float32))

    def forward(self, x):
        return oneblob(x, self.n_bins)
