"""``get_encoder`` -- the reference's encoder factory (model/encodings.py:6-97) without tinycudann.

Only OneBlob is reachable in the reference (model/scene_rep.py:157; the hash-grid call at :160 is
commented out).  The module returned here has tinycudann's surface (``n_output_dims``, a zero-size
``params`` Parameter -> state_dict key ``embedpos_fn.params``) and runs the stand-alone HIP kernel.
"""
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


def get_encoder(encoding, input_dim=3, degree=4, n_bins=16, n_frequencies=12, n_levels=16, level_dim=2,
                base_resolution=16, log2_hashmap_size=19, desired_resolution=512):
    """Same signature and return value ``(module, out_dim)`` as the reference factory."""
    name = encoding.lower()
    if "blob" in name:
        embed = OneBlobEncoding(input_dim, n_bins)
        return embed, embed.n_output_dims
    raise NotImplementedError(
        f"encoding '{encoding}' is not wired in the reference's mapping path (model/scene_rep.py:160 is "
        "commented out) and is not provided by this build yet")
