"""HashJointEncoding -- the scene model with the hash-grid call put back (EXTENSION, parity unpinned).

The reference keeps Co-SLAM's sparse parametric encoding commented out (model/scene_rep.py:160 ``self.embed_fn,
self.input_ch = get_encoder(config['grid']['enc'], log2_hashmap_size=config['grid']['hash_size'],
desired_resolution=self.resolution_sdf)``, :243 ``embedded = self.embed_fn(inputs_flat)``) and runs tri-planes
instead; BASELINE.json's headline config nevertheless quotes the factory's defaults ("16-level hash grid (T=2^19) +
2x64 MLP", model/encodings.py:6-10, :31-46).  This class is that wiring: the grid features replace the tri-plane
features as the first decoder input, everything downstream (OneBlob, ColorSDFNet_v2, compositing, losses) is the
tri-plane path's.

Layout choice: the decoder keeps a 64-wide feature slot (``input_ch`` 64, as with planes); the grid fills its first
``n_levels * 2`` columns and the rest are dead inputs whose weight columns are zero-initialised and receive zero
gradients -- mathematically the [hidden, 32 + 48] first layer of the commented-out wiring, and the same HIP decoder
kernels serve both encodings.

Only the mapping iteration is provided for this model (``mneslam_amd.fused.HashFusedStep``, bench workload
``replica_office0_hashT19_2x64_2048x128``): there is no reference behaviour to mirror for the other entry points.
"""
import torch

from .decoder import ColorSDFNet_v2
from .encodings import get_encoder
from .scene_rep import JointEncoding
from .utils import batchify


class HashJointEncoding(JointEncoding):
    FEATURE_SLOT = 64

    def get_encoding(self, config):
        g = config["grid"]
        if not g["oneGrid"]:
            raise NotImplementedError("the hash-grid wiring has one grid (Co-SLAM's oneGrid: True)")
        self.embedpos_fn, self.input_ch_pos = get_encoder(config["pos"]["enc"], n_bins=config["pos"]["n_bins"])
        self.embed_fn, self.n_grid_features = get_encoder(g["enc"], log2_hashmap_size=g["hash_size"],
                                                          desired_resolution=g.get("desired_resolution", 512))
        if self.n_grid_features > self.FEATURE_SLOT or self.embed_fn.cfg.n_features != 2 or self.embed_fn.cfg.n_levels > 16:
            raise NotImplementedError("the fused form takes at most 16 levels of 2 features")
        self.embed_fn.to(self.device)
        self.input_ch = self.FEATURE_SLOT
        self.input_ch_pos = config["model"]["input_ch_pos"]
        self.all_planes = ()

    def get_decoder(self, config):
        self.decoder = ColorSDFNet_v2(config, input_ch=self.input_ch, input_ch_pos=self.input_ch_pos)
        with torch.no_grad():                           # dead feature columns (see the module docstring)
            self.decoder.sdf_net.model[0].weight[:, self.n_grid_features:self.FEATURE_SLOT] = 0.0
        self.color_net = batchify(self.decoder.color_net, None)
        self.sdf_net = batchify(self.decoder.sdf_net, None)

    def _info(self):
        info = super()._info()
        info["n_planes"] = 0
        return info

    def _no_planes(self, *a, **k):
        raise NotImplementedError("HashJointEncoding provides the fused mapping iteration only (fused.HashFusedStep); "
                                  "render_rays / forward / queries exist for the tri-plane model the reference runs")

    _render = render_rays = forward = render_img = render_maps = query_sdf = query_color = query_color_sdf = _no_planes
    run_network = run_network_flat = render_surface_color = _no_planes
