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

The mapping iteration is ``mneslam_amd.fused.HashFusedStep`` (bench workload ``replica_office0_hashT19_2x64_2048x128``);
``render_rays`` / ``forward`` / ``render_img`` / ``render_maps`` / ``query_*`` / ``run_network*`` keep JointEncoding's
signatures and run the same kernels with the grid features as caller-supplied feature rows (``hip_path.Hash*``),
differentiable w.r.t. the table, the decoder AND the rays (R13: OneBlob share from the render backward + the grid's
trilinear-weight share, ``mne_hash_ray_grad``), so the pose loops of loop closure run on this model through the host's
autograd loop.  ``grid.enc: dense`` gives BASELINE configs[0]'s 16^3 grid.
"""
import torch

from .decoder import ColorSDFNet_v2
from .encodings import get_encoder
from .scene_rep import JointEncoding
from .. import hip_path
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

    # ------------------------------------------------------------------ rendering / queries (HIP, hip_path.Hash*)
    def _render(self, rays_o, rays_d, target_rgb, target_d, u=None):
        dev = rays_o.device
        has_d = target_d is not None
        if not has_d and not self.config["training"].get("n_samples"):
            raise KeyError("n_samples")            # the reference raises the same
        tables = hip_path.linspace_tables(self.config, has_d, dev)
        seed_offset = (0, 0)
        if u is None:
            tr = self.config["training"]
            S = (tr["n_range_d"] + tr["n_samples_d"]) if has_d else tr["n_samples"]
            u, seed_offset = self._jitter(rays_o.shape[0], S, rays_o)
        return hip_path.HashRenderFunction.apply(self._info(), self.embed_fn.cfg, tables, rays_o, rays_d, target_rgb, target_d, u,
                                                 seed_offset, self.embed_fn.params, *self.decoder.hip_weights())

    def render_maps(self, rays_o, rays_d, target_d=None, u=None, stats=None):
        if stats is not None:
            raise NotImplementedError("decoded-sample statistics are kept for the tri-plane render only")
        dev = rays_o.device
        has_d = target_d is not None
        if not has_d and not self.config["training"].get("n_samples"):
            raise KeyError("n_samples")
        tr = self.config["training"]
        S = (tr["n_range_d"] + tr["n_samples_d"]) if has_d else tr["n_samples"]
        seed_offset = (0, 0)
        if u is None:
            u, seed_offset = self._jitter(rays_o.shape[0], S, rays_o)
        rgb, depth, disp, acc, var = hip_path.hash_render_maps(self._info(), self.embed_fn.cfg, hip_path.linspace_tables(self.config, has_d, dev),
                                                               rays_o, rays_d, target_d, u, seed_offset, self.embed_fn.params,
                                                               self.decoder.hip_weights())
        return {"rgb": rgb, "depth": depth, "disp_map": disp, "acc_map": acc, "depth_var": var}

    # render_rays / forward / render_img / query_color / run_network(_flat) / render_surface_color are JointEncoding's: they
    # go through _render, render_maps and _query.  Whole-frame renders hold one 256-byte feature row per sample:
    render_chunk_rays = 1 << 16

    def _query(self, pts, want_raw=True, want_geo=False, want_feat=False, **unsupported):
        if unsupported.get("normalised") or unsupported.get("want_corner_idx"):
            raise NotImplementedError("plane coordinates / corner indices belong to the tri-plane encoding")
        return hip_path.hash_query_points(self._info(), self.embed_fn.cfg, self.embed_fn.params, self.decoder.hip_weights(), pts,
                                          want_raw=want_raw, want_geo=want_geo, want_feat=want_feat)

    def sample_plane_feature(self, *a, **k):
        raise NotImplementedError("HashJointEncoding has no planes (model/scene_rep.py:28-53 belongs to the tri-plane wiring)")
