"""KeyFrameDatabase -- the ray store that feeds every mapping iteration.

Interface contract (what the host's tracker / mapper touch, reference model/keyframe.py:6-132 and its callers
mp_slam/mapper.py:92,135-138,204,290, tracker/factor_graph.py:184): constructor arguments, the attributes ``rays``
([num_kf, num_rays_to_save, 7] = dir3 | rgb3 | depth1, fp32, host memory), ``frame_ids``, ``num_rays_to_save``, ``H``,
``W`` and the methods below.  Seeded runs must sample the same rays as the reference, so the python ``random``
draws happen with the same populations in the same order (one ``random.sample`` per stored keyframe, one per global
batch); everything else is this build's own design:

* the store keeps a DEVICE mirror for the fused mapping step (``device_rays``), synchronised through a set of dirty
  slots -- ``add_keyframe`` / ``del_keyframe`` mark what they rewrite, so re-adding a slot or shrinking the store can
  never leave stale rays on the device (``touch()`` marks slots written from outside through ``.rays``);
* ``sample_overlap_keyframe`` projects all probe points into all keyframes with one batched matrix product instead of
  a per-keyframe numpy loop.
"""
import random

import numpy as np
import torch

RAY_CH = 7          # dir(3) | rgb(3) | depth(1)


def _draw(population, k):
    """k distinct indices out of ``population`` with python's generator (the reference's RNG for ray selection)."""
    return random.sample(range(0, int(population)), int(k))


class KeyFrameDatabase(object):
    def __init__(self, config, H, W, num_kf, num_rays_to_save, device) -> None:
        self.config, self.device = config, device
        self.H, self.W = H, W
        self.num_rays_to_save = num_rays_to_save
        self.keyframes = {}
        self.rays = torch.zeros((num_kf, num_rays_to_save, RAY_CH))
        self.all_frame_ids = torch.arange(0, num_kf, dtype=torch.int32)
        self.frame_ids = [0]
        self._mirror, self._dirty = None, set()

    # ------------------------------------------------------------------ size
    def __len__(self):
        return len(self.frame_ids)

    def get_length(self):
        return len(self)

    # ------------------------------------------------------------------ writes
    @staticmethod
    def _pack(batch):
        """[1, H*W, 7] view of a frame dict (datasets/dataset.py:121-128)."""
        cols = (batch["direction"], batch["rgb"], batch["depth"].unsqueeze(-1))
        return torch.cat(cols, dim=-1).reshape(1, -1, RAY_CH)

    def sample_single_keyframe_rays(self, rays, option="random"):
        """Choose ``num_rays_to_save`` pixels of one frame.  ``rays`` is [1, n, 7]; returns [1, num_rays_to_save, 7].
        ``random``: uniformly.  ``filter_depth``: the reference draws positions below the NUMBER of pixels with
        0 < depth <= cam.depth_trunc but then indexes the unfiltered pixel list with them (model/keyframe.py:33-42);
        that behaviour is kept bit for bit (the option is off in every shipped config)."""
        if option == "random":
            pick = _draw(self.H * self.W, self.num_rays_to_save)
        elif option == "filter_depth":
            depth = rays[..., -1]
            n_valid = int(((depth > 0.0) & (depth <= self.config["cam"]["depth_trunc"])).sum())
            pick = _draw(n_valid, self.num_rays_to_save)
        else:
            raise NotImplementedError(option)
        return rays[:, pick]

    def add_keyframe(self, batch, counter, filter_depth=False):
        """Store the rays of keyframe number ``counter`` (1-based; it becomes the last live keyframe)."""
        counter = int(counter)
        chosen = self.sample_single_keyframe_rays(self._pack(batch), "filter_depth" if filter_depth else "random")
        self.frame_ids = self.all_frame_ids[:counter]
        self.rays[counter - 1] = chosen
        self._dirty.add(counter - 1)

    def del_keyframe(self, idx):
        """Drop keyframe slot ``idx`` when at least one later keyframe exists (the tracker removes redundant
        keyframes, tracker/factor_graph.py:184): later slots move down by one, the id list is cut after ``idx``."""
        if len(self.frame_ids) < idx + 2:
            return
        keep = [k for k in range(self.rays.shape[0]) if k != idx]
        self.rays = self.rays[keep]
        self.frame_ids = self.frame_ids[:idx + 1]
        self._mirror = None                                  # shape changed: rebuild the device mirror

    def touch(self, slots=None):
        """Tell the store that ``rays[slots]`` (default: everything) was written from outside."""
        self._dirty.update(range(self.rays.shape[0]) if slots is None else slots)

    # ------------------------------------------------------------------ reads
    def sample_global_rays(self, bs):
        """``bs`` distinct rays over all live keyframes + the id of the keyframe each came from."""
        per, live = self.num_rays_to_save, len(self.frame_ids)
        flat = torch.tensor(_draw(live * per, bs))
        owner = torch.div(flat, per, rounding_mode="trunc")
        return self.rays[:live].reshape(-1, RAY_CH)[flat], self.frame_ids[owner]

    def sample_global_keyframe(self, window_size, n_fixed=1):
        """Whole keyframes: ``window_size`` random ones among all but the last ``n_fixed``, plus those last ones
        (everything when the store is smaller than the window)."""
        live = len(self.frame_ids)
        if window_size >= live:
            return self.rays[:live], self.frame_ids
        slots = _draw(live - n_fixed, window_size) + list(range(live - n_fixed, live))
        return self.rays[slots], torch.cat([self.frame_ids[slots[:window_size]], self.frame_ids[-n_fixed:]], dim=0)

    @torch.no_grad()
    def sample_overlap_keyframe(self, batch, frame_id, est_c2w_list, k_frame, n_samples=16, n_pixel=100, dataset=None):
        """Keyframes that see the current frame's geometry (NICE-SLAM style): ``n_pixel`` random pixels are lifted to
        ``n_samples`` points each between 0.8 d and d + 0.5, projected into every keyframe, and keyframes are ranked
        by the fraction of points landing inside the image (20 px margin, in front of the camera).  Up to ``k_frame``
        of those with any overlap are drawn in random order; the newest keyframe is always included.
        Returns (rays [k, num_rays_to_save, 7], list of the chosen slots)."""
        cam = self.config["cam"]
        c2w = est_c2w_list[frame_id]
        pix = torch.randint(dataset.H * dataset.W, (n_pixel,))
        d_cam = batch["direction"].reshape(-1, 3)[pix].to(self.device)
        depth = batch["depth"].reshape(-1, 1)[pix].to(self.device)
        t = torch.linspace(0.0, 1.0, steps=n_samples).to(depth)
        z = (depth * 0.8) * (1.0 - t) + (depth + 0.5) * t                          # [n_pixel, n_samples]
        d_world = torch.sum(d_cam[..., None, :] * c2w[:3, :3].to(self.device), -1)
        pts = c2w[:3, -1].to(self.device) + d_world[:, None, :] * z[..., None]
        pts_h = torch.cat([pts.reshape(-1, 3), torch.ones(pts.numel() // 3, 1, device=pts.device)], -1).double().cpu()
        ids = [int(f) for f in self.frame_ids]
        w2c = torch.linalg.inv(torch.stack([est_c2w_list[f].double().cpu() for f in ids]))      # [K,4,4]
        cam_pts = torch.einsum("kij,nj->kni", w2c, pts_h)[..., :3]                  # all points in all cameras
        cam_pts[..., 0] *= -1.0
        K = torch.tensor([[cam["fx"], 0.0, cam["cx"]], [0.0, cam["fy"], cam["cy"]], [0.0, 0.0, 1.0]], dtype=torch.float64)
        uvw = torch.einsum("ij,knj->kni", K, cam_pts)
        zc = uvw[..., 2] + 1e-5
        u, v = (uvw[..., 0] / zc).float(), (uvw[..., 1] / zc).float()
        edge = 20
        inside = (u < cam["W"] - edge) & (u > edge) & (v < cam["H"] - edge) & (v > edge) & (zc < 0)
        score = inside.double().mean(dim=1)
        ranked = sorted(range(len(ids)), key=lambda s: float(score[s]), reverse=True)
        overlapping = [s for s in ranked if float(score[s]) > 0.0]
        chosen = list(np.random.permutation(np.array(overlapping))[:k_frame]) if overlapping else []
        newest = len(ids) - 1
        if newest not in chosen:
            chosen.append(newest)
        chosen = [int(s) for s in chosen]
        return self.rays[chosen], chosen

    # ------------------------------------------------------------------ device mirror (fused path)
    def device_rays(self, device):
        """Device copy of ``rays``: created on first use, afterwards only the slots written since the previous call
        are uploaded."""
        device = torch.device(device)
        if self._mirror is None or self._mirror.device != device or self._mirror.shape != self.rays.shape:
            self._mirror = self.rays.to(device)
            self._dirty.clear()
        elif self._dirty:
            slots = sorted(s for s in self._dirty if s < self.rays.shape[0])
            self._mirror[slots] = self.rays[slots].to(device)
            self._dirty.clear()
        return self._mirror
