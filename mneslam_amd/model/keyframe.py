"""KeyFrameDatabase -- the ray database feeding every mapping iteration (reference:
model/keyframe.py:6-103).  ``rays[k]`` holds ``num_rays_to_save`` random rays of keyframe k as
[dir3, rgb3, depth1].  Host RNG is python ``random`` exactly as in the reference so that seeded runs
sample the same rays; ``rays`` lives on the CPU like the reference's (a device-resident mirror for
the fused path is kept in sync by ``device_rays``)."""
import random

import torch


class KeyFrameDatabase(object):
    def __init__(self, config, H, W, num_kf, num_rays_to_save, device) -> None:
        self.config = config
        self.keyframes = {}
        self.device = device
        self.rays = torch.zeros((num_kf, num_rays_to_save, 7))
        self.num_rays_to_save = num_rays_to_save
        self.frame_ids = [0]
        self.all_frame_ids = torch.arange(0, num_kf, dtype=torch.int32)
        self.H, self.W = H, W
        self._dev_rays = None
        self._dev_count = 0

    def __len__(self):
        return len(self.frame_ids)

    def get_length(self):
        return self.__len__()

    def sample_single_keyframe_rays(self, rays, option="random"):
        """reference: model/keyframe.py:26-44"""
        if option == "random":
            idxs = random.sample(range(0, self.H * self.W), self.num_rays_to_save)
        elif option == "filter_depth":
            valid = (rays[..., -1] > 0.0) & (rays[..., -1] <= self.config["cam"]["depth_trunc"])
            rays = rays[valid, :][None]
            idxs = random.sample(range(0, rays.shape[1]), self.num_rays_to_save)
        else:
            raise NotImplementedError()
        return rays[:, idxs]

    def add_keyframe(self, batch, counter, filter_depth=False):
        """reference: model/keyframe.py:64-89"""
        rays = torch.cat([batch["direction"], batch["rgb"], batch["depth"][..., None]], dim=-1)
        rays = rays.reshape(1, -1, rays.shape[-1])
        rays = self.sample_single_keyframe_rays(rays, "filter_depth" if filter_depth else "random")
        if isinstance(counter, torch.Tensor):
            counter = int(counter)
        self.frame_ids = self.all_frame_ids[:counter]
        self.rays[counter - 1] = rays

    def sample_global_rays(self, bs):
        """reference: model/keyframe.py:91-103"""
        num_kf = len(self.frame_ids)
        idxs = torch.tensor(random.sample(range(num_kf * self.num_rays_to_save), bs))
        sample_rays = self.rays[:num_kf].reshape(-1, 7)[idxs]
        frame_ids = self.frame_ids[torch.div(idxs, self.num_rays_to_save, rounding_mode="trunc")]
        return sample_rays, frame_ids

    def device_rays(self, device):
        """Device mirror of the first len(self) keyframes (uploaded incrementally)."""
        n = len(self.frame_ids)
        if self._dev_rays is None or self._dev_rays.device != torch.device(device):
            self._dev_rays = torch.zeros(self.rays.shape, device=device)
            self._dev_count = 0
        if self._dev_count < n:
            self._dev_rays[self._dev_count:n].copy_(self.rays[self._dev_count:n])
            self._dev_count = n
        return self._dev_rays
