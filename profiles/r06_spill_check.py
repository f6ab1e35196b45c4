"""Do the tile lists of the list-bound workloads overflow their per-XCD segments?  (A tile with an overflowed segment walks the whole spill area.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mneslam_amd import configs, _lib

for name in sys.argv[1:] or ["indoor", "scannet", "office0"]:
    cfg = configs.WORKLOADS[name][0]()
    ag = bench.Agent(cfg, torch.device("cuda"), seed=0, n_keyframes=20)
    f = ag.fused
    for _ in range(60):
        ag.step(prefetch=True)
    torch.cuda.synchronize()
    # one more step with the plane update held back so that the counters are still there
    orig = f.lib.mne_tile_adam
    f.lib.mne_tile_adam = lambda *a: 0
    ag.step(prefetch=False)
    torch.cuda.synchronize()
    f.lib.mne_tile_adam = orig
    cnt = f.tile_counts.view(-1, _lib.LIST_SEGMENTS).cpu()
    tiles = [((p.shape[2] + 15) // 16) * ((p.shape[3] + 15) // 16) for p in f.planes]
    caps = [int(f.bins.plane_cap[k]) or int(f.bins.cap) for k in range(len(f.planes))]
    off, over_t, over_e = 0, 0, 0
    for k, (nt, cap) in enumerate(zip(tiles, caps)):
        c = cnt[off:off + nt]
        seg_cap = (cap - cap % _lib.LIST_SEGMENTS) // _lib.LIST_SEGMENTS
        ov = (c > seg_cap)
        over_t += int(ov.any(1).sum())
        over_e += int((c - seg_cap).clamp(min=0).sum())
        print(f"{name} plane {k}: tiles {nt} cap {cap} (segment {seg_cap}) entries {int(c.sum())} longest list {int(c.sum(1).max())} longest segment {int(c.max())} "
              f"tiles with an overflowed segment {int(ov.any(1).sum())}")
        off += nt
    print(f"{name}: entries {int(cnt.sum())}, tiles with overflow {over_t}, entries beyond their segments {over_e}, spill_count {int(f.spill_count.item())}, dropped {int(f.dropped.item())}")
    del ag
    torch.cuda.empty_cache()
