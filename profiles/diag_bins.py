"""Diagnostic: distribution of the per-tile list lengths of the binned scatter (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mneslam_amd import configs
cfg = configs.bench_office0()
ag = bench.Agent(cfg, torch.device("cuda"), seed=0, n_keyframes=20, path="fused")
f = ag.fused
import ctypes as C
from mneslam_amd import _lib
# run the pieces of one step up to (not including) tile_adam by monkeypatching
orig = f.lib.mne_tile_adam
class Dummy:
    def __call__(self, *a): return 0
f.lib.mne_tile_adam = Dummy()
ag.step()
torch.cuda.synchronize()
cnt = f.tile_counts.cpu()
print("tiles", cnt.numel(), "entries", int(cnt.sum()), "max", int(cnt.max()), "mean", float(cnt.float().mean()),
      "over_cap", int((cnt > f.bins.cap).sum()), "spill", int(f.spill_count.cpu()), "tape_rows", int(f.tape_rows.cpu()))
print("hist", torch.histc(cnt.float(), bins=16, min=0, max=float(cnt.max()) + 1).tolist())
nz = (cnt > 0)
print("touched tiles", int(nz.sum()), "of", cnt.numel(), "=", float(nz.float().mean()))
import ctypes as C
base = [0]
# per plane: touched fraction (planes in [set][orient][level] order; tile grid = ceil(h/16) x ceil(w/16))
off = 0
for k, p in enumerate(f.planes):
    h, w = p.shape[2], p.shape[3]
    nt = ((h + 15) // 16) * ((w + 15) // 16)
    c = cnt[off:off + nt]
    print("plane", k, (h, w), "tiles", nt, "touched", float((c > 0).float().mean()), "entries", int(c.sum()), "max", int(c.max()))
    off += nt
