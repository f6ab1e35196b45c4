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
