"""INS Indoor iteration: how many rays go to the deferred / long / heavy lists and how the backward tiles are distributed."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from mneslam_amd import configs
cfg = configs.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "indoor"][0]()
ag = bench.Agent(cfg, torch.device("cuda"), seed=0, n_keyframes=20)
for it in range(60):
    ag.step(prefetch=True)
ag.fused.synchronize(); torch.cuda.synchronize()
f = ag.fused
R, S = f.n_active, f.S
a16 = lambda x: (x + 15) & ~15
off = a16(f.R * S * 16) + 4 * a16(f.R * 4) + a16(f.R * 32)
tail = f.ws[off:off + 32].view(torch.int32).cpu().tolist()
rt = f.ray_tiles[:R].cpu()
rc = f.ray_counts[:R].cpu()
need = rc[:, 6]
print("R", R, "S", S, "deferred", tail[0], "heavy", tail[2], "long", tail[4])
print("backward tiles per ray: mean %.2f max %d" % (rt.float().mean(), rt.max()), "hist", torch.bincount(rt.clamp(max=33)).tolist())
print("a-priori samples per ray: mean %.1f, rays with more than 256: %d" % (need.float().mean(), int((need > 256).sum())))
print("valid depth rays", int((f.tgt_d[:R] > 0).sum()), "mean depth %.2f" % float(f.tgt_d[:R][f.tgt_d[:R] > 0].mean()))
