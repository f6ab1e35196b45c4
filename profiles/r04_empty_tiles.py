"""How many plane tiles receive NO gradient in a mapping iteration (their Adam sweep depends on m, v only and could run
before / beside the render)?  Per plane, from the final list lengths of the last tile_adam launch (prev_counts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mneslam_amd import configs
for name in sys.argv[1:] or ["office0"]:
    cfg = configs.WORKLOADS[name][0]()
    ag = bench.Agent(cfg, torch.device("cuda"), seed=0, n_keyframes=20)
    for warm in (12, 60, 300):
        for i in range(warm):
            ag.step(prefetch=True)
        torch.cuda.synchronize()
        fs = ag.fused
        c = fs.prev_counts.cpu()
        tiles = [((p.shape[2] + 15) // 16) * ((p.shape[3] + 15) // 16) for p in fs.planes]
        params = [p.numel() for p in fs.planes]
        off, empty_params, tot = 0, 0, 0
        row = []
        for t, n in zip(tiles, params):
            z = int((c[off:off + t] == 0).sum())
            row.append(f"{z}/{t}")
            empty_params += n * z / t
            tot += n
            off += t
        print(f"{name} after {warm:3d} more steps: empty tiles per plane {' '.join(row)}; {100 * empty_params / tot:.1f} % of the parameters sit in empty tiles; "
              f"entries {int(c.sum())}, tiles {len(c)}, max list {int(c.max())}")
    del ag
    torch.cuda.empty_cache()
