"""Per-step kernel times + work counters of the first iterations of a fresh agent."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mneslam_amd import configs
cfg = configs.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "office0"][0]()
dev = torch.device("cuda:0")
agent = bench.Agent(cfg, dev, seed=0, n_keyframes=20)
f = agent.fused
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for i in range(N):
    timers = {}
    agent.step(timers, prefetch=True)
    torch.cuda.synchronize()
    t = {k[:6]: round(sum(a.elapsed_time(b) for a, b in v) * 1e3) for k, v in timers.items()}
    R = f.R
    dec = int((f.ray_tiles[:R].long()).sum())
    print(i, t, "tape_rows", int(f.tape_rows), "tiles_walked", dec, "adapt", f.adapt_state.tolist() if f.adapt_state is not None else None,
          "maxlist", int(f.prev_counts.max()), "spill", int(f.spill_count))
