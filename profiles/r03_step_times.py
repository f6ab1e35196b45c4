"""Per-step device time of the first iterations of a fresh agent (driver form = 5 warm-up + 20 timed steps)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mneslam_amd import configs
make_cfg, workload = configs.WORKLOADS["office0"]
cfg = make_cfg()
dev = torch.device("cuda:0")
agent = bench.Agent(cfg, dev, seed=0, n_keyframes=20)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
torch.cuda.synchronize()
ev[0].record()
for i in range(N):
    t0 = time.perf_counter()
    agent.step(None, prefetch=True)
    host.append((time.perf_counter() - t0) * 1e6)
    ev[i + 1].record()
torch.cuda.synchronize()
dt = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(N)]
print("device us/step:", " ".join(f"{x:.0f}" for x in dt))
print("host   us/step:", " ".join(f"{x:.0f}" for x in host))
print("steps 5..24 mean", sum(dt[5:25]) / 20, " steps 40.. mean", sum(dt[40:]) / len(dt[40:]))
