"""Experiment (needs a -DTILE_PROFILE build of tile_adam.hip as the in-tree library): per-phase wall time
of tile_adam_kernel workgroups, from the 100 MHz clock stamps the kernel leaves in the spill tail."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mneslam_amd import configs
cfg = configs.bench_office0()
ag = bench.Agent(cfg, torch.device("cuda"), seed=0, n_keyframes=20, path="fused")
for _ in range(int(os.environ.get("ITERS", "60"))):
    ag.step()
torch.cuda.synchronize()
f = ag.fused
raw = f.spill[-8192:].contiguous().view(torch.int64).view(-1, 8)[:4096].cpu().double()
t = raw[:, :7] * 10.0 / 1000.0                      # 100 MHz ticks -> us
cnt = raw[:, 7]
t0 = t[:, 0].min()
names = ["order+count+zero", "hist zero+barrier", "A stage+rank", "B prefix + C scatter", "D accumulate (all passes)", "Adam"]
print("blocks", t.shape[0], "kernel span us", float(t[:, 6].max() - t0))
for lo, hi in [(0, 1), (1, 64), (64, 256), (256, 512), (512, 1024), (1024, 100000)]:
    m = (cnt >= lo) & (cnt < hi)
    if m.sum() == 0:
        continue
    d = (t[m][:, 1:] - t[m][:, :-1]).mean(0)
    print(f"entries [{lo},{hi}) n={int(m.sum())}: total {float((t[m][:, 6] - t[m][:, 0]).mean()):.2f} us | " +
          " | ".join(f"{n} {float(x):.2f}" for n, x in zip(names, d)))
start = t[:, 0] - t0
print("block start time percentiles us:", [float(torch.quantile(start, q)) for q in (0.1, 0.5, 0.9, 0.99, 1.0)])
end = t[:, 6] - t0
print("block end time percentiles us:", [float(torch.quantile(end, q)) for q in (0.1, 0.5, 0.9, 0.99, 1.0)])
