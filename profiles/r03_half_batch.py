"""Kernel times at full / half / quarter ray batches on the office0 planes: are the render kernels latency- or throughput-bound?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mneslam_amd import configs
dev = torch.device("cuda:0")
for n in (2048, 1024, 512):
    cfg = configs.WORKLOADS["office0"][0]()
    cfg["mapping"]["sample"] = n
    cfg["mapping"]["min_pixels_cur"] = max(n // 20, 1)
    agent = bench.Agent(cfg, dev, seed=0, n_keyframes=20)
    for _ in range(40):
        agent.step(None, prefetch=True)
    timers = {}
    for i in range(100):
        agent.step(timers if i % 5 == 0 else None, prefetch=True)
    torch.cuda.synchronize()
    t = {k[:8]: round(sum(a.elapsed_time(b) for a, b in v) * 1e3 / len(v), 1) for k, v in timers.items()}
    print(n, agent.fused.R, t, flush=True)
    del agent
