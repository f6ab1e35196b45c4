"""Soak run: tens of thousands of prefetching mapping iterations per workload on one MI355X -- losses stay finite, no list entry is dropped
(FusedStep.check), quality keeps improving, the rate does not drift.  python profiles/r06_soak.py [iterations] [workloads...]"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mneslam_amd import configs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
names = sys.argv[2:] or ["office0", "office0_hash", "scannet", "indoor"]
for name in names:
    cfg = configs.WORKLOADS[name][0]()
    ag = bench.Agent(cfg, torch.device("cuda"), seed=0, n_keyframes=20, path="fused")
    block = n // 5
    print(f"{name}: {n} iterations in blocks of {block}")
    for b in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for it in range(block):
            ag.step(prefetch=it + 1 < block)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ag.fused.check()
        psnr, l1 = ag.quality()
        L = ag.fused.losses.detach().float().cpu()
        ok = bool(torch.isfinite(L[:2]).all()) and math.isfinite(psnr) and math.isfinite(l1)
        print(f"  block {b}: {block / dt:8.1f} it/s  psnr {psnr:6.2f}  depth-L1 {l1:.5f}  rgb/depth loss {float(L[0]):.3e} {float(L[1]):.3e}  finite {ok}  mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
        assert ok
    del ag
    torch.cuda.empty_cache()
