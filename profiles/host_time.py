"""Host-side cost of one fused step (Python + ctypes launches), measured by timing the launch loop alone."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mneslam_amd import configs
ag = bench.Agent(configs.bench_office0(), torch.device("cuda"), seed=0, n_keyframes=20, path="fused")
for _ in range(20):
    ag.step(prefetch=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):           # short enough that the launch queue does not fill up and throttle the host
    ag.step(prefetch=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host launch time per step: %.1f us; GPU drain afterwards: %.1f us/step" % ((t1 - t0) / 50 * 1e6, (t2 - t1) / 50 * 1e6))
