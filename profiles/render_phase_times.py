"""Experiment (needs a -DRENDER_PROFILE build of render.hip as the in-tree library): per-phase wall time of
decode_kernel tile tasks and ray_kernel rays, from the 100 MHz clock stamps the kernels leave in the spill tail."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mneslam_amd import configs
cfg = configs.bench_office0()
ag = bench.Agent(cfg, torch.device("cuda"), seed=0, n_keyframes=20, path="fused")
f = ag.fused
for _ in range(int(os.environ.get("ITERS", "60"))):
    ag.step()
torch.cuda.synchronize()
f.spill[-65536:].zero_()
ag.step()
torch.cuda.synchronize()
us = 10.0 / 1000.0
dec = f.spill[-65536:-32768].contiguous().view(torch.int64).view(-1, 16).cpu().double()
dec = dec[dec[:, 0] > 0]
t = dec[:, :7] * us
t0 = t[:, 0].min()
names = ["z/rays/coords", "gather", "store feat rows", "oneblob", "MLP fwd (120 MFMA)", "raw/mask + staged tape"]
print(f"decode_kernel: {t.shape[0]} stamped tile tasks, kernel span {float(t[:, 6].max() - t0):.1f} us")
for which, sel in (("first task of a wave", torch.arange(t.shape[0]) % 2 == 0), ("second task", torch.arange(t.shape[0]) % 2 == 1)):
    pass
d = (t[:, 1:] - t[:, :-1])
print("  mean per phase us: " + " | ".join(f"{n} {float(x):.2f}" for n, x in zip(names, d.mean(0))) + f" | total {float((t[:, 6] - t[:, 0]).mean()):.2f}")
print("  p90  per phase us: " + " | ".join(f"{n} {float(torch.quantile(d[:, k], 0.9)):.2f}" for k, n in enumerate(names)))
print("  task start percentiles us:", [round(float(torch.quantile(t[:, 0] - t0, q)), 1) for q in (0.1, 0.5, 0.9, 1.0)])
print("  task end   percentiles us:", [round(float(torch.quantile(t[:, 6] - t0, q)), 1) for q in (0.1, 0.5, 0.9, 1.0)])
for c in (1, 2, 3, 4):
    m = dec[:, 7] == c
    if m.sum():
        print(f"  tile index {c - 1}: n={int(m.sum())} mean total {float((t[m][:, 6] - t[m][:, 0]).mean()):.2f} us, start median {float((t[m][:, 0] - t0).median()):.1f}")
ray = f.spill[-32768:].contiguous().view(torch.int64).view(-1, 32)[:f.R].cpu().double()
nb = ray[:, 31]
tr = ray[:, :31] * us
r0 = tr[:, 0][tr[:, 0] > 0].min()
print(f"ray_kernel: {int((tr[:, 0] > 0).sum())} rays, kernel span {float(tr[:, :23].max() - r0):.1f} us; backward tiles per ray: mean {float(nb.mean()):.2f} max {int(nb.max())}")
pn = ["load raws", "resolve", "composite", "contrib scan"]
d = tr[:, 1:5] - tr[:, 0:4]
print("  per ray us: " + " | ".join(f"{n} {float(x):.2f}" for n, x in zip(pn, d.mean(0))))
tn = ["coords/mask/dsdc", "MFMA bwd", "raygrad/pn", "staged tape", "append grouping", "atomics+entries"]
for c in range(3):
    m = nb > c
    if m.sum() == 0:
        continue
    base = 5 + 6 * c
    dd = tr[m][:, base + 1:base + 6] - tr[m][:, base:base + 5]
    print(f"  tile {c} (n={int(m.sum())}) us: " + " | ".join(f"{n} {float(x):.2f}" for n, x in zip(tn, dd.mean(0))) +
          f" | total {float((tr[m][:, base + 5] - tr[m][:, base]).mean()):.2f}")
end = torch.stack([tr[k, 4 + 6 * int(nb[k])] if nb[k] > 0 else tr[k, 4] for k in range(tr.shape[0])])
print("  ray start percentiles us:", [round(float(torch.quantile(tr[:, 0] - r0, q)), 1) for q in (0.1, 0.5, 0.9, 0.99, 1.0)])
print("  ray end   percentiles us:", [round(float(torch.quantile(end - r0, q)), 1) for q in (0.1, 0.5, 0.9, 0.99, 1.0)])
print("  ray duration us: mean", round(float((end - tr[:, 0]).mean()), 1), "p90", round(float(torch.quantile(end - tr[:, 0], 0.9)), 1))
