"""decode_kernel by wave (variant build: git apply profiles/patches/r06_decode_wave_stamps.patch, build.build_variant("decprof", ["-DDECODE_PROFILE"]), git apply -R): when does every wave enter,
finish staging, start / end each of its tile tasks -- where do the 60 us of the first-pass decode go?  100 MHz clock stamps, lane 0 of each
wave, in the tail of the spill area.  python profiles/r06_decode_waves.py [config]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from mneslam_amd import _lib, build, configs
_lib.load(build.variant_path(os.environ.get("VARIANT", "decprof")))
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "office0"
cfg = configs.WORKLOADS[name][0]()
ag = bench.Agent(cfg, torch.device("cuda"), seed=0, n_keyframes=20, path="fused")
f = ag.fused
for _ in range(int(os.environ.get("ITERS", "300"))):
    ag.step(prefetch=True)
torch.cuda.synchronize()
for rep in range(3):
    f.spill[-65536:].zero_()
    ag.step(prefetch=True)
    torch.cuda.synchronize()
    st = f.spill[-65536:].contiguous().view(torch.int64).view(-1, 16).cpu().double()
    st = st[st[:, 0] > 0]
    us = 0.01
    t0 = st[:, 0].min()
    n = st[:, 14].long()
    q = lambda x, ps=(0.0, 0.1, 0.5, 0.9, 1.0): [round(float(torch.quantile(x, p)), 1) for p in ps]
    end = torch.stack([st[k, 3 + 2 * (int(n[k]) - 1)] if n[k] > 0 else st[k, 1] for k in range(st.shape[0])])
    print(f"{name} rep {rep}: {st.shape[0]} waves; tasks per wave {q(n.double())}; extension tiles decoded {int(st[:, 15].sum())}")
    print("  wave entry (min 10% 50% 90% max) us:", q((st[:, 0] - t0) * us), " staging done:", q((st[:, 1] - t0) * us))
    print("  wave end us:", q((end - t0) * us), " kernel span", round(float((end.max() - t0) * us), 1))
    for k in range(4):
        m = n > k
        if m.sum():
            d = (st[m][:, 3 + 2 * k] - st[m][:, 2 + 2 * k]) * us
            print(f"  task {k}: n={int(m.sum())} start {q((st[m][:, 2 + 2 * k] - t0) * us)} duration {q(d)}")
    slow = (end - t0) * us > torch.quantile((end - t0) * us, 0.97)
    print("  the 3% waves that end last: tasks", q(n[slow].double()), "extension tiles", q(st[slow][:, 15]), "entry", q((st[slow][:, 0] - t0) * us))
