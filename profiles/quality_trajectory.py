#!/usr/bin/env python3
"""Matched-quality evidence: PSNR / depth-L1 trajectories of the fused MI355X path and of the CPU oracle
trained on IDENTICAL batches (the device draws rays and jittered samples, the oracle replays them) from the same initial
parameters.  office0 bound at reduced plane resolution so that the oracle keeps up; usage: quality_trajectory.py [iters]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import parity_cases as pc
from mneslam_amd import configs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = configs.bench_office0()
cfg["planes_res"] = {"coarse": 0.08, "fine": 0.04, "bound_dividable": 0.08}
cfg["mapping"]["sample"] = 1024
torch.set_num_threads(min(os.cpu_count() or 1, 32))
rows = pc.quality_trajectory("cuda", cfg, n_iters=n, n_keyframes=8, seed=0, small=False)
print("office0 bound, planes 0.08/0.04 m, 1024+128 rays x 128 samples, identical batches; iteration | PSNR hip | PSNR oracle | depth-L1 hip | depth-L1 oracle")
for it, ph, po, dh, do in rows:
    if it < 5 or it % 10 == 9:
        print(f"{it + 1:5d}  {ph:8.3f}  {po:8.3f}   {dh:9.5f}  {do:9.5f}")
import statistics
last = rows[-20:]
print("mean of last 20: PSNR hip %.3f oracle %.3f | depth-L1 hip %.5f oracle %.5f" % (
    statistics.mean(r[1] for r in last), statistics.mean(r[2] for r in last), statistics.mean(r[3] for r in last), statistics.mean(r[4] for r in last)))
