import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mneslam_amd import configs
cfg = configs.bench_office0(); dev = torch.device("cuda")
fin = {}
for mode in ("binned", "atomics"):
    ag = bench.Agent(cfg, dev, seed=3, n_keyframes=4, path="fused", scatter=mode)
    for _ in range(6): ag.step()
    fin[mode] = [p.detach().clone() for lst in ag.model.all_planes for p in lst] + [p.detach().clone() for p in ag.model.decoder.parameters()]
    print(mode, "losses", ag.fused.losses.tolist())
    del ag; torch.cuda.empty_cache()
for k, (a, b) in enumerate(zip(fin["binned"], fin["atomics"])):
    d = (a - b).abs()
    print(k, tuple(a.shape), "max", float(d.max()), "mean", float(d.mean()), "frac>1e-4", float((d > 1e-4).float().mean()), "nonzero frac", float((a != 0).float().mean()))
