"""Round 6, step 0 of VERDICT r05 #1: how many plane parameters sit in tiles that an iteration neither writes (empty list)
nor can read?  Run on the GPU box.  For iterations t of a steady-state run it prints the fraction of parameters in

  W_t        tiles with a non-empty list in iteration t (prev_counts after the step),
  Nap_{t+1}  tiles under the bilinear footprint of the a-priori sample prefix of batch t+1 (what gather_kernel reads),
  Nx1_{t+1}  the same with the resolver's one-tile extension (32 more samples per ray),
  Nall_{t+1} tiles under the footprint of ALL R x S samples (superset of anything a deferred ray can read),

and of their unions with W_t, plus the fraction no iteration of a window of 20 / 50 touches at all.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from mneslam_amd import _lib, configs


def tile_params(planes):
    out = []
    for p in planes:
        h, w = p.shape[2], p.shape[3]
        ty = torch.arange((h + 15) // 16)
        tx = torch.arange((w + 15) // 16)
        ch = torch.clamp(h - 16 * ty, max=16)[:, None]
        cw = torch.clamp(w - 16 * tx, max=16)[None, :]
        out.append((ch * cw * 32).reshape(-1))
    return torch.cat(out)


def footprint(f, model, mask):
    """bool [n_tiles]: tiles holding a bilinear corner of a masked sample (ATen grid_sampler_2d, align_corners, border)."""
    dev = f.rays_o.device
    R = f.n_active
    pts = f.rays_o[:R, None, :] + f.rays_d[:R, None, :] * f.z_vals[:R, :, None]
    pts = pts[mask[:R]]
    lo = model.bound[:, 0].to(dev)
    hi = model.bound[:, 1].to(dev)
    pn = ((pts - lo) / (hi - lo)) * 2.0 - 1.0
    axes = [(0, 1), (0, 2), (1, 2)]                      # xy: W = X, H = Y; xz: W = X, H = Z; yz: W = Y, H = Z
    hit = []
    k = 0
    for s in range(len(f.planes) // 6):
        for o in range(3):
            for l in range(2):
                p = f.planes[k]
                k += 1
                h, w = p.shape[2], p.shape[3]
                ntx, nty = (w + 15) // 16, (h + 15) // 16
                ix = torch.clamp(((pn[:, axes[o][0]] + 1) / 2) * (w - 1), 0, w - 1)
                iy = torch.clamp(((pn[:, axes[o][1]] + 1) / 2) * (h - 1), 0, h - 1)
                x0, y0 = ix.floor().long(), iy.floor().long()
                t = torch.zeros(nty * ntx, dtype=torch.bool, device=dev)
                for dx in (0, 1):
                    for dy in (0, 1):
                        x, y = torch.clamp(x0 + dx, max=w - 1), torch.clamp(y0 + dy, max=h - 1)
                        t[(y // 16) * ntx + x // 16] = True
                hit.append(t)
    return torch.cat(hit)


def run(name, cfg, warm, n_iter, fresh=False):
    dev = torch.device("cuda")
    ag = bench.Agent(cfg, dev, seed=0, n_keyframes=20, path="fused")
    f = ag.fused
    for _ in range(warm):
        ag.step(prefetch=True)
    torch.cuda.synchronize()
    S = f.S
    ntile = (S + 31) // 32
    wts = tile_params(f.planes).to(dev).double()
    tot = wts.sum()
    frac = lambda m: float((wts * m.double()).sum() / tot)
    ar = torch.arange(S, device=dev)[None, :]
    prevN = None
    touched20 = touched50 = None
    rows = []
    for it in range(n_iter):
        ag.step(prefetch=True)
        torch.cuda.synchronize()
        Wt = f.prev_counts > 0
        need = f.ray_counts[:f.n_active, _lib.C_NEED]
        tl = torch.clamp((need + 31) // 32, 1, ntile)
        adapt = int(f.adapt_state[0]) if f.adapt_state is not None else 0
        m_ap = ar < (tl * 32)[:, None]
        m_x1 = ar < (torch.clamp(tl + 1, max=ntile) * 32)[:, None]
        m_all = torch.ones_like(m_ap)
        Nap, Nx1, Nall = (footprint(f, ag.model, m) for m in (m_ap, m_x1, m_all))
        rows.append((it, adapt, frac(Wt), frac(Nap), frac(Nx1), frac(Nall), frac(Wt | Nap), frac(Wt | Nx1), frac(Wt | Nall)))
        if prevN is not None:
            pass
        u = Wt | Nx1
        touched20 = u.clone() if it % 20 == 0 or touched20 is None else (touched20 | u)
        touched50 = u.clone() if it % 50 == 0 or touched50 is None else (touched50 | u)
        if it % 20 == 19:
            print(f"{name}: window of 20 ending at {it}: {1 - frac(touched20):.3f} of the parameters untouched by W | Nx1")
        if it % 50 == 49:
            print(f"{name}: window of 50 ending at {it}: {1 - frac(touched50):.3f} of the parameters untouched by W | Nx1")
    print(f"{name}: tiles {wts.numel()}, parameters {int(tot)}; per iteration: adapt-mode, fraction of parameters in")
    print("   it mode   W_t    Nap    Nx1   Nall  W|Nap  W|Nx1 W|Nall")
    for r in rows[:8] + rows[-4:]:
        print("  %3d  %d   %.3f  %.3f  %.3f  %.3f  %.3f  %.3f  %.3f" % r)
    t = torch.tensor([r[2:] for r in rows])
    print("  mean     " + "  ".join("%.3f" % v for v in t.mean(0).tolist()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["office0", "office0_fresh", "scannet", "apartment"]
    for w in which:
        if w == "office0":
            run(w, configs.bench_office0(), 300, 100)
        elif w == "office0_fresh":
            run(w, configs.bench_office0(), 0, 30)
        else:
            run(w, configs.WORKLOADS[w][0](), 100, 40)
