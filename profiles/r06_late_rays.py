"""Round 6: how many rays of batch t+1 read a tile whose list was EMPTY in iteration t?

A tile with an empty list only gets its zero-gradient Adam step; if that part of the sweep ran beside the next iteration's
front end, only the rays that read such a tile would have to wait for it (they could go with the deferred rays).  This counts
them: per steady-state iteration, rays whose a-priori prefix + one extension tile (what gather / decode / the resolver read)
touches a tile that is empty in the PREVIOUS iteration's lists, split by rays with / without a valid target depth.
Run on the GPU box: python profiles/r06_late_rays.py [office0 office0_fresh scannet apartment indoor]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from mneslam_amd import _lib, configs
from r06_lazy_potential import tile_params


def ray_hits(f, model, mask, empty):
    """bool [R]: ray has a masked sample with a bilinear corner in a tile flagged in `empty` (bool over all tiles, plane-major)."""
    dev = f.rays_o.device
    R = f.n_active
    S = f.S
    pts = f.rays_o[:R, None, :] + f.rays_d[:R, None, :] * f.z_vals[:R, :, None]
    lo = model.bound[:, 0].to(dev)
    hi = model.bound[:, 1].to(dev)
    pn = ((pts - lo) / (hi - lo)) * 2.0 - 1.0
    axes = [(0, 1), (0, 2), (1, 2)]
    hit = torch.zeros(R, S, dtype=torch.bool, device=dev)
    k = 0
    base = 0
    for s in range(len(f.planes) // 6):
        for o in range(3):
            for l in range(2):
                p = f.planes[k]
                k += 1
                h, w = p.shape[2], p.shape[3]
                ntx, nty = (w + 15) // 16, (h + 15) // 16
                ix = torch.clamp(((pn[..., axes[o][0]] + 1) / 2) * (w - 1), 0, w - 1)
                iy = torch.clamp(((pn[..., axes[o][1]] + 1) / 2) * (h - 1), 0, h - 1)
                x0, y0 = ix.floor().long(), iy.floor().long()
                e = empty[base:base + nty * ntx]
                for dx in (0, 1):
                    for dy in (0, 1):
                        x, y = torch.clamp(x0 + dx, max=w - 1), torch.clamp(y0 + dy, max=h - 1)
                        hit |= e[(y // 16) * ntx + x // 16]
                base += nty * ntx
    return (hit & mask[:R]).any(1)


def run(name, cfg, warm, n_iter):
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
    ar = torch.arange(S, device=dev)[None, :]
    prev_empty = None
    rows = []
    for it in range(n_iter):
        ag.step(prefetch=True)
        torch.cuda.synchronize()
        # after step t: f holds batch t+1 (prefetched) and prev_counts = list lengths of iteration t
        empty = f.prev_counts == 0
        R = f.n_active
        need = f.ray_counts[:R, _lib.C_NEED]
        tl = torch.clamp((need + 31) // 32, 1, ntile)
        m_ap = ar < (tl * 32)[:, None]
        m_x1 = ar < (torch.clamp(tl + 1, max=ntile) * 32)[:, None]
        m_all = torch.ones_like(m_ap)
        nodepth = need >= S                                  # rays no loss mask can cut short
        late_ap = ray_hits(f, ag.model, m_ap, empty)
        late_x1 = ray_hits(f, ag.model, m_x1, empty)
        late_all = ray_hits(f, ag.model, m_all, empty)
        adapt = int(f.adapt_state[0]) if f.adapt_state is not None else 0
        rows.append((it, adapt, float((wts * empty.double()).sum() / tot), R, int(nodepth.sum()), int(late_ap.sum()), int(late_x1.sum()),
                     int((late_x1 & ~nodepth).sum()), int(late_all.sum()), int(tl[late_x1].sum()), int(tl.sum())))
    print(f"{name}: per iteration: adapt mode, fraction of parameters in empty-list tiles, rays, rays needing all samples, late rays by")
    print("  a-priori prefix | + one extension tile | of those with a cut prefix | by ALL samples | a-priori tiles of the late rays / of all rays")
    for r in rows[:10] + rows[-4:]:
        print("  %3d  %d  %.3f  R %5d  full %4d  late_ap %5d  late_x1 %5d  (cut %5d)  late_all %5d  tiles %6d / %6d" % r)
    t = torch.tensor([r[2:] for r in rows], dtype=torch.double).mean(0).tolist()
    print("  mean    %.3f  R %5.0f  full %4.0f  late_ap %5.0f  late_x1 %5.0f  (cut %5.0f)  late_all %5.0f  tiles %6.0f / %6.0f" % tuple(t))


if __name__ == "__main__":
    which = sys.argv[1:] or ["office0", "office0_fresh", "scannet", "apartment", "indoor"]
    for w in which:
        if w == "office0":
            run(w, configs.bench_office0(), 300, 40)
        elif w == "office0_fresh":
            run(w, configs.bench_office0(), 0, 30)
        else:
            run(w, configs.WORKLOADS[w][0](), 100, 20)
