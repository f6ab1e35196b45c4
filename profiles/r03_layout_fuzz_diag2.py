"""Which tape rows / columns does the failing build leave unwritten?  (follow-up of r03_layout_fuzz_diag.py --poison)

    python profiles/r03_layout_fuzz_diag2.py <lib>

ScanNet 2x64 + colour planes, 2 warm-up steps, tape poisoned with NaN, one step; then for every column block of the tape
row the number of rows inside the region the weight-gradient pass reads (first ray_tiles[r]*32 samples of every ray) that
still hold a NaN, broken down by tile index, slot inside the tile (row % 32) and ray index modulo the 12 waves of a workgroup.
"""
import os
import sys

import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    lib_path = os.path.abspath(sys.argv[1])
    from mneslam_amd import _lib, configs
    _lib.unload()
    _lib.load(lib_path)
    import bench
    cfg = configs.WORKLOADS["scannet"][0](64)
    cfg["mapping"]["sample"] = 1024
    dev = torch.device("cuda")
    ag = bench.Agent(cfg, dev, seed=7, n_keyframes=4, path="fused", scatter="binned")
    fs = ag.fused
    for _ in range(2):
        ag.step()
    fs.synchronize()
    torch.cuda.synchronize()
    fs.tape.fill_(float("nan"))
    fs.ray_tiles.fill_(-7)
    torch.cuda.synchronize()
    ag.step()
    fs.synchronize()
    torch.cuda.synchronize()
    R, S = fs.R, fs.S
    tape = fs.tape.view(R, S, -1).cpu()
    rt = fs.ray_tiles[:R].cpu().long()
    print(f"lib {os.path.relpath(lib_path, REPO)}: R={R} S={S} row={tape.shape[-1]} ray_tiles min {int(rt.min())} max {int(rt.max())} hist {torch.bincount(rt.clamp(min=0)).tolist()}  unset(-7): {int((rt == -7).sum())}")
    ws = fs.ws
    # workspace carve (render.hip carve_workspace): masks | defer_list | dec_tiles | long_list | counters
    a16 = lambda x: (x + 15) & ~15
    off = a16(R * S * 16)
    defer_list = ws[off:off + 4 * R].view(torch.int32).cpu(); off += a16(4 * R)
    dec_tiles = ws[off:off + 4 * R].view(torch.int32).cpu().long(); off += a16(4 * R)
    off += a16(4 * R)
    cnt = ws[off:off + 32].view(torch.int32).cpu()
    print(f"  defer_count {int(cnt[0])} long_count {int(cnt[4])}  dec_tiles hist {torch.bincount(dec_tiles.clamp(min=0, max=8)).tolist()}")
    read = (torch.arange(S)[None, :] < (rt[:, None] * 32).clamp(max=S))            # rows wgrad reads
    blocks = [("X.feat", 0, 64), ("X.pos", 64, 112), ("OUT", 112, 128), ("H", 128, 192), ("HC", 192, 256), ("CF", 256, 320),
              ("DH", 320, 384), ("DHC", 384, 448), ("DOUT", 448, 464), ("DC", 464, 468), ("PN", 468, 472)]
    nanrow_any = torch.zeros(R, S, dtype=torch.bool)
    for name, a, b in blocks:
        bad = torch.isnan(tape[:, :, a:b]).any(-1) & read
        nanrow_any |= bad
        if int(bad.sum()) == 0:
            print(f"  {name:7s}: all {int(read.sum())} rows written")
            continue
        r_idx, s_idx = bad.nonzero(as_tuple=True)
        by_tile = torch.bincount(s_idx // 32, minlength=4).tolist()
        by_slot = torch.bincount(s_idx % 32, minlength=32).tolist()
        by_wave = torch.bincount(r_idx % 12, minlength=12).tolist()
        rays = r_idx.unique()
        print(f"  {name:7s}: {int(bad.sum())} unwritten rows in {rays.numel()} rays | by tile {by_tile} | by slot {by_slot} | by ray%12 {by_wave}")
        print(f"           first rays {rays[:12].tolist()}  their ray_tiles {rt[rays[:12]].tolist()} dec_tiles {dec_tiles[rays[:12]].tolist()}")
        # are the missing rows whole tiles?
        tiles = (bad.view(R, -1)[:, :S // 32 * 32].view(R, S // 32, 32).sum(-1))
        print(f"           rows missing per (ray, tile): histogram {torch.bincount(tiles.flatten()).tolist()}")
    deferred = set(defer_list[:int(cnt[0])].tolist())
    bad_rays = nanrow_any.any(1).nonzero().flatten().tolist()
    print(f"  rays with any unwritten row: {len(bad_rays)}; of them deferred in pass 1: {sum(1 for r in bad_rays if r in deferred)} (deferred total {len(deferred)})")


if __name__ == "__main__":
    main()
