"""Round-3 root-cause experiment for the kernel-argument-layout-sensitive 2x64 + colour-plane training kernel
(DESIGN.md section 9).  Runs on the GPU box:

    python profiles/r03_layout_fuzz_diag.py <libmneslam_hip.so variant> [--poison]

One fused iteration of the ScanNet 2x64 shape (1024 + share rays x 117 samples, colour planes, after 2 warm-up steps) of
the given library build against one oracle iteration on the batch the device drew; prints, per decoder matrix, where the
gradient differs: by 32-row block of the hidden units and by input-column group.  --poison fills the tape with NaN before
the iteration under test: a NaN in a gradient then proves that the weight-gradient pass read a tape row no kernel wrote in
this iteration (a stale-read bug that the zero-initialised host emulator cannot see).
"""
import os
import sys

import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    lib_path = os.path.abspath(sys.argv[1])
    poison = "--poison" in sys.argv
    from mneslam_amd import _lib, configs
    _lib.unload()
    _lib.load(lib_path)
    import bench
    from oracle import mapping as omap
    from oracle.scene_rep import OracleScene
    workload = "scannet"
    hidden = 64
    for a in sys.argv[2:]:
        if a.startswith("--workload="):
            workload = a.split("=")[1]
        if a.startswith("--hidden="):
            hidden = int(a.split("=")[1])
    cfg = configs.WORKLOADS[workload][0](hidden)
    cfg["mapping"]["sample"] = 1024
    dev = torch.device("cuda")
    ag = bench.Agent(cfg, dev, seed=7, n_keyframes=4, path="fused", scatter="binned")
    fs, m = ag.fused, ag.model
    for _ in range(2):
        ag.step()
    fs.synchronize()
    torch.cuda.synchronize()
    cpu = lambda t: t.detach().to("cpu", copy=True)
    planes0 = [[cpu(p).contiguous() for p in lst] for lst in m.all_planes]
    dec0 = {k: cpu(v) for k, v in m.decoder.state_dict().items()}
    if poison:
        fs.tape.fill_(float("nan"))
        torch.cuda.synchronize()
    ag.step()
    fs.synchronize()
    torch.cuda.synchronize()
    fs.check()
    R, S = fs.R, fs.S
    rays_o, rays_d, tgt_rgb, tgt_d, z = cpu(fs.rays_o), cpu(fs.rays_d), cpu(fs.tgt_rgb), cpu(fs.tgt_d), cpu(fs.z_vals)
    sc = OracleScene(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float64), build=False)
    sc.all_planes = tuple([p.contiguous() for p in lst] for lst in planes0)
    sc.col_w = [dec0["color_net.model.0.weight"], dec0["color_net.model.2.weight"]]
    sc.sdf_w = [dec0["sdf_net.model.0.weight"], dec0["sdf_net.model.2.weight"]]
    sc.requires_grad_(True)
    ret = sc.forward(rays_o, rays_d, tgt_rgb, tgt_d[:, None], impl="grid_sample", z_vals=z)
    omap.loss_from_ret(cfg, ret, is_co_sdf=cfg["is_co_sdf"]).backward()
    n0, n1, n2 = [w.numel() for w in (sc.col_w[0], sc.col_w[1], sc.sdf_w[0])]
    dg = cpu(fs.dec_grad)
    got = [dg[:n0], dg[n0:n0 + n1], dg[n0 + n1:n0 + n1 + n2], dg[n0 + n1 + n2:]]
    names = ["color_net.0 [HIDC][pos48|cf64|geo15]", "color_net.2 [3][HIDC]", "sdf_net.0 [HID][feat64|pos48]", "sdf_net.2 [16][HID]"]
    print(f"lib {os.path.relpath(lib_path, REPO)}  poison={poison}  R={R} S={S} deferred-capable, contributing={int(fs.tape_rows.item())}")
    worst = 0.0
    for gk, w, nm in zip(got, sc.decoder_list(), names):
        ref = w.grad
        g = gk.reshape(ref.shape)
        scale = float(ref.abs().max())
        err = (g - ref).abs() / max(scale, 1e-30)
        n_nan = int(torch.isnan(g).sum())
        e = torch.nan_to_num(err, nan=1e9)
        worst = max(worst, float(e.max()))
        print(f"  {nm}: max err / max|ref| = {float(e.max()):.3e}  NaNs {n_nan}  (ref scale {scale:.3e})")
        rows, cols = ref.shape
        for r0 in range(0, rows, 32):
            blk = e[r0:r0 + 32]
            if cols > 64:
                groups = ([("pos", 0, 48), ("cf", 48, 112), ("geo", 112, cols)] if "color_net.0" in nm and cols > 100
                          else [("pos", 0, 48), ("geo", 48, cols)] if "color_net.0" in nm
                          else [("coarse", 0, 32), ("fine", 32, 64), ("pos", 64, cols)])
                desc = "  ".join(f"{n}:{float(blk[:, a:b].max()):.2e}" for n, a, b in groups)
            else:
                desc = f"all:{float(blk.max()):.2e}"
            print(f"      rows {r0:2d}..{min(r0 + 32, rows) - 1:2d}  {desc}")
    # plane parameters after the step are not compared here (tests do); first moments of the colour planes, as a proxy
    print("RESULT", "PASS" if worst < 2e-3 else "FAIL", f"worst {worst:.3e}")


if __name__ == "__main__":
    main()
