"""world_size-2 gloo tests of the multi-agent path (CPU): peer map hand-off, pose exchange, the
shared-decoder gradient all-reduce (extension) and bench.py's max-over-ranks timing rule."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _worker(rank, world, port, ret):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from mneslam_amd import configs, dist as mdist
    from mneslam_amd.model.scene_rep import JointEncoding
    r, w, dev = mdist.init_agents(backend="gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    cfg = configs.small_test_config(one_grid=(rank == 0))          # the two agents differ in shape: 6 vs 12 planes
    cfg["mapping"]["bound"] = [[-1.0, 1.0 + 0.4 * rank], [-1.2, 1.1], [-0.8, 0.9]]
    torch.manual_seed(100 + rank)
    bb = torch.tensor(cfg["mapping"]["bound"], dtype=torch.float64)
    model = JointEncoding(cfg, bb)
    # 1) whole-map hand-off rank 1 -> rank 0 (loop closure / fusion path of the reference)
    if rank == 1:
        mdist.send_model(model, dst=0)
        ref = [p.clone() for lst in model.all_planes for p in lst]
        torch.save({"planes": ref, "dec": [w_.detach().clone() for w_ in model.decoder.parameters()],
                    "bound": model.bound, "bb": bb}, ret + ".ref")
    else:
        cfg1 = configs.small_test_config(one_grid=False)
        shared = JointEncoding(cfg1, bb)                            # placeholder shapes, replaced wholesale
        mdist.recv_model_into(shared, src=1)
        dist.barrier()
        exp = torch.load(ret + ".ref")
        got = [p for lst in shared.all_planes for p in lst]
        assert len(got) == 12 and not shared.training
        for a, b in zip(got, exp["planes"]):
            assert a.shape == b.shape and torch.equal(a, b)
            assert a.is_contiguous(memory_format=torch.channels_last)
        for a, b in zip(shared.decoder.parameters(), exp["dec"]):
            assert torch.equal(a, b)
        assert torch.equal(shared.bound, exp["bound"]) and torch.equal(shared.bounding_box, exp["bb"])
    if rank == 1:
        dist.barrier()
    # 2) keyframe poses
    poses = torch.eye(4)[None].repeat(2 + rank, 1, 1) * (rank + 1)
    allp = mdist.gather_keyframe_poses(poses, torch.arange(2 + rank))
    assert [p.shape[0] for p, _ in allp] == [2, 3] and float(allp[1][0][0, 0, 0]) == 2.0
    # 3) shared-decoder gradient (extension): mean over agents
    g = torch.full((6208,), float(rank + 1))
    mdist.allreduce_mean_(g)
    assert torch.allclose(g, torch.full((6208,), 1.5))
    # 3b) overlap-region plane gradients (extension): two agents on one global lattice, 0.4 m apart along x
    from mneslam_amd import dist as md
    gb = [[-1.0, 1.4], [-1.2, 1.2], [-0.8, 0.8]]
    b0, b1 = md.aligned_agent_bounds(gb, 2, axis=0, overlap=0.4, cell=0.2)
    assert b0[0][0] == -1.0 and b1[0][1] == pytest.approx(1.4) and b0[0][1] > b1[0][0]
    mine_b, peer_b = (b0, b1) if rank == 0 else (b1, b0)

    def geom(bnd, res):                      # planes that span `bnd` with node spacing `res` (xy, xz, yz)
        n = [int(round((hi - lo) / res)) + 1 for lo, hi in bnd]
        return [((n[1], n[0]), bnd, (0, 1)), ((n[2], n[0]), bnd, (0, 2)), ((n[2], n[1]), bnd, (1, 2))]
    my_geo, peer_geo = geom(mine_b, 0.1), geom(peer_b, 0.1)
    gen = torch.Generator().manual_seed(5 + rank)
    grads = [torch.randn(1, 4, *shape, generator=gen) for shape, _, _ in my_geo]
    before = [g_.clone() for g_ in grads]
    torch.save(before, ret + f".g{rank}")
    md.exchange_overlap_gradients(grads, my_geo, 1 - rank, peer_geo)
    dist.barrier()
    other = torch.load(ret + f".g{1 - rank}")
    for g_, b_, o_, (shape, bnd, axes), (pshape, pbnd, _) in zip(grads, before, other, my_geo, peer_geo):
        (ys, xs), (pys, pxs) = md.overlap_slices(bnd, pbnd, shape, pshape, axes)
        assert (xs.stop - xs.start) * (ys.stop - ys.start) > 0
        exp = b_.clone()
        exp[:, :, ys, xs] += o_[:, :, pys, pxs]
        assert torch.equal(g_, exp)
    with pytest.raises(ValueError):          # a peer whose lattice is shifted by half a cell is refused
        md.overlap_slices([[0.0, 1.0]] * 3, [[0.05, 1.05]] * 3, (11, 11), (11, 11), (0, 1))
    # 4) timing rule
    assert mdist.max_over_ranks(0.1 * (rank + 1), dev) == pytest.approx(0.2)
    dist.destroy_process_group()
    open(ret + f".ok{rank}", "w").write("ok")


def test_two_agents_gloo(tmp_path):
    port = 29600 + (os.getpid() % 300)
    ret = str(tmp_path / "r")
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert os.path.exists(ret + ".ok0") and os.path.exists(ret + ".ok1")


def _shared_decoder_worker(rank, world, port, ret):
    """Two agents, each with its own planes and ray batch, ONE decoder (EXTENSION, BASELINE multi-GPU configs):
    FusedStep(shared_decoder=True) averages the decoder gradient over the agents before the decoder's Adam step,
    so decoders that start equal stay bit-equal while the planes diverge.  Kernels run through the host emulator."""
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    sys.path.insert(0, os.path.join(REPO, "tests", "hostemu"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import build_emu
    from mneslam_amd import _lib, configs, dist as mdist, slam_glue
    from mneslam_amd.fused import FusedStep
    import parity_cases as pc
    from helpers import load_golden
    _lib.unload()
    _lib.load(build_emu.build())
    mdist.init_agents(backend="gloo")
    g = load_golden("mapping3_onegrid_esdf")
    cfg = configs.small_test_config(one_grid=True, is_co_sdf=False)
    n_rays = 48
    model = pc.model_from_golden(g, cfg, "cpu", prefix="init.").train()           # same decoder on both ranks
    for lst in model.all_planes:                                                  # different maps
        for l in range(2):
            lst[l] = (lst[l] * (1.0 + 0.25 * rank)).contiguous(memory_format=torch.channels_last)
    opt = slam_glue.create_optimizer(model, cfg)
    fs = FusedStep(model, opt, cfg, n_rays, "cpu", shared_decoder=True)
    H, W = int(g["H"]), int(g["W"])
    k = 1 + rank                                                                  # different frame per agent
    cur = torch.cat([torch.from_numpy(g["direction"]), torch.from_numpy(g[f"frame{k}.rgb"]),
                     torch.from_numpy(g[f"frame{k}.depth"])[..., None]], -1).reshape(-1, 7).contiguous()
    poses = torch.from_numpy(g[f"frame{k}.c2w"]).reshape(1, 4, 4).contiguous()
    gen = torch.Generator().manual_seed(7 + rank)
    for it in range(2):
        idx = torch.randperm(H * W, generator=gen)[:n_rays]
        fs.step(None, 0, 1, cur, poses, 0, n_rays, idx_cur=idx, u=torch.rand(n_rays, fs.S, generator=gen))
    dec = torch.cat([p.detach().reshape(-1) for p in model.decoder.parameters()])
    gathered = [torch.zeros_like(dec) for _ in range(world)]
    dist.all_gather(gathered, dec)
    assert torch.equal(gathered[0], gathered[1]), "shared decoder diverged between the agents"
    init = torch.cat([torch.from_numpy(g[f"init.dec.{kk}"]).reshape(-1) for kk in pc.DEC_KEYS])
    assert torch.isfinite(dec).all() and not torch.equal(dec.sort().values, init.sort().values), "decoder did not train"
    pl = torch.cat([p.detach().reshape(-1)[:4096] for lst in model.all_planes for p in lst])
    gp = [torch.zeros_like(pl) for _ in range(world)]
    dist.all_gather(gp, pl)
    assert not torch.equal(gp[0], gp[1])                                          # planes stay per-agent
    dist.destroy_process_group()
    open(ret + f".ok{rank}", "w").write("ok")


def test_two_agents_shared_decoder_fused_step(tmp_path):
    port = 29900 + (os.getpid() % 90)
    ret = str(tmp_path / "s")
    sys.path.insert(0, os.path.join(REPO, "tests", "hostemu"))
    import build_emu
    build_emu.build()                           # once, before the workers race for it
    mp.spawn(_shared_decoder_worker, args=(2, port, ret), nprocs=2, join=True)
    assert os.path.exists(ret + ".ok0") and os.path.exists(ret + ".ok1")


# --------------------------------------------------------------------------------------------------------------
# EXTENSION: two agents on one lattice, binned plane update, shared cells + shared decoder
# --------------------------------------------------------------------------------------------------------------
def _lattice_config(rank):
    """Two slabs along x whose planes sit on ONE lattice on both levels: extended x length 2.4 m with 13 / 25 nodes
    (spacing 0.2 / 0.1 m), agent 1 shifted by 1.4 m = 7 coarse / 14 fine nodes; y and z extents are the agents' common ones."""
    from mneslam_amd import configs
    cfg = configs.small_test_config(one_grid=True, is_co_sdf=False, n_samples_d=21, n_range_d=11)
    x0 = -1.0 + 1.4 * rank
    cfg["mapping"]["bound"] = [[x0, x0 + 2.3], [-1.2, 1.1], [-0.8, 0.9]]             # extended by bound_dividable to 2.4 / 2.4 / 1.8
    cfg["planes_res"] = {"coarse": 0.181, "fine": 0.095, "bound_dividable": 0.2}
    room = [[x0 + 0.2, x0 + 2.2], [-1.0, 0.9], [-0.6, 0.7]]
    return cfg, room


def _binned_overlap_worker(rank, world, port, ret):
    """Two FusedSteps (binned plane update, overlap_peers + shared_decoder) against two oracle agents with the exchange
    written out in tensor ops: plane gradients summed over the node rectangles both agents hold, decoder gradient averaged,
    then Adam.  Equal by construction to ONE model over the union lattice trained on the union batch wherever the cells are
    shared.  Checks, over two iterations: every plane and decoder parameter against the oracle agents, and the shared cells
    bit-equal between the two HIP agents."""
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    sys.path.insert(0, os.path.join(REPO, "tests", "hostemu"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import build_emu
    from mneslam_amd import _lib, dist as mdist, slam_glue, synthetic
    from mneslam_amd.fused import FusedStep
    from mneslam_amd.model.scene_rep import JointEncoding
    from oracle import mapping as omap
    from oracle.scene_rep import OracleScene
    from helpers import assert_close
    _lib.unload()
    _lib.load(build_emu.build())
    mdist.init_agents(backend="gloo")
    cfg, room = _lattice_config(rank)
    bb = torch.tensor(cfg["mapping"]["bound"], dtype=torch.float64)
    torch.manual_seed(11)                                        # the same decoder on both agents
    model = JointEncoding(cfg, bb).train()
    geo = mdist.plane_geometry(model)
    geos = [None, None]
    dist.all_gather_object(geos, geo)
    peer_geo = geos[1 - rank]
    # planes = windows of one field over the union lattice (same seed on both ranks), so shared cells start equal
    flat = [p for lst in model.all_planes for p in lst]
    rects = []
    for k, (p, (shape, bnd, axes), (pshape, pbnd, _)) in enumerate(zip(flat, geo, peer_geo)):
        (ys, xs), (pys, pxs) = mdist.overlap_slices(bnd, pbnd, shape, pshape, axes)
        rects.append(((ys, xs), (pys, pxs)))
        off = (xs.start - pxs.start) if rank == 1 else 0        # this agent's first node on the union lattice (x only)
        if axes[0] != 0:
            assert xs == slice(0, shape[1]) and ys == slice(0, shape[0])          # yz planes: shared as a whole
        shift = pxs.start if rank == 0 else 0                    # union width = own + peer - shared
        width = shape[1] + pshape[1] - (xs.stop - xs.start)
        field = 0.05 * torch.randn(1, p.shape[1], shape[0], width, generator=torch.Generator().manual_seed(100 + k))
        start = 0 if (rank == 0 or axes[0] != 0) else width - shape[1]
        with torch.no_grad():
            p.copy_(field[..., start:start + shape[1]])
        del off, shift
    opt = slam_glue.create_optimizer(model, cfg)
    n_rays = 40
    fs = FusedStep(model, opt, cfg, n_rays, "cpu", scatter="binned", shared_decoder=True, overlap_peers=[(1 - rank, peer_geo)])
    assert fs.tile_overlap is not None and fs.ov_send[0].numel() > 32 * 25
    H, W = 34, 60
    frames = synthetic.make_frames(2, H, W, 30.0, 30.0, 29.5, 16.5, room, seed=3 + rank)
    fr = frames[1]
    cur = torch.cat([fr["direction"], fr["rgb"], fr["depth"][..., None]], -1).reshape(-1, 7).contiguous()
    poses = fr["c2w"].reshape(1, 4, 4).contiguous()
    # ---- oracle agent
    cpu = lambda t: t.detach().clone()
    sc = OracleScene(cfg, bb, build=False)
    sc.all_planes = tuple([cpu(p).contiguous() for p in lst] for lst in model.all_planes)
    sd = model.decoder.state_dict()
    sc.col_w = [cpu(sd["color_net.model.0.weight"]), cpu(sd["color_net.model.2.weight"])]
    sc.sdf_w = [cpu(sd["sdf_net.model.0.weight"]), cpu(sd["sdf_net.model.2.weight"])]
    sc.requires_grad_(True)
    oopt = omap.OracleAdam(sc, cfg)
    gen = torch.Generator().manual_seed(7 + rank)
    for it in range(2):
        idx = torch.randperm(H * W, generator=gen)[:n_rays]
        fs.step(None, 0, 1, cur, poses, 0, n_rays, idx_cur=idx, u=torch.rand(n_rays, fs.S, generator=gen))
        fs.synchronize()
        fs.check()
        oopt.zero_grad()
        r = sc.forward(fs.rays_o.clone(), fs.rays_d.clone(), fs.tgt_rgb.clone(), fs.tgt_d.clone()[:, None], impl="grid_sample",
                       z_vals=fs.z_vals.clone())
        omap.loss_from_ret(cfg, r, is_co_sdf=False).backward()
        mine = [p.grad.clone() for p in sc.plane_list()]
        theirs = [None, None]
        dist.all_gather_object(theirs, mine)
        with torch.no_grad():
            for p, g_peer, ((ys, xs), (pys, pxs)) in zip(sc.plane_list(), theirs[1 - rank], rects):
                p.grad[:, :, ys, xs] += g_peer[:, :, pys, pxs]
            for w in sc.decoder_list():
                gs = [None, None]
                dist.all_gather_object(gs, w.grad.clone())
                w.grad.copy_((gs[0] + gs[1]) / 2)
        oopt.step()
        lr = opt.param_groups[1]["lr"]
        for k, (p, ref) in enumerate(zip(flat, sc.plane_list())):
            d = (p.detach() - ref.detach()).abs()
            assert float(d.mean()) < 2e-3 * lr and float((d > 0.05 * lr).float().mean()) < 2e-3, \
                f"iteration {it} plane {k}: mean {float(d.mean()):.3e}, outliers {float((d > 0.05 * lr).float().mean()):.3e}"
        for w_hip, w_ref in zip(model.decoder.parameters(), sc.decoder_list()):
            d = (w_hip.detach() - w_ref.detach()).abs()
            assert float(d.mean()) < 2e-3 * opt.param_groups[0]["lr"], f"iteration {it}: decoder after Adam"
    # the exchange carried something: the shared cells' first moments hold the peer's share too
    ex = fs.ov_recv[0]
    assert float(ex.abs().max()) > 0
    # shared cells are bit-equal on the two agents (a + b == b + a; same moments, same step), the rest is not
    for k, (p, ((ys, xs), (pys, pxs))) in enumerate(zip(flat, rects)):
        both = [None, None]
        dist.all_gather_object(both, p.detach()[:, :, ys, xs].clone())
        assert torch.equal(both[0], both[1]), f"plane {k}: shared cells differ between the agents"
    dec = torch.cat([p.detach().reshape(-1) for p in model.decoder.parameters()])
    both = [None, None]
    dist.all_gather_object(both, dec)
    assert torch.equal(both[0], both[1])
    dist.destroy_process_group()
    open(ret + f".ok{rank}", "w").write("ok")


def test_two_agents_binned_overlap_shared_decoder(tmp_path):
    port = 29700 + (os.getpid() % 90)
    ret = str(tmp_path / "o")
    sys.path.insert(0, os.path.join(REPO, "tests", "hostemu"))
    import build_emu
    build_emu.build()                           # once, before the workers race for it
    mp.spawn(_binned_overlap_worker, args=(2, port, ret), nprocs=2, join=True)
    assert os.path.exists(ret + ".ok0") and os.path.exists(ret + ".ok1")


def test_bench_launcher_dry_run_two_ranks(tmp_path):
    """The driver's multi-GPU command line, as written in the task contract, on two ranks over gloo with the kernels in the
    host emulator: rendezvous, one agent per rank, warm-up, barrier-bracketed timed steps, max over ranks, decoder-gradient
    all-reduce, ONE JSON line from rank 0.  (Functional only: 16 + 4 rays on a tiny scene.)"""
    import json
    import subprocess
    sys.path.insert(0, os.path.join(REPO, "tests", "hostemu"))
    import build_emu
    env = dict(os.environ, MNE_EMULATED_LIBRARY=build_emu.build(), PYTHONPATH=REPO,
               MNE_NO_TILE_SPLIT="1")          # (the emulator pays one OS thread per work-item: no 2048 spare split items)
    port = 29800 + (os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--share-decoder", "--small", "--rays", "16", "--keyframes", "2"]
    out = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["agents"] == 2 and "all-reduce" in d["config"]["parallelism"] and "DRY RUN" in d["data"]
    assert abs(d["value"] - 2 * 1 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]          # whole-job rate = all agents' steps / time
