"""Parity of the HIP path on a real MI355X (through the C ABI of libmneslam_hip.so) against the
golden vectors captured from the reference and against the CPU oracle.  Run by the driver with
``-m gpu``; bodies shared with the host-emulator run live in tests/parity_cases.py."""
import math
import os

import pytest
import torch

import parity_cases as pc
from mneslam_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def real_library():
    if not torch.cuda.is_available():
        pytest.skip("these tests need an MI355X", allow_module_level=False)
    _lib.unload()
    lib = _lib.load()                       # in-tree libmneslam_hip.so only; raises if missing
    assert os.path.samefile(lib._name, _lib.LIB_PATH)
    yield
    torch.cuda.synchronize()


def test_oneblob():
    pc.check_oneblob(DEV)


def test_spherical_frequency_identity_encodings():
    """get_encoder('SphericalHarmonics' | 'Frequency' | 'Identity'), model/encodings.py:48-58, 73-95"""
    pc.check_misc_encodings(DEV)


def test_adam():
    pc.check_adam(DEV)


@pytest.mark.parametrize("name", list(pc.FWD_CASES))
def test_forward_matches_reference(name):
    pc.check_forward(name, DEV)


@pytest.mark.parametrize("name", list(pc.FWD_CASES))
@pytest.mark.parametrize("co", [False, True])
def test_gradients_match_reference(name, co):
    pc.check_backward(name, co, DEV)


def test_mfma_wgrad_matches_scalar_crosscheck():
    pc.check_backward("fwd_onegrid", False, DEV, wgrad_impl=1)


def test_all_invalid_depth_nan_losses():
    pc.check_all_invalid(DEV)


def test_render_without_depth():
    pc.check_render_nodepth(DEV)


def test_point_queries():
    pc.check_queries(DEV)


@pytest.mark.parametrize("name", list(pc.FWD_CASES))
def test_triplane_corner_indices_bit_exact(name):
    """R6: integer corner indices of the HIP gather == the oracle's, bit for bit (golden points + cell-edge adversaries)."""
    assert pc.check_corner_indices(DEV, name) > 10000


@pytest.mark.parametrize("name,one_grid,co,seed", [("mapping3_onegrid_esdf", True, False, 21),
                                                   ("mapping3_colorplanes_cosdf", False, True, 22)])
def test_three_mapping_iterations_match_reference(name, one_grid, co, seed):
    pc.check_mapping3(name, one_grid, co, seed, DEV)


@pytest.mark.parametrize("name,co", [("fwd_onegrid", False), ("fwd_onegrid", True), ("fwd_colorplanes", False), ("fwd_colorplanes", True)])
def test_ray_gradients_match_reference(name, co):
    pc.check_ray_gradients(name, co, DEV)


def test_render_without_depth_pose_gradients():
    pc.check_render_nodepth_pose_gradients(DEV)


@pytest.mark.parametrize("kind", ["hash", "dense"])
def test_grid_encoding_surface(kind):
    pc.check_grid_encoding(DEV, kind)


def test_device_sampler():
    pc.check_device_sampler(DEV)


def test_device_clock():
    pc.check_device_clock(DEV)


def test_quality_trajectory_matches_oracle_on_identical_batches():
    """Matched PSNR / depth-L1: fused path vs oracle trained on the batches the device drew."""
    pc.check_quality_trajectory(DEV, n_iters=30)


def test_render_maps_fast_path_and_render_img():
    pc.check_render_maps_fast_path(DEV)


@pytest.mark.parametrize("name,one_grid,co,seed", [("mapping3_onegrid_esdf", True, False, 21),
                                                   ("mapping3_colorplanes_cosdf", False, True, 22)])
@pytest.mark.parametrize("scatter", ["binned", "atomics"])
def test_three_fused_mapping_iterations_match_reference(name, one_grid, co, seed, scatter):
    pc.check_mapping3(name, one_grid, co, seed, DEV, compute="fused", scatter=scatter)


def test_binned_scatter_with_list_overflow():
    pc.check_mapping3("mapping3_onegrid_esdf", True, False, 21, DEV, compute="fused", scatter="binned", tile_capacity=8)


@pytest.mark.parametrize("hidden,one_grid", [(64, True), (64, False), (32, True)])
def test_random_scene_vs_oracle(hidden, one_grid):
    pc.check_oracle_random_scene(DEV, hidden=hidden, one_grid=one_grid, n_rays=96, S_d=96, S_r=32)


@pytest.mark.parametrize("n_rays,S_d,S_r", [(1, 4, 3), (5, 20, 13), (3, 1, 1), (257, 43, 21)])
def test_ragged_sizes_vs_oracle(n_rays, S_d, S_r):
    pc.check_oracle_random_scene(DEV, n_rays=n_rays, S_d=S_d, S_r=S_r, invalid_every=0 if n_rays < 10 else 5)


@pytest.mark.parametrize("hidden,one_grid,co", [(64, False, True), (64, True, False), (32, False, False)])
def test_fused_step_matches_autograd_path(hidden, one_grid, co):
    pc.check_fused_vs_autograd(DEV, hidden=hidden, one_grid=one_grid, co=co)


def test_fp16_planes_autograd_path_matches_fused_path():
    """Half-precision plane storage on the drop-in (autograd) path: the fp32 gradient sums reach the optimizer (``grad32``)."""
    pc.check_fused_vs_autograd(DEV, hidden=32, one_grid=True, co=False, plane_dtype="fp16")


@pytest.mark.parametrize("compute,absolute", [("autograd", False), ("fused", False), ("fused", True), ("fused", "quat")])
def test_loop_closure_pose_alignment(compute, absolute):
    pc.check_pose_alignment(DEV, compute, absolute)


def test_pose_alignment_falls_back_for_other_parameterisations():
    """A host whose matrix_from_tensor is neither the axis-angle nor the quaternion map keeps its own loop."""
    from mneslam_amd import hip_path
    other = lambda rot, trans: torch.eye(4)[None].repeat(rot.shape[0], 1, 1)
    assert hip_path.probe_axis_angle(other, torch.zeros(1, 4), torch.zeros(1, 3)) is None
    assert hip_path.probe_axis_angle(other, torch.tensor([[0.9, 0.1, 0.2, 0.3]]), torch.zeros(1, 3)) is None
    assert hip_path.probe_axis_angle(other, torch.tensor([[0.1, 0.2, 0.3]]), torch.zeros(1, 3)) is None
    assert hip_path.probe_axis_angle(other, torch.zeros(1, 6), torch.zeros(1, 3)) is None          # e.g. a 6-D rotation


def test_checkpoint_handoff_between_device_models(tmp_path):
    pc.check_checkpoint_handoff(DEV, tmp_path)


@pytest.mark.parametrize("compute", ["autograd", "fused"])
def test_loop_closure_distillation(compute):
    pc.check_distillation(DEV, compute)


def test_full_size_paths_agree_and_learn():
    """BASELINE-size workload (office0 planes 38.4 M params, 2150 rays x 128 samples): the fused path with
    binned scatter, the fused path with global atomics and the drop-in autograd path, driven with the
    SAME device-sampled batches and Philox jitter, must reach the same parameters after a few
    iterations (size-independent property: the three are different schedules of the same math), the
    loss must fall, and nothing may be NaN."""
    import bench
    from mneslam_amd import configs
    cfg = configs.bench_office0()
    dev = torch.device("cuda")
    finals, losses = {}, {}
    for mode in ("binned", "atomics", "binned+prefetch"):
        ag = bench.Agent(cfg, dev, seed=3, n_keyframes=4, path="fused", scatter=mode.split("+")[0])
        hist = []
        for it in range(6):
            # "+prefetch": the two HIP streams swap roles every iteration and the next batch is drawn early
            ag.step(prefetch=mode.endswith("prefetch") and it < 5)
            ag.fused.synchronize()
            hist.append(float(ag.fused.losses[0] + ag.fused.losses[1]))
        finals[mode] = [p.detach().clone() for lst in ag.model.all_planes for p in lst] + \
                       [p.detach().clone() for p in ag.model.decoder.parameters()]
        losses[mode] = hist
        del ag
        torch.cuda.empty_cache()
    # Adam with eps=1e-15 is scale-free: a cell whose gradient is at fp32-noise level takes a full lr-sized
    # step whose sign depends on the summation order, so a handful of elements may differ by O(lr);
    # everything else must agree to rounding.
    for a, b in zip(finals["binned"], finals["atomics"]):
        assert torch.isfinite(a).all()
        d = (a - b).abs()
        assert float(d.mean()) < 1e-7, "binned and atomic scatter disagree"
        assert float((d > 1e-4).float().mean()) < 1e-5 and float(d.max()) < 0.05
    for a, b in zip(finals["binned"], finals["binned+prefetch"]):
        d = (a - b).abs()
        assert float(d.mean()) < 1e-7 and float((d > 1e-4).float().mean()) < 1e-5, "prefetching / stream alternation changes the result"
    for x, y, z in zip(losses["binned"], losses["atomics"], losses["binned+prefetch"]):
        assert x == x and abs(x - y) <= 1e-3 * abs(y) and abs(x - z) <= 1e-3 * abs(x), "loss histories of the schedules diverge"
    # the mean absolute update is non-trivial (dense Adam moved the touched cells)
    assert float((finals["binned"][1] != 0).float().mean()) > 0.5


def test_full_size_properties():
    """BASELINE-size batch (2150 rays x 128 samples, office0 planes), size-independent properties of the path:
    the forward is deterministic bit for bit; sampled z are sorted inside [near, far]; rendered weights sum to <= 1,
    depth lies inside the sampled interval; the backward's tape holds exactly the samples that the reference's
    masks select (render window, Co-SLAM / ESLAM truncation masks), recomputed here with torch from raw and z."""
    import bench
    from mneslam_amd import configs
    cfg = configs.bench_office0()
    dev = torch.device("cuda")
    ag = bench.Agent(cfg, dev, seed=5, n_keyframes=4, path="fused", scatter="binned")
    for _ in range(3):
        ag.step()                                         # a few updates so that the SDF has sign changes
    fs, m = ag.fused, ag.model
    torch.cuda.synchronize()
    rays_o, rays_d, tgt_d = fs.rays_o.clone(), fs.rays_d.clone(), fs.tgt_d.clone()
    U = torch.rand(fs.R, fs.S, device=dev)
    outs = [m._render(rays_o, rays_d, None, tgt_d[:, None], u=U) for _ in range(2)]
    for a, b in zip(outs[0][:7], outs[1][:7]):
        assert torch.equal(a, b), "forward is not deterministic"
    rgb, depth, _, acc, var, z, raw = [t.detach() for t in outs[0][:7]]
    near, far = cfg["cam"]["near"], cfg["cam"]["far"]
    assert torch.all(z[:, 1:] >= z[:, :-1]) and float(z.min()) >= min(near, float((tgt_d - cfg["training"]["range_d"]).min())) - 1e-6
    assert float(acc.max()) <= 1.0 + 1e-5 and float(acc.min()) >= 0.0
    assert torch.all(depth >= z[:, 0] * acc - 1e-4) and torch.all(depth <= z[:, -1] * acc + 1e-4)
    assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0 + 1e-5 and float(var.min()) >= -1e-6
    # contributing samples of the LAST fused step vs the masks of the reference, from that step's own raw / z
    ag.step()
    torch.cuda.synchronize()
    z, sdf, td = fs.z_vals, fs.raw[..., 3], fs.tgt_d[:, None]
    tr, T = cfg["training"]["trunc"], cfg["model"]["truncation"]
    win = cfg["data"]["sc_factor"] * tr
    sign = (sdf[:, 1:] * sdf[:, :-1]) < 0
    first = torch.where(sign.any(1), sign.float().argmax(1), torch.zeros(fs.R, dtype=torch.long, device=dev))
    z_lim = z.gather(1, first[:, None]) + win
    has_d = td > 0
    front, back = z < (td - T), z > (td + T)
    center = (z > (td - 0.4 * T)) & (z < (td + 0.4 * T))
    eslam = has_d & (front | center | (~front & ~back & ~center))          # = every sample not behind the surface band
    expected = int(((z < z_lim) | eslam).sum())
    assert int(fs.tape_rows.item()) == expected, (int(fs.tape_rows.item()), expected)


def test_rebinding_tensors_between_fused_steps():
    """Planes / Adam moments are ordinary tensors that the host may re-bind between calls ("never cache
    data_ptr()"): a run in which every plane and moment is moved to fresh storage half-way must equal an undisturbed run."""
    import bench
    from mneslam_amd import configs
    cfg = configs.bench_office0()
    cfg["mapping"]["bound"] = [[-1.0, 1.0], [-1.2, 1.1], [-0.8, 0.9]]
    cfg["planes_res"] = {"coarse": 0.1, "fine": 0.05, "bound_dividable": 0.1}
    finals = []
    for rebind in (False, True):
        ag = bench.Agent(cfg, torch.device("cuda"), seed=9, n_keyframes=8, small=True, path="fused", scatter="binned")
        for it in range(4):
            if rebind and it == 2:
                torch.cuda.synchronize()
                for lst in ag.model.all_planes:
                    for p in lst:
                        p.data = p.data.clone(memory_format=torch.preserve_format)
                        st = ag.opt._state(p)
                        st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
            ag.step()
        torch.cuda.synchronize()
        finals.append([p.detach().clone() for lst in ag.model.all_planes for p in lst] +
                      [p.detach().clone() for p in ag.model.decoder.parameters()])
    for a, b in zip(*finals):
        d = (a - b).abs()
        assert torch.isfinite(a).all() and float(d.mean()) < 1e-7 and float((d > 1e-4).float().mean()) < 1e-4


@pytest.mark.parametrize("sample,n_samples_d,n_range_d", [(8192, 96, 32), (2048, 200, 56), (333, 5, 2)])
def test_batch_shapes_binned_vs_atomics(sample, n_samples_d, n_range_d):
    """Other batch shapes on the office0 planes (4x the rays; 256 samples per ray; 333+ rays x 7 samples), prefetching
    steps: the binned scatter (lists, counting sort, tile Adam) and the atomic scatter + streaming Adam are two
    schedules of the same sums and must agree; no list entry may be dropped."""
    import bench
    from mneslam_amd import configs
    cfg = configs.bench_office0()
    cfg["mapping"]["sample"] = sample
    cfg["training"]["n_samples_d"], cfg["training"]["n_range_d"] = n_samples_d, n_range_d
    finals = {}
    for mode in ("binned", "atomics"):
        ag = bench.Agent(cfg, torch.device("cuda"), seed=4, n_keyframes=6, path="fused", scatter=mode)
        for it in range(4):
            ag.step(prefetch=it < 3)
        ag.fused.check()
        torch.cuda.synchronize()
        finals[mode] = torch.cat([p.detach().reshape(-1) for lst in ag.model.all_planes for p in lst])
        assert torch.isfinite(ag.fused.losses[:2]).all()
        del ag
        torch.cuda.empty_cache()
    d = (finals["binned"] - finals["atomics"]).abs()
    assert float(d.mean()) < 1e-7 and float((d > 1e-4).float().mean()) < 1e-5


# ------------------------------------------------------------------------------------------------------------
# full-size parity: the bench path (device sampler, Philox jitter, two streams) vs the oracle on the same batch
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("warm", [0, 3])
def test_full_size_fused_step_vs_oracle(warm):
    """BASELINE configs[1] as wired: office0 planes (38.4 M parameters), 2150 rays x 128 samples.  One fused
    iteration of the bench path vs one oracle iteration on the batch the DEVICE drew (warm = 0: from the initial
    state, where the plane gradients can be read back from Adam's first moment; warm = 3: after three updates,
    continuing the optimizer state)."""
    from mneslam_amd import configs
    out = pc.check_fused_step_vs_oracle(DEV, configs.bench_office0(), n_keyframes=4, seed=3, warm_steps=warm)
    assert out["R"] == 2048 + 512 and out["S"] == 128 and out["contributing"] > 10000
    assert out["adam_stats"]["skipped_params"] > 0      # tiles no ray has reached yet are skipped, not swept (an untrained map's rays
    #                                                      run to `far`: most of the volume is touched within the first iterations)


@pytest.mark.parametrize("workload,hidden,rays", [("office0", 64, 2048), ("apartment", 32, 2048), ("scannet", 32, 2048),
                                                  ("scannet", 64, 1024), ("indoor", 32, 2048)])
def test_baseline_config_shapes_vs_oracle(workload, hidden, rays):
    """The other BASELINE.json configurations on ONE GPU at their FULL plane sizes, fused bench path vs the oracle on
    the device-drawn batch: C2 with the 2x64 decoders, C3 (Replica apartment agent: 62.4 M plane parameters), C4
    (ScanNet scene0000: colour planes, 69.3 M parameters, 460x620 frames, 117 samples; hidden 32 as configured and
    64 as BASELINE words it) and C5's shape (INS Indoor agent: 1045 samples per ray, far = 60 m; the full 2048-ray batch,
    the oracle evaluated in chunks of 256 rays -- oracle.mapping.forward_backward_chunked -- so that its autograd graph
    fits in host memory)."""
    from mneslam_amd import configs
    cfg = configs.WORKLOADS[workload][0](hidden)
    cfg["mapping"]["sample"] = rays
    out = pc.check_fused_step_vs_oracle(DEV, cfg, n_keyframes=4, seed=7, warm_steps=2,
                                        oracle_chunk=256 if workload == "indoor" else None)
    S = cfg["training"]["n_range_d"] + cfg["training"]["n_samples_d"]
    assert out["S"] == S and out["contributing"] > 0


@pytest.mark.parametrize("workload", ["apartment", "scannet", "indoor"])
def test_baseline_config_shapes_full_batch_properties(workload):
    """Full batches (2048 + share rays) of the C3 / C4 / C5 shapes: binned scatter + tile Adam and global atomics +
    streaming Adam are two schedules of the same sums and must agree after a few prefetching iterations; the losses
    must be finite and equal between the two schedules; no list entry may be dropped."""
    import bench
    from mneslam_amd import configs
    cfg = configs.WORKLOADS[workload][0](32)
    finals, hist = {}, {}
    for mode in ("binned", "atomics"):
        ag = bench.Agent(cfg, torch.device("cuda"), seed=4, n_keyframes=5, path="fused", scatter=mode)
        h = []
        for it in range(5):
            ag.step(prefetch=it < 4)
            ag.fused.synchronize()
            h.append(float(ag.fused.losses[0] + ag.fused.losses[1]))
        ag.fused.check()
        torch.cuda.synchronize()
        finals[mode] = torch.cat([p.detach().reshape(-1) for lst in ag.model.all_planes for p in lst])
        hist[mode] = h
        del ag
        torch.cuda.empty_cache()
    d = (finals["binned"] - finals["atomics"]).abs()
    assert float(d.mean()) < 1e-7 and float((d > 1e-4).float().mean()) < 1e-5
    assert all(x == x and x < 1e4 for x in hist["binned"])          # finite (every iteration draws a different batch)
    for x, y in zip(hist["binned"], hist["atomics"]):
        assert abs(x - y) <= 1e-3 * abs(y)


def test_rccl_branch_single_rank():
    """The RCCL (backend "nccl") branch of the multi-agent plumbing on a real GPU, world size 1: process-group set-up of
    dist.init_agents, a collective on a device tensor, the pose gather and bench.py's timing rule.  (8-GPU runs are the
    driver's; the 2-agent logic is covered on CPU with gloo.)"""
    import os
    import torch.distributed as dist
    from mneslam_amd import dist as mdist
    env = {k: os.environ.get(k) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    try:
        world = int(os.environ["WORLD_SIZE"])
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=world, device_id=dev)
        g = torch.full((6208,), 3.0, device=dev)
        dist.all_reduce(g)                                   # 24.8 KB decoder-gradient buffer through RCCL
        mdist.allreduce_mean_(g)
        assert torch.equal(g, torch.full_like(g, 3.0))
        poses = mdist.gather_keyframe_poses(torch.eye(4, device=dev)[None], torch.arange(1))
        assert len(poses) == 1 and poses[0][0].shape == (1, 4, 4)
        assert mdist.max_over_ranks(0.25, dev) == pytest.approx(0.25)
        dist.barrier()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# ------------------------------------------------------------------------------------------------------------
# NS-a: the hash-grid wiring (EXTENSION; parity unpinned -- tinycudann is not in the reference tree, the checker is the
# build's own oracle/hashgrid.py + oracle.scene_rep.OracleHashScene)
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hash_size,hidden,warm", [(12, 32, 0), (14, 64, 0), (14, 64, 3)])
def test_hash_grid_fused_step_vs_oracle(hash_size, hidden, warm):
    """Reduced tables (heavy collisions on the hashed levels), 256 + 100 rays x 43 samples: table indices bit-exact,
    forward / losses / gradients / post-Adam state against the oracle on the device-drawn batch."""
    cfg = pc.hash_test_config(hash_size=hash_size, hidden=hidden)
    out = pc.check_hash_fused_step_vs_oracle(DEV, cfg, n_keyframes=3, seed=5, warm_steps=warm, small=True)
    assert out["S"] == 43 and out["touched_entries"] > 1000


def test_hash_grid_headline_config_vs_oracle():
    """BASELINE configs[1] in its literal form: 16 levels, T = 2^19 (10.49 M table floats), 2x64 MLPs, 2048 + 512 rays x
    128 samples, 1200x680 frames -- one fused iteration vs the oracle on the same batch."""
    from mneslam_amd import configs
    cfg = configs.bench_office0_hash()
    out = pc.check_hash_fused_step_vs_oracle(DEV, cfg, n_keyframes=4, seed=3, warm_steps=2, small=False)
    assert out["R"] == 2048 + 512 and out["S"] == 128


def test_hash_table_update_is_bit_reproducible_at_the_headline_size():
    """BASELINE configs[1] literal (T = 2^19 x 16 levels, 2560 rays x 128 samples): the table update of a pipeline iteration --
    run beside the weight-gradient kernel -- repeated three times stand-alone on copies of the pre-step state gives the table,
    exp_avg and exp_avg_sq bit for bit (order-independent 64-bit fixed-point sums; a size-independent property of the path)."""
    from mneslam_amd import configs
    out = pc.check_hash_update_bit_reproducible(DEV, configs.bench_office0_hash(), n_keyframes=4, seed=3, warm_steps=3, small=False, repeats=3)
    assert out["R"] == 2048 + 512 and out["moved"] > 100000


@pytest.mark.parametrize("hidden,co", [(32, False), (64, True)])
def test_hash_scene_api_vs_oracle(hidden, co):
    """The whole JointEncoding surface of the hash-grid model -- render_rays (with / without depth), forward + autograd
    backward (table and decoder gradients), render_maps, render_img (chunked and whole), query_sdf / query_color /
    query_color_sdf / run_network -- against oracle.scene_rep.OracleHashScene."""
    cfg = pc.hash_test_config(hash_size=14, hidden=hidden)
    cfg["training"]["n_samples"] = 64
    pc.check_hash_scene_api(DEV, cfg, n_rays=300, co=co)


@pytest.mark.parametrize("warm", [0, 2])
def test_dense_grid_configs0_fused_step_vs_oracle(warm):
    """BASELINE.json configs[0] in its as-north-star form on the HIP path: 16^3 dense grid (4 levels x 2 features) + 2x32
    MLPs, 512 rays x 64 samples: one fused mapping iteration against the oracle on the device-drawn batch."""
    out = pc.check_hash_fused_step_vs_oracle(DEV, pc.dense_grid_config(), n_keyframes=4, seed=3, warm_steps=warm, small=False)
    assert out["S"] == 64 and out["R"] >= 512


def test_dense_grid_scene_api_vs_oracle():
    cfg = pc.dense_grid_config()
    cfg["training"]["n_samples"] = 48
    pc.check_hash_scene_api(DEV, cfg, n_rays=200)


@pytest.mark.parametrize("compute", ["fused", "autograd"])
def test_loop_closure_pose_alignment_on_the_hash_model(compute):
    """R13 on the north-star encoding: the pose loop of loop closure differentiates the hash-grid render w.r.t. its rays."""
    pc.check_pose_alignment_hash(DEV, compute)


def test_hash_iteration_on_more_than_512k_rows():
    """8192 rays x 128 samples = 1.1 M tape rows on the headline hash grid (T = 2^19, 2x64): the slice kernel's chunk groups."""
    from mneslam_amd import configs
    cfg = configs.bench_office0_hash()
    cfg["mapping"]["sample"] = 8192
    out = pc.check_hash_large_batch(DEV, cfg)
    assert out["rows"] > (1 << 20)


def test_hash_grid_training_learns():
    """The hash-grid iteration trains: PSNR rises and depth L1 falls over 150 prefetching iterations at full size."""
    import bench
    from mneslam_amd import configs
    ag = bench.Agent(configs.bench_office0_hash(), torch.device("cuda"), seed=1, n_keyframes=5, path="fused")
    ag.step()
    p0, d0 = ag.quality()
    for it in range(150):
        ag.step(prefetch=it < 149)
    p1, d1 = ag.quality()
    assert math.isfinite(p1) and p1 > p0 + 3.0 and d1 < 0.5 * d0, (p0, d0, p1, d1)


# ------------------------------------------------------------------------------------------------------------
# NS-b / BASELINE configs[4]: half-precision plane STORAGE (fp32 accumulate) and the hipGraph-captured iteration (EXTENSIONS)
# ------------------------------------------------------------------------------------------------------------
def _small_bench_cfg(plane_dtype="fp32"):
    from mneslam_amd import configs
    cfg = configs.bench_office0()
    cfg["mapping"]["bound"] = [[-1.0, 1.0], [-1.2, 1.1], [-0.8, 0.9]]
    cfg["planes_res"] = {"coarse": 0.1, "fine": 0.05, "bound_dividable": 0.1}
    cfg["grid"]["plane_dtype"] = plane_dtype
    return cfg


def _graph_vs_eager(cfg, mode, n_steps=9, small=True, n_keyframes=8):
    """The recorded iteration (one hipGraphLaunch per step, iteration / Adam step from the device clock) against the same
    steps launched one by one: identical ray batches and z samples (bit for bit), same parameters up to the summation
    order of the plane-gradient lists."""
    import bench
    out = {}
    for how in ("eager", mode):
        ag = bench.Agent(cfg, torch.device("cuda"), seed=6, n_keyframes=n_keyframes, small=small, path="fused", scatter="binned")
        ag.fused.use_graph = None if how == "eager" else how
        for it in range(n_steps):
            ag.step(prefetch=it < n_steps - 1)
        ag.fused.check()
        torch.cuda.synchronize()
        assert (len(ag.fused._graphs) > 0 and all(ag.fused._graphs.values())) == (how != "eager")
        out[how] = (ag.fused.idx.clone(), ag.fused.z_vals.clone(), ag.fused.losses.clone(),
                    [p.detach().float().clone() for lst in ag.model.all_planes for p in lst]
                    + [p.detach().clone() for p in ag.model.decoder.parameters()],
                    ag.opt._state(ag.fused.planes[0])["step"], ag.fused.iteration)
        del ag
        torch.cuda.empty_cache()
    e, g = out["eager"], out[mode]
    assert torch.equal(e[0], g[0]) and torch.equal(e[1], g[1])
    assert e[4] == g[4] == n_steps and e[5] == g[5] == n_steps
    assert torch.allclose(e[2], g[2], rtol=1e-3, atol=1e-6, equal_nan=True)
    for a, b in zip(e[3], g[3]):
        d = (a - b).abs()
        assert torch.isfinite(b).all() and float(d.mean()) < 2e-6 and float((d > 1e-4).float().mean()) < 2e-4


@pytest.mark.parametrize("mode", ["two_stream", "one_stream"])
@pytest.mark.parametrize("plane_dtype", ["fp32", "fp16"])
def test_graph_replay_matches_eager_launches(plane_dtype, mode):
    _graph_vs_eager(_small_bench_cfg(plane_dtype), mode)


@pytest.mark.parametrize("workload,rays,warm,chunk", [("office0", 2048, 0, None), ("office0", 2048, 3, None), ("indoor", 2048, 2, 256)])
def test_fp16_plane_storage_step_vs_oracle(workload, rays, warm, chunk):
    """Full plane sizes, planes stored ONLY in fp16: forward, losses, gradients (fp32 sums) and the post-Adam parameters --
    round_to_nearest(Adam(float(p16))) with fp32 moments -- against the oracle evaluated at the same stored values."""
    from mneslam_amd import configs
    cfg = configs.WORKLOADS[workload][0]()
    cfg["mapping"]["sample"] = rays
    cfg["grid"]["plane_dtype"] = "fp16"
    out = pc.check_fused_step_vs_oracle(DEV, cfg, n_keyframes=4, seed=9, warm_steps=warm, oracle_chunk=chunk)
    assert out["contributing"] > 0


def test_configs4_indoor_fp16_graph_captured_iteration():
    """BASELINE.json configs[4] as worded, one agent of it on one GPU: INS Indoor agent bounds (S = 21 + 1024 samples per
    ray, far 60 m, planes 0.24 / 0.06 m), fp16 feature storage + fp32 accumulate, hipGraph-captured mapping iteration
    (configs.WORKLOADS['indoor_fp16']; bench.py reports it as variants.indoor_fp16_graph).  Replay vs eager launches of
    the same steps at the full batch; the per-iteration numerics of this storage are pinned against the oracle by
    test_fp16_plane_storage_step_vs_oracle[indoor]."""
    from mneslam_amd import configs
    cfg = configs.WORKLOADS["indoor_fp16"][0]()
    for mode in ("two_stream", "one_stream"):
        _graph_vs_eager(cfg, mode, n_steps=6, small=False, n_keyframes=5)


def test_fp16_plane_storage_trains_like_fp32():
    """Matched quality of the extension: 200 iterations on the same device-drawn batches, fp16 vs fp32 plane storage:
    PSNR within 0.5 dB and depth L1 within 10 % at the end (the planes are N(0, 0.01^2)-scale features: fp16 keeps 11
    bits of each; the update has no fp32 master to accumulate sub-ulp steps in)."""
    import bench
    from mneslam_amd import configs
    res = {}
    for ps in ("fp32", "fp16"):
        cfg = configs.bench_office0()
        cfg["grid"]["plane_dtype"] = ps
        ag = bench.Agent(cfg, torch.device("cuda"), seed=2, n_keyframes=5, path="fused")
        hist = []
        for it in range(200):
            ag.step(prefetch=it < 199)
            if it >= 180:
                hist.append(ag.quality())
        res[ps] = (sum(h[0] for h in hist) / len(hist), sum(h[1] for h in hist) / len(hist))
        del ag
        torch.cuda.empty_cache()
    assert abs(res["fp16"][0] - res["fp32"][0]) < 0.5 and abs(res["fp16"][1] - res["fp32"][1]) < 0.1 * res["fp32"][1], res


def test_forced_split_lists_and_capped_ray_lds(monkeypatch):
    """The two load-balance mechanisms that only engage on large shapes -- tile lists split over several workgroups
    (tile_adam.hip) and the training ray kernel's LDS sample cap with its long-ray pass (render.hip) -- forced on a small
    scene and checked against the oracle like every other fused step."""
    from mneslam_amd import configs
    monkeypatch.setenv("MNE_TILE_SPLIT_MIN", "16")
    monkeypatch.setenv("MNE_HOT_LDS_SAMPLES", "48")
    cfg = configs.bench_office0()
    cfg["mapping"]["bound"] = [[-1.0, 1.0], [-1.2, 1.1], [-0.8, 0.9]]
    cfg["planes_res"] = {"coarse": 0.2, "fine": 0.1, "bound_dividable": 0.2}
    cfg["mapping"]["sample"] = 512
    out = pc.check_fused_step_vs_oracle(DEV, cfg, n_keyframes=4, seed=11, warm_steps=2, small=True)
    assert out["contributing"] > 1000


# ------------------------------------------------------------------------------------------------------------
# 8e on real hardware as far as one GPU allows: the multi-agent forms of the plane update (tile_adam_kernel<1>, <2>),
# two agents = two threads of this process on the same device, the exchange done by device-to-device copies
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("geometry,world", [("lattice", 2), ("apartment", 2), ("lattice", 3), ("scannet", 4)])
def test_two_agents_binned_overlap_on_one_device(monkeypatch, geometry, world):
    """geometry "apartment": BASELINE configs[2] as worded at full plane size (the two slabs bench.py --split gives its ranks);
    "scannet", 4 agents: configs[3] as worded -- ScanNet scene0000 with colour planes split 4-way, agents 1 and 2 INTERIOR (two
    neighbours each: two rectangles per plane in mne_tile_grad_export / mne_tile_adam_shared); "lattice", 3: the
    small chain with one interior agent.
    parity_cases.run_overlap_agent with all agents on cuda:0: mne_tile_grad_export / mne_tile_adam_shared on the GPU,
    FusedStep(overlap_peers, shared_decoder) on its two streams; what torch.distributed would carry (the send / recv
    buffers of the shared cells, the decoder-gradient mean) is copied between the agents' buffers under a barrier."""
    import threading
    from mneslam_amd import dist as mdist
    bar, slots, local = threading.Barrier(world), [None] * world, threading.local()

    class ThreadComm:
        lock = threading.Lock()

        def all_gather(self, obj):
            slots[local.rank] = obj
            bar.wait()
            out = list(slots)
            bar.wait()
            return out
    comm = ThreadComm()

    def exchange(peers, send, recv):
        torch.cuda.current_stream().synchronize()                 # my export has finished
        everyone = comm.all_gather((list(peers), send))
        for k, peer in enumerate(peers):                          # what the peer exported for ME
            their_peers, their_send = everyone[peer]
            recv[k].copy_(their_send[their_peers.index(local.rank)])
        torch.cuda.current_stream().synchronize()
        bar.wait()                                                # the peers have read my send buffers

    def allreduce_mean(buf):
        torch.cuda.current_stream().synchronize()
        every = comm.all_gather(buf)
        mean = sum(every[1:], every[0].clone()) / world
        torch.cuda.current_stream().synchronize()
        bar.wait()
        buf.copy_(mean)
        return buf

    def allreduce_sum_into(send, recv):
        torch.cuda.current_stream().synchronize()
        every = comm.all_gather(send)
        total = sum(every[1:], every[0].clone())                  # (rank order: the same bits on every agent)
        recv.copy_(total)
        torch.cuda.current_stream().synchronize()
        bar.wait()                                                # everyone has read my send buffer
        send.zero_()
    monkeypatch.setattr(mdist, "exchange_buffers", exchange)
    monkeypatch.setattr(mdist, "allreduce_mean_", allreduce_mean)
    monkeypatch.setattr(mdist, "allreduce_sum_into", allreduce_sum_into)
    errors = []

    def agent(rank):
        local.rank = rank
        try:
            with torch.cuda.device(0):
                pc.run_overlap_agent(rank, DEV, comm, geometry=geometry, world=world)
        except BaseException as e:          # noqa: BLE001 -- reported by the main thread
            import traceback
            errors.append((rank, e, traceback.format_exc()[-1500:]))
            bar.abort()
    threads = [threading.Thread(target=agent, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_sample_z_frame_sized_batch_counts():
    """whole-frame batches: the striped counts reduction of mne_sample_z"""
    pc.check_sample_z_frame_counts(DEV)


# (last in the module: N child processes on the device; nothing that initialises RCCL runs behind them)
@pytest.mark.parametrize("form,world", [("self", 2), ("split-scannet", 4), ("split-indoor", 8)])
def test_multi_agent_processes_share_the_gpu(tmp_path, form, world):
    """The process-per-agent paths of bench.py as REAL processes on the HIP library (round 6): N ranks share this box's GPU(s) with
    gloo as the transport (MNE_SHARE_GPUS=1, dist.init_agents: RCCL refuses two ranks on one device) -- rendezvous, barrier-bracketed
    timing, max over ranks, ONE JSON line; "self": N independent agents + the side records (decoder-gradient all-reduce; BASELINE
    configs[2] as worded: one scene, two overlapping slabs); "split-*": configs[3] / configs[4] geometry, interior agents exchange the
    overlap rectangles with TWO neighbours (batch_isend_irecv on device buffers, tile_adam_kernel<1> / <2>) and the planes without the
    slab axis are reduced over all agents.  Small scenes (functional, a few steps); the full-size runs of the same command lines are
    recorded in profiles/r06_shared_gpu_ranks.txt.  What this cannot show is RCCL itself and any rate."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MNE_EMULATED_LIBRARY")}
    env.update(MNE_SHARE_GPUS="1", PYTHONPATH=repo, MNE_SIDE_RECORD_LIMIT_S="600")
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "2", "--cpu-iters", "0",
           "--small", "--rays", "256", "--keyframes", "3"]
    if form != "self":
        cmd += ["--split", "--config", form.split("-")[1], "--no-variants"]
    out = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    if out.returncode != 0:                    # (one retry: N fresh processes + a TCP rendezvous on a box that has just run the rest of the suite)
        first = out.stderr[-1500:]
        out = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, first + "\n---- retry ----\n" + out.stderr[-1500:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["config"]["ranks_seen"] == world and d["config"]["collective_backend"] == "gloo" and d["value"] > 0
    assert "RANKS SHARE GPUs" in d["data"] and "DRY RUN" not in d["data"]
    per = d["per_rank"]
    assert len(per) == world and all(r["it_per_s"] > 0 and math.isfinite(r["psnr_last_iter"]) and math.isfinite(r["depth_l1_last_iter"]) for r in per)
    if form == "self":
        assert "no data-path collective" in d["config"]["parallelism"]
        side = d["variants"]
        assert side["share_decoder"].get("value", 0) > 0, side
        worded = side["as_worded"]
        assert worded.get("value", 0) > 0 and worded["baseline_config"] == "configs[2]", worded
        assert len(worded["per_rank"]) == 2 and all(r["overlap_exchange_bytes_per_iter"] > 0 and math.isfinite(r["psnr_last_iter"]) for r in worded["per_rank"])
    else:
        assert d["config"]["workload"].endswith(f"_scene_split{world}_SMALL") and "all agents" in d["config"]["parallelism"]
        ends = (per[0]["overlap_exchange_bytes_per_iter"] + per[-1]["overlap_exchange_bytes_per_iter"]) / 2
        assert ends > 0
        for r in per[1:-1]:            # two neighbours: more to exchange than an end slab
            assert 1.2 * ends < r["overlap_exchange_bytes_per_iter"] < 2.5 * ends, [q["overlap_exchange_bytes_per_iter"] for q in per]
