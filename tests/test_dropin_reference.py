"""Drop-in proof with the REFERENCE'S OWN callers (build container only: needs /root/reference).

The reference's unmodified ``mp_slam.mapper.Mapper`` / ``model.keyframe.KeyFrameDatabase`` / ``model.scene_rep.JointEncoding``
are imported through tests/golden/ref_harness.py (stubs for the third-party modules that are not installed) and
combined with this repository's classes the way INTEGRATION.md describes.  Kernels run through the host emulator
(tests/hostemu), results are compared with the golden fixtures the reference itself produced.
Skipped wherever the reference tree is absent (e.g. on the GPU box)."""
import contextlib
import os
import random
import sys
import threading
import types
from unittest import mock

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "hostemu"))
sys.path.insert(0, os.path.join(HERE, "golden"))

REF = os.environ.get("MNESLAM_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
# (every case runs by default, including the end-to-end ones: the reference's own Mapper driving this repository's model
# through first_frame_mapping / mapping_optimize against the golden post-step parameters)

from helpers import DEC_KEYS, assert_close, load_golden, n_plane_sets  # noqa: E402
import parity_cases as pc  # noqa: E402
from mneslam_amd import _lib, configs, slam_glue  # noqa: E402
from mneslam_amd.model.keyframe import KeyFrameDatabase as RepoKeyFrameDatabase  # noqa: E402
from mneslam_amd.model.scene_rep import JointEncoding as RepoJointEncoding  # noqa: E402
from mneslam_amd.mp_slam import mapper as repo_mapper  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def emulator_library():
    import build_emu
    _lib.unload()
    _lib.load(build_emu.build())
    torch.set_num_threads(2)
    yield
    _lib.unload()


@pytest.fixture(scope="module")
def ref():
    import ref_harness
    ref_harness.install()
    from model.keyframe import KeyFrameDatabase
    from model.scene_rep import JointEncoding
    from mp_slam.mapper import Mapper
    return types.SimpleNamespace(Mapper=Mapper, KeyFrameDatabase=KeyFrameDatabase, JointEncoding=JointEncoding)


class _Counter:
    def __init__(self, v=0):
        self.value = v


def _fake_slam(cfg, model, opt, kfdb, H, W, direction, tmp_path, model_shared=None):
    """The fields the reference's Mapper reads (mp_slam/mapper.py:12-50) + the MNESLAM methods its loops call."""
    cfg.setdefault("data", {}).update(output=str(tmp_path), exp_name="dropin")
    cfg.setdefault("mesh", {"voxel_eval": 0.05})
    n_kf = 8
    video = types.SimpleNamespace(keyframe=kfdb, map_counter=_Counter(0), counter=_Counter(1),
                                  timestamp=torch.arange(n_kf, dtype=torch.float32),
                                  get_lock=lambda: contextlib.nullcontext(),
                                  get_all_pose=lambda device: torch.eye(4).repeat(n_kf, 1, 1))
    slam = types.SimpleNamespace(
        config=cfg, model=model, model_shared=model_shared if model_shared is not None else model, map_optimizer=opt,
        device=torch.device("cpu"), video=video,
        dataset=types.SimpleNamespace(H=H, W=W, fx=30.0, fy=30.0, cx=15.0, cy=11.0, rays_d=direction),
        tracking_idx=None, mapping_idx=None, mapping_first_frame=torch.zeros(1, dtype=torch.int32), keyframe_dict=[],
        mesher=None, all_agent_bounds=None, keyframe_dict_lock=threading.Lock(), descriptor_db_lock=threading.Lock(),
        rank=0, world_size=1,
        get_loss_from_ret=lambda ret, **kw: slam_glue.get_loss_from_ret(cfg, ret, **kw),
        select_samples=slam_glue.select_samples,
        save_imgs=mock.MagicMock(), save_latest_checkpoint=mock.MagicMock(), save_mesh=mock.MagicMock())
    return slam


def _mapping3_setup(name, one_grid, co, seed, kf_cls, tmp_path):
    g = load_golden(name)
    cfg = configs.small_test_config(one_grid=one_grid, is_co_sdf=co)
    cfg["mapping"].update(sample=64, min_pixels_cur=10, iters=3, n_pixels=0.25)
    H, W, n_save = int(g["H"]), int(g["W"]), int(g["n_save"])
    m = pc.model_from_golden(g, cfg, "cpu", prefix="init.").train()
    opt = slam_glue.create_optimizer(m, cfg)               # FusedAdam over the reference's param groups
    direction = torch.from_numpy(g["direction"])
    frames = [dict(frame_id=k, c2w=torch.from_numpy(g[f"frame{k}.c2w"]), rgb=torch.from_numpy(g[f"frame{k}.rgb"]),
                   depth=torch.from_numpy(g[f"frame{k}.depth"]), direction=direction) for k in range(4)]
    random.seed(seed)
    torch.manual_seed(seed)
    kfdb = kf_cls(cfg, H, W, 8, n_save, torch.device("cpu"))
    for k in range(3):
        kfdb.add_keyframe(frames[k], k + 1)
    assert_close(kfdb.rays[:3], g["kf.rays"], rtol=0, atol=0, what="keyframe ray DB")
    slam = _fake_slam(cfg, m, opt, kfdb, H, W, direction, tmp_path)
    return g, cfg, m, frames, slam


def _assert_final(g, m):
    for s in range(n_plane_sets(g, "init.")):
        for l in range(2):
            assert_close(m.all_planes[s][l].detach().cpu(), g[f"final.plane_{s}_{l}"], rtol=1e-3, atol=1e-4,
                         what=f"plane {s},{l} after 3 iterations")
    sd = dict(m.decoder.named_parameters())
    for k in DEC_KEYS:
        assert_close(sd[k].detach().cpu(), g[f"final.dec.{k}"], rtol=1e-3, atol=1e-4, what=f"decoder {k}")


@pytest.mark.parametrize("kf_side", ["reference", "repo"])
def test_reference_mapper_drives_repo_model(ref, tmp_path, kf_side):
    """Integration level 1: the reference's UNMODIFIED Mapper.mapping_optimize (its own loop, its own ray assembly)
    on this repository's JointEncoding + FusedAdam reaches the parameters the reference reached with its own model."""
    kf_cls = ref.KeyFrameDatabase if kf_side == "reference" else RepoKeyFrameDatabase
    g, cfg, m, frames, slam = _mapping3_setup("mapping3_onegrid_esdf", True, False, 21, kf_cls, tmp_path)
    mapper = ref.Mapper(cfg, slam)
    assert type(mapper).mapping_optimize.__module__ == "mp_slam.mapper"
    poses = torch.stack([f["c2w"] for f in frames])
    random.seed(22)
    torch.manual_seed(22)
    mapper.mapping_optimize(frames[3], poses)
    _assert_final(g, m)


def test_bound_mapper_class_structure(ref):
    """bind(host Mapper): the mixin sits in front of the reference's class and replaces exactly the two training loops."""
    Bound = repo_mapper.bind(ref.Mapper, compute="fused", sampler="host")
    assert Bound.__mro__[1] is repo_mapper.FusedMappingMixin and Bound.__mro__[2] is ref.Mapper
    for name in ("run", "final_run", "handle_loop_closure", "bound_based_fusion",
                 "save_keyframe_data_atomic", "__init__"):
        assert getattr(Bound, name) is getattr(ref.Mapper, name), f"{name} must stay the host's"
    for name in ("mapping_optimize", "first_frame_mapping", "load_foreign_model"):
        assert getattr(Bound, name) is getattr(repo_mapper.FusedMappingMixin, name)
    # load_foreign_model: the host's own method unless the SLAM object carries a running map exchange (process group)
    import types
    calls = []
    fake = types.SimpleNamespace(slam=types.SimpleNamespace())
    orig = ref.Mapper.load_foreign_model
    try:
        ref.Mapper.load_foreign_model = lambda self, other_rank: calls.append(other_rank) or "host"
        obj = Bound.__new__(Bound)
        obj.slam = fake.slam
        assert obj.load_foreign_model(3) == "host" and calls == [3]
    finally:
        ref.Mapper.load_foreign_model = orig
    with pytest.raises(ValueError):
        repo_mapper.bind(ref.Mapper, compute="autograd", sampler="device")


def test_fused_mixin_over_reference_mapper(ref, tmp_path):
    """Integration level 2: bind(reference Mapper) replaces mapping_optimize by the fused iteration (host RNG draws in
    the reference's order); everything else of the class is the reference's."""
    g, cfg, m, frames, slam = _mapping3_setup("mapping3_onegrid_esdf", True, False, 21, ref.KeyFrameDatabase, tmp_path)
    Bound = repo_mapper.bind(ref.Mapper, compute="fused", sampler="host")
    mapper = Bound(cfg, slam)
    assert Bound.__mro__[1] is repo_mapper.FusedMappingMixin and Bound.__mro__[2] is ref.Mapper
    for name in ("run", "final_run", "handle_loop_closure", "bound_based_fusion", "save_keyframe_data_atomic"):
        assert getattr(Bound, name) is getattr(ref.Mapper, name), f"{name} must stay the host's"
    poses = torch.stack([f["c2w"] for f in frames])
    random.seed(22)
    torch.manual_seed(22)
    mapper.mapping_optimize(frames[3], poses)
    _assert_final(g, m)


def test_first_frame_mapping_keeps_host_bookkeeping(ref, tmp_path):
    """The fused first-frame loop hands over to the reference's own first_frame_mapping (zero iterations) for the
    bookkeeping of mp_slam/mapper.py:91-116: first keyframe, keyframe_dict entry, flag, dumps, pose files."""
    g = load_golden("mapping3_onegrid_esdf")
    cfg = configs.small_test_config(one_grid=True, is_co_sdf=False)
    cfg["mapping"].update(sample=64, min_pixels_cur=10, iters=3, n_pixels=0.25)
    H, W, n_save = int(g["H"]), int(g["W"]), int(g["n_save"])
    finals = {}
    for kind in ("host-loop", "fused-loop"):
        m = pc.model_from_golden(g, cfg, "cpu", prefix="init.").train()
        opt = slam_glue.create_optimizer(m, cfg)
        direction = torch.from_numpy(g["direction"])
        frame0 = dict(frame_id=0, c2w=torch.from_numpy(g["frame0.c2w"]), rgb=torch.from_numpy(g["frame0.rgb"]),
                      depth=torch.from_numpy(g["frame0.depth"]), direction=direction)
        kfdb = ref.KeyFrameDatabase(cfg, H, W, 8, n_save, torch.device("cpu"))
        slam = _fake_slam(cfg, m, opt, kfdb, H, W, direction, tmp_path / kind)
        cls = ref.Mapper if kind == "host-loop" else repo_mapper.bind(ref.Mapper, compute="fused", sampler="host")
        mapper = cls(cfg, slam)
        random.seed(3)
        torch.manual_seed(3)
        mapper.first_frame_mapping(frame0, n_iters=2)
        assert float(kfdb.rays[0].abs().sum()) > 0 and len(kfdb.frame_ids) == 1, "first keyframe was not stored"
        assert len(slam.keyframe_dict) == 1 and slam.keyframe_dict[0]["frame_id"] == 0
        assert int(slam.mapping_first_frame[0]) == 1 and slam.video.map_counter.value == 1
        assert slam.save_imgs.call_count == 1 and slam.save_latest_checkpoint.call_count == 1 and slam.save_mesh.call_count == 1
        out_dir = slam_glue.agent_dir(cfg, 0)
        assert os.path.exists(os.path.join(out_dir, "key_est_poses.npy"))
        poses, stamps = slam_glue.load_keyframe_poses(cfg, 0)            # the host's writer, this repo's reader
        assert poses.shape == (1, 4, 4) and stamps.shape == (1,)
        finals[kind] = ([p.detach().clone() for lst in m.all_planes for p in lst] +
                        [p.detach().clone() for p in m.decoder.parameters()], kfdb.rays[0].clone())
    for a, b in zip(finals["host-loop"][0], finals["fused-loop"][0]):
        assert_close(b, a, rtol=1e-3, atol=1e-4, what="fused vs host first-frame loop")
    assert torch.equal(finals["host-loop"][1], finals["fused-loop"][1]), "both paths must consume the RNG identically"


def test_keyframe_database_matches_reference(ref):
    """Seeded call-for-call equality of the repository's KeyFrameDatabase with the reference's (all public methods)."""
    cfg = {"cam": {"depth_trunc": 3.0, "fx": 60.0, "fy": 60.0, "cx": 59.0, "cy": 33.0, "H": 68, "W": 120}}
    H, W = 68, 120
    gen = torch.Generator().manual_seed(0)
    frames = [{"direction": torch.randn(H, W, 3, generator=gen), "rgb": torch.rand(H, W, 3, generator=gen),
               "depth": torch.rand(H, W, generator=gen) * 4} for _ in range(6)]
    poses = [torch.eye(4) for _ in range(8)]
    for i, p in enumerate(poses):
        p[:3, 3] = torch.tensor([0.1 * i, 0.0, 0.0])
    outs = []
    for cls in (ref.KeyFrameDatabase, RepoKeyFrameDatabase):
        random.seed(5), np.random.seed(5), torch.manual_seed(5)
        kf = cls(cfg, H, W, 8, 400, "cpu")
        for k in range(5):
            kf.add_keyframe(frames[k], k + 1, filter_depth=(k == 3))
        got = [kf.sample_global_rays(300), kf.sample_global_keyframe(2)]
        kf.del_keyframe(2)
        got += [kf.sample_global_rays(100), kf.sample_global_keyframe(10),
                kf.sample_overlap_keyframe(frames[5], 3, poses, 2, dataset=types.SimpleNamespace(H=H, W=W))]
        outs.append(got + [(kf.rays.clone(), kf.frame_ids)])
    for x, y in zip(*outs):
        for u, v in zip(x, y):
            assert torch.equal(torch.as_tensor(u).double(), torch.as_tensor(v).double())


def test_ray_helpers_match_reference(ref):
    """get_rays / normalize_3d_coordinate of this repository (own implementations) == the reference's, bit for bit."""
    from model import utils as ref_utils
    from mneslam_amd.model import utils as repo_utils
    c2w = torch.randn(4, 4, generator=torch.Generator().manual_seed(1))
    for H, W, fx, fy, cx, cy in [(680, 1200, 600.0, 600.0, 599.0, 339.0), (460, 620, 577.0, 578.0, 308.0, 232.0), (12, 16, 16.0, 16.0, 8.0, 6.0)]:
        a, b = ref_utils.get_rays(H, W, fx, fy, cx, cy, c2w, "cpu"), repo_utils.get_rays(H, W, fx, fy, cx, cy, c2w, "cpu")
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    p = torch.randn(100, 3)
    bound = torch.tensor([[-3.0, 3.02], [-4.0, 2.52], [-2.0, 2.52]])
    assert torch.equal(ref_utils.normalize_3d_coordinate(p.clone(), bound), repo_utils.normalize_3d_coordinate(p, bound))


def test_state_dict_keys_and_plane_shapes_match_reference(ref):
    """same state_dict keys / shapes (incl. the duplicated color_net / sdf_net aliases and the empty
    embedpos_fn.params) and the same logical plane shapes as the reference's JointEncoding, for both grid modes."""
    for one_grid in (True, False):
        cfg = configs.small_test_config(one_grid=one_grid)
        bb = torch.tensor(cfg["mapping"]["bound"], dtype=torch.float64)
        torch.manual_seed(0)
        theirs = ref.JointEncoding(cfg, bb)
        torch.manual_seed(0)
        ours = RepoJointEncoding(cfg, bb)
        sd_t, sd_o = theirs.state_dict(), ours.state_dict()
        assert list(sd_t) == list(sd_o), (list(sd_t), list(sd_o))
        for k in sd_t:
            assert sd_t[k].shape == sd_o[k].shape and sd_t[k].dtype == sd_o[k].dtype, k
        assert len(theirs.all_planes) == len(ours.all_planes)
        for lt, lo in zip(theirs.all_planes, ours.all_planes):
            for pt, po in zip(lt, lo):
                assert pt.shape == po.shape
                assert torch.equal(pt, po), "same seed -> same initial planes (same CPU draws in the same order)"
        assert torch.equal(theirs.bound, ours.bound)


def test_checkpoint_files_cross_compatible(ref, tmp_path):
    """N4 file compatibility, both directions: (a) ``latest_checkpoint.pt`` written by this repository is read by the
    reference's UNMODIFIED Mapper.load_foreign_model into the reference's own JointEncoding, which then renders what
    this repository's model renders; (b) a checkpoint written from the reference's model in the reference's layout is
    read by this repository's reader into this repository's model, same check."""
    g = load_golden("fwd_onegrid")
    cfg = configs.small_test_config(one_grid=True)
    cfg["data"].update(output=str(tmp_path), exp_name="ckpt")
    bb = torch.from_numpy(g["bounding_box"])
    ours = pc.model_from_golden(g, cfg, "cpu").eval()
    rays_o, rays_d, rgb, d, U = [torch.from_numpy(g[k]) for k in ("rays_o", "rays_d", "target_rgb", "target_d", "U")]

    def render(model):
        torch.manual_seed(9)                     # both sides draw the jitter from the CPU generator
        with torch.no_grad():
            out = model.render_rays(rays_o, rays_d, target_d=d)
        return out["rgb"], out["depth"]

    want = render(ours)
    # (a) repo writer -> reference reader
    slam_glue.save_latest_checkpoint(ours, cfg, rank=1)
    theirs = ref.JointEncoding(cfg, bb)
    host = types.SimpleNamespace(config=cfg, device=torch.device("cpu"), model_shared=theirs)
    ckpt = ref.Mapper.load_foreign_model(host, 1)
    assert set(ckpt) == {"model", "all_planes", "bound", "bounding_box"}
    got = render(theirs)
    assert_close(got[0], want[0], rtol=1e-4, atol=1e-5, what="rgb through the reference after load_foreign_model")
    assert_close(got[1], want[1], rtol=1e-4, atol=1e-5, what="depth through the reference after load_foreign_model")
    # (b) reference-layout writer (mneslam_mp.py:294-315 on the reference's model) -> repo reader
    path = os.path.join(slam_glue.agent_dir(cfg, 2), "latest_checkpoint.pt")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"model": theirs.state_dict(), "all_planes": theirs.all_planes, "bound": theirs.bound.cpu(),
                "bounding_box": theirs.bounding_box.cpu()}, path)
    fresh = RepoJointEncoding(cfg, bb.clone() * 0.5)
    slam_glue.load_foreign_model(fresh, cfg, 2, torch.device("cpu"))
    assert not fresh.training
    for lst in fresh.all_planes:
        for p in lst:
            assert p.is_contiguous(memory_format=torch.channels_last)
    got = render(fresh)
    assert_close(got[0], want[0], rtol=1e-4, atol=1e-5, what="rgb after reading a reference-layout checkpoint")
    assert_close(got[1], want[1], rtol=1e-4, atol=1e-5, what="depth after reading a reference-layout checkpoint")
