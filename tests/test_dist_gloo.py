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
    # 4) timing rule
    assert mdist.max_over_ranks(0.1 * (rank + 1), dev) == pytest.approx(0.2)
    dist.destroy_process_group()
    open(ret + f".ok{rank}", "w").write("ok")


def test_two_agents_gloo(tmp_path):
    port = 29600 + (os.getpid() % 300)
    ret = str(tmp_path / "r")
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert os.path.exists(ret + ".ok0") and os.path.exists(ret + ".ok1")
