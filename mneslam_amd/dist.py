"""Multi-agent plumbing over torch.distributed (backend "nccl" = RCCL over xGMI on MI355X, "gloo" on CPU).

The reference runs one agent per GPU with NO collective on the mapping path: agents exchange state
through files -- ``latest_checkpoint.pt`` (planes + decoder + bounds, mneslam_mp.py:294-315) read by
``Mapper.load_foreign_model`` (mp_slam/mapper.py:708-726), and ``key_est_poses.npy`` /
``key_timestamps.npy`` (mp_slam/mapper.py:565-592).  This module re-expresses those two exchange steps
as point-to-point transfers between the agents' processes, and adds the one
extension BASELINE.json's multi-GPU configs ask for that does not exist in the reference: a shared
decoder, i.e. an all-reduce (mean) of the 6,208-float decoder gradient before the decoder's Adam step.
"""
import os

import torch
import torch.distributed as dist


def init_agents(backend=None):
    """One process per agent/GPU.  Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment
    (torch.distributed.run); returns (rank, world_size, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        backend = backend or ("nccl" if use_gpu else "gloo")
        if backend == "nccl":
            dist.init_process_group(backend, device_id=device)
        else:
            dist.init_process_group(backend)
    return rank, world, device


def _meta_tensor(model):
    """[n_planes, then (C,H,W) per plane, then 6 bound + 6 bounding-box values, then 1.0 if the bounding box is
    float64 (it is in the live system, mneslam_mp.py:223) else 0.0] as float64."""
    planes = [p for lst in model.all_planes for p in lst]
    vals = [float(len(planes))]
    for p in planes:
        vals += [float(p.shape[1]), float(p.shape[2]), float(p.shape[3])]
    vals += [float(v) for v in model.bound.reshape(-1)]
    bb = torch.as_tensor(model.bounding_box).detach().cpu()
    vals += [float(v) for v in bb.reshape(-1)]
    vals.append(1.0 if bb.dtype == torch.float64 else 0.0)
    return torch.tensor(vals, dtype=torch.float64)


def send_model(model, dst, device=None):
    """Hand this agent's map (planes, decoder, bounds) to agent ``dst`` -- what the reference does by
    writing ``latest_checkpoint.pt`` for a peer to ``torch.load`` (mneslam_mp.py:294-315)."""
    planes = [p for lst in model.all_planes for p in lst]
    device = device or planes[0].device
    meta = _meta_tensor(model).to(device)
    n = torch.tensor([meta.numel()], dtype=torch.int64, device=device)
    dist.send(n, dst)
    dist.send(meta, dst)
    for p in planes:                     # logical NCHW order on the wire, whatever the physical layout
        dist.send(p.detach().contiguous(), dst)
    for w in model.decoder.parameters():
        dist.send(w.detach().contiguous(), dst)


def recv_model_into(model_shared, src, device=None):
    """Receive a peer's map into ``model_shared`` (mp_slam/mapper.py:708-726: replaces all_planes, bound,
    bounding_box and the decoder weights wholesale; the receiving model is put in eval mode)."""
    device = device or next(model_shared.decoder.parameters()).device
    n = torch.zeros(1, dtype=torch.int64, device=device)
    dist.recv(n, src)
    meta = torch.zeros(int(n.item()), dtype=torch.float64, device=device)
    dist.recv(meta, src)
    meta = meta.cpu()
    n_planes = int(meta[0].item())
    shapes = meta[1:1 + 3 * n_planes].reshape(n_planes, 3).to(torch.int64).tolist()
    bound = meta[1 + 3 * n_planes:7 + 3 * n_planes].reshape(3, 2)
    bbox = meta[7 + 3 * n_planes:13 + 3 * n_planes].reshape(3, 2)
    if float(meta[13 + 3 * n_planes]) == 0.0:        # the sender's box was fp32: keep its dtype (torch.load would)
        bbox = bbox.float()
    planes = []
    for c, h, w in shapes:
        buf = torch.empty(1, c, h, w, device=device)
        dist.recv(buf, src)
        planes.append(buf.contiguous(memory_format=torch.channels_last))
    lists = [planes[i:i + 2] for i in range(0, n_planes, 2)]        # [coarse, fine] per orientation
    model_shared.all_planes = tuple(lists)
    model_shared.bound = bound.float()
    model_shared.bounding_box = bbox.to(device)
    for w in model_shared.decoder.parameters():
        buf = torch.empty_like(w)
        dist.recv(buf, src)
        with torch.no_grad():
            w.copy_(buf)
    model_shared.eval()
    return model_shared


def gather_keyframe_poses(poses, timestamps):
    """All agents' keyframe poses/timestamps (the reference's key_est_poses.npy / key_timestamps.npy,
    mp_slam/mapper.py:565-592) -> list indexed by rank."""
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, (poses.detach().cpu(), timestamps.detach().cpu()))
    return out


def allreduce_mean_(buf):
    """EXTENSION (not reference behaviour): average a gradient buffer over all agents, e.g. the
    24.8 KB decoder gradient of a shared decoder.  Latency-bound on xGMI: one fused buffer, one call."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        buf.div_(dist.get_world_size())
    return buf


def max_over_ranks(seconds, device):
    """bench.py timing rule: the job time is the slowest rank's."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])
