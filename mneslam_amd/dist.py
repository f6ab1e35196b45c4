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
    n_dev = torch.cuda.device_count() if use_gpu else 0
    # More ranks on this node than GPUs?  The deployment is one agent per GPU (RCCL refuses two ranks on one device); MNE_SHARE_GPUS=1
    # lets the ranks share the devices round robin with gloo as the transport (it takes device tensors): a FUNCTIONAL run of the
    # multi-agent data path on the HIP library -- what a one-GPU box can execute of it (tests/test_hip_parity_gpu.py; bench.py marks
    # its line).  Every rank of the node takes the same branch (LOCAL_WORLD_SIZE is the launcher's).
    shared = use_gpu and int(os.environ.get("LOCAL_WORLD_SIZE", world)) > n_dev
    if shared:
        if os.environ.get("MNE_SHARE_GPUS", "0") != "1":
            raise RuntimeError(f"{os.environ.get('LOCAL_WORLD_SIZE', world)} ranks on a node with {n_dev} GPU(s): one rank per GPU "
                               "(MNE_SHARE_GPUS=1: ranks share devices over gloo, functional runs only)")
        local = local % n_dev
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        backend = backend or ("gloo" if shared else "nccl" if use_gpu else "gloo")       # (every rank of a node takes the same branch)
        if backend == "nccl":
            dist.init_process_group(backend, device_id=device)
        else:
            dist.init_process_group(backend)
    return rank, world, device


def _meta_tensor(model):
    """[n_planes, then (C,H,W) per plane, then 6 bound + 6 bounding-box values, then 1.0 if the bounding box is
    float64 (it is in the live system, mneslam_mp.py:223) else 0.0, then 1.0 if the planes are stored in fp16] as float64."""
    planes = [p for lst in model.all_planes for p in lst]
    vals = [float(len(planes))]
    for p in planes:
        vals += [float(p.shape[1]), float(p.shape[2]), float(p.shape[3])]
    vals += [float(v) for v in model.bound.reshape(-1)]
    bb = torch.as_tensor(model.bounding_box).detach().cpu()
    vals += [float(v) for v in bb.reshape(-1)]
    vals.append(1.0 if bb.dtype == torch.float64 else 0.0)
    vals.append(1.0 if planes and planes[0].dtype == torch.float16 else 0.0)      # half-precision plane storage (extension)
    return torch.tensor(vals, dtype=torch.float64)


def send_model(model, dst, device=None, group=None):
    """Hand this agent's map (planes, decoder, bounds) to agent ``dst`` -- what the reference does by
    writing ``latest_checkpoint.pt`` for a peer to ``torch.load`` (mneslam_mp.py:294-315).  ``dst`` is a global rank."""
    planes = [p for lst in model.all_planes for p in lst]
    device = device or planes[0].device
    meta = _meta_tensor(model).to(device)
    n = torch.tensor([meta.numel()], dtype=torch.int64, device=device)
    dist.send(n, dst, group=group)
    dist.send(meta, dst, group=group)
    for p in planes:                     # logical NCHW order on the wire, whatever the physical layout
        dist.send(p.detach().contiguous(), dst, group=group)
    for w in model.decoder.parameters():
        dist.send(w.detach().contiguous(), dst, group=group)


def recv_model_into(model_shared, src, device=None, group=None):
    """Receive a peer's map into ``model_shared`` (mp_slam/mapper.py:708-726: replaces all_planes, bound,
    bounding_box and the decoder weights wholesale; the receiving model is put in eval mode)."""
    device = device or next(model_shared.decoder.parameters()).device
    n = torch.zeros(1, dtype=torch.int64, device=device)
    dist.recv(n, src, group=group)
    meta = torch.zeros(int(n.item()), dtype=torch.float64, device=device)
    dist.recv(meta, src, group=group)
    meta = meta.cpu()
    n_planes = int(meta[0].item())
    shapes = meta[1:1 + 3 * n_planes].reshape(n_planes, 3).to(torch.int64).tolist()
    bound = meta[1 + 3 * n_planes:7 + 3 * n_planes].reshape(3, 2)
    bbox = meta[7 + 3 * n_planes:13 + 3 * n_planes].reshape(3, 2)
    if float(meta[13 + 3 * n_planes]) == 0.0:        # the sender's box was fp32: keep its dtype (torch.load would)
        bbox = bbox.float()
    plane_dtype = torch.float16 if meta.numel() > 14 + 3 * n_planes and float(meta[14 + 3 * n_planes]) != 0.0 else torch.float32
    planes = []
    for c, h, w in shapes:
        buf = torch.empty(1, c, h, w, device=device, dtype=plane_dtype)
        dist.recv(buf, src, group=group)
        planes.append(buf.contiguous(memory_format=torch.channels_last))
    lists = [planes[i:i + 2] for i in range(0, n_planes, 2)]        # [coarse, fine] per orientation
    model_shared.all_planes = tuple(lists)
    model_shared.bound = bound.float()
    model_shared.bounding_box = bbox.to(device)
    for w in model_shared.decoder.parameters():
        buf = torch.empty_like(w)
        dist.recv(buf, src, group=group)
        with torch.no_grad():
            w.copy_(buf)
    model_shared.eval()
    return model_shared


class ModelExchange:
    """The reference's map hand-off at loop closure / fusion is ONE-SIDED: the agent that detects the loop reads the peer's
    last ``latest_checkpoint.pt`` whenever it wants (mp_slam/mapper.py:708-726); the peer does not take part.  Over a
    process group that becomes a tiny service: every agent runs ``serve()`` in a daemon thread which waits for a request
    from ANY peer on a control group (gloo: any-source receive) and answers with ``send_model`` on a data group (RCCL
    point-to-point over xGMI between GPUs; gloo on CPU), on its own HIP stream; ``fetch(model_shared, src)`` is what
    ``Mapper.load_foreign_model`` calls instead of ``torch.load``.  All groups are private to the exchange (created
    collectively in ``__init__``), so it never interleaves with the agents' own collectives (shared-decoder all-reduce,
    overlap rectangles).

    * **One data group per DIRECTION** (``data_up``: the sender's rank is below the receiver's, ``data_down``: above).
      Loop closure is symmetric, so two agents can fetch each other at the same moment: each then has a receive posted by
      its mapping thread and a send posted by its service thread for the SAME peer.  On one RCCL communicator those two
      point-to-point operations are executed in the order they were enqueued -- receive first on both sides, each waiting
      for a send that sits behind the other's receive.  With a group per direction a communicator only ever
      carries transfers from ONE side of a pair to the other, so no receive can be queued in front of a send.
    * **A consistent map.**  ``lock`` is held while a map is being sent, and the mapper (``FusedMappingMixin``) holds it
      while it updates the map -- ``mapping_optimize`` / ``first_frame_mapping`` / ``distillation`` -- and leaves with every
      stream of the fused step joined (``FusedStep.check``): peers see maps at keyframe boundaries only, where the
      reference writes its checkpoint (mneslam_mp.py:294-315), never planes and decoder of different iterations.  The
      service additionally waits for the work enqueued so far on the stream the exchange was created on (the mapper's).
    * ``stop()`` is collective (a barrier over the control group, then the STOP messages): no agent's service ends while a
      peer may still fetch from it, and every service thread has left ``recv`` before ``destroy_process_group``."""

    STOP, FETCH = 0, 1

    def __init__(self, model, device=None):
        import threading
        if not dist.is_initialized():
            raise RuntimeError("ModelExchange needs an initialised process group (dist.init_agents)")
        self.model, self.rank, self.world = model, dist.get_rank(), dist.get_world_size()
        self.device = torch.device(device) if device is not None else next(model.decoder.parameters()).device
        on_gpu = self.device.type == "cuda"
        # collective calls: every agent builds its exchange at the same point of its start-up
        self.ctrl = dist.new_group(backend="gloo")
        self.data_up = dist.new_group(backend="nccl" if on_gpu else "gloo")
        self.data_down = dist.new_group(backend="nccl" if on_gpu else "gloo")
        self.lock = threading.RLock()
        self.producer_stream = torch.cuda.current_stream(self.device) if on_gpu else None
        self._thread, self.served = None, 0

    def _data(self, sender, receiver):
        return self.data_up if sender < receiver else self.data_down

    def start(self, producer_stream=None):
        """``producer_stream``: the HIP stream the mapping thread enqueues its map updates on (default: the current stream
        of the thread that calls ``start`` -- a mapper working on another stream passes its own)."""
        import threading
        if self.device.type == "cuda":
            self.producer_stream = producer_stream if producer_stream is not None else torch.cuda.current_stream(self.device)
        if self.world == 1:
            return self                  # nobody to serve: no thread (gloo has no send-to-self that could ever stop it)
        if self._thread is None:
            self._thread = threading.Thread(target=self.serve, name=f"mne-model-exchange-{self.rank}", daemon=True)
            self._thread.start()
        return self

    def serve(self):
        stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        while True:
            req = torch.zeros(2, dtype=torch.int64)
            src = dist.recv(req, group=self.ctrl)                  # any source
            if int(req[0]) == self.STOP:
                return
            group = self._data(self.rank, src)
            with self.lock:                                        # not while the mapper is inside an update
                if stream is not None:
                    stream.wait_stream(self.producer_stream)       # the map as enqueued so far by the mapping thread
                    with torch.cuda.stream(stream):
                        send_model(self.model, src, self.device, group=group)
                    stream.synchronize()
                else:
                    send_model(self.model, src, self.device, group=group)
            self.served += 1

    def fetch(self, model_shared, src):
        """The peer's CURRENT map into ``model_shared`` (planes, decoder, both boxes; eval mode): one request + the transfer.
        The caller must not hold ``lock`` (two agents fetching each other would then wait for each other's service)."""
        if src == self.rank:
            raise ValueError("an agent does not fetch its own map")
        dist.send(torch.tensor([self.FETCH, self.rank], dtype=torch.int64), src, group=self.ctrl)
        return recv_model_into(model_shared, src, self.device, group=self._data(src, self.rank))

    def stop(self):
        """COLLECTIVE: every agent calls it once it will fetch no more (before destroy_process_group).  The barrier makes
        sure nobody's service ends while a peer is still fetching; then rank r ends the service of rank r + 1 (gloo has no
        send-to-self) and joins its own thread, which rank r - 1 is ending at the same time."""
        if self._thread is not None:
            if self.world > 1:
                dist.barrier(group=self.ctrl)
                dist.send(torch.tensor([self.STOP, self.rank], dtype=torch.int64), (self.rank + 1) % self.world, group=self.ctrl)
            self._thread.join(timeout=60)
            if self._thread.is_alive():
                raise RuntimeError("ModelExchange.stop: the service thread did not end (did every agent call stop()?)")
            self._thread = None


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


# --------------------------------------------------------------------------------------------------------------
# EXTENSION (BASELINE configs 3-5; the reference has no such exchange): plane gradients of the region two
# agents both map.  The reference's agents own unrelated lattices (each plane spans its agent's own bound,
# model/scene_rep.py:96-109), so cells of different agents do not coincide in general; the exchange is defined for
# agents laid out on ONE global lattice: node spacing h per level, every agent's bound starts on a node of that lattice
# and spans a whole number of cells.  ``aligned_agent_bounds`` produces such bounds (the overlap boxes of
# mp_slam/mapper.py:491-509 / configs/Indoor/indoor.yaml:169-173 are then whole sub-rectangles of both agents' planes).
# --------------------------------------------------------------------------------------------------------------
def aligned_agent_bounds(global_bound, n_agents, axis, overlap, cell):
    """Split ``global_bound`` ([3][2]) into ``n_agents`` slabs along ``axis`` that overlap by about ``overlap`` metres,
    every slab edge on a multiple of ``cell`` from the global lower corner (``cell`` = the coarsest plane cell, which
    the finer levels divide).  Returns a list of [3][2] bounds."""
    lo, hi = global_bound[axis]
    n_cells = int(round((hi - lo) / cell))
    per = n_cells // n_agents
    ov = max(int(round(overlap / cell)), 1)
    out = []
    for k in range(n_agents):
        a = max(k * per - (ov // 2 if k else 0), 0)
        b = n_cells if k == n_agents - 1 else min((k + 1) * per + (ov - ov // 2), n_cells)
        bnd = [list(map(float, x)) for x in global_bound]
        bnd[axis] = [lo + a * cell, lo + b * cell]
        out.append(bnd)
    return out


def overlap_slices(bound_a, bound_b, shape_a, shape_b, axes, tol=1e-4):
    """Index rectangles of the region both planes cover.  ``bound_*``: [3][2] extents the planes span (node 0 at lo, last
    node at hi: align_corners=True), ``shape_*`` = (H, W) of the two planes, ``axes`` = (axis of W, axis of H) -- (0,1) for
    xy, (0,2) for xz, (1,2) for yz.  Returns ((ys_a, xs_a), (ys_b, xs_b)) slices, or None when the boxes do not
    intersect; raises when the two lattices do not coincide (different node spacing or an offset off the lattice)."""
    sl = [[None, None], [None, None]]
    for which, ax in ((1, axes[0]), (0, axes[1])):                      # which: 1 = W (x of the plane), 0 = H
        na, nb = shape_a[which], shape_b[which]
        la, ha = bound_a[ax]
        lb, hb = bound_b[ax]
        h_a, h_b = (ha - la) / (na - 1), (hb - lb) / (nb - 1)
        if abs(h_a - h_b) > tol * h_a:
            raise ValueError(f"axis {ax}: node spacings differ ({h_a} vs {h_b}): the agents do not share a lattice")
        off = (lb - la) / h_a
        if abs(off - round(off)) > tol * max(1.0, abs(off)):
            raise ValueError(f"axis {ax}: peer lattice is shifted by a fraction of a cell")
        off = int(round(off))
        lo, hi = max(0, off), min(na - 1, off + nb - 1)                  # node range in a's indices
        if hi < lo:
            return None
        sl[0][which] = slice(lo, hi + 1)
        sl[1][which] = slice(lo - off, hi - off + 1)
    return (sl[0][0], sl[0][1]), (sl[1][0], sl[1][1])


def exchange_overlap_gradients(grads, shapes_bounds, peer, peer_shapes_bounds):
    """Add the peer agent's plane gradients over the region both agents map, and hand ours to the peer (pairwise,
    symmetric; RCCL send/recv over xGMI on GPUs).  ``grads``: list of [1,C,H,W] gradient buffers in all_planes order;
    ``shapes_bounds`` / ``peer_shapes_bounds``: per plane ((H, W), bound [3][2], axes) of this agent / the peer (exchanged once at
    start-up).  Planes without overlap are skipped.  Afterwards both agents hold the SUM on the shared cells."""
    rank = dist.get_rank()
    for g, (shape, bound, axes), (pshape, pbound, _) in zip(grads, shapes_bounds, peer_shapes_bounds):
        ov = overlap_slices(bound, pbound, shape, pshape, axes)
        if ov is None:
            continue
        (ys, xs), _ = ov
        mine = g[:, :, ys, xs].contiguous()
        theirs = torch.empty_like(mine)
        if rank < peer:
            dist.send(mine, peer)
            dist.recv(theirs, peer)
        else:
            dist.recv(theirs, peer)
            dist.send(mine, peer)
        g[:, :, ys, xs] += theirs


def exchange_buffers(peers, send, recv):
    """One message each way per peer, all posted together (the binned path's overlap exchange: ``send[k]`` holds this agent's
    gradients of the cells shared with ``peers[k]``, ``recv[k]`` receives the peer's; RCCL point-to-point over xGMI on
    GPUs, stream-ordered with the caller's current stream)."""
    ops = []
    for peer, s, r in zip(peers, send, recv):
        ops.append(dist.P2POp(dist.isend, s, peer))
        ops.append(dist.P2POp(dist.irecv, r, peer))
    for w in dist.batch_isend_irecv(ops):
        w.wait()


def allreduce_sum_into(send, recv):
    """Group slot of the binned overlap exchange (more than two agents, FusedStep(overlap_group_axis=...)): ``recv`` = the sum of
    every agent's ``send`` (one all-reduce: reduce-scatter + all-gather over the xGMI links on GPUs; every rank receives the same
    bits), ``send`` zeroed -- mne_tile_adam_shared adds send + recv, i.e. exactly the total, on every agent."""
    recv.copy_(send)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(recv, op=dist.ReduceOp.SUM)
    send.zero_()


def plane_geometry(model):
    """((H, W), extended bound as [3][2] floats, plane axes) for every plane of ``model`` in all_planes order."""
    bound = [[float(lo), float(hi)] for lo, hi in torch.as_tensor(model.bound).cpu()]
    axes = [(0, 1), (0, 2), (1, 2)]
    out = []
    for k, lst in enumerate(model.all_planes):
        for p in lst:
            out.append(((p.shape[2], p.shape[3]), bound, axes[k % 3]))
    return out


def max_over_ranks(seconds, device):
    """bench.py timing rule: the job time is the slowest rank's."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])
