"""The fused mapping iteration: sample rays -> z -> decode / composite / backward -> decoder
weight-gradient pass + decoder Adam, and -- concurrently on a second HIP stream -- the per-tile plane
scatter fused with Adam.  No autograd graph, no per-iteration allocation, no host synchronisation: every
buffer is allocated once per ``FusedStep`` and the ~10 C-ABI calls of an iteration are asynchronous
launches on the caller's stream and one private side stream (joined with events, see ``step``).

Semantically one ``step`` equals the reference's
``forward -> get_loss_from_ret -> backward -> map_optimizer.step() -> zero_grad()``
(mp_slam/mapper.py:155-162) on the same ray batch; tests run both paths against the reference's
golden parameters.
"""
import ctypes as C
import os

import torch

from . import _lib, hip_path, slam_glue
from .optim import FusedAdam


class _null_ctx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


class FusedStep:
    LIST_BUDGET_BYTES = 4 << 30          # upper bound for the tile lists of the binned plane update (all tiles together)

    def __init__(self, model, optimizer, config, n_rays, device, is_co_sdf=None, scatter="binned",
                 tile_capacity=None, spill_capacity=None, shared_decoder=False, overlap=True, overlap_peers=None,
                 use_graph=None, overlap_group_axis=None):
        """scatter="binned": plane gradients are accumulated per 16x16-cell tile in LDS and Adam is applied
        in the same kernel (csrc/tile_adam.hip; no gradient buffers).  scatter="atomics": global
        atomic adds into persistent gradient buffers + the streaming Adam kernel.
        use_graph (EXTENSION, BASELINE configs[4] "hipGraph-captured mapping iteration"; default off / env MNE_GRAPH):
        steady-state iterations are recorded once into a HIP graph and replayed -- "two_stream" captures the two-stream
        schedule of ``step``, "one_stream" the same launches on one stream (see ``_record``)."""
        if scatter not in ("binned", "atomics"):
            raise ValueError("scatter must be binned|atomics")
        self.scatter = scatter
        # EXTENSION (BASELINE multi-GPU configs; not reference behaviour): agents share one decoder, so the
        # decoder gradient is averaged over all ranks (RCCL all-reduce over xGMI) before its Adam step
        self.shared_decoder = shared_decoder
        # EXTENSION: [(peer rank, peer plane geometry)] of agents on the same global lattice whose bounds overlap ours:
        # the plane gradients of the shared region are summed pairwise before Adam (dist.exchange_overlap_gradients)
        # (binned path: the shared cells' gradients travel as rectangles cut out of the tiles' LDS sums, one message each
        # way per peer -- mne_tile_grad_export / mne_tile_adam_shared; atomics path: slices of the gradient buffers)
        self.overlap_peers = list(overlap_peers or [])
        self.tile_overlap = None
        # EXTENSION, more than two agents in a chain of slabs along ``overlap_group_axis``: the planes that do not contain that
        # axis (yz for slabs along x) are held as a whole by EVERY agent, so their gradient is summed over ALL agents (one
        # all-reduce per iteration, dist.allreduce_sum_into) instead of pairwise with the neighbours -- pairwise sums would
        # leave the middle agent with g0 + g1 + g2 and its neighbours with g0 + g1 / g1 + g2.  None: every plane pairwise
        # (two agents).
        self.overlap_group_axis = overlap_group_axis
        if overlap_group_axis is not None and (scatter != "binned" or not self.overlap_peers):
            raise ValueError("overlap_group_axis belongs to the binned plane update with overlap_peers")
        n_slots = len(self.overlap_peers) + (1 if overlap_group_axis is not None else 0)
        if n_slots > _lib.MAX_OVERLAP_PEERS and scatter != "atomics":
            raise ValueError(f"the binned plane update exchanges with at most two neighbours (slabs along one axis) + the group")
        if not isinstance(optimizer, FusedAdam):
            raise TypeError("the fused mapping step needs mneslam_amd.optim.FusedAdam "
                            "(slam_glue.create_optimizer builds it with the reference's groups)")
        self.lib = _lib.load()
        self.model, self.opt, self.cfg, self.R = model, optimizer, config, int(n_rays)
        self.device = torch.device(device)
        dev, f32 = self.device, torch.float32
        self.info = model._info()
        self.rc = self.info["render_cfg"]
        self.S = self.lib.mne_num_samples(C.byref(self.rc), 1)
        R, S = self.R, self.S
        self.planes = [p for lst in model.all_planes for p in lst]
        for p in self.planes:
            if not isinstance(p, torch.nn.Parameter):
                raise ValueError("planes must be nn.Parameters (call create_optimizer first)")
            if not (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and p.stride(1) == 1) \
                    or p.device.type != dev.type:
                raise ValueError("fused step needs channels_last planes on the compute device")
        self.dec_w = model.decoder.hip_weights()
        self.group_of = {p: g for g in optimizer.param_groups for p in g["params"]}
        self.group_of_dec = self.group_of[self.dec_w[0]]
        if any(self.group_of[w] is not self.group_of_dec for w in self.dec_w):
            raise ValueError("the decoder tensors must be in one param group (slam_glue.create_optimizer builds it that way)")
        self.grads = None
        if scatter == "atomics":
            # persistent gradient accumulators (zeroed by the Adam kernel itself after each use)
            self.grads = []
            for p in self.planes:
                st = optimizer._state(p)
                if "grad_buffer" not in st:
                    st["grad_buffer"] = torch.zeros_like(p.data, dtype=torch.float32)
                self.grads.append(st["grad_buffer"])
        self._graph_mode = use_graph
        self.scene = hip_path.scene_struct(self.info, [p.data for p in self.planes], [w.data for w in self.dec_w], self.grads)
        self._alloc_buffers(config, is_co_sdf)
        self.bins = None
        if scatter == "atomics":
            self.grad_map.update({p: g for p, g in zip(self.planes, self.grads)})
        else:
            n_tiles = self.lib.mne_tile_count(C.byref(self.scene))
            b = _lib.TileBins()
            if tile_capacity is None:
                # One capacity PER PLANE: 4x its mean list length if every sample contributed (a sample touches ~1.3 tiles of
                # a plane), at least 4096 entries.  A coarse plane's few tiles take 10^4-10^5 entries each, a fine plane's
                # thousands of tiles a few hundred: one capacity for all reserved 17.5 GB on ScanNet with colour planes and
                # 30 GB on INS Indoor, and capping THAT sent ScanNet's coarse lists into the spill area (every
                # overflowing tile scans all of it: 0.44 -> 1.16 ms).  Per plane: office0 0.65 GB, ScanNet 1.27, Indoor 2.24.
                # LIST_BUDGET_BYTES still bounds the sum (capacities scaled down together); overflow stays correct (spill
                # area sized for the worst case), only slower.
                tiles = [((p.shape[2] + 15) // 16) * ((p.shape[3] + 15) // 16) for p in self.planes]
                caps = [int(max(4096, 4 * 1.3 * R * S / t)) for t in tiles]
                need = 32 * sum(t * c for t, c in zip(tiles, caps))
                if need > self.LIST_BUDGET_BYTES:
                    caps = [max(256, int(c * self.LIST_BUDGET_BYTES / need)) for c in caps]
                tile_capacity = max(caps)
                for k, c in enumerate(caps):
                    b.plane_cap[k] = c
            b.cap = tile_capacity
            n_entries = self.lib.mne_tile_list_entries(C.byref(self.scene), C.byref(b))
            # (only the counters need to start at zero: an entry is read after it was written)
            self.tile_lists = torch.empty(n_entries, 8, device=dev, dtype=torch.int32)
            # one cursor per (list, XCD): every list is cut into LIST_SEGMENTS segments (csrc/mne_launch.h)
            self.tile_counts = torch.zeros(n_tiles * _lib.LIST_SEGMENTS, device=dev, dtype=torch.int32)
            if spill_capacity is None:
                # worst case: every sample appends to 4 tiles of every plane and every entry overflows its list --
                # then nothing can ever be dropped (small scenes put >4096 samples into most tiles; office0 none)
                spill_capacity = R * S * len(self.planes) * 4
            self.spill = torch.empty(spill_capacity, 8, device=dev, dtype=torch.int32)
            self.spill_count = torch.zeros(1, device=dev, dtype=torch.int32)
            self.dropped = torch.zeros(1, device=dev, dtype=torch.int32)
            b.lists, b.counts = self.tile_lists.data_ptr(), self.tile_counts.data_ptr()
            b.spill, b.spill_count = self.spill.data_ptr(), self.spill_count.data_ptr()
            # processing order of the tiles (heaviest lists first), recomputed every iteration from that iteration's list
            # lengths.  (Sorting by the PREVIOUS iteration's lengths after the plane update, off the critical path, was
            # measured: tile_adam_kernel 244 us instead of 225 us, iteration 557 us instead of 544 us -- profiles/r02_order_ab.txt.)
            # long lists are cut into parts processed by several workgroups (load balance; tile_adam.hip): scratch for their
            # partial gradient tiles + arrival counters.  MNE_NO_TILE_SPLIT=1 keeps one workgroup per tile (A/B).
            split = os.environ.get("MNE_NO_TILE_SPLIT", "0") != "1"
            self.tile_order = torch.arange(n_tiles + (_lib.TILE_SPLIT_PARTS if split else 0), device=dev, dtype=torch.int32)
            b.order = self.tile_order.data_ptr()
            if split:
                self.split_scratch = torch.empty(_lib.TILE_SPLIT_PARTS, 16 * 16 * 32, device=dev)
                self.split_state = torch.zeros(n_tiles + 1, device=dev, dtype=torch.int32)
                b.split_scratch, b.split_state = self.split_scratch.data_ptr(), self.split_state.data_ptr()
            self._order_early = n_tiles <= _lib.TILE_ORDER_SNAPSHOT
            self.prev_counts = torch.zeros(n_tiles, device=dev, dtype=torch.int32)
            b.prev_counts = self.prev_counts.data_ptr()
            # tiles that never received a gradient keep m = v = 0 and are skipped by the plane update (bit-identical: Adam
            # does not move them).  Valid while the moments are the ones THIS object has seen grow from zero: an optimizer
            # that already stepped, or moments re-bound later (load_state_dict), make every tile live (_refresh_pointers).
            fresh = all(optimizer._state(p)["step"] == 0 for p in self.planes) and os.environ.get("MNE_SWEEP_ALL_TILES", "0") != "1"
            self.tile_live = (torch.zeros if fresh else torch.ones)(n_tiles, device=dev, dtype=torch.int32)
            b.live = self.tile_live.data_ptr()
            self._moment_ptrs = None
            b.cap, b.spill_cap = tile_capacity, spill_capacity
            if self.overlap_peers:
                from . import dist as mdist
                ov = _lib.TileOverlap()
                n_nb, gax = len(self.overlap_peers), self.overlap_group_axis
                ov.n_peers = n_nb + (1 if gax is not None else 0)
                geo = mdist.plane_geometry(model)
                for k, (peer, peer_geo) in enumerate(self.overlap_peers):
                    for pi, ((shape, bound, axes), (pshape, pbound, _)) in enumerate(zip(geo, peer_geo)):
                        if gax is not None and gax not in axes:          # held by every agent: the group slot below
                            if tuple(shape) != tuple(pshape) or any(abs(bound[a][q] - pbound[a][q]) > 1e-6 for a in axes for q in (0, 1)):
                                raise ValueError(f"plane {pi} does not contain the slab axis but differs between the agents")
                            continue
                        sl = mdist.overlap_slices(bound, pbound, shape, pshape, axes)
                        if sl is not None:
                            (ys, xs), _ = sl
                            r = ov.rect[k][pi]
                            r.x0, r.x1, r.y0, r.y1 = xs.start, xs.stop, ys.start, ys.stop
                if gax is not None:
                    for pi, (shape, bound, axes) in enumerate(geo):
                        if gax not in axes:
                            r = ov.rect[n_nb][pi]
                            r.x0, r.x1, r.y0, r.y1 = 0, shape[1], 0, shape[0]
                self.ov_send, self.ov_recv = [], []
                for k in range(ov.n_peers):
                    n = self.lib.mne_tile_overlap_floats(C.byref(self.scene), C.byref(ov), k)
                    self.ov_send.append(torch.zeros(max(n, 1), device=dev))
                    self.ov_recv.append(torch.zeros(max(n, 1), device=dev))
                    ov.send[k], ov.recv[k] = self.ov_send[k].data_ptr(), self.ov_recv[k].data_ptr()
                self.tile_overlap = ov
            b.dropped = self.dropped.data_ptr()
            self.bins = b
            self.plane_opt = (_lib.PlaneOpt * len(self.planes))()
            for k, p in enumerate(self.planes):
                st, grp = optimizer._state(p), self.group_of[p]
                o = self.plane_opt[k]
                o.m, o.v = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                o.lr, (o.beta1, o.beta2) = float(grp["lr"]), map(float, grp["betas"])
                o.eps, o.weight_decay = float(grp["eps"]), float(grp["weight_decay"])
        self._finish_init(overlap)

    def _alloc_buffers(self, config, is_co_sdf):
        """Batch, output, tape and decoder-gradient buffers (everything that does not depend on the scene encoding)."""
        dev, f32, R, S = self.device, torch.float32, self.R, self.S
        e = lambda *shape, dtype=f32: torch.empty(*shape, device=dev, dtype=dtype)
        self.rays_o, self.rays_d, self.tgt_rgb, self.tgt_d = e(R, 3), e(R, 3), e(R, 3), e(R)
        self.idx = e(R, dtype=torch.int64)
        self.z_vals, self.raw = e(R, S), e(R, S, 4)
        self.counts, self.ray_counts = e(_lib.N_COUNT, dtype=torch.int32), e(R, _lib.N_COUNT, dtype=torch.int32)
        self.coef, self.ray_sums, self.losses = e(_lib.N_LOSS), e(R, _lib.N_LOSS), torch.zeros(_lib.N_LOSS, device=dev)
        self.rgb, self.depth = e(R, 3), e(R)
        self.packed = e(self.lib.mne_packed_decoder_floats(C.byref(self.scene)))
        self.tape = e(R * S, self.lib.mne_tape_row_floats(C.byref(self.scene)))
        self.tape_rows = torch.zeros(1, device=dev, dtype=torch.int32)
        self.ray_tiles = torch.zeros(R, device=dev, dtype=torch.int32)
        self.ws_bytes = self.lib.mne_render_workspace_bytes(R, S)
        self.ws = e(self.ws_bytes, dtype=torch.uint8)
        self.partials = e(self.lib.mne_wgrad_partial_floats(C.byref(self.scene)))
        self.dec_grad = e(self.lib.mne_decoder_param_floats(C.byref(self.scene)))
        co = config["is_co_sdf"] if is_co_sdf is None else is_co_sdf
        self._loss_w_host = slam_glue.loss_weight_vector(config, co) + [0.0]
        self.loss_w = torch.tensor(self._loss_w_host, device=dev, dtype=f32)
        self.tables = hip_path.linspace_tables(config, True, dev)
        w_sdf0, w_sdf1, w_col0, w_col1 = self.dec_w
        n0, n1, n2 = w_col0.numel(), w_col1.numel(), w_sdf0.numel()
        self.dec_grad_views = {w_col0: self.dec_grad[:n0].view_as(w_col0), w_col1: self.dec_grad[n0:n0 + n1].view_as(w_col1),
                               w_sdf0: self.dec_grad[n0 + n1:n0 + n1 + n2].view_as(w_sdf0),
                               w_sdf1: self.dec_grad[n0 + n1 + n2:].view_as(w_sdf1)}
        self.grad_map = dict(self.dec_grad_views)

    def _finish_init(self, overlap):
        # exact early ray termination (decode only the samples a ray needs; csrc/render.hip); False = decode everything
        self.early_termination = os.environ.get("MNE_NO_EARLY_TERMINATION", "0") != "1"
        # Steady-state iterations can be recorded into a HIP graph and replayed (device sampler, prefetching steps): one
        # hipGraphLaunch per iteration; what changes between iterations (sampling keys, jitter counter, Adam step) is read
        # from a device clock (mne_clock_t).  OFF by default: the iteration is GPU-bound and its launches are enqueued far
        # ahead anyway -- replay measured 0.570 vs 0.533 ms (round 2) and 0.669 vs 0.488 ms (round 3, two captured streams:
        # the runtime serialises them; profiles/r03_nsb_graph_fp16.txt).  use_graph / MNE_GRAPH = two_stream | one_stream.
        mode = getattr(self, "_graph_mode", None) or os.environ.get("MNE_GRAPH", "")
        mode = {"": None, "0": None, "1": "two_stream", "2": "one_stream"}.get(mode, mode)
        if mode not in (None, "two_stream", "one_stream"):
            raise ValueError("use_graph must be None | 'two_stream' | 'one_stream'")
        self.use_graph = mode if (self.device.type == "cuda" and self.bins is not None) else None
        self._graphs = {}
        self.events = None          # set to {} to record HIP events around the dominant launches
        self.overlap = overlap
        self._side, self._ev, self._prefetched, self._planes_pending = None, None, None, False
        self._packed_key = None
        self._ev_decode = None       # event "prefix decode enqueued" (two-stream runs with the binned plane update)
        self._bin_pending = False
        self.concurrent_bin = os.environ.get("MNE_SERIAL_BIN", "0") != "1"
        # tests (one-stream runs, the host emulator): hand the resolved rays' appends to mne_tile_bin(pass 0) as the two-stream
        # schedule does, here AFTER the render call returned -- i.e. after the deferred pass and its appends (pass 1): pass 0 must
        # then still see exactly the rays the a-priori prefix resolved (dec_tiles untouched by the list decode), ADVICE r03
        self.split_bin_calls = os.environ.get("MNE_FORCE_EXTERNAL_BIN", "0") == "1"
        # Adaptive a-priori prefix (mne_fused_opts_t::adapt_state): 4 device words that carry the schedule decision from
        # one iteration to the next (mode 0: prefix + deferred pass; mode 1: decode everything a priori while most rays are
        # unresolved, i.e. while the SDF is untrained).  Exact either way.  MNE_NO_ADAPT=1 pins mode 0 (A/B).
        self.adapt_state = (torch.zeros(4, device=self.device, dtype=torch.int32)
                            if os.environ.get("MNE_NO_ADAPT", "0") != "1" else None)
        self.iteration = 0
        self.seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF

    # ---------------------------------------------------------------- stream plumbing
    def _streams(self):
        """(main, side) as (torch stream, raw handle) pairs: the caller's current stream, and a private stream
        for the plane update (tile_adam_kernel, HBM-bound), which is independent of the decoder chain (weight
        gradients -> decoder Adam -> loss scalars -> next batch) once the backward kernel is done.  With
        scatter="atomics", or on the host emulator, everything stays on one stream."""
        main_h = _lib.stream_for(self.rays_o)
        if not self.rays_o.is_cuda or self.bins is None or not self.overlap:
            return (None, main_h), (None, main_h)
        if self._side is None:
            # (a high-priority side stream was measured: no effect, 0.522 vs 0.521 ms -- profiles/r02_variants.txt)
            self._side = torch.cuda.Stream(self.device)
            # (events created with hipEventReleaseToDevice were measured: 1.5 % slower than the runtime's default events,
            # profiles/r03_events_negative.txt)
            self._ev = [torch.cuda.Event() for _ in range(4)]
            if self.concurrent_bin:
                self._ev_decode = torch.cuda.Event()
                self._ev_decode.record(torch.cuda.current_stream(self.device))      # (recorded once so that the handle exists)
        return (torch.cuda.current_stream(self.device), main_h), (self._side, C.c_void_p(self._side.cuda_stream))

    @staticmethod
    def _after(waiter, ev, producer):
        """`waiter` stream continues only after everything enqueued so far on `producer`."""
        if waiter is not None and producer is not None and waiter is not producer:
            ev.record(producer)
            waiter.wait_event(ev)

    def _batch_key(self, kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur):
        return (None if kf_rays is None else kf_rays.data_ptr(), int(n_kf_rays), int(n_save), cur_rays.data_ptr(),
                poses.data_ptr(), poses.shape[0], int(n_global), int(n_cur), self.iteration)

    def _sample_batch(self, kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur, idx_global, idx_cur, u, st,
                      clock=None):
        """R1-R3 for iteration `self.iteration` (or, with ``clock``, for the iteration the device clock holds): ray
        batch, z samples (+ mask counts), loss coefficients."""
        lib, P, R, S = self.lib, _lib.ptr, int(n_global + n_cur), self.S
        it = self.iteration if clock is None else 0
        ck = C.byref(clock) if clock is not None else None
        # one launch: ray draw + z samples + mask counts + loss coefficients (csrc/render.hip, batch_kernel)
        _lib.check(lib.mne_sample_batch(P(kf_rays), int(n_kf_rays), int(n_save), None, P(cur_rays), cur_rays.shape[0],
                                        P(poses), poses.shape[0], n_global, n_cur, P(idx_global), P(idx_cur),
                                        self.seed, it, P(self.rays_o), P(self.rays_d), P(self.tgt_rgb), P(self.tgt_d),
                                        P(self.idx), C.byref(self.rc), P(u), P(self.tables), it * ((R * S + 3) // 4),
                                        P(self.z_vals), P(self.counts), P(self.ray_counts), P(self.loss_w), P(self.coef),
                                        ck, st), "mne_sample_batch")

    def _decoder_chain(self, R, S, st, clock=None):
        """wgrad -> (all-reduce) -> decoder Adam -> loss scalars -> decoder tables of the NEXT render.  Two launches
        (weight-gradient pass; decoder_update_kernel) unless the decoder gradient is shared between agents or a
        cross-check implementation of the weight gradients is selected."""
        lib, P = self.lib, _lib.ptr
        fused = not self.shared_decoder and self.model.wgrad_impl == 0
        _lib.check(lib.mne_decoder_wgrad(C.byref(self.scene), P(self.tape), P(self.ray_tiles), R, S, P(self.partials),
                                         P(self.dec_grad), 3 if fused else self.model.wgrad_impl, st), "mne_decoder_wgrad")
        if not fused:
            if self.shared_decoder:
                from . import dist as mdist
                mdist.allreduce_mean_(self.dec_grad)
            self.opt.step(zero_grad=False, grad_buffers=self.dec_grad_views, clock=clock)      # decoder tensors
            _lib.check(lib.mne_loss_finalize(R, S, P(self.ray_sums), P(self.counts), P(self.losses), st), "mne_loss_finalize")
            self._packed_key = None
            return
        w_sdf0, w_sdf1, w_col0, w_col1 = self.dec_w
        grp = self.group_of_dec
        o = _lib.DecoderOpt()
        steps = set()
        for k, w in enumerate((w_col0, w_col1, w_sdf0, w_sdf1)):       # decoder.parameters() order
            stt = self.opt._state(w)
            if clock is None:
                stt["step"] += 1
            steps.add(stt["step"])
            o.m[k], o.v[k] = stt["exp_avg"].data_ptr(), stt["exp_avg_sq"].data_ptr()
        if len(steps) != 1:
            raise RuntimeError("the decoder tensors must share one Adam step count")
        o.lr, (o.beta1, o.beta2) = float(grp["lr"]), map(float, grp["betas"])
        o.eps, o.weight_decay = float(grp["eps"]), float(grp["weight_decay"])
        o.step = steps.pop() + (1 if clock is not None else 0)
        _lib.check(lib.mne_decoder_update(C.byref(self.scene), P(self.partials), R, P(self.dec_grad), C.byref(o),
                                          S, P(self.ray_sums), P(self.counts), P(self.losses),
                                          C.byref(clock) if clock is not None else None, st), "mne_decoder_update")
        # the NEXT render's decoder tables, here in the decoder chain (beside the plane update), not in front of that render
        _lib.check(lib.mne_pack_decoder(C.byref(self.scene), P(self.packed), st), "mne_pack_decoder")
        self._packed_key = self._decoder_key()              # self.packed now holds the tables of these weights

    def _decoder_key(self):
        return tuple((w.data_ptr(), w._version) for w in self.dec_w)

    def _pack_if_stale(self, st):
        """Decoder tables for the render: repacked only when the weights are not the ones the last decoder update packed
        (first step, re-bound or externally modified weights)."""
        if self._packed_key is None or self._packed_key != self._decoder_key():
            _lib.check(self.lib.mne_pack_decoder(C.byref(self.scene), _lib.ptr(self.packed), st), "mne_pack_decoder")
            self._packed_key = self._decoder_key()

    def _tile_bin(self, pass_, opts, st):
        R, S, P = self.n_active, self.S, _lib.ptr
        _lib.check(self.lib.mne_tile_bin(C.byref(self.scene), C.byref(self.rc), R, S, P(self.rays_o), P(self.rays_d),
                                         P(self.tgt_d), P(self.z_vals),
                                         P(self.ray_counts) if self.early_termination else None, P(self.coef), P(self.raw),
                                         C.byref(self.bins), P(self.ws), self.ws_bytes, pass_, C.byref(opts), st), "mne_tile_bin")

    def _refresh_pointers(self):
        """Planes, decoder weights and Adam moments are ordinary tensors owned by Python: their storage may be
        re-bound between calls (``p.data = ...``, ``optimizer.load_state_dict``), so no device pointer survives
        from one step to the next (ownership)."""
        n = 0
        for s in range(self.scene.n_sets):
            for o in range(3):
                for l in range(2):
                    p = self.planes[n]
                    if not (p.is_contiguous(memory_format=torch.channels_last) and p.stride(1) == 1):
                        raise ValueError("fused step needs channels_last planes (a plane was re-bound with another layout)")
                    if (p.dtype == torch.float16) != bool(self.scene.plane_f16):
                        raise ValueError("a plane was re-bound with another dtype than the step was built for")
                    pl = self.scene.plane[s][o][l]
                    pl.data = p.data_ptr()
                    if self.grads is not None:
                        pl.grad = self.grads[n].data_ptr()
                    if self.bins is not None:
                        st = self.opt._state(p)
                        self.plane_opt[n].m, self.plane_opt[n].v = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    n += 1
        if self.bins is not None:
            ptrs = tuple((o.m, o.v) for o in self.plane_opt)
            if self._moment_ptrs is not None and ptrs != self._moment_ptrs:
                self.tile_live.fill_(1)              # moments re-bound (optimizer.load_state_dict): nothing is known to be zero
            self._moment_ptrs = ptrs
        w_sdf0, w_sdf1, w_col0, w_col1 = self.dec_w
        self.scene.w_sdf0, self.scene.w_sdf1 = w_sdf0.data_ptr(), w_sdf1.data_ptr()
        self.scene.w_col0, self.scene.w_col1 = w_col0.data_ptr(), w_col1.data_ptr()

    # ---------------------------------------------------------------- recorded iteration (HIP graph)
    def _graph_key(self, kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur):
        ptrs = tuple(p.data_ptr() for p in self.planes) + tuple(w.data_ptr() for w in self.dec_w)
        ptrs += tuple(self.opt._state(p)[k].data_ptr() for p in list(self.planes) + list(self.dec_w)
                      for k in ("exp_avg", "exp_avg_sq"))
        # kernel arguments the capture bakes in: a changed learning rate / eps / weight decay / betas of a param group or
        # changed loss weights must record a new graph, not replay the old values
        hyper = tuple((float(g["lr"]), float(g["eps"]), float(g["weight_decay"]), tuple(map(float, g["betas"])))
                      for g in self.opt.param_groups)
        return (None if kf_rays is None else kf_rays.data_ptr(), int(n_kf_rays), int(n_save), cur_rays.data_ptr(),
                cur_rays.shape[0], poses.data_ptr(), poses.shape[0], int(n_global), int(n_cur), ptrs, hyper,
                tuple(self._loss_w_host))

    def _make_clock(self, R, n_table=8192):
        groups = self.opt.param_groups
        b1, b2 = groups[0]["betas"]
        if any(tuple(g["betas"]) != (b1, b2) for g in groups):
            return None                                   # the bias table holds one pair of betas
        dev = self.device
        self.clk_iter = torch.zeros(1, device=dev, dtype=torch.int64)
        self.clk_step = torch.zeros(1, device=dev, dtype=torch.int32)
        t = [(1.0 - float(b1) ** k, 1.0 - float(b2) ** k) for k in range(1, n_table + 1)]     # C pow(), as the host path
        self.clk_table = torch.tensor(t, dtype=torch.float64, device=dev)
        ck = _lib.Clock()
        ck.iteration, ck.step_offset, ck.bias_table = self.clk_iter.data_ptr(), self.clk_step.data_ptr(), self.clk_table.data_ptr()
        ck.n_table, ck.beta1, ck.beta2 = n_table, float(b1), float(b2)
        ck.z_offset_stride = (R * self.S + 3) // 4
        return ck

    def _record(self, args):
        """Record one steady-state iteration -- render of the batch that is already in the buffers, plane update, decoder
        chain, then the NEXT batch -- into a HIP graph.  Everything that changes between iterations (ray-sampling keys,
        jitter counter, Adam step) is read from the device clock.  'two_stream': the schedule of ``step`` (list appends and
        plane update on the side stream, joined by captured events); 'one_stream': the same launches in one queue (the
        render call does its own appends), i.e. no concurrency between the plane update and the decoder chain."""
        kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur = args
        lib, P = self.lib, _lib.ptr
        R, S = int(n_global + n_cur), self.S
        clock = self._make_clock(R)
        if clock is None:
            return None
        steps = {self.opt._state(p)["step"] for p in list(self.planes) + list(self.dec_w)}
        if len(steps) != 1:
            return None                                   # one step offset serves every tensor
        t0 = steps.pop()
        two = self.use_graph == "two_stream"
        if self._side is None:
            self._streams()
        side = self._side
        cap = torch.cuda.Stream(self.device)
        cap.wait_stream(torch.cuda.current_stream(self.device))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            st = C.c_void_p(cap.cuda_stream)
            st2 = C.c_void_p(side.cuda_stream) if two else st
            opts = self._render_opts(None)[0]
            if not two:
                opts.external_bin, opts.event_after_decode = 0, None
            self.n_active = R
            _lib.check(lib.mne_render_fused(C.byref(self.scene), C.byref(self.rc), R, S, P(self.rays_o), P(self.rays_d),
                                            P(self.tgt_rgb), P(self.tgt_d), P(self.z_vals),
                                            P(self.ray_counts) if self.early_termination else None, P(self.packed),
                                            P(self.coef), P(self.rgb), P(self.depth), P(self.raw), P(self.ray_sums),
                                            P(self.tape), R * S, P(self.tape_rows), P(self.ray_tiles), C.byref(self.bins),
                                            P(self.ws), self.ws_bytes, C.byref(opts), st), "mne_render_fused")
            early = two and self._ev_decode is not None
            if early:
                side.wait_event(self._ev_decode)
                self._tile_bin(0, opts, st2)
                if self._order_early:
                    _lib.check(lib.mne_tile_order(C.byref(self.scene), C.byref(self.bins), st2), "mne_tile_order")
                self._ev[2].record(side)
            if two:
                side.wait_stream(cap)
            if not (early and self._order_early):
                _lib.check(lib.mne_tile_order(C.byref(self.scene), C.byref(self.bins), st2), "mne_tile_order")
            for k in range(len(self.planes)):
                self.plane_opt[k].step = t0 + 1
            _lib.check(lib.mne_tile_adam(C.byref(self.scene), self.plane_opt, P(self.tape), C.byref(self.bins),
                                         C.byref(clock), st2), "mne_tile_adam")
            self._decoder_chain(R, S, st, clock=clock)      # leaves the NEXT iteration's decoder tables in self.packed
            _lib.check(lib.mne_clock_advance(P(self.clk_iter), None, st), "mne_clock_advance")
            if early:
                cap.wait_event(self._ev[2])               # the appends have read this batch: its buffers may be overwritten
            self._sample_batch(kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur, None, None, None, st, clock=clock)
            if two:
                cap.wait_stream(side)
            _lib.check(lib.mne_clock_advance(None, P(self.clk_step), st), "mne_clock_advance")
        return {"graph": g, "clock": clock, "t0": t0, "iter_dev": None, "step_dev": None, "packed_for": None,
                "keep": (self.clk_iter, self.clk_step, self.clk_table)}

    def _clock_covers(self, rec, step_next):
        """The bias-correction table of a recorded graph holds ``n_table`` steps; beyond it the kernel clamps, which is exact
        only once 1 - beta^n has rounded to 1 for both betas (true for (0.9, 0.99) at 8192, not for e.g. beta2 = 0.9999)."""
        ck = rec["clock"]
        if step_next <= ck.n_table:
            return True
        return (1.0 - ck.beta1 ** ck.n_table) == 1.0 and (1.0 - ck.beta2 ** ck.n_table) == 1.0

    def _replay(self, rec):
        """Launch the recorded iteration; the device clock is first brought in line with the host's counters when
        eager steps ran in between."""
        if self._planes_pending:
            torch.cuda.current_stream(self.device).wait_event(self._ev[1])
            self._planes_pending = False
        step_now = self.opt._state(self.planes[0])["step"]
        clk_iter, clk_step, _ = rec["keep"]
        if rec["iter_dev"] != self.iteration:
            clk_iter.fill_(self.iteration)
        if rec["step_dev"] != step_now - rec["t0"]:
            clk_step.fill_(step_now - rec["t0"])
        if rec["packed_for"] != self.iteration:           # after eager steps: the decoder tables are one Adam step old
            _lib.check(self.lib.mne_pack_decoder(C.byref(self.scene), _lib.ptr(self.packed), _lib.stream_for(self.rays_o)),
                       "mne_pack_decoder")
        rec["graph"].replay()
        self.iteration += 1
        for p in list(self.planes) + list(self.dec_w):
            self.opt._state(p)["step"] += 1
        rec["iter_dev"], rec["step_dev"], rec["packed_for"] = self.iteration, step_now + 1 - rec["t0"], self.iteration
        self._packed_key = self._decoder_key()

    def step(self, kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur, idx_global=None, idx_cur=None, u=None,
             prefetch=False):
        """One mapping iteration.  kf_rays [*,7] / cur_rays [H*W,7] / poses [N,4,4] live on the device;
        idx_global / idx_cur (int64 device tensors) and u [R,S] reproduce a host-RNG batch.
        prefetch=True promises that the NEXT call has the same ray sources and poses (the iterations of
        one keyframe, mp_slam/mapper.py:133): its batch is then drawn while this iteration's plane update
        still runs on the side stream (device sampler only; the batch buffers rays_o/tgt_*/z_vals then already
        hold the next batch when this call returns, and the plane update is only joined by the next step or by
        ``synchronize()``; a step with prefetch=False, as Mapper's last one, leaves everything joined)."""
        lib, P = self.lib, _lib.ptr
        R, S = int(n_global + n_cur), self.S
        if R > self.R or R < 1:
            raise ValueError(f"this FusedStep was built for at most {self.R} rays, got {n_global}+{n_cur}")
        self.n_active = R            # every buffer is [ray][...]: a smaller batch uses the leading rows
        self._refresh_pointers()
        (main, st), (side, st2) = self._streams()
        ev = self._ev if side is not None else [None] * 4
        host_batch = idx_global is not None or idx_cur is not None or u is not None
        key = self._batch_key(kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur)
        # ---- steady state (batch already drawn, next call promised alike): one hipGraphLaunch instead of ~14 launches
        if (self.use_graph and prefetch and not host_batch and side is not None and self._prefetched == key
                and self.events is None and not self.shared_decoder and self.tile_overlap is None):
            gkey = self._graph_key(kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur)
            rec = self._graphs.get(gkey)
            if rec is None:
                if len(self._graphs) >= 4:
                    self._graphs.clear()
                self.synchronize()
                rec = self._graphs[gkey] = self._record((kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur)) or False
            steps_now = {self.opt._state(p)["step"] for p in list(self.planes) + list(self.dec_w)}
            if rec and len(steps_now) == 1 and self._clock_covers(rec, steps_now.pop() + 1):   # else: this step runs eagerly
                self._replay(rec)
                self._prefetched = self._batch_key(kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur)
                return
        # Stream roles.  The caller's stream carries the whole dependency chain of the decoder:
        #   [batch] -> pack -> render (decode .. backward) -> wgrad -> decoder Adam -> loss scalars -> [next batch]
        # and the plane update (tile_order + tile_adam, HBM-bound, the longest kernel) runs on the side stream
        # between two events: "backward done" (recorded on the caller's stream) and "planes updated" (waited for
        # by the caller's stream right before the next decode, or at the end of a step without prefetch).  The
        # caller's stream has long parked on that wait when the plane update finishes, so the next decode starts
        # ~2 us later; joining the other way round (the decoder chain on the side stream) put an event wait
        # BEHIND the long kernel on the same queue and cost 18-20 us per iteration (profiles/r01_gap_analysis.txt).
        if host_batch or self._prefetched != key:
            self._sample_batch(kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur, idx_global, idx_cur, u, st)
        self._prefetched = None
        self._pack_if_stale(st)
        if self._planes_pending:                                  # previous step's plane update (side stream)
            main.wait_event(ev[1])
            self._planes_pending = False
        e0 = self._mark("render")
        opts, marks = self._render_opts(main)
        _lib.check(lib.mne_render_fused(C.byref(self.scene), C.byref(self.rc), R, S, P(self.rays_o), P(self.rays_d),
                                        P(self.tgt_rgb), P(self.tgt_d), P(self.z_vals),
                                        P(self.ray_counts) if self.early_termination else None, P(self.packed),
                                        P(self.coef), P(self.rgb), P(self.depth), P(self.raw), P(self.ray_sums), P(self.tape),
                                        R * S, P(self.tape_rows), P(self.ray_tiles),
                                        C.byref(self.bins) if self.bins is not None else None,
                                        P(self.ws), self.ws_bytes, C.byref(opts), st),
                   "mne_render_fused")
        self._mark("render", e0)
        if marks:
            for name, a, b in self.MARK_NAMES:
                if name != "bin_kernel" or self._ev_decode is None:      # (timed on the side stream when it runs there)
                    self.events.setdefault(name, []).append((marks[a], marks[b]))
        if self.bins is not None:
            # ---- plane update on the side stream; with _ev_decode the list appends (mne_tile_bin) are there too, pass 0
            # beside the backward kernels (it waits for the decode only), pass 1 (deferred rays) behind them
            if self._ev_decode is not None and side is not None:
                side.wait_event(self._ev_decode)
                e0 = self._mark("bin_kernel", stream=side)
                self._tile_bin(0, opts, st2)
                self._mark("bin_kernel", e0, stream=side)
                # processing order of the tiles from the list lengths as they are now -- the deferred rays' appends (a
                # handful in steady state) come later and do not change which lists are the heavy ones; tile_adam_kernel
                # reads the final lengths itself.  Keeps tile_order_kernel (12 us + a launch gap) off the critical path.
                # (Only while the kernel can snapshot every list length, MNE_TILE_ORDER_SNAPSHOT: beyond that it would re-read
                # lengths that the deferred rays' appends are changing under it)
                if self._order_early:
                    _lib.check(lib.mne_tile_order(C.byref(self.scene), C.byref(self.bins), st2), "mne_tile_order")
                ev[2].record(side)                      # "appends done": the batch buffers (rays, z, targets, coefficients)
                self._bin_pending = True                # may be overwritten by the next batch only after this
            if (self._ev_decode is None or side is None) and self.split_bin_calls:
                self._tile_bin(0, opts, st2)            # (tests) pass 0 as a call of its own, behind the whole render call
            self._after(side, ev[0], main)              # the deferred rays' appends are part of the render call (caller's stream)
            if self._ev_decode is None or side is None or not self._order_early:
                _lib.check(lib.mne_tile_order(C.byref(self.scene), C.byref(self.bins), st2), "mne_tile_order")
            for k, p in enumerate(self.planes):
                stt = self.opt._state(p)
                stt["step"] += 1
                self.plane_opt[k].step = stt["step"]
            e0 = self._mark("adam", stream=side)
            if self.tile_overlap is not None:
                # EXTENSION: the peers' gradients of the shared cells join ours before the update.  Export -> one batched
                # exchange per peer (RCCL point-to-point on the side stream; gloo on the host) -> update.
                from . import dist as mdist
                _lib.check(lib.mne_tile_grad_export(C.byref(self.scene), P(self.tape), C.byref(self.bins),
                                                    C.byref(self.tile_overlap), st2), "mne_tile_grad_export")
                with torch.cuda.stream(side) if side is not None else _null_ctx():
                    n_nb = len(self.overlap_peers)
                    mdist.exchange_buffers([peer for peer, _ in self.overlap_peers], self.ov_send[:n_nb], self.ov_recv[:n_nb])
                    if self.overlap_group_axis is not None:     # planes every agent holds: the total over all agents
                        mdist.allreduce_sum_into(self.ov_send[n_nb], self.ov_recv[n_nb])
                _lib.check(lib.mne_tile_adam_shared(C.byref(self.scene), self.plane_opt, P(self.tape), C.byref(self.bins),
                                                    C.byref(self.tile_overlap), None, st2), "mne_tile_adam_shared")
            else:
                _lib.check(lib.mne_tile_adam(C.byref(self.scene), self.plane_opt, P(self.tape), C.byref(self.bins), None, st2),
                           "mne_tile_adam")
            self._mark("adam", e0, stream=side)
            if side is not None:
                ev[1].record(side)                      # "planes updated": what the next decode waits for
                self._planes_pending = True
            # ---- decoder chain on the caller's stream, concurrent with the plane update
            self._decoder_chain(R, S, st)
        else:
            _lib.check(lib.mne_decoder_wgrad(C.byref(self.scene), P(self.tape), P(self.ray_tiles), R, S, P(self.partials),
                                             P(self.dec_grad), self.model.wgrad_impl, st), "mne_decoder_wgrad")
            if self.shared_decoder:
                from . import dist as mdist
                mdist.allreduce_mean_(self.dec_grad)
            if self.overlap_peers:
                from . import dist as mdist
                geo = mdist.plane_geometry(self.model)
                for peer, peer_geo in self.overlap_peers:
                    mdist.exchange_overlap_gradients(self.grads, geo, peer, peer_geo)
            e0 = self._mark("adam")
            self.opt.step(zero_grad=True, grad_buffers=self.grad_map)
            self._mark("adam", e0)
            self._packed_key = None
            _lib.check(lib.mne_loss_finalize(R, S, P(self.ray_sums), P(self.counts), P(self.losses), st), "mne_loss_finalize")
        self.iteration += 1
        if self._bin_pending:                                     # the side stream's appends read this batch's buffers
            main.wait_event(ev[2])
            self._bin_pending = False
        if prefetch and not host_batch and side is not None:
            self._sample_batch(kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur, None, None, None, st)
            self._prefetched = self._batch_key(kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur)
        elif self._planes_pending:                                # whatever the caller enqueues next sees the updated planes
            main.wait_event(ev[1])
            self._planes_pending = False

    def check(self):
        """Host-synchronising sanity check (call once per mapping_optimize, not per step): raises if list entries
        were lost because a caller-chosen spill_capacity was too small."""
        self.synchronize()
        if self.bins is not None and int(self.dropped.item()) != 0:
            raise RuntimeError(f"binned scatter lost {int(self.dropped.item())} list entries: spill_capacity too small")

    def synchronize(self):
        """Make the caller's current stream wait for a plane update still running on the side stream (only needed
        after a step with prefetch=True if planes are read before the next step)."""
        if self._planes_pending:
            torch.cuda.current_stream(self.device).wait_event(self._ev[1])
            self._planes_pending = False

    N_MARKS = 6

    def _render_opts(self, stream, force=False):
        """(mne_fused_opts_t, marks): the per-call extras of the fused render -- the adaptive-schedule words and, when
        ``events`` is set (bench.py's live kernel timing), six HIP events the call records between its kernels."""
        o = _lib.FusedOpts()
        if self.adapt_state is not None:
            o.adapt_state = self.adapt_state.data_ptr()
        if self._ev_decode is not None:                # the list appends run on the side stream beside the backward kernels
            o.external_bin, o.event_after_decode = 1, self._ev_decode.cuda_event
        elif getattr(self, "split_bin_calls", False) and self.bins is not None:
            o.external_bin = 1
        marks = None
        if self.events is not None and self.rays_o.is_cuda and (self.bins is not None or force):
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(self.N_MARKS)]
            for e in marks:
                e.record(stream)                     # recorded once here so that the handle exists
            self._mark_arr = (C.c_void_p * self.N_MARKS)(*[e.cuda_event for e in marks])
            o.timing_events, o.n_timing_events = self._mark_arr, self.N_MARKS
        return o, marks

    MARK_NAMES = (("gather_kernel", 0, 1), ("decode_kernel", 1, 2), ("ray_kernel", 2, 3), ("deferred_pass", 3, 4),
                  ("bin_kernel", 4, 5))

    def _mark(self, name, start=None, stream=None):
        """HIP events on the launch stream around one launch (bench.py's live kernel timing)."""
        if self.events is None or not self.rays_o.is_cuda:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream) if stream is not None else e.record()
        if start is not None:
            self.events.setdefault(name, []).append((start, e))
        return e

    def loss_dict(self):
        L = self.losses
        n = getattr(self, "n_active", self.R)
        return {"rgb": self.rgb[:n], "depth": self.depth[:n], "rgb_loss": L[_lib.L_RGB], "depth_loss": L[_lib.L_DEPTH],
                "co_sdf_loss": L[_lib.L_CO_SDF], "co_fs_loss": L[_lib.L_CO_FS], "e_fs_loss": L[_lib.L_E_FS],
                "e_center_loss": L[_lib.L_E_CENTER], "e_tail_loss": L[_lib.L_E_TAIL], "psnr": L[_lib.L_PSNR:_lib.L_PSNR + 1]}


class HashFusedStep(FusedStep):
    """The mapping iteration of ``model.scene_rep_hash.HashJointEncoding`` (EXTENSION: the hash-grid wiring the reference
    keeps commented out, parity unpinned).  Same batch sampler, decoder / compositing / loss kernels and weight-gradient
    kernels as FusedStep; the plane gather and the binned plane update are replaced by the grid's (include/mneslam_hip.h,
    section NS-a).  One iteration on two HIP streams and TWO tapes (alternating: the next gather may overwrite feature
    columns that the weight-gradient pass of this iteration is still reading):

      caller's stream   gather(i) [rows the early termination can decode] -> wait "decoder(i-1) updated" -> decode .. backward
                        (mne_render_fused_features) -> *event "backward done"* -> table update (mne_hash_slice_adam: binned rows,
                        exact fixed-point LDS sums, Adam fused) -> loss scalars -> batch(i+1) -> gather(i+1) ...
      side stream       wait "backward done" -> mne_decoder_wgrad (no reduction) -> mne_decoder_update (reduce + decoder Adam)
                        -> decoder tables of the next render -> *event "decoder updated"*

    The table's critical chain is  backward -> table update -> next gather -> next decode;  the decoder's chain (~0.2 ms with
    2x64 decoders) runs beside the table update AND the next gather."""

    def __init__(self, model, optimizer, config, n_rays, device, is_co_sdf=None):
        if not isinstance(optimizer, FusedAdam):
            raise TypeError("the fused mapping step needs mneslam_amd.optim.FusedAdam")
        if getattr(model, "embed_fn", None) is None:
            raise TypeError("HashFusedStep drives a HashJointEncoding")
        self.scatter, self.shared_decoder, self.overlap_peers = "hash", False, []
        self.lib = _lib.load()
        self.model, self.opt, self.cfg, self.R = model, optimizer, config, int(n_rays)
        self.device = torch.device(device)
        self.info = model._info()
        self.rc = self.info["render_cfg"]
        self.S = self.lib.mne_num_samples(C.byref(self.rc), 1)
        self.planes, self.grads, self.bins = [], None, None
        self.table = model.embed_fn.params
        if self.table.device.type != self.device.type:
            raise ValueError("the hash table must live on the compute device")
        self.grid_cfg = model.embed_fn.cfg
        self.dec_w = model.decoder.hip_weights()
        self.group_of = {p: g for g in optimizer.param_groups for p in g["params"]}
        self.group_of_dec = self.group_of[self.dec_w[0]]
        self.scene = hip_path.scene_struct(self.info, [], [w.data for w in self.dec_w], None)
        self.scene.n_sets = 1
        self._alloc_buffers(config, is_co_sdf)
        self.tape.zero_()                                  # feature columns the grid does not fill must read as zero
        self.tapes = [self.tape, torch.zeros_like(self.tape)]
        st = optimizer._state(self.table)
        if "grad_buffer" not in st:
            st["grad_buffer"] = torch.zeros_like(self.table.data)
        self.table_grad = st["grad_buffer"]
        # table update: "slices" = slice-binned rows + exact LDS sums + fused Adam (no float atomics, no gradient buffer; default);
        # "atomics" = run-reduced global atomics into a gradient buffer + the streaming Adam kernel (cross-check)
        self.table_update = os.environ.get("MNE_HASH_UPDATE", "slices")
        if self.table_update == "atomics":
            self.grad_map[self.table] = self.table_grad
        else:
            self.hash_ws_bytes = self.lib.mne_hash_workspace_bytes(C.byref(self.grid_cfg), self.R, self.S)
            self.hash_ws = torch.zeros(self.hash_ws_bytes, device=self.device, dtype=torch.uint8)
            grp = next(g for g in optimizer.param_groups if any(p is self.table for p in g["params"]))
            o = self.table_opt = _lib.PlaneOpt()
            o.lr, (o.beta1, o.beta2) = float(grp["lr"]), map(float, grp["betas"])
            o.eps, o.weight_decay = float(grp["eps"]), float(grp["weight_decay"])
        self._finish_init(overlap=os.environ.get("MNE_NO_OVERLAP", "0") != "1")
        self.use_graph = None
        self._decoder_pending = False
        self._gathered = None                 # (batch key, tape index): the first-pass rows of that batch are in that tape already

    def _refresh_pointers(self):
        w_sdf0, w_sdf1, w_col0, w_col1 = self.dec_w
        self.scene.w_sdf0, self.scene.w_sdf1 = w_sdf0.data_ptr(), w_sdf1.data_ptr()
        self.scene.w_col0, self.scene.w_col1 = w_col0.data_ptr(), w_col1.data_ptr()

    def synchronize(self):
        if self._decoder_pending:
            torch.cuda.current_stream(self.device).wait_event(self._ev[1])
            self._decoder_pending = False

    def check(self):
        self.synchronize()

    def _gather(self, tape, st, e_name="hash_gather"):
        """The grid features of the rows the first pass of the render can decode (every row without early termination)."""
        R, S, P = self.n_active, self.S, _lib.ptr
        e0 = self._mark(e_name)
        _lib.check(self.lib.mne_hash_gather(C.byref(self.grid_cfg), C.byref(self.scene), R, S, P(self.rays_o), P(self.rays_d),
                                            P(self.z_vals), P(self.ray_counts) if self.early_termination else None,
                                            P(self.table.data), P(tape), st), "mne_hash_gather")
        self._mark(e_name, e0)

    def step(self, kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur, idx_global=None, idx_cur=None, u=None,
             prefetch=False):
        lib, P = self.lib, _lib.ptr
        R, S = int(n_global + n_cur), self.S
        if R > self.R or R < 1:
            raise ValueError(f"this step was built for at most {self.R} rays, got {n_global}+{n_cur}")
        self.n_active = R
        self._refresh_pointers()
        st = _lib.stream_for(self.rays_o)
        main = side = None
        slices = self.table_update != "atomics"
        if self.rays_o.is_cuda and self.overlap and slices:
            if self._side is None:
                self._side = torch.cuda.Stream(self.device)
                self._ev = [torch.cuda.Event() for _ in range(2)]
                for e in self._ev:
                    e.record(torch.cuda.current_stream(self.device))        # (recorded once so that the handles exist)
            main, side = torch.cuda.current_stream(self.device), self._side
        host_batch = idx_global is not None or idx_cur is not None or u is not None
        key = self._batch_key(kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur)
        if host_batch or self._prefetched != key:
            self._sample_batch(kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur, idx_global, idx_cur, u, st)
            self._gathered = None                                   # a re-drawn batch: rows gathered for the prefetched one are stale
        self._prefetched = None
        sc, gc = C.byref(self.scene), C.byref(self.grid_cfg)
        ti = self.iteration & 1
        tape = self.tapes[ti]
        self.tape = tape                                            # (what tests and bench read back)
        if self._gathered != (key, ti):                             # not pre-gathered at the end of the previous step
            self._gather(tape, st)
        self._gathered = None
        if self._decoder_pending:                                   # previous step's decoder update + tables (side stream)
            main.wait_event(self._ev[1])
            self._decoder_pending = False
        self._pack_if_stale(st)
        e0 = self._mark("render")
        opts, marks = self._render_opts(torch.cuda.current_stream(self.device) if self.rays_o.is_cuda else None, force=True)
        opts.adapt_state = None                           # (the hash gather sizes its row set from the a-priori prefix)
        opts.external_bin, opts.event_after_decode = 0, None
        opts.features_pregathered = 1
        _lib.check(lib.mne_render_fused_features(sc, C.byref(self.rc), R, S, P(self.rays_o), P(self.rays_d), P(self.tgt_rgb),
                                                 P(self.tgt_d), P(self.z_vals),
                                                 P(self.ray_counts) if self.early_termination else None, P(self.packed), P(self.coef), P(self.rgb),
                                                 P(self.depth), P(self.raw), P(self.ray_sums), P(tape), R * S,
                                                 P(self.tape_rows), P(self.ray_tiles), P(self.ws), self.ws_bytes,
                                                 gc, P(self.table.data), C.byref(opts), st),
                   "mne_render_fused_features")
        self._mark("render", e0)
        if marks:
            for name, a, b in (("decode_kernel", 1, 2), ("ray_kernel", 2, 3), ("deferred_pass", 3, 4)):
                self.events.setdefault(name, []).append((marks[a], marks[b]))
        e0 = self._mark("hash_scatter")
        if not slices:
            _lib.check(lib.mne_hash_scatter(gc, sc, R, S, P(self.rays_o), P(self.rays_d), P(self.z_vals), P(tape),
                                            P(self.ray_tiles), P(self.table_grad), st), "mne_hash_scatter")
        else:
            stt, o = self.opt._state(self.table), self.table_opt
            stt["step"] += 1
            o.m, o.v, o.step = stt["exp_avg"].data_ptr(), stt["exp_avg_sq"].data_ptr(), stt["step"]
            # "rows binned": recorded by the call between its binning and its slice launch -- the decoder chain starts there
            _lib.check(lib.mne_hash_slice_adam(gc, sc, R, S, P(self.rays_o), P(self.rays_d), P(self.z_vals), P(tape),
                                               P(self.ray_tiles), P(self.table.data), C.byref(o), P(self.hash_ws),
                                               self.hash_ws_bytes, C.c_void_p(self._ev[0].cuda_event) if side is not None else None, st),
                       "mne_hash_slice_adam")
        if side is not None:
            side.wait_event(self._ev[0])
            # decoder chain (weight gradients -> reduce + decoder Adam -> next tables) on the side stream, beside the slice /
            # Adam launch of the table update and the next gather (not beside the latency-bound binning, which it slowed by 70 %)
            with torch.cuda.stream(side):
                e1 = self._mark("wgrad", stream=side)
                self._decoder_chain_hash(R, S, tape, C.c_void_p(side.cuda_stream))
                self._mark("wgrad", e1, stream=side)
                self._ev[1].record(side)
            self._decoder_pending = True
        self._mark("hash_scatter", e0)
        if side is None:
            e0 = self._mark("wgrad")
            if slices:
                self._decoder_chain_hash(R, S, tape, st)
            else:
                _lib.check(lib.mne_decoder_wgrad(sc, P(tape), P(self.ray_tiles), R, S, P(self.partials), P(self.dec_grad),
                                                 self.model.wgrad_impl, st), "mne_decoder_wgrad")
                self.opt.step(zero_grad=True, grad_buffers=self.grad_map)            # table + decoder tensors, one launch
                self._packed_key = None
            self._mark("wgrad", e0)
        _lib.check(lib.mne_loss_finalize(R, S, P(self.ray_sums), P(self.counts), P(self.losses), st), "mne_loss_finalize")
        self.iteration += 1
        if prefetch and not host_batch:
            self._sample_batch(kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur, None, None, None, st)
            self._prefetched = self._batch_key(kf_rays, n_kf_rays, n_save, cur_rays, poses, n_global, n_cur)
            if side is not None:        # the next batch's rows into the OTHER tape, while the decoder chain still reads this one
                self._gather(self.tapes[self.iteration & 1], st, e_name="hash_gather")
                self._gathered = (self._prefetched, self.iteration & 1)
        else:
            self.synchronize()                                # whatever the caller enqueues next sees the updated decoder

    def _decoder_chain_hash(self, R, S, tape, st):
        """wgrad (no reduction) -> decoder_update_kernel (fixed-order reduce + decoder Adam) -> the next render's tables: three
        launches, as in FusedStep (the loss scalars stay on the caller's stream: they read this batch's mask counts)."""
        lib, P = self.lib, _lib.ptr
        fused = self.model.wgrad_impl == 0
        _lib.check(lib.mne_decoder_wgrad(C.byref(self.scene), P(tape), P(self.ray_tiles), R, S, P(self.partials),
                                         P(self.dec_grad), 3 if fused else self.model.wgrad_impl, st), "mne_decoder_wgrad")
        if not fused:
            self.opt.step(zero_grad=False, grad_buffers=self.dec_grad_views)
            self._packed_key = None
            return
        w_sdf0, w_sdf1, w_col0, w_col1 = self.dec_w
        grp = self.group_of_dec
        o = _lib.DecoderOpt()
        steps = set()
        for k, w in enumerate((w_col0, w_col1, w_sdf0, w_sdf1)):       # decoder.parameters() order
            stt = self.opt._state(w)
            stt["step"] += 1
            steps.add(stt["step"])
            o.m[k], o.v[k] = stt["exp_avg"].data_ptr(), stt["exp_avg_sq"].data_ptr()
        if len(steps) != 1:
            raise RuntimeError("the decoder tensors must share one Adam step count")
        o.lr, (o.beta1, o.beta2) = float(grp["lr"]), map(float, grp["betas"])
        o.eps, o.weight_decay = float(grp["eps"]), float(grp["weight_decay"])
        o.step = steps.pop()
        _lib.check(lib.mne_decoder_update(C.byref(self.scene), P(self.partials), R, P(self.dec_grad), C.byref(o),
                                          S, None, None, None, None, st), "mne_decoder_update")
        _lib.check(lib.mne_pack_decoder(C.byref(self.scene), P(self.packed), st), "mne_pack_decoder")
        self._packed_key = self._decoder_key()
