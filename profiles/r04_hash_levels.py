"""Where the hash-grid table update spends its time: mne_hash_slice_adam timed stand-alone (HIP events, nothing beside it) on the
tape of a real iteration for grids of the first k levels, k = 1..16 -- differences = the cost of each level (offsets + pack + bin
+ slice + finish kernels).  usage: r04_hash_levels.py [library.so]"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mneslam_amd import _lib
if len(sys.argv) > 1:
    _lib.load(sys.argv[1])
import bench
from mneslam_amd import configs
cfg = configs.WORKLOADS["office0_hash"][0]()
ag = bench.Agent(cfg, torch.device("cuda"), seed=0, n_keyframes=20)
for i in range(60):
    ag.step(prefetch=i < 59)
torch.cuda.synchronize()
fs = ag.fused
lib, P = fs.lib, _lib.ptr
R, S = fs.R, fs.S
st = _lib.stream_for(fs.rays_o)
print("rows with gradient:", int((fs.ray_tiles[:R].long() * 32).clamp(max=S).sum()), "library:", lib._name)
prev = 0.0
for k in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16):
    gc = _lib.GridCfg()
    for f in ("n_features", "base_resolution", "log2_hashmap_size", "grid_type", "per_level_scale"):
        setattr(gc, f, getattr(fs.grid_cfg, f))
    gc.n_levels = k
    nb = lib.mne_hash_workspace_bytes(C.byref(gc), R, S)
    ws = torch.zeros(nb, device="cuda", dtype=torch.uint8)
    o = fs.table_opt
    stt = fs.opt._state(fs.table)
    o.m, o.v, o.step = stt["exp_avg"].data_ptr(), stt["exp_avg_sq"].data_ptr(), max(stt["step"], 1)
    def call():
        _lib.check(lib.mne_hash_slice_adam(C.byref(gc), C.byref(fs.scene), R, S, P(fs.rays_o), P(fs.rays_d), P(fs.z_vals), P(fs.tape),
                                           P(fs.ray_tiles), P(fs.table.data), C.byref(o), P(ws), nb, None, st), "slice")
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    print(f"levels 0..{k - 1:2d}: {t:8.1f} us   (+{t - prev:7.1f})")
    prev = t
