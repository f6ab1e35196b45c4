"""Timing-only ablations of the hash-grid table update (variant builds with -DHASH_ABL=bits; results are WRONG by construction):
mne_hash_slice_adam stand-alone on the tape of a real iteration, 16 levels.  usage: r04_hash_ablate.py [library.so]   (run under
rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mneslam_amd import _lib
import bench
from mneslam_amd import configs
cfg = configs.WORKLOADS["office0_hash"][0]()
ag = bench.Agent(cfg, torch.device("cuda"), seed=0, n_keyframes=20)
for i in range(40):
    ag.step(prefetch=i < 39)
torch.cuda.synchronize()
fs = ag.fused
R, S = fs.R, fs.S
args = [fs.rays_o, fs.rays_d, fs.z_vals, fs.tape, fs.ray_tiles, fs.table.data]
for name in sys.argv[1:] or ["main"]:
    lib = _lib.load() if name == "main" else C.CDLL(name)
    if name != "main":
        for fn in ("mne_hash_slice_adam", "mne_hash_workspace_bytes"):
            getattr(lib, fn).restype, getattr(lib, fn).argtypes = _lib._PROTOS[fn]
    P, st = _lib.ptr, _lib.stream_for(fs.rays_o)
    nb = lib.mne_hash_workspace_bytes(C.byref(fs.grid_cfg), R, S)
    ws = torch.zeros(nb, device="cuda", dtype=torch.uint8)
    table = fs.table.data.clone()
    o = fs.table_opt
    stt = fs.opt._state(fs.table)
    m, v = stt["exp_avg"].clone(), stt["exp_avg_sq"].clone()
    o.m, o.v, o.step = m.data_ptr(), v.data_ptr(), max(stt["step"], 1)
    def call():
        rc = lib.mne_hash_slice_adam(C.byref(fs.grid_cfg), C.byref(fs.scene), R, S, P(fs.rays_o), P(fs.rays_d), P(fs.z_vals), P(fs.tape),
                                     P(fs.ray_tiles), P(table), C.byref(o), P(ws), nb, None, st)
        assert rc == 0, rc
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    print(f"{os.path.basename(os.path.dirname(name)) if name != 'main' else 'main':>14s}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per table update (bin + slice/Adam + finish), stand-alone")
