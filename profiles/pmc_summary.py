#!/usr/bin/env python3
"""HBM traffic and matrix-pipe busy time per mapping iteration from rocprofv3 --pmc passes (rocpd sqlite), separate
passes with --kernel-trace only.
usage: pmc_summary.py fetch.db write.db sq.db|- out.json out.txt [workload label] [key kernel]
HBM bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B
(MI355X_MICROARCH.md, HBM / rocprofv3 section); both counters are reported in KB.  Per ITERATION = sum over the launches
of a kernel / number of iterations (= tile_adam_kernel launches): decode_kernel and ray_kernel run twice per iteration."""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    return {k: (n, v) for k, n, v in db.execute(
        "select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (counter,))}


LABEL = sys.argv[6] if len(sys.argv) > 6 else "replica_office0_triplane_asWired_2048x128"
KEY = sys.argv[7] if len(sys.argv) > 7 else "tile_adam_kernel"        # the kernel launched once per iteration
fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
have_sq = sys.argv[3] != "-"
mfma = per_kernel(sys.argv[3], "SQ_VALU_MFMA_BUSY_CYCLES") if have_sq else {}
n_it_f = max(n for k, (n, v) in fetch.items() if KEY in k)
n_it_w = max(n for k, (n, v) in write.items() if KEY in k)
n_it_s = max([n for k, (n, v) in mfma.items() if KEY in k] or [1])
rows = []
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, (0, 0.0))[1] / n_it_f, write.get(k, (0, 0.0))[1] / n_it_w
    rows.append((1024.0 * (2.0 * f + w), k, f, w, fetch.get(k, (0, 0))[0] / n_it_f))
rows.sort(reverse=True)
with open(sys.argv[5], "w") as fh:
    fh.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ counters (separate passes, kernel-trace only), bench.py --steps 10 --warmup 3,\n"
             f"# workload {LABEL}.  KB per ITERATION; HBM bytes = 1024 * (2*FETCH_SIZE + WRITE_SIZE)\n")
    fh.write(f"{'kernel':64s} {'launch/it':>9s} {'FETCH_KB':>12s} {'WRITE_KB':>12s} {'HBM_MB':>10s} {'MFMA busy Mcycles':>18s}\n")
    for b, k, f, w, n in rows[:20]:
        fh.write(f"{k[:64]:64s} {n:9.1f} {f:12.1f} {w:12.1f} {b / 1e6:10.1f} {mfma.get(k, (0, 0.0))[1] / n_it_s / 1e6:18.2f}\n")


def total(pred):
    return sum(b for b, k, *_ in rows if pred(k))


def mf(pred):
    return sum(v / n_it_s for k, (n, v) in mfma.items() if pred(k))


render = lambda k: any(t in k for t in ("gather_kernel", "decode_kernel", "ray_kernel", "composite_kernel"))
out = {"workload": LABEL, "path": "fused", "scatter": "binned",
       "hbm_bytes_per_launch": {"adam": total(lambda k: KEY in k), "render": total(render)},
       "mfma_busy_cycles_per_launch": {"adam": 0.0, "render": mf(render)},
       "per_kernel_hbm_bytes_per_iteration": {k[:60]: b for b, k, *_ in rows[:12]},
       "per_kernel_mfma_busy_cycles_per_iteration": {k[:60]: v / n_it_s for k, (n, v) in mfma.items() if v > 0},
       "method": "rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE, SQ counters in separate passes; bytes = 1024*(2*FETCH_SIZE+WRITE_SIZE) "
                 "(gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md); 'render' = gather + decode + ray kernels of one mne_render_fused "
                 "call; SQ_VALU_MFMA_BUSY_CYCLES summed over the chip's 1024 SIMDs"}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(open(sys.argv[5]).read())
