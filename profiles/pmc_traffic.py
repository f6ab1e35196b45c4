#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite).
usage: pmc_traffic.py fetch.db write.db out.json out.txt
bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B
(MI355X_MICROARCH.md, HBM / rocprofv3 section); both counters are reported in KB."""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    return {k: (n, v) for k, n, v in db.execute(
        "select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (counter,))}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, (0, 0.0))[1], write.get(k, (0, 0.0))[1]
    rows.append((1024.0 * (2.0 * f + w), k, f, w, fetch.get(k, (0, 0))[0]))
rows.sort(reverse=True)
with open(sys.argv[4], "w") as fh:
    fh.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only), bench.py --steps 10 --warmup 3,\n"
             "# fused + binned, office0 2150x128.  KB per launch as reported; HBM bytes = 1024 * (2*FETCH_SIZE + WRITE_SIZE)\n")
    fh.write(f"{'kernel':72s} {'launches':>8s} {'FETCH_KB':>12s} {'WRITE_KB':>12s} {'HBM_MB':>10s}\n")
    for b, k, f, w, n in rows[:24]:
        fh.write(f"{k[:72]:72s} {n:8d} {f:12.1f} {w:12.1f} {b / 1e6:10.1f}\n")


def total(pred):
    return sum(b for b, k, *_ in rows if pred(k))


out = {"workload": "replica_office0_triplane_asWired_2048x128", "path": "fused", "scatter": "binned",
       "hbm_bytes_per_launch": {
           "adam": total(lambda k: k.startswith("tile_adam_kernel")),
           "render": total(lambda k: any(t in k for t in ("decode_kernel", "composite_kernel", "scan_kernel", "backward_kernel")))},
       "per_kernel_hbm_bytes": {k[:60]: b for b, k, *_ in rows[:12]},
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = 1024*(2*FETCH_SIZE+WRITE_SIZE) "
                 "(gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md); 'render' = decode+composite+scan+backward of one mne_render_fused call"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(open(sys.argv[4]).read())
