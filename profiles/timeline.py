#!/usr/bin/env python3
"""One iteration's kernel timeline from a rocprofv3 kernel trace (rocpd sqlite): start / end relative to the
iteration's decode_kernel, stream (queue) id, and a steady-state average over iterations.
usage: timeline.py results.db [first_iter] [n_iters]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n_it = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = db.execute(f"select name, start, end, {q} from kernels order by start").fetchall()
short = lambda n: n.replace("void ", "").split("(")[0].split("<")[0][:26]
marker = sys.argv[4] if len(sys.argv) > 4 else "pack_decoder_kernel"     # one launch per iteration, right before the decode
dec = [i for i, r in enumerate(rows) if marker in r[0]]
acc = {}
for k in range(first, min(first + n_it, len(dec) - 1)):
    t0 = rows[dec[k]][1]
    seen = {}
    for n, s, e, qid in rows[dec[k]:dec[k + 1]]:
        nm = short(n)
        seen[nm] = seen.get(nm, 0) + 1
        key = (nm, seen[nm], qid)
        a = acc.setdefault(key, [0.0, 0.0, 0])
        a[0] += (s - t0) / 1e3; a[1] += (e - t0) / 1e3; a[2] += 1
    a = acc.setdefault(("<next iteration>", 1, -1), [0.0, 0.0, 0])
    a[0] += (rows[dec[k + 1]][1] - t0) / 1e3; a[1] += (rows[dec[k + 1]][1] - t0) / 1e3; a[2] += 1
print(f"{'kernel':28s} {'queue':>6s} {'start us':>9s} {'end us':>9s} {'dur':>8s}")
for (nm, j, qid), (s, e, c) in sorted(acc.items(), key=lambda kv: kv[1][0] / kv[1][2]):
    print(f"{nm:28s} {str(qid):>6s} {s / c:9.1f} {e / c:9.1f} {(e - s) / c:8.1f}")
