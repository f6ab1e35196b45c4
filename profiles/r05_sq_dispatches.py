#!/usr/bin/env python3
"""rocprofv3 --pmc passes (rocpd sqlite): counter values PER DISPATCH of the kernels matching a substring, in dispatch order
(the passes run the same program, so dispatch k of one pass is dispatch k of another).
usage: r05_sq_dispatches.py <filter-substring> <last-n-dispatches> db1 [db2 ...]"""
import sqlite3
import sys

flt, last = sys.argv[1], int(sys.argv[2])
for path in sys.argv[3:]:
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    key = "dispatch_id" if "dispatch_id" in cols else cols[0]
    rows = {}
    for k, c, d, v in db.execute(f"select kernel_name, counter_name, {key}, sum(value) from counters_collection "
                                 f"group by kernel_name, counter_name, {key} order by {key}"):
        if flt in k:
            rows.setdefault((k[:70], c), []).append(v)
    for (k, c), vals in sorted(rows.items()):
        print(f"{k}  {c:30s} " + " ".join(f"{v:.4g}" for v in vals[-last:]))
