#!/usr/bin/env python3
"""rocprofv3 (rocpd sqlite) kernel trace: mean / min duration of the LAST n launches of every kernel whose name contains `pattern`
usage: last_calls.py results.db n pattern"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n, pat = int(sys.argv[2]), sys.argv[3]
names = [r[0] for r in db.execute("select distinct name from kernels where name like ?", (f"%{pat}%",))]
for name in names:
    d = [r[0] / 1e3 for r in db.execute("select end-start from kernels where name = ? order by start desc limit ?", (name, n))]
    print(f"  {name[:40]:40s} last {len(d):3d} launches: mean {sum(d) / len(d):8.1f} us  min {min(d):8.1f}")
