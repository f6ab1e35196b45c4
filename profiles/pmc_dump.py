#!/usr/bin/env python3
"""rocprofv3 --pmc pass (rocpd sqlite): per kernel whose name contains `pattern`, the mean of every collected counter per launch
usage: pmc_dump.py results.db pattern"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection where kernel_name like ? "
                  "group by kernel_name, counter_name order by kernel_name, counter_name", (f"%{sys.argv[2]}%",)).fetchall()
last = None
for k, c, n, v in rows:
    if k != last:
        print(k[:60]); last = k
    print(f"    {c:28s} {v / n:16.1f}  ({n} launches)")
