#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (rocpd sqlite): average counter value per launch, per kernel.
usage: summarize_pmc_db.py <filter-substring> db1 [db2 ...]"""
import sqlite3
import sys

flt = sys.argv[1]
table = {}
for path in sys.argv[2:]:
    db = sqlite3.connect(path)
    for k, c, n, v in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                 "group by kernel_name, counter_name"):
        if flt in k:
            table.setdefault(k[:60], {})[c] = (n, v)
for k, d in table.items():
    print(k)
    for c in sorted(d):
        print(f"    {c:34s} {d[c][1]:16.1f}   (n={d[c][0]})")
