#!/usr/bin/env python3
"""Idle time between the kernels of the iteration's critical chain, from a rocprofv3 kernel trace (rocpd sqlite).
usage: gap_analysis.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
chain = ["decode_kernel", "composite_kernel", "scan_kernel", "backward_kernel", "tile_order_kernel", "tile_adam_kernel"]


def short(n):
    for c in chain + ["wgrad_fused", "wgrad_reduce", "adam_kernel", "loss_finalize", "sample_rays", "sample_z", "counts_reduce",
                      "loss_coef", "pack_decoder"]:
        if c in n:
            return c
    return None


ev = [(short(n), s, e) for n, s, e in rows if short(n)]
gaps, durs = {}, {}
prev = None
for n, s, e in ev:
    if n not in chain:
        continue
    if prev is not None:
        key = f"{prev[0]} -> {n}"
        gaps.setdefault(key, []).append((s - prev[2]) / 1e3)
    durs.setdefault(n, []).append((e - s) / 1e3)
    prev = (n, s, e)
print("critical-chain kernel durations (avg us):")
for n in chain:
    v = durs.get(n, [])
    print(f"  {n:22s} {sum(v) / max(len(v), 1):8.1f}  (n={len(v)})")
print("idle gaps between consecutive chain kernels (avg us; skip first 20):")
tot = 0.0
for k, v in gaps.items():
    v = v[20:] if len(v) > 40 else v
    a = sum(v) / max(len(v), 1)
    tot += a
    print(f"  {k:44s} {a:8.2f}")
print(f"  sum of gaps per iteration {tot:.1f} us")
# when does the side chain finish relative to tile_adam?
last_side, slack = None, []
for n, s, e in ev:
    if n == "pack_decoder":
        last_side = e
    if n == "decode_kernel" and last_side is not None:
        slack.append((s - last_side) / 1e3)
print("decode start minus pack_decoder end (side chain slack, avg us):", sum(slack[20:]) / max(len(slack[20:]), 1))
