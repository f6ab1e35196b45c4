"""cpu_baseline of bench.py (the CPU oracle on the default workload's shape) at 32 / 64 / 128 host threads: the record behind
"32 threads: more only add contention" (VERDICT r04, weak #11).  usage: python profiles/r05_cpu_threads.py [iters]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from mneslam_amd import configs  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = configs.bench_office0()
for cores in (16, 32, 64, 128):
    r = bench.cpu_baseline(cfg, 20, iters, cores=cores)
    print(json.dumps({"threads": cores, "it_per_s": r["value"], "s_per_iter_min_median_max": r["s_per_iter_min_median_max"],
                      "host_cores": r["host_cores"], "host_cores_usable": r["host_cores_usable"]}), flush=True)
