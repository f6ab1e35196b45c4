"""profiles/rNN_spill_table.txt from the ISA report the build writes (mneslam_amd/isa_report.json).
    python profiles/spill_table.py > profiles/r03_spill_table.txt"""
import json
import os
import re
import subprocess

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
r = json.load(open(os.path.join(REPO, "mneslam_amd", "isa_report.json")))
names = [k["kernel"] for k in r["kernels"]]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
print("# every kernel of libmneslam_hip.so (mneslam_amd/isa_report.json, written by the build from the compiler's own kernel info)")
print("# kernels carrying the compiler defect of DESIGN.md section 9 (spill store in front of an exec restore): %d" % len(r["hazards"]))
print("%5s %5s %5s %8s %4s %8s %8s  %s" % ("SGPR", "VGPR", "AGPR", "scratchB", "occ", "sgprSpil", "vgprSpil", "kernel"))
for k, d in sorted(zip(r["kernels"], dem), key=lambda x: (-x[0].get("scratch", 0), x[1])):
    d = re.sub(r"\(.*\)$", "", d.replace("void ", ""))
    print("%5d %5d %5d %8d %4d %8d %8d  %s" % (k.get("sgpr", -1), k.get("vgpr", -1), k.get("agpr", -1), k.get("scratch", -1),
                                               k.get("occupancy", -1), k.get("sgpr_spill", -1), k.get("vgpr_spill", -1), d))
