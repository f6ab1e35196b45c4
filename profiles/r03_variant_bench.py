"""bench.py on a variant build of the library (mneslam_amd/_fuzz/<name>/): python profiles/r03_variant_bench.py <name> [bench args]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mneslam_amd import _lib, build
name = sys.argv[1]
sys.argv = ["bench.py"] + sys.argv[2:]
if name != "main":
    _lib.load(build.variant_path(name))
import bench
bench.main()
