"""render_phase_times.py on the -DRENDER_PROFILE variant build (mneslam_amd/_fuzz/x_prof, `build.build_variant("x_prof", ["-DRENDER_PROFILE"])`)."""
import os, runpy, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mneslam_amd import _lib, build
_lib.load(build.variant_path("x_prof"))
runpy.run_path(os.path.join(REPO, "profiles", "render_phase_times.py"), run_name="__main__")
