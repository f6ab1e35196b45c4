"""Import harness for the *reference* (dtc111111/MNESLAM at /root/reference).

Only ``tests/golden/make_golden.py`` uses this, and only in the build container
(the reference does not exist on the GPU box).  It makes the reference's hot-path
modules importable on CPU without touching the reference tree:

* third-party modules that are not installed (cv2, lietorch, droid_backends,
  pytorch3d, torchvision, trimesh, open3d, ...) become inert stub modules;
* ``tinycudann`` becomes a stub whose ``Encoding`` supports only OneBlob and
  is backed by this build's frozen OneBlob spec (``oracle/oneblob.py``) -- the
  tinycudann arithmetic is not in the reference tree ("parity unpinned");
* HuggingFace ``datasets`` (installed here) is shadowed by the reference's own
  ``datasets/`` namespace package.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest import mock

REF = os.environ.get("MNESLAM_REFERENCE", "/root/reference")
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))

_MISSING = ["cv2", "lietorch", "droid_backends", "trimesh", "open3d", "pytorch3d",
            "marching_cubes", "skimage", "evo", "colorama", "mathutils", "torchvision",
            "torch_scatter"]


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = mock.MagicMock(name=f"{self.__name__}.{name}")
        setattr(self, name, m)
        return m


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _MISSING:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _make_tcnn_stub():
    import torch
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    from oracle.oneblob import OneBlobEncoding

    tcnn = types.ModuleType("tinycudann")

    class Encoding(OneBlobEncoding):
        def __init__(self, n_input_dims, encoding_config, dtype=torch.float):
            if encoding_config.get("otype") != "OneBlob":
                raise NotImplementedError("reference harness stubs only the OneBlob encoding")
            super().__init__(n_input_dims, encoding_config["n_bins"])

    tcnn.Encoding = Encoding
    tcnn.Network = None
    return tcnn


_installed = False


def install():
    """Install the stubs and put the reference on sys.path (idempotent)."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}")
    sys.meta_path.insert(0, _StubFinder())
    sys.modules["tinycudann"] = _make_tcnn_stub()
    pkg = types.ModuleType("datasets")
    pkg.__path__ = [os.path.join(REF, "datasets")]
    sys.modules["datasets"] = pkg
    sys.path.insert(0, REF)
    _installed = True


def load_config(rel_path):
    """Load one of the reference's YAML configs (its loader resolves inherit_from
    relative to the reference root, config.py:21-23)."""
    install()
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import config as ref_config
        return ref_config.load_config(rel_path)
    finally:
        os.chdir(cwd)
