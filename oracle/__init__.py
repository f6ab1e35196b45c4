"""CPU oracle for the MNE-SLAM mapping hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32/fp64) restatement of the reference
algorithm on the path  model/scene_rep.py + model/encodings.py +
model/decoder.py + model/utils.py + mp_slam/mapper.py:118-162 +
mneslam_mp.py:342-372,431-469 + model/keyframe.py:64-103  of
dtc111111/MNESLAM.  Every function cites the reference file:line it follows.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the *checker* / reported CPU
baseline, never as the thing measured or shipped.  Nothing under
``mneslam_amd/`` imports this package; the product path raises if the HIP
library is missing instead of falling back to it.

Pinning status
--------------
* Tri-plane lookup, decoder, compositing, losses, loss weighting, Adam and the
  mapping loop are PINNED against golden vectors captured by importing the
  reference itself on CPU (``tests/golden/make_golden.py``; fixtures in
  ``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).
* OneBlob and the hash/dense multiresolution grid live in tinycudann, which is
  NOT vendored in the reference tree (requirements.txt:120, un-pinned git
  HEAD).  ``oracle/oneblob.py`` and ``oracle/hashgrid.py`` restate tinycudann's
  published algorithm as this build's frozen spec: **parity unpinned** for
  those two encodings (the golden vectors use this spec for the OneBlob stub).
"""
