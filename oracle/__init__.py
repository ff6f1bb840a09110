"""CPU oracle for the PointRegGPT generative data path.  TEST INFRASTRUCTURE ONLY.

A plain restatement (torch-CPU / numpy, fp32 unless the reference itself promotes to fp64) of the
algorithms on the hot path named by BASELINE.json, each function citing the reference file:line it
follows.  It exists to *check* the HIP path:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
  * nothing under ``pointreggpt_amd/`` imports it, and the product path has no CPU fallback.

Pinning: every function here is checked against golden vectors produced by importing the real reference
in the build container (``tools/make_goldens.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).
Third-party arithmetic the reference delegates to packages that are absent from the image (open3d
crop / voxel_down_sample / KD-tree, torchvision transforms, ema_pytorch key layout, cv2 PNG writing) has
no fixture and is "parity unpinned" — see DESIGN.md.
"""
