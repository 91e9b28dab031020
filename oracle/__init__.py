"""CPU oracles for the two hot paths (TEST INFRASTRUCTURE ONLY).

Nothing under ``oracle/`` is shipped or measured as the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Pinning status (details: oracle/ASSUMPTIONS.md):

* BA, in-tree half — PINNED.  ``oracle/_ref/libvins_ref.so`` is the reference's own ``estimator.cpp``,
  ``feature_manager.cpp``, ``factor/*`` and ``utility/utility.*`` compiled UNCHANGED from /root/reference by
  ``oracle/Makefile`` (target ``ref``) against header stand-ins for the absent Eigen / Ceres / ROS / OpenCV
  (``oracle/ref_stubs``).  ``tests/test_ref_parity.py`` holds both restatements (``ba_numpy.py``, ``ba_cpu.cpp``) and the
  HIP path to it: factor residuals / Jacobians, pre-integration, problem construction, gauge fix, marginalization,
  triangulation, window shift.  ``tests/golden/golden_ba.npz`` is generated from it.
* BA, third-party half — restated, unpinned: the Ceres trust-region minimiser (``ref_stubs/ceres/solver_stub.cc``,
  ``ba_numpy.solve``, ``ba_cpu.cpp``) and the dense kernels of Eigen (inverse, LLT, eigen-solver, SVD inside the
  stand-in).  Ceres 1.14 / Eigen 3 are not in /root/reference and not in this image.
* FE — PARITY UNPINNED.  The arithmetic (OpenCV 3.3.1: GFTT, pyramidal LK, CLAHE, findFundamentalMat) is third-party
  and absent; ``fe_cpu.cpp`` / ``fe_numpy.py`` restate the published algorithms.
"""
