"""CPU oracles for the two hot paths (TEST INFRASTRUCTURE ONLY).

Nothing under ``oracle/`` is shipped or measured as the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

PARITY UNPINNED: the reference (HKUST-Aerial-Robotics/VINS-Mono) has no tests, no golden
vectors, and its FE arithmetic (OpenCV 3.3.1) and BA minimiser (Ceres 1.14 / Eigen 3) are
third-party libraries that are absent from /root/reference and from this image.  The in-tree
factor math (vins_estimator/src/factor/*) is restated line-by-line; the third-party parts are
restated from their published algorithms (see oracle/ASSUMPTIONS.md).
"""
