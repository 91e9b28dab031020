#!/usr/bin/env python3
"""DIFF KIT — goldens of the REAL Ceres for rows B6 / B7 of SURVEY.md 8(a): `ceres::Solve` as
Estimator::optimization() calls it (vins_estimator/src/estimator.cpp:803-818: DENSE_SCHUR, DOGLEG, NUM_ITERATIONS).

Cannot run in the graft image (no Eigen, no Ceres).  On a machine that has Eigen 3 + Ceres 1.14 (and this repo next to a checkout of
the reference):

    make -C oracle ref_real REF=<reference>/vins_estimator/src          # the reference's TUs + oracle/ref_stubs/ref_driver.cpp on the real libraries
    VINS_REF_LIB=oracle/_ref/libvins_ref_real.so python tests/golden/make_golden_ceres.py

writes tests/golden/golden_ceres.npz: for every fixture of tests/ba_fixtures.BRANCH_FIXTURES (each forces one branch of the
trust-region loop) and for chained windows with a prior / extrinsic / td, the per-iteration rows [valid, accepted, cost, candidate cost,
model cost change, radius, step norm] recorded from ceres::IterationSummary by the link-time wrapper
(oracle/ref_stubs_real/ceres_real_trace.cc), termination type, costs, and the gauge-fixed states after double2vector().
tests/test_golden.py::test_restated_minimiser_against_real_ceres picks the file up when present and holds oracle/ba_numpy.py and
oracle/ba_cpu.cpp to it (and, with -m gpu, the HIP path).  Until then B6 / B7 stay PARITY UNPINNED.

Run against the stand-in build (default library) the script refuses to write: those would be goldens of the restatement."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
import ba_fixtures as FX  # noqa: E402
from oracle import ref as R  # noqa: E402
from vins_mono_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
COLS = ('valid', 'accepted', 'cost', 'cost_cand', 'model_change', 'radius', 'step_norm')


def rows_of(sm):
    return np.array([[float(it.get(c, 0) or 0) for c in COLS] for it in sm['iterations']]).reshape(-1, len(COLS))


def cases():
    """name -> prob; the same constructions tests/test_ref_parity.py uses"""
    out = {}
    for name in sorted(FX.BRANCH_FIXTURES):
        out["branch_" + name] = FX.BRANCH_FIXTURES[name][0]()
    for k, (ex, td) in enumerate([(0, 0), (1, 0), (1, 1)]):
        seq = synth.SyntheticSequence(4 + ex + td, L=60, estimate_extrinsic=ex, estimate_td=td)
        p1 = seq.window(0)
        st, _, pr = R.optimization(p1, 0)
        out["prior_ex%d_td%d" % (ex, td)] = seq.next_window(st, pr, 1)
    for seed in (1, 2, 3):
        out["euroc_window_%d" % seed] = synth.SyntheticSequence(seed, L=150).window(0)
    return out


def main(force=False, path=None):
    """path: where to write (default tests/golden/golden_ceres.npz; tests/test_diff_kit.py passes a scratch file)"""
    L = R.lib()
    if not L.vref_real_ceres() and not force:
        sys.exit("oracle/ref.py loaded a library built on the STAND-IN Ceres: goldens of the restatement are not goldens.\n"
                 "Build `make -C oracle ref_real` where Eigen + Ceres exist and set VINS_REF_LIB to oracle/_ref/libvins_ref_real.so.")
    out = dict(columns=np.array(COLS), real_ceres=np.array(int(L.vref_real_ceres())))
    for name, prob in cases().items():
        st, sm, _ = R.optimization(prob, 1)
        out[name + "/rows"] = rows_of(sm)
        out[name + "/termination"] = np.array(sm['termination'])
        out[name + "/costs"] = np.array([sm['initial_cost'], sm['final_cost']])
        for k in ('pose', 'sb', 'ex', 'inv_depth'):
            out[name + "/" + k] = np.asarray(st[k])
        out[name + "/td"] = np.array(float(st['td']))
        print("%-32s %2d iterations  %-14s  cost %.6e -> %.6e" % (name, len(sm['iterations']), sm['termination'], sm['initial_cost'], sm['final_cost']))
    path = path or os.path.join(HERE, "golden_ceres.npz" if L.vref_real_ceres() else "golden_ceres_STANDIN_DO_NOT_COMMIT.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)
    return path


if __name__ == "__main__":
    main(force="--force-standin" in sys.argv)
