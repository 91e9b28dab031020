#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz.

PARITY UNPINNED: the reference has no golden vectors and cannot be built here, so these files hold
the outputs of THIS repo's CPU oracles (oracle/ba_numpy.py, oracle/fe_cpu.cpp) on small seeded inputs.  They are
regression anchors — they pin the oracles (and through the parity tests the HIP path) against silent drift, e.g. a
NumPy/LAPACK or compiler change — not evidence of agreement with Ceres / OpenCV.

usage: python tests/golden/make_golden.py        (run from the repository root; CPU only)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from vins_mono_amd import synth  # noqa: E402
from oracle import ba_numpy as B, fe_cpu as F  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def ba_case():
    seq = synth.SyntheticSequence(1234, L=24)
    prob = seq.window(0)
    st, summ, pr = B.optimization(prob, B.MARGIN_OLD)
    prob2 = seq.next_window(st, pr, 1)
    st2, summ2, pr2 = B.optimization(prob2, B.MARGIN_OLD)
    return dict(seed=1234, L=24,
                w1_pose=st['pose'], w1_sb=st['sb'], w1_inv_depth=st['inv_depth'], w1_final_cost=summ['final_cost'],
                w1_iters=summ['num_iterations'], w1_H=pr['J0'].T @ pr['J0'], w1_g=pr['J0'].T @ pr['r0'],
                w2_pose=st2['pose'], w2_sb=st2['sb'], w2_final_cost=summ2['final_cost'], w2_iters=summ2['num_iterations'])


def fe_case():
    a = synth.synth_frame(77, 256, 160)
    b = synth.warp_frame(a, 78)
    corners = F.gftt(a, 40, 0.01, 12.0)
    nxt, st, err = F.lk(a, b, corners)
    return dict(seed=77, size=np.array([256, 160]), frame_a_crc=np.array([int(a.astype(np.uint64).sum()), int((a.astype(np.uint64) ** 2).sum())]),
                corners=corners, lk_next=nxt, lk_status=st, lk_err=err, pyr1=F.pyrdown(a), clahe=F.clahe(a))


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "golden_ba.npz"), **ba_case())
    np.savez_compressed(os.path.join(HERE, "golden_fe.npz"), **fe_case())
    print("wrote", os.listdir(HERE))
