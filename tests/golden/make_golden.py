#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz.

golden_ba.npz holds outputs of THE REFERENCE: oracle/_ref/libvins_ref.so = the reference's own estimator / factor /
marginalization sources compiled unchanged (oracle/Makefile `ref`, needs /root/reference, i.e. this container) — two chained
Estimator::optimization() calls with MARGIN_OLD and the factor tables of the second window.  The minimiser inside is the
restated Ceres of oracle/ref_stubs/ceres (third party, absent).  It travels to the GPU box as a fixture: there the reference
library may be missing, the vectors are not.

golden_fe.npz: PARITY UNPINNED — OpenCV is not in /root/reference and not in this image, so the FE file holds the outputs of
this repo's restatement (oracle/fe_cpu.cpp): a regression anchor, not evidence of agreement with OpenCV.

usage: python tests/golden/make_golden.py        (run from the repository root; CPU only)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from vins_mono_amd import synth  # noqa: E402
from oracle import ba_numpy as B, fe_cpu as F, ref as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def ba_case():
    seq = synth.SyntheticSequence(1234, L=24)
    prob = seq.window(0)
    st, summ, pr = R.optimization(prob, 0)
    _, H1, g1, _ = R.canonical_prior(pr)
    prob2 = seq.next_window(st, pr, 1)
    proj_r, proj_J, imu_r, imu_J = R.factor_tables(prob2)
    st2, summ2, pr2 = R.optimization(prob2, 0)
    return dict(seed=1234, L=24, source="oracle/_ref (reference sources compiled unchanged)",
                w1_pose=st['pose'], w1_sb=st['sb'], w1_inv_depth=st['inv_depth'], w1_final_cost=summ['final_cost'],
                w1_iters=summ['num_iterations'], w1_H=H1, w1_g=g1,
                w1_prior_n=pr['n'], w1_prior_blocks=np.array(pr['blocks'], np.int32), w1_prior_J0=pr['J0'], w1_prior_r0=pr['r0'],
                w1_prior_x0=np.concatenate(pr['x0']),
                w2_proj_r=proj_r, w2_proj_J=proj_J, w2_imu_r=imu_r, w2_imu_J=imu_J,
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
