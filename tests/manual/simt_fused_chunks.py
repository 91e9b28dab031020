"""MANUAL: the fused projection kernel at chunk boundaries (emulated): windows whose factor count sits just above a multiple of the chunk size."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, numpy as np
from vins_mono_amd import ba, synth
from oracle import ba_numpy as B
h = conftest._simt_handle()
for L in (226, 227, 228, 229, 230):
    p = synth.SyntheticSequence(3, L=L).window(0)
    F = int(sum(p['lm_nobs']) - len(p['lm_nobs']))
    st, sm, _ = h.ba_optimize(p)
    x, smo = B.solve(p)
    ref = B.double2vector(p, x)
    print(L, F, sm['num_iterations'], np.abs(st['pose'] - ref['pose']).max())
