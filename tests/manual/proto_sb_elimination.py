"""CPU prototype for a round-2 restructuring of the BA solve ("structural"): eliminate the
speed-bias blocks of the reduced camera system by a block-Thomas (block-tridiagonal Schur) sweep BEFORE the dense
Cholesky, so that the dense factorisation shrinks from R = 165 to the 66 (+6 +1) pose / extrinsic / td columns.

Checks on a real window (with prior, after the landmark Schur complement, Jacobi-scaled + damped like the solver):
  * the speed-bias block of S is block-tridiagonal (bandwidth = one 9x9 block),
  * block-Thomas elimination + 73-dim Cholesky reproduces the dense 165-dim solve,
  * flop count of both routes.
Run: python tests/manual/proto_sb_elimination.py   (CPU only)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g  # noqa: E402

g.load_package()
from vins_mono_amd import synth  # noqa: E402
from oracle import ba_numpy as B  # noqa: E402

seq = synth.SyntheticSequence(5, L=150)
p1 = seq.window(0)
st, _, pr = B.optimization(p1, B.MARGIN_OLD)
prob = seq.next_window(st, pr, 1)
lay = B.Layout(prob)
x = B.state_of(prob)
_, r, J = B.evaluate(prob, x, need_jac=True)
R, K = lay.R, lay.K
Jp, Jl = J[:, :R], J[:, R:]
# Jacobi scaling 1/(1 + ||J_col||) and the mu = 1e-8 damping of the dogleg's Gauss-Newton system (ASSUMPTIONS C3, C4)
sc = 1.0 / (1.0 + np.sqrt((J * J).sum(axis=0)))
J = J * sc
Jp, Jl = J[:, :R], J[:, R:]
dg = np.clip((J * J).sum(axis=0), 1e-6, 1e32)
H = Jp.T @ Jp + 1e-8 * np.diag(dg[:R])
hll = np.einsum('ij,ij->j', Jl, Jl) + 1e-8 * dg[R:]
W = Jp.T @ Jl
S = H - (W / hll) @ W.T
gvec = Jp.T @ r - W @ ((Jl.T @ r) / hll)
sb = np.concatenate([np.arange(o, o + 9) for o in lay.sb_off])
cam = np.array([c for c in range(R) if c not in set(sb)])
Sss, Ssc, Scc = S[np.ix_(sb, sb)], S[np.ix_(sb, cam)], S[np.ix_(cam, cam)]
# block structure of the speed-bias part
nb = K
blk = np.array([[np.abs(Sss[9 * a:9 * a + 9, 9 * b:9 * b + 9]).max() for b in range(nb)] for a in range(nb)])
band = max(abs(a - b) for a in range(nb) for b in range(nb) if blk[a, b] > 1e-9 * blk.max())
print("speed-bias block bandwidth (in 9x9 blocks):", band, " (prior couples sb_0 only:", bool(blk[0, 2:].max() < 1e-9 * blk.max()), ")")
# reference: dense Cholesky of the full reduced system
y_ref = np.linalg.solve(S, gvec)
# block-Thomas elimination of the chain sb_0 .. sb_{K-1} (forward sweep), accumulating the Schur complement on cam
D = [Sss[9 * a:9 * a + 9, 9 * a:9 * a + 9].copy() for a in range(nb)]
U = [Sss[9 * a:9 * a + 9, 9 * a + 9:9 * a + 18].copy() for a in range(nb - 1)]
C = [Ssc[9 * a:9 * a + 9, :].copy() for a in range(nb)]
gs = [gvec[sb][9 * a:9 * a + 9].copy() for a in range(nb)]
Sc, gc = Scc.copy(), gvec[cam].copy()
flops = 0
for a in range(nb):
    Di = np.linalg.inv(D[a])
    flops += 2 * 9 ** 3
    X = Di @ C[a]
    z = Di @ gs[a]
    flops += 2 * 81 * len(cam)
    Sc -= C[a].T @ X
    gc -= C[a].T @ z
    flops += 2 * 9 * len(cam) ** 2
    if a + 1 < nb:
        Ut = U[a].T                       # coupling of sb_{a+1} with sb_a
        D[a + 1] -= Ut @ Di @ U[a]
        C[a + 1] -= Ut @ X
        gs[a + 1] -= Ut @ z
        flops += 2 * (2 * 9 ** 3 + 81 * len(cam))
yc = np.linalg.solve(Sc, gc)
flops += len(cam) ** 3 / 3
y = np.zeros(R)
y[cam] = yc
# (the chain back-substitution needs the eliminated forms; solve the tridiagonal system directly for the check)
ysb = np.linalg.solve(Sss, gvec[sb] - Ssc @ yc)
y[sb] = ysb
print("residuals |S y - g| / |g|: dense %.1e, block-Thomas %.1e ; cond(S) = %.1e" % (np.linalg.norm(S @ y_ref - gvec) / np.linalg.norm(gvec), np.linalg.norm(S @ y - gvec) / np.linalg.norm(gvec), np.linalg.cond(S)))
print("dense 165-dim solve vs block-Thomas + %d-dim Cholesky: max rel diff %.2e" % (len(cam), np.abs(y - y_ref).max() / np.abs(y_ref).max()))
print("flops: dense Cholesky R^3/3 = %.2f M ; block-Thomas + reduced Cholesky = %.2f M" % (R ** 3 / 3 / 1e6, flops / 1e6))
