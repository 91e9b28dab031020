import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
g.load_package()
import numpy as np
np.set_printoptions(linewidth=200, precision=4)
from vins_mono_amd import ba, synth
from oracle import ba_numpy as B
h = ba.Handle()
seq = synth.SyntheticSequence(40, L=60)
prob = seq.window(0)
x, _ = B.solve(prob)
at = dict(prob); at.update(pose=x['pose'], sb=x['sb'], ex=x['ex'], td=x['td'], inv_depth=x['inv_depth'], max_iters=0)
st_o, _, po = B.optimization(at, B.MARGIN_OLD)
st_g, sm, pg = h.ba_optimize(at, ba.VG_MARGIN_OLD)
print('state diff', np.abs(st_g['pose']-st_o['pose']).max(), np.abs(st_g['sb']-st_o['sb']).max(), np.abs(st_g['inv_depth']-st_o['inv_depth']).max())
Hg, Ho = pg['J0'].T@pg['J0'], po['J0'].T@po['J0']
gg, go = pg['J0'].T@pg['r0'], po['J0'].T@po['r0']
print('H rel', np.abs(Hg-Ho).max()/np.abs(Ho).max(), 'H vs A (oracle)', np.abs(Ho-po['A']).max()/np.abs(Ho).max(), 'Hg vs A', np.abs(Hg-po['A']).max()/np.abs(Ho).max())
print('g gpu ', gg[:12]); print('g ora ', go[:12]); print('b ora ', po['b'][:12])
print('|dg|', np.abs(gg-go).max(), 'g vs b (oracle)', np.abs(go-po['b']).max(), 'gg vs b', np.abs(gg-po['b']).max())
w = np.linalg.eigvalsh(po['A']); print('eig A min/max', w[:6], w[-3:])
print('r0 norms', np.linalg.norm(pg['r0']), np.linalg.norm(po['r0']))
