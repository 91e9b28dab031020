"""manual GPU debugging aid (not a test)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
g.load_package()
import numpy as np
from vins_mono_amd import ba, synth
from oracle import ba_numpy as B
h = ba.Handle()
seq = synth.SyntheticSequence(1, L=int(os.environ.get('NL', '60')))
prob = seq.window(0)
out = h.ba_eval_factors(prob)
import tests.test_ba_gpu as T
pr, pJ, ir, iJ, qr = T._oracle_factor_tables(prob)
print('proj r err', np.abs(out['proj_r']-pr).max(), 'J err', np.abs(out['proj_J'][:,:,:19]-pJ[:,:,:19]).max(), np.abs(pJ).max())
print('imu r rel', np.abs(out['imu_r']-ir).max()/np.abs(ir).max(), 'imu J rel', np.abs(out['imu_J']-iJ).max()/np.abs(iJ).max())
t=time.time(); st, sm, _ = h.ba_optimize(prob); print('gpu optimize wall', time.time()-t)
x, summ = B.solve(prob)
ref = B.double2vector(prob, x)
print('gpu', sm['initial_cost'], sm['final_cost'], sm['num_iterations'], sm['it_flags'], sm['status'], sm['termination'])
print('ora', summ['initial_cost'], summ['final_cost'], summ['num_iterations'])
for k, it in enumerate(summ['iterations']):
    print(k, 'cand', sm['it_cost_cand'][k], it.get('cost_cand'), 'model', sm['it_model'][k], it.get('model_change'), 'rad', sm['it_radius'][k], it.get('radius'), 'sn', sm['it_step_norm'][k], it.get('step_norm'))
print('pose err', np.abs(st['pose']-ref['pose']).max(), 'sb err', np.abs(st['sb']-ref['sb']).max(), 'lam err', np.abs(st['inv_depth']-ref['inv_depth']).max())
for n in (1, 16, 256):
    probs = [prob]*n
    h.ba_upload(probs)
    h.ba_run_async(); h.sync()
    h.timer_start(); h.ba_run_async(); ms = h.timer_stop()
    print(f'batch {n}: {ms:.3f} ms -> {n/ms*1e3:.0f} solves/s')
names = ['PRO','IMU','PRIOR','PROJ','ACC','LMACC','IMUACC','PRACC','JVEC','BUILD','SCHUR','CHOL','BACK','CAND','MISC','TOTAL']
seq = synth.SyntheticSequence(3, L=150)
p1 = seq.window(0)
st1, _, pr1 = h.ba_optimize(p1, ba.VG_MARGIN_OLD)
p2 = seq.next_window(st1, pr1, 1)
st, sm, _ = h.ba_optimize(p2, ba.VG_MARGIN_OLD)
tot = sm['prof'][15]
if tot > 0:
    print('phase cycles (single window, with prior), total %.0f cycles = %.3f ms @2.4GHz' % (tot, tot/2.4e6))
    for n, v in zip(names, sm['prof']): print(f'  {n:8s} {v:12.0f} {100*v/tot:6.1f}%')
h.ba_upload([p2], [ba.VG_MARGIN_OLD]); print('timed (solve_ms, marg_ms):', [h.ba_run_timed() for _ in range(3)])
