import sys
sys.path.insert(0,'/root/repo')
import __graft_entry__ as g
g.load_package()
from vins_mono_amd import ba, synth
h=ba.Handle()
prob=synth.SyntheticSequence(1,L=60).window(0)
try:
    st,sm,_=h.ba_optimize(prob)
    print('ok',sm['num_iterations'],sm['final_cost'])
except Exception as e:
    print('ERR',e)
