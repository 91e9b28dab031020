"""Phase profile of ba_solve_kernel (library built with -DBA_PROFILE: make -C vins-mono_amd/csrc OBJDIR=../build_prof
LIB=../lib/libvinsgpu_prof.so EXTRA=-DBA_PROFILE).  Prints shader cycles per phase, summed over the rounds of a solve."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), "libvinsgpu_prof.so")
from vins_mono_amd import ba, synth
from oracle import ba_numpy as B
h = ba.Handle()
seq = synth.SyntheticSequence(5, L=150)
p1 = seq.window(0)
st, sm, pr = h.ba_optimize(p1, 0)
prob = seq.next_window(st, pr, 1)
names = ["judge", "assemble", "dg", "build", "chain", "schur", "chol", "back", "chain_back", "lm_y", "norms", "cand", "tail"]
for rep in range(2):
    st2, sm2, _ = h.ba_optimize(prob, 0)
tot = sum(sm2['prof'][:13])
print("iterations", sm2['num_iterations'], "total solve-kernel cycles (thread 0)", tot)
for n, v in zip(names, sm2['prof'][:13]):
    print(f"  {n:<12}{v:>12.0f}  {100 * v / tot:5.1f} %   per round {v / max(sm2['num_iterations'], 1):>9.0f}")
