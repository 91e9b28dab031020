"""Phase profile of ba_solve_big_kernel on the enlarged window (library built with -DBA_PROFILE, see gpu_phase_profile.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), "libvinsgpu_prof.so")
from vins_mono_amd import ba, synth
h = ba.Handle()
ex = int(sys.argv[1]) if len(sys.argv) > 1 else 0
seq = synth.SyntheticSequence(5 + ex, n_frames=32, K=31, L=2000, estimate_extrinsic=ex, estimate_td=ex)
prob = synth.SyntheticSequence.anchor_prior(seq.window(0))
names = ["judge", "assemble", "dg", "build", "chain", "schur", "chol", "back", "chain_back", "lm_y", "norms", "cand", "tail"]
for rep in range(2):
    st, sm, _ = h.ba_optimize(prob, ba.VG_MARGIN_NONE)
tot = sum(sm['prof'][:13])
print("iterations", sm['num_iterations'], "total solve-big cycles (thread 0)", tot)
for n, v in zip(names, sm['prof'][:13]):
    print(f"  {n:<12}{v:>12.0f}  {100 * v / max(tot, 1):5.1f} %   per round {v / max(sm['num_iterations'], 1):>9.0f}")
t = [h.ba_run_timed()[0] for _ in range(5)]
print("solve pipeline ms", min(t))
