"""Development aid: run ONE window through a -DBA_DEBUG_DUMP build of the library and save the LDS image of the solve kernel after
each phase of its first iteration (vg_debug_dump).  Run it once against the emulated build and once on the GPU, then diff:
    python tests/manual/dbg_dump.py tests/simt/_build_dbg/libvinsgpu_simt.so /tmp/emu.npz
    gpurun -- 'python tests/manual/dbg_dump.py vins-mono_amd/lib/libvinsgpu_dbg.so gpurun_out/gpu.npz'
    python tests/manual/dbg_dump.py --diff /tmp/emu.npz gpurun_out/gpu.npz"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SLOT = 16384


def run(libpath, out):
    import __graft_entry__ as g
    pkg = g.load_package()
    pkg._lib, pkg.LIB_PATH = ctypes.CDLL(os.path.abspath(libpath), mode=ctypes.RTLD_GLOBAL), os.path.abspath(libpath)
    from vins_mono_amd import ba, synth
    seq = synth.SyntheticSequence(7, L=40)
    prob = seq.window(0)
    h = ba.Handle()
    st, sm, _ = h.ba_optimize(prob)
    buf = np.zeros(6 * SLOT)
    rc = pkg._lib.vg_debug_dump(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(buf.size))
    assert rc == 0
    np.savez(out, dump=buf, iters=sm['num_iterations'], flags=np.array(sm['it_flags']), cost=sm['final_cost'])
    print("iterations", sm['num_iterations'], "flags", list(sm['it_flags']), "final cost", sm['final_cost'])


def diff(a, b):
    A, B = np.load(a), np.load(b)
    print("iterations", A['iters'], B['iters'])
    da, db = A['dump'], B['dump']
    names = ["after assemble", "after build_scaled", "after chain_schur", "after cholesky", "-", "xp (HBM)"]
    for s in range(6):
        x, y = da[s * SLOT:(s + 1) * SLOT], db[s * SLOT:(s + 1) * SLOT]
        bad = np.where(~np.isclose(x, y, rtol=1e-9, atol=1e-12, equal_nan=True))[0]
        print(f"slot {s} {names[s]}: {bad.size} differing entries", bad[:12], "nan in b:", int(np.isnan(y).sum()))
        for i in bad[:6]:
            print("    ", i, x[i], y[i])


if __name__ == "__main__":
    if sys.argv[1] == "--diff":
        diff(sys.argv[2], sys.argv[3])
    else:
        run(sys.argv[1], sys.argv[2])
