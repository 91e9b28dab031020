"""vg_create_config (ABI 10, VERDICT r4 'structure' item 12): a handle made from a vg_config never consults the environment -- the
development variables VG_BA_LAUNCH_MODE / VG_BA_FUSED / VG_BA_FUSED_MIN / VG_BA_SOLVE_W8_BELOW / VG_PACK_THREADS only shape handles of plain vg_create().
Run in a subprocess (the library reads those variables once per process)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes, json, os, sys
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
pkg = g.load_package()
if EMULATED:
    import conftest
    conftest._build_simt()
    pkg._lib, pkg.LIB_PATH = ctypes.CDLL(conftest.SIMT_LIB, mode=ctypes.RTLD_LOCAL), conftest.SIMT_LIB
from vins_mono_amd import ba, synth
probs = [ba.PackedProblem(synth.SyntheticSequence(20 + i, L=12).window(0)) for i in range(32)]
out = {}
for name, h in (("plain", ba.Handle()), ("config", ba.Handle(config=dict(device=None, launch_mode="direct")))):
    h.ba_upload(probs, [ba.VG_MARGIN_NONE] * 32)
    h.ba_run_async()
    st, sm, _ = h.ba_download()
    out[name] = dict(fused=int(h.lib.vg_ba_batch_is_fused(h.h)), mode=h.ba_launch_stats()["mode"], cost=[s["final_cost"] for s in sm])
    h.close()
print("RESULT " + json.dumps(out))
'''


def _run(emulated):
    env = dict(os.environ, VG_BA_FUSED="0", VG_BA_LAUNCH_MODE="graph")
    code = f"ROOT = {ROOT!r}\nEMULATED = {emulated!r}\n" + SCRIPT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    # plain vg_create: the development variables apply (spread kernels, graph launches) ...
    assert d["plain"]["fused"] == 0 and d["plain"]["mode"] in ("graph", 1)
    # ... a handle from a vg_config ignores them (32 windows: fused factor kernel; direct launches as the config says)
    assert d["config"]["fused"] == 1 and d["config"]["mode"] in ("direct", 0)
    # and both solve the same windows to the same costs (fused vs spread kernels: summation order differs in the last digits)
    for a, b in zip(d["plain"]["cost"], d["config"]["cost"]):
        assert abs(a - b) <= 1e-9 * max(1.0, abs(b))


def test_config_handle_ignores_the_environment_on_emulated_kernels():
    _run(True)


@pytest.mark.gpu
def test_config_handle_ignores_the_environment():
    _run(False)
