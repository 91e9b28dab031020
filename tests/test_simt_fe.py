"""The front-end tracking kernels under the CPU fiber emulation (tests/simt, TEST INFRASTRUCTURE): fe_pyrdown_kernel and
fe_lk_kernel compiled from the unchanged sources, run with the fibers of a wavefront scheduled forward, in reverse and
shuffled.  The emulation has no implicit wavefront lock-step, so a missing barrier between lanes that exchange data through
LDS shows up as an order-dependent (and oracle-divergent) result.  Bit-exact against oracle/fe_cpu.cpp, like on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import conftest
from oracle import fe_cpu as F
from vins_mono_amd import fe, synth
h = conftest._simt_handle()
W, H = 256, 160
a = synth.synth_frame(21, W, H)
b = synth.warp_frame(a, 22, shift=(2.6, -1.7), angle_deg=0.6)
tr = fe.FrontEnd(h, W, H, 1, 64)
tr.push_frames([a])
ref_lvl = a
for lvl in range(3):
    assert np.array_equal(tr.get_level(0, lvl), ref_lvl), lvl
    ref_lvl = F.pyrdown(ref_lvl)
tr.push_frames([b])
pts = np.concatenate([F.gftt(a, 24, 0.01, 12.0), np.array([[0.0, 0.0], [255.0, 159.0], [3.2, 150.7], [250.1, 2.5], [-3.0, 80.0]], np.float32)])
nxt, st, err = tr.track(0, pts)
rn, rs, re = F.lk(a, b, pts)
assert np.array_equal(st, rs), (st, rs)
assert np.array_equal(nxt.view(np.uint32), rn.view(np.uint32))
assert np.array_equal(err.view(np.uint32), re.view(np.uint32))
assert st.sum() >= 20
print("OK", int(st.sum()))
"""


@pytest.mark.parametrize("order", ["forward", "reverse", "shuffle"])
def test_emulated_lk_is_bit_exact_in_every_fiber_order(order):
    env = dict(os.environ, SIMT_ORDER=order)
    r = subprocess.run([sys.executable, "-c", _CHILD % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
