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


_CHILD_F = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import conftest
from oracle import fe_cpu as F
from vins_mono_amd import fe
from test_fe_oracle import _two_view
h = conftest._simt_handle()
tr = fe.FrontEnd(h, 64, 64, 1, 8)
for seed, n, n_out in ((3, 60, 8), (5, 40, 0), (8, 12, 2), (9, 9, 0)):       # n >= 15: RANSAC; n < 15: LMedS
    p1, p2, out = _two_view(seed, n=n, n_out=n_out)
    st_g, F_g = tr.reject_with_f(p1, p2, 1.0)
    st_o, F_o = F.reject_with_f(p1, p2, 1.0)
    assert np.array_equal(st_g, st_o), (seed, st_g, st_o)
    s = np.abs(F_o).max()
    assert np.abs(F_g - F_o).max() < 1e-6 * s, (seed, F_g, F_o)
print("OK")
"""


def test_emulated_reject_with_f_matches_restatement():
    """vg_fe_reject_with_f (host schedule + fe_ransac7_kernel + host bookkeeping) against the sequential restatement of
    OpenCV's loop: same inlier mask, same model — RANSAC for n >= 15, LMedS below."""
    r = subprocess.run([sys.executable, "-c", _CHILD_F % dict(root=ROOT)], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
