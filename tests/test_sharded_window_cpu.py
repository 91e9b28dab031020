"""Landmark shards of one window over two ranks (BASELINE configs[4]) on CPU: world_size-2 gloo rendezvous
on 127.0.0.1, every rank runs the kernel sources of the large-window path under the fiber emulator (tests/simt) on ITS
share of the landmarks, the two reduce buffers travel through the all-reduce hook of the C-ABI (vg_ba_set_allreduce).
Checked: both ranks end with bit-identical frame states; the sharded solve equals the single-rank solve of the whole
window (first step to 1e-12, per-iteration costs and states to 1e-7) and the NumPy oracle (trace + states, as every other BA parity test)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import conftest
from vins_mono_amd import dist_util as D, synth, shard
import ba_fixtures as FX
rank, local, world = D.env_rank()
assert D.init("gloo")
case = sys.argv[1]
if case == "plain":
    prob = synth.SyntheticSequence(3, L=30).window(0)
elif case == "extd":
    prob = synth.SyntheticSequence(73, n_frames=6, K=5, L=22, estimate_extrinsic=1, estimate_td=1).window(0)
else:
    prob = FX.BRANCH_FIXTURES[case][0]()
h = conftest._simt_handle()
h.ba_set_large_window(True)
# the whole window on this rank alone (no hook), then this rank's share with the hook
st1, sm1, _ = h.ba_optimize(prob)
sub = shard.shard_problem(prob, rank, world)
h.ba_set_allreduce(shard.torch_allreduce_hook())
st, sm, _ = h.ba_optimize(sub)
counts = h.ba_reduce_layout()
D.barrier()
# marginalization of the sharded window (shard.marginalize_sharded) against the single-rank solve + marginalization of the whole
# window, and against the whole window marginalized at the SHARDED state (max_iters = 0)
import torch.distributed as dist
fl = lambda a: [float(v) for v in np.asarray(a).ravel()]
def gather(obj):
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out
marg = None
if case in ("plain", "extd"):
    from vins_mono_amd import ba
    h2 = conftest._simt_handle()                       # the marginalization runs on a handle of its own (no hook)
    h2.ba_set_large_window(True)
    pr_sh = shard.marginalize_sharded(h2, sub, st, ba.VG_MARGIN_OLD, gather)
    h.ba_set_allreduce(None)
    _, _, pr_one = h.ba_optimize(prob, ba.VG_MARGIN_OLD)
    at = dict(prob)
    lam_all = gather([float(v) for v in st["inv_depth"]])
    at.update(pose=st["pose"], sb=st["sb"], ex=st["ex"], td=st["td"], inv_depth=np.array([v for p in lam_all for v in p]), max_iters=0)
    _, _, pr_at = h.ba_optimize(at, ba.VG_MARGIN_OLD)
    import hashlib
    def prod(p):
        return p["J0"].T @ p["J0"], p["J0"].T @ p["r0"]
    (A, b), (A1, b1), (Aat, bat) = prod(pr_sh), prod(pr_one), prod(pr_at)
    scale = float(np.abs(Aat).max())
    marg = dict(n=[int(pr_sh["n"]), int(pr_one["n"]), int(pr_at["n"])], blocks_equal=bool(pr_sh["blocks"] == pr_one["blocks"] == pr_at["blocks"]),
                sha=hashlib.sha1(np.ascontiguousarray(pr_sh["J0"]).tobytes() + np.ascontiguousarray(pr_sh["r0"]).tobytes()).hexdigest(),
                dA_at=float(np.abs(A - Aat).max()) / scale, db_at=float(np.abs(b - bat).max()) / max(1.0, float(np.abs(bat).max())),
                dA_one=float(np.abs(A - A1).max()) / scale, db_one=float(np.abs(b - b1).max()) / max(1.0, float(np.abs(b1).max())))
lo, hi = (int(v) for v in sub["shard"])
def pack(s, m):
    return dict(pose=fl(s["pose"]), sb=fl(s["sb"]), ex=fl(s["ex"]), td=float(s["td"]), lam=fl(s["inv_depth"]),
                it_cost=fl(m["it_cost"]), it_cost_cand=fl(m["it_cost_cand"]), it_flags=[int(v) for v in m["it_flags"]], it_radius=fl(m["it_radius"]),
                n=int(m["num_iterations"]), term=int(m["termination"]), status=int(m["status"]), final_cost=float(m["final_cost"]))
sys.stdout.write(json.dumps(dict(rank=rank, lo=lo, hi=hi, counts=[int(v) for v in counts], sharded=pack(st, sm), single=pack(st1, sm1), marg=marg)) + "\n")
sys.stdout.flush()
D.finish()
''' % ROOT


def _run(tmp_path, case, port):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script), case], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rows, dec, txt, pos = [], json.JSONDecoder(), r.stdout, 0
    while (pos := txt.find('{"rank"', pos)) >= 0:
        obj, pos = dec.raw_decode(txt, pos)
        rows.append(obj)
    assert len(rows) == 2
    rows.sort(key=lambda d: d["rank"])
    return rows


def _check(rows, first=1e-12, rest=1e-7, state=1e-7):
    a, b = rows[0]["sharded"], rows[1]["sharded"]
    one = rows[0]["single"]
    assert rows[0]["counts"] == rows[1]["counts"]                     # rank-invariant reduce-buffer sizes
    assert rows[0]["hi"] == rows[1]["lo"] and rows[0]["lo"] == 0       # disjoint, contiguous landmark shares
    for k in ("pose", "sb", "ex", "td", "it_cost", "it_cost_cand", "it_flags", "it_radius", "n", "term", "status", "final_cost"):
        assert a[k] == b[k], k                                        # replicated part: bit-identical on both ranks
    assert a["status"] == 0 and a["n"] == one["n"] and a["it_flags"] == one["it_flags"] and a["term"] == one["term"]
    n = a["n"]
    # the first step only depends on the rank-summed reduced system of the initial point: 1e-12; later iterations carry the
    # rounding differences of the different summation order through the nonlinear iteration
    np.testing.assert_allclose(a["it_cost"][:1], one["it_cost"][:1], rtol=1e-13)
    np.testing.assert_allclose(a["it_cost_cand"][:1], one["it_cost_cand"][:1], rtol=first, atol=1e-300)
    np.testing.assert_allclose(a["it_cost"][:n], one["it_cost"][:n], rtol=rest)
    np.testing.assert_allclose(a["it_cost_cand"][:n], one["it_cost_cand"][:n], rtol=rest, atol=1e-300)
    for k in ("pose", "sb", "ex"):
        np.testing.assert_allclose(np.array(a[k]), np.array(one[k]), rtol=state, atol=1e-2 * state)
    lam = np.array(a["lam"] + b["lam"])
    np.testing.assert_allclose(lam, np.array(one["lam"]), rtol=10 * state, atol=1e-2 * state)
    return a, lam


def _check_marg(rows, tol_single=1e-5):
    """Both ranks hold the identical new prior; it equals the whole window marginalized at the same (sharded) state to rounding
    and the single-rank solve + marginalization to what the states differ by (J0^T J0 relative to its largest entry, J0^T r0)."""
    m0, m1 = rows[0]["marg"], rows[1]["marg"]
    assert m0["sha"] == m1["sha"]                                      # bit-identical on both ranks, no broadcast
    assert m0["n"][0] == m0["n"][1] == m0["n"][2] and m0["blocks_equal"]
    assert m0["dA_at"] <= 1e-9 and m0["db_at"] <= 1e-9, m0
    assert m0["dA_one"] <= tol_single and m0["db_one"] <= tol_single, m0


def test_two_rank_landmark_shards_equal_the_single_rank_solve(tmp_path):
    rows = _run(tmp_path, "plain", 29541)
    a, lam = _check(rows)
    _check_marg(rows)
    # and the oracle: same bounds as tests/test_ba_gpu.py::_check_solve
    from oracle import ba_numpy as B
    from vins_mono_amd import synth
    prob = synth.SyntheticSequence(3, L=30).window(0)
    x, summ = B.solve(prob)
    ref = B.double2vector(prob, x)
    assert a["n"] == summ["num_iterations"]
    np.testing.assert_allclose(a["final_cost"], summ["final_cost"], rtol=1e-6)
    assert np.abs(np.array(a["pose"]).reshape(-1, 7) - ref["pose"]).max() < 1e-4 * max(1.0, np.abs(ref["pose"][:, :3]).max())
    assert np.abs(np.array(a["sb"]).reshape(-1, 9) - ref["sb"]).max() < 1e-4 * max(1.0, np.abs(ref["sb"]).max())
    np.testing.assert_allclose(lam, ref["inv_depth"], rtol=1e-4, atol=1e-6)


def test_two_rank_shards_with_rejected_steps_extrinsic_and_td(tmp_path):
    # (the low-parallax window is ill-conditioned by construction: tests/ba_fixtures.py COST_RTOL)
    _check(_run(tmp_path, "low_parallax", 29542), first=1e-9, rest=1e-5, state=1e-5)
    rows = _run(tmp_path, "extd", 29543)
    _check(rows)
    _check_marg(rows)
