"""The C++ host of the landmark-sharded window (vins-mono_amd/host/sharded_estimator.{h,cpp}; BASELINE configs[4], VERDICT r4
"missing" 3) on two ranks: vins_gpu::ShardedWindow -- landmark partition by sum (6 n_l)^2, sub-problem tables, the sharded solve
through the all-reduce hook of its solve handle, the frame-0 all-gather through the caller's transport, the marginalization of the
reduced problem on its second handle -- driven through its C entry points from two gloo ranks on the emulated kernels.  Checked
against vins-mono_amd/shard.py (the Python driver it replaces on the product side: same partition, bit-identical states and prior)
and against the single-rank solve of the whole window."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes as C, hashlib, json, os, sys
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch, torch.distributed as dist
import conftest
from vins_mono_amd import ba, dist_util as D, synth, shard
rank, local, world = D.env_rank()
assert D.init("gloo")
prob = synth.SyntheticSequence(3, L=30).window(0) if sys.argv[1] == "plain" else \
    synth.SyntheticSequence(73, n_frames=6, K=5, L=22, estimate_extrinsic=1, estimate_td=1).window(0)
h = conftest._simt_handle()                      # (loads the emulated library into the package)
SH = C.CDLL(os.path.join(ROOT, "tests", "simt", "_build", "libvins_shard_simt.so"))
GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
def gather_cb(user, src, nbytes, dst):
    mine = torch.frombuffer((C.c_ubyte * nbytes).from_address(src), dtype=torch.uint8).clone()
    parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = torch.cat(parts).numpy()
    C.memmove(dst, out.ctypes.data, out.nbytes)
    return 0
gcb = GATHER(gather_cb)
SH.vins_sharded_create.restype = C.c_void_p
SH.vins_sharded_create.argtypes = [C.c_int, C.c_int, GATHER, C.c_void_p]
SH.vins_sharded_solve_handle.restype = C.c_void_p
SH.vins_sharded_solve_handle.argtypes = [C.c_void_p]
SH.vins_sharded_optimize.argtypes = [C.c_void_p, C.POINTER(ba.Problem), C.c_int, C.POINTER(ba.State), C.POINTER(ba.Summary), C.POINTER(ba.Prior)]
SH.vins_sharded_range.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
SH.vins_sharded_last_error.restype = C.c_char_p
SH.vins_sharded_last_error.argtypes = [C.c_void_p]
SH.vins_sharded_destroy.argtypes = [C.c_void_p]
w = SH.vins_sharded_create(rank, world, gcb, None)
assert w
# the reduction of the solve handle: the same torch hook the Python driver installs (RCCL inside the library on GPUs)
hook = shard.torch_allreduce_hook()
HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
def hook_cb(user, buf, count, stream):
    hook(buf, count, stream)
    return 0
hcb = HOOK(hook_cb)
h.lib.vg_ba_set_allreduce.argtypes = [C.c_void_p, HOOK, C.c_void_p]
assert h.lib.vg_ba_set_allreduce(C.c_void_p(SH.vins_sharded_solve_handle(w)), hcb, None) == 0
pk = ba.PackedProblem(prob)
K, L = prob["pose"].shape[0], len(prob["inv_depth"])
pose, sb, ex, td, lam = np.zeros((K, 7)), np.zeros((K, 9)), np.zeros(7), np.zeros(1), np.zeros(max(L, 1))
_p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
st = ba.State(pose=_p(pose), speedbias=_p(sb), ex_pose=_p(ex), td=_p(td), inv_depth=_p(lam), relo_pose=None)
sm = ba.Summary()
cap, capb = 9 * K + 32, K + 8
J0, r0, x0 = np.zeros((cap, cap)), np.zeros(cap), np.zeros(9 * capb)
kind, idx = np.zeros(capb, np.int32), np.zeros(capb, np.int32)
pr = ba.Prior(cap=cap, cap_blocks=capb, block_kind=kind.ctypes.data_as(C.POINTER(C.c_int)), block_index=idx.ctypes.data_as(C.POINTER(C.c_int)),
              J0=_p(J0), r0=_p(r0), x0=_p(x0))
rc = SH.vins_sharded_optimize(w, C.byref(pk.struct), ba.VG_MARGIN_OLD, C.byref(st), C.byref(sm), C.byref(pr))
assert rc == 0, (rc, SH.vins_sharded_last_error(w))
lo, hi = C.c_int(), C.c_int()
SH.vins_sharded_range(w, C.byref(lo), C.byref(hi))
n = pr.n
J0n = J0.ravel()[:n * n].reshape(n, n)          # (the ABI writes J0 as n x n row-major, whatever the capacity)
# ---- the Python driver on the same ranks (vins-mono_amd/shard.py): partition, sharded solve, sharded marginalization
sub = shard.shard_problem(prob, rank, world)
h.ba_set_large_window(True)
h.ba_set_allreduce(shard.torch_allreduce_hook())
st_py, sm_py, _ = h.ba_optimize(sub)
def gather(obj):
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out
h2 = conftest._simt_handle()
h2.ba_set_large_window(True)
pr_py = shard.marginalize_sharded(h2, sub, st_py, ba.VG_MARGIN_OLD, gather)
# ---- and the whole window on this rank alone
h.ba_set_allreduce(None)
st1, sm1, pr1 = h.ba_optimize(prob, ba.VG_MARGIN_OLD)
sha = lambda *a: hashlib.sha1(b"".join(np.ascontiguousarray(x).tobytes() for x in a)).hexdigest()
A, b = J0n.T @ J0n, J0n.T @ r0[:n]
A1, b1 = pr1["J0"].T @ pr1["J0"], pr1["J0"].T @ pr1["r0"]
out = dict(rank=rank, lo=lo.value, hi=hi.value, py_shard=[int(v) for v in sub["shard"]], status=int(sm.status), iters=int(sm.num_iterations),
           flags=[int(v) for v in sm.it_flags[:sm.num_iterations]], flags_one=[int(v) for v in sm1["it_flags"][:sm1["num_iterations"]]],
           state_sha=sha(pose, sb, ex, td), state_sha_py=sha(st_py["pose"], st_py["sb"], st_py["ex"], np.array([st_py["td"]])),
           lam=[float(v) for v in lam[:hi.value - lo.value]], lam_py=[float(v) for v in st_py["inv_depth"]],
           prior_valid=int(pr.valid), prior_n=int(n), prior_sha=sha(J0n, r0[:n]), prior_sha_py=sha(pr_py["J0"], pr_py["r0"]),
           prior_n_py=int(pr_py["n"]), blocks=[(int(kind[i]), int(idx[i])) for i in range(pr.nblocks)], blocks_py=[(int(a), int(b)) for a, b in pr_py["blocks"]],
           dstate_one=float(max(np.abs(pose - st1["pose"]).max(), np.abs(sb - st1["sb"]).max())),
           dA_one=float(np.abs(A - A1).max() / np.abs(A1).max()), db_one=float(np.abs(b - b1).max() / max(1.0, np.abs(b1).max())))
SH.vins_sharded_destroy(w)
sys.stdout.write(json.dumps(out) + "\n")
sys.stdout.flush()
D.finish()
''' % ROOT


def _run(tmp_path, case, port):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script), case], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    rows, dec, txt, pos = [], json.JSONDecoder(), r.stdout, 0
    while (pos := txt.find('{"rank"', pos)) >= 0:
        obj, pos = dec.raw_decode(txt, pos)
        rows.append(obj)
    assert len(rows) == 2
    rows.sort(key=lambda d: d["rank"])
    return rows


def _check(rows):
    a, b = rows
    assert a["lo"] == 0 and a["hi"] == b["lo"] and a["hi"] > 0 and b["hi"] > b["lo"]                   # disjoint, contiguous, both non-empty
    for r in rows:
        assert [r["lo"], r["hi"]] == r["py_shard"]                      # the same partition as the Python driver
        assert r["status"] == 0 and r["flags"] == r["flags_one"]        # the same trust-region decisions as the single-rank solve
        assert r["state_sha"] == r["state_sha_py"] and r["lam"] == r["lam_py"]          # bit-identical to the Python-driven sharded solve
        assert r["prior_valid"] == 1 and r["prior_n"] == r["prior_n_py"] and r["prior_sha"] == r["prior_sha_py"] and r["blocks"] == r["blocks_py"]
        assert r["dstate_one"] < 1e-6 and r["dA_one"] < 1e-5 and r["db_one"] < 1e-5, r
    assert a["state_sha"] == b["state_sha"] and a["prior_sha"] == b["prior_sha"]       # replicated results: identical on both ranks, no broadcast


def test_cpp_sharded_window_on_two_ranks(tmp_path):
    _check(_run(tmp_path, "plain", 29571))


def test_cpp_sharded_window_with_extrinsic_and_td_columns(tmp_path):
    _check(_run(tmp_path, "extd", 29573))


def test_landmark_partition_balances_the_schur_work():
    """landmark_shards (C++) == shard.landmark_shards (Python) on ragged track lengths, and every share is a contiguous range."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.load_package()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest
    conftest._build_simt()
    from vins_mono_amd import shard
    from oracle.ref import dlopen_own_scope
    SH = dlopen_own_scope(os.path.join(ROOT, "tests", "simt", "_build", "libvins_shard_simt.so"))
    rng = np.random.default_rng(5)
    for L, world in ((1, 2), (7, 3), (200, 8), (2000, 8), (5, 8)):
        nobs = rng.integers(2, 12, L).astype(np.int32)
        out = np.zeros(2 * world, np.int32)
        SH.vins_sharded_landmark_shards(nobs.ctypes.data_as(C.POINTER(C.c_int)), L, world, out.ctypes.data_as(C.POINTER(C.c_int)))
        got = [(int(out[2 * r]), int(out[2 * r + 1])) for r in range(world)]
        assert got == [(int(a), int(b)) for a, b in shard.landmark_shards(nobs, world)], (L, world)
        assert got[0][0] == 0 and got[-1][1] == L and all(got[r][1] == got[r + 1][0] for r in range(world - 1))


def test_sharded_window_refuses_alike_on_every_rank_before_any_collective():
    """ADVICE r5 (medium / low): relocalisation matches that fall into some shards only would give the ranks reduced systems of different
    sizes; VG_PRIOR_RESIDENT would pick up the marginalization handle's own slot; a contiguity error seen by one rank only would leave
    the others in the all-reduce.  ShardedWindow::optimize refuses all three from the FULL problem -- the same verdict on every rank,
    whatever its share -- and shard.py raises for the first two.  No transport is installed: reaching a collective would fail differently."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.load_package()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest
    import pytest
    h = conftest._simt_handle()
    from vins_mono_amd import ba, shard, synth
    from oracle.ref import dlopen_own_scope
    SH = dlopen_own_scope(os.path.join(ROOT, "tests", "simt", "_build", "libvins_shard_simt.so"))
    GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
    SH.vins_sharded_create.restype = C.c_void_p
    SH.vins_sharded_create.argtypes = [C.c_int, C.c_int, GATHER, C.c_void_p]
    SH.vins_sharded_optimize.argtypes = [C.c_void_p, C.POINTER(ba.Problem), C.c_int, C.POINTER(ba.State), C.POINTER(ba.Summary), C.POINTER(ba.Prior)]
    SH.vins_sharded_last_error.restype = C.c_char_p
    SH.vins_sharded_last_error.argtypes = [C.c_void_p]
    SH.vins_sharded_destroy.argtypes = [C.c_void_p]
    prob = synth.SyntheticSequence(3, L=30).window(0)
    L = len(prob["inv_depth"])
    # every match in the LAST landmarks: rank 0 of 2 would see none of them
    relo = dict(prob, relo=dict(pose=np.array(prob["pose"][2], float), match=[(L - 1, 0.01, 0.02), (L - 2, -0.03, 0.01)]))
    assert all(lm >= shard.landmark_shards(prob["lm_nobs"], 2)[0][1] for lm, _, _ in relo["relo"]["match"])
    holes = dict(prob, obs_off=np.asarray(prob["obs_off"]).copy())
    holes["obs_off"][L - 1] += 1                                           # a gap before the last landmark: rank 1's share only
    holes["obs"] = np.concatenate([np.asarray(prob["obs"], float).reshape(-1, 7), np.zeros((1, 7))])
    K = prob["pose"].shape[0]
    _p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for rank in (0, 1):
        w = SH.vins_sharded_create(rank, 2, GATHER(0), None)
        assert w
        for case, want, word in ((relo, -3, b"relocalisation"),
                                 (holes, -1, b"consecutive"), ("resident", -3, b"RESIDENT")):
            pk = ba.PackedProblem(prob if case == "resident" else case)
            if case == "resident":
                pk.struct.prior_n = ba.VG_PRIOR_RESIDENT
            pose, sb, ex, td, lam, rp = np.zeros((K, 7)), np.zeros((K, 9)), np.zeros(7), np.zeros(1), np.zeros(L), np.zeros(7)
            st = ba.State(pose=_p(pose), speedbias=_p(sb), ex_pose=_p(ex), td=_p(td), inv_depth=_p(lam), relo_pose=_p(rp))
            sm = ba.Summary()
            rc = SH.vins_sharded_optimize(w, C.byref(pk.struct), ba.VG_MARGIN_NONE, C.byref(st), C.byref(sm), None)
            assert rc == want and word in SH.vins_sharded_last_error(w), (rank, rc, SH.vins_sharded_last_error(w))
        SH.vins_sharded_destroy(w)
        with pytest.raises(ValueError, match="relocalisation"):
            shard.shard_problem(relo, rank, 2)
        with pytest.raises(ValueError, match="consecutive"):
            shard.shard_problem(holes, rank, 2)
    shard.shard_problem(relo, 0, 1)                                        # one rank: relocalisation factors are fine
    h.close()
