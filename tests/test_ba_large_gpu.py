"""Large-window path on the GPU (BASELINE configs[4]): forced on the ordinary fixtures it must agree
with the NumPy oracle exactly like the single-workgroup pipeline does (tests/test_ba_gpu.py::_check_solve: iteration
trace, flags, termination, gauge-fixed states); at the enlarged size (31 frames x 2000 landmarks, where only the
C++ restatement oracle/ba_cpu.cpp is fast enough) states within 1e-4 and the same accept / reject trace."""
import numpy as np
import pytest

from oracle import ba_cpu, ba_numpy as B
from vins_mono_amd import ba, synth

import ba_fixtures as FX
from test_ba_gpu import _check_solve

pytestmark = pytest.mark.gpu


@pytest.fixture()
def large(handle):
    handle.ba_set_large_window(True)
    yield handle
    handle.ba_set_large_window(False)


@pytest.mark.parametrize("seed", [1, 4, 9])
def test_forced_large_path_matches_oracle(large, seed):
    _check_solve(large, synth.SyntheticSequence(seed).window(0))


@pytest.mark.parametrize("name", sorted(FX.BRANCH_FIXTURES))
def test_forced_large_path_trust_region_branches(large, name):
    build, need = FX.BRANCH_FIXTURES[name]
    _, _, summ = _check_solve(large, build(), rtol_cost=FX.COST_RTOL.get(name, 1e-6))
    assert need <= FX.trace_features(summ)


def test_forced_large_path_with_prior_relocalisation_extrinsic_td(large, handle):
    seq = synth.SyntheticSequence(31, estimate_extrinsic=1, estimate_td=1)
    p0 = seq.window(0)
    handle.ba_set_large_window(False)
    st, sm, pr = handle.ba_optimize(p0, ba.VG_MARGIN_OLD)          # a real marginalization prior from the ordinary path
    handle.ba_set_large_window(True)
    _check_solve(large, seq.next_window(st, pr, 1))
    _check_solve(large, synth.SyntheticSequence(23, n_frames=13, K=12, L=60).window(0))


def test_forced_large_path_marginalizes_like_the_single_workgroup_path(large, handle):
    """MARGIN_OLD and MARGIN_SECOND_NEW on the large-window path (forced on a window that also fits the other path): the same
    kernel, fed by the other solve pipeline, must produce the same prior as the oracle."""
    from test_ba_gpu import _check_prior, _window_with_prior
    seq = synth.SyntheticSequence(40, L=60)
    prob = seq.window(0)
    x, _ = B.solve(prob)
    at = dict(prob)
    at.update(pose=x['pose'], sb=x['sb'], ex=x['ex'], td=x['td'], inv_depth=x['inv_depth'], max_iters=0)
    _, _, pr_o = B.optimization(at, B.MARGIN_OLD)
    _, sm, pr_g = large.ba_optimize(at, ba.VG_MARGIN_OLD)
    assert sm['status'] == 0
    _check_prior(pr_g, pr_o)
    _, _, prob2 = _window_with_prior(6, L=150)
    prob2 = dict(prob2, max_iters=0)
    _, _, pr2_o = B.optimization(prob2, B.MARGIN_SECOND_NEW)
    _, _, pr2_g = large.ba_optimize(prob2, ba.VG_MARGIN_SECOND_NEW)
    _check_prior(pr2_g, pr2_o)


def test_enlarged_window_marginalization(handle):
    """A 31-frame window (large-window path by itself): kept block n = 6 * 30 + 9 + 7 = 196 columns (pivoted Cholesky in global
    memory), camera part of the projection assembly in five entry passes, dropped block [pose 0 | speed-bias 0 | the landmarks
    anchored at frame 0] by block elimination."""
    from test_ba_gpu import marginalize_many_frame0_landmarks
    marginalize_many_frame0_landmarks(handle, K=31, L=400, w0=3, n_frames=40, min_m=60)


def _vs_cpp_oracle(h, prob, rtol_state=1e-4):
    st_o, sm_o, _ = ba_cpu.optimize(prob, margin_flag=ba.VG_MARGIN_NONE)
    st, sm, _ = h.ba_optimize(prob)
    assert sm['status'] == 0 and sm_o['status'] == 0
    n = sm_o['num_iterations']
    assert sm['num_iterations'] == n and sm['termination'] == sm_o['termination']
    assert list(sm['it_flags'][:n]) == list(sm_o['it_flags'][:n])
    assert np.isclose(sm['initial_cost'], sm_o['initial_cost'], rtol=1e-9)
    np.testing.assert_allclose(sm['it_cost'][:n], sm_o['it_cost'][:n], rtol=1e-6)
    np.testing.assert_allclose(sm['it_radius'][:n], sm_o['it_radius'][:n], rtol=1e-6)
    assert np.isclose(sm['final_cost'], sm_o['final_cost'], rtol=1e-6)
    scale_p = max(1.0, np.abs(st_o['pose'][:, :3]).max())
    assert np.abs(st['pose'][:, :3] - st_o['pose'][:, :3]).max() < rtol_state * scale_p
    assert np.abs(st['pose'][:, 3:] - st_o['pose'][:, 3:]).max() < rtol_state
    assert np.abs(st['sb'] - st_o['sb']).max() < rtol_state * max(1.0, np.abs(st_o['sb']).max())
    assert np.allclose(st['inv_depth'], st_o['inv_depth'], rtol=1e-4, atol=1e-6)
    return st, sm


@pytest.mark.parametrize("ex,td", [(0, 0), (1, 1)])
def test_enlarged_window_31_frames_2000_landmarks(handle, ex, td):
    """BASELINE configs[4] at full size: K = 31 (WINDOW_SIZE 30), 2000 landmarks, ~16K projection factors, 30 IMU factors,
    prior on the oldest frame; Rc = 186 / 193 -> the large-window path is taken automatically."""
    seq = synth.SyntheticSequence(5 + ex, n_frames=32, K=31, L=2000, estimate_extrinsic=ex, estimate_td=td)
    prob = synth.SyntheticSequence.anchor_prior(seq.window(0))
    st, sm = _vs_cpp_oracle(handle, prob)
    assert sm['num_accepted'] >= 3
    # a second solve from the optimum stays there (idempotence at full size)
    again = dict(prob, pose=st['pose'], sb=st['sb'], ex=st['ex'], td=float(st['td']), inv_depth=st['inv_depth'])
    st2, sm2, _ = handle.ba_optimize(again)
    assert sm2['final_cost'] <= sm['final_cost'] * (1 + 1e-9)


def test_enlarged_window_without_prior(handle):
    seq = synth.SyntheticSequence(8, n_frames=32, K=31, L=2000)
    _vs_cpp_oracle(handle, seq.window(0))


def test_allreduce_hook_over_rccl_single_rank(handle):
    """The hook path end to end on one GPU: torch.distributed 'nccl' (= RCCL) world of ONE rank, the device buffers of
    the library wrapped zero-copy (__cuda_array_interface__), the collective enqueued on the library's HIP stream.  Summing
    over one rank must not change a bit of the result."""
    import os
    import torch
    import torch.distributed as dist
    from vins_mono_amd import shard
    prob = synth.SyntheticSequence.anchor_prior(synth.SyntheticSequence(5, n_frames=17, K=16, L=300).window(0))
    st0, sm0, _ = handle.ba_optimize(prob)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29551")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    calls = []
    inner = shard.torch_allreduce_hook()

    def hook(ptr, count, stream):
        calls.append(count)
        inner(ptr, count, stream)
    try:
        handle.ba_set_allreduce(hook)
        st1, sm1, _ = handle.ba_optimize(prob)
        c1, c2 = handle.ba_reduce_layout()
    finally:
        handle.ba_set_allreduce(None)
        if created:
            dist.destroy_process_group()
    assert sm1['status'] == 0 and set(calls) == {c1, c2} and len(calls) >= 2 * sm1['num_iterations']
    assert np.array_equal(st0['pose'], st1['pose']) and np.array_equal(st0['inv_depth'], st1['inv_depth'])
    assert list(sm0['it_cost']) == list(sm1['it_cost'])


def test_allreduce_hook_in_the_library_single_rank(handle):
    """vg_ba_rccl_init: the RCCL communicator and the ncclAllReduce hook live inside libvinsgpu.so (csrc/vg_rccl.hip, librccl
    opened with dlopen).  World of ONE rank on the box's GPU: the reductions really go through ncclAllReduce on the launch
    stream, and summing over one rank must not change a bit."""
    prob = synth.SyntheticSequence.anchor_prior(synth.SyntheticSequence(5, n_frames=17, K=16, L=300).window(0))
    st0, sm0, _ = handle.ba_optimize(prob)
    h2 = ba.Handle()
    try:
        h2.ba_rccl_init(1, 0, h2.rccl_unique_id())
        st1, sm1, _ = h2.ba_optimize(prob)
        st2, sm2, _ = h2.ba_optimize(prob)          # the communicator is reused
        h2.ba_rccl_finalize()
        st3, sm3, _ = h2.ba_optimize(prob)          # hook removed again
    finally:
        h2.close()
    for st, sm in ((st1, sm1), (st2, sm2), (st3, sm3)):
        assert sm['status'] == 0 and list(sm0['it_cost']) == list(sm['it_cost'])
        assert np.array_equal(st0['pose'], st['pose']) and np.array_equal(st0['inv_depth'], st['inv_depth'])


SHARD_WORKER = r'''
import os, sys, json
ROOT = %r
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g
g.load_package()
from vins_mono_amd import ba, dist_util as D, synth, shard
rank, local, world = D.env_rank()
torch.cuda.set_device(0)                                  # both ranks share the one GPU of the box
assert D.init("gloo")
seq = synth.SyntheticSequence(5, n_frames=32, K=31, L=600)
prob = synth.SyntheticSequence.anchor_prior(seq.window(0))
h = ba.Handle()
st1, sm1, _ = h.ba_optimize(prob)                         # the whole window on this rank alone
sub = shard.shard_problem(prob, rank, world)
h.ba_set_allreduce(shard.torch_allreduce_hook(device_buffers=True))
st, sm, _ = h.ba_optimize(sub)
D.barrier()
lo, hi = (int(v) for v in sub["shard"])
fl = lambda a: [float(v) for v in np.asarray(a).ravel()]
pack = lambda s, m: dict(pose=fl(s["pose"]), sb=fl(s["sb"]), lam=fl(s["inv_depth"]), it_cost=fl(m["it_cost"]),
                         it_flags=[int(v) for v in m["it_flags"]], n=int(m["num_iterations"]), status=int(m["status"]))
sys.stdout.write(json.dumps(dict(rank=rank, lo=lo, hi=hi, sharded=pack(st, sm), single=pack(st1, sm1))) + "\n")
sys.stdout.flush()
D.finish()
'''


def test_two_ranks_share_one_gpu_landmark_shards(tmp_path):
    """The sharded window with the REAL kernels and two ranks: both processes run on the single GPU of the box (RCCL refuses two
    ranks on one device, so the reduce buffers are staged through gloo), each with its contiguous share of the landmarks of a
    31-frame x 600-landmark window.  Both ranks must end with bit-identical frame states, equal to the single-rank solve."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(SHARD_WORKER % root)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29561", str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rows, dec, txt, pos = [], json.JSONDecoder(), r.stdout, 0
    while (pos := txt.find('{"rank"', pos)) >= 0:
        obj, pos = dec.raw_decode(txt, pos)
        rows.append(obj)
    assert len(rows) == 2
    rows.sort(key=lambda d: d["rank"])
    a, b, one = rows[0]["sharded"], rows[1]["sharded"], rows[0]["single"]
    assert rows[0]["lo"] == 0 and rows[0]["hi"] == rows[1]["lo"] and rows[1]["hi"] == 600
    for k in ("pose", "sb", "it_cost", "it_flags", "n", "status"):
        assert a[k] == b[k], k                                        # replicated part: bit-identical on both ranks
    assert a["status"] == 0 and a["n"] == one["n"] and a["it_flags"] == one["it_flags"]
    n = a["n"]
    np.testing.assert_allclose(a["it_cost"][:n], one["it_cost"][:n], rtol=1e-7)
    np.testing.assert_allclose(np.array(a["pose"]), np.array(one["pose"]), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.array(a["lam"] + b["lam"]), np.array(one["lam"]), rtol=1e-6, atol=1e-9)
