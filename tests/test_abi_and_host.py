"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/vinsgpu.h declares, the ctypes structs
mirror the header, the host-side packing rejects malformed problems, and the front-end oracle obeys the integer
identities its OpenCV originals have (the reference has no fixtures to pin it against)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import __graft_entry__ as graft
from oracle import fe_cpu as F
from vins_mono_amd import ba, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "vinsgpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pkg):
    lib = C.CDLL(pkg.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    lib.vg_abi_version.restype = C.c_int
    assert lib.vg_abi_version() == 12 == ba.VG_ABI_VERSION


def test_config_struct_is_validated_before_any_device_is_touched(pkg):
    """vg_create_config (ABI 10): a malformed vg_config is VG_ERR_BAD_ARG whether or not a GPU is present."""
    lib = C.CDLL(pkg.LIB_PATH)
    lib.vg_create_config.argtypes = [C.POINTER(ba.Config), C.POINTER(C.c_void_p)]
    h = C.c_void_p()
    assert lib.vg_create_config(None, C.byref(h)) == -1
    for bad in (dict(struct_size=4), dict(launch_mode=7), dict(marg_mode=5), dict(fused_min_windows=-2), dict(pack_threads=65), dict(imu_info_mode=2)):
        cfg = ba.Config(struct_size=C.sizeof(ba.Config), device=0)
        for k, v in bad.items():
            setattr(cfg, k, v)
        assert lib.vg_create_config(C.byref(cfg), C.byref(h)) == -1, bad
    assert C.sizeof(ba.Config) == 28


def test_struct_sizes_match_header():
    # sizes implied by include/vinsgpu.h on LP64
    assert C.sizeof(ba.ImuPreint) == 8 * (1 + 3 + 4 + 3 + 3 + 3 + 225 + 225) + 8
    assert C.sizeof(ba.Summary) == 16 + 24 + 5 * 8 * 32 + 4 * 32 + 8 * 16 + 8 * 12
    assert C.sizeof(ba.State) == 6 * 8
    assert C.sizeof(ba.Prior) == 6 * 4 + 5 * 8


def test_no_gpu_means_loud_failure(pkg):
    """There is no CPU fallback: without a device vg_create must fail (VG_ERR_NO_DEVICE), never silently succeed."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        ba.Handle()


def test_packed_problem_roundtrip_fields():
    prob = synth.SyntheticSequence(5, L=20).window(0)
    p = ba.PackedProblem(prob)
    s = p.struct
    assert s.K == 11 and s.L == len(prob['inv_depth']) and s.n_obs == prob['obs'].shape[0]
    assert np.isclose(s.imu[3].sum_dt, prob['imu'][3]['sum_dt']) and s.imu[3].valid == 1
    assert s.prior_n == 0 and s.relo_n == 0 and s.max_iters == 8
    assert np.allclose(np.ctypeslib.as_array(s.pose, (11, 7)), prob['pose'])


# ---------------------------------------------------------------- front-end oracle identities
def test_pyrdown_constant_and_sizes():
    img = np.full((480, 752), 93, np.uint8)
    d = F.pyrdown(img)
    assert d.shape == (240, 376) and np.all(d == 93)          # kernel sums to 256
    odd = np.arange(35 * 51, dtype=np.uint8).reshape(35, 51)
    assert F.pyrdown(odd).shape == (18, 26)


def test_pyrdown_matches_numpy_restatement():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (61, 94), dtype=np.uint8)
    k = np.array([1, 4, 6, 4, 1])
    h, w = img.shape
    ref = lambda i, n: -i if i < 0 else (2 * n - 2 - i if i >= n else i)
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    for y in range(out.shape[0]):
        for x in range(out.shape[1]):
            acc = 0
            for v in range(5):
                for u in range(5):
                    acc += int(k[v]) * int(k[u]) * int(img[ref(2 * y - 2 + v, h), ref(2 * x - 2 + u, w)])
            out[y, x] = (acc + 128) >> 8
    assert np.array_equal(F.pyrdown(img), out)


def test_scharr_on_ramps():
    x = np.tile(np.arange(60, dtype=np.uint8) * 2, (40, 1))
    d = F.scharr(x)
    assert np.all(d[1:-1, 1:-1, 0] == 2 * 2 * 16) and np.all(d[:, :, 1] == 0)     # slope 2 per px, gain 32 per unit slope
    assert np.all(d[:, 0, 0] == 0)                                                  # reflect-101: t[-1] = t[1]


def test_lk_recovers_pure_translation_and_flags_flat_patches():
    a = synth.synth_frame(21)
    b = synth.warp_frame(a, 22, shift=(2.0, -3.0), angle_deg=0.0, noise=0.0)
    pts = F.gftt(a, 80)
    out, st, err = F.lk(a, b, pts)
    ok = st == 1
    assert ok.sum() >= 75
    flow = out[ok] - pts[ok]
    assert np.abs(np.median(flow[:, 0]) - 2.0) < 0.05 and np.abs(np.median(flow[:, 1]) + 3.0) < 0.05
    flat = np.full_like(a, 128)
    _, st2, _ = F.lk(flat, flat, pts[:5])
    assert np.all(st2 == 0)


def test_gftt_respects_min_distance_mask_and_order():
    a = synth.synth_frame(23)
    mask = np.full(a.shape, 255, np.uint8)
    mask[:, 300:] = 0
    c = F.gftt(a, 150, 0.01, 30.0, mask)
    assert len(c) > 20 and np.all(c[:, 0] < 300)
    d = np.linalg.norm(c[:, None, :] - c[None, :, :], axis=2) + np.eye(len(c)) * 1e9
    assert d.min() >= 30.0
    e = F.mineig(a)
    vals = e[c[:, 1].astype(int), c[:, 0].astype(int)]
    assert np.all(np.diff(vals) <= 0)                  # accepted in descending corner-response order
    assert np.array_equal(F.gftt(a, 10, 0.01, 30.0, mask), c[:10])


def test_clahe_properties():
    a = synth.synth_frame(24)
    out = F.clahe(a)
    assert out.shape == a.shape and out.std() > a.std() * 0.8
    const = np.full((480, 752), 10, np.uint8)
    o2 = F.clahe(const)
    assert len(np.unique(o2)) == 1                      # a constant image stays constant
