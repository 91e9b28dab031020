"""ctypes binding of oracle/_ref/libvins_ref.so — TEST INFRASTRUCTURE.

libvins_ref.so is the REFERENCE'S OWN back-end code (vins_estimator/src/{estimator,feature_manager}.cpp, factor/*,
utility/utility.*), compiled unchanged where it lies under /root/reference by oracle/Makefile (`make ref`) against the header
stand-ins of oracle/ref_stubs/ (Eigen / Ceres / ROS / OpenCV are not installed here).  It is what PINS the restatements
(oracle/ba_numpy.py, oracle/ba_cpu.cpp) and, through them and directly, the HIP path — see tests/test_ref_parity.py.

The library is built in this container only; it travels to the GPU box as a prebuilt file.  `available()` says whether it is
there; nothing in the product imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_ref", "libvins_ref.so")
_REF_SRC = "/root/reference/vins_estimator/src/estimator.cpp"
_LIB_GPU = os.path.join(_HERE, "_ref", "libvins_ref_gpu.so")
_lib = None
_lib_gpu = None
K_REF = 11                       # WINDOW_SIZE + 1 is a compile-time constant of the reference (parameters.h:12)
DP, IP = C.POINTER(C.c_double), C.POINTER(C.c_int)
KIND_POSE, KIND_SB, KIND_EX, KIND_TD = 0, 1, 2, 3
GSIZE = {0: 7, 1: 9, 2: 7, 3: 1}
LSIZE = {0: 6, 1: 9, 2: 6, 3: 1}


def available():
    return os.path.exists(_LIB) or os.path.exists(_REF_SRC)


def lib():
    """libvins_ref.so — or, with VINS_REF_LIB=<path> in the environment, another build of the same driver (the diff kit's
    libvins_ref_real.so: the reference's translation units on the REAL Eigen + Ceres, `make -C oracle ref_real`)."""
    global _lib
    if _lib is None and os.environ.get("VINS_REF_LIB"):
        _lib = _prepare(C.CDLL(os.environ["VINS_REF_LIB"]))
        return _lib
    if _lib is None:
        if os.path.exists(_REF_SRC):          # (re)build when the reference is present; a no-op when up to date
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
        if not os.path.exists(_LIB):
            raise RuntimeError("oracle/_ref/libvins_ref.so is missing and /root/reference is not here to build it")
        _lib = _prepare(C.CDLL(_LIB))
        assert _lib.vref_has_gpu_optimization() == 0
    return _lib


def _prepare(L):
    assert L.vref_abi_version() == 1 and L.vref_window_size() == K_REF - 1
    for name in ("vref_preint_create", "vref_preint_from_terms", "vref_est_create", "vref_est_get_preintegration"):
        getattr(L, name).restype = C.c_void_p
    return L


def gpu_available():
    return os.path.exists(_LIB_GPU)


def lib_gpu():
    """libvins_ref_gpu.so: the same reference objects with Estimator::optimization() replaced by the product's drop-in body
    (vins-mono_amd/host/dropin/estimator_optimization.cpp -> libvinsgpu.so).  Needs a GPU at the first optimization()."""
    global _lib_gpu
    if _lib_gpu is None:
        lib()                                    # (builds both when the reference is present)
        if not os.path.exists(_LIB_GPU):
            raise RuntimeError("oracle/_ref/libvins_ref_gpu.so is missing")
        _lib_gpu = _prepare(C.CDLL(_LIB_GPU))
        assert _lib_gpu.vref_has_gpu_optimization() == 1
    return _lib_gpu


_lib_simt = None


def dlopen_own_scope(path):
    """ctypes.CDLL of a library that is linked against the EMULATED kernel library (tests/simt/_build/libvinsgpu_simt.so) such that its
    vg_* references bind to THAT library even when the process has the product's libvinsgpu.so in the global scope already
    (vins-mono_amd/__init__.py loads it RTLD_GLOBAL: an xdist worker that ran an ABI test first handed the emulated drop-ins the real
    library's vg_create -- "no device" on this GPU-less box, an order-dependent failure).  RTLD_DEEPBIND puts the library's own
    dependency chain in front of the global scope; the sanitizer runtimes refuse that flag, so a process that preloads one (the ASAN
    runs of tests/test_simt_asan.py, which never load the product library) gets the plain local load."""
    mode = os.RTLD_NOW | os.RTLD_LOCAL
    if "asan" not in os.environ.get("LD_PRELOAD", "") and "ubsan" not in os.environ.get("LD_PRELOAD", ""):
        mode |= getattr(os, "RTLD_DEEPBIND", 0)
    return C.CDLL(path, mode=mode)


def simt_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libvins_ref_simt.so")) or os.path.exists(_REF_SRC)


def lib_simt():
    """libvins_ref_simt.so: libvins_ref_gpu.so's objects linked against the EMULATED kernel library (tests/simt) — the drop-in body
    of Estimator::optimization() on the CPU, for the `not gpu` suite."""
    global _lib_simt
    if _lib_simt is None:
        lib()
        if os.path.exists(_REF_SRC):
            subprocess.check_call(["make", "-C", os.path.join(_HERE, "..", "tests", "simt")], stdout=subprocess.DEVNULL)
            subprocess.check_call(["make", "-C", _HERE, "ref_simt"], stdout=subprocess.DEVNULL)
        path = os.path.join(_HERE, "_ref", "libvins_ref_simt.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/_ref/libvins_ref_simt.so is missing")
        _lib_simt = _prepare(dlopen_own_scope(path))
        assert _lib_simt.vref_has_gpu_optimization() == 1
    return _lib_simt


def _d(a):
    return np.ascontiguousarray(a, np.float64)


def _p(a):
    return a.ctypes.data_as(DP)


def _q2R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def configure(acc_n=0.08, acc_w=0.00004, gyr_n=0.004, gyr_w=2.0e-6, g_norm=9.81007, estimate_extrinsic=0, estimate_td=0,
              td=0.0, tr=0.0, row=480.0, num_iterations=8, init_depth=5.0, min_parallax=10.0 / 460.0, ric=None, tic=None, L=None):
    """The globals readParameters() would fill (vins_estimator/src/parameters.cpp:42-137).  They are per library: pass L =
    lib_gpu() to configure the drop-in build."""
    ric = _d(np.eye(3) if ric is None else ric)
    tic = _d(np.zeros(3) if tic is None else tic)
    (L or lib()).vref_set_config(C.c_double(acc_n), C.c_double(acc_w), C.c_double(gyr_n), C.c_double(gyr_w), C.c_double(g_norm),
                          int(estimate_extrinsic), int(estimate_td), C.c_double(td), C.c_double(tr), C.c_double(row), int(num_iterations),
                          C.c_double(init_depth), C.c_double(min_parallax), _p(ric), _p(tic))


def configure_for(prob, cfg=None, L=None, min_parallax=10.0 / 460.0):
    """Globals for a test window (`vins_mono_amd.synth` layout)."""
    from vins_mono_amd import synth
    c = dict(synth.EUROC if cfg is None else cfg)
    assert prob['focal'] == 460.0, "FOCAL_LENGTH is a compile-time constant of the reference (parameters.h:11)"
    configure(c['acc_n'], c['acc_w'], c['gyr_n'], c['gyr_w'], prob['g_norm'], prob['estimate_extrinsic'], prob['estimate_td'],
              float(prob['td']), float(prob['tr']), float(prob['row']), int(prob['max_iters']), 5.0, min_parallax,
              _q2R(prob['ex'][3:]), prob['ex'][:3], L=L)


# ------------------------------------------------------------------------------------------ small entry points
def R2ypr(R):
    out = np.zeros(3)
    lib().vref_R2ypr(_p(_d(R)), _p(out))
    return out


def ypr2R(ypr):
    out = np.zeros(9)
    lib().vref_ypr2R(_p(_d(ypr)), _p(out))
    return out.reshape(3, 3)


def quat_from_R(R):
    out = np.zeros(4)
    lib().vref_quat_from_R(_p(_d(R)), _p(out))
    return out


def pose_plus(x, delta):
    out = np.zeros(7)
    lib().vref_pose_plus(_p(_d(x)), _p(_d(delta)), _p(out))
    return out


def pose_plus_jacobian(x):
    out = np.zeros(42)
    lib().vref_pose_plus_jacobian(_p(_d(x)), _p(out))
    return out.reshape(7, 6)


class Preintegration:
    """IntegrationBase of the reference (factor/integration_base.h)."""

    def __init__(self, acc_0=None, gyr_0=None, ba=None, bg=None, terms=None, handle=None, own=True):
        L = lib()
        self.own = own
        if handle is not None:
            self.h = handle
        elif terms is not None:
            t = terms
            self.h = L.vref_preint_from_terms(C.c_double(float(t['sum_dt'])), _p(_d(t['delta_p'])), _p(_d(t['delta_q'])), _p(_d(t['delta_v'])),
                                              _p(_d(t['lin_ba'])), _p(_d(t['lin_bg'])), _p(_d(t['jacobian'])), _p(_d(t['covariance'])))
        else:
            self.h = L.vref_preint_create(_p(_d(acc_0)), _p(_d(gyr_0)), _p(_d(ba)), _p(_d(bg)))

    def push_back(self, dt, acc, gyr):
        lib().vref_preint_push(C.c_void_p(self.h), C.c_double(dt), _p(_d(acc)), _p(_d(gyr)))

    def repropagate(self, ba, bg):
        lib().vref_preint_repropagate(C.c_void_p(self.h), _p(_d(ba)), _p(_d(bg)))

    def as_dict(self):
        s = C.c_double()
        dp, dq, dv, ba, bg, J, P = np.zeros(3), np.zeros(4), np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(225), np.zeros(225)
        lib().vref_preint_get(C.c_void_p(self.h), C.byref(s), _p(dp), _p(dq), _p(dv), _p(ba), _p(bg), _p(J), _p(P))
        return dict(sum_dt=s.value, delta_p=dp, delta_q=dq, delta_v=dv, lin_ba=ba, lin_bg=bg, jacobian=J.reshape(15, 15),
                    covariance=P.reshape(15, 15))

    def release(self):
        """Hand the object over (an Estimator takes ownership)."""
        self.own = False
        return self.h

    def __del__(self):
        if getattr(self, 'own', False) and getattr(self, 'h', None):
            lib().vref_preint_destroy(C.c_void_p(self.h))
            self.h = None


def preintegrate(samples, ba, bg):
    """samples = [(0, acc_0, gyr_0), (dt, acc, gyr), ...] as `synth.preintegrate` takes them (noise terms: configure())."""
    pre = Preintegration(samples[0][1], samples[0][2], ba, bg)
    for dt, a, g in samples[1:]:
        pre.push_back(dt, a, g)
    return pre.as_dict()


def imu_factor(pre, pose_i, sb_i, pose_j, sb_j, need_jac=True):
    """IMUFactor::Evaluate (factor/imu_factor.h:19-179).  `pre`: Preintegration or a dict of its result terms.
    Returns r(15), [15x7, 15x9, 15x7, 15x9] (global-size Jacobians, the 7th pose column is the zero w column)."""
    if isinstance(pre, dict):
        pre = Preintegration(terms=pre)
    r = np.zeros(15)
    J = [np.zeros((15, 7)), np.zeros((15, 9)), np.zeros((15, 7)), np.zeros((15, 9))]
    a = [_p(j) for j in J] if need_jac else [None] * 4
    lib().vref_imu_factor(C.c_void_p(pre.h), _p(_d(pose_i)), _p(_d(sb_i)), _p(_d(pose_j)), _p(_d(sb_j)), _p(r), *a)
    return r, (J if need_jac else None)


def projection_factor(pose_i, pose_j, ex, inv_dep, pts_i, pts_j, need_jac=True):
    """ProjectionFactor::Evaluate (factor/projection_factor.cpp:21-121); sqrt_info = FOCAL_LENGTH / 1.5 I (estimator.cpp:17)."""
    r = np.zeros(2)
    J = [np.zeros((2, 7)), np.zeros((2, 7)), np.zeros((2, 7)), np.zeros((2, 1))]
    a = [_p(j) for j in J] if need_jac else [None] * 4
    lib().vref_projection_factor(_p(_d(pts_i)), _p(_d(pts_j)), _p(_d(pose_i)), _p(_d(pose_j)), _p(_d(ex)), C.c_double(inv_dep), _p(r), *a)
    return r, (J if need_jac else None)


def projection_td_factor(pose_i, pose_j, ex, inv_dep, td, obs_i, obs_j, need_jac=True):
    """ProjectionTdFactor::Evaluate (factor/projection_td_factor.cpp:34-141); obs rows [x y u v vx vy cur_td]; TR / ROW: configure()."""
    r = np.zeros(2)
    J = [np.zeros((2, 7)), np.zeros((2, 7)), np.zeros((2, 7)), np.zeros((2, 1)), np.zeros((2, 1))]
    a = [_p(j) for j in J] if need_jac else [None] * 5
    oi, oj = _d(obs_i), _d(obs_j)
    lib().vref_projection_td_factor(_p(_d([oi[0], oi[1], 1.0])), _p(_d([oj[0], oj[1], 1.0])), _p(_d(oi[4:6])), _p(_d(oj[4:6])),
                                    C.c_double(oi[6]), C.c_double(oj[6]), C.c_double(oi[3]), C.c_double(oj[3]),
                                    _p(_d(pose_i)), _p(_d(pose_j)), _p(_d(ex)), C.c_double(inv_dep), C.c_double(td), _p(r), *a)
    return r, (J if need_jac else None)


# ------------------------------------------------------------------------------------------ Estimator
class Estimator:
    """The reference's `Estimator` object (estimator.h:26-139)."""

    def __init__(self, L=None):
        self.L = L or lib()
        self.h = self.L.vref_est_create()

    def close(self):
        if self.h:
            self.L.vref_est_destroy(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        self.close()

    # ---- raw accessors
    def set_frame(self, i, pose7, sb9):
        R = _d(_q2R(pose7[3:]))
        self.L.vref_est_set_frame(C.c_void_p(self.h), int(i), _p(_d(pose7[:3])), _p(R), _p(_d(sb9[0:3])), _p(_d(sb9[3:6])), _p(_d(sb9[6:9])))

    def get_frame(self, i):
        P, R, V, Ba, Bg = np.zeros(3), np.zeros(9), np.zeros(3), np.zeros(3), np.zeros(3)
        self.L.vref_est_get_frame(C.c_void_p(self.h), int(i), _p(P), _p(R), _p(V), _p(Ba), _p(Bg))
        return P, R.reshape(3, 3), V, Ba, Bg

    def para(self):
        """vector2double(): pose (K,7), sb (K,9), ex (7,), td."""
        pose, sb, ex, td = np.zeros((K_REF, 7)), np.zeros((K_REF, 9)), np.zeros(7), C.c_double()
        self.L.vref_est_get_para(C.c_void_p(self.h), _p(pose), _p(sb), _p(ex), C.byref(td))
        return pose, sb, ex, td.value

    def set_prior(self, prior):
        if prior is None:
            self.L.vref_est_set_prior(C.c_void_p(self.h), 0, 0, None, None, None, None, None)
            return
        kinds = np.array([b[0] for b in prior['blocks']], np.int32)
        idxs = np.array([b[1] for b in prior['blocks']], np.int32)
        x0 = _d(np.concatenate([np.asarray(x, float).ravel() for x in prior['x0']]))
        J0 = _d(prior['J0'])
        assert J0.shape == (prior['n'], sum(LSIZE[k] for k in kinds))
        self.L.vref_est_set_prior(C.c_void_p(self.h), int(prior['n']), len(kinds), kinds.ctypes.data_as(IP), idxs.ctypes.data_as(IP),
                                  _p(J0), _p(_d(prior['r0'])), _p(x0))

    def get_prior(self):
        nb, ncols = C.c_int(), C.c_int()
        kinds, idxs = np.zeros(64, np.int32), np.zeros(64, np.int32)
        n = self.L.vref_est_get_prior(C.c_void_p(self.h), C.byref(nb), kinds.ctypes.data_as(IP), idxs.ctypes.data_as(IP), C.byref(ncols), None, None, None)
        if n == 0:
            return None
        J0, r0, x0 = np.zeros((n, ncols.value)), np.zeros(n), np.zeros(9 * 64)
        self.L.vref_est_get_prior(C.c_void_p(self.h), C.byref(nb), kinds.ctypes.data_as(IP), idxs.ctypes.data_as(IP), C.byref(ncols), _p(J0), _p(r0), _p(x0))
        blocks = [(int(kinds[b]), int(idxs[b])) for b in range(nb.value)]
        xs, o = [], 0
        for k, _ in blocks:
            xs.append(x0[o:o + GSIZE[k]].copy())
            o += GSIZE[k]
        return dict(n=n, blocks=blocks, J0=J0, r0=r0, x0=xs)

    def features(self):
        cap = self.L.vref_est_num_features(C.c_void_p(self.h))
        ids, st, nb, fl = (np.zeros(max(cap, 1), np.int32) for _ in range(4))
        dep = np.zeros(max(cap, 1))
        n = self.L.vref_est_get_features(C.c_void_p(self.h), cap, ids.ctypes.data_as(IP), st.ctypes.data_as(IP), nb.ctypes.data_as(IP), _p(dep), fl.ctypes.data_as(IP))
        return dict(id=ids[:n], start=st[:n], nobs=nb[:n], depth=dep[:n], solve_flag=fl[:n])

    def gpu_set_option(self, option, value):
        """Drop-in build only: vins_gpu_set_option (1 = forward SOLVER_TIME as max_solver_time_in_seconds, 2 = eigen form of the prior)."""
        self.L.vref_est_gpu_set_option.restype = C.c_int
        rc = self.L.vref_est_gpu_set_option(C.c_void_p(self.h), C.c_int(option), C.c_int(value))
        if rc:
            raise RuntimeError(f"vins_gpu_set_option({option}, {value}) -> {rc}")

    def set_solver_time(self, t):
        self.L.vref_set_solver_time(C.c_double(t))

    def last_trace(self):
        """Per-iteration rows [valid, accepted, cost, candidate cost, radius, step norm] of this estimator's last optimization()."""
        rows = np.zeros((40, 6))
        n = self.L.vref_est_last_trace(C.c_void_p(self.h), 40, _p(rows))
        return rows[:min(n, 40)].copy()

    def solve_trace(self):
        rows, ic, fc, term = np.zeros((64, 10)), C.c_double(), C.c_double(), C.c_int()
        n = self.L.vref_last_solve_trace(64, _p(rows), C.byref(ic), C.byref(fc), C.byref(term))
        its = []
        for r in rows[1:n]:                          # row 0 is Ceres' "iteration 0" (the initial evaluation)
            its.append(dict(iter=int(r[0]), valid=bool(r[1]), accepted=bool(r[2]), cost=r[3], cost_cand=r[4], model_change=r[5], radius=r[6],
                            step_norm=r[7], mu=r[8], exit={0: None, 1: 'parameter_tolerance', 2: 'function_tolerance', 3: 'gradient_tolerance'}[int(r[9])]))
        return dict(initial_cost=ic.value, final_cost=fc.value, num_iterations=len(its), iterations=its,
                    termination={0: 'CONVERGENCE', 1: 'NO_CONVERGENCE', 2: 'FAILURE'}[term.value])

    def marg_factors(self):
        """The linearised factors the last marginalization summed (ResidualBlockInfo::Evaluate, marginalization_factor.cpp:3-69):
        list of (r, [J_block ...], [(kind, idx) ...], [dropped ...]) with kind 4 = landmark, and (m, n)."""
        need = self.L.vref_est_marg_factors(C.c_void_p(self.h), 0, None)
        if need == 0:
            return [], 0, 0
        buf = np.zeros(need)
        self.L.vref_est_marg_factors(C.c_void_p(self.h), need, _p(buf))
        nf, m, n = int(buf[0]), int(buf[1]), int(buf[2])
        o, out = 3, []
        for _ in range(nf):
            nres, nb = int(buf[o]), int(buf[o + 1])
            o += 2
            blocks, ls, drop = [], [], []
            for b in range(nb):
                blocks.append((int(buf[o]), int(buf[o + 1])))
                ls.append(int(buf[o + 2]))
                drop.append(bool(buf[o + 3]))
                o += 4
            r = buf[o:o + nres].copy()
            o += nres
            Js = []
            for b in range(nb):
                Js.append(buf[o:o + nres * ls[b]].reshape(nres, ls[b]).copy())
                o += nres * ls[b]
            out.append((r, Js, blocks, drop))
        assert o == need
        return out, m, n

    # ---- a test window in, Estimator::optimization(), results out
    def load_window(self, prob):
        """Fill the members optimization() reads from a `synth` window dict (K = 11)."""
        K = prob['pose'].shape[0]
        assert K == K_REF, "the reference's window size is a compile-time constant"
        H = C.c_void_p(self.h)
        self.L.vref_est_set_solver_flag(H, 1)
        self.L.vref_est_set_frame_count(H, K - 1)
        for i in range(K):
            self.set_frame(i, prob['pose'][i], prob['sb'][i])
        self.L.vref_est_set_extrinsic(H, _p(_d(_q2R(prob['ex'][3:]))), _p(_d(prob['ex'][:3])), C.c_double(float(prob['td'])))
        for j in range(1, K):
            t = prob['imu'][j - 1]
            if t is None:                            # "no factor": the reference skips intervals longer than 10 s (estimator.cpp:709)
                t = dict(sum_dt=100.0, delta_p=np.zeros(3), delta_q=np.array([0, 0, 0, 1.0]), delta_v=np.zeros(3), lin_ba=np.zeros(3),
                         lin_bg=np.zeros(3), jacobian=np.eye(15), covariance=np.eye(15))
            self.L.vref_est_set_preintegration(H, j, C.c_void_p(Preintegration(terms=t).release()))
        self.L.vref_est_clear_features(H)
        for l in range(len(prob['inv_depth'])):
            o, n = int(prob['obs_off'][l]), int(prob['lm_nobs'][l])
            self.L.vref_est_add_feature(H, l, int(prob['lm_start'][l]), n, _p(_d(prob['obs'][o:o + n])), C.c_double(1.0 / prob['inv_depth'][l]))
        self.set_prior(prob.get('prior'))
        relo = prob.get('relo')
        if relo is not None:
            m = _d([[x, y, float(l)] for (l, x, y) in relo['match']])      # feature_id == landmark index here
            self.L.vref_est_set_relo(H, int(relo.get('local_index', 0)), _p(_d(relo['pose'])), len(m), _p(m),
                                     _p(_d(relo.get('prev_t', np.zeros(3)))), _p(_d(relo.get('prev_r', np.eye(3)))))

    def optimization(self, flag):
        H = C.c_void_p(self.h)
        self.L.vref_est_set_marginalization_flag(H, int(flag))
        self.L.vref_est_optimization(H)

    def state(self, L):
        pose, sb, ex, td = self.para()
        inv = np.zeros(max(L, 1))
        assert self.L.vref_est_feature_count(C.c_void_p(self.h)) == L
        self.L.vref_est_get_depth_vector(C.c_void_p(self.h), _p(inv))
        return dict(pose=pose, sb=sb, ex=ex, td=td, inv_depth=inv[:L])


def optimization(prob, flag, cfg=None):
    """Estimator::optimization() of the REFERENCE on a test window: (state, summary, new prior) like `ba_numpy.optimization`.
    flag: 0 MARGIN_OLD, 1 MARGIN_SECOND_NEW."""
    configure_for(prob, cfg)
    e = Estimator()
    try:
        e.load_window(prob)
        e.optimization(flag)
        st = e.state(len(prob['inv_depth']))
        if prob.get('relo') is not None:
            rp, rt, rq, ry, dr, dt = np.zeros(7), np.zeros(3), np.zeros(4), C.c_double(), np.zeros(9), np.zeros(3)
            e.L.vref_est_get_relo(C.c_void_p(e.h), _p(rp), _p(rt), _p(rq), C.byref(ry), _p(dr), _p(dt))
            st.update(relo_pose=rp, relo_relative_t=rt, relo_relative_q=rq, relo_relative_yaw=ry.value, drift_correct_r=dr.reshape(3, 3), drift_correct_t=dt)
        return st, e.solve_trace(), e.get_prior()
    finally:
        e.close()


def canonical_prior(prior):
    """Reorder a prior's blocks to (pose asc, speed-bias asc, ex, td): the reference's own order follows an
    unordered_map over addresses (marginalization_factor.cpp:176-194).  Returns (blocks, H = J0^T J0, b = J0^T r0, x0)."""
    order = sorted(range(len(prior['blocks'])), key=lambda i: (prior['blocks'][i][0], prior['blocks'][i][1]))
    off, o = [], 0
    for k, _ in prior['blocks']:
        off.append(o)
        o += LSIZE[k]
    cols = np.concatenate([np.arange(off[i], off[i] + LSIZE[prior['blocks'][i][0]]) for i in order])
    J = prior['J0'][:, cols]
    return [prior['blocks'][i] for i in order], J.T @ J, J.T @ prior['r0'], [prior['x0'][i] for i in order]


def assemble_marginalization(facs):
    """A, b of MarginalizationInfo::marginalize (marginalization_factor.cpp:197-255) from the reference's evaluated factors, in
    the canonical order [dropped: pose, speed-bias, landmarks asc | kept: pose asc, speed-bias, ex, td].  Plain sums: the
    arithmetic under test is the factors' (the reference's); this only places the blocks."""
    LS = {0: 6, 1: 9, 2: 6, 3: 1, 4: 1}
    order = {0: 0, 1: 1, 2: 2, 3: 3, 4: 4}
    seen, dropped = set(), set()
    for (_, _, blocks, drop) in facs:
        for b, d in zip(blocks, drop):
            seen.add(b)
            if d:
                dropped.add(b)
    key = lambda b: (order[b[0]], b[1])
    drop_list, keep_list = sorted(dropped, key=key), sorted(seen - dropped, key=key)
    idx, pos = {}, 0
    for b in drop_list + keep_list:
        idx[b] = pos
        pos += LS[b[0]]
    m = sum(LS[b[0]] for b in drop_list)
    A, bv, babs = np.zeros((pos, pos)), np.zeros(pos), np.zeros(pos)
    for (r, Js, blocks, _) in facs:
        for i, bi in enumerate(blocks):
            bv[idx[bi]:idx[bi] + LS[bi[0]]] += Js[i].T @ r
            babs[idx[bi]:idx[bi] + LS[bi[0]]] += np.abs(Js[i]).T @ np.abs(r)
            for j, bj in enumerate(blocks):
                A[idx[bi]:idx[bi] + LS[bi[0]], idx[bj]:idx[bj] + LS[bj[0]]] += Js[i].T @ Js[j]
    return A, bv, babs, m, drop_list, keep_list


def replay(plan, cfg=None):
    """tests/replay_util.py's N-window plan through the REFERENCE: Estimator::optimization() + Estimator::slideWindow()
    (estimator.cpp:670-1126) with the feature list, the depths (removeBackShiftDepth, feature_manager.cpp:275-313), the prior
    and the states carried by the reference's own members.  Returns the CSV rows of pubOdometry (visualization.cpp:157-172)."""
    from vins_mono_amd import synth
    seq, K, W = plan['seq'], plan['K'], plan['W']
    assert K == K_REF
    c = seq.cfg
    base = seq._base()
    configure_for(base, c)
    L = lib()
    pre = [synth.preintegrate(s, seq.ba_lin, seq.bg_lin, c['acc_n'], c['gyr_n'], c['acc_w'], c['gyr_w']) for s in plan['intervals']]
    e = Estimator()
    H = C.c_void_p(e.h)
    out = []
    try:
        L.vref_est_set_solver_flag(H, 1)
        L.vref_est_set_frame_count(H, K - 1)
        L.vref_est_set_extrinsic(H, _p(_d(_q2R(base['ex'][3:]))), _p(_d(base['ex'][:3])), C.c_double(0.0))
        for i in range(K):
            t, pose, sb = plan['frames'][i]
            e.set_frame(i, pose, sb)
            L.vref_est_set_stamp(H, i, C.c_double(t))
            L.vref_est_add_image_frame(H, C.c_double(t))
        for j in range(1, K):
            L.vref_est_set_preintegration(H, j, C.c_void_p(Preintegration(terms=pre[j - 1]).release()))
        for w in range(W):
            if w > 0:
                L.vref_est_set_marginalization_flag(H, 0)
                L.vref_est_slide_window(H)                     # the reference's own shift of states, pre-integrations, depths
                t, pose, sb = plan['frames'][w + K - 1]
                e.set_frame(K - 1, pose, sb)
                L.vref_est_set_stamp(H, K - 1, C.c_double(t))
                L.vref_est_add_image_frame(H, C.c_double(t))
                L.vref_est_set_preintegration(H, K - 1, C.c_void_p(Preintegration(terms=pre[w + K - 2]).release()))
            carried = dict(zip(e.features()['id'].tolist(), e.features()['depth'].tolist()))
            L.vref_est_clear_features(H)
            for r in plan['tables'][w]:
                depth = carried.get(r['id'], 1.0 / r['init'])
                L.vref_est_add_feature(H, int(r['id']), int(r['start']), int(r['nobs']), _p(_d(r['obs'])), C.c_double(depth))
            e.optimization(0)
            P, R, V, _, _ = e.get_frame(K - 1)
            q = quat_from_R(R)
            out.append([plan['frames'][w + K - 1][0] * 1e9, *P, q[3], q[0], q[1], q[2], *V])
    finally:
        e.close()
    return np.array(out)


def factor_tables(prob):
    """Residuals / Jacobians of every factor of a window at its initial state, by the REFERENCE's Evaluate() methods, in the
    table layout of `Handle.ba_eval_factors`: proj_r (F,2), proj_J (F,2,20) [pose_i 6 | pose_j 6 | ex 6 | lambda | td],
    imu_r (K-1,15), imu_J (K-1,15,30) [pose_i 6 | sb_i 9 | pose_j 6 | sb_j 9]."""
    from oracle import ba_numpy as B          # window bookkeeping only (factor list order, state dict)
    configure_for(prob)
    st = B.state_of(prob)
    lay = B.Layout(prob)
    facs = B.factor_list(prob)
    pr, pJ = np.zeros((len(facs), 2)), np.zeros((len(facs), 2, 20))

    def pose(i):
        return st['pose'][i] if i < lay.K else st['relo_pose']
    for f, (l, fi, fj, oi, oj) in enumerate(facs):
        if lay.est_td:
            r, J = projection_td_factor(pose(fi), pose(fj), st['ex'], st['inv_depth'][l], st['td'], oi, oj)
            pJ[f, :, 19] = J[4][:, 0]
        else:
            r, J = projection_factor(pose(fi), pose(fj), st['ex'], st['inv_depth'][l], [oi[0], oi[1], 1.0], [oj[0], oj[1], 1.0])
        pr[f] = r
        pJ[f, :, 0:6], pJ[f, :, 6:12], pJ[f, :, 12:18], pJ[f, :, 18] = J[0][:, :6], J[1][:, :6], J[2][:, :6], J[3][:, 0]
        assert not J[0][:, 6].any() and not J[1][:, 6].any() and not J[2][:, 6].any()      # the w column of the global Jacobians
    K = lay.K
    ir, iJ = np.zeros((K - 1, 15)), np.zeros((K - 1, 15, 30))
    for k in range(K - 1):
        if prob['imu'][k] is None:
            continue
        r, J = imu_factor(prob['imu'][k], st['pose'][k], st['sb'][k], st['pose'][k + 1], st['sb'][k + 1])
        ir[k] = r
        iJ[k] = np.hstack([J[0][:, :6], J[1], J[2][:, :6], J[3]])
    return pr, pJ, ir, iJ


def run_sequence(seq, n_frames, L=None, min_parallax=10.0 / 460.0, noise_seed=0, reset_at=None, collect_priors=True, gpu_options=None):
    """The reference's own per-frame loop on a synthetic sequence: Estimator::processIMU for every IMU sample and
    Estimator::processImage for every frame (estimator.cpp:81-215) — feature bookkeeping, key-frame decision by parallax,
    triangulation, optimization(), failure detection, slideWindow() for BOTH marginalization flags with the IMU buffers merged
    (:1069-1099), removeFailures.  The SfM bootstrap (initial/*) is bypassed: the first WINDOW_SIZE frames are collected in
    INITIAL mode exactly as the reference does, then the window is given noisy ground-truth states and switched to
    NON_LINEAR (what initialStructure() + visualInitialAlign() hand over).
    L: lib() (all reference) or lib_gpu() (optimization() = the product's drop-in).  Returns a list of per-frame records.
    reset_at: a frame index before which the estimator is reset (clearState + setParameter) and bootstrapped again.
    collect_priors=False leaves the drop-in's marginalization result on the device between frames (its normal operation;
    get_prior() fetches it early).  gpu_options: {option: value} for vins_gpu_set_option (drop-in builds)."""
    L = L or lib()
    K = K_REF
    c = seq.cfg
    base = seq._base()
    configure_for(base, c, L=L, min_parallax=min_parallax)
    rng = np.random.default_rng(noise_seed)
    e = Estimator(L=L)
    for opt, val in (gpu_options or {}).items():
        e.gpu_set_option(opt, val)
    H = C.c_void_p(e.h)
    h = seq.frame_dt / seq.imu_per_frame

    def noisy_state(f):
        th = rng.normal(0, np.radians(0.3), 3)
        Rn = seq.Rm[f] @ (np.eye(3) + np.array([[0, -th[2], th[1]], [th[2], 0, -th[0]], [-th[1], th[0], 0]]))
        q = quat_from_R(Rn)
        q = q / np.linalg.norm(q)
        return np.concatenate([seq.P[f] + rng.normal(0, 0.03, 3), q]), np.concatenate([seq.V[f] + rng.normal(0, 0.03, 3), seq.ba_lin, seq.bg_lin])

    def feed_imu(f):                                   # the samples between frame f and f + 1
        t = seq.times[f]
        for s_ in range(1, seq.imu_per_frame + 1):
            a, g = seq._imu_sample(t + s_ * h)
            L.vref_est_process_imu(H, C.c_double(h), _p(_d(a)), _p(_d(g)))

    def image(f):
        ids, rows = [], []
        for lid, lm in enumerate(seq.lm):
            k = f - lm['f0']
            if 0 <= k < len(lm['obs']):
                xy = lm['obs'][k]
                prev = lm['obs'][k - 1] if k > 0 else xy
                vel = (xy - prev) / seq.frame_dt
                ids.append(lid)
                rows.append([xy[0], xy[1], 1.0, c['fx'] * xy[0] + c['cx'], c['fy'] * xy[1] + c['cy'], vel[0], vel[1]])
        ids = np.array(ids, np.int32)
        rows = _d(rows)
        L.vref_est_process_image(H, C.c_double(float(seq.times[f])), len(ids), ids.ctypes.data_as(IP), _p(rows))

    def bootstrap(f0):
        """frames f0 .. f0 + WINDOW_SIZE - 1 in INITIAL mode (they are only counted in), then noisy ground-truth states"""
        L.vref_est_set_extrinsic(H, _p(_d(_q2R(base['ex'][3:]))), _p(_d(base['ex'][:3])), C.c_double(0.0))
        L.vref_est_set_g(H, _p(_d([0.0, 0.0, c['g_norm']])))
        zero = np.zeros(3)
        for i in range(K):                             # linearisation biases of the pre-integrations created in INITIAL mode
            L.vref_est_set_frame(H, i, _p(zero), _p(_d(np.eye(3))), _p(zero), _p(_d(seq.ba_lin)), _p(_d(seq.bg_lin)))
        a0, g0 = seq._imu_sample(seq.times[f0])
        L.vref_est_process_imu(H, C.c_double(0.0), _p(_d(a0)), _p(_d(g0)))     # first_imu: acc_0 / gyr_0
        for f in range(f0, f0 + K - 1):                # frames 0 .. WINDOW_SIZE-1: INITIAL mode only counts them in
            if f > f0:
                feed_imu(f - 1)
            image(f)
        assert L.vref_est_get_frame_count(H) == K - 1 and L.vref_est_get_solver_flag(H) == 0
        for i in range(K - 1):
            pose, sb = noisy_state(f0 + i)
            e.set_frame(i, pose, sb)
        pose, sb = noisy_state(f0 + K - 2)             # slot WINDOW_SIZE starts from the previous frame and is propagated by processIMU
        e.set_frame(K - 1, pose, sb)
        L.vref_est_set_solver_flag(H, 1)
        L.vref_est_set_last_from_window(H)

    out = []
    try:
        bootstrap(0)
        f = K - 1
        while f < n_frames:
            if reset_at is not None and f == reset_at:
                # restart_callback (estimator_node.cpp:182-198) between two frames: clearState() + setParameter(), then the
                # estimator collects a fresh window -- whatever the previous optimization() left behind must not reach it
                L.vref_est_clear_state(H)
                bootstrap(f)
                f += K - 1
                reset_at = None
                continue
            feed_imu(f - 1)
            image(f)
            pose, sb, ex, td = e.para()
            feats = e.features()
            out.append(dict(frame=f, flag=int(L.vref_est_get_marginalization_flag(H)), pose=pose.copy(), sb=sb.copy(), ex=ex.copy(), td=td,
                            n_features=len(feats['id']), depth=dict(zip(feats['id'].tolist(), feats['depth'].tolist())),
                            solver_flag=int(L.vref_est_get_solver_flag(H)), prior=e.get_prior() if collect_priors else False, iterations=int(L.vref_est_last_iterations(H)),
                            trace=e.last_trace()))
            f += 1
    finally:
        e.close()
    return out
