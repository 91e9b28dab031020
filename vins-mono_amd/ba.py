"""ctypes binding of the BA entry points of include/vinsgpu.h (thin; no arithmetic lives here).

`prob` dicts are the ones produced by `synth.py` (and by the tests): see synth.py's docstring.
"""
import time
import ctypes as C
import numpy as np

from . import lib

VG_MAX_ITERS = 32
VG_MARGIN_OLD, VG_MARGIN_SECOND_NEW, VG_MARGIN_NONE = 0, 1, 2
VG_MARG_SQRT, VG_MARG_EIGEN = 0, 1     # vg_ba_set_marg_mode
VG_IMU_INFO_FACTOR, VG_IMU_INFO_REFERENCE = 0, 1     # vg_ba_set_imu_info_mode
VG_OK = 0
VG_ABI_VERSION = 12         # include/vinsgpu.h
VG_LAUNCH_DIRECT, VG_LAUNCH_GRAPH = 0, 1   # vg_ba_set_launch_mode
VG_LAUNCH_DEFAULT = VG_LAUNCH_DIRECT        # include/vinsgpu.h
VG_PRIOR_RESIDENT = -1
_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int)


class ImuPreint(C.Structure):
    _fields_ = [("sum_dt", C.c_double), ("delta_p", C.c_double * 3), ("delta_q", C.c_double * 4),
                ("delta_v", C.c_double * 3), ("linearized_ba", C.c_double * 3), ("linearized_bg", C.c_double * 3),
                ("jacobian", C.c_double * 225), ("covariance", C.c_double * 225), ("valid", C.c_int), ("_pad", C.c_int)]


class Problem(C.Structure):
    _fields_ = [("K", C.c_int), ("L", C.c_int), ("n_obs", C.c_int),
                ("pose", _pd), ("speedbias", _pd), ("ex_pose", _pd), ("td", C.c_double), ("inv_depth", _pd),
                ("lm_start", _pi), ("lm_nobs", _pi), ("lm_obs_off", _pi), ("obs", _pd), ("imu", C.POINTER(ImuPreint)),
                ("prior_n", C.c_int), ("prior_nblocks", C.c_int), ("prior_block_kind", _pi), ("prior_block_index", _pi),
                ("prior_J0", _pd), ("prior_r0", _pd), ("prior_x0", _pd),
                ("relo_n", C.c_int), ("relo_pose", _pd), ("relo_lm", _pi), ("relo_xy", _pd),
                ("estimate_extrinsic", C.c_int), ("estimate_td", C.c_int), ("max_iters", C.c_int),
                ("focal", C.c_double), ("tr", C.c_double), ("row", C.c_double), ("g_norm", C.c_double),
                ("max_solver_time_s", C.c_double)]


class State(C.Structure):
    _fields_ = [("pose", _pd), ("speedbias", _pd), ("ex_pose", _pd), ("td", _pd), ("inv_depth", _pd), ("relo_pose", _pd)]


class Summary(C.Structure):
    _fields_ = [("status", C.c_int), ("termination", C.c_int), ("num_iterations", C.c_int), ("num_accepted", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("final_radius", C.c_double),
                ("it_cost", C.c_double * VG_MAX_ITERS), ("it_cost_cand", C.c_double * VG_MAX_ITERS),
                ("it_model", C.c_double * VG_MAX_ITERS), ("it_radius", C.c_double * VG_MAX_ITERS),
                ("it_step_norm", C.c_double * VG_MAX_ITERS), ("it_flags", C.c_int * VG_MAX_ITERS), ("prof", C.c_double * 16),
                ("gauge_rot", C.c_double * 9), ("gauge_p0", C.c_double * 3)]


class Prior(C.Structure):
    _fields_ = [("cap", C.c_int), ("cap_blocks", C.c_int), ("n", C.c_int), ("m", C.c_int), ("nblocks", C.c_int),
                ("valid", C.c_int), ("block_kind", _pi), ("block_index", _pi), ("J0", _pd), ("r0", _pd), ("x0", _pd)]


class Tracks(C.Structure):            # vg_ba_tracks
    _fields_ = [("n_features", C.c_int), ("feature_id", _pi), ("start_frame", _pi), ("n_obs", _pi), ("solve_flag", _pi),
                ("depth", _pd), ("obs", _pd)]


class SeqConfig(C.Structure):         # vg_ba_seq_config
    _fields_ = [("max_features", C.c_int), ("max_new_obs", C.c_int), ("max_landmarks", C.c_int), ("max_factors", C.c_int),
                ("init_depth", C.c_double), ("min_parallax", C.c_double)]


class Frame(C.Structure):             # vg_ba_frame
    _fields_ = [("pose", C.c_double * 7), ("speedbias", C.c_double * 9), ("imu_new", C.POINTER(ImuPreint)),
                ("imu_merged", C.POINTER(ImuPreint)), ("n_obs", C.c_int), ("feature_id", _pi), ("obs", _pd)]


SEQ_INFO = ("flag", "n_features", "n_tracked", "n_parallax", "n_landmarks", "n_factors", "status", "n_after")


def imu_struct(m):
    """vg_imu_preint from a pre-integration dict (synth.preintegrate / Handle.imu_preintegrate); None: no factor."""
    q = ImuPreint()
    if m is None:
        q.valid = 0
        return q
    q.sum_dt = float(m['sum_dt'])
    q.delta_p[:] = list(m['delta_p']); q.delta_q[:] = list(m['delta_q']); q.delta_v[:] = list(m['delta_v'])
    q.linearized_ba[:] = list(m['lin_ba']); q.linearized_bg[:] = list(m['lin_bg'])
    q.jacobian[:] = list(np.asarray(m['jacobian'], float).ravel())
    q.covariance[:] = list(np.asarray(m['covariance'], float).ravel())
    q.valid = 1
    return q


def _dp(a):
    return a.ctypes.data_as(_pd)


def _ip(a):
    return a.ctypes.data_as(_pi)


_GS = {0: 7, 1: 9, 2: 7, 3: 1}


class PackedProblem:
    """Owns contiguous copies of a prob dict's arrays and the ctypes struct pointing at them."""

    def __init__(self, prob):
        f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i4 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        self.K = int(prob['pose'].shape[0])
        self.L = int(prob['inv_depth'].shape[0])
        self.keep = k = {}
        k['pose'], k['sb'], k['ex'] = f8(prob['pose']), f8(prob['sb']), f8(prob['ex'])
        k['lam'] = f8(prob['inv_depth'])
        k['lm_start'], k['lm_nobs'], k['obs_off'] = i4(prob['lm_start']), i4(prob['lm_nobs']), i4(prob['obs_off'])
        k['obs'] = f8(prob['obs']).reshape(-1, 7) if self.L else np.zeros((0, 7))
        imu = (ImuPreint * (self.K - 1))()
        for i, m in enumerate(prob['imu']):
            if m is None:
                imu[i].valid = 0
                continue
            imu[i].sum_dt = float(m['sum_dt'])
            imu[i].delta_p[:] = list(m['delta_p'])
            imu[i].delta_q[:] = list(m['delta_q'])
            imu[i].delta_v[:] = list(m['delta_v'])
            imu[i].linearized_ba[:] = list(m['lin_ba'])
            imu[i].linearized_bg[:] = list(m['lin_bg'])
            imu[i].jacobian[:] = list(np.asarray(m['jacobian'], float).ravel())
            imu[i].covariance[:] = list(np.asarray(m['covariance'], float).ravel())
            imu[i].valid = 1
        k['imu'] = imu
        p = Problem()
        p.K, p.L, p.n_obs = self.K, self.L, int(k['obs'].shape[0])
        p.pose, p.speedbias, p.ex_pose, p.td = _dp(k['pose']), _dp(k['sb']), _dp(k['ex']), float(prob['td'])
        p.inv_depth, p.lm_start, p.lm_nobs, p.lm_obs_off = _dp(k['lam']), _ip(k['lm_start']), _ip(k['lm_nobs']), _ip(k['obs_off'])
        p.obs, p.imu = _dp(k['obs']), imu
        pr = prob.get('prior')
        if isinstance(pr, str) and pr == 'resident':      # the prior the batch slot holds on the device
            p.prior_n = VG_PRIOR_RESIDENT
        elif pr is not None:
            k['pk'] = i4([b[0] for b in pr['blocks']])
            k['pidx'] = i4([b[1] for b in pr['blocks']])
            k['J0'], k['r0'] = f8(pr['J0']), f8(pr['r0'])
            k['x0'] = f8(np.concatenate([np.atleast_1d(np.asarray(v, float)) for v in pr['x0']]))
            p.prior_n, p.prior_nblocks = int(pr['n']), len(pr['blocks'])
            p.prior_block_kind, p.prior_block_index = _ip(k['pk']), _ip(k['pidx'])
            p.prior_J0, p.prior_r0, p.prior_x0 = _dp(k['J0']), _dp(k['r0']), _dp(k['x0'])
        relo = prob.get('relo')
        if relo is not None:
            k['relo_pose'] = f8(relo['pose'])
            k['relo_lm'] = i4([m[0] for m in relo['match']])
            k['relo_xy'] = f8([[m[1], m[2]] for m in relo['match']])
            p.relo_n, p.relo_pose, p.relo_lm, p.relo_xy = len(relo['match']), _dp(k['relo_pose']), _ip(k['relo_lm']), _dp(k['relo_xy'])
        p.estimate_extrinsic, p.estimate_td, p.max_iters = int(prob['estimate_extrinsic']), int(prob['estimate_td']), int(prob['max_iters'])
        p.focal, p.tr, p.row, p.g_norm = float(prob['focal']), float(prob['tr']), float(prob['row']), float(prob['g_norm'])
        p.max_solver_time_s = float(prob.get('max_solver_time_s', 0.0))
        self.struct = p
        self.has_relo = relo is not None


class PackedBatch:
    """The vg_ba_problem* array of a batch, built once: what a native caller holds between frames."""

    def __init__(self, probs):
        self.packed = [p if isinstance(p, PackedProblem) else PackedProblem(p) for p in probs]
        self.arr = (C.POINTER(Problem) * len(self.packed))(*[C.pointer(p.struct) for p in self.packed])


class _Out:
    def __init__(self, K, L, has_relo, want_prior):
        self.pose, self.sb = np.zeros((K, 7)), np.zeros((K, 9))
        self.ex, self.td, self.lam = np.zeros(7), np.zeros(1), np.zeros(max(L, 1))
        self.relo = np.zeros(7)
        self.L = L
        self.state = State(_dp(self.pose), _dp(self.sb), _dp(self.ex), _dp(self.td), _dp(self.lam),
                           _dp(self.relo) if has_relo else None)
        self.prior = None
        if want_prior:
            cap, capb = 6 * K + 32, K + 8
            self.pk, self.pidx = np.zeros(capb, np.int32), np.zeros(capb, np.int32)
            self.J0, self.r0, self.x0 = np.zeros(cap * cap), np.zeros(cap), np.zeros(9 * capb)
            self.prior = Prior(cap, capb, 0, 0, 0, 0, _ip(self.pk), _ip(self.pidx), _dp(self.J0), _dp(self.r0), _dp(self.x0))

    def state_dict(self, has_relo):
        st = dict(pose=self.pose.copy(), sb=self.sb.copy(), ex=self.ex.copy(), td=float(self.td[0]),
                  inv_depth=self.lam[:self.L].copy())
        if has_relo:
            st['relo_pose'] = self.relo.copy()
        return st

    def prior_dict(self):
        q = self.prior
        if q is None or not q.valid:
            return None
        n, nb = q.n, q.nblocks
        blocks = [(int(self.pk[b]), int(self.pidx[b])) for b in range(nb)]
        x0, off = [], 0
        for kind, _ in blocks:
            x0.append(self.x0[off:off + _GS[kind]].copy())
            off += _GS[kind]
        return dict(n=n, m=q.m, blocks=blocks, J0=self.J0[:n * n].reshape(n, n).copy(), r0=self.r0[:n].copy(), x0=x0)


def summary_dict(s):
    n = s.num_iterations
    return dict(status=s.status, termination=s.termination, num_iterations=n, num_accepted=s.num_accepted,
                initial_cost=s.initial_cost, final_cost=s.final_cost, final_radius=s.final_radius,
                it_cost=np.array(s.it_cost[:n]), it_cost_cand=np.array(s.it_cost_cand[:n]),
                it_model=np.array(s.it_model[:n]), it_radius=np.array(s.it_radius[:n]),
                it_step_norm=np.array(s.it_step_norm[:n]), it_flags=np.array(s.it_flags[:n]), prof=np.array(s.prof[:]),
                gauge_rot=np.array(s.gauge_rot[:]).reshape(3, 3), gauge_p0=np.array(s.gauge_p0[:]))


# vg_allreduce_fn(user, device_buf, count, stream) -> 0 on success
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class Config(C.Structure):
    """vg_config (include/vinsgpu.h, ABI 12): everything that shapes a handle in one struct; a handle made from it never reads the environment."""
    _fields_ = [("struct_size", C.c_int), ("device", C.c_int), ("launch_mode", C.c_int), ("marg_mode", C.c_int),
                ("fused_min_windows", C.c_int), ("pack_threads", C.c_int), ("imu_info_mode", C.c_int)]


class Handle:
    """vg_create / vg_destroy wrapper; raises RuntimeError with vg_last_error on failures.  config: a dict of vg_config fields
    (device: None = the current one, else the HIP device index; launch_mode = 'graph' | 'direct', marg_mode, fused_min_windows,
    pack_threads, imu_info_mode) -> vg_create_config."""

    def __init__(self, config=None):
        self.lib = lib()
        L = self.lib
        L.vg_create.argtypes = [C.POINTER(C.c_void_p)]
        L.vg_destroy.argtypes = [C.c_void_p]
        L.vg_sync.argtypes = [C.c_void_p]
        L.vg_last_error.argtypes = [C.c_void_p]
        L.vg_last_error.restype = C.c_char_p
        L.vg_timer_start.argtypes = [C.c_void_p]
        L.vg_timer_stop.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.vg_ba_batch_upload.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(Problem)), _pi]
        L.vg_ba_batch_run_async.argtypes = [C.c_void_p]
        L.vg_ba_batch_download.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(State)), C.POINTER(Summary),
                                           C.POINTER(C.POINTER(Prior))]
        L.vg_ba_batch_download_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(State)), C.POINTER(Summary)]
        L.vg_ba_batch_download_prior.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(Prior))]
        L.vg_ba_batch_run_timed.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.vg_ba_batch_run_profiled.argtypes = [C.c_void_p, C.POINTER(C.c_float), _pi]
        L.vg_ba_batch_flops_by_kernel.argtypes = [C.c_void_p, _pd]
        L.vg_ba_batch_info.argtypes = [C.c_void_p, _pd, _pd, _pd, _pi]
        L.vg_ba_batch_flops.argtypes = [C.c_void_p, _pd, _pd]
        L.vg_ba_eval_factors.argtypes = [C.c_void_p, C.POINTER(Problem), _pd, _pd, _pd, _pd, _pd]
        L.vg_ba_set_large_window.argtypes = [C.c_void_p, C.c_int]
        L.vg_ba_set_marg_mode.argtypes = [C.c_void_p, C.c_int]
        L.vg_ba_set_launch_mode.argtypes = [C.c_void_p, C.c_int]
        L.vg_ba_set_fused_min_windows.argtypes = [C.c_void_p, C.c_int]
        L.vg_host_register.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.vg_host_unregister.argtypes = [C.c_void_p, C.c_void_p]
        L.vg_ba_launch_stats.argtypes = [C.c_void_p, _pi, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        if hasattr(L, 'vg_ba_rccl_init'):            # (absent from the CPU-emulated build of tests/simt)
            L.vg_ba_rccl_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
            L.vg_ba_rccl_finalize.argtypes = [C.c_void_p]
        L.vg_ba_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p]
        L.vg_ba_reduce_layout.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.vg_triangulate.argtypes = [C.c_void_p, C.c_int, _pd, _pd, _pd, _pd, C.c_int, _pi, _pi, _pi, _pd, C.c_double, _pd]
        L.vg_imu_preintegrate.argtypes = [C.c_void_p, C.c_int, _pi, _pd, _pd, _pd, _pd, C.POINTER(ImuPreint)]
        L.vg_ba_reserve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.vg_ba_seq_begin.argtypes = [C.c_void_p, C.c_int, C.POINTER(SeqConfig), C.POINTER(C.POINTER(Problem)), C.POINTER(C.POINTER(Tracks))]
        L.vg_ba_seq_step_async.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(Frame))]
        L.vg_ba_seq_info.argtypes = [C.c_void_p, C.c_int, _pi]
        L.vg_ba_seq_get_tracks.argtypes = [C.c_void_p, C.c_int, C.c_int, _pi, _pi, _pi, _pi, _pi, _pd, _pd]
        L.vg_ba_seq_end.argtypes = [C.c_void_p]
        L.vg_ba_seq_export.argtypes = [C.c_void_p, C.c_int, _pd, _pd, _pd, _pd, C.POINTER(ImuPreint), C.POINTER(Prior)]
        L.vg_ba_seq_import.argtypes = [C.c_void_p, C.c_int, C.POINTER(Problem), C.POINTER(Tracks)]
        if L.vg_abi_version() != VG_ABI_VERSION:
            raise RuntimeError(f"libvinsgpu.so reports ABI version {L.vg_abi_version()}, this binding was written for {VG_ABI_VERSION}")
        self.h = C.c_void_p()
        if config is None:
            rc = L.vg_create(C.byref(self.h))
        else:
            dev = config.get("device")                      # vg_config::device: 0 = the current device, k + 1 = device k
            cfg = Config(struct_size=C.sizeof(Config), device=0 if dev is None or int(dev) < 0 else int(dev) + 1,
                         launch_mode={None: 0, "direct": 1, "graph": 2}[config.get("launch_mode")], marg_mode=int(config.get("marg_mode", 0)),
                         fused_min_windows=int(config.get("fused_min_windows", 0)), pack_threads=int(config.get("pack_threads", 0)),
                         imu_info_mode=int(config.get("imu_info_mode", 0)))
            L.vg_create_config.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
            rc = L.vg_create_config(C.byref(cfg), C.byref(self.h))
        if rc != VG_OK:
            raise RuntimeError(f"vg_create failed with status {rc} (no HIP device?)")

    def close(self):
        if self.h:
            self.lib.vg_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != VG_OK:
            raise RuntimeError(f"{what} failed: status {rc}: {self.lib.vg_last_error(self.h).decode()}")

    # ---- timers
    def timer_start(self):
        self._chk(self.lib.vg_timer_start(self.h), "vg_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        self._chk(self.lib.vg_timer_stop(self.h, C.byref(ms)), "vg_timer_stop")
        return float(ms.value)

    def sync(self):
        self._chk(self.lib.vg_sync(self.h), "vg_sync")

    # ---- BA
    def ba_upload(self, probs, margin_flags=None):
        if isinstance(probs, PackedBatch):                # pointer array built once (the boundary loops of bench.py)
            self._packed, arr = probs.packed, probs.arr
            n = len(self._packed)
        else:
            self._packed = [p if isinstance(p, PackedProblem) else PackedProblem(p) for p in probs]
            n = len(self._packed)
            arr = (C.POINTER(Problem) * n)(*[C.pointer(p.struct) for p in self._packed])
        mf = np.ascontiguousarray(margin_flags if margin_flags is not None else [VG_MARGIN_NONE] * n, dtype=np.int32)
        self._margin = mf
        t0 = time.perf_counter()
        self._chk(self.lib.vg_ba_batch_upload(self.h, n, arr, _ip(mf)), "vg_ba_batch_upload")
        self.last_upload_call_ms = (time.perf_counter() - t0) * 1e3        # the C-ABI call alone (pack + H2D + sync)

    def ba_run_async(self):
        self._chk(self.lib.vg_ba_batch_run_async(self.h), "vg_ba_batch_run_async")

    def ba_set_launch_mode(self, mode):
        """VG_LAUNCH_DIRECT (0) / VG_LAUNCH_GRAPH (1): one launch per kernel, or the captured pipeline replayed as a hipGraph."""
        self._chk(self.lib.vg_ba_set_launch_mode(self.h, int(mode)), "vg_ba_set_launch_mode")

    def host_register(self, arr):
        """Page-lock a numpy array that will be handed to the upload entry points repeatedly (vg_host_register)."""
        self._chk(self.lib.vg_host_register(self.h, C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes)), "vg_host_register")

    def host_unregister(self, arr):
        self._chk(self.lib.vg_host_unregister(self.h, C.c_void_p(arr.ctypes.data)), "vg_host_unregister")

    def ba_set_fused_min_windows(self, n):
        """Batches of at least n windows take the fused linearise + accumulate kernel (0 = never; default 32): vg_ba_set_fused_min_windows."""
        self._chk(self.lib.vg_ba_set_fused_min_windows(self.h, int(n)), "vg_ba_set_fused_min_windows")

    def probe_clocks(self):
        """vg_probe_clocks: {ns per dependent FP64 FMA, clock64 ticks per us} with one wavefront / with every CU loaded."""
        out = (C.c_double * 4)()
        self.lib.vg_probe_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self._chk(self.lib.vg_probe_clocks(self.h, out), "vg_probe_clocks")
        return {"lone_wave": {"ns_per_dependent_fma": out[0], "clock64_ticks_per_us": out[1]},
                "all_cus_loaded": {"ns_per_dependent_fma": out[2], "clock64_ticks_per_us": out[3]}}

    def ba_launch_stats(self):
        mode, nl, ncap = C.c_int(), C.c_longlong(), C.c_longlong()
        self._chk(self.lib.vg_ba_launch_stats(self.h, C.byref(mode), C.byref(nl), C.byref(ncap)), "vg_ba_launch_stats")
        return {"mode": "graph" if mode.value == VG_LAUNCH_GRAPH else "direct", "graph_launches": nl.value, "graph_captures": ncap.value}

    # ---- large windows / landmark shards (include/vinsgpu.h "Large windows and landmark shards")
    def ba_set_marg_mode(self, mode):
        """VG_MARG_SQRT (0, default) / VG_MARG_EIGEN (1): form of the prior factor (include/vinsgpu.h)."""
        self._chk(self.lib.vg_ba_set_marg_mode(self.h, int(mode)), "vg_ba_set_marg_mode")

    def ba_set_imu_info_mode(self, mode):
        """VG_IMU_INFO_FACTOR (0, default) / VG_IMU_INFO_REFERENCE (1): form of the IMU factors' sqrt_info (include/vinsgpu.h)."""
        self.lib.vg_ba_set_imu_info_mode.argtypes = [C.c_void_p, C.c_int]
        self._chk(self.lib.vg_ba_set_imu_info_mode(self.h, int(mode)), "vg_ba_set_imu_info_mode")

    def rccl_unique_id(self):
        """128 bytes from ncclGetUniqueId (rank 0 calls this and broadcasts them)."""
        buf = C.create_string_buffer(128)
        self._chk(self.lib.vg_rccl_unique_id(buf), "vg_rccl_unique_id")
        return buf.raw

    def ba_rccl_init(self, nranks, rank, unique_id):
        """Collective: RCCL communicator on this handle's device + the C all-reduce hook (include/vinsgpu.h)."""
        self._chk(self.lib.vg_ba_rccl_init(self.h, int(nranks), int(rank), bytes(unique_id)), "vg_ba_rccl_init")

    def ba_rccl_finalize(self):
        self._chk(self.lib.vg_ba_rccl_finalize(self.h), "vg_ba_rccl_finalize")

    def ba_set_large_window(self, force=True):
        self._chk(self.lib.vg_ba_set_large_window(self.h, 1 if force else 0), "vg_ba_set_large_window")

    def ba_set_allreduce(self, fn):
        """fn(device_ptr: int, count: int, stream: int) must sum `count` doubles at device_ptr over all ranks in place,
        stream-ordered on the HIP stream `stream`; None removes the hook (single rank)."""
        if fn is None:
            self._hook = None
            self._chk(self.lib.vg_ba_set_allreduce(self.h, ALLREDUCE_FN(), None), "vg_ba_set_allreduce")
            return

        def tramp(user, buf, count, stream):
            try:
                fn(int(buf or 0), int(count), int(stream or 0))
                return 0
            except Exception as exc:                      # an exception must not unwind through the C frame
                self._hook_error = exc
                return 1
        self._hook = ALLREDUCE_FN(tramp)                  # keep the trampoline alive as long as the handle uses it
        self._chk(self.lib.vg_ba_set_allreduce(self.h, self._hook, None), "vg_ba_set_allreduce")

    def ba_reduce_layout(self):
        a, b = C.c_size_t(), C.c_size_t()
        self._chk(self.lib.vg_ba_reduce_layout(self.h, C.byref(a), C.byref(b)), "vg_ba_reduce_layout")
        return int(a.value), int(b.value)

    def ba_run_timed(self):
        """Synchronous run; returns (solve_kernel_ms, marg_kernel_ms) from HIP events on the launch stream."""
        a, b = C.c_float(), C.c_float()
        self._chk(self.lib.vg_ba_batch_run_timed(self.h, C.byref(a), C.byref(b)), "vg_ba_batch_run_timed")
        return float(a.value), float(b.value)

    KERNEL_CLASSES = ("ba_prologue_kernel", "ba_linearize_imu_kernel+ba_linearize_proj_kernel", "ba_accumulate_kernel", "ba_solve_kernel",
                      "ba_final_kernel", "ba_marg_kernel", "ba_big_schur_kernel", "ba_solve_big_kernel", "ba_big_step_kernel")

    def kernel_classes(self):
        """Names of the launch classes of the uploaded batch (VG_BA_KERNEL_*): with the fused projection kernel the accumulate class
        IS ba_linacc_proj_kernel (IMU + prior + projection factors: linearise + accumulate) and the linearize class is left with the
        cost-only pass of the last candidate."""
        k = list(self.KERNEL_CLASSES)
        mode = self.lib.vg_ba_batch_is_fused(self.h)
        if mode == 1:
            k[2] = "ba_linacc_proj_kernel"
        return k

    def ba_run_profiled(self):
        """Synchronous run with a HIP event after every launch: {kernel: (summed ms, launches)}."""
        ms = (C.c_float * len(self.KERNEL_CLASSES))()
        n = (C.c_int * len(self.KERNEL_CLASSES))()
        self._chk(self.lib.vg_ba_batch_run_profiled(self.h, ms, n), "vg_ba_batch_run_profiled")
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(self.kernel_classes())}

    def ba_info(self):
        fl, bi, bo, lds = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        self._chk(self.lib.vg_ba_batch_info(self.h, C.byref(fl), C.byref(bi), C.byref(bo), C.byref(lds)), "vg_ba_batch_info")
        fs, fm = C.c_double(), C.c_double()
        self._chk(self.lib.vg_ba_batch_flops(self.h, C.byref(fs), C.byref(fm)), "vg_ba_batch_flops")
        fk = (C.c_double * len(self.KERNEL_CLASSES))()
        self._chk(self.lib.vg_ba_batch_flops_by_kernel(self.h, fk), "vg_ba_batch_flops_by_kernel")
        return dict(flops=fl.value, flops_solve=fs.value, flops_marg=fm.value, bytes_in=bi.value, bytes_out=bo.value,
                    lds_bytes=lds.value, flops_by_kernel={k: float(fk[i]) for i, k in enumerate(self.kernel_classes())})

    def ba_prepare_download(self):
        """Allocate the host output buffers of the uploaded batch once (reused by every ba_download_raw())."""
        n = len(self._packed)
        outs = [_Out(p.K, p.L, p.has_relo, self._margin[i] != VG_MARGIN_NONE) for i, p in enumerate(self._packed)]
        st = (C.POINTER(State) * n)(*[C.pointer(o.state) for o in outs])
        pri = (C.POINTER(Prior) * n)(*[C.pointer(o.prior) if o.prior is not None else None for o in outs])
        sm = (Summary * n)()
        self._dl = (outs, st, pri, sm, list(self._packed))
        return self._dl

    def ba_download_raw(self):
        """vg_ba_batch_download into the buffers of ba_prepare_download(); returns the C-ABI status."""
        outs, st, pri, sm, _ = self._dl
        t0 = time.perf_counter()
        rc = self.lib.vg_ba_batch_download(self.h, len(outs), st, sm, pri)
        self.last_download_call_ms = (time.perf_counter() - t0) * 1e3      # the C-ABI call alone (sync + D2H + unpack)
        return rc

    def ba_download(self, allow_numeric_failure=False):
        outs, st, pri, sm, packed = self.ba_prepare_download()
        n = len(outs)
        rc = self.ba_download_raw()
        if rc != VG_OK and not (allow_numeric_failure and rc == -4):
            self._chk(rc, "vg_ba_batch_download")
        return ([o.state_dict(p.has_relo) for o, p in zip(outs, packed)],
                [summary_dict(sm[i]) for i in range(n)],
                [o.prior_dict() for o in outs])

    def ba_download_state_raw(self, dl=None):
        """vg_ba_batch_download_state into the buffers of ba_prepare_download(): returns while the marginalization runs."""
        outs, st, pri, sm, _ = dl if dl is not None else self._dl
        t0 = time.perf_counter()
        rc = self.lib.vg_ba_batch_download_state(self.h, len(outs), st, sm)
        self.last_download_call_ms = (time.perf_counter() - t0) * 1e3
        return rc

    def ba_download_prior_raw(self):
        outs, st, pri, sm, _ = self._dl
        return self.lib.vg_ba_batch_download_prior(self.h, len(outs), pri)

    def ba_optimize_split(self, prob, margin_flag=VG_MARGIN_NONE):
        """vg_ba_optimize in two parts: returns (state, summary, prior, ms until the states were on the host)."""
        t0 = time.perf_counter()
        self.ba_upload([prob], [margin_flag])
        self.ba_run_async()
        outs, st, pri, sm, packed = self.ba_prepare_download()
        self._chk(self.ba_download_state_raw(), "vg_ba_batch_download_state")
        t_state = (time.perf_counter() - t0) * 1e3
        state, summ = outs[0].state_dict(packed[0].has_relo), summary_dict(sm[0])
        self._chk(self.ba_download_prior_raw(), "vg_ba_batch_download_prior")
        return state, summ, outs[0].prior_dict(), t_state

    def ba_optimize(self, prob, margin_flag=VG_MARGIN_NONE):
        """One Estimator::optimization(): returns (state, summary, new_prior)."""
        self.ba_upload([prob], [margin_flag])
        self.ba_run_async()
        st, sm, pr = self.ba_download()
        return st[0], sm[0], pr[0]

    # ---- windows that stay on the device (include/vinsgpu.h vg_ba_seq_*)
    def ba_reserve(self, max_landmarks=0, max_factors=0, max_obs=0, max_prior_n=0):
        self._chk(self.lib.vg_ba_reserve(self.h, int(max_landmarks), int(max_factors), int(max_obs), int(max_prior_n)), "vg_ba_reserve")

    def seq_begin(self, windows, tracks, max_features=512, max_new_obs=512, max_landmarks=0, max_factors=0, init_depth=5.0,
                  min_parallax=10.0 / 460.0):
        """windows: prob dicts (states, imu, prior, options; landmark tables ignored); tracks[w]: dict(id, start, nobs, depth, obs
        [sum nobs x 8: x y z u v vx vy cur_td], optional solve_flag)."""
        n = len(windows)
        self._packed = [PackedProblem(p) for p in windows]
        arr = (C.POINTER(Problem) * n)(*[C.pointer(p.struct) for p in self._packed])
        self._margin = np.zeros(n, np.int32)
        keep, ts = [], (Tracks * n)()
        for w, t in enumerate(tracks):
            i4 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
            a = dict(id=i4(t['id']), start=i4(t['start']), nobs=i4(t['nobs']), depth=np.ascontiguousarray(t['depth'], dtype=np.float64),
                     obs=np.ascontiguousarray(t['obs'], dtype=np.float64).reshape(-1, 8),
                     flag=i4(t['solve_flag']) if t.get('solve_flag') is not None else None)
            keep.append(a)
            ts[w].n_features = len(a['id'])
            ts[w].feature_id, ts[w].start_frame, ts[w].n_obs = _ip(a['id']), _ip(a['start']), _ip(a['nobs'])
            ts[w].solve_flag = _ip(a['flag']) if a['flag'] is not None else None
            ts[w].depth, ts[w].obs = _dp(a['depth']), _dp(a['obs'])
        tp = (C.POINTER(Tracks) * n)(*[C.pointer(ts[w]) for w in range(n)])
        cfg = SeqConfig(int(max_features), int(max_new_obs), int(max_landmarks), int(max_factors), float(init_depth), float(min_parallax))
        self._chk(self.lib.vg_ba_seq_begin(self.h, n, C.byref(cfg), arr, tp), "vg_ba_seq_begin")
        self._seq_n = n
        for p in self._packed:                            # size of the inverse-depth slab a state download fills
            p.L = (int(max_landmarks or max_features) + 15) // 16 * 16

    @staticmethod
    def seq_pack_frames(frames):
        """The vg_ba_frame* array of one step, built once (what a native caller holds): frames[w] = dict(pose (7,), sb (9,), imu_new,
        imu_merged (pre-integration dicts; imu_merged None unless the frame before was dropped as a non-keyframe), ids (n,),
        obs (n, 7) [x y z u v vx vy])."""
        n = len(frames)
        fs, keep = (Frame * n)(), []
        for w, f in enumerate(frames):
            ids = np.ascontiguousarray(f['ids'], dtype=np.int32)
            obs = np.ascontiguousarray(f['obs'], dtype=np.float64).reshape(-1, 7)
            inew = imu_struct(f['imu_new'])
            imrg = imu_struct(f['imu_merged']) if f.get('imu_merged') is not None else None
            keep.append((ids, obs, inew, imrg))
            fs[w].pose[:] = list(np.asarray(f['pose'], float)); fs[w].speedbias[:] = list(np.asarray(f['sb'], float))
            fs[w].imu_new = C.pointer(inew)
            fs[w].imu_merged = C.pointer(imrg) if imrg is not None else None
            fs[w].n_obs = len(ids)
            fs[w].feature_id, fs[w].obs = _ip(ids), _dp(obs)
        fp = (C.POINTER(Frame) * n)(*[C.pointer(fs[w]) for w in range(n)])
        return (n, fp, fs, keep)

    def seq_step(self, frames):
        """One frame for every window (vg_ba_seq_step_async); frames: a list of dicts or the result of seq_pack_frames."""
        packed = frames if isinstance(frames, tuple) else self.seq_pack_frames(frames)
        self._chk(self.lib.vg_ba_seq_step_async(self.h, packed[0], packed[1]), "vg_ba_seq_step_async")

    def seq_states(self, allow_numeric_failure=False):
        """States of the windows as the last step solved them (before the slide) and the summaries."""
        outs, st, pri, sm, packed = self.ba_prepare_download()
        rc = self.ba_download_state_raw()
        if rc != VG_OK and not (allow_numeric_failure and rc == -4):
            self._chk(rc, "vg_ba_batch_download_state")
        return [o.state_dict(False) for o in outs], [summary_dict(sm[i]) for i in range(len(outs))]

    def seq_priors(self):
        """parity tap: the priors the last step's marginalization produced (None where the old one stays)."""
        outs, st, pri, sm, packed = self._dl
        self._chk(self.ba_download_prior_raw(), "vg_ba_batch_download_prior")
        return [o.prior_dict() for o in outs]

    def seq_info(self):
        a = np.zeros((self._seq_n, len(SEQ_INFO)), np.int32)
        self._chk(self.lib.vg_ba_seq_info(self.h, self._seq_n, _ip(a)), "vg_ba_seq_info")
        return [dict(zip(SEQ_INFO, map(int, r))) for r in a]

    def seq_tracks(self, w, K, cap=1024):
        n = C.c_int()
        ids, st, nb, fl = (np.zeros(cap, np.int32) for _ in range(4))
        dep, obs = np.zeros(cap), np.zeros((cap, K, 8))
        self._chk(self.lib.vg_ba_seq_get_tracks(self.h, int(w), cap, C.byref(n), _ip(ids), _ip(st), _ip(nb), _ip(fl), _dp(dep), _dp(obs)),
                  "vg_ba_seq_get_tracks")
        m = n.value
        return dict(id=ids[:m].copy(), start=st[:m].copy(), nobs=nb[:m].copy(), solve_flag=fl[:m].copy(), depth=dep[:m].copy(), obs=obs[:m].copy())

    def seq_export(self, w, K):
        """The window of slot w between two frames, as vg_ba_seq_begin takes it: (prob-dict fields, tracks dict)."""
        o = _Out(K, 0, False, True)
        imu = (ImuPreint * (K - 1))()
        self._chk(self.lib.vg_ba_seq_export(self.h, int(w), _dp(o.pose), _dp(o.sb), _dp(o.ex), _dp(o.td), imu, C.byref(o.prior)), "vg_ba_seq_export")
        recs = []
        for q in imu:
            recs.append(None if not q.valid else dict(
                sum_dt=q.sum_dt, delta_p=np.array(q.delta_p), delta_q=np.array(q.delta_q), delta_v=np.array(q.delta_v), lin_ba=np.array(q.linearized_ba),
                lin_bg=np.array(q.linearized_bg), jacobian=np.array(q.jacobian).reshape(15, 15), covariance=np.array(q.covariance).reshape(15, 15)))
        t = self.seq_tracks(w, K)
        rows = np.concatenate([t['obs'][f, :n][:, [0, 1, 7, 2, 3, 4, 5, 6]] for f, n in enumerate(t['nobs'])]) if len(t['id']) else np.zeros((0, 8))
        tracks = dict(id=t['id'], start=t['start'], nobs=t['nobs'], depth=t['depth'], solve_flag=t['solve_flag'], obs=rows)
        return dict(pose=o.pose.copy(), sb=o.sb.copy(), ex=o.ex.copy(), td=float(o.td[0]), imu=recs, prior=o.prior_dict()), tracks

    def seq_import(self, w, prob, tracks):
        """Replace what slot w of the running sequence holds (vg_ba_seq_import); prob / tracks as for seq_begin."""
        p = PackedProblem(prob)
        i4 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        a = dict(id=i4(tracks['id']), start=i4(tracks['start']), nobs=i4(tracks['nobs']), depth=np.ascontiguousarray(tracks['depth'], dtype=np.float64),
                 obs=np.ascontiguousarray(tracks['obs'], dtype=np.float64).reshape(-1, 8),
                 flag=i4(tracks['solve_flag']) if tracks.get('solve_flag') is not None else None)
        t = Tracks()
        t.n_features = len(a['id'])
        t.feature_id, t.start_frame, t.n_obs = _ip(a['id']), _ip(a['start']), _ip(a['nobs'])
        t.solve_flag = _ip(a['flag']) if a['flag'] is not None else None
        t.depth, t.obs = _dp(a['depth']), _dp(a['obs'])
        self._chk(self.lib.vg_ba_seq_import(self.h, int(w), C.byref(p.struct), C.byref(t)), "vg_ba_seq_import")

    def seq_end(self):
        self._chk(self.lib.vg_ba_seq_end(self.h), "vg_ba_seq_end")

    def imu_preintegrate(self, intervals, biases, noise):
        """Batched IntegrationBase (integration_base.h:30-158).  intervals[k] = [(dt, acc, gyr), ...] with entry 0 =
        (0, acc_0, gyr_0) the first measurement (the layout of synth.preintegrate); biases[k] = (ba, bg);
        noise = (acc_n, gyr_n, acc_w, gyr_w).  Returns the list of pre-integration dicts vg_ba_problem consumes."""
        n = len(intervals)
        off = np.zeros(n + 1, np.int32)
        rows, first = [], np.zeros((n, 6))
        for k, iv in enumerate(intervals):
            first[k, :3], first[k, 3:] = iv[0][1], iv[0][2]
            for dt, a, g in iv[1:]:
                rows.append([dt, a[0], a[1], a[2], g[0], g[1], g[2]])
            off[k + 1] = len(rows)
        smp = np.ascontiguousarray(rows if rows else np.zeros((1, 7)), dtype=np.float64)
        bias = np.ascontiguousarray([np.concatenate([np.asarray(b[0], float), np.asarray(b[1], float)]) for b in biases])
        nz = np.ascontiguousarray(noise, dtype=np.float64)
        out = (ImuPreint * n)()
        self._chk(self.lib.vg_imu_preintegrate(self.h, n, _ip(off), _dp(smp), _dp(np.ascontiguousarray(first)), _dp(bias), _dp(nz), out),
                  "vg_imu_preintegrate")
        res = []
        for q in out:
            res.append(dict(sum_dt=q.sum_dt, delta_p=np.array(q.delta_p), delta_q=np.array(q.delta_q), delta_v=np.array(q.delta_v),
                            lin_ba=np.array(q.linearized_ba), lin_bg=np.array(q.linearized_bg),
                            jacobian=np.array(q.jacobian).reshape(15, 15), covariance=np.array(q.covariance).reshape(15, 15)))
        return res

    def triangulate(self, Ps, Rs, tic, ric, start, nobs, obs_off, points, init_depth=5.0):
        """FeatureManager::triangulate (feature_manager.cpp:202-257); returns the depth per landmark."""
        Ps = np.ascontiguousarray(Ps, np.float64); Rs = np.ascontiguousarray(Rs, np.float64)
        K, Ln = len(Ps), len(start)
        st = np.ascontiguousarray(start, np.int32); nb = np.ascontiguousarray(nobs, np.int32); oo = np.ascontiguousarray(obs_off, np.int32)
        pts = np.ascontiguousarray(points, np.float64)
        out = np.zeros(max(Ln, 1))
        self._chk(self.lib.vg_triangulate(self.h, K, _dp(Ps), _dp(Rs), _dp(np.ascontiguousarray(tic, np.float64)),
                                          _dp(np.ascontiguousarray(ric, np.float64)), Ln, _ip(st), _ip(nb), _ip(oo), _dp(pts),
                                          float(init_depth), _dp(out)), "vg_triangulate")
        return out[:Ln]

    def ba_eval_factors(self, prob):
        p = PackedProblem(prob)
        F = int((p.keep['lm_nobs'] - 1).sum()) + (len(prob['relo']['match']) if prob.get('relo') else 0)
        K = p.K
        n = int(prob['prior']['n']) if prob.get('prior') is not None else 0
        pr, pJ = np.zeros((F, 2)), np.zeros((F, 2, 20))
        ir, iJ = np.zeros((K - 1, 15)), np.zeros((K - 1, 15, 30))
        qr = np.zeros(max(n, 1))
        self._chk(self.lib.vg_ba_eval_factors(self.h, C.byref(p.struct), _dp(pr), _dp(pJ), _dp(ir), _dp(iJ), _dp(qr)),
                  "vg_ba_eval_factors")
        return dict(proj_r=pr, proj_J=pJ, imu_r=ir, imu_J=iJ, prior_r=qr[:n])
