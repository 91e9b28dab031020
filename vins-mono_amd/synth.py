"""EuRoC-shaped synthetic sliding windows (host side, NumPy) for parity tests and bench.py.

Scene follows the reference's simulator (data_generator/src/data_generator.cpp:81-111: Lissajous
position, sinusoidal roll/pitch) with the EuRoC camera/IMU calibration and noise parameters
(config/euroc/euroc_config.yaml:13-63).  The IMU constants of each frame pair are produced by the
mid-point pre-integration of IntegrationBase::push_back (factor/integration_base.h:30-158), here
`preintegrate()` (host code: the step upstream of the BA hot path).

A problem ("window") is a plain dict of NumPy arrays; `ba.py` packs it for the C-ABI:
  pose (K,7) [px py pz qx qy qz qw] | sb (K,9) [v ba bg] | ex (7,) | td | inv_depth (L,)
  lm_start (L,), lm_nobs (L,), obs_off (L,), obs (sum n_l, 7) [x y u v vx vy cur_td]
  imu: list of K-1 dicts {sum_dt, delta_p, delta_q, delta_v, lin_ba, lin_bg, jacobian, covariance}
  prior: None | {n, blocks [(kind, idx)], J0 (n,n), r0 (n,), x0 [arrays]}
  flags: estimate_extrinsic, estimate_td, max_iters, focal, tr, row, g_norm
"""
import numpy as np

EUROC = dict(
    fx=461.6, fy=460.3, cx=363.0, cy=248.1, width=752, height=480,
    ric=np.array([[0.0148655429818, -0.999880929698, 0.00414029679422],
                  [0.999557249008, 0.0149672133247, 0.025715529948],
                  [-0.0257744366974, 0.00375618835797, 0.999660727178]]),
    tic=np.array([-0.0216401454975, -0.064676986768, 0.00981073058949]),
    acc_n=0.08, gyr_n=0.004, acc_w=0.00004, gyr_w=2.0e-6, g_norm=9.81007,
    focal=460.0, max_iters=8,
)


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def _qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def _q2R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _R2q(m):
    w = np.sqrt(max(1e-300, 1 + m[0, 0] + m[1, 1] + m[2, 2])) / 2
    if w > 1e-3:
        return np.array([(m[2, 1] - m[1, 2]) / (4 * w), (m[0, 2] - m[2, 0]) / (4 * w), (m[1, 0] - m[0, 1]) / (4 * w), w])
    i = int(np.argmax(np.diag(m)))
    j, k = (i + 1) % 3, (i + 2) % 3
    t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
    q = np.zeros(4)
    q[i] = 0.5 * t
    t = 0.5 / t
    q[3] = (m[k, j] - m[j, k]) * t
    q[j] = (m[j, i] + m[i, j]) * t
    q[k] = (m[k, i] + m[i, k]) * t
    return q


def preintegrate(samples, ba, bg, acc_n, gyr_n, acc_w, gyr_w):
    """Mid-point IMU pre-integration of one frame interval (integration_base.h:54-158).
    samples = [(dt, acc, gyr), ...] with samples[0] = (0, acc_0, gyr_0) the first measurement."""
    ba, bg = np.asarray(ba, float), np.asarray(bg, float)
    acc_0, gyr_0 = np.asarray(samples[0][1], float), np.asarray(samples[0][2], float)
    Jm, P = np.eye(15), np.zeros((15, 15))
    dp, dv, dq, sum_dt = np.zeros(3), np.zeros(3), np.array([0, 0, 0, 1.0]), 0.0
    nz = np.zeros(18)
    nz[0:3] = nz[6:9] = acc_n ** 2
    nz[3:6] = nz[9:12] = gyr_n ** 2
    nz[12:15] = acc_w ** 2
    nz[15:18] = gyr_w ** 2
    N, I3 = np.diag(nz), np.eye(3)
    for dt, acc_1, gyr_1 in samples[1:]:
        acc_1, gyr_1 = np.asarray(acc_1, float), np.asarray(gyr_1, float)
        Rq = _q2R(dq)
        w = 0.5 * (gyr_0 + gyr_1) - bg
        rq = _qmul(dq, np.array([w[0] * dt / 2, w[1] * dt / 2, w[2] * dt / 2, 1.0]))
        Rr = _q2R(rq)
        a0, a1 = acc_0 - ba, acc_1 - ba
        un_acc = 0.5 * (Rq @ a0 + Rr @ a1)
        Rw, Ra0, Ra1 = _skew(w), _skew(a0), _skew(a1)
        F = np.zeros((15, 15))
        F[0:3, 0:3] = I3
        F[0:3, 3:6] = -0.25 * Rq @ Ra0 * dt * dt - 0.25 * Rr @ Ra1 @ (I3 - Rw * dt) * dt * dt
        F[0:3, 6:9] = I3 * dt
        F[0:3, 9:12] = -0.25 * (Rq + Rr) * dt * dt
        F[0:3, 12:15] = 0.25 * Rr @ Ra1 * dt * dt * dt
        F[3:6, 3:6] = I3 - Rw * dt
        F[3:6, 12:15] = -I3 * dt
        F[6:9, 3:6] = -0.5 * Rq @ Ra0 * dt - 0.5 * Rr @ Ra1 @ (I3 - Rw * dt) * dt
        F[6:9, 6:9] = I3
        F[6:9, 9:12] = -0.5 * (Rq + Rr) * dt
        F[6:9, 12:15] = 0.5 * Rr @ Ra1 * dt * dt
        F[9:12, 9:12] = I3
        F[12:15, 12:15] = I3
        V = np.zeros((15, 18))
        V[0:3, 0:3] = 0.25 * Rq * dt * dt
        V[0:3, 3:6] = -0.125 * Rr @ Ra1 * dt * dt * dt
        V[0:3, 6:9] = 0.25 * Rr * dt * dt
        V[0:3, 9:12] = V[0:3, 3:6]
        V[3:6, 3:6] = 0.5 * I3 * dt
        V[3:6, 9:12] = 0.5 * I3 * dt
        V[6:9, 0:3] = 0.5 * Rq * dt
        V[6:9, 3:6] = -0.25 * Rr @ Ra1 * dt * dt
        V[6:9, 6:9] = 0.5 * Rr * dt
        V[6:9, 9:12] = V[6:9, 3:6]
        V[9:12, 12:15] = I3 * dt
        V[12:15, 15:18] = I3 * dt
        Jm = F @ Jm
        P = F @ P @ F.T + V @ N @ V.T
        dp = dp + dv * dt + 0.5 * un_acc * dt * dt
        dv = dv + un_acc * dt
        dq = rq / np.linalg.norm(rq)
        sum_dt += dt
        acc_0, gyr_0 = acc_1, gyr_1
    return dict(sum_dt=sum_dt, delta_p=dp, delta_q=dq, delta_v=dv, lin_ba=ba.copy(), lin_bg=bg.copy(),
                jacobian=Jm, covariance=P)


class SyntheticSequence:
    """Ground-truth trajectory + landmark tracks over `n_frames` key frames at 10 Hz with 200 Hz IMU."""

    def __init__(self, seed, n_frames=12, K=11, L=150, period=10.0, imu_per_frame=20, frame_dt=0.1,
                 estimate_extrinsic=0, estimate_td=0, cfg=None):
        self.cfg = dict(EUROC if cfg is None else cfg)
        self.rng = np.random.default_rng(seed)
        self.K, self.L, self.n_frames = K, L, n_frames
        self.period, self.frame_dt, self.imu_per_frame = period, frame_dt, imu_per_frame
        self.estimate_extrinsic, self.estimate_td = estimate_extrinsic, estimate_td
        self.t0 = float(self.rng.uniform(0.0, period))
        self.ba_true = self.rng.normal(0, 0.02, 3)
        self.bg_true = np.array([0.02, 0.03, 0.04])
        self.ba_lin = self.ba_true + self.rng.normal(0, 0.01, 3)
        self.bg_lin = self.bg_true + self.rng.normal(0, 0.002, 3)
        self.td_true = 0.0
        self.times = self.t0 + frame_dt * np.arange(n_frames)
        self.P = np.array([self.position(t) for t in self.times])
        self.Rm = np.array([self.rotation(t) for t in self.times])
        self.V = np.array([self.velocity(t) for t in self.times])
        self._make_imu()
        self._make_landmarks()

    # ---- trajectory (data_generator.cpp:81-111 with MAX_BOX=10)
    def position(self, t):
        T = self.period
        return np.array([5 + 5 * np.cos(t / T * np.pi), 5 + 5 * np.cos(t / T * np.pi * 2),
                         5 + 5 * np.cos(t / T * np.pi * 4)])

    def velocity(self, t):
        T = self.period
        return np.array([-5 * np.pi / T * np.sin(t / T * np.pi), -5 * 2 * np.pi / T * np.sin(t / T * np.pi * 2),
                         -5 * 4 * np.pi / T * np.sin(t / T * np.pi * 4)])

    def acceleration(self, t):
        T = self.period
        return np.array([-5 * (np.pi / T) ** 2 * np.cos(t / T * np.pi),
                         -5 * (2 * np.pi / T) ** 2 * np.cos(t / T * np.pi * 2),
                         -5 * (4 * np.pi / T) ** 2 * np.cos(t / T * np.pi * 4)])

    def _angles(self, t):
        T = self.period
        s, c = np.sin(t / T * np.pi * 2), np.cos(t / T * np.pi * 2) * 2 * np.pi / T
        return np.radians(30) * s, np.radians(40) * s, np.radians(30) * c, np.radians(40) * c

    def rotation(self, t):
        a, b, _, _ = self._angles(t)
        Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        return Rx @ Ry

    def angular_velocity(self, t):
        _, b, da, db = self._angles(t)
        Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        return Ry.T @ np.array([da, 0, 0]) + np.array([0, db, 0])

    def _imu_sample(self, t):
        G = np.array([0, 0, self.cfg['g_norm']])
        acc = self.rotation(t).T @ (self.acceleration(t) + G) + self.ba_true
        gyr = self.angular_velocity(t) + self.bg_true
        return acc, gyr

    def _make_imu(self):
        c = self.cfg
        self.imu = []
        h = self.frame_dt / self.imu_per_frame
        for k in range(self.n_frames - 1):
            t = self.times[k]
            samples = [(0.0,) + self._imu_sample(t)]
            for s in range(1, self.imu_per_frame + 1):
                samples.append((h,) + self._imu_sample(t + s * h))
            self.imu.append(preintegrate(samples, self.ba_lin, self.bg_lin, c['acc_n'], c['gyr_n'], c['acc_w'], c['gyr_w']))

    def _make_landmarks(self):
        """Tracks: global first frame f0 ~ U{0..K-4}, length n ~ U{2..n_frames-f0}; the world point is
        drawn in the frustum of the first observing camera (depth 2-12 m)."""
        rng, c = self.rng, self.cfg
        self.lm = []
        for _ in range(self.L):
            f0 = int(rng.integers(0, self.K - 3))
            n = int(rng.integers(2, self.n_frames - f0 + 1))
            x, y, d = rng.uniform(-0.5, 0.5), rng.uniform(-0.35, 0.35), rng.uniform(2.0, 12.0)
            pc = np.array([x, y, 1.0]) * d
            Xw = self.Rm[f0] @ (c['ric'] @ pc + c['tic']) + self.P[f0]
            obs = []
            for f in range(f0, f0 + n):
                p = c['ric'].T @ (self.Rm[f].T @ (Xw - self.P[f]) - c['tic'])
                if p[2] < 0.2:
                    break
                obs.append(p[:2] / p[2] + rng.normal(0, 0.3 / 460.0, 2))
            if len(obs) < 2:
                continue
            self.lm.append(dict(f0=f0, Xw=Xw, obs=np.array(obs)))

    # ---- window assembly
    def _landmark_tables(self, w0):
        K, c = self.K, self.cfg
        start, nobs, off, obs, depth = [], [], [], [], []
        for lm in self.lm:
            lo, hi = max(lm['f0'], w0), min(lm['f0'] + len(lm['obs']) - 1, w0 + K - 1)
            n = hi - lo + 1
            s = lo - w0
            if n < 2 or not (s < K - 3):      # used_num >= 2 && start_frame < WINDOW_SIZE - 2
                continue
            start.append(s)
            nobs.append(n)
            off.append(len(obs))
            prev = None
            for f in range(lo, hi + 1):
                xy = lm['obs'][f - lm['f0']]
                u, v = c['fx'] * xy[0] + c['cx'], c['fy'] * xy[1] + c['cy']
                vel = (xy - prev) / self.frame_dt if prev is not None else np.zeros(2)
                prev = xy
                obs.append([xy[0], xy[1], u, v, vel[0], vel[1], 0.0])
            pc = c['ric'].T @ (self.Rm[lo].T @ (lm['Xw'] - self.P[lo]) - c['tic'])
            depth.append(pc[2])
        return (np.array(start, np.int32), np.array(nobs, np.int32), np.array(off, np.int32),
                np.array(obs, float), np.array(depth, float))

    def _noisy_pose(self, f):
        rng = self.rng
        th = rng.normal(0, np.radians(0.5), 3)
        dq = np.array([th[0] / 2, th[1] / 2, th[2] / 2, 1.0])
        q = _qmul(_R2q(self.Rm[f]), dq)
        q /= np.linalg.norm(q)
        return np.concatenate([self.P[f] + rng.normal(0, 0.05, 3), q])

    def _noisy_sb(self, f):
        return np.concatenate([self.V[f] + self.rng.normal(0, 0.05, 3), self.ba_lin, self.bg_lin])

    def _base(self):
        c = self.cfg
        ex = np.concatenate([c['tic'], _R2q(c['ric'])])
        return dict(ex=ex, td=0.0, estimate_extrinsic=self.estimate_extrinsic, estimate_td=self.estimate_td,
                    max_iters=c['max_iters'], focal=c['focal'], tr=0.0, row=float(c['height']),
                    g_norm=c['g_norm'], prior=None, relo=None)

    def window(self, w0=0):
        """A fresh window starting at global frame w0: truth (+) noise, no prior."""
        K = self.K
        prob = self._base()
        prob['pose'] = np.array([self._noisy_pose(w0 + i) for i in range(K)])
        prob['sb'] = np.array([self._noisy_sb(w0 + i) for i in range(K)])
        s, n, o, obs, depth = self._landmark_tables(w0)
        prob.update(lm_start=s, lm_nobs=n, obs_off=o, obs=obs)
        prob['inv_depth'] = 1.0 / (depth * (1.0 + self.rng.normal(0, 0.1, depth.shape[0])))
        prob['imu'] = [dict((k, np.copy(v)) for k, v in self.imu[w0 + i].items()) for i in range(K - 1)]
        if self.estimate_extrinsic:
            prob['ex'] = prob['ex'].copy()
            prob['ex'][:3] += self.rng.normal(0, 0.01, 3)
        if self.estimate_td:
            prob['td'] = 0.002
        return prob

    @staticmethod
    def anchor_prior(prob, sigma_p=0.02, sigma_q=0.01, sigma_v=0.05, sigma_b=0.02):
        """A prior factor on the oldest frame (pose 0, speed-bias 0 and, if estimated, the extrinsic pose / td) in the form
        MarginalizationFactor consumes: r = r0 + J0 (x (-) x0) with J0 = diag(1 / sigma), r0 = 0, x0 = the window's own
        initial values.  Stands in for the marginalization result where none can be produced (the enlarged window of
        BASELINE configs[4] has no predecessor): it fixes the gauge like the reference's prior does."""
        blocks = [(0, 0), (1, 0)]                     # (VG_BLK_POSE, 0), (VG_BLK_SPEEDBIAS, 0)
        w = [1.0 / sigma_p] * 3 + [1.0 / sigma_q] * 3 + [1.0 / sigma_v] * 3 + [1.0 / sigma_b] * 6
        x0 = [np.array(prob['pose'][0], float), np.array(prob['sb'][0], float)]
        if prob['estimate_extrinsic']:
            blocks.append((2, 0)); w += [1.0 / sigma_p] * 3 + [1.0 / sigma_q] * 3; x0.append(np.array(prob['ex'], float))
        if prob['estimate_td']:
            blocks.append((3, 0)); w += [1.0 / 0.01]; x0.append(np.array([float(prob['td'])]))
        n = len(w)
        out = dict(prob)
        out['prior'] = dict(n=n, blocks=blocks, J0=np.diag(np.array(w, float)), r0=np.zeros(n), x0=x0)
        return out

    def next_window(self, prev_state, prior, w0):
        """Window w0 (= previous + 1) after MARGIN_OLD: frames 0..K-2 carry the previous optimum,
        the newest frame is truth (+) noise, `prior` is the previous marginalization result."""
        K = self.K
        prob = self._base()
        prob['ex'] = prev_state['ex'].copy()
        prob['td'] = float(prev_state['td'])
        prob['pose'] = np.vstack([prev_state['pose'][1:K], self._noisy_pose(w0 + K - 1)[None]])
        prob['sb'] = np.vstack([prev_state['sb'][1:K], self._noisy_sb(w0 + K - 1)[None]])
        s, n, o, obs, depth = self._landmark_tables(w0)
        prob.update(lm_start=s, lm_nobs=n, obs_off=o, obs=obs)
        prob['inv_depth'] = 1.0 / (depth * (1.0 + self.rng.normal(0, 0.1, depth.shape[0])))
        prob['imu'] = [dict((k, np.copy(v)) for k, v in self.imu[w0 + i].items()) for i in range(K - 1)]
        prob['prior'] = prior
        return prob


class FrameSource:
    """Per-frame inputs of a sliding-window estimator (device-resident sequences, vg_ba_seq_*).
    Frames of a synth.SyntheticSequence the way the estimator node receives them: the `image` map of a frame (feature id =
    landmark number, rows [x y 1 u v vx vy]), the IMU samples of an interval, noisy state guesses."""

    def __init__(self, seq, noise_seed=0):
        self.seq, self.rng = seq, np.random.default_rng(noise_seed)
        self.c = seq.cfg
        self.h = seq.frame_dt / seq.imu_per_frame

    def image(self, f):
        seq, c = self.seq, self.c
        ids, rows = [], []
        for lid, lm in enumerate(seq.lm):
            k = f - lm['f0']
            if 0 <= k < len(lm['obs']):
                xy = lm['obs'][k]
                prev = lm['obs'][k - 1] if k > 0 else xy
                vel = (xy - prev) / seq.frame_dt
                ids.append(lid)
                rows.append([xy[0], xy[1], 1.0, c['fx'] * xy[0] + c['cx'], c['fy'] * xy[1] + c['cy'], vel[0], vel[1]])
        return np.array(ids, np.int32), np.array(rows, float).reshape(-1, 7)

    def samples(self, f):
        """(dt, acc, gyr) of the interval f -> f + 1; entry 0 = the first measurement (dt 0)."""
        seq = self.seq
        t = seq.times[f]
        out = [(0.0,) + seq._imu_sample(t)]
        for s in range(1, seq.imu_per_frame + 1):
            out.append((self.h,) + seq._imu_sample(t + s * self.h))
        return out

    def preintegrate(self, samples, ba, bg):
        c = self.c
        return preintegrate(samples, ba, bg, c['acc_n'], c['gyr_n'], c['acc_w'], c['gyr_w'])

    def guess(self, f):
        rng, seq = self.rng, self.seq
        th = rng.normal(0, np.radians(0.3), 3)
        q = _qmul(_R2q(seq.Rm[f]), np.array([th[0] / 2, th[1] / 2, th[2] / 2, 1.0]))
        q /= np.linalg.norm(q)
        return (np.concatenate([seq.P[f] + rng.normal(0, 0.03, 3), q]), np.concatenate([seq.V[f] + rng.normal(0, 0.03, 3), seq.ba_lin, seq.bg_lin]))

    def initial_window(self, K, g0=0):
        """The state between two frames: global frames g0 .. g0 + K - 2 are in the window, the newest slot is a copy of the
        frame before it (what slideWindow leaves), no prior yet, no depths yet."""
        seq = self.seq
        pose, sb = zip(*[self.guess(g0 + i) for i in range(K - 1)])
        pose, sb = list(pose) + [pose[-1]], list(sb) + [sb[-1]]
        smp = [self.samples(g0 + i) for i in range(K - 2)] + [None]
        imu = [self.preintegrate(s, seq.ba_lin, seq.bg_lin) for s in smp[:-1]] + [None]
        feats = {}
        for f in range(g0, g0 + K - 1):
            ids, rows = self.image(f)
            for fid, r in zip(ids, rows):
                ft = feats.setdefault(int(fid), dict(id=int(fid), start=f - g0, obs=[], depth=-1.0))
                ft['obs'].append(list(r) + [0.0])
        tracks = list(feats.values())                     # list order of f_manager.feature: by first frame, ids ascending inside a frame
        return dict(K=K, base=seq._base(), pose=np.array(pose), sb=np.array(sb), imu=imu, samples=smp, tracks=tracks)


def sequence_inputs(win):
    """(prob dict, tracks dict) of an initial window (FrameSource.initial_window) for vg_ba_seq_begin."""
    prob = dict(win['base'])
    prob.update(pose=win['pose'].copy(), sb=win['sb'].copy(), prior=None, relo=None, imu=[None if m is None else dict(m) for m in win['imu']],
                lm_start=np.zeros(0, np.int32), lm_nobs=np.zeros(0, np.int32), obs_off=np.zeros(0, np.int32), obs=np.zeros((0, 7)),
                inv_depth=np.zeros(0))
    t = win['tracks']
    tracks = dict(id=np.array([f['id'] for f in t], np.int32), start=np.array([f['start'] for f in t], np.int32),
                  nobs=np.array([len(f['obs']) for f in t], np.int32), depth=np.array([f['depth'] for f in t], float),
                  solve_flag=np.array([f.get('flag', 0) for f in t], np.int32),
                  obs=np.array([r for f in t for r in f['obs']], float).reshape(-1, 8))
    return prob, tracks


# ----------------------------------------------------------------------------- front-end frames
def synth_frame(seed, width=752, height=480):
    """EuRoC-shaped textured frame (SURVEY.md 8(d) configs[1]): value noise on a (width/8 x height/8) lattice,
    bilinear x8 upsample, plus uniform +-8 fine noise, clamped to u8."""
    rng = np.random.default_rng(seed)
    gw, gh = width // 8 + 2, height // 8 + 2
    lat = rng.uniform(0, 255, (gh, gw))
    ys, xs = (np.arange(height) + 0.5) / 8.0, (np.arange(width) + 0.5) / 8.0
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    img = (lat[y0][:, x0] * (1 - fy) * (1 - fx) + lat[y0][:, x0 + 1] * (1 - fy) * fx
           + lat[y0 + 1][:, x0] * fy * (1 - fx) + lat[y0 + 1][:, x0 + 1] * fy * fx)
    img = img + rng.uniform(-8, 8, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def warp_frame(img, seed, shift=(3.7, -2.2), angle_deg=0.5, noise=2.0):
    """Next frame: `img` resampled (bilinear) with a translation + small rotation about the centre + Gaussian noise."""
    rng = np.random.default_rng(seed)
    h, w = img.shape
    a = np.radians(angle_deg)
    cy, cx = (h - 1) / 2.0, (w - 1) / 2.0
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    # inverse map: where does output pixel (x, y) come from
    xs = np.cos(a) * (xx - cx - shift[0]) + np.sin(a) * (yy - cy - shift[1]) + cx
    ys = -np.sin(a) * (xx - cx - shift[0]) + np.cos(a) * (yy - cy - shift[1]) + cy
    xs, ys = np.clip(xs, 0, w - 1.001), np.clip(ys, 0, h - 1.001)
    x0, y0 = np.floor(xs).astype(int), np.floor(ys).astype(int)
    fx, fy = xs - x0, ys - y0
    f = img.astype(np.float64)
    out = f[y0, x0] * (1 - fy) * (1 - fx) + f[y0, x0 + 1] * (1 - fy) * fx + f[y0 + 1, x0] * fy * (1 - fx) + f[y0 + 1, x0 + 1] * fy * fx
    out = out + rng.normal(0, noise, out.shape)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)
