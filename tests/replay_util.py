"""BA replay plan shared by the C++ replay harness (vins-mono_amd/host/replay_main.cpp, `vins_replay ba`) and its
oracle-side mirror: N consecutive sliding windows with the prior, the states and the landmark depths carried from window
to window exactly as Estimator::optimization() + slideWindow() do (estimator.cpp:825-1000, :1005-1126;
feature_manager.cpp:275-313)."""
import struct

import numpy as np

from oracle import ba_numpy as B
from vins_mono_amd import synth

INIT_DEPTH = 5.0


def make_plan(seed, n_windows, L=60):
    K = 11
    seq = synth.SyntheticSequence(seed, n_frames=K + n_windows, K=K, L=L)
    c = seq.cfg
    nf = K + n_windows - 1
    frames = [(float(seq.times[f]), seq._noisy_pose(f), seq._noisy_sb(f)) for f in range(nf)]
    h = seq.frame_dt / seq.imu_per_frame
    intervals = []
    for k in range(nf - 1):
        t = seq.times[k]
        smp = [(0.0,) + seq._imu_sample(t)]
        for s in range(1, seq.imu_per_frame + 1):
            smp.append((h,) + seq._imu_sample(t + s * h))
        intervals.append(smp)
    tables = []
    for w0 in range(n_windows):
        rows = []
        for lid, lm in enumerate(seq.lm):
            lo, hi = max(lm['f0'], w0), min(lm['f0'] + len(lm['obs']) - 1, w0 + K - 1)
            n, s = hi - lo + 1, lo - w0
            if n < 2 or not (s < K - 3):
                continue
            obs, prev = [], None
            for f in range(lo, hi + 1):
                xy = lm['obs'][f - lm['f0']]
                vel = (xy - prev) / seq.frame_dt if prev is not None else np.zeros(2)
                prev = xy
                obs.append([xy[0], xy[1], c['fx'] * xy[0] + c['cx'], c['fy'] * xy[1] + c['cy'], vel[0], vel[1], 0.0])
            pc = c['ric'].T @ (seq.Rm[lo].T @ (lm['Xw'] - seq.P[lo]) - c['tic'])
            init = 1.0 / (pc[2] * (1.0 + seq.rng.normal(0, 0.1)))
            rows.append(dict(id=lid, start=s, nobs=n, obs=np.array(obs), init=float(init)))
        tables.append(rows)
    return dict(seq=seq, K=K, W=n_windows, frames=frames, intervals=intervals, tables=tables)


def write_sequence(plan, path):
    seq, K, W = plan['seq'], plan['K'], plan['W']
    c = seq.cfg
    S = seq.imu_per_frame
    base = seq._base()

    def interval(f, k):
        smp = plan['intervals'][k]
        f.write(struct.pack("6d", *smp[0][1], *smp[0][2]))
        for dt, a, g in smp[1:]:
            f.write(struct.pack("7d", dt, *a, *g))

    with open(path, "wb") as f:
        f.write(struct.pack("4i", 0x31414256, W, K, S))
        f.write(struct.pack("7d", *base['ex']))
        f.write(struct.pack("4d", c['acc_n'], c['gyr_n'], c['acc_w'], c['gyr_w']))
        f.write(struct.pack("2d", c['g_norm'], c['focal']))
        f.write(struct.pack("6d", *seq.ba_lin, *seq.bg_lin))
        for i in range(K):
            t, pose, sb = plan['frames'][i]
            f.write(struct.pack("17d", t, *pose, *sb))
        for k in range(K - 1):
            interval(f, k)
        for w in range(W):
            if w > 0:
                t, pose, sb = plan['frames'][w + K - 1]
                f.write(struct.pack("17d", t, *pose, *sb))
                interval(f, w + K - 2)
            rows = plan['tables'][w]
            f.write(struct.pack("i", len(rows)))
            for r in rows:
                f.write(struct.pack("3i", r['id'], r['start'], r['nobs']))
                f.write(struct.pack("d", r['init']))
                f.write(r['obs'].astype(np.float64).tobytes())


def run_oracle(plan):
    """The same replay through oracle/ba_numpy.py; returns the CSV rows [stamp_ns, P(3), Q(w x y z), V(3)]."""
    seq, K, W = plan['seq'], plan['K'], plan['W']
    c = seq.cfg
    pre = [synth.preintegrate(s, seq.ba_lin, seq.bg_lin, c['acc_n'], c['gyr_n'], c['acc_w'], c['gyr_w']) for s in plan['intervals']]
    depth_by_id, prior, st_prev, out = {}, None, None, []
    for w in range(W):
        prob = seq._base()
        if w == 0:
            prob['pose'] = np.array([plan['frames'][i][1] for i in range(K)])
            prob['sb'] = np.array([plan['frames'][i][2] for i in range(K)])
        else:
            prob['pose'] = np.vstack([st_prev['pose'][1:K], plan['frames'][w + K - 1][1][None]])
            prob['sb'] = np.vstack([st_prev['sb'][1:K], plan['frames'][w + K - 1][2][None]])
        rows = plan['tables'][w]
        prob['lm_start'] = np.array([r['start'] for r in rows], np.int32)
        prob['lm_nobs'] = np.array([r['nobs'] for r in rows], np.int32)
        prob['obs_off'] = np.cumsum([0] + [r['nobs'] for r in rows])[:-1].astype(np.int32)
        prob['obs'] = np.vstack([r['obs'] for r in rows])
        prob['inv_depth'] = np.array([1.0 / depth_by_id[r['id']] if r['id'] in depth_by_id else r['init'] for r in rows])
        prob['imu'] = [dict((k, np.copy(v)) for k, v in pre[w + i].items()) for i in range(K - 1)]
        prob['prior'] = prior
        st, _, prior = B.optimization(prob, B.MARGIN_OLD)
        q = st['pose'][K - 1][3:]
        q = B.R2q(B.q2R(q))                      # Quaterniond(Rs[WINDOW_SIZE]) of pubOdometry
        out.append([plan['frames'][w + K - 1][0] * 1e9, *st['pose'][K - 1][:3], q[3], q[0], q[1], q[2], *st['sb'][K - 1][:3]])
        # slideWindowOld -> removeBackShiftDepth with the optimised frames 0 and 1 (estimator.cpp:1115-1126)
        ric, tic = B.q2R(st['ex'][3:]), st['ex'][:3]
        Rb0, Rb1 = B.q2R(st['pose'][0][3:]), B.q2R(st['pose'][1][3:])
        R0, P0 = Rb0 @ ric, st['pose'][0][:3] + Rb0 @ tic
        R1, P1 = Rb1 @ ric, st['pose'][1][:3] + Rb1 @ tic
        nxt = {}
        for l, r in enumerate(rows):
            depth = 1.0 / st['inv_depth'][l]
            if r['start'] != 0:
                nxt[r['id']] = depth
            elif r['nobs'] - 1 >= 2:
                pj = R1.T @ (R0 @ (np.array([r['obs'][0][0], r['obs'][0][1], 1.0]) * depth) + P0 - P1)
                nxt[r['id']] = pj[2] if pj[2] > 0 else INIT_DEPTH
        depth_by_id, st_prev = nxt, st
    return np.array(out)
