"""float64 NumPy oracle for the sliding-window BA hot path (TEST INFRASTRUCTURE — never shipped).

Pinned against the reference itself for everything in-tree (oracle/_ref, tests/test_ref_parity.py); the Ceres minimiser
stays restated / unpinned (see oracle/__init__.py).  Restates, citing /root/reference:

* quaternion / so(3) helpers ............ vins_estimator/src/utility/utility.h:16-68
* PoseLocalParameterization::Plus ...... factor/pose_local_parameterization.cpp:3-18
* ProjectionFactor::Evaluate ........... factor/projection_factor.cpp:21-121
* ProjectionTdFactor::Evaluate ......... factor/projection_td_factor.cpp:34-141
* IntegrationBase::{propagate,evaluate}  factor/integration_base.h:54-186
* IMUFactor::Evaluate .................. factor/imu_factor.h:19-179
* MarginalizationFactor::Evaluate ...... factor/marginalization_factor.cpp:333-381
* ResidualBlockInfo::Evaluate (loss) ... factor/marginalization_factor.cpp:3-69
* MarginalizationInfo::marginalize ..... factor/marginalization_factor.cpp:174-297
* Estimator::optimization .............. estimator.cpp:670-1003
* Estimator::double2vector (gauge fix) . estimator.cpp:530-619
* ceres::Solve(DENSE_SCHUR, DOGLEG) .... third party (Ceres 1.14, not in tree): restated from
  trust_region_minimizer.cc / dogleg_strategy.cc / corrector.cc / schur_eliminator

Everything here is dense and slow on purpose: the Jacobian is a dense (rows x cols) array and the
Schur complement is written with plain matrix algebra so it can be cross-checked against a full
normal-equation solve.
"""
import copy
import numpy as np

KIND_POSE, KIND_SB, KIND_EX, KIND_TD, KIND_LM = 0, 1, 2, 3, 4
GSIZE = {KIND_POSE: 7, KIND_SB: 9, KIND_EX: 7, KIND_TD: 1, KIND_LM: 1}
LSIZE = {KIND_POSE: 6, KIND_SB: 9, KIND_EX: 6, KIND_TD: 1, KIND_LM: 1}
MARGIN_OLD, MARGIN_SECOND_NEW = 0, 1
O_P, O_R, O_V, O_BA, O_BG = 0, 3, 6, 9, 12


# ----------------------------------------------------------------------------- quaternion helpers
# storage order everywhere: q = [x, y, z, w]  (estimator.cpp:490-497)
def qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def qinv(q):
    n2 = float(np.dot(q, q))
    return np.array([-q[0], -q[1], -q[2], q[3]]) / n2


def qnormalized(q):
    return q / np.sqrt(np.dot(q, q))


def q2R(q):
    """Eigen::Quaternion::toRotationMatrix (no normalisation)."""
    x, y, z, w = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def R2q(m):
    """Eigen quaternion-from-matrix (trace branch method)."""
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def deltaQ(theta):
    """utility.h:16-28 — first order, NOT normalised."""
    return np.array([theta[0] / 2, theta[1] / 2, theta[2] / 2, 1.0])


def Qleft(q):
    """utility.h:51-59, 4x4 in (w,x,y,z) order."""
    v, w = q[:3], q[3]
    m = np.zeros((4, 4))
    m[0, 0] = w
    m[0, 1:] = -v
    m[1:, 0] = v
    m[1:, 1:] = w * np.eye(3) + skew(v)
    return m


def Qright(q):
    """utility.h:61-68."""
    v, w = q[:3], q[3]
    m = np.zeros((4, 4))
    m[0, 0] = w
    m[0, 1:] = -v
    m[1:, 0] = v
    m[1:, 1:] = w * np.eye(3) - skew(v)
    return m


def R2ypr(R):
    """utility.h:70-86 (degrees)."""
    n, o, a = R[:, 0], R[:, 1], R[:, 2]
    y = np.arctan2(n[1], n[0])
    p = np.arctan2(-n[2], n[0] * np.cos(y) + n[1] * np.sin(y))
    r = np.arctan2(a[0] * np.sin(y) - a[1] * np.cos(y), -o[0] * np.sin(y) + o[1] * np.cos(y))
    return np.array([y, p, r]) / np.pi * 180.0


def ypr2R(ypr):
    """utility.h:88-112."""
    y, p, r = np.asarray(ypr) / 180.0 * np.pi
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1.0]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    return Rz @ Ry @ Rx


def pose_plus(x, d):
    """PoseLocalParameterization::Plus (pose_local_parameterization.cpp:3-18)."""
    out = np.empty(7)
    out[:3] = x[:3] + d[:3]
    out[3:] = qnormalized(qmul(x[3:], deltaQ(d[3:6])))
    return out


# ----------------------------------------------------------------------------- projection factors
def projection_factor(pose_i, pose_j, ex, inv_dep, pts_i, pts_j, focal=460.0, need_jac=True):
    """ProjectionFactor::Evaluate (projection_factor.cpp:21-121).  pts_* are (x,y,1) normalised obs.
    Returns r(2), [J_pose_i(2x6), J_pose_j(2x6), J_ex(2x6), J_lambda(2x1)] in tangent columns."""
    sqrt_info = focal / 1.5
    Pi, Qi = pose_i[:3], pose_i[3:]
    Pj, Qj = pose_j[:3], pose_j[3:]
    tic, qic = ex[:3], ex[3:]
    Ri, Rj, ric = q2R(Qi), q2R(Qj), q2R(qic)
    pts_camera_i = pts_i / inv_dep
    pts_imu_i = ric @ pts_camera_i + tic
    pts_w = Ri @ pts_imu_i + Pi
    pts_imu_j = q2R(qinv(Qj)) @ (pts_w - Pj)
    pts_camera_j = q2R(qinv(qic)) @ (pts_imu_j - tic)
    dep_j = pts_camera_j[2]
    r = sqrt_info * ((pts_camera_j / dep_j)[:2] - pts_j[:2])
    if not need_jac:
        return r, None
    reduce = sqrt_info * np.array([[1.0 / dep_j, 0, -pts_camera_j[0] / (dep_j * dep_j)],
                                   [0, 1.0 / dep_j, -pts_camera_j[1] / (dep_j * dep_j)]])
    jaco_i = np.hstack([ric.T @ Rj.T, ric.T @ Rj.T @ Ri @ -skew(pts_imu_i)])
    jaco_j = np.hstack([ric.T @ -Rj.T, ric.T @ skew(pts_imu_j)])
    tmp_r = ric.T @ Rj.T @ Ri @ ric
    jaco_ex = np.hstack([ric.T @ (Rj.T @ Ri - np.eye(3)),
                         -tmp_r @ skew(pts_camera_i) + skew(tmp_r @ pts_camera_i)
                         + skew(ric.T @ (Rj.T @ (Ri @ tic + Pi - Pj) - tic))])
    j_l = (reduce @ ric.T @ Rj.T @ Ri @ ric @ pts_i * -1.0 / (inv_dep * inv_dep)).reshape(2, 1)
    return r, [reduce @ jaco_i, reduce @ jaco_j, reduce @ jaco_ex, j_l]


def projection_td_factor(pose_i, pose_j, ex, inv_dep, td, obs_i, obs_j, focal, tr, row, need_jac=True):
    """ProjectionTdFactor::Evaluate (projection_td_factor.cpp:34-141).
    obs_* = [x, y, u, v, vx, vy, cur_td]; row_* = v - ROW/2 (ctor :18-19)."""
    sqrt_info = focal / 1.5
    pts_i = np.array([obs_i[0], obs_i[1], 1.0])
    pts_j = np.array([obs_j[0], obs_j[1], 1.0])
    vel_i = np.array([obs_i[4], obs_i[5], 0.0])
    vel_j = np.array([obs_j[4], obs_j[5], 0.0])
    td_i, td_j = obs_i[6], obs_j[6]
    row_i, row_j = obs_i[3] - row / 2, obs_j[3] - row / 2
    Pi, Qi = pose_i[:3], pose_i[3:]
    Pj, Qj = pose_j[:3], pose_j[3:]
    tic, qic = ex[:3], ex[3:]
    Ri, Rj, ric = q2R(Qi), q2R(Qj), q2R(qic)
    pts_i_td = pts_i - (td - td_i + tr / row * row_i) * vel_i
    pts_j_td = pts_j - (td - td_j + tr / row * row_j) * vel_j
    pts_camera_i = pts_i_td / inv_dep
    pts_imu_i = ric @ pts_camera_i + tic
    pts_w = Ri @ pts_imu_i + Pi
    pts_imu_j = q2R(qinv(Qj)) @ (pts_w - Pj)
    pts_camera_j = q2R(qinv(qic)) @ (pts_imu_j - tic)
    dep_j = pts_camera_j[2]
    r = sqrt_info * ((pts_camera_j / dep_j)[:2] - pts_j_td[:2])
    if not need_jac:
        return r, None
    reduce = sqrt_info * np.array([[1.0 / dep_j, 0, -pts_camera_j[0] / (dep_j * dep_j)],
                                   [0, 1.0 / dep_j, -pts_camera_j[1] / (dep_j * dep_j)]])
    jaco_i = np.hstack([ric.T @ Rj.T, ric.T @ Rj.T @ Ri @ -skew(pts_imu_i)])
    jaco_j = np.hstack([ric.T @ -Rj.T, ric.T @ skew(pts_imu_j)])
    tmp_r = ric.T @ Rj.T @ Ri @ ric
    jaco_ex = np.hstack([ric.T @ (Rj.T @ Ri - np.eye(3)),
                         -tmp_r @ skew(pts_camera_i) + skew(tmp_r @ pts_camera_i)
                         + skew(ric.T @ (Rj.T @ (Ri @ tic + Pi - Pj) - tic))])
    j_l = (reduce @ tmp_r @ pts_i_td * -1.0 / (inv_dep * inv_dep)).reshape(2, 1)
    j_td = (reduce @ tmp_r @ vel_i / inv_dep * -1.0 + sqrt_info * vel_j[:2]).reshape(2, 1)
    return r, [reduce @ jaco_i, reduce @ jaco_j, reduce @ jaco_ex, j_l, j_td]


# ----------------------------------------------------------------------------- IMU pre-integration
class Preintegration:
    """IntegrationBase (integration_base.h:9-209): mid-point rule + 15x15 jacobian/covariance."""

    def __init__(self, acc_0, gyr_0, ba, bg, acc_n, gyr_n, acc_w, gyr_w):
        self.acc_0, self.gyr_0 = np.array(acc_0, float), np.array(gyr_0, float)
        self.linearized_ba, self.linearized_bg = np.array(ba, float), np.array(bg, float)
        self.jacobian = np.eye(15)
        self.covariance = np.zeros((15, 15))
        self.sum_dt = 0.0
        self.delta_p = np.zeros(3)
        self.delta_q = np.array([0, 0, 0, 1.0])
        self.delta_v = np.zeros(3)
        n = np.zeros(18)
        n[0:3] = acc_n * acc_n
        n[3:6] = gyr_n * gyr_n
        n[6:9] = acc_n * acc_n
        n[9:12] = gyr_n * gyr_n
        n[12:15] = acc_w * acc_w
        n[15:18] = gyr_w * gyr_w
        self.noise = np.diag(n)

    def push_back(self, dt, acc_1, gyr_1):
        """propagate + midPointIntegration (integration_base.h:54-158)."""
        acc_1, gyr_1 = np.asarray(acc_1, float), np.asarray(gyr_1, float)
        acc_0, gyr_0 = self.acc_0, self.gyr_0
        ba, bg = self.linearized_ba, self.linearized_bg
        dq, dp, dv = self.delta_q, self.delta_p, self.delta_v
        Rq = q2R(dq)
        un_acc_0 = Rq @ (acc_0 - ba)
        un_gyr = 0.5 * (gyr_0 + gyr_1) - bg
        res_q = qmul(dq, np.array([un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2, 1.0]))
        Rr = q2R(res_q)  # NB: un-normalised result_delta_q, as the reference uses it
        un_acc_1 = Rr @ (acc_1 - ba)
        un_acc = 0.5 * (un_acc_0 + un_acc_1)
        res_p = dp + dv * dt + 0.5 * un_acc * dt * dt
        res_v = dv + un_acc * dt
        w_x = 0.5 * (gyr_0 + gyr_1) - bg
        R_w_x, R_a_0_x, R_a_1_x = skew(w_x), skew(acc_0 - ba), skew(acc_1 - ba)
        I3 = np.eye(3)
        F = np.zeros((15, 15))
        F[0:3, 0:3] = I3
        F[0:3, 3:6] = -0.25 * Rq @ R_a_0_x * dt * dt + -0.25 * Rr @ R_a_1_x @ (I3 - R_w_x * dt) * dt * dt
        F[0:3, 6:9] = I3 * dt
        F[0:3, 9:12] = -0.25 * (Rq + Rr) * dt * dt
        F[0:3, 12:15] = -0.25 * Rr @ R_a_1_x * dt * dt * -dt
        F[3:6, 3:6] = I3 - R_w_x * dt
        F[3:6, 12:15] = -1.0 * I3 * dt
        F[6:9, 3:6] = -0.5 * Rq @ R_a_0_x * dt + -0.5 * Rr @ R_a_1_x @ (I3 - R_w_x * dt) * dt
        F[6:9, 6:9] = I3
        F[6:9, 9:12] = -0.5 * (Rq + Rr) * dt
        F[6:9, 12:15] = -0.5 * Rr @ R_a_1_x * dt * -dt
        F[9:12, 9:12] = I3
        F[12:15, 12:15] = I3
        V = np.zeros((15, 18))
        V[0:3, 0:3] = 0.25 * Rq * dt * dt
        V[0:3, 3:6] = 0.25 * -Rr @ R_a_1_x * dt * dt * 0.5 * dt
        V[0:3, 6:9] = 0.25 * Rr * dt * dt
        V[0:3, 9:12] = V[0:3, 3:6]
        V[3:6, 3:6] = 0.5 * I3 * dt
        V[3:6, 9:12] = 0.5 * I3 * dt
        V[6:9, 0:3] = 0.5 * Rq * dt
        V[6:9, 3:6] = 0.5 * -Rr @ R_a_1_x * dt * 0.5 * dt
        V[6:9, 6:9] = 0.5 * Rr * dt
        V[6:9, 9:12] = V[6:9, 3:6]
        V[9:12, 12:15] = I3 * dt
        V[12:15, 15:18] = I3 * dt
        self.jacobian = F @ self.jacobian
        self.covariance = F @ self.covariance @ F.T + V @ self.noise @ V.T
        self.delta_p, self.delta_v = res_p, res_v
        self.delta_q = qnormalized(res_q)
        self.sum_dt += dt
        self.acc_0, self.gyr_0 = acc_1, gyr_1

    def as_dict(self):
        return dict(sum_dt=self.sum_dt, delta_p=self.delta_p.copy(), delta_q=self.delta_q.copy(),
                    delta_v=self.delta_v.copy(), lin_ba=self.linearized_ba.copy(),
                    lin_bg=self.linearized_bg.copy(), jacobian=self.jacobian.copy(),
                    covariance=self.covariance.copy())


def imu_sqrt_info(cov):
    """imu_factor.h:64 — LLT(covariance^-1).matrixL().transpose()."""
    return np.linalg.cholesky(np.linalg.inv(cov)).T


def imu_factor(pre, pose_i, sb_i, pose_j, sb_j, g_norm, need_jac=True):
    """IMUFactor::Evaluate (imu_factor.h:19-179) + IntegrationBase::evaluate (integration_base.h:160-186).
    Returns r(15), [J_pose_i 15x6, J_sb_i 15x9, J_pose_j 15x6, J_sb_j 15x9]."""
    G = np.array([0, 0, g_norm])
    Pi, Qi = pose_i[:3], pose_i[3:]
    Pj, Qj = pose_j[:3], pose_j[3:]
    Vi, Bai, Bgi = sb_i[0:3], sb_i[3:6], sb_i[6:9]
    Vj, Baj, Bgj = sb_j[0:3], sb_j[3:6], sb_j[6:9]
    Jm = pre['jacobian']
    dp_dba, dp_dbg = Jm[O_P:O_P + 3, O_BA:O_BA + 3], Jm[O_P:O_P + 3, O_BG:O_BG + 3]
    dq_dbg = Jm[O_R:O_R + 3, O_BG:O_BG + 3]
    dv_dba, dv_dbg = Jm[O_V:O_V + 3, O_BA:O_BA + 3], Jm[O_V:O_V + 3, O_BG:O_BG + 3]
    sum_dt = pre['sum_dt']
    dba, dbg = Bai - pre['lin_ba'], Bgi - pre['lin_bg']
    corrected_delta_q = qmul(pre['delta_q'], deltaQ(dq_dbg @ dbg))
    corrected_delta_v = pre['delta_v'] + dv_dba @ dba + dv_dbg @ dbg
    corrected_delta_p = pre['delta_p'] + dp_dba @ dba + dp_dbg @ dbg
    Qi_inv = qinv(Qi)
    Ri_inv = q2R(Qi_inv)
    r = np.zeros(15)
    r[O_P:O_P + 3] = Ri_inv @ (0.5 * G * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt) - corrected_delta_p
    r[O_R:O_R + 3] = 2 * qmul(qinv(corrected_delta_q), qmul(Qi_inv, Qj))[:3]
    r[O_V:O_V + 3] = Ri_inv @ (G * sum_dt + Vj - Vi) - corrected_delta_v
    r[O_BA:O_BA + 3] = Baj - Bai
    r[O_BG:O_BG + 3] = Bgj - Bgi
    sqrt_info = imu_sqrt_info(pre['covariance'])
    r = sqrt_info @ r
    if not need_jac:
        return r, None
    Qj_inv = qinv(Qj)
    J0 = np.zeros((15, 6))
    J0[O_P:O_P + 3, 0:3] = -Ri_inv
    J0[O_P:O_P + 3, 3:6] = skew(Ri_inv @ (0.5 * G * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt))
    J0[O_R:O_R + 3, 3:6] = -(Qleft(qmul(Qj_inv, Qi)) @ Qright(corrected_delta_q))[1:, 1:]
    J0[O_V:O_V + 3, 3:6] = skew(Ri_inv @ (G * sum_dt + Vj - Vi))
    J1 = np.zeros((15, 9))
    J1[O_P:O_P + 3, 0:3] = -Ri_inv * sum_dt
    J1[O_P:O_P + 3, 3:6] = -dp_dba
    J1[O_P:O_P + 3, 6:9] = -dp_dbg
    J1[O_R:O_R + 3, 6:9] = -Qleft(qmul(qmul(Qj_inv, Qi), pre['delta_q']))[1:, 1:] @ dq_dbg
    J1[O_V:O_V + 3, 0:3] = -Ri_inv
    J1[O_V:O_V + 3, 3:6] = -dv_dba
    J1[O_V:O_V + 3, 6:9] = -dv_dbg
    J1[O_BA:O_BA + 3, 3:6] = -np.eye(3)
    J1[O_BG:O_BG + 3, 6:9] = -np.eye(3)
    J2 = np.zeros((15, 6))
    J2[O_P:O_P + 3, 0:3] = Ri_inv
    J2[O_R:O_R + 3, 3:6] = Qleft(qmul(qinv(corrected_delta_q), qmul(Qi_inv, Qj)))[1:, 1:]
    J3 = np.zeros((15, 9))
    J3[O_V:O_V + 3, 0:3] = Ri_inv
    J3[O_BA:O_BA + 3, 3:6] = np.eye(3)
    J3[O_BG:O_BG + 3, 6:9] = np.eye(3)
    return r, [sqrt_info @ J0, sqrt_info @ J1, sqrt_info @ J2, sqrt_info @ J3]


# ----------------------------------------------------------------------------- prior factor
def prior_dx(prior, blocks_now):
    """dx of MarginalizationFactor::Evaluate (marginalization_factor.cpp:343-363)."""
    dx = np.zeros(prior['n'])
    off = 0
    for (kind, _), x, x0 in zip(prior['blocks'], blocks_now, prior['x0']):
        if GSIZE[kind] != 7:
            dx[off:off + LSIZE[kind]] = x - x0
        else:
            dx[off:off + 3] = x[:3] - x0[:3]
            dq = qmul(qinv(x0[3:]), x[3:])
            dx[off + 3:off + 6] = 2.0 * dq[:3]
            if not (dq[3] >= 0):
                dx[off + 3:off + 6] = 2.0 * -dq[:3]
        off += LSIZE[kind]
    return dx


def prior_factor(prior, blocks_now):
    """r = r0 + J0 dx; Jacobian = J0 (marginalization_factor.cpp:364-378)."""
    return prior['r0'] + prior['J0'] @ prior_dx(prior, blocks_now), prior['J0']


# ----------------------------------------------------------------------------- problem bookkeeping
def cauchy(s):
    """ceres::CauchyLoss(1.0): rho, rho', rho''."""
    return np.log1p(s), 1.0 / (1.0 + s), -1.0 / ((1.0 + s) * (1.0 + s))


def state_of(prob):
    """The optimised state as a dict of arrays (copy)."""
    st = dict(pose=prob['pose'].copy(), sb=prob['sb'].copy(), ex=prob['ex'].copy(),
              td=float(prob['td']), inv_depth=prob['inv_depth'].copy())
    if prob.get('relo') is not None:
        st['relo_pose'] = prob['relo']['pose'].copy()
    return st


def get_block(st, kind, idx):
    if kind == KIND_POSE:
        if idx >= st['pose'].shape[0]:
            return st['relo_pose']
        return st['pose'][idx]
    if kind == KIND_SB:
        return st['sb'][idx]
    if kind == KIND_EX:
        return st['ex']
    if kind == KIND_TD:
        return np.array([st['td']])
    return st['inv_depth'][idx:idx + 1]


class Layout:
    """Tangent-space column layout: poses | speed-biases | ex | td | landmarks."""

    def __init__(self, prob):
        K = prob['pose'].shape[0]
        self.K = K
        self.Kp = K + (1 if prob.get('relo') is not None else 0)
        self.L = prob['inv_depth'].shape[0]
        self.est_ex = bool(prob['estimate_extrinsic'])
        self.est_td = bool(prob['estimate_td'])
        off = 0
        self.pose_off = [off + 6 * i for i in range(self.Kp)]
        off += 6 * self.Kp
        self.sb_off = [off + 9 * i for i in range(K)]
        off += 9 * K
        self.ex_off = off if self.est_ex else -1
        off += 6 if self.est_ex else 0
        self.td_off = off if self.est_td else -1
        off += 1 if self.est_td else 0
        self.R = off                      # reduced (camera-side) dimension
        self.lm_off = off
        self.ncols = off + self.L

    def col(self, kind, idx):
        if kind == KIND_POSE:
            return self.pose_off[idx]
        if kind == KIND_SB:
            return self.sb_off[idx]
        if kind == KIND_EX:
            return self.ex_off
        if kind == KIND_TD:
            return self.td_off
        return self.lm_off + idx


def factor_list(prob):
    """Projection factor list in the reference's order (estimator.cpp:719-764): landmark-major,
    anchor = first observation, one factor per later observation; then relocalisation factors (:769-801).
    Each entry: (landmark, frame_i, frame_j, obs_i(7), obs_j(7))."""
    out = []
    for l in range(prob['inv_depth'].shape[0]):
        s, n, o = int(prob['lm_start'][l]), int(prob['lm_nobs'][l]), int(prob['obs_off'][l])
        for k in range(1, n):
            out.append((l, s, s + k, prob['obs'][o], prob['obs'][o + k]))
    relo = prob.get('relo')
    if relo is not None:
        K = prob['pose'].shape[0]
        for (l, x, y) in relo['match']:
            l = int(l)
            o = int(prob['obs_off'][l])
            oj = np.array([x, y, 0, 0, 0, 0, 0.0])
            out.append((l, int(prob['lm_start'][l]), K, prob['obs'][o], oj))
    return out


def evaluate(prob, st, need_jac=True, use_td_factor=None):
    """Ceres evaluator restated: cost = 1/2 sum rho(||r||^2); residuals/Jacobians loss-corrected
    (corrector.cc; Cauchy always takes the rho''<=0 branch).  Dense J in tangent columns."""
    lay = Layout(prob)
    K = lay.K
    facs = factor_list(prob)
    est_td = lay.est_td if use_td_factor is None else use_td_factor
    nprior = prob['prior']['n'] if prob.get('prior') is not None else 0
    imu_valid = [k for k in range(K - 1) if prob['imu'][k] is not None and prob['imu'][k]['sum_dt'] <= 10.0]
    nrows = nprior + 15 * len(imu_valid) + 2 * len(facs)
    r = np.zeros(nrows)
    J = np.zeros((nrows, lay.ncols)) if need_jac else None
    cost = 0.0
    row = 0
    def pose(i):
        return st['pose'][i] if i < K else st['relo_pose']
    if nprior:
        pr = prob['prior']
        now = [get_block(st, k, i) for (k, i) in pr['blocks']]
        rp, J0 = prior_factor(pr, now)
        r[0:nprior] = rp
        cost += 0.5 * rp @ rp
        if need_jac:
            off = 0
            for (k, i) in pr['blocks']:
                c = lay.col(k, i)
                if c >= 0:
                    J[0:nprior, c:c + LSIZE[k]] = J0[:, off:off + LSIZE[k]]
                off += LSIZE[k]
        row += nprior
    for k in imu_valid:
        ri, Js = imu_factor(prob['imu'][k], st['pose'][k], st['sb'][k], st['pose'][k + 1], st['sb'][k + 1],
                            prob['g_norm'], need_jac)
        r[row:row + 15] = ri
        cost += 0.5 * ri @ ri
        if need_jac:
            J[row:row + 15, lay.pose_off[k]:lay.pose_off[k] + 6] = Js[0]
            J[row:row + 15, lay.sb_off[k]:lay.sb_off[k] + 9] = Js[1]
            J[row:row + 15, lay.pose_off[k + 1]:lay.pose_off[k + 1] + 6] = Js[2]
            J[row:row + 15, lay.sb_off[k + 1]:lay.sb_off[k + 1] + 9] = Js[3]
        row += 15
    for (l, fi, fj, oi, oj) in facs:
        lam = st['inv_depth'][l]
        if est_td:
            rf, Js = projection_td_factor(pose(fi), pose(fj), st['ex'], lam, st['td'], oi, oj,
                                          prob['focal'], prob['tr'], prob['row'], need_jac)
        else:
            rf, Js = projection_factor(pose(fi), pose(fj), st['ex'], lam,
                                       np.array([oi[0], oi[1], 1.0]), np.array([oj[0], oj[1], 1.0]),
                                       prob['focal'], need_jac)
        s = rf @ rf
        rho0, rho1, _ = cauchy(s)
        cost += 0.5 * rho0
        sq = np.sqrt(rho1)
        r[row:row + 2] = sq * rf
        if need_jac:
            J[row:row + 2, lay.pose_off[fi]:lay.pose_off[fi] + 6] = sq * Js[0]
            J[row:row + 2, lay.pose_off[fj]:lay.pose_off[fj] + 6] = sq * Js[1]
            if lay.est_ex:
                J[row:row + 2, lay.ex_off:lay.ex_off + 6] = sq * Js[2]
            J[row:row + 2, lay.lm_off + l:lay.lm_off + l + 1] = sq * Js[3]
            if est_td:
                J[row:row + 2, lay.td_off:lay.td_off + 1] = sq * Js[4]
        row += 2
    return cost, r, J


def plus(prob, st, delta):
    """Apply a tangent-space step to the whole state."""
    lay = Layout(prob)
    out = copy.deepcopy(st)
    for i in range(lay.K):
        out['pose'][i] = pose_plus(st['pose'][i], delta[lay.pose_off[i]:lay.pose_off[i] + 6])
        out['sb'][i] = st['sb'][i] + delta[lay.sb_off[i]:lay.sb_off[i] + 9]
    if lay.Kp > lay.K:
        c = lay.pose_off[lay.K]
        out['relo_pose'] = pose_plus(st['relo_pose'], delta[c:c + 6])
    if lay.est_ex:
        out['ex'] = pose_plus(st['ex'], delta[lay.ex_off:lay.ex_off + 6])
    if lay.est_td:
        out['td'] = st['td'] + delta[lay.td_off]
    out['inv_depth'] = st['inv_depth'] + delta[lay.lm_off:]
    return out


def ambient_vector(prob, st):
    """All non-constant parameters in ambient coordinates (for the parameter-tolerance test)."""
    lay = Layout(prob)
    parts = [st['pose'].ravel(), st['sb'].ravel()]
    if lay.Kp > lay.K:
        parts.append(st['relo_pose'])
    if lay.est_ex:
        parts.append(st['ex'])
    if lay.est_td:
        parts.append(np.array([st['td']]))
    parts.append(st['inv_depth'])
    return np.concatenate(parts)


def dense_schur_solve(J, r, D, R):
    """DENSE_SCHUR: min ||J y - r||^2 + ||D y||^2 with the last (ncols-R) 1-wide landmark columns
    eliminated first; reduced system by dense LLT.  Returns y or None on failure."""
    Jp, Jl = J[:, :R], J[:, R:]
    Hpp = Jp.T @ Jp + np.diag(D[:R] ** 2)
    hll = np.einsum('ij,ij->j', Jl, Jl) + D[R:] ** 2
    W = Jp.T @ Jl
    gp, gl = Jp.T @ r, Jl.T @ r
    if np.any(hll <= 0) or not np.all(np.isfinite(hll)):
        return None
    S = Hpp - (W / hll) @ W.T
    gr = gp - W @ (gl / hll)
    try:
        Lc = np.linalg.cholesky(S)
    except np.linalg.LinAlgError:
        return None
    yp = np.linalg.solve(Lc.T, np.linalg.solve(Lc, gr))
    yl = (gl - W.T @ yp) / hll
    y = np.concatenate([yp, yl])
    return y if np.all(np.isfinite(y)) else None


def solve(prob, trace=None, min_mu=1e-8):
    """ceres::Solve as configured at estimator.cpp:803-818 (DENSE_SCHUR, DOGLEG, max 8 iterations,
    wall-clock cap OFF).  Returns (state, summary)."""
    lay = Layout(prob)
    R = lay.R
    x = state_of(prob)
    max_iters = int(prob['max_iters'])
    cost, r, J = evaluate(prob, x)
    scale = 1.0 / (1.0 + np.sqrt(np.einsum('ij,ij->j', J, J)))
    J = J * scale
    g = J.T @ r   # scaled-space gradient (used by dogleg); unscaled max-norm for the tolerance test
    summary = dict(initial_cost=cost, iterations=[], termination='NO_CONVERGENCE')
    if np.max(np.abs(g / scale)) <= 1e-10:
        summary['termination'] = 'CONVERGENCE'
        summary['final_cost'] = cost
        return x, summary
    radius, mu, reuse = 1e4, min_mu, False
    max_mu = 1.0
    x_norm = np.linalg.norm(ambient_vector(prob, x))
    num_invalid = 0
    Dg = gt = gn = None
    alpha = 0.0
    dogleg_norm = 0.0
    it = 0
    while True:
        if it >= max_iters:
            break
        it += 1
        rec = dict(iter=it)
        # ---- DoglegStrategy::ComputeStep
        ok = True
        if not reuse:
            reuse = True
            Dg = np.sqrt(np.clip(np.einsum('ij,ij->j', J, J), 1e-6, 1e32))
            gt = (J.T @ r) / Dg
            Jg = J @ (gt / Dg)
            alpha = (gt @ gt) / (Jg @ Jg)
            y = None
            mu_tries = 0
            while mu < max_mu:
                y = dense_schur_solve(J, r, Dg * np.sqrt(mu), R)
                if y is None:
                    mu *= 10.0
                    mu_tries += 1
                    continue
                break
            rec.update(mu=mu, mu_tries=mu_tries)
            if y is None:
                ok = False
            else:
                gn = -(y * Dg)
        if ok:
            gnorm, gnn = np.linalg.norm(gt), np.linalg.norm(gn)
            if gnn <= radius:
                s, dogleg_norm = gn.copy(), gnn
                rec.update(branch='gn')
            elif gnorm * alpha >= radius:
                s, dogleg_norm = -(radius / gnorm) * gt, radius
                rec.update(branch='cauchy')
            else:
                rec.update(branch='dogleg')
                b_dot_a = -alpha * (gt @ gn)
                a_sq = (alpha * gnorm) ** 2
                bma_sq = a_sq - 2 * b_dot_a + gnn ** 2
                c = b_dot_a - a_sq
                d = np.sqrt(c * c + bma_sq * (radius ** 2 - a_sq))
                beta = (d - c) / bma_sq if c <= 0 else (radius * radius - a_sq) / (d + c)
                s = (-alpha * (1.0 - beta)) * gt + beta * gn
                dogleg_norm = np.linalg.norm(s)
            step = s / Dg
            Jstep = J @ step
            model_change = -Jstep @ (r + Jstep / 2.0)
        if (not ok) or not (model_change > 0):
            num_invalid += 1
            rec.update(valid=False)
            summary['iterations'].append(rec)
            if num_invalid >= 5:
                summary['termination'] = 'FAILURE'
                break
            mu *= 10.0
            reuse = False
            continue
        num_invalid = 0
        delta = step * scale
        x_cand = plus(prob, x, delta)
        cost_cand, _, _ = evaluate(prob, x_cand, need_jac=False)
        rec.update(valid=True, cost=cost, cost_cand=cost_cand, model_change=model_change, radius=radius,
                   step_norm=dogleg_norm)
        step_norm = np.linalg.norm(ambient_vector(prob, x) - ambient_vector(prob, x_cand))
        if step_norm <= 1e-8 * (x_norm + 1e-8):
            rec.update(accepted=False, exit='parameter_tolerance')
            summary['iterations'].append(rec)
            summary['termination'] = 'CONVERGENCE'
            break
        if abs(cost - cost_cand) <= 1e-6 * cost:
            rec.update(accepted=False, exit='function_tolerance')
            summary['iterations'].append(rec)
            summary['termination'] = 'CONVERGENCE'
            break
        rho = (cost - cost_cand) / model_change
        rec.update(rho=rho)
        if rho > 1e-3:
            x = x_cand
            x_norm = np.linalg.norm(ambient_vector(prob, x))
            cost, r, J = evaluate(prob, x)
            J = J * scale
            rec.update(accepted=True)
            summary['iterations'].append(rec)
            if rho < 0.25:
                radius *= 0.5
            if rho > 0.75:
                radius = max(radius, 3.0 * dogleg_norm)
            mu = max(min_mu, 2.0 * mu / 10.0)
            reuse = False
            if np.max(np.abs((J.T @ r) / scale)) <= 1e-10:
                summary['termination'] = 'CONVERGENCE'
                break
        else:
            rec.update(accepted=False)
            summary['iterations'].append(rec)
            radius *= 0.5
            reuse = True
    summary['final_cost'] = cost
    summary['num_iterations'] = it
    if trace is not None:
        trace.update(summary)
    return x, summary


# ----------------------------------------------------------------------------- gauge fix (double2vector)
def double2vector(prob, st):
    """Estimator::double2vector (estimator.cpp:530-619) followed by vector2double's R->q repack
    (:486-528): yaw/position of frame 0 are pinned to their pre-solve values."""
    K = prob['pose'].shape[0]
    Rs0 = q2R(prob['pose'][0][3:])
    origin_R0 = R2ypr(Rs0)
    origin_P0 = prob['pose'][0][:3].copy()
    R00 = q2R(st['pose'][0][3:])
    origin_R00 = R2ypr(R00)
    y_diff = origin_R0[0] - origin_R00[0]
    rot_diff = ypr2R(np.array([y_diff, 0, 0]))
    if abs(abs(origin_R0[1]) - 90) < 1.0 or abs(abs(origin_R00[1]) - 90) < 1.0:
        rot_diff = Rs0 @ R00.T
    out = copy.deepcopy(st)
    for i in range(K):
        Ri = rot_diff @ q2R(qnormalized(st['pose'][i][3:]))
        out['pose'][i][:3] = rot_diff @ (st['pose'][i][:3] - st['pose'][0][:3]) + origin_P0
        out['pose'][i][3:] = R2q(Ri)
        out['sb'][i][:3] = rot_diff @ st['sb'][i][:3]
    out['ex'][3:] = R2q(q2R(st['ex'][3:]))
    # setDepth stores 1/x, getDepthVector returns 1/that (feature_manager.cpp:141-200)
    out['inv_depth'] = 1.0 / (1.0 / st['inv_depth'])
    return out


# ----------------------------------------------------------------------------- marginalization
def _marg_factors(prob, st, flag):
    """The ResidualBlockInfo list of estimator.cpp:828-903 (MARGIN_OLD) / :935-956 (SECOND_NEW).
    Each entry: (r, [J_block...], [(kind, idx)...], drop_positions)."""
    lay = Layout(prob)
    K = lay.K
    out = []
    pr = prob.get('prior')
    if flag == MARGIN_OLD:
        if pr is not None:
            now = [get_block(st, k, i) for (k, i) in pr['blocks']]
            rp, J0 = prior_factor(pr, now)
            Js, off = [], 0
            for (k, i) in pr['blocks']:
                Js.append(J0[:, off:off + LSIZE[k]])
                off += LSIZE[k]
            drop = [p for p, (k, i) in enumerate(pr['blocks']) if (k, i) in ((KIND_POSE, 0), (KIND_SB, 0))]
            out.append((rp, Js, list(pr['blocks']), drop))
        pre = prob['imu'][0]
        if pre is not None and pre['sum_dt'] < 10.0:
            ri, Js = imu_factor(pre, st['pose'][0], st['sb'][0], st['pose'][1], st['sb'][1], prob['g_norm'])
            out.append((ri, Js, [(KIND_POSE, 0), (KIND_SB, 0), (KIND_POSE, 1), (KIND_SB, 1)], [0, 1]))
        for l in range(lay.L):
            s, n, o = int(prob['lm_start'][l]), int(prob['lm_nobs'][l]), int(prob['obs_off'][l])
            if s != 0:
                continue
            for k in range(1, n):
                oi, oj = prob['obs'][o], prob['obs'][o + k]
                if lay.est_td:
                    rf, Js = projection_td_factor(st['pose'][0], st['pose'][k], st['ex'], st['inv_depth'][l],
                                                  st['td'], oi, oj, prob['focal'], prob['tr'], prob['row'])
                    blocks = [(KIND_POSE, 0), (KIND_POSE, k), (KIND_EX, 0), (KIND_LM, l), (KIND_TD, 0)]
                else:
                    rf, Js = projection_factor(st['pose'][0], st['pose'][k], st['ex'], st['inv_depth'][l],
                                               np.array([oi[0], oi[1], 1.0]), np.array([oj[0], oj[1], 1.0]),
                                               prob['focal'])
                    blocks = [(KIND_POSE, 0), (KIND_POSE, k), (KIND_EX, 0), (KIND_LM, l)]
                # ResidualBlockInfo::Evaluate loss correction (marginalization_factor.cpp:37-68)
                sq_norm = rf @ rf
                _, rho1, rho2 = cauchy(sq_norm)
                sqrt_rho1 = np.sqrt(rho1)
                if sq_norm == 0.0 or rho2 <= 0.0:
                    rs, a = sqrt_rho1, 0.0
                else:  # pragma: no cover  (never for Cauchy)
                    Dd = 1.0 + 2.0 * sq_norm * rho2 / rho1
                    al = 1.0 - np.sqrt(Dd)
                    rs, a = sqrt_rho1 / (1 - al), al / sq_norm
                Js = [sqrt_rho1 * (Jb - a * np.outer(rf, rf @ Jb)) for Jb in Js]
                out.append((rf * rs, Js, blocks, [0, 3]))
    else:
        if pr is not None:
            now = [get_block(st, k, i) for (k, i) in pr['blocks']]
            rp, J0 = prior_factor(pr, now)
            Js, off = [], 0
            for (k, i) in pr['blocks']:
                Js.append(J0[:, off:off + LSIZE[k]])
                off += LSIZE[k]
            drop = [p for p, (k, i) in enumerate(pr['blocks']) if (k, i) == (KIND_POSE, K - 2)]
            out.append((rp, Js, list(pr['blocks']), drop))
    return out


def _block_sort_key(b):
    kind, idx = b
    order = {KIND_POSE: 0, KIND_SB: 1, KIND_EX: 2, KIND_TD: 3, KIND_LM: 4}[kind]
    return (order, idx)


def marginalize(prob, st, flag, eps=1e-8):
    """Estimator::optimization's marginalization step (estimator.cpp:825-1000) with
    MarginalizationInfo::{preMarginalize,marginalize,getParameterBlocks}
    (marginalization_factor.cpp:110-319).  `st` must be the post-double2vector state.
    Canonical block order (the reference's is unordered_map-over-addresses, i.e. arbitrary):
    dropped = poses, speed-biases, landmarks ascending; kept = poses asc, speed-biases, ex, td.
    Returns the new prior dict (blocks already re-labelled for the slid window) or None."""
    K = prob['pose'].shape[0]
    pr = prob.get('prior')
    if flag == MARGIN_SECOND_NEW:
        if pr is None or (KIND_POSE, K - 2) not in pr['blocks']:
            return pr  # estimator.cpp:935-936: prior untouched
    facs = _marg_factors(prob, st, flag)
    if not facs:
        return None
    dropped, seen = set(), set()
    for (_, _, blocks, drop) in facs:
        for b in blocks:
            seen.add(b)
        for p in drop:
            dropped.add(blocks[p])
    drop_list = sorted(dropped, key=_block_sort_key)
    keep_list = sorted(seen - dropped, key=_block_sort_key)
    idx, pos = {}, 0
    for b in drop_list:
        idx[b] = pos
        pos += LSIZE[b[0]]
    m = pos
    for b in keep_list:
        idx[b] = pos
        pos += LSIZE[b[0]]
    n = pos - m
    A = np.zeros((pos, pos))
    bvec = np.zeros(pos)
    for (rf, Js, blocks, _) in facs:
        for a_i, ba in enumerate(blocks):
            ia, sa = idx[ba], LSIZE[ba[0]]
            for b_i in range(a_i, len(blocks)):
                bb = blocks[b_i]
                ib, sb_ = idx[bb], LSIZE[bb[0]]
                blk = Js[a_i].T @ Js[b_i]
                if a_i == b_i:
                    A[ia:ia + sa, ib:ib + sb_] += blk
                else:
                    A[ia:ia + sa, ib:ib + sb_] += blk
                    A[ib:ib + sb_, ia:ia + sa] = A[ia:ia + sa, ib:ib + sb_].T
            bvec[ia:ia + sa] += Js[a_i].T @ rf
    Amm = 0.5 * (A[:m, :m] + A[:m, :m].T)
    w, V = np.linalg.eigh(Amm)
    winv = np.where(w > eps, 1.0 / np.where(w > eps, w, 1.0), 0.0)
    Amm_inv = V @ np.diag(winv) @ V.T
    bmm, Amr, Arm, Arr, brr = bvec[:m], A[:m, m:], A[m:, :m], A[m:, m:], bvec[m:]
    A2 = Arr - Arm @ Amm_inv @ Amr
    b2 = brr - Arm @ Amm_inv @ bmm
    w2, V2 = np.linalg.eigh(A2)
    S = np.where(w2 > eps, w2, 0.0)
    S_inv = np.where(w2 > eps, 1.0 / np.where(w2 > eps, w2, 1.0), 0.0)
    J0 = np.diag(np.sqrt(S)) @ V2.T
    r0 = np.diag(np.sqrt(S_inv)) @ V2.T @ b2
    # getParameterBlocks + addr_shift (estimator.cpp:913-930 / :969-996)
    new_blocks, x0 = [], []
    for b in keep_list:
        kind, i = b
        x0.append(np.array(get_block(st, kind, i), float).copy())
        if kind in (KIND_POSE, KIND_SB):
            if flag == MARGIN_OLD:
                new_blocks.append((kind, i - 1))
            else:
                new_blocks.append((kind, i - 1 if i == K - 1 else i))
        else:
            new_blocks.append((kind, i))
    return dict(n=n, m=m, blocks=new_blocks, J0=J0, r0=r0, x0=x0, A=A2, b=b2, A_full=A, b_full=bvec)


def schur_extended(A, bvec, m):
    """Schur complement of the leading m x m block in 80-bit extended precision (np.longdouble), exact inverse
    (no eps cut).  NOT part of the reference algorithm: a yardstick that tells whose double-precision
    eigen-based pseudo-inverse (numpy's or the device's) is closer to the truth when they disagree at 1e-8."""
    Al = np.array(A, dtype=np.longdouble)
    bl = np.array(bvec, dtype=np.longdouble)
    M = 0.5 * (Al[:m, :m] + Al[:m, :m].T)
    X = np.concatenate([Al[:m, m:], bl[:m, None]], axis=1)
    # Gauss-Jordan on the SPD block (symmetric pivots are fine)
    M = M.copy()
    for k in range(m):
        piv = M[k, k]
        M[k, :] /= piv
        X[k, :] /= piv
        for i in range(m):
            if i != k:
                f = M[i, k]
                if f != 0:
                    M[i, :] -= f * M[k, :]
                    X[i, :] -= f * X[k, :]
    A2 = Al[m:, m:] - Al[m:, :m] @ X[:, :-1]
    b2 = bl[m:] - Al[m:, :m] @ X[:, -1]
    return np.array(A2, dtype=np.float64), np.array(b2, dtype=np.float64)


def optimization(prob, flag):
    """Estimator::optimization(): solve, gauge-fix, marginalize.  Returns (state, summary, new_prior)."""
    x, summary = solve(prob)
    st = double2vector(prob, x)
    new_prior = marginalize(prob, st, flag)
    return st, summary, new_prior


def triangulate(Ps, Rs, tic, ric, start, nobs, obs_off, points, init_depth=5.0):
    """FeatureManager::triangulate (feature_manager.cpp:202-257): DLT rows f0*P2 - f2*P0, f1*P2 - f2*P1 relative to the
    first observing frame (:222-241), svd_V = last right singular vector (:244), depth = V[2]/V[3], INIT_DEPTH if < 0.1."""
    Ps, Rs, points = np.asarray(Ps, float), np.asarray(Rs, float).reshape(-1, 3, 3), np.asarray(points, float).reshape(-1, 3)
    tic, ric = np.asarray(tic, float), np.asarray(ric, float).reshape(3, 3)
    out = np.zeros(len(start))
    for l in range(len(start)):
        i0, n = int(start[l]), int(nobs[l])
        t0 = Ps[i0] + Rs[i0] @ tic
        R0 = Rs[i0] @ ric
        A = np.zeros((2 * n, 4))
        for j in range(n):
            f_ = i0 + j
            t1 = Ps[f_] + Rs[f_] @ tic
            R1 = Rs[f_] @ ric
            t = R0.T @ (t1 - t0)
            R = R0.T @ R1
            P = np.zeros((3, 4))
            P[:, :3] = R.T
            P[:, 3] = -R.T @ t
            f = points[int(obs_off[l]) + j]
            f = f / np.linalg.norm(f)
            A[2 * j] = f[0] * P[2] - f[2] * P[0]
            A[2 * j + 1] = f[1] * P[2] - f[2] * P[1]
        v = np.linalg.svd(A)[2][-1]
        d = v[2] / v[3]
        out[l] = init_depth if d < 0.1 else d
    return out
