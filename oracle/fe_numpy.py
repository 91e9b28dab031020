"""oracle/fe_numpy.py — SECOND, independent CPU restatement of the front-end arithmetic (TEST INFRASTRUCTURE).

Written from the algorithm descriptions of SURVEY.md Appendix B and oracle/ASSUMPTIONS.md (rows F1-F6), NOT from
oracle/fe_cpu.cpp: whole-array NumPy formulations (padding + slicing, integer tensors, scipy.ndimage for the rank / box
filters) instead of the C++ oracle's per-pixel loops, so that an error of recall or of indexing in one restatement
shows up as a disagreement (tests/test_fe_oracle.py).  The third-party behaviour restated ([3P], OpenCV 3.3-era):

  pyrdown   cv::pyrDown as used by buildOpticalFlowPyramid            feature_tracker.cpp:113 (calcOpticalFlowPyrLK)
  scharr    calcSharrDeriv                                              same call
  lk        LKTrackerInvoker, window 21x21, maxLevel 3, 30 its, eps .01 same call
  mineig / gftt   cv::goodFeaturesToTrack(img, n, 0.01, minDist, mask)  feature_tracker.cpp:149
  clahe     cv::createCLAHE(3.0, Size(8,8))->apply                      feature_tracker.cpp:87-93

Only tests/ may import this module.  It is slow (pure NumPy) and meant for small inputs.
"""
import numpy as np
from scipy import ndimage

WIN = 21
HALF = 10
W_BITS = 14


def _reflect101(img, pad):
    return np.pad(img, pad, mode='reflect')          # numpy 'reflect' == BORDER_REFLECT_101 (edge pixel not repeated)


# ----------------------------------------------------------------------------- F1 pyrDown
def pyrdown(img):
    """[1 4 6 4 1] x [1 4 6 4 1], REFLECT_101, dst = ((W+1)/2, (H+1)/2), (sum + 128) >> 8."""
    img = np.asarray(img, np.uint8)
    h, w = img.shape
    dh, dw = (h + 1) // 2, (w + 1) // 2
    p = _reflect101(img.astype(np.int64), 2)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    # horizontal pass at the even columns, all (padded) rows
    cols = 2 * np.arange(dw)
    rowp = sum(k[t] * p[:, cols + t] for t in range(5))            # centre 2x <-> padded index 2x + 2
    rows = 2 * np.arange(dh)
    out = sum(k[t] * rowp[rows + t, :] for t in range(5))
    return ((out + 128) >> 8).astype(np.uint8)


def pyramid(img, max_level=3):
    """Levels while both dimensions of the NEXT level stay larger than the 21-px window."""
    lv = [np.asarray(img, np.uint8)]
    while len(lv) <= max_level:
        h, w = lv[-1].shape
        nh, nw = (h + 1) // 2, (w + 1) // 2
        if nw <= WIN or nh <= WIN:
            break
        lv.append(pyrdown(lv[-1]))
    return lv


# ----------------------------------------------------------------------------- F2 Scharr
def scharr(img):
    """(Ix, Iy) int16, un-normalised (gain 32), reflect-101 at the image edge."""
    p = _reflect101(np.asarray(img, np.uint8).astype(np.int32), 1)
    up, mid, dn = p[:-2, :], p[1:-1, :], p[2:, :]
    t0 = 3 * (up + dn) + 10 * mid                    # vertical smooth, all padded columns
    t1 = dn - up                                     # vertical difference
    ix = t0[:, 2:] - t0[:, :-2]
    iy = 3 * (t1[:, :-2] + t1[:, 2:]) + 10 * t1[:, 1:-1]
    return ix.astype(np.int16), iy.astype(np.int16)


# ----------------------------------------------------------------------------- F3 LK
def _round_half_even(x):
    return np.rint(x)                                # numpy rint = IEEE round-half-to-even = cvRound


def _weights(fx, fy):
    """14-bit bilinear weights from float32 fractions: iw00, iw01, iw10 by cvRound, iw11 = remainder."""
    a, b = np.float32(fx), np.float32(fy)
    one = np.float32(1.0)
    s = np.float32(1 << W_BITS)
    iw00 = int(_round_half_even((one - a) * (one - b) * s))
    iw01 = int(_round_half_even(a * (one - b) * s))
    iw10 = int(_round_half_even((one - a) * b * s))
    return iw00, iw01, iw10, (1 << W_BITS) - iw00 - iw01 - iw10


def _descale(v, n):
    return (v + (1 << (n - 1))) >> n                 # arithmetic shift on int64 arrays


def _patch(padded, pad, ix, iy, w4, shift):
    """Bilinear 21x21 sample of an integer image stored with `pad` pixels of border, top-left at integer (ix, iy)."""
    y0, x0 = iy + pad, ix + pad
    q = padded[y0:y0 + WIN + 1, x0:x0 + WIN + 1].astype(np.int64)
    v = q[:-1, :-1] * w4[0] + q[:-1, 1:] * w4[1] + q[1:, :-1] * w4[2] + q[1:, 1:] * w4[3]
    return _descale(v, shift)


def lk(prev, nxt, pts, max_level=3, max_count=30, eps=0.01, min_eig_threshold=1e-4):
    """calcOpticalFlowPyrLK(prev, next, pts, Size(21,21), max_level).  Returns (next_pts float32 [n,2], status u8, err f32).
    A / b are accumulated exactly in int64 and converted once to float32 (ASSUMPTIONS F3)."""
    pI, pJ = pyramid(prev, max_level), pyramid(nxt, max_level)
    nl = len(pI) - 1
    pts = np.asarray(pts, np.float32).reshape(-1, 2)
    n = pts.shape[0]
    out = np.zeros((n, 2), np.float32)
    status = np.ones(n, np.uint8)
    err = np.zeros(n, np.float32)
    PAD = WIN + 2
    levels = []
    for lvl in range(nl + 1):
        I, J = pI[lvl], pJ[lvl]
        dx, dy = scharr(I)
        # pyramid images carry a REFLECT_101 border, the derivative a border of zeros
        levels.append((_reflect101(I.astype(np.int64), PAD), _reflect101(J.astype(np.int64), PAD),
                       np.pad(dx.astype(np.int64), PAD), np.pad(dy.astype(np.int64), PAD), I.shape))
    f32 = np.float32
    scale_ab = f32(1.0 / (1 << 20))
    for lvl in range(nl, -1, -1):
        Ip, Jp, Dxp, Dyp, (H, W) = levels[lvl]
        for i in range(n):
            prevpt = pts[i] * f32(1.0 / (1 << lvl))
            nextpt = prevpt.copy() if lvl == nl else out[i] * f32(2.0)
            out[i] = nextpt
            prevpt = prevpt - f32(HALF)
            ipx, ipy = int(np.floor(prevpt[0])), int(np.floor(prevpt[1]))
            if ipx < -WIN or ipx >= W or ipy < -WIN or ipy >= H:
                if lvl == 0:
                    status[i] = 0
                    err[i] = 0
                continue
            w4 = _weights(prevpt[0] - f32(ipx), prevpt[1] - f32(ipy))
            Ipatch = _patch(Ip, PAD, ipx, ipy, w4, W_BITS - 5)
            dIx = _patch(Dxp, PAD, ipx, ipy, w4, W_BITS)
            dIy = _patch(Dyp, PAD, ipx, ipy, w4, W_BITS)
            A11 = f32(int((dIx * dIx).sum())) * scale_ab
            A12 = f32(int((dIx * dIy).sum())) * scale_ab
            A22 = f32(int((dIy * dIy).sum())) * scale_ab
            D = A11 * A22 - A12 * A12
            min_eig = (A22 + A11 - np.sqrt((A11 - A22) * (A11 - A22) + f32(4.0) * A12 * A12, dtype=np.float32)) / f32(2 * WIN * WIN)
            if min_eig < f32(min_eig_threshold) or D < np.finfo(np.float32).eps:
                if lvl == 0:
                    status[i] = 0
                continue
            D = f32(1.0) / D
            nextpt = nextpt - f32(HALF)
            prev_delta = np.zeros(2, np.float32)
            for j in range(max_count):
                inx, iny = int(np.floor(nextpt[0])), int(np.floor(nextpt[1]))
                if inx < -WIN or inx >= W or iny < -WIN or iny >= H:
                    if lvl == 0:
                        status[i] = 0
                    break
                wj = _weights(nextpt[0] - f32(inx), nextpt[1] - f32(iny))
                diff = _patch(Jp, PAD, inx, iny, wj, W_BITS - 5) - Ipatch
                b1 = f32(int((diff * dIx).sum())) * scale_ab
                b2 = f32(int((diff * dIy).sum())) * scale_ab
                delta = np.array([(A12 * b2 - A22 * b1) * D, (A12 * b1 - A11 * b2) * D], np.float32)
                nextpt = nextpt + delta
                out[i] = nextpt + f32(HALF)
                if delta[0] * delta[0] + delta[1] * delta[1] <= f32(eps * eps):
                    break
                if j > 0 and abs(delta[0] + prev_delta[0]) < 0.01 and abs(delta[1] + prev_delta[1]) < 0.01:
                    out[i] = out[i] - delta * f32(0.5)
                    break
                prev_delta = delta
            if lvl == 0 and status[i]:
                npt = out[i] - f32(HALF)
                inx, iny = int(np.floor(npt[0])), int(np.floor(npt[1]))
                if inx < -WIN or inx >= W or iny < -WIN or iny >= H:
                    status[i] = 0
                    continue
                wj = _weights(npt[0] - f32(inx), npt[1] - f32(iny))
                diff = _patch(Jp, PAD, inx, iny, wj, W_BITS - 5) - Ipatch
                err[i] = f32(int(np.abs(diff).sum())) / f32(32 * WIN * WIN)
    return out, status, err


# ----------------------------------------------------------------------------- F4-F5 Shi-Tomasi
def mineig(img):
    """cornerMinEigenVal(blockSize 3, ksize 3): float32 Sobel with the 1/3060 scale folded into the smoothing taps,
    cov in float32, 3x3 un-normalised box in double, eig = (a + c) - sqrtf((a - c)^2 + b^2)."""
    f32 = np.float32
    p = _reflect101(np.asarray(img, np.uint8).astype(np.int32), 1)
    s = 1.0 / (4.0 * 3.0 * 255.0)
    k1, k2 = f32(s), f32(2.0 * s)
    # Dx: rows [-1 0 1] (exact integers), columns [s 2s s] as centre*2s + s*(pair sum)
    r = (p[:, 2:] - p[:, :-2]).astype(np.float32)
    dx = k2 * r[1:-1, :] + k1 * (r[:-2, :] + r[2:, :])
    # Dy: rows [s 2s s] (pair sum of uchar = exact integer), columns [-1 0 1]
    pf = p.astype(np.float32)
    rr = k2 * pf[:, 1:-1] + k1 * (pf[:, :-2] + pf[:, 2:])
    dy = rr[2:, :] - rr[:-2, :]
    dx, dy = dx.astype(np.float32), dy.astype(np.float32)
    cov = [(dx * dx).astype(np.float32), (dx * dy).astype(np.float32), (dy * dy).astype(np.float32)]
    # the box filter of the reference is separable (RowSum<float, double>, then ColumnSum<double, float>; REFLECT_101): three taps
    # of a row summed in double, left to right, then three row sums, top to bottom (round 5: the same order as oracle/fe_cpu.cpp)
    box = []
    for c in cov:
        q = _reflect101(c.astype(np.float64), 1)
        hh, ww = c.shape
        rs = (q[:, 0:ww] + q[:, 1:ww + 1]) + q[:, 2:ww + 2]
        acc = (rs[0:hh] + rs[1:hh + 1]) + rs[2:hh + 2]
        box.append(acc.astype(np.float32))
    a, b, cc = box[0] * f32(0.5), box[1], box[2] * f32(0.5)
    return ((a + cc) - np.sqrt((a - cc) * (a - cc) + b * b, dtype=np.float32)).astype(np.float32)


def gftt(img, max_corners, quality=0.01, min_dist=30.0, mask=None):
    """goodFeaturesToTrack: threshold at (float)(max * quality), 3x3 non-maximum test, sort by value (ties: larger
    linear index first), grid-accelerated minimum distance, integer coordinates in acceptance order."""
    eig = mineig(img)
    h, w = eig.shape
    m = np.ones((h, w), bool) if mask is None else (np.asarray(mask) != 0)
    if not m.any():
        return np.zeros((0, 2), np.float32)
    max_val = float(eig[m].max())
    thr = np.float32(max_val * quality)
    eig = np.where(eig > thr, eig, np.float32(0)).astype(np.float32)
    dil = ndimage.maximum_filter(eig, size=3, mode='constant', cval=-np.inf)
    cand = (eig != 0) & (eig == dil) & m
    cand[0, :] = cand[-1, :] = False
    cand[:, 0] = cand[:, -1] = False
    ys, xs = np.nonzero(cand)
    lin = ys * w + xs
    order = np.lexsort((-lin, -eig[ys, xs].astype(np.float64)))          # value descending, then larger index first
    out = []
    if min_dist >= 1:
        cell = int(np.rint(min_dist))
        gw, gh = (w + cell - 1) // cell, (h + cell - 1) // cell
        grid = [[[] for _ in range(gw)] for _ in range(gh)]
        md2 = np.float32(min_dist) * np.float32(min_dist)
        for k in order:
            x, y = int(xs[k]), int(ys[k])
            cx, cy = x // cell, y // cell
            good = True
            for yy in range(max(0, cy - 1), min(gh - 1, cy + 1) + 1):
                for xx in range(max(0, cx - 1), min(gw - 1, cx + 1) + 1):
                    for (px, py) in grid[yy][xx]:
                        ddx, ddy = np.float32(x - px), np.float32(y - py)
                        if ddx * ddx + ddy * ddy < md2:
                            good = False
                            break
                    if not good:
                        break
                if not good:
                    break
            if good:
                grid[cy][cx].append((x, y))
                out.append((x, y))
                if 0 < max_corners <= len(out):
                    break
    else:
        for k in order:
            out.append((int(xs[k]), int(ys[k])))
            if 0 < max_corners <= len(out):
                break
    return np.array(out, np.float32).reshape(-1, 2)


# ----------------------------------------------------------------------------- F6 CLAHE
def clahe(img, clip=3.0, tiles=(8, 8)):
    """createCLAHE(clip, tiles)->apply for images whose size divides into the tiles (752x480 -> 94x60)."""
    img = np.asarray(img, np.uint8)
    h, w = img.shape
    tx, ty = tiles
    tw, th = w // tx, h // ty
    assert tw * tx == w and th * ty == h, "restated for the exact-division case only"
    area = tw * th
    clip_limit = max(int(clip * area / 256), 1)
    lut_scale = np.float32(255.0) / np.float32(area)
    luts = np.zeros((ty, tx, 256), np.uint8)
    for j in range(ty):
        for i in range(tx):
            hist = np.bincount(img[j * th:(j + 1) * th, i * tw:(i + 1) * tw].ravel(), minlength=256).astype(np.int64)
            clipped = int(np.maximum(hist - clip_limit, 0).sum())
            hist = np.minimum(hist, clip_limit)
            hist += clipped // 256
            residual = clipped % 256
            if residual:
                step = max(256 // residual, 1)
                idx = np.arange(0, 256, step)[:residual]
                hist[idx] += 1
            cdf = np.cumsum(hist).astype(np.float32) * lut_scale
            luts[j, i] = np.clip(np.rint(cdf), 0, 255).astype(np.uint8)
    f32 = np.float32
    xs = np.arange(w, dtype=np.float32) * f32(1.0 / tw) - f32(0.5)
    ys = np.arange(h, dtype=np.float32) * f32(1.0 / th) - f32(0.5)
    tx1 = np.floor(xs).astype(np.int64)
    ty1 = np.floor(ys).astype(np.int64)
    xa = (xs - tx1.astype(np.float32)).astype(np.float32)
    ya = (ys - ty1.astype(np.float32)).astype(np.float32)
    tx2, ty2 = np.minimum(tx1 + 1, tx - 1), np.minimum(ty1 + 1, ty - 1)
    tx1, ty1 = np.maximum(tx1, 0), np.maximum(ty1, 0)
    v = img.astype(np.int64)
    Y1, X1 = ty1[:, None], tx1[None, :]
    Y2, X2 = ty2[:, None], tx2[None, :]
    l11 = luts[Y1, X1, v].astype(np.float32)
    l12 = luts[Y1, X2, v].astype(np.float32)
    l21 = luts[Y2, X1, v].astype(np.float32)
    l22 = luts[Y2, X2, v].astype(np.float32)
    XA, YA = xa[None, :], ya[:, None]
    one = f32(1.0)
    res = (l11 * (one - XA) + l12 * XA) * (one - YA) + (l21 * (one - XA) + l22 * XA) * YA
    return np.clip(np.rint(res.astype(np.float32)), 0, 255).astype(np.uint8)
