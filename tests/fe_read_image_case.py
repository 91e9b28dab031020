"""vg_fe_read_image (one library call per frame) against the SAME frame composed from the step-by-step entry points of the C-ABI on a
second handle -- vg_fe_push_frames, vg_fe_track, the border test, liftProjective in double (NumPy), vg_fe_reject_with_f (host-made
exact sample schedule + host bookkeeping), vg_fe_set_mask, vg_fe_detect_masked, vg_fe_undistort -- i.e. the path the drop-in took until
round 5, which tests/test_fe_gpu.py / test_simt_fe.py hold to the oracle call by call.  Bit-identical statuses, positions, kept lists, corners
and lifted points are required, on streams chosen to reach every branch of the call:

  normal       RANSAC on the device (resident schedule, device bookkeeping), walk order from the callback
  lmeds        8 <= survivors < 15: the estimate goes back to the host inside the call (RI_FB_LMEDS)
  collinear    cur_pts on image rows + a camera without distortion: samples OpenCV would redraw (RI_FB_COLLINEAR)
  few          fewer than 8 survivors: no rejectWithF; no survivor at all; no point at all (first frame)
  no-callback  order == NULL walks the list as it stands = the callback returning the identity
  fisheye      a base mask under setMask

Used by tests/test_fe_read_image.py under the emulator (`not gpu`) and on the device (`gpu`)."""
import numpy as np

from vins_mono_amd import fe

import fe_scene

INTR = (196.4, 195.9, 154.5, 124.0, -2.917e-01, 8.228e-02, 5.333e-05, -1.578e-04)       # at 320 x 240 (scaled with the frame)
INTR_PLAIN = (460.0, 460.0, 160.0, 120.0, 0.0, 0.0, 0.0, 0.0)
FOCAL = 460.0


def lift64(pts, intr):
    """PinholeCamera::liftProjective (PinholeCamera.cc:450-510) in double, as rejectWithF uses it (feature_tracker.cpp:176-187)"""
    fx, fy, cx, cy, k1, k2, p1, p2 = [np.float64(v) for v in intr]
    p = np.asarray(pts, np.float32).astype(np.float64).reshape(-1, 2)
    mx_d = (1.0 / fx) * p[:, 0] + (-cx / fx)
    my_d = (1.0 / fy) * p[:, 1] + (-cy / fy)
    mx_u, my_u = mx_d.copy(), my_d.copy()
    for _ in range(8):
        mx2, my2, mxy = mx_u * mx_u, my_u * my_u, mx_u * my_u
        rho2 = mx2 + my2
        rad = k1 * rho2 + k2 * rho2 * rho2
        dx = mx_u * rad + 2.0 * p1 * mxy + p2 * (rho2 + 2.0 * mx2)
        dy = my_u * rad + 2.0 * p2 * mxy + p1 * (rho2 + 2.0 * my2)
        mx_u, my_u = mx_d - dx, my_d - dy
    return mx_u, my_u


def stepwise(tr, W, H, img, cur, cnt, publish, intr, max_cnt, min_dist, equalize, base_mask, order_fn):
    """one frame from the fine-grained calls; returns the same dictionary as FrontEnd.read_image (+ the new count list)"""
    tr.push_frames([img], equalize=equalize)
    cur = np.asarray(cur, np.float32).reshape(-1, 2)
    n = len(cur)
    out = dict(ransac_ran=False, status_f=None, kept=None, new_xy=None)
    if n:
        forw, st, _ = tr.track(0, cur)
        ix, iy = np.rint(forw[:, 0].astype(np.float64)), np.rint(forw[:, 1].astype(np.float64))          # cvRound
        st = (st != 0) & (1 <= ix) & (ix < W - 1) & (1 <= iy) & (iy < H - 1)
    else:
        forw, st = np.zeros((0, 2), np.float32), np.zeros(0, bool)
    out["status_lk"], out["forw_xy"] = st.astype(np.uint8), forw
    cur1, forw1, cnt1 = cur[st], forw[st], np.asarray(cnt, np.int64)[st] + 1
    out["n1"] = out["n2"] = len(forw1)
    if not publish:
        out["n_final"] = len(forw1)
        out["un_xy"] = tr.undistort(forw1, intr) if len(forw1) else np.zeros((0, 2), np.float32)
        return out, forw1, cnt1
    if len(forw1) >= 8:
        cx, cy = lift64(cur1, intr)
        fx, fy = lift64(forw1, intr)
        p1 = np.stack([FOCAL * cx / 1.0 + W / 2.0, FOCAL * cy / 1.0 + H / 2.0], 1).astype(np.float32)
        p2 = np.stack([FOCAL * fx / 1.0 + W / 2.0, FOCAL * fy / 1.0 + H / 2.0], 1).astype(np.float32)
        sf, _ = tr.reject_with_f(p1, p2, 1.0)
        out["ransac_ran"], out["status_f"] = True, sf
        keep = sf != 0
        forw1, cnt1 = forw1[keep], cnt1[keep]
        out["n2"] = len(forw1)
    order = np.asarray(order_fn(cnt1), np.int64) if len(forw1) else np.zeros(0, np.int64)
    pts_o, cnt_o = forw1[order], cnt1[order]
    # (strictly decreasing counts: vg_fe_set_mask's own stable sort is then the identity and the walk follows `order`)
    kept = tr.set_mask([pts_o], [np.arange(len(cnt_o), 0, -1)], min_dist, base_masks=None if base_mask is None else [base_mask])[0]
    room = max_cnt - len(kept)
    new = tr.detect_masked(0, room, 0.01, float(min_dist)) if room > 0 else np.zeros((0, 2), np.float32)
    final = np.concatenate([pts_o[kept], new]) if len(kept) + len(new) else np.zeros((0, 2), np.float32)
    out.update(kept=np.asarray(kept, np.int32), new_xy=new, n_kept=len(kept), n_new=len(new), n_final=len(final))
    out["un_xy"] = tr.undistort(final, intr) if len(final) else np.zeros((0, 2), np.float32)
    return out, final, np.concatenate([cnt_o[kept], np.ones(len(new), np.int64)])


def _same(a, b, what):
    assert a["n1"] == b["n1"] and a["n2"] == b["n2"] and a["ransac_ran"] == b["ransac_ran"] and a["n_final"] == b["n_final"], (what, {k: (a[k], b[k]) for k in ("n1", "n2", "ransac_ran", "n_final")})
    assert np.array_equal(a["status_lk"], b["status_lk"]), what
    assert np.array_equal(a["forw_xy"].view(np.uint32), b["forw_xy"].view(np.uint32)), what
    if a["ransac_ran"]:
        assert np.array_equal(a["status_f"], b["status_f"]), (what, a["status_f"], b["status_f"])
    if b["kept"] is not None:
        assert np.array_equal(a["kept"], b["kept"]), (what, a["kept"], b["kept"])
        assert np.array_equal(a["new_xy"].view(np.uint32), b["new_xy"].view(np.uint32)), what
    assert np.array_equal(a["un_xy"].view(np.uint32), b["un_xy"].view(np.uint32)), what


def run(handle_a, handle_b, W=320, H=240, n_frames=7):
    """returns what happened (for the caller's assertions on coverage)"""
    cap = 160
    one, ref = fe.FrontEnd(handle_a, W, H, 1, cap), fe.FrontEnd(handle_b, W, H, 1, cap)
    seen = dict(ransac_device=0, fb_lmeds=0, fb_collinear=0, no_ransac=0, published=0, niters=[])
    rng = np.random.default_rng(11)

    def unstable_like(cnt):
        """a walk order that is NOT the stable one: equal counts in reversed order (what an unstable sort may do)"""
        c = np.asarray(cnt)
        return np.lexsort((-np.arange(len(c)), -c))

    def stream(name, frames, intr, max_cnt, min_dist, equalize, base_mask=None, first_pts=None, callback=True, order_fn=unstable_like, pub=lambda k: k % 2 == 0):
        pts = np.zeros((0, 2), np.float32) if first_pts is None else np.asarray(first_pts, np.float32)
        cnt = np.ones(len(pts), np.int64)
        for k, img in enumerate(frames):
            publish = bool(pub(k))
            box = {}

            def cb(st, sf, fw, n2):
                c = cnt[st != 0] + 1
                if sf is not None:
                    c = c[sf != 0]
                assert len(c) == n2
                box["order"] = order_fn(c)
                return box["order"]

            got = one.read_image(img, pts, publish, intr, max_cnt=max_cnt, min_dist=min_dist, equalize=equalize, base_mask=base_mask,
                                 order=cb if callback else None)
            want, pts_next, cnt_next = stepwise(ref, W, H, img, pts, cnt, publish, intr, max_cnt, min_dist, equalize, base_mask,
                                                order_fn if callback else (lambda c: np.arange(len(c))))
            _same(got, want, (name, k))
            if publish:
                seen["published"] += 1
                if got["ransac_ran"]:
                    if got["fallback"] & 2: seen["fb_lmeds"] += 1
                    elif got["fallback"] & 1: seen["fb_collinear"] += 1
                    else:
                        seen["ransac_device"] += 1
                        seen["niters"].append(got["ransac_niters"])
                else:
                    seen["no_ransac"] += 1
            pts, cnt = pts_next, cnt_next
        return pts

    frames = fe_scene.moving_scene(n_frames, seed=4, width=W, height=H, velocity=(3.1, -1.4))
    yy, xx = np.mgrid[0:H, 0:W]
    fish = np.where((xx - W / 2) ** 2 + (yy - H / 2) ** 2 < (0.55 * H) ** 2, 255, 0).astype(np.uint8)
    sc = W / 320.0                                               # (distances scale with the frame: the detection's cell grid holds 1024 cells)
    md = lambda v: int(round(v * sc))
    intr = tuple(v * (sc if i < 4 else 1.0) for i, v in enumerate(INTR))
    plain = tuple(v * (sc if i < 4 else 1.0) for i, v in enumerate(INTR_PLAIN))
    stream("normal", frames, intr, 60, md(14), True)
    stream("fisheye", frames[:5], intr, 50, md(12), False, base_mask=fish)
    stream("no-callback", frames[:5], intr, 40, md(16), True, callback=False)
    stream("lmeds", frames[:5], intr, 12, md(30), True)
    stream("few", frames[:4], intr, 5, md(40), False)
    # cur_pts on four image rows, a camera without distortion: three points of a row stay exactly collinear after the lifting
    grid = np.array([[x * sc, y * sc] for y in (50.0, 90.0, 130.0, 170.0) for x in np.arange(30.0, 290.0, 20.0)], np.float32)
    stream("collinear", frames[:3], plain, 70, md(10), False, first_pts=grid, pub=lambda k: True)
    # points that all leave the image / fail: nothing survives the tracking on a published frame
    stream("none", frames[:2], intr, 30, md(20), False, first_pts=np.array([[0.2, 0.3], [W - 0.6, H - 0.7], [0.4, H - 0.8]], np.float32), pub=lambda k: True)
    return seen
