"""Synthetic image streams for the front-end drop-in tests: a window sliding (and slowly rotating) over a large textured canvas, sensor
noise per frame, and optionally a textured patch that moves on its own (its tracks violate the epipolar geometry of the rest: work for
rejectWithF).  Features leave at one border and new texture enters at the other, so masks, detection and new ids occur on every
published frame."""
import numpy as np

from vins_mono_amd import synth


def moving_scene(n, seed=5, width=752, height=480, velocity=(5.3, -2.1), rot_deg_per_frame=0.15, noise=2.0, patch=True):
    rng = np.random.default_rng(seed)
    margin = 64 + int(np.ceil(max(abs(velocity[0]), abs(velocity[1])) * n))
    cw, ch = (width + 2 * margin + 7) // 8 * 8, (height + 2 * margin + 7) // 8 * 8
    canvas = synth.synth_frame(seed, cw, ch).astype(np.float64)
    patch_tex = synth.synth_frame(seed + 1000, 96, 96).astype(np.float64)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
    frames = []
    for k in range(n):
        a = np.radians(rot_deg_per_frame * k)
        ox, oy = (cw - width) / 2.0 + velocity[0] * k, (ch - height) / 2.0 + velocity[1] * k
        xs = np.cos(a) * (xx - cx) - np.sin(a) * (yy - cy) + cx + ox
        ys = np.sin(a) * (xx - cx) + np.cos(a) * (yy - cy) + cy + oy
        x0, y0 = np.floor(xs).astype(int), np.floor(ys).astype(int)
        fx, fy = xs - x0, ys - y0
        img = canvas[y0, x0] * (1 - fy) * (1 - fx) + canvas[y0, x0 + 1] * (1 - fy) * fx + canvas[y0 + 1, x0] * fy * (1 - fx) + canvas[y0 + 1, x0 + 1] * fy * fx
        if patch:
            px, py = int(round(width * 0.3 + 3.0 * k)), int(round(height * 0.55 + 4.0 * k))
            if 0 <= px and px + 96 <= width and 0 <= py and py + 96 <= height:
                img[py:py + 96, px:px + 96] = patch_tex
        img = img + rng.normal(0, noise, img.shape)
        frames.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
    return frames
