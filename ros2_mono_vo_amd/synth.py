"""Seeded synthetic inputs for parity tests and bench.py.

* ``gen_stream``  — mono8 image streams: value-noise + random rectangles texture seen through a smooth
  similarity warp, plus +-2 grey levels of uniform noise (the "Synthetic WxH mono8 stream" configs of
  BASELINE.json).
* ``gen_scene`` / ``project`` — 3-D point sets with a known camera motion, pixel noise and outliers, for
  the RANSAC / PnP / recoverPose stages.
* ``render_planes`` — a small multi-plane 3-D scene renderer (true parallax) so the Initializer /
  Tracker state machines can run end to end on images.

Everything is numpy; nothing here is part of the measured hot path.
"""
from __future__ import annotations

import numpy as np


def _value_noise(rng: np.random.Generator, h: int, w: int, cell: int, amp: float) -> np.ndarray:
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.uniform(-1.0, 1.0, size=(gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = ys.astype(np.int32)
    x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)


def base_texture(w: int, h: int, seed: int, margin: int = 96) -> np.ndarray:
    """Float32 texture of size (h+2m, w+2m): 3 octaves of value noise + grey rectangles."""
    rng = np.random.default_rng(seed)
    H, W = h + 2 * margin, w + 2 * margin
    t = np.full((H, W), 128.0, np.float32)
    for cell, amp in ((64, 96.0), (16, 48.0), (4, 24.0)):
        t += _value_noise(rng, H, W, cell, amp)
    nrect = int(400 * (w * h) / 307200)
    for _ in range(nrect):
        rw, rh = rng.integers(6, 48, size=2)
        x = rng.integers(0, W - rw)
        y = rng.integers(0, H - rh)
        t[y:y + rh, x:x + rw] = rng.integers(0, 256)
    return np.clip(t, 0, 255)


def warp_frame(tex: np.ndarray, w: int, h: int, k: int, seed: int, margin: int = 96,
               step=(1.5, 0.5, 0.05, 0.0005)) -> np.ndarray:
    """Frame k of the stream: similarity warp (tx,ty px, rot deg, zoom per frame) + +-2 noise."""
    tx, ty, rot, zoom = step[0] * k, step[1] * k, np.deg2rad(step[2] * k), 1.0 + step[3] * k
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    cx, cy = w / 2.0, h / 2.0
    c, s = np.cos(rot) * zoom, np.sin(rot) * zoom
    u = c * (xs - cx) - s * (ys - cy) + cx + tx + margin
    v = s * (xs - cx) + c * (ys - cy) + cy + ty + margin
    u = np.clip(u, 0, tex.shape[1] - 2)
    v = np.clip(v, 0, tex.shape[0] - 2)
    x0 = u.astype(np.int32)
    y0 = v.astype(np.int32)
    fx = u - x0
    fy = v - y0
    img = (tex[y0, x0] * (1 - fx) + tex[y0, x0 + 1] * fx) * (1 - fy) + \
          (tex[y0 + 1, x0] * (1 - fx) + tex[y0 + 1, x0 + 1] * fx) * fy
    rng = np.random.default_rng(seed * 1000003 + k)
    img = img + rng.integers(-2, 3, size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def gen_stream(w: int, h: int, seed: int, n_frames: int, step=(1.5, 0.5, 0.05, 0.0005)) -> np.ndarray:
    """uint8 [n_frames, h, w]."""
    tex = base_texture(w, h, seed)
    return np.stack([warp_frame(tex, w, h, k, seed, step=step) for k in range(n_frames)])


# ---------------------------------------------------------------------------------------------------
def default_K(w: int, h: int) -> np.ndarray:
    return np.array([[0.9 * w, 0, w / 2.0], [0, 0.9 * w, h / 2.0], [0, 0, 1]], np.float64)


def rot_y(deg: float) -> np.ndarray:
    a = np.deg2rad(deg)
    return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float64)


def gen_scene(P: int, seed: int, w: int = 1280, h: int = 720, planar: bool = False,
              outlier_frac: float = 0.2, noise_px: float = 0.5, steps: int = 1):
    """P 3-D points in the frustum of camera 0 and their noisy projections in camera 0 and camera
    `steps` (motion per step t=(0.05,0.01,0.30) m, 0.5 deg about y).  Returns a dict with X (P,3) f32,
    p1, p2 (P,2) f32, K, R, t (X_cam2 = R X + t), inlier (P,) bool."""
    rng = np.random.default_rng(seed)
    K = default_K(w, h)
    z = np.full(P, 10.0) if planar else rng.uniform(4.0, 40.0, P)
    u = rng.uniform(0, w, P)
    v = rng.uniform(0, h, P)
    X = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], 1)
    R = np.eye(3)
    t = np.zeros(3)
    for _ in range(steps):
        Rs = rot_y(0.5)
        ts = -Rs @ np.array([0.05, 0.01, 0.30])
        R = Rs @ R
        t = Rs @ t + ts
    X2 = X @ R.T + t
    p1 = (X @ K.T)
    p1 = p1[:, :2] / p1[:, 2:]
    p2 = (X2 @ K.T)
    p2 = p2[:, :2] / p2[:, 2:]
    p1 = p1 + rng.normal(0, noise_px, p1.shape)
    p2 = p2 + rng.normal(0, noise_px, p2.shape)
    inl = np.ones(P, bool)
    nout = int(round(outlier_frac * P))
    if nout:
        idx = rng.choice(P, nout, replace=False)
        p2[idx] = np.stack([rng.uniform(0, w, nout), rng.uniform(0, h, nout)], 1)
        inl[idx] = False
    return dict(X=X.astype(np.float32), p1=p1.astype(np.float32), p2=p2.astype(np.float32), K=K, R=R, t=t,
                inlier=inl, w=w, h=h)


# ---------------------------------------------------------------------------------------------------
def render_planes(w: int, h: int, K: np.ndarray, R_cw: np.ndarray, t_cw: np.ndarray, planes, seed: int = 0):
    """Render fronto-parallel textured planes (world z = const, infinite extent, nearest wins inside
    its rectangle) seen from camera pose (R_cw, t_cw): x_cam = R_cw x_w + t_cw.
    planes: list of dict(z, x0, x1, y0, y1, tex (float32 HxW), ppm (texels per metre))."""
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    d_cam = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs)], -1)
    R_wc = R_cw.T
    c_w = -R_wc @ t_cw
    d_w = d_cam @ R_wc.T
    img = np.zeros((h, w), np.float32)
    depth = np.full((h, w), np.inf)
    for pl in planes:
        s = (pl["z"] - c_w[2]) / d_w[..., 2]
        X = c_w[0] + s * d_w[..., 0]
        Y = c_w[1] + s * d_w[..., 1]
        ok = (s > 0) & (X >= pl["x0"]) & (X < pl["x1"]) & (Y >= pl["y0"]) & (Y < pl["y1"]) & (s < depth)
        tex = pl["tex"]
        u = np.clip((X - pl["x0"]) * pl["ppm"], 0, tex.shape[1] - 2)
        v = np.clip((Y - pl["y0"]) * pl["ppm"], 0, tex.shape[0] - 2)
        x0 = u.astype(np.int32)
        y0 = v.astype(np.int32)
        fx = (u - x0).astype(np.float32)
        fy = (v - y0).astype(np.float32)
        val = (tex[y0, x0] * (1 - fx) + tex[y0, x0 + 1] * fx) * (1 - fy) + \
              (tex[y0 + 1, x0] * (1 - fx) + tex[y0 + 1, x0 + 1] * fx) * fy
        img = np.where(ok, val, img)
        depth = np.where(ok, s, depth)
    rng = np.random.default_rng(seed)
    img = img + rng.integers(-2, 3, size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8), depth


def make_plane_scene(seed: int = 7, scale: float = 1.0):
    """A background wall + a few nearer billboards: genuine parallax for the Initializer.  `scale` shrinks the
    whole scene (depths 30 / 8-20 m at scale 1)."""
    planes = []
    tex = base_texture(2048, 1536, seed, margin=0)
    planes.append(dict(z=30.0 * scale, x0=-60.0 * scale, x1=60.0 * scale, y0=-40.0 * scale, y1=40.0 * scale, tex=tex,
                       ppm=tex.shape[1] / (120.0 * scale)))
    rng = np.random.default_rng(seed + 1)
    for i in range(6):
        z = float(rng.uniform(8.0, 20.0)) * scale
        cx = float(rng.uniform(-8, 8)) * scale
        cy = float(rng.uniform(-5, 5)) * scale
        sw, sh = float(rng.uniform(3, 6)) * scale, float(rng.uniform(2, 4)) * scale
        t = base_texture(640, 480, seed + 10 + i, margin=0)
        planes.append(dict(z=z, x0=cx - sw, x1=cx + sw, y0=cy - sh, y1=cy + sh, tex=t, ppm=t.shape[1] / (2 * sw)))
    return planes


def depth_landmarks(K: np.ndarray, depth0: np.ndarray, xy: np.ndarray) -> np.ndarray:
    """Landmarks of frame-0 key-points from a renderer's depth map (camera 0 = world): X = z K^-1 (x, y, 1), float32."""
    xi = np.clip(np.rint(xy[:, 0]).astype(int), 0, depth0.shape[1] - 1)
    yi = np.clip(np.rint(xy[:, 1]).astype(int), 0, depth0.shape[0] - 1)
    z = depth0[yi, xi]
    z = np.where(np.isfinite(z), z, 10.0).astype(np.float32)
    return np.stack([(xy[:, 0] - np.float32(K[0, 2])) / np.float32(K[0, 0]) * z,
                     (xy[:, 1] - np.float32(K[1, 2])) / np.float32(K[1, 1]) * z, z], 1).astype(np.float32)
