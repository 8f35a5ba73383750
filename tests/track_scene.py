"""Rendered multi-plane sequences (true parallax) whose cameras move differently, so that the streams of one batch take
different branches of Tracker::update (test helper): steady lateral motion, a camera that only pans, a fast mover and a
stream whose content changes mid-way (tracking lost)."""
import numpy as np

from ros2_mono_vo_amd import synth

W, H = 640, 480
KINDS = ("lateral", "pan", "fast", "cut")


def camera(kind, k):
    if kind == "lateral":
        R = synth.rot_y(-0.05 * k)
        c = np.array([0.045 * k, 0.004 * k, 0.01 * k])
    elif kind == "pan":
        R = synth.rot_y(0.12 * k)
        c = np.zeros(3)
    elif kind == "fast":
        R = synth.rot_y(-0.1 * k)
        c = np.array([0.16 * k, 0.015 * k, 0.02 * k])
    else:
        R = synth.rot_y(-0.05 * k)
        c = np.array([0.05 * k, 0.0, 0.0])
    return R, -R @ c


_cache = {}


def stream(kind, n, seed=7):
    """-> (frames [n, H, W] uint8, depth of frame 0 [H, W])."""
    key = (kind, n, seed)
    if key not in _cache:
        K = synth.default_K(W, H)
        planes = synth.make_plane_scene(seed, scale=0.3)
        other = synth.make_plane_scene(seed + 100, scale=0.3)
        fr, d0 = [], None
        for k in range(n):
            pl = other if (kind == "cut" and k >= 9) else planes
            img, depth = synth.render_planes(W, H, K, *camera(kind, k), pl, seed=seed * 1000 + k)
            if k == 0:
                d0 = depth
            fr.append(img)
        _cache[key] = (np.stack(fr), d0)
    return _cache[key]


def depth_landmarks(K, depth):
    """Landmarks of frame-0 key-points from the renderer's depth (camera 0 = world): X = z * K^-1 (x, y, 1)."""
    def f(xy):
        xi = np.clip(np.rint(xy[:, 0]).astype(int), 0, depth.shape[1] - 1)
        yi = np.clip(np.rint(xy[:, 1]).astype(int), 0, depth.shape[0] - 1)
        z = depth[yi, xi]
        z = np.where(np.isfinite(z), z, 10.0).astype(np.float32)
        return np.stack([(xy[:, 0] - np.float32(K[0, 2])) / np.float32(K[0, 0]) * z,
                         (xy[:, 1] - np.float32(K[1, 2])) / np.float32(K[1, 1]) * z, z], 1).astype(np.float32)
    return f
