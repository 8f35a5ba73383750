"""Rendered multi-plane sequence with true parallax for the VO state-machine tests (test helper)."""
import numpy as np

from ros2_mono_vo_amd import synth

W, H = 640, 480


def camera(k):
    R = synth.rot_y(-0.15 * k)
    c = np.array([0.25 * k, 0.02 * k, 0.03 * k])
    return R, -R @ c


_cache = {}


def frames(n):
    if n not in _cache:
        K = synth.default_K(W, H)
        planes = synth.make_plane_scene(7, scale=0.3)
        _cache[n] = [synth.render_planes(W, H, K, *camera(k), planes, seed=k)[0] for k in range(n)]
    return _cache[n]
