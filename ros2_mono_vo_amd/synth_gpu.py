"""Synthetic camera streams rendered on the GPU with torch — bench / test tooling, NOT part of the measured hot path.

bench.py needs hundreds of DISTINCT 1280x720 streams with true parallax (so that the reference's key-frame policy can
pass its has_parallax test, src/tracker.cpp:237-268); numpy rendering (synth.render_planes) would take hours.  Same
scene model as synth.make_plane_scene / render_planes: a textured back wall and a few nearer fronto-parallel billboards
seen from a moving camera, +-2 grey levels of noise; every stream has its own billboard layout, trajectory and noise.
Frames come out as uint8 tensors on the device together with the depth map of frame 0 (the seed landmarks)."""
from __future__ import annotations

import math

import numpy as np
import torch


def _value_noise(g: torch.Generator, h: int, w: int, cell: int, amp: float, dev) -> torch.Tensor:
    gh, gw = h // cell + 2, w // cell + 2
    grid = (torch.rand((1, 1, gh, gw), generator=g, device=dev) * 2 - 1)
    up = torch.nn.functional.interpolate(grid, size=(gh * cell, gw * cell), mode="bilinear", align_corners=False)
    return amp * up[0, 0, :h, :w]


def make_texture(h: int, w: int, seed: int, dev) -> torch.Tensor:
    """float32 [h, w] in 0..255: three octaves of value noise + a mosaic of flat grey blocks (corners for FAST)."""
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    t = torch.full((h, w), 128.0, device=dev)
    for cell, amp in ((64, 96.0), (16, 48.0), (4, 24.0)):
        t = t + _value_noise(g, h, w, cell, amp, dev)
    for cell, p in ((41, 0.10), (17, 0.12), (7, 0.10)):
        gh, gw = h // cell + 1, w // cell + 1
        on = torch.rand((gh, gw), generator=g, device=dev) < p
        val = torch.rand((gh, gw), generator=g, device=dev) * 255.0
        on = on.repeat_interleave(cell, 0).repeat_interleave(cell, 1)[:h, :w]
        val = val.repeat_interleave(cell, 0).repeat_interleave(cell, 1)[:h, :w]
        t = torch.where(on, val, t)
    return t.clamp(0, 255)


class SceneBank:
    """Shared textures (one big wall texture, a few billboard textures); streams differ in layout, motion and noise."""

    def __init__(self, dev, seed: int = 11, n_board_tex: int = 8, wall_hw=(3072, 4096), board_hw=(720, 960)):
        self.dev = dev
        self.wall = make_texture(wall_hw[0], wall_hw[1], seed, dev)
        self.boards = [make_texture(board_hw[0], board_hw[1], seed + 1 + i, dev) for i in range(n_board_tex)]

    def stream_params(self, seed: int, scale: float = 0.3):
        """Per-stream scene + motion: dict(planes=[(z, x0, x1, y0, y1, tex, ppm)], vel (3,), yaw_deg_per_frame)."""
        rng = np.random.default_rng(seed)
        planes = [(30.0 * scale, -60.0 * scale, 60.0 * scale, -40.0 * scale, 40.0 * scale, self.wall, self.wall.shape[1] / (120.0 * scale))]
        for i in range(6):
            # depths stratified over 8 .. 20 m (x scale): six billboards at nearly one depth in front of most of the view make
            # the scene a plane, and solvePnP(ITERATIVE)'s DLT start on near-planar landmarks is ill-conditioned - the
            # reference's pose is then chaotic (rvec ~ 1e9) and there is nothing to compare a pose with
            z = (8.0 + 12.0 * (i + float(rng.uniform(0.1, 0.9))) / 6.0) * scale
            cx, cy = float(rng.uniform(-8, 8)) * scale, float(rng.uniform(-5, 5)) * scale
            sw, sh = float(rng.uniform(2.5, 4.5)) * scale, float(rng.uniform(1.5, 3)) * scale
            tex = self.boards[int(rng.integers(len(self.boards)))]
            planes.append((z, cx - sw, cx + sw, cy - sh, cy + sh, tex, tex.shape[1] / (2 * sw)))
        ang = float(rng.uniform(0, 2 * math.pi))
        speed = float(rng.uniform(0.03, 0.06))                    # metres per frame, mostly sideways
        vel = np.array([speed * math.cos(ang), 0.35 * speed * math.sin(ang), float(rng.uniform(-0.012, 0.012))])
        return dict(planes=planes, vel=vel, yaw=float(rng.uniform(-0.06, 0.06)), noise_seed=int(rng.integers(1 << 31)))


def render_stream(bank: SceneBank, params, K: np.ndarray, w: int, h: int, n_frames: int):
    """-> (frames uint8 [n_frames, h, w] on the device, depth of frame 0 float32 [h, w] on the device)."""
    dev = bank.dev
    ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
    dx = (xs - float(K[0, 2])) / float(K[0, 0])
    dy = (ys - float(K[1, 2])) / float(K[1, 1])
    k = torch.arange(n_frames, device=dev, dtype=torch.float32).view(-1, 1, 1)
    yaw = torch.deg2rad(k * params["yaw"])
    c, s = torch.cos(yaw), torch.sin(yaw)
    # x_cam = R_cw (x_w - c_w), R_cw = rot_y(yaw): ray in world = R_cw^T d_cam
    dwx = c * dx - s * 1.0
    dwy = dy.expand(n_frames, h, w)
    dwz = s * dx + c * 1.0
    vel = params["vel"]
    cwx, cwy, cwz = k * float(vel[0]), k * float(vel[1]), k * float(vel[2])
    img = torch.zeros((n_frames, h, w), device=dev)
    depth = torch.full((n_frames, h, w), float("inf"), device=dev)
    for (z, x0, x1, y0, y1, tex, ppm) in params["planes"]:
        sdist = (z - cwz) / dwz
        X = cwx + sdist * dwx
        Y = cwy + sdist * dwy
        ok = (sdist > 0) & (X >= x0) & (X < x1) & (Y >= y0) & (Y < y1) & (sdist < depth)
        u = ((X - x0) * ppm).clamp(0, tex.shape[1] - 2)
        v = ((Y - y0) * ppm).clamp(0, tex.shape[0] - 2)
        ui, vi = u.long(), v.long()
        fx, fy = u - ui, v - vi
        val = (tex[vi, ui] * (1 - fx) + tex[vi, ui + 1] * fx) * (1 - fy) + (tex[vi + 1, ui] * (1 - fx) + tex[vi + 1, ui + 1] * fx) * fy
        img = torch.where(ok, val, img)
        depth = torch.where(ok, sdist, depth)
    g = torch.Generator(device=dev)
    g.manual_seed(params["noise_seed"])
    img = img + torch.randint(-2, 3, img.shape, generator=g, device=dev).float()
    # depth along the optical axis of camera 0 (yaw 0, at the origin): the ray parameter with d_cam.z = 1
    return img.round().clamp(0, 255).to(torch.uint8), depth[0]


from .synth import depth_landmarks  # noqa: E402,F401  (numpy only: the CPU-oracle workers use it without importing torch)
