"""mvo_run - the ROS-free harness of the path ("what calls it").

Feeds frames through the same dispatch as MonoVO::image_callback (src/mono_vo.cpp:83-131: Initializer until it
succeeds, then Tracker), with every stage on the HIP path, and prints one line per frame plus, optionally, the
trajectory in the cv->ROS convention of src/utils.cpp:85-121 (TUM format: "t x y z qx qy qz qw").

Frame sources
  --raw FILE --width W --height H [--encoding mono8|bgr8|rgb8|bgra8|rgba8]   headerless frames, back to back
  --synthetic parallax|plane [--frames N]   the rendered multi-plane sequence (true parallax: initialises and tracks)
                                            or the similarity-warp stream of bench.py (planar: exercises the stages,
                                            the initializer keeps rejecting it for lack of parallax, as the reference would)

There is no CPU fallback: without the HIP library / a GPU the Context constructor raises.

    python -m ros2_mono_vo_amd.mvo_run --synthetic parallax --frames 12 --tum /tmp/traj.txt
"""
from __future__ import annotations

import argparse
import sys
import time

import numpy as np

from . import ros_io, synth, vo

CHANNELS = {"mono8": 1, "bgr8": 3, "rgb8": 3, "bgra8": 4, "rgba8": 4}


def raw_frames(path: str, width: int, height: int, encoding: str):
    """Headerless frames of width x height x channels(encoding) bytes; a trailing partial frame is an error."""
    ch = CHANNELS[encoding]
    nbytes = width * height * ch
    with open(path, "rb") as f:
        k = 0
        while True:
            buf = f.read(nbytes)
            if not buf:
                return
            if len(buf) != nbytes:
                raise ValueError(f"{path}: frame {k} is truncated ({len(buf)} of {nbytes} bytes)")
            a = np.frombuffer(buf, np.uint8)
            yield a.reshape(height, width) if ch == 1 else a.reshape(height, width, ch)
            k += 1


def to_bgr8(img: np.ndarray, encoding: str) -> np.ndarray:
    """cv_bridge::toCvCopy(msg, BGR8) of the node (src/mono_vo.cpp:92-100) for the colour encodings: channel reorder,
    alpha dropped.  mono8 stays single-channel (the device ingest replicates it exactly like the BGR8 conversion)."""
    if encoding in ("mono8", "bgr8"):
        return img
    if encoding == "rgb8":
        return np.ascontiguousarray(img[..., ::-1])
    if encoding == "bgra8":
        return np.ascontiguousarray(img[..., :3])
    if encoding == "rgba8":
        return np.ascontiguousarray(img[..., 2::-1])
    raise ValueError(encoding)


def synthetic_frames(kind: str, width: int, height: int, n: int, seed: int):
    if kind == "plane":
        yield from synth.gen_stream(width, height, seed, n)
        return
    K = synth.default_K(width, height)
    planes = synth.make_plane_scene(7, scale=0.3)
    for k in range(n):
        R = synth.rot_y(-0.15 * k)
        c = np.array([0.25 * k, 0.02 * k, 0.03 * k])
        yield synth.render_planes(width, height, K, R, -R @ c, planes, seed=k)[0]


def parse_args(argv=None):
    ap = argparse.ArgumentParser(prog="mvo_run", description=__doc__.split("\n\n")[0])
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--raw", metavar="FILE")
    src.add_argument("--synthetic", choices=("parallax", "plane"))
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--encoding", choices=sorted(CHANNELS), default="mono8")
    ap.add_argument("--frames", type=int, default=12, help="synthetic: frames to render; raw: stop after this many (0 = all)")
    ap.add_argument("--seed", type=lambda v: int(v, 0), default=0x5EED0002)
    ap.add_argument("--nfeatures", type=int, default=1000, help="ORB features (the node hard-codes 1000, src/mono_vo.cpp:16)")
    ap.add_argument("--intrinsics", type=float, nargs=4, metavar=("FX", "FY", "CX", "CY"),
                    help="default: fx = fy = 0.9 W, principal point at the centre")
    ap.add_argument("--distortion", type=float, nargs=5, metavar=("K1", "K2", "P1", "P2", "K3"), default=[0.0] * 5)
    ap.add_argument("--fps", type=float, default=30.0, help="time stamps of the trajectory file")
    ap.add_argument("--tum", metavar="FILE", help="write the trajectory (cv->ROS convention) in TUM format")
    ap.add_argument("--device", type=int, default=0)
    return ap.parse_args(argv)


def intrinsics(args) -> np.ndarray:
    if args.intrinsics is None:
        return synth.default_K(args.width, args.height)
    fx, fy, cx, cy = args.intrinsics
    return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)


def main(argv=None) -> int:
    args = parse_args(argv)
    from .context import Context  # raises when the HIP library or the GPU is missing

    K = intrinsics(args)
    d = np.asarray(args.distortion, np.float64)
    if args.raw:
        frames = raw_frames(args.raw, args.width, args.height, args.encoding)
    else:
        frames = synthetic_frames(args.synthetic, args.width, args.height, args.frames, args.seed)
    path = ros_io.PathAccumulator()
    traj = []
    with Context(max_width=args.width, max_height=args.height, nfeatures=args.nfeatures, max_points=4096,
                 device=args.device) as ctx:
        odo = vo.VisualOdometry(ctx, K, d, nfeatures=args.nfeatures)
        for k, img in enumerate(frames):
            if args.raw and args.frames and k >= args.frames:
                break
            img = to_bgr8(img, args.encoding)
            t0 = time.perf_counter()
            pose = odo.process(img)
            ms = (time.perf_counter() - t0) * 1e3
            init, trk = odo.initializer, odo.tracker
            line = f"frame {k:5d}  init={init.state.name:<12s} tracker={trk.state.name:<12s} {ms:7.2f} ms"
            if pose is not None:
                msg = path.push(pose[:3, :3], pose[:3, 3], k / args.fps)["poses"][-1]
                p_ros, q_ros = msg["position"], msg["orientation"]
                traj.append((k / args.fps, p_ros, q_ros))
                line += (f"  tracked={trk.last.get('n_tracked', 0):4d} pnp_inliers={trk.last.get('n_pnp_inliers', 0):4d}"
                         f"  p_ros=({p_ros[0]:+.4f},{p_ros[1]:+.4f},{p_ros[2]:+.4f})")
            print(line, flush=True)
        print(f"key-frames {len(odo.map.keyframes)}  landmarks {len(odo.map.landmarks)}  poses {len(traj)}", flush=True)
    if args.tum:
        with open(args.tum, "w") as f:
            for t, p, q in traj:
                f.write(f"{t:.6f} {p[0]:.9f} {p[1]:.9f} {p[2]:.9f} {q[0]:.9f} {q[1]:.9f} {q[2]:.9f} {q[3]:.9f}\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
