#!/usr/bin/env python3
"""Generates tests/golden/lk_colour_v1.npz: calcOpticalFlowPyrLK on a true-colour (BGR8) pair, outputs of the CPU oracle
(orc_lk_track_color) on inputs that regenerate from seeds: golden_v1's frames as a three-channel image (a gain, the identity and a
tone curve of the gray value), its key-points plus border / outside points.  Provenance as make_golden.py: the oracle restates
OpenCV-4.6 semantics, real OpenCV is unavailable offline - "parity unpinned"; the vectors freeze the oracle's behaviour and the
HIP path is checked against them (tests/test_golden.py).

Run:  python tests/golden/make_golden_lk_colour.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_py as O  # noqa: E402


def tone(g):
    f = g.astype(np.float64)
    return np.stack([(f * 0.85).round(), f, 255.0 * (f / 255.0) ** 0.7], -1).round().clip(0, 255).astype(np.uint8)


def inputs():
    G = np.load(os.path.join(HERE, "golden_v1.npz"))
    extra = np.array([[0, 0], [319, 239], [-5, 10], [340, 100], [3.25, 230.75], [160.5, 120.5], [12, 12], [310, 5]], np.float32)
    return tone(G["frame0"]), tone(G["frame1"]), np.concatenate([G["lk_in"], extra]).astype(np.float32)


if __name__ == "__main__":
    a, b, pts = inputs()
    p, s, e = O.lk_track(a, b, pts)
    np.savez_compressed(os.path.join(HERE, "lk_colour_v1.npz"), pts=p, status=s, err=e)
    print("wrote lk_colour_v1.npz:", len(pts), "points,", int(s.sum()), "tracked")
