#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.npz: small input/output vectors for every stage of the hot path.

Provenance: the reference (Tatsuya-2/ros2_mono_vo) ships no tests, fixtures or golden vectors and its arithmetic
lives in un-vendored OpenCV, which is not available offline.  These vectors are therefore produced by the CPU
oracle (oracle/, a restatement of OpenCV-4.6 semantics) on seeded synthetic inputs — "parity unpinned" with
respect to real OpenCV.  They freeze the oracle's behaviour so that (a) oracle regressions and (b) HIP-path
regressions are both caught against committed data.  Inputs are regenerated from seeds by the tests and checked
against the stored copies, so the fixture is self-describing.

Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_py as O  # noqa: E402
from ros2_mono_vo_amd import synth  # noqa: E402

W, H, NF = 320, 240, 300


def build():
    g = {}
    fr = synth.gen_stream(W, H, 0x601D, 2)
    g["frame0"], g["frame1"] = fr[0], fr[1]
    g["pyrdown0"] = O.pyrdown(fr[0])
    g["fast0"] = O.fast9_nms(fr[0], 20)
    k0, d0 = O.orb_detect_and_compute(fr[0], NF)
    k1, d1 = O.orb_detect_and_compute(fr[1], NF)
    for name, k, d in (("0", k0, d0), ("1", k1, d1)):
        g["orb_kp" + name] = np.stack([k[f].astype(np.float64) for f in ("x", "y", "size", "angle", "response", "octave", "class_id")], 1)
        g["orb_desc" + name] = d
    pts = np.stack([k0["x"], k0["y"]], 1).astype(np.float32)
    p, s, e = O.lk_track(fr[0], fr[1], pts, cn=3)
    g["lk_in"], g["lk_pts"], g["lk_status"], g["lk_err"] = pts, p, s, e
    m = O.match_knn2_ratio(d0, d1, 0.7)
    g["match"] = np.stack([m["query_idx"], m["train_idx"], m["distance"].astype(np.int32)], 1)
    sc = synth.gen_scene(200, 0xC0FFEE00 + 200, w=W, h=H)
    g["sc_X"], g["sc_p1"], g["sc_p2"], g["sc_K"] = sc["X"], sc["p1"], sc["p2"], sc["K"]
    r, mask, Hm, st = O.find_homography_ransac(sc["p1"], sc["p2"], 1.0)
    g["h_mask"], g["h_model"], g["h_stats"] = mask, Hm, st
    r, mask, F, st = O.find_fundamental_ransac(sc["p1"], sc["p2"], 1.0, 0.99)
    g["f_mask"], g["f_model"], g["f_stats"] = mask, F, st
    rc, rv, tv, idx, st = O.solve_pnp_ransac(sc["X"], sc["p2"], sc["K"])
    g["pnp_rvec"], g["pnp_tvec"], g["pnp_inliers"], g["pnp_stats"] = rv, tv, idx, st
    r, mask, E, st = O.find_essential_ransac(sc["p1"], sc["p2"], sc["K"], 0.99, 1.0)
    g["e_mask"], g["e_model"], g["e_stats"] = mask, E, st
    gg, R, t, m2 = O.recover_pose(E, sc["p1"], sc["p2"], sc["K"], mask=mask)
    g["rp_R"], g["rp_t"], g["rp_mask"], g["rp_good"] = R, t, m2, np.array([gg])
    P1 = sc["K"] @ np.hstack([np.eye(3), np.zeros((3, 1))])
    P2 = sc["K"] @ np.hstack([O.rodrigues(rv), tv[:, None]])
    g["tri_P1"], g["tri_P2"] = P1, P2
    g["tri_X3"], _ = O.triangulate(P1, P2, sc["p1"], sc["p2"])
    g["rng"] = np.array(O.rng_sequence(0xFFFFFFFFFFFFFFFF, 16), np.uint32)
    return g


if __name__ == "__main__":
    g = build()
    out = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(out, **g)
    print(out, os.path.getsize(out), "bytes;", {k: v.shape for k, v in g.items()})
