"""Oracle-driven restatement of the frame-batch step (csrc/pipeline.hip) for one camera stream — TEST
INFRASTRUCTURE ONLY.  Same data flow as mvo_batch_track under policy 1 (key-frame branch on every tracked frame), every stage computed by the
CPU oracle: used by tests/test_gpu_pipeline.py as the end-to-end checker and by bench.py's cpu_baseline leg.

The flow is the reference's steady-state Tracker::update (src/tracker.cpp:274-333) with the key-frame
branch taken every frame: LK -> status/err filter -> solvePnPRansac -> findHomography/findFundamentalMat ->
ORB -> knn2+ratio vs the last key-frame -> triangulate + cheirality -> landmark hand-over -> new tracks."""
from __future__ import annotations

import numpy as np

import oracle_py as O


def _matmul_seq(A, B):
    """Row-by-column products accumulated left to right in IEEE double without FMA (the device / cv::gemm order)."""
    A = np.asarray(A, np.float64); B = np.asarray(B, np.float64)
    out = np.zeros((A.shape[0], B.shape[1]))
    for i in range(A.shape[0]):
        for j in range(B.shape[1]):
            acc = float(A[i, 0]) * float(B[0, j])
            for k in range(1, A.shape[1]):
                acc = acc + float(A[i, k]) * float(B[k, j])
            out[i, j] = acc
    return out


def _cam_z(T, X3):
    """z of T * (x,y,z,1) for float32 points, double accumulation left to right, rounded to float32."""
    x, y, z = (X3[:, k].astype(np.float64) for k in range(3))
    return (((T[2, 0] * x + T[2, 1] * y) + T[2, 2] * z) + T[2, 3]).astype(np.float32)


class StreamRef:
    def __init__(self, K, nfeatures=2000, cfg=None):
        self.K = np.asarray(K, np.float64).reshape(3, 3)
        self.nfeatures = nfeatures
        self.err_thresh = np.float32(30.0)
        self.ratio = 0.7
        self.thr = 1.0

    def seed(self, img, landmarks_fn):
        """ORB on the first frame; every key-point becomes a track with landmark landmarks_fn(xy)."""
        self.prev_img = img
        kps, desc = O.orb_detect_and_compute(img, self.nfeatures)
        xy = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
        lm = landmarks_fn(xy).astype(np.float32)
        self.kf_xy, self.kf_desc = xy, desc
        self.kf_has = np.ones(len(xy), bool)
        self.kf_lm = lm.copy()
        self.kf_pose = (np.zeros(3), np.zeros(3))
        self.trk_xy, self.trk_lm, self.trk_kf = xy.copy(), lm.copy(), xy.copy()
        return len(xy)

    def step(self, img):
        res = {}
        # LK + filter (src/tracker.cpp:68-77)
        nxt, st, err = O.lk_track(self.prev_img, img, self.trk_xy, cn=3)
        keep = (st != 0) & (err < self.err_thresh)
        cur_xy, cur_lm, cur_kf = nxt[keep], self.trk_lm[keep], self.trk_kf[keep]
        res["n_prev"], res["n_tracked"] = len(self.trk_xy), int(keep.sum())
        # PnP (src/tracker.cpp:309)
        rc, rvec, tvec, idx, _ = O.solve_pnp_ransac(cur_lm, cur_xy, self.K)
        res["pnp_ok"], res["n_pnp_inliers"], res["rvec"], res["tvec"] = rc == 1, len(idx), rvec, tvec
        # has_parallax (src/tracker.cpp:243-249)
        rh, _, _, _ = O.find_homography_ransac(cur_kf, cur_xy, self.thr, 2000, 0.995)
        rf, _, _, _ = O.find_fundamental_ransac(cur_kf, cur_xy, self.thr, 0.99, 1000)
        res["score_h"], res["score_f"] = max(rh, 0), max(rf, 0)
        # add_new_keyframe (src/tracker.cpp:182-235)
        kps, desc = O.orb_detect_and_compute(img, self.nfeatures)
        xy = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
        m = O.match_knn2_ratio(self.kf_desc, desc, self.ratio)
        res["n_keypoints"], res["n_matches"] = len(xy), len(m)
        pr, pc = self.kf_xy[m["query_idx"]], xy[m["train_idx"]]
        Rk, Rc = O.rodrigues(self.kf_pose[0]), O.rodrigues(rvec)
        Tk = np.hstack([Rk, np.asarray(self.kf_pose[1]).reshape(3, 1)])
        Tc = np.hstack([Rc, np.asarray(tvec).reshape(3, 1)])
        X3, _ = O.triangulate(_matmul_seq(self.K, Tk), _matmul_seq(self.K, Tc), pr, pc) if len(m) else (np.zeros((0, 3), np.float32), None)
        zr = _cam_z(Tk, X3) if len(m) else np.zeros(0, np.float32)
        zc = _cam_z(Tc, X3) if len(m) else np.zeros(0, np.float32)
        valid = (zr > 0) & (zc > 0) & bool(rc == 1)
        res["n_triangulated"] = int(valid.sum())
        cur_has = np.zeros(len(xy), bool)
        cur_lmk = np.zeros((len(xy), 3), np.float32)
        for i in range(len(m)):          # sequential: a later match overwrites (src/tracker.cpp:219-226)
            if valid[i]:
                q, t = m["query_idx"][i], m["train_idx"][i]
                cur_has[t] = True
                cur_lmk[t] = self.kf_lm[q] if self.kf_has[q] else X3[i]
        self.kf_xy, self.kf_desc, self.kf_has, self.kf_lm = xy, desc, cur_has, cur_lmk
        self.kf_pose = (rvec, tvec)
        self.trk_xy, self.trk_lm, self.trk_kf = xy[cur_has].copy(), cur_lmk[cur_has].copy(), xy[cur_has].copy()
        self.prev_img = img
        res["n_new_tracks"] = int(cur_has.sum())
        return res
