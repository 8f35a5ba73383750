"""Stage backend computed by the CPU oracle, with the method surface of ros2_mono_vo_amd.Context — TEST
INFRASTRUCTURE ONLY.  Lets the host mirror (ros2_mono_vo_amd/vo.py) run without a GPU and serves as the
end-to-end checker for the HIP backend."""
import numpy as np

import oracle_py as O
from ros2_mono_vo_amd._lib import KP_DTYPE, MATCH_DTYPE


class OracleBackend:
    def __init__(self, nfeatures=1000, lk_channels=3):
        self.nfeatures = nfeatures
        self.cn = lk_channels

    def orb_detect_and_compute(self, img):
        k, d = O.orb_detect_and_compute(img, self.nfeatures)
        return k.astype(KP_DTYPE), d

    def orb_detect(self, img):
        return self.orb_detect_and_compute(img)[0]

    def match_knn2_ratio(self, q, t, ratio=0.7):
        return O.match_knn2_ratio(q, t, ratio).astype(MATCH_DTYPE)

    def lk_track(self, prev, nxt, pts):
        if len(pts) == 0:
            return np.zeros((0, 2), np.float32), np.zeros(0, np.uint8), np.zeros(0, np.float32)
        return O.lk_track(prev, nxt, pts, cn=self.cn)

    def find_homography_ransac(self, p1, p2, thr=1.0, max_iters=2000, confidence=0.995):
        r, mask, H, _ = O.find_homography_ransac(p1, p2, thr, max_iters, confidence)
        if r < 0:
            raise ValueError("findHomography needs at least 4 correspondences")
        return r > 0, mask, H, max(r, 0)

    def find_fundamental_ransac(self, p1, p2, thr=1.0, confidence=0.99, max_iters=1000):
        r, mask, F, _ = O.find_fundamental_ransac(p1, p2, thr, confidence, max_iters)
        if r == -2:
            raise ValueError("LMedS branch not restated")
        return r > 0, mask, F, max(r, 0)

    def find_essential_ransac(self, p1, p2, K, prob=0.99, thr=1.0, max_iters=1000):
        r, mask, E, _ = O.find_essential_ransac(p1, p2, K, prob, thr, max_iters)
        return r > 0, mask, E, max(r, 0)

    def recover_pose(self, E, p1, p2, K, mask=None):
        return O.recover_pose(E, p1, p2, K, mask=mask)

    def solve_pnp_ransac(self, obj, img, K, d=None, iters=100, reproj=8.0, confidence=0.99):
        rc, r, t, idx, _ = O.solve_pnp_ransac(obj, img, K, d, iters, reproj, confidence)
        return rc == 1, r, t, idx

    def triangulate(self, P1, P2, p1, p2):
        if len(p1) == 0:
            return np.zeros((0, 3), np.float32)
        return O.triangulate(P1, P2, p1, p2)[0]
