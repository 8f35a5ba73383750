// csrc/track_policy.h - the per-slot decisions of Tracker::update that ride at the end of another kernel's workgroup instead
// of being launches of their own (a 5 us kernel costs 0.3-0.7 ms of stream time when three other contexts keep the chip busy).
#pragma once
#include <cfloat>
#include <cmath>

#include "../../include/mvo.h"

// After the status/err filter: min_tracked_points -> LOST (src/tracker.cpp:292-296); the others go on to PnP.
struct TrkLostPolicy {
  int* state;   // null: no policy (the per-call API)
  long long min_tracked;
  int* n_pnp;
  int* flags;
  mvo_step_result* res;
};
__device__ __forceinline__ void trk_policy_lost_slot(const TrkLostPolicy& P, int s, int ncur) {
  int n = 0;
  if (P.state[s] == MVO_TRACK_TRACKING) {
    P.res[s].n_tracked = ncur;
    if ((long long)ncur < P.min_tracked) { P.state[s] = MVO_TRACK_LOST; P.flags[s] |= MVO_STEP_LOST_NOW; }
    else n = ncur;
  }
  P.n_pnp[s] = n;
}

__device__ inline void trk_rodrigues(const double r_[3], double R[9]) {   // calibration.cpp cvRodrigues2, vector -> matrix
  double rx = r_[0], ry = r_[1], rz = r_[2];
  const double theta = sqrt(rx * rx + ry * ry + rz * rz);
  if (theta < DBL_EPSILON) {
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1 : 0;
    return;
  }
  const double c = cos(theta), s = sin(theta), c1 = 1. - c, it = 1. / theta;
  rx *= it; ry *= it; rz *= it;
  const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
  const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
  for (int k = 0; k < 9; k++) R[k] = c * ((k % 4 == 0) ? 1. : 0.) + c1 * rrt[k] + s * r_x[k];
}

// After PnP: pose, ++tracking_count_from_keyframe_, should_add_keyframe (src/tracker.cpp:318-319, 118-136, 92-116).
// The reference does not look at solvePnPRansac's return value; without a model its pose for the frame is whatever an
// uninitialised 3x1 Mat holds (include/mvo.h, MVO_STEP_PNP_FAILED).  Defined here: no pose for the frame, the count still
// advances, no key-frame test, the stream keeps tracking (trk_finalize_kernel carries the LK survivors forward).
struct TrkKeyframePolicy {
  int* state;   // null: no policy (the per-call API)
  int* count;
  const int* n_pnp;
  const double* kf_pose;
  long long min_obs, max_after;
  double max_trans, max_rot;
  int policy;
  int* n_hf;
  int* flags;
  mvo_step_result* res;
  int* hf_ctr;   // [2] slot queues of the H and F launches that follow: zeroed by slot 0
};
__device__ inline void trk_policy_keyframe_slot(const TrkKeyframePolicy& P, int s, const int* pnp_result /* this slot's [8] */,
                                                const double* p /* this slot's rvec, tvec */) {
  if (s == 0) { P.hf_ctr[0] = 0; P.hf_ctr[1] = 0; }
  int nhf = 0;
  const int n = P.n_pnp[s];
  if (n > 0 && P.state[s] == MVO_TRACK_TRACKING) {
    const int ok = pnp_result[0] && pnp_result[6];
    P.res[s].pnp_ok = ok;
    P.res[s].n_pnp_inliers = pnp_result[5];
    if (!ok) {
      P.flags[s] |= MVO_STEP_PNP_FAILED;
      P.count[s] = P.count[s] + 1;
    } else {
      for (int k = 0; k < 3; k++) { P.res[s].rvec[k] = p[k]; P.res[s].tvec[k] = p[3 + k]; }
      P.flags[s] |= MVO_STEP_POSE;
      const int c = P.count[s] + 1;
      P.count[s] = c;
      bool add = (long long)n < P.min_obs || (long long)c > P.max_after;
      if (!add) {
        // has_significant_motion: relative pose kf_wc^-1 * cur_wc = T_kf_cw * T_cur_cw^-1
        double Rk[9], Rc[9];
        trk_rodrigues(P.kf_pose + 8 * s, Rk);
        trk_rodrigues(p, Rc);
        const double* tk = P.kf_pose + 8 * s + 3;
        double Rr[9];
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) Rr[3 * i + j] = Rk[3 * i] * Rc[3 * j] + Rk[3 * i + 1] * Rc[3 * j + 1] + Rk[3 * i + 2] * Rc[3 * j + 2];
        double tr[3];
        for (int i = 0; i < 3; i++) tr[i] = tk[i] - (Rr[3 * i] * p[3] + Rr[3 * i + 1] * p[4] + Rr[3 * i + 2] * p[5]);
        const double translation = sqrt(tr[0] * tr[0] + tr[1] * tr[1] + tr[2] * tr[2]);
        if (translation > P.max_trans) add = true;
        else {
          const double rotation = acos((Rr[0] + Rr[4] + Rr[8] - 1.0) / 2.0);   // NaN outside [-1, 1]: the test below is false
          add = rotation > P.max_rot;
        }
      }
      if (P.policy == 1) add = true;
      if (P.policy == 2) add = false;
      if (add) { nhf = n; P.flags[s] |= MVO_STEP_KF_CHECKED; }
    }
  }
  P.n_hf[s] = nhf;
}
