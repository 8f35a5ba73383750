// tests/cxx/mvo_oracle_shim.cpp - TEST INFRASTRUCTURE ONLY.  The per-call subset of include/mvo.h that include/mono_vo_hip.hpp's
// Backend binds, implemented over the CPU oracle (oracle/mvo_oracle.h) instead of the HIP library: it lets the C++ restatement
// of the reference's Tracker / Map / KeyFrame run on a CPU, where tests/test_cxx_tracker_over_oracle.py diffs it against the
// golden vectors made by the PYTHON restatement (ros2_mono_vo_amd/vo.py) over the same oracle - two independent readings of
// src/tracker.cpp checked against each other.  Never linked into the product.
#include <cstring>
#include <string>
#include <vector>

#include "mvo.h"
#include "mvo_oracle.h"

struct mvo_ctx { mvo_config cfg; std::string err; };

extern "C" {
void mvo_config_default(mvo_config* c) {
  std::memset(c, 0, sizeof(*c));
  c->max_width = 1280; c->max_height = 720; c->batch = 1; c->max_points = 8192; c->nfeatures = 1000; c->fast_threshold = 20;
  c->lk_channels = 3; c->lk_win = 21; c->lk_max_level = 3; c->lk_max_count = 30; c->lk_epsilon = 0.01; c->lk_min_eig = 1e-4;
  c->device = -1;
}
int mvo_create(const mvo_config* cfg, mvo_ctx** out) { *out = new mvo_ctx{*cfg, ""}; return MVO_OK; }
void mvo_destroy(mvo_ctx* ctx) { delete ctx; }
const char* mvo_last_error(const mvo_ctx* ctx) { return ctx ? ctx->err.c_str() : "null"; }

int mvo_orb_detect_and_compute(mvo_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels, mvo_keypoint* kps, uint8_t* desc, int cap, int* n) {
  static_assert(sizeof(mvo_keypoint) == sizeof(orc_keypoint), "cv::KeyPoint layout on both sides");
  const int cnt = orc_orb_detect_and_compute(img, w, h, stride, channels, ctx->cfg.nfeatures, ctx->cfg.fast_threshold, ctx->cfg.orb_blur_mode,
                                             (orc_keypoint*)kps, desc, cap);
  *n = cnt;
  return cnt > cap ? MVO_E_CAPACITY : MVO_OK;
}
int mvo_match_knn2_ratio(mvo_ctx*, const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio, mvo_match* out, int cap, int* n) {
  static_assert(sizeof(mvo_match) == sizeof(orc_match), "cv::DMatch layout on both sides");
  *n = orc_match_knn2_ratio(q, nq, t, nt, ratio, (orc_match*)out, cap);
  return MVO_OK;
}
int mvo_lk_track(mvo_ctx* ctx, const uint8_t* prev, const uint8_t* next, int w, int h, int stride, int channels, const float* prev_pts, int n,
                 float* next_pts, uint8_t* status, float* err) {
  if (channels != 1) {   // BGR8 / RGB8 / BGRA8 / RGBA8: the sums run over the three channels (identical channels give the one-plane result)
    const int bpp = channels < 0 ? -channels : channels;
    if (bpp != 3 && bpp != 4) { ctx->err = "channels must be 1, +-3 or +-4"; return MVO_E_ARG; }
    orc_lk_track_color(prev, next, w, h, stride, bpp, prev_pts, n, next_pts, status, err, ctx->cfg.lk_win, ctx->cfg.lk_max_level,
                       ctx->cfg.lk_max_count, ctx->cfg.lk_epsilon, ctx->cfg.lk_min_eig);
    return MVO_OK;
  }
  orc_lk_track(prev, next, w, h, stride, ctx->cfg.lk_channels, prev_pts, n, next_pts, status, err, ctx->cfg.lk_win, ctx->cfg.lk_max_level,
               ctx->cfg.lk_max_count, ctx->cfg.lk_epsilon, ctx->cfg.lk_min_eig);
  return MVO_OK;
}
static int count(const uint8_t* m, int n) { int c = 0; for (int i = 0; i < n; i++) c += m[i] != 0; return c; }
int mvo_find_homography_ransac(mvo_ctx* ctx, const float* p1, const float* p2, int n, double thr, int max_iters, double conf, uint8_t* mask, double H[9],
                               int* n_inliers) {
  int st[4] = {0, 0, 0, 0};
  const int r = orc_find_homography_ransac(p1, p2, n, thr, max_iters, conf, mask, H, st);
  if (r < 0) { ctx->err = "findHomography needs at least 4 correspondences"; return MVO_E_ARG; }
  *n_inliers = r > 0 ? count(mask, n) : 0;
  return r > 0 ? MVO_OK : MVO_E_DEGENERATE;
}
int mvo_find_fundamental_ransac(mvo_ctx*, const float* p1, const float* p2, int n, double thr, double conf, int max_iters, uint8_t* mask, double F[9],
                                int* n_inliers) {
  int st[4] = {0, 0, 0, 0};
  const int r = orc_find_fundamental_ransac(p1, p2, n, thr, conf, max_iters, mask, F, st);
  *n_inliers = r > 0 ? count(mask, n) : 0;
  return r > 0 ? MVO_OK : MVO_E_DEGENERATE;
}
int mvo_find_essential_ransac(mvo_ctx*, const float* p1, const float* p2, int n, const double K[9], double prob, double thr, int max_iters, uint8_t* mask,
                              double E[9], int* n_inliers) {
  int st[4] = {0, 0, 0, 0};
  const int r = orc_find_essential_ransac(p1, p2, n, K, prob, thr, max_iters, mask, E, st);
  *n_inliers = r > 0 ? count(mask, n) : 0;
  return r > 0 ? MVO_OK : MVO_E_DEGENERATE;
}
int mvo_recover_pose(mvo_ctx*, const double E[9], const float* p1, const float* p2, int n, const double K[9], double R[9], double t[3], uint8_t* mask_io,
                     int* n_good) {
  *n_good = orc_recover_pose(E, p1, p2, n, K, R, t, mask_io);
  return MVO_OK;
}
int mvo_solve_pnp_ransac(mvo_ctx* ctx, const float* obj, const float* img, int n, const double K[9], const double d[5], int iters, float reproj_err,
                         double confidence, double rvec[3], double tvec[3], int* inlier_idx, int* n_inliers) {
  int st[4] = {0, 0, 0, 0};
  const int r = orc_solve_pnp_ransac(obj, img, n, K, d, iters, reproj_err, confidence, rvec, tvec, inlier_idx, n_inliers, st);
  if (r < 0) { ctx->err = "solvePnPRansac: bad input"; return MVO_E_ARG; }
  if (r != 1) *n_inliers = 0;
  return r == 1 ? MVO_OK : MVO_E_DEGENERATE;
}
int mvo_triangulate(mvo_ctx*, const double P1[12], const double P2[12], const float* p1, const float* p2, int n, float* X3) {
  std::vector<float> x4((size_t)n * 4 + 4);
  orc_triangulate(P1, P2, p1, p2, n, X3, x4.data());
  return MVO_OK;
}
}
