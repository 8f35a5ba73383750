// oracle/orc_geom.cpp — TEST INFRASTRUCTURE ONLY (see orc_common.h).  PARITY UNPINNED.
//
// CPU restatement of the two-view geometry calls of the reference (OpenCV 4.6 calib3d semantics,
// SURVEY.md A.5, A.6, A.8):
//   cv::findHomography(p1,p2,RANSAC,thr,mask)              src/tracker.cpp:243, src/initializer.cpp:82
//   cv::findFundamentalMat(p1,p2,FM_RANSAC,thr,0.99,mask)  src/tracker.cpp:248, src/initializer.cpp:87
//   cv::triangulatePoints + convertPointsFromHomogeneous    src/tracker.cpp:149-152, src/initializer.cpp:125-131
//   cv::recoverPose(E,p1,p2,K,R,t,mask)                     src/initializer.cpp:236
// findHomography's post-RANSAC refit + LM polish is NOT restated: the reference reads only the mask
// (which OpenCV returns untouched by the polish) and discards H.
#include "orc_common.h"
#include "orc_linalg.h"
#include "orc_ransac.h"
#include "mvo_oracle.h"

namespace orc {

// ---- homography (fundam.cpp HomographyEstimatorCallback) -----------------------------------------
struct HomographyCb : RansacCb {
  HomographyCb() { d1 = 2; d2 = 2; model_size = 9; }
  bool check_subset(const float* ms1, const float* ms2, int count) const override {
    if (have_collinear_points(ms1, count) || have_collinear_points(ms2, count)) return false;
    if (count == 4) {
      static const int tt[][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
      int negative = 0;
      for (int i = 0; i < 4; i++) {
        const int* t = tt[i];
        double A[9] = {ms1[2 * t[0]], ms1[2 * t[0] + 1], 1., ms1[2 * t[1]], ms1[2 * t[1] + 1], 1.,
                       ms1[2 * t[2]], ms1[2 * t[2] + 1], 1.};
        double B[9] = {ms2[2 * t[0]], ms2[2 * t[0] + 1], 1., ms2[2 * t[1]], ms2[2 * t[1] + 1], 1.,
                       ms2[2 * t[2]], ms2[2 * t[2] + 1], 1.};
        negative += det3(A) * det3(B) < 0;
      }
      if (negative != 0 && negative != 4) return false;
    }
    return true;
  }
  int run_kernel(const float* M, const float* m, int count, double* model) const override {
    double LtL[9][9], W[9], V[9][9];
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; i++) {
      cmx += m[2 * i]; cmy += m[2 * i + 1];
      cMx += M[2 * i]; cMy += M[2 * i + 1];
    }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; i++) {
      smx += std::fabs(m[2 * i] - cmx);
      smy += std::fabs(m[2 * i + 1] - cmy);
      sMx += std::fabs(M[2 * i] - cMx);
      sMy += std::fabs(M[2 * i + 1] - cMy);
    }
    if (std::fabs(smx) < DBL_EPSILON || std::fabs(smy) < DBL_EPSILON || std::fabs(sMx) < DBL_EPSILON ||
        std::fabs(sMy) < DBL_EPSILON)
      return 0;
    smx = count / smx; smy = count / smy;
    sMx = count / sMx; sMy = count / sMy;
    double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    memset(LtL, 0, sizeof(LtL));
    for (int i = 0; i < count; i++) {
      double x = (m[2 * i] - cmx) * smx, y = (m[2 * i + 1] - cmy) * smy;
      double X = (M[2 * i] - cMx) * sMx, Y = (M[2 * i + 1] - cMy) * sMy;
      double Lx[] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
      double Ly[] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
      for (int j = 0; j < 9; j++)
        for (int k = j; k < 9; k++) LtL[j][k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; j++)
      for (int k = 0; k < j; k++) LtL[j][k] = LtL[k][j];  // completeSymm
    jacobi_eigen(&LtL[0][0], 9, W, &V[0][0]);
    double Htemp[9], H0[9];
    mat3mul(invHnorm, V[8], Htemp);
    mat3mul(Htemp, Hnorm2, H0);
    double s = 1. / H0[8];
    for (int i = 0; i < 9; i++) model[i] = H0[i] * s;
    return 1;
  }
  void compute_error(const float* M, const float* m, int count, const double* H, float* err) const override {
    float Hf[] = {(float)H[0], (float)H[1], (float)H[2], (float)H[3], (float)H[4], (float)H[5], (float)H[6], (float)H[7]};
    for (int i = 0; i < count; i++) {
      float ww = 1.f / (Hf[6] * M[2 * i] + Hf[7] * M[2 * i + 1] + 1.f);
      float dx = (Hf[0] * M[2 * i] + Hf[1] * M[2 * i + 1] + Hf[2]) * ww - m[2 * i];
      float dy = (Hf[3] * M[2 * i] + Hf[4] * M[2 * i + 1] + Hf[5]) * ww - m[2 * i + 1];
      err[i] = dx * dx + dy * dy;
    }
  }
};

// ---- fundamental matrix, 7-point (fundam.cpp FMEstimatorCallback / run7Point) ---------------------
struct FundamentalCb : RansacCb {
  FundamentalCb() { d1 = 2; d2 = 2; model_size = 9; }
  bool check_subset(const float* ms1, const float* ms2, int count) const override {
    return !have_collinear_points(ms1, count) && !have_collinear_points(ms2, count);
  }
  int run_kernel(const float* m1, const float* m2, int count_, double* fmatrix) const override {
    (void)count_;
    double a[7 * 9], w[7], v[9 * 9], c[4], r[3] = {0};
    double *f1, *f2;
    double t0, t1, t2;
    double m1cx = 0, m1cy = 0, m2cx = 0, m2cy = 0;
    double t, scale1 = 0, scale2 = 0;
    const int count = 7;
    for (int i = 0; i < count; i++) {
      m1cx += m1[2 * i]; m1cy += m1[2 * i + 1];
      m2cx += m2[2 * i]; m2cy += m2[2 * i + 1];
    }
    t = 1. / count;
    m1cx *= t; m1cy *= t; m2cx *= t; m2cy *= t;
    for (int i = 0; i < count; i++) {
      double ax = m1[2 * i] - m1cx, ay = m1[2 * i + 1] - m1cy;
      double bx = m2[2 * i] - m2cx, by = m2[2 * i + 1] - m2cy;
      scale1 += std::sqrt(ax * ax + ay * ay);
      scale2 += std::sqrt(bx * bx + by * by);
    }
    scale1 *= t; scale2 *= t;
    if (scale1 < FLT_EPSILON || scale2 < FLT_EPSILON) return 0;
    scale1 = std::sqrt(2.) / scale1;
    scale2 = std::sqrt(2.) / scale2;
    for (int i = 0; i < 7; i++) {
      double x0 = (m1[2 * i] - m1cx) * scale1;
      double y0 = (m1[2 * i + 1] - m1cy) * scale1;
      double x1 = (m2[2 * i] - m2cx) * scale2;
      double y1 = (m2[2 * i + 1] - m2cy) * scale2;
      a[i * 9 + 0] = x1 * x0; a[i * 9 + 1] = x1 * y0; a[i * 9 + 2] = x1;
      a[i * 9 + 3] = y1 * x0; a[i * 9 + 4] = y1 * y0; a[i * 9 + 5] = y1;
      a[i * 9 + 6] = x0; a[i * 9 + 7] = y0; a[i * 9 + 8] = 1;
    }
    svd_compute(a, 7, 9, w, nullptr, v, true);  // SVDecomp(A, W, U, Vt, MODIFY_A + FULL_UV)
    f1 = v + 7 * 9;
    f2 = v + 8 * 9;
    for (int i = 0; i < 9; i++) f1[i] -= f2[i];
    t0 = f2[4] * f2[8] - f2[5] * f2[7];
    t1 = f2[3] * f2[8] - f2[5] * f2[6];
    t2 = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) +
           f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) - f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
           f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
           f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7];
    t1 = f1[3] * f1[8] - f1[5] * f1[6];
    t2 = f1[3] * f1[7] - f1[4] * f1[6];
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) +
           f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) - f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
           f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
           f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    int n = solve_cubic(c, r);
    if (n < 1 || n > 3) return n;
    double T1[9] = {scale1, 0, -scale1 * m1cx, 0, scale1, -scale1 * m1cy, 0, 0, 1};
    double T2[9] = {scale2, 0, -scale2 * m2cx, 0, scale2, -scale2 * m2cy, 0, 0, 1};
    double T2t[9] = {T2[0], T2[3], T2[6], T2[1], T2[4], T2[7], T2[2], T2[5], T2[8]};
    for (int k = 0; k < n; k++, fmatrix += 9) {
      double lambda = r[k], mu = 1.;
      double s = f1[8] * r[k] + f2[8];
      if (std::fabs(s) > DBL_EPSILON) {
        mu = 1. / s;
        lambda *= mu;
        fmatrix[8] = 1.;
      } else
        fmatrix[8] = 0.;
      for (int i = 0; i < 8; i++) fmatrix[i] = f1[i] * lambda + f2[i] * mu;
      double tmp[9];
      mat3mul(T2t, fmatrix, tmp);   // F = T2.t() * F * T1 (MatExpr: (T2t*F)*T1)
      mat3mul(tmp, T1, fmatrix);
      if (std::fabs(fmatrix[8]) > FLT_EPSILON) {
        double sc = 1. / fmatrix[8];
        for (int i = 0; i < 9; i++) fmatrix[i] *= sc;
      }
    }
    return n;
  }
  void compute_error(const float* m1, const float* m2, int count, const double* F, float* err) const override {
    for (int i = 0; i < count; i++) {
      double a, b, c, d1, d2, s1, s2;
      a = F[0] * m1[2 * i] + F[1] * m1[2 * i + 1] + F[2];
      b = F[3] * m1[2 * i] + F[4] * m1[2 * i + 1] + F[5];
      c = F[6] * m1[2 * i] + F[7] * m1[2 * i + 1] + F[8];
      s2 = 1. / (a * a + b * b);
      d2 = m2[2 * i] * a + m2[2 * i + 1] * b + c;
      a = F[0] * m2[2 * i] + F[3] * m2[2 * i + 1] + F[6];
      b = F[1] * m2[2 * i] + F[4] * m2[2 * i + 1] + F[7];
      c = F[2] * m2[2 * i] + F[5] * m2[2 * i + 1] + F[8];
      s1 = 1. / (a * a + b * b);
      d1 = m1[2 * i] * a + m1[2 * i + 1] * b + c;
      err[i] = (float)std::max(d1 * d1 * s1, d2 * d2 * s2);
    }
  }
};

// ---- triangulatePoints (triangulate.cpp icvTriangulatePoints, 4.6: per point 4x4 A, SVD, last Vt row) --
static void triangulate_one(const double* P1, const double* P2, double x1, double y1, double x2, double y2, double X[4]) {
  double A[16], w[4], vt[16];
  for (int k = 0; k < 4; k++) {
    A[0 * 4 + k] = x1 * P1[8 + k] - P1[0 + k];
    A[1 * 4 + k] = y1 * P1[8 + k] - P1[4 + k];
    A[2 * 4 + k] = x2 * P2[8 + k] - P2[0 + k];
    A[3 * 4 + k] = y2 * P2[8 + k] - P2[4 + k];
  }
  svd_compute(A, 4, 4, w, nullptr, vt, false);
  for (int k = 0; k < 4; k++) X[k] = vt[12 + k];
}

}  // namespace orc

using namespace orc;

extern "C" int orc_find_homography_ransac(const float* p1, const float* p2, int n, double thr, int max_iters,
                                          double confidence, unsigned char* mask, double* H, int* stats) {
  if (n < 4) return -1;
  if (thr <= 0) thr = 3;
  HomographyCb cb;
  bool ok;
  RansacStats st;
  if (n == 4) {
    memset(mask, 1, n);
    ok = cb.run_kernel(p1, p2, n, H) > 0;
  } else {
    ok = ransac_run(cb, p1, p2, n, 4, thr, confidence, max_iters, H, mask, &st);
  }
  if (stats) { stats[0] = st.iters_run; stats[1] = st.niters_final; stats[2] = st.hyp_models; }
  if (!ok) { memset(mask, 0, n); return 0; }
  int c = 0;
  for (int i = 0; i < n; i++) c += mask[i] != 0;
  return c;
}

extern "C" int orc_find_fundamental_ransac(const float* p1, const float* p2, int n, double thr, double confidence,
                                           int max_iters, unsigned char* mask, double* F, int* stats) {
  if (n < 7) return -1;  // OpenCV returns an empty Mat and leaves the mask untouched
  FundamentalCb cb;
  RansacStats st;
  bool ok;
  if (n == 7) {
    double models[27];
    int nm = cb.run_kernel(p1, p2, n, models);
    memset(mask, 1, n);
    ok = nm > 0;
    if (ok) memcpy(F, models, 9 * sizeof(double));
  } else {
    if (thr <= 0) thr = 3;
    if (confidence < DBL_EPSILON || confidence > 1 - DBL_EPSILON) confidence = 0.99;
    // fundam.cpp: FM_RANSAC needs >= 15 points, below that findFundamentalMat silently switches to LMedS
    // (createLMeDSPointSetRegistrator(cb, 7, confidence), default maxIters 1000)
    if (n >= 15) ok = ransac_run(cb, p1, p2, n, 7, thr, confidence, max_iters, F, mask, &st);
    else ok = lmeds_run(cb, p1, p2, n, 7, confidence, 1000, F, mask, &st);
  }
  if (stats) { stats[0] = st.iters_run; stats[1] = st.niters_final; stats[2] = st.hyp_models; }
  if (!ok) { memset(mask, 0, n); return 0; }
  int c = 0;
  for (int i = 0; i < n; i++) c += mask[i] != 0;
  return c;
}

// Building blocks exposed for tests.
extern "C" int orc_h4_kernel(const float* p1, const float* p2, int n, double* H) { HomographyCb cb; return cb.run_kernel(p1, p2, n, H); }
extern "C" int orc_f7_kernel(const float* p1, const float* p2, double* F) { FundamentalCb cb; return cb.run_kernel(p1, p2, 7, F); }
extern "C" void orc_svd(const double* A, int m, int n, double* w, double* U, double* Vt, int full) { svd_compute(A, m, n, w, U, Vt, full != 0); }
extern "C" void orc_eigen_sym(const double* A, int n, double* W, double* V) {
  std::vector<double> a(A, A + n * n);
  jacobi_eigen(a.data(), n, W, V);
}
extern "C" int orc_solve_cubic(const double* c, double* x) { return solve_cubic(c, x); }
extern "C" unsigned orc_rng_next(unsigned long long* state) { RNG r(*state); unsigned v = r.next(); *state = r.state; return v; }
extern "C" int orc_ransac_update_num_iters(double p, double ep, int mp, int mi) { return ransac_update_num_iters(p, ep, mp, mi); }

// cv::triangulatePoints(P1, P2, p1 (float), p2 (float)) -> 4 x n float; then
// cv::convertPointsFromHomogeneous (float: scale = w != 0 ? 1.f/w : 1.f).
extern "C" int orc_triangulate(const double* P1, const double* P2, const float* p1, const float* p2, int n, float* X3, float* X4) {
  for (int i = 0; i < n; i++) {
    double X[4];
    triangulate_one(P1, P2, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1], X);
    float xf[4] = {(float)X[0], (float)X[1], (float)X[2], (float)X[3]};
    if (X4) memcpy(X4 + 4 * i, xf, sizeof(xf));
    float scale = xf[3] != 0.f ? 1.f / xf[3] : 1.f;
    X3[3 * i] = xf[0] * scale; X3[3 * i + 1] = xf[1] * scale; X3[3 * i + 2] = xf[2] * scale;
  }
  return n;
}

// cv::decomposeEssentialMat + cv::recoverPose(E, p1, p2, K, R, t, distanceThresh = 50, mask) (five-point.cpp).
extern "C" int orc_recover_pose(const double* E, const float* p1f, const float* p2f, int n, const double* K, double* Rout,
                                double* tout, unsigned char* mask_io) {
  const double dist = 50.0;
  double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  std::vector<double> x1(2 * n), x2(2 * n);
  for (int i = 0; i < n; i++) {
    x1[2 * i] = (p1f[2 * i] - cx) / fx; x1[2 * i + 1] = (p1f[2 * i + 1] - cy) / fy;
    x2[2 * i] = (p2f[2 * i] - cx) / fx; x2[2 * i + 1] = (p2f[2 * i + 1] - cy) / fy;
  }
  // decomposeEssentialMat
  double w[3], U[9], Vt[9];
  svd_compute(E, 3, 3, w, U, Vt, false);
  if (det3(U) < 0) for (double& v : U) v *= -1.;
  if (det3(Vt) < 0) for (double& v : Vt) v *= -1.;
  const double Wm[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
  const double Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
  double R1[9], R2[9], tmp[9], t[3];
  mat3mul(U, Wm, tmp); mat3mul(tmp, Vt, R1);
  mat3mul(U, Wt, tmp); mat3mul(tmp, Vt, R2);
  t[0] = U[2]; t[1] = U[5]; t[2] = U[8];
  const double* Rs[4] = {R1, R2, R1, R2};
  const double ts[4] = {1, 1, -1, -1};
  std::vector<unsigned char> masks[4];
  int good[4];
  for (int c = 0; c < 4; c++) {
    double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, P[12];
    for (int r = 0; r < 3; r++) {
      for (int k = 0; k < 3; k++) P[r * 4 + k] = Rs[c][r * 3 + k];
      P[r * 4 + 3] = t[r] * ts[c];
    }
    masks[c].assign(n, 0);
    good[c] = 0;
    for (int i = 0; i < n; i++) {
      double Q[4];
      triangulate_one(P0, P, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1], Q);
      bool m = (Q[2] * Q[3]) > 0;
      double q0 = Q[0] / Q[3], q1 = Q[1] / Q[3], q2 = Q[2] / Q[3];
      m = m && (q2 < dist);
      double z2 = P[8] * q0 + P[9] * q1 + P[10] * q2 + P[11] * 1.0;
      m = m && (z2 > 0) && (z2 < dist);
      if (mask_io) m = m && (mask_io[i] != 0);
      masks[c][i] = m ? 1 : 0;
      good[c] += m ? 1 : 0;
    }
  }
  int best;
  if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) best = 0;
  else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) best = 1;
  else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) best = 2;
  else best = 3;
  memcpy(Rout, Rs[best], 9 * sizeof(double));
  for (int k = 0; k < 3; k++) tout[k] = t[k] * ts[best];
  if (mask_io) memcpy(mask_io, masks[best].data(), n);
  return good[best];
}
