// csrc/geom_models.h — minimal solvers + residuals of the RANSAC models (device code, one hypothesis per
// lane).  Each model policy provides:
//   MP          sample size            MAXM   models per sample        MS   doubles per model
//   PT1 / PT2   floats per point in set 1 / set 2
//   check_subset(ms1, ms2)             OpenCV's Callback::checkSubset
//   solve(P, ms1, ms2, models, ws)     OpenCV's Callback::runKernel   -> number of models
//   CH          hypotheses solved per RANSAC round (lanes of the wavefront that run the solver)
//   WS          doubles of per-lane LDS workspace `ws` for the solver's dense matrices (0: all in private memory).
//               Private (scratch) matrices cost an L2 round trip per element and, with per-lane pivots (Jacobi
//               eigen), 64 different cache lines per access; in LDS they cost a bank access.
//   Scorer                             per-model constants + err(p1, p2) -> float (Callback::computeError)
// Semantics: OpenCV 4.6 calib3d (fundam.cpp, solvepnp.cpp, epnp.cpp, calibration.cpp) as restated in
// SURVEY.md Appendix A.5-A.7; arithmetic order matches oracle/orc_geom.cpp / orc_pnp.cpp.
#pragma once
#include "geom_linalg.h"

// Intrinsics + the 5-coefficient plumb-bob distortion (k1, k2, p1, p2, k3) the reference forwards from CameraInfo::d
// (src/tracker.cpp:309).  `dist` = any coefficient non-zero; brace-initialising the first four members leaves it off.
struct CamK { double fx, fy, cx, cy; double k[5]; int dist; };
inline CamK make_camk(const double K[9], const double* d) {
  CamK c{K[0], K[4], K[2], K[5], {0, 0, 0, 0, 0}, 0};
  for (int i = 0; i < 5; i++) { c.k[i] = d ? d[i] : 0.0; c.dist = c.dist || c.k[i] != 0.0; }
  return c;
}
struct ModelParams { CamK cam; };

// ---------------------------------------------------------------------------------------------------
// Homography, 4 points (HomographyEstimatorCallback)
// ---------------------------------------------------------------------------------------------------
// Homography rounds: RS_H_NW wavefronts per stream, RS_H_CH hypotheses each.  A hypothesis' workspace is 135 doubles (LtL as
// packed upper triangle, W, V); 4 x 16 = 64 per round is 82 KB of LDS (106 KB with the full 9 x 9 matrix before), 4 x 24 =
// 96 per round 124 KB.  One stream, 2000 iterations, 20 % outliers (profiles/tools/ransac_h_wall.py): 4 x 16 13.4 ms, 6 x 16
// 12.8 ms (two SIMDs carry two wavefronts: a round takes as much longer as it is wider), 4 x 24 and 3 x 32 9.9 ms, 2 x 48
// 13.7 ms, 1 x 64 24.7 ms - the f64 pipe works through a wavefront in quarters of 16 lanes and skips empty ones.  In the
// bench (four contexts) the 96-wide round first did not pay (72.5 k against 73.3 k frames/s over three runs each: the
// other contexts' LK wavefronts found less LDS); with the LK launches taking turns it does: 73.0 k against 71.0 k, the
// homography stage 2.3 instead of 2.8 ms per step.
#ifndef RS_H_NW
#define RS_H_NW 4
#endif
#ifndef RS_H_CH
#define RS_H_CH 24
#endif
#ifndef RS_H_OVERDRAW
#define RS_H_OVERDRAW 1
#endif
struct HModel {
  static constexpr int LANES = 1;
  static constexpr bool SEQ_SCORE = false;   // score all hypotheses of a round, then replay (geom.hip)
  static constexpr int MP = 4, MAXM = 1, MS = 9, PT1 = 2, PT2 = 2;
  // 16 hypotheses per round, every round in LDS (16 x 193 doubles = 25 KB): a 64-wide round with 48 workspaces in
  // private memory took 2.85 ms against 0.8 ms for an LDS round, i.e. more per hypothesis; wider LDS rounds (24, 32)
  // made the workgroup wait for LDS beside the image kernels
  // LtL, W, V.  In LDS (the RANSAC kernel) LtL is the packed upper triangle (45): 135 doubles per hypothesis; anywhere
  // else (host checks of this header) the generic routine wants the full matrix
#if defined(__HIP_DEVICE_COMPILE__)
  static constexpr int CH = RS_H_CH, WS = 45 + 9 + 81;
#else
  static constexpr int CH = RS_H_CH, WS = 81 + 9 + 81;
#endif
  static constexpr bool WIDE = false;
  static constexpr int LMEDS_BELOW = 0;
  static constexpr int MP_ALT = 0;   // no second sample size
  static constexpr bool OVERDRAW = RS_H_OVERDRAW != 0;   // checkSubset rejects most samples of a non-planar scene: refill the queue with one candidate per thread

  __device__ static bool check_subset(const float* ms1, const float* ms2) {
    if (gl_have_collinear(ms1, 4) || gl_have_collinear(ms2, 4)) return false;
    const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
    int negative = 0;
    for (int i = 0; i < 4; i++) {
      const int* t = tt[i];
      double A[9] = {ms1[2 * t[0]], ms1[2 * t[0] + 1], 1., ms1[2 * t[1]], ms1[2 * t[1] + 1], 1., ms1[2 * t[2]], ms1[2 * t[2] + 1], 1.};
      double B[9] = {ms2[2 * t[0]], ms2[2 * t[0] + 1], 1., ms2[2 * t[1]], ms2[2 * t[1] + 1], 1., ms2[2 * t[2]], ms2[2 * t[2] + 1], 1.};
      negative += gl_det3(A) * gl_det3(B) < 0;
    }
    return !(negative != 0 && negative != 4);
  }

  // runKernel for `count` float points (count = 4 inside RANSAC)
  __device__ GL_NOINLINE static int solve_n(const float* M, const float* m, int count, double* model, double* ws) {
    const bool packed = gl_is_lds(ws);
    double *LtL = ws, *W = ws + (packed ? 45 : 81), *V = W + 9;
#ifdef RS_TIMING
    const long long tq0 = wall_clock64();
#endif
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; i++) {
      cmx += m[2 * i]; cmy += m[2 * i + 1];
      cMx += M[2 * i]; cMy += M[2 * i + 1];
    }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; i++) {
      smx += fabs(m[2 * i] - cmx);
      smy += fabs(m[2 * i + 1] - cmy);
      sMx += fabs(M[2 * i] - cMx);
      sMy += fabs(M[2 * i + 1] - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON) return 0;
    smx = count / smx; smy = count / smy;
    sMx = count / sMx; sMy = count / sMy;
    double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    for (int i = 0; i < (packed ? 45 : 81); i++) LtL[i] = 0;
    for (int i = 0; i < count; i++) {
      double x = (m[2 * i] - cmx) * smx, y = (m[2 * i + 1] - cmy) * smy;
      double X = (M[2 * i] - cMx) * sMx, Y = (M[2 * i + 1] - cMy) * sMy;
      double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
      double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
      // rolled on purpose: fully unrolled, these loops alone take ~250 VGPRs and the wave then crowds out its SIMD
#pragma unroll 1
      for (int j = 0; j < 9; j++) {
        const int row = packed ? (j * (17 - j)) >> 1 : j * 9;   // gl_tri9
#pragma unroll 1
        for (int k = j; k < 9; k++) LtL[row + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
      }
    }
    if (!packed) {
#pragma unroll 1
      for (int j = 0; j < 9; j++)
#pragma unroll 1
        for (int k = 0; k < j; k++) LtL[j * 9 + k] = LtL[k * 9 + j];
    }
#ifdef RS_TIMING
    const long long tq1 = wall_clock64();
#endif
    // in the RANSAC kernel the workspace is LDS: the load-batched routine (same rotations, same values)
    if (packed) gl_jacobi_eigen9_lds<true>((gl_ldsd*)LtL, (gl_ldsd*)W, (gl_ldsd*)V);
    else gl_jacobi_eigen(LtL, 9, W, V);
#ifdef RS_TIMING
    const long long tq2 = wall_clock64();
#endif
    double Htemp[9], H0[9];
    gl_mat3mul(invHnorm, V + 72, Htemp);
    gl_mat3mul(Htemp, Hnorm2, H0);
    double s = 1. / H0[8];
    for (int i = 0; i < 9; i++) model[i] = H0[i] * s;
#ifdef RS_TIMING
    if (threadIdx.x == 0 && blockIdx.x == 0) printf("H solve: build %lld eigen %lld tail %lld ticks\n", tq1 - tq0, tq2 - tq1, wall_clock64() - tq2);
#endif
    return 1;
  }
  __device__ static int solve(const ModelParams&, const float* ms1, const float* ms2, double* models, double* ws) {
    return solve_n(ms1, ms2, 4, models, ws);
  }

  struct Scorer {
    float Hf[8];
    __device__ void init(const ModelParams&, const double* H) {
      for (int i = 0; i < 8; i++) Hf[i] = (float)H[i];
    }
    __device__ __forceinline__ float err(const float* M, const float* m) const {
      float ww = 1.f / (Hf[6] * M[0] + Hf[7] * M[1] + 1.f);
      float dx = (Hf[0] * M[0] + Hf[1] * M[1] + Hf[2]) * ww - m[0];
      float dy = (Hf[3] * M[0] + Hf[4] * M[1] + Hf[5]) * ww - m[1];
      return dx * dx + dy * dy;
    }
  };
};

// ---------------------------------------------------------------------------------------------------
// Fundamental matrix, 7 points (FMEstimatorCallback / run7Point)
// ---------------------------------------------------------------------------------------------------
struct FModel {
  static constexpr int LANES = 1;
  static constexpr bool SEQ_SCORE = false;   // score all hypotheses of a round, then replay (geom.hip)
  static constexpr int MP = 7, MAXM = 3, MS = 9, PT1 = 2, PT2 = 2;
  static constexpr int CH = 16, WS = 63 + 81 + 81 + 49;  // a, v, ta, tv
  static constexpr bool WIDE = true;
  static constexpr int LMEDS_BELOW = 15;  // findFundamentalMat: FM_RANSAC with fewer than 15 points runs LMedS (fundam.cpp)
  static constexpr int MP_ALT = 0;   // no second sample size
  static constexpr bool OVERDRAW = false;

  __device__ static bool check_subset(const float* ms1, const float* ms2) {
    return !gl_have_collinear(ms1, 7) && !gl_have_collinear(ms2, 7);
  }

  __device__ GL_NOINLINE static int solve(const ModelParams&, const float* m1, const float* m2, double* fmatrix, double* ws) {
    double *a = ws, *v = ws + 63, *ta = ws + 144, *tv = ws + 225;
    double w[7], c[4], r[3] = {0, 0, 0};
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
      scale1 += sqrt(ax * ax + ay * ay);
      scale2 += sqrt(bx * bx + by * by);
    }
    scale1 *= t; scale2 *= t;
    if (scale1 < FLT_EPSILON || scale2 < FLT_EPSILON) return 0;
    scale1 = sqrt(2.) / scale1;
    scale2 = sqrt(2.) / scale2;
    for (int i = 0; i < 7; i++) {
      double x0 = (m1[2 * i] - m1cx) * scale1;
      double y0 = (m1[2 * i + 1] - m1cy) * scale1;
      double x1 = (m2[2 * i] - m2cx) * scale2;
      double y1 = (m2[2 * i + 1] - m2cy) * scale2;
      a[i * 9 + 0] = x1 * x0; a[i * 9 + 1] = x1 * y0; a[i * 9 + 2] = x1;
      a[i * 9 + 3] = y1 * x0; a[i * 9 + 4] = y1 * y0; a[i * 9 + 5] = y1;
      a[i * 9 + 6] = x0; a[i * 9 + 7] = y0; a[i * 9 + 8] = 1;
    }
    gl_svd_compute(a, 7, 9, w, nullptr, v, true, ta, tv);
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
    int n = gl_solve_cubic(c, r);
    if (n < 1 || n > 3) return n;
    double T1[9] = {scale1, 0, -scale1 * m1cx, 0, scale1, -scale1 * m1cy, 0, 0, 1};
    double T2t[9] = {scale2, 0, 0, 0, scale2, 0, -scale2 * m2cx, -scale2 * m2cy, 1};
    for (int k = 0; k < n; k++, fmatrix += 9) {
      double lambda = r[k], mu = 1.;
      double s = f1[8] * r[k] + f2[8];
      if (fabs(s) > DBL_EPSILON) {
        mu = 1. / s;
        lambda *= mu;
        fmatrix[8] = 1.;
      } else
        fmatrix[8] = 0.;
      for (int i = 0; i < 8; i++) fmatrix[i] = f1[i] * lambda + f2[i] * mu;
      double tmp[9];
      gl_mat3mul(T2t, fmatrix, tmp);
      gl_mat3mul(tmp, T1, fmatrix);
      if (fabs(fmatrix[8]) > FLT_EPSILON) {
        double sc = 1. / fmatrix[8];
        for (int i = 0; i < 9; i++) fmatrix[i] *= sc;
      }
    }
    return n;
  }

  struct Scorer {
    double F[9];
    __device__ void init(const ModelParams&, const double* f) {
      for (int i = 0; i < 9; i++) F[i] = f[i];
    }
    __device__ __forceinline__ float err(const float* m1, const float* m2) const {
      double a, b, c, d1, d2, s1, s2;
      a = F[0] * m1[0] + F[1] * m1[1] + F[2];
      b = F[3] * m1[0] + F[4] * m1[1] + F[5];
      c = F[6] * m1[0] + F[7] * m1[1] + F[8];
      s2 = 1. / (a * a + b * b);
      d2 = m2[0] * a + m2[1] * b + c;
      a = F[0] * m2[0] + F[3] * m2[1] + F[6];
      b = F[1] * m2[0] + F[4] * m2[1] + F[7];
      c = F[2] * m2[0] + F[5] * m2[1] + F[8];
      s1 = 1. / (a * a + b * b);
      d1 = m1[0] * a + m1[1] * b + c;
      return (float)fmax(d1 * d1 * s1, d2 * d2 * s2);
    }
  };
};

// ---------------------------------------------------------------------------------------------------
// Rodrigues / projection (calibration.cpp cvRodrigues2, cvProjectPoints2 with zero distortion)
// ---------------------------------------------------------------------------------------------------
__device__ GL_NOINLINE void gm_rodrigues_v2m(const double r_[3], double R[9], double* J /* 27 or null */) {
  double rx = r_[0], ry = r_[1], rz = r_[2];
  double theta = sqrt(rx * rx + ry * ry + rz * rz);
  if (theta < DBL_EPSILON) {
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1 : 0;
    if (J) {
      for (int i = 0; i < 27; i++) J[i] = 0;
      J[5] = J[15] = J[19] = -1;
      J[7] = J[11] = J[21] = 1;
    }
    return;
  }
  double c = cos(theta), s = sin(theta), c1 = 1. - c;
  double itheta = theta ? 1. / theta : 0.;
  rx *= itheta; ry *= itheta; rz *= itheta;
  double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
  double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
  if (J) {
    const double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0, 0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                             0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
    const double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; i++) {
      double ri = i == 0 ? rx : i == 1 ? ry : rz;
      double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
      double a3 = (c - s * itheta) * ri, a4 = s * itheta;
      for (int k = 0; k < 9; k++)
        J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
    }
  }
}

__device__ GL_NOINLINE void gm_rodrigues_m2v(const double Rin[9], double r[3]) {
  for (int i = 0; i < 9; i++)
    if (!(Rin[i] > -100 && Rin[i] < 100)) { r[0] = r[1] = r[2] = 0; return; }
  double w[3], U[9], Vt[9], R[9];
  gl_svd3(Rin, w, U, Vt);
  gl_mat3mul(U, Vt, R);
  double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
  double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  double theta = acos(c);
  if (s < 1e-5) {
    double t;
    if (c > 0) rx = ry = rz = 0;
    else {
      t = (R[0] + 1) * 0.5;
      rx = sqrt(fmax(t, 0.));
      t = (R[4] + 1) * 0.5;
      ry = sqrt(fmax(t, 0.)) * (R[1] < 0 ? -1. : 1.);
      t = (R[8] + 1) * 0.5;
      rz = sqrt(fmax(t, 0.)) * (R[2] < 0 ? -1. : 1.);
      if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
      theta /= sqrt(rx * rx + ry * ry + rz * rz);
      rx *= theta; ry *= theta; rz *= theta;
    }
  } else {
    double vth = 1 / (2 * s);
    vth *= theta;
    rx *= vth; ry *= vth; rz *= vth;
  }
  r[0] = rx; r[1] = ry; r[2] = rz;
}

// cvUndistortPointsInternal for one pixel (identity R, no P): 5 fixed-point iterations (TermCriteria COUNT 5)
__device__ inline void gm_undistort_point(const CamK& cam, double u, double v, double& xo, double& yo) {
  const double ifx = 1. / cam.fx, ify = 1. / cam.fy;
  double x = (u - cam.cx) * ifx, y = (v - cam.cy) * ify;
  if (cam.dist) {
    const double* k = cam.k;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
      double r2 = x * x + y * y;
      double icdist = 1. / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      if (icdist < 0) { x = (u - cam.cx) * ifx; y = (v - cam.cy) * ify; break; }
      double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
      double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
  }
  xo = x; yo = y;
}

// cvProjectPoints2 for one point, optional d m / d r (2x3) and d m / d t (2x3)
__device__ inline void gm_project_point(const double R[9], const double* dRdr, const double t[3], const CamK& cam,
                                        const double M[3], double m[2], double* dpdr, double* dpdt) {
  const double* k = cam.k;
  double X = M[0], Y = M[1], Z = M[2];
  double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
  double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
  double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
  z = z ? 1. / z : 1;
  x *= z; y *= z;
  double r2 = 0, r4 = 0, cdist = 1, xd = x, yd = y;
  if (cam.dist) {
    r2 = x * x + y * y; r4 = r2 * r2;
    double r6 = r4 * r2, a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
    xd = x * cdist + k[2] * a1 + k[3] * a2;
    yd = y * cdist + k[2] * a3 + k[3] * a1;
  }
  m[0] = xd * cam.fx + cam.cx;
  m[1] = yd * cam.fy + cam.cy;
  if (dpdt) {
    double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
    for (int j = 0; j < 3; j++) {
      double dmxdt = dxdt[j], dmydt = dydt[j];
      if (cam.dist) {
        double dr2dt = 2 * x * dxdt[j] + 2 * y * dydt[j];
        double dcdist_dt = k[0] * dr2dt + 2 * k[1] * r2 * dr2dt + 3 * k[4] * r4 * dr2dt;
        double da1dt = 2 * (x * dydt[j] + y * dxdt[j]);
        dmxdt = dxdt[j] * cdist + x * dcdist_dt + k[2] * da1dt + k[3] * (dr2dt + 4 * x * dxdt[j]);
        dmydt = dydt[j] * cdist + y * dcdist_dt + k[2] * (dr2dt + 4 * y * dydt[j]) + k[3] * da1dt;
      }
      dpdt[j] = cam.fx * dmxdt;
      dpdt[3 + j] = cam.fy * dmydt;
    }
  }
  if (dpdr) {
    double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11],
                       X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
    double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14],
                       X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
    double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17],
                       X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
    for (int j = 0; j < 3; j++) {
      double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
      double dydr = z * (dy0dr[j] - y * dz0dr[j]);
      double dmxdr = dxdr, dmydr = dydr;
      if (cam.dist) {
        double dr2dr = 2 * x * dxdr + 2 * y * dydr;
        double dcdist_dr = (k[0] + 2 * k[1] * r2 + 3 * k[4] * r4) * dr2dr;
        double da1dr = 2 * (x * dydr + y * dxdr);
        dmxdr = dxdr * cdist + x * dcdist_dr + k[2] * da1dr + k[3] * (dr2dr + 4 * x * dxdr);
        dmydr = dydr * cdist + y * dcdist_dr + k[2] * (dr2dr + 4 * y * dydr) + k[3] * da1dr;
      }
      dpdr[j] = cam.fx * dmxdr;
      dpdr[3 + j] = cam.fy * dmydr;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// solvePnPRansac with exactly four correspondences (src/tracker.cpp:309 with min_tracked_points lowered to 4):
// model_points = 4 == npoints, so OpenCV runs solvePnP(SOLVEPNP_P3P) on all four and returns them as inliers - no RANSAC,
// no refinement.  solveP3P: undistortPoints -> p3p::solve (Gao's P3P: quartic by closed form, Horn alignment with a 4x4
// Jacobi, the fourth point orders the solutions) -> smallest squared reprojection error over the four points.  Same IEEE
// sequence as oracle/orc_p3p.inc (checked bit for bit on x86 by tests/test_sanitize_geom.py); cos / acos / pow come from
// OCML on the device.
// ---------------------------------------------------------------------------------------------------
#define GM_P3_PI 3.1415926535897932384626433832795   // CV_PI

__device__ GL_NOINLINE int gm_p3_solve_deg2(double a, double b, double c, double& x1, double& x2) {
  double delta = b * b - 4 * a * c;
  if (delta < 0) return 0;
  double inv_2a = 0.5 / a;
  if (delta == 0) {
    x1 = -b * inv_2a;
    x2 = x1;
    return 1;
  }
  double sqrt_delta = sqrt(delta);
  x1 = (-b + sqrt_delta) * inv_2a;
  x2 = (-b - sqrt_delta) * inv_2a;
  return 2;
}

__device__ GL_NOINLINE int gm_p3_solve_deg3(double a, double b, double c, double d, double& x0, double& x1, double& x2) {
  if (a == 0) {
    if (b == 0) {
      if (c == 0) return 0;
      x0 = -d / c;
      return 1;
    }
    x2 = 0;
    return gm_p3_solve_deg2(b, c, d, x0, x1);
  }
  double inv_a = 1. / a;
  double b_a = inv_a * b, b_a2 = b_a * b_a;
  double c_a = inv_a * c;
  double d_a = inv_a * d;
  double Q = (3 * c_a - b_a2) / 9;
  double R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
  double Q3 = Q * Q * Q;
  double D = Q3 + R * R;
  double b_a_3 = (1. / 3.) * b_a;
  if (Q == 0) {
    if (R == 0) {
      x0 = x1 = x2 = -b_a_3;
      return 3;
    } else {
      x0 = pow(2 * R, 1 / 3.0) - b_a_3;
      return 1;
    }
  }
  if (D <= 0) {
    double theta = acos(R / sqrt(-Q3));
    double sqrt_Q = sqrt(-Q);
    x0 = 2 * sqrt_Q * cos(theta / 3.0) - b_a_3;
    x1 = 2 * sqrt_Q * cos((theta + 2 * GM_P3_PI) / 3.0) - b_a_3;
    x2 = 2 * sqrt_Q * cos((theta + 4 * GM_P3_PI) / 3.0) - b_a_3;
    return 3;
  }
  double AD = pow(fabs(R) + sqrt(D), 1.0 / 3.0) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
  double BD = (AD == 0) ? 0 : -Q / AD;
  x0 = AD + BD - b_a_3;
  return 1;
}

__device__ GL_NOINLINE int gm_p3_solve_deg4(double a, double b, double c, double d, double e, double& x0, double& x1, double& x2, double& x3) {
  if (a == 0) {
    x3 = 0;
    return gm_p3_solve_deg3(b, c, d, e, x0, x1, x2);
  }
  double inv_a = 1. / a;
  b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
  double b2 = b * b, bc = b * c, b3 = b2 * b;
  double r0, r1, r2;
  int n = gm_p3_solve_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, r0, r1, r2);
  if (n == 0) return 0;
  double R2 = 0.25 * b2 - c + r0, R;
  if (R2 < 0) return 0;
  R = sqrt(R2);
  double inv_R = 1. / R;
  int nb_real_roots = 0;
  double D2, E2;
  if (R < 10E-12) {
    double temp = r0 * r0 - 4 * e;
    if (temp < 0)
      D2 = E2 = -1;
    else {
      double sqrt_temp = sqrt(temp);
      D2 = 0.75 * b2 - 2 * c + 2 * sqrt_temp;
      E2 = D2 - 4 * sqrt_temp;
    }
  } else {
    double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
    D2 = u + v;
    E2 = u - v;
  }
  double b_4 = 0.25 * b, R_2 = 0.5 * R;
  if (D2 >= 0) {
    double D = sqrt(D2);
    nb_real_roots = 2;
    double D_2 = 0.5 * D;
    x0 = R_2 + D_2 - b_4;
    x1 = x0 - D;
  }
  if (E2 >= 0) {
    double E = sqrt(E2);
    double E_2 = 0.5 * E;
    if (nb_real_roots == 0) {
      x0 = -R_2 + E_2 - b_4;
      x1 = x0 - E;
      nb_real_roots = 2;
    } else {
      x2 = -R_2 + E_2 - b_4;
      x3 = x2 - E;
      nb_real_roots = 4;
    }
  }
  return nb_real_roots;
}

// p3p::jacobi_4x4: cyclic Jacobi, A symmetric 4x4 (upper triangle used), D eigenvalues, U eigenvectors in columns
__device__ GL_NOINLINE bool gm_p3_jacobi_4x4(double* A, double* D, double* U) {
  double B[4] = {}, Z[4] = {};
  double Id[16] = {1., 0., 0., 0., 0., 1., 0., 0., 0., 0., 1., 0., 0., 0., 0., 1.};
  for (int i = 0; i < 16; i++) U[i] = Id[i];
  B[0] = A[0]; B[1] = A[5]; B[2] = A[10]; B[3] = A[15];
  for (int i = 0; i < 4; i++) D[i] = B[i];
  for (int iter = 0; iter < 50; iter++) {
    double sum = fabs(A[1]) + fabs(A[2]) + fabs(A[3]) + fabs(A[6]) + fabs(A[7]) + fabs(A[11]);
    if (sum == 0.0) return true;
    double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
    for (int i = 0; i < 3; i++) {
      double* pAij = A + 5 * i + 1;
      for (int j = i + 1; j < 4; j++) {
        double Aij = *pAij;
        double eps_machine = 100.0 * fabs(Aij);
        if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) && fabs(D[j]) + eps_machine == fabs(D[j]))
          *pAij = 0.0;
        else if (fabs(Aij) > tresh) {
          double hh = D[j] - D[i], t;
          if (fabs(hh) + eps_machine == fabs(hh))
            t = Aij / hh;
          else {
            double theta = 0.5 * hh / Aij;
            t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
            if (theta < 0.0) t = -t;
          }
          hh = t * Aij;
          Z[i] -= hh;
          Z[j] += hh;
          D[i] -= hh;
          D[j] += hh;
          *pAij = 0.0;
          double c = 1.0 / sqrt(1 + t * t);
          double s = t * c;
          double tau = s / (1.0 + c);
          for (int k = 0; k <= i - 1; k++) {
            double g = A[k * 4 + i], h = A[k * 4 + j];
            A[k * 4 + i] = g - s * (h + g * tau);
            A[k * 4 + j] = h + s * (g - h * tau);
          }
          for (int k = i + 1; k <= j - 1; k++) {
            double g = A[i * 4 + k], h = A[k * 4 + j];
            A[i * 4 + k] = g - s * (h + g * tau);
            A[k * 4 + j] = h + s * (g - h * tau);
          }
          for (int k = j + 1; k < 4; k++) {
            double g = A[i * 4 + k], h = A[j * 4 + k];
            A[i * 4 + k] = g - s * (h + g * tau);
            A[j * 4 + k] = h + s * (g - h * tau);
          }
          for (int k = 0; k < 4; k++) {
            double g = U[k * 4 + i], h = U[k * 4 + j];
            U[k * 4 + i] = g - s * (h + g * tau);
            U[k * 4 + j] = h + s * (g - h * tau);
          }
        }
        pAij++;
      }
    }
    for (int i = 0; i < 4; i++) B[i] += Z[i];
    for (int i = 0; i < 4; i++) D[i] = B[i];
    for (int i = 0; i < 4; i++) Z[i] = 0;
  }
  return false;
}

// p3p::align: rigid motion taking (X_i, Y_i, Z_i) onto M_end[i] (Horn 1987, unit quaternion = top eigenvector of Qs)
__device__ GL_NOINLINE bool gm_p3_align(double M_end[3][3], double X0, double Y0, double Z0, double X1, double Y1, double Z1, double X2, double Y2,
                     double Z2, double R[3][3], double T[3]) {
  double C_start[3] = {}, C_end[3] = {};
  for (int i = 0; i < 3; i++) C_end[i] = (M_end[0][i] + M_end[1][i] + M_end[2][i]) / 3;
  C_start[0] = (X0 + X1 + X2) / 3;
  C_start[1] = (Y0 + Y1 + Y2) / 3;
  C_start[2] = (Z0 + Z1 + Z2) / 3;
  double s[3 * 3] = {};
  for (int j = 0; j < 3; j++) {
    s[0 * 3 + j] = (X0 * M_end[0][j] + X1 * M_end[1][j] + X2 * M_end[2][j]) / 3 - C_end[j] * C_start[0];
    s[1 * 3 + j] = (Y0 * M_end[0][j] + Y1 * M_end[1][j] + Y2 * M_end[2][j]) / 3 - C_end[j] * C_start[1];
    s[2 * 3 + j] = (Z0 * M_end[0][j] + Z1 * M_end[1][j] + Z2 * M_end[2][j]) / 3 - C_end[j] * C_start[2];
  }
  double Qs[16] = {}, evs[4] = {}, U[16] = {};
  Qs[0 * 4 + 0] = s[0 * 3 + 0] + s[1 * 3 + 1] + s[2 * 3 + 2];
  Qs[1 * 4 + 1] = s[0 * 3 + 0] - s[1 * 3 + 1] - s[2 * 3 + 2];
  Qs[2 * 4 + 2] = s[1 * 3 + 1] - s[2 * 3 + 2] - s[0 * 3 + 0];
  Qs[3 * 4 + 3] = s[2 * 3 + 2] - s[0 * 3 + 0] - s[1 * 3 + 1];
  Qs[1 * 4 + 0] = Qs[0 * 4 + 1] = s[1 * 3 + 2] - s[2 * 3 + 1];
  Qs[2 * 4 + 0] = Qs[0 * 4 + 2] = s[2 * 3 + 0] - s[0 * 3 + 2];
  Qs[3 * 4 + 0] = Qs[0 * 4 + 3] = s[0 * 3 + 1] - s[1 * 3 + 0];
  Qs[2 * 4 + 1] = Qs[1 * 4 + 2] = s[1 * 3 + 0] + s[0 * 3 + 1];
  Qs[3 * 4 + 1] = Qs[1 * 4 + 3] = s[2 * 3 + 0] + s[0 * 3 + 2];
  Qs[3 * 4 + 2] = Qs[2 * 4 + 3] = s[2 * 3 + 1] + s[1 * 3 + 2];
  gm_p3_jacobi_4x4(Qs, evs, U);
  int i_ev = 0;
  double ev_max = evs[i_ev];
  for (int i = 1; i < 4; i++)
    if (evs[i] > ev_max) ev_max = evs[i_ev = i];
  double q[4];
  for (int i = 0; i < 4; i++) q[i] = U[i * 4 + i_ev];
  double q02 = q[0] * q[0], q12 = q[1] * q[1], q22 = q[2] * q[2], q32 = q[3] * q[3];
  double q0_1 = q[0] * q[1], q0_2 = q[0] * q[2], q0_3 = q[0] * q[3];
  double q1_2 = q[1] * q[2], q1_3 = q[1] * q[3];
  double q2_3 = q[2] * q[3];
  R[0][0] = q02 + q12 - q22 - q32;
  R[0][1] = 2. * (q1_2 - q0_3);
  R[0][2] = 2. * (q1_3 + q0_2);
  R[1][0] = 2. * (q1_2 + q0_3);
  R[1][1] = q02 + q22 - q12 - q32;
  R[1][2] = 2. * (q2_3 - q0_1);
  R[2][0] = 2. * (q1_3 - q0_2);
  R[2][1] = 2. * (q2_3 + q0_1);
  R[2][2] = q02 + q32 - q12 - q22;
  for (int i = 0; i < 3; i++) T[i] = C_end[i] - (R[i][0] * C_start[0] + R[i][1] * C_start[1] + R[i][2] * C_start[2]);
  return true;
}

// p3p::solve_for_lengths: |PA|, |PB|, |PC| from the pairwise distances |BC|, |AC|, |AB| and the cosines of the angles
// at the projection centre; up to four solutions (main branch of Gao et al.)
__device__ GL_NOINLINE int gm_p3_solve_for_lengths(double lengths[4][3], double distances[3], double cosines[3]) {
  double p = cosines[0] * 2;
  double q = cosines[1] * 2;
  double r = cosines[2] * 2;
  double inv_d22 = 1. / (distances[2] * distances[2]);
  double a = inv_d22 * (distances[0] * distances[0]);
  double b = inv_d22 * (distances[1] * distances[1]);
  double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
  double pr = p * r, pqr = q * pr;
  if (p2 + q2 + r2 - pqr - 1 == 0) return 0;
  double ab = a * b, a_2 = 2 * a;
  double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
  if (A == 0) return 0;
  double a_4 = 4 * a;
  double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
  double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
  double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
  double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
  double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
  double b0 = b * temp * temp;
  if (b0 == 0) return 0;
  double real_roots[4];
  int n = gm_p3_solve_deg4(A, B, C, D, E, real_roots[0], real_roots[1], real_roots[2], real_roots[3]);
  if (n == 0) return 0;
  int nb_solutions = 0;
  double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
  double inv_b0 = 1. / b0;
  for (int i = 0; i < n; i++) {
    double x = real_roots[i];
    if (x <= 0) continue;
    double x2 = x * x;
    double b1 = ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
                (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
                  (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
                 (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
                  pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
                 2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
                 p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
    if (b1 <= 0) continue;
    double y = inv_b0 * b1;
    double v = x2 + y * y - x * y * r;
    if (v <= 0) continue;
    double Z = distances[2] / sqrt(v);
    double X = x * Z;
    double Y = y * Z;
    lengths[nb_solutions][0] = X;
    lengths[nb_solutions][1] = Y;
    lengths[nb_solutions][2] = Z;
    nb_solutions++;
  }
  return nb_solutions;
}

// p3p::solve with p4p = true.  mu, mv: pixel coordinates (the undistorted normalised point times fx plus cx, as
// p3p::extract_points forms them); X, Y, Z: object points.
__device__ GL_NOINLINE int gm_p3_solve(const CamK& cam, double R[4][3][3], double t[4][3], const double mu_[4], const double mv_[4], const double X[4],
                    const double Y[4], const double Z[4]) {
  const double inv_fx = 1. / cam.fx, inv_fy = 1. / cam.fy, cx_fx = cam.cx / cam.fx, cy_fy = cam.cy / cam.fy;
  double mu[4], mv[4], mk[3];
  for (int i = 0; i < 3; i++) {
    mu[i] = inv_fx * mu_[i] - cx_fx;
    mv[i] = inv_fy * mv_[i] - cy_fy;
    double norm = sqrt(mu[i] * mu[i] + mv[i] * mv[i] + 1);
    mk[i] = 1. / norm; mu[i] *= mk[i]; mv[i] *= mk[i];
  }
  mu[3] = inv_fx * mu_[3] - cx_fx;
  mv[3] = inv_fy * mv_[3] - cy_fy;
  double distances[3];
  distances[0] = sqrt((X[1] - X[2]) * (X[1] - X[2]) + (Y[1] - Y[2]) * (Y[1] - Y[2]) + (Z[1] - Z[2]) * (Z[1] - Z[2]));
  distances[1] = sqrt((X[0] - X[2]) * (X[0] - X[2]) + (Y[0] - Y[2]) * (Y[0] - Y[2]) + (Z[0] - Z[2]) * (Z[0] - Z[2]));
  distances[2] = sqrt((X[0] - X[1]) * (X[0] - X[1]) + (Y[0] - Y[1]) * (Y[0] - Y[1]) + (Z[0] - Z[1]) * (Z[0] - Z[1]));
  double cosines[3];
  cosines[0] = mu[1] * mu[2] + mv[1] * mv[2] + mk[1] * mk[2];
  cosines[1] = mu[0] * mu[2] + mv[0] * mv[2] + mk[0] * mk[2];
  cosines[2] = mu[0] * mu[1] + mv[0] * mv[1] + mk[0] * mk[1];
  double lengths[4][3] = {};
  int n = gm_p3_solve_for_lengths(lengths, distances, cosines);
  int nb_solutions = 0;
  double reproj_errors[4];
  for (int i = 0; i < n; i++) {
    double M_orig[3][3];
    for (int k = 0; k < 3; k++) {
      M_orig[k][0] = lengths[i][k] * mu[k];
      M_orig[k][1] = lengths[i][k] * mv[k];
      M_orig[k][2] = lengths[i][k] * mk[k];
    }
    if (!gm_p3_align(M_orig, X[0], Y[0], Z[0], X[1], Y[1], Z[1], X[2], Y[2], Z[2], R[nb_solutions], t[nb_solutions])) continue;
    {
      double(*Rn)[3] = R[nb_solutions];
      double* tn = t[nb_solutions];
      double X3p = Rn[0][0] * X[3] + Rn[0][1] * Y[3] + Rn[0][2] * Z[3] + tn[0];
      double Y3p = Rn[1][0] * X[3] + Rn[1][1] * Y[3] + Rn[1][2] * Z[3] + tn[1];
      double Z3p = Rn[2][0] * X[3] + Rn[2][1] * Y[3] + Rn[2][2] * Z[3] + tn[2];
      double mu3p = X3p / Z3p;
      double mv3p = Y3p / Z3p;
      reproj_errors[nb_solutions] = (mu3p - mu[3]) * (mu3p - mu[3]) + (mv3p - mv[3]) * (mv3p - mv[3]);
    }
    nb_solutions++;
  }
  for (int i = 1; i < nb_solutions; i++)
    for (int j = i; j > 0 && reproj_errors[j - 1] > reproj_errors[j]; j--) {
      { double tmp_ = reproj_errors[j]; reproj_errors[j] = reproj_errors[j - 1]; reproj_errors[j - 1] = tmp_; }
      for (int k = 0; k < 9; k++) { double tmp_ = R[j][k / 3][k % 3]; R[j][k / 3][k % 3] = R[j - 1][k / 3][k % 3]; R[j - 1][k / 3][k % 3] = tmp_; }
      for (int k = 0; k < 3; k++) { double tmp_ = t[j][k]; t[j][k] = t[j - 1][k]; t[j - 1][k] = tmp_; }
    }
  return nb_solutions;
}

// solvePnP(SOLVEPNP_P3P) on four float correspondences -> rvec, tvec of the solution with the smallest squared
// reprojection error over the four points.  Returns the number of solutions (0: solvePnPRansac returns false).
__device__ GL_NOINLINE int gm_p3p4(const float* obj, const float* img, const CamK& cam, double rvec[3], double tvec[3]) {
  double mu[4], mv[4], X[4], Y[4], Z[4];
  for (int i = 0; i < 4; i++) {
    double xu, yu;
    gm_undistort_point(cam, (double)img[2 * i], (double)img[2 * i + 1], xu, yu);   // cvUndistortPoints writes CV_32FC2
    float xn = (float)xu, yn = (float)yu;
    mu[i] = xn * cam.fx + cam.cx;   // p3p::extract_points
    mv[i] = yn * cam.fy + cam.cy;
    X[i] = obj[3 * i]; Y[i] = obj[3 * i + 1]; Z[i] = obj[3 * i + 2];
  }
  double Rs[4][3][3] = {}, ts[4][3] = {};
  int solutions = gm_p3_solve(cam, Rs, ts, mu, mv, X, Y, Z);
  if (solutions == 0) return 0;
  double rv[4][3], errs[4];
  int order[4] = {0, 1, 2, 3};
  for (int s = 0; s < solutions; s++) {
    double Rm[9];
    for (int k = 0; k < 9; k++) Rm[k] = Rs[s][k / 3][k % 3];
    gm_rodrigues_m2v(Rm, rv[s]);
    double R2[9];
    gm_rodrigues_v2m(rv[s], R2, nullptr);   // projectPoints starts from the rotation vector
    double e = 0;
    for (int i = 0; i < 4; i++) {
      double M[3] = {X[i], Y[i], Z[i]}, m[2];
      gm_project_point(R2, nullptr, ts[s], cam, M, m, nullptr, nullptr);
      double ex = (double)img[2 * i] - m[0], ey = (double)img[2 * i + 1] - m[1];
      e += ex * ex;
      e += ey * ey;
    }
    errs[s] = e;
  }
  for (int i = 1; i < solutions; i++)
    for (int j = i; j > 0 && errs[j - 1] > errs[j]; j--) { double te_ = errs[j]; errs[j] = errs[j - 1]; errs[j - 1] = te_; int to_ = order[j]; order[j] = order[j - 1]; order[j - 1] = to_; }
  for (int k = 0; k < 3; k++) { rvec[k] = rv[order[0]][k]; tvec[k] = ts[order[0]][k]; }
  return solutions;
}


// ---------------------------------------------------------------------------------------------------
// EPnP on 5 points (epnp.cpp) — the solvePnPRansac minimal kernel
// ---------------------------------------------------------------------------------------------------
#define EP_N 5
struct EpnpState {
  double fu, fv, uc, vc;
  double pws[3 * EP_N], us[2 * EP_N], alphas[4 * EP_N], pcs[3 * EP_N];
  double cws[4][3], ccs[4][3];
};

__device__ __forceinline__ double ep_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ double ep_dist2(const double* p1, const double* p2) {
  return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
}

__device__ GL_NOINLINE void ep_qr_solve(double* pA, double* pb, double* pX, int nr, int nc) {
  double A1[6], A2[6];
  double* ppAkk = pA;
  for (int k = 0; k < nc; k++) {
    double *ppAik1 = ppAkk, eta = fabs(*ppAik1);
    for (int i = k + 1; i < nr; i++) {
      double elt = fabs(*ppAik1);
      if (eta < elt) eta = elt;
      ppAik1 += nc;
    }
    if (eta == 0) { A1[k] = A2[k] = 0.0; return; }
    double *ppAik2 = ppAkk, sum2 = 0.0, inv_eta = 1. / eta;
    for (int i = k; i < nr; i++) {
      *ppAik2 *= inv_eta;
      sum2 += *ppAik2 * *ppAik2;
      ppAik2 += nc;
    }
    double sigma = sqrt(sum2);
    if (*ppAkk < 0) sigma = -sigma;
    *ppAkk += sigma;
    A1[k] = sigma * *ppAkk;
    A2[k] = -eta * sigma;
    for (int j = k + 1; j < nc; j++) {
      double *ppAik = ppAkk, sum = 0;
      for (int i = k; i < nr; i++) { sum += *ppAik * ppAik[j - k]; ppAik += nc; }
      double tau = sum / A1[k];
      ppAik = ppAkk;
      for (int i = k; i < nr; i++) { ppAik[j - k] -= tau * *ppAik; ppAik += nc; }
    }
    ppAkk += nc + 1;
  }
  double* ppAjj = pA;
  for (int j = 0; j < nc; j++) {
    double *ppAij = ppAjj, tau = 0;
    for (int i = j; i < nr; i++) { tau += *ppAij * pb[i]; ppAij += nc; }
    tau /= A1[j];
    ppAij = ppAjj;
    for (int i = j; i < nr; i++) { pb[i] -= tau * *ppAij; ppAij += nc; }
    ppAjj += nc + 1;
  }
  pX[nc - 1] = pb[nc - 1] / A2[nc - 1];
  for (int i = nc - 2; i >= 0; i--) {
    double *ppAij = pA + i * nc + (i + 1), sum = 0;
    for (int j = i + 1; j < nc; j++) { sum += *ppAij * pX[j]; ppAij++; }
    pX[i] = (pb[i] - sum) / A2[i];
  }
}

__device__ GL_NOINLINE void ep_gauss_newton(const double* L, const double* rho, double betas[4]) {
  double a[24], b[6], x[4] = {0, 0, 0, 0};
  for (int k = 0; k < 5; k++) {
    for (int i = 0; i < 6; i++) {
      const double* rowL = L + i * 10;
      double* rowA = a + i * 4;
      rowA[0] = 2 * rowL[0] * betas[0] + rowL[1] * betas[1] + rowL[3] * betas[2] + rowL[6] * betas[3];
      rowA[1] = rowL[1] * betas[0] + 2 * rowL[2] * betas[1] + rowL[4] * betas[2] + rowL[7] * betas[3];
      rowA[2] = rowL[3] * betas[0] + rowL[4] * betas[1] + 2 * rowL[5] * betas[2] + rowL[8] * betas[3];
      rowA[3] = rowL[6] * betas[0] + rowL[7] * betas[1] + rowL[8] * betas[2] + 2 * rowL[9] * betas[3];
      b[i] = rho[i] - (rowL[0] * betas[0] * betas[0] + rowL[1] * betas[0] * betas[1] + rowL[2] * betas[1] * betas[1] +
                       rowL[3] * betas[0] * betas[2] + rowL[4] * betas[1] * betas[2] + rowL[5] * betas[2] * betas[2] +
                       rowL[6] * betas[0] * betas[3] + rowL[7] * betas[1] * betas[3] + rowL[8] * betas[2] * betas[3] +
                       rowL[9] * betas[3] * betas[3]);
    }
    ep_qr_solve(a, b, x, 6, 4);
    for (int i = 0; i < 4; i++) betas[i] += x[i];
  }
}

__device__ GL_NOINLINE double ep_compute_R_and_t(EpnpState& e, const double* ut, const double* betas, double R[3][3], double t[3]) {
  const int n = EP_N;
  for (int i = 0; i < 4; i++) e.ccs[i][0] = e.ccs[i][1] = e.ccs[i][2] = 0.0f;
  for (int i = 0; i < 4; i++) {
    const double* v = ut + 12 * (11 - i);
    for (int j = 0; j < 4; j++)
      for (int k = 0; k < 3; k++) e.ccs[j][k] += betas[i] * v[3 * j + k];
  }
  for (int i = 0; i < n; i++) {
    double* a = &e.alphas[4 * i];
    double* pc = &e.pcs[3 * i];
    for (int j = 0; j < 3; j++) pc[j] = a[0] * e.ccs[0][j] + a[1] * e.ccs[1][j] + a[2] * e.ccs[2][j] + a[3] * e.ccs[3][j];
  }
  if (e.pcs[2] < 0.0) {
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 3; j++) e.ccs[i][j] = -e.ccs[i][j];
    for (int i = 0; i < n; i++) { e.pcs[3 * i] = -e.pcs[3 * i]; e.pcs[3 * i + 1] = -e.pcs[3 * i + 1]; e.pcs[3 * i + 2] = -e.pcs[3 * i + 2]; }
  }
  // estimate_R_and_t
  double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 3; j++) { pc0[j] += e.pcs[3 * i + j]; pw0[j] += e.pws[3 * i + j]; }
  for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
  double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, abt_d[3], abt_u[9], abt_vt[9], abt_v[9];
  for (int i = 0; i < n; i++) {
    double* pc = &e.pcs[3 * i];
    double* pw = &e.pws[3 * i];
    for (int j = 0; j < 3; j++) {
      abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
      abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
      abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
    }
  }
  gl_svd3(abt, abt_d, abt_u, abt_vt);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) abt_v[i * 3 + j] = abt_vt[j * 3 + i];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[i][j] = ep_dot(abt_u + 3 * i, abt_v + 3 * j);
  const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                     R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
  if (det < 0) { R[2][0] = -R[2][0]; R[2][1] = -R[2][1]; R[2][2] = -R[2][2]; }
  t[0] = pc0[0] - ep_dot(R[0], pw0);
  t[1] = pc0[1] - ep_dot(R[1], pw0);
  t[2] = pc0[2] - ep_dot(R[2], pw0);
  // reprojection_error
  double sum2 = 0.0;
  for (int i = 0; i < n; i++) {
    double* pw = &e.pws[3 * i];
    double Xc = ep_dot(R[0], pw) + t[0];
    double Yc = ep_dot(R[1], pw) + t[1];
    double inv_Zc = 1.0 / (ep_dot(R[2], pw) + t[2]);
    double ue = e.uc + e.fu * Xc * inv_Zc;
    double ve = e.vc + e.fv * Yc * inv_Zc;
    double u = e.us[2 * i], v = e.us[2 * i + 1];
    sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
  }
  return sum2 / n;
}

// solvePnP(SOLVEPNP_EPNP) for 5 float correspondences -> rvec, tvec
// ws: 288 doubles.  [0,144) MtM, rotated in place into Ut by the SVD; [144,276) first M (2n x 12 = 120), then dv (72) + L (60) and,
// over the dead dv, the workspaces of the three small least-squares solves; [276,288) singular values.
// sub < 0: one lane does everything.  sub = 0, 1, 2: THREE ADJACENT LANES of a wavefront solve one problem together (all three
// call with the same inputs and the same LDS workspace): the 12 x 12 SVD - half of the 0.74 ms of a solve, pinned to its
// sequential pair and summation order by bit-exactness - stays on lane 0, but the three beta approximations with their
// Gauss-Newton and R, t estimation (46 %) are independent given Ut and L and run one per lane, each the same instruction
// sequence as in the one-lane form, with workspaces in the dead parts of the hypothesis's LDS block (dv, rows 0..7 of Ut).
// The winner is chosen in the one-lane order; lane 0 returns it.  Results are bit-identical to the one-lane form.
#ifdef EP_TIMING
__device__ long long g_ep_ticks[8];
#define EP_TICK(k) { long long tn_ = wall_clock64(); if (threadIdx.x == 0 && blockIdx.x == 0) g_ep_ticks[k] += tn_ - tq_; tq_ = tn_; }
#else
#define EP_TICK(k)
#endif
__device__ GL_NOINLINE void gm_epnp5(const float* obj, const float* img, const CamK& cam, double rvec[3], double tvec[3], double* ws, const int sub = -1) {
  const int n = EP_N;
#if defined(__HIP_DEVICE_COMPILE__)
  const bool coop = sub >= 0;
#define EP_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
  const bool coop = false;   // the host checks of this header run the one-lane form
  (void)sub;
#define EP_FENCE() do {} while (0)
#endif
#ifdef EP_TIMING
  long long tq_ = wall_clock64();
#endif
  EpnpState e;
  e.fu = cam.fx; e.fv = cam.fy; e.uc = cam.cx; e.vc = cam.cy;
  for (int i = 0; i < n; i++) {
    e.pws[3 * i] = obj[3 * i]; e.pws[3 * i + 1] = obj[3 * i + 1]; e.pws[3 * i + 2] = obj[3 * i + 2];
    double xu, yu;
    gm_undistort_point(cam, (double)img[2 * i], (double)img[2 * i + 1], xu, yu);
    float xn = (float)xu, yn = (float)yu;
    e.us[2 * i] = xn * e.fu + e.uc;
    e.us[2 * i + 1] = yn * e.fv + e.vc;
  }
  // choose_control_points
  e.cws[0][0] = e.cws[0][1] = e.cws[0][2] = 0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 3; j++) e.cws[0][j] += e.pws[3 * i + j];
  for (int j = 0; j < 3; j++) e.cws[0][j] /= n;
  {
    double pw0[3 * EP_N];
    for (int i = 0; i < n; i++)
      for (int j = 0; j < 3; j++) pw0[3 * i + j] = e.pws[3 * i + j] - e.cws[0][j];
    double pw0tpw0[9], dc[3], U[9];
    for (int a = 0; a < 3; a++)
      for (int b = a; b < 3; b++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += pw0[3 * i + a] * pw0[3 * i + b];
        pw0tpw0[a * 3 + b] = pw0tpw0[b * 3 + a] = s;
      }
    gl_svd3(pw0tpw0, dc, U, nullptr);
    for (int i = 1; i < 4; i++) {
      double k = sqrt(dc[i - 1] / n);
      for (int j = 0; j < 3; j++) e.cws[i][j] = e.cws[0][j] + k * U[j * 3 + (i - 1)];  // uct[3*(i-1)+j] = U[j][i-1]
    }
  }
  // compute_barycentric_coordinates
  {
    double cc[9], ci[9];
    for (int i = 0; i < 3; i++)
      for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = e.cws[j][i] - e.cws[0][i];
    gl_invert3_svd(cc, ci);
    for (int i = 0; i < n; i++) {
      double* pi = &e.pws[3 * i];
      double* a = &e.alphas[4 * i];
      for (int j = 0; j < 3; j++)
        a[1 + j] = ci[3 * j] * (pi[0] - e.cws[0][0]) + ci[3 * j + 1] * (pi[1] - e.cws[0][1]) + ci[3 * j + 2] * (pi[2] - e.cws[0][2]);
      a[0] = 1.0f - a[1] - a[2] - a[3];
    }
  }
  EP_TICK(0)
  // M (2n x 12), MtM, SVD
  double* ut = ws;
  double* ws2 = ws + 144;
  if (!coop || sub == 0) {
    double* M = ws2;
    for (int i = 0; i < n; i++) {
      const double* as = &e.alphas[4 * i];
      double u = e.us[2 * i], v = e.us[2 * i + 1];
      double* M1 = &M[(2 * i) * 12];
      double* M2 = M1 + 12;
      for (int k = 0; k < 4; k++) {
        M1[3 * k] = as[k] * e.fu; M1[3 * k + 1] = 0.0; M1[3 * k + 2] = as[k] * (e.uc - u);
        M2[3 * k] = 0.0; M2[3 * k + 1] = as[k] * e.fv; M2[3 * k + 2] = as[k] * (e.vc - v);
      }
    }
    double* mtm = ws;
    double d[12];
    for (int a = 0; a < 12; a++)
      for (int b = a; b < 12; b++) {
        double s = 0;
        for (int i = 0; i < 2 * n; i++) s += M[i * 12 + a] * M[i * 12 + b];
        mtm[a * 12 + b] = mtm[b * 12 + a] = s;
      }
    // Ut rows = left singular vectors = rows of the rotated A^T: the one-sided Jacobi runs directly on mtm^T, and mtm
    // is bit-for-bit symmetric (both halves are stored from the same sum), so no transposition is needed
    // ut == mtm.  Lanes whose workspace is in LDS take the unrolled 12 x 12 routine; the private-memory lanes of a wide
    // RANSAC round take the general one (same arithmetic)
    // cvSVD(MtM, D, Ut, 0): only the left vectors are read, V is not formed (no value of Ut depends on it)
    if (gl_is_lds(ws)) gl_jacobi_svd12_lds<false>((gl_lds_double*)mtm, (gl_lds_double*)(ws + 276), nullptr);
    else gl_jacobi_svd(mtm, 12, d, nullptr, 12, 12, 12, true);
  }
  EP_TICK(1)
  // dv (4 x 6 x 3) and L_6x10 live in the workspace (the V block is free after the SVD) and their loops stay rolled:
  // as register arrays they pushed this function to ~250 VGPRs
  double* dvm = ws2;         // [4][6][3], dead once L is built
  double* l_6x10 = ws2 + 72; // [6][10]; the small solves below use ws2[0, 61)
  double rho[6];
  if (coop) EP_FENCE();
  if (!coop || sub == 0) {
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
      const double* vi = ut + 12 * (11 - i);
      int a = 0, b = 1;
#pragma unroll 1
      for (int j = 0; j < 6; j++) {
        double* d = dvm + (i * 6 + j) * 3;
        d[0] = vi[3 * a] - vi[3 * b];
        d[1] = vi[3 * a + 1] - vi[3 * b + 1];
        d[2] = vi[3 * a + 2] - vi[3 * b + 2];
        b++;
        if (b > 3) { a++; b = a + 1; }
      }
    }
#pragma unroll 1
    for (int i = 0; i < 6; i++) {
      double* row = l_6x10 + 10 * i;
      const double *d0 = dvm + (0 * 6 + i) * 3, *d1 = dvm + (1 * 6 + i) * 3, *d2 = dvm + (2 * 6 + i) * 3, *d3 = dvm + (3 * 6 + i) * 3;
      row[0] = ep_dot(d0, d0);
      row[1] = 2.0f * ep_dot(d0, d1);
      row[2] = ep_dot(d1, d1);
      row[3] = 2.0f * ep_dot(d0, d2);
      row[4] = 2.0f * ep_dot(d1, d2);
      row[5] = ep_dot(d2, d2);
      row[6] = 2.0f * ep_dot(d0, d3);
      row[7] = 2.0f * ep_dot(d1, d3);
      row[8] = 2.0f * ep_dot(d2, d3);
      row[9] = ep_dot(d3, d3);
    }
  }
  if (coop) EP_FENCE();
  {
    rho[0] = ep_dist2(e.cws[0], e.cws[1]); rho[1] = ep_dist2(e.cws[0], e.cws[2]); rho[2] = ep_dist2(e.cws[0], e.cws[3]);
    rho[3] = ep_dist2(e.cws[1], e.cws[2]); rho[4] = ep_dist2(e.cws[1], e.cws[3]); rho[5] = ep_dist2(e.cws[2], e.cws[3]);
  }
  EP_TICK(2)
  double Betas[4][4], rep_errors[4] = {0, 0, 0, 0};
  double Rs[4][3][3], ts[4][3];
  // workspaces of the three small solves: one after the other in ws2[0, 61) on one lane; side by side when three lanes share
  // the problem: dv[0, 40), dv[40, 67) and rows 0..4 of Ut (dead: R, t only read rows 8..11)
  // ONE instruction stream for the three approximations (lanes that share a problem must not diverge, or they take turns):
  // approximation a solves the 6 x ncol[a] system of its columns of L, turns the solution into betas by its own rule, then
  // Gauss-Newton and R, t - the same calls for all three.  One lane runs a = 0, 1, 2 in turn, three lanes one each.
  for (int a = coop ? sub : 0; a < (coop ? sub + 1 : 3); a++) {
    const int nc = a == 0 ? 4 : (a == 1 ? 3 : 5);
    double* w = !coop ? ws2 : (a == 0 ? ws2 : (a == 1 ? ws2 + 40 : ut));
    double l[30], bb[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < nc; j++) {
        const int col = a == 0 ? (j == 2 ? 3 : (j == 3 ? 6 : j)) : j;   // approximation 1 takes columns 0, 1, 3, 6 of L
        l[i * nc + j] = l_6x10[i * 10 + col];
      }
    gl_solve_svd_ws(l, 6, nc, rho, bb, w, w + 6 * nc);
    double* betas = Betas[1 + a];
    if (a == 0) {   // find_betas_approx_1
      if (bb[0] < 0) { betas[0] = sqrt(-bb[0]); betas[1] = -bb[1] / betas[0]; betas[2] = -bb[2] / betas[0]; betas[3] = -bb[3] / betas[0]; }
      else { betas[0] = sqrt(bb[0]); betas[1] = bb[1] / betas[0]; betas[2] = bb[2] / betas[0]; betas[3] = bb[3] / betas[0]; }
    } else {        // find_betas_approx_2 / _3
      if (bb[0] < 0) { betas[0] = sqrt(-bb[0]); betas[1] = (bb[2] < 0) ? sqrt(-bb[2]) : 0.0; }
      else { betas[0] = sqrt(bb[0]); betas[1] = (bb[2] > 0) ? sqrt(bb[2]) : 0.0; }
      if (bb[1] < 0) betas[0] = -betas[0];
      betas[2] = a == 2 ? bb[3] / betas[0] : 0.0;
      betas[3] = 0.0;
    }
    EP_TICK(3)
    ep_gauss_newton(l_6x10, rho, betas);
    EP_TICK(4)
    rep_errors[1 + a] = ep_compute_R_and_t(e, ut, betas, Rs[1 + a], ts[1 + a]);
    EP_TICK(5)
  }
  EP_TICK(6)
#if defined(__HIP_DEVICE_COMPILE__)
  if (coop) {
    // the three lanes' candidates side by side: errors to every lane, then the winner's R, t to lane 0 (one-lane order of the tests)
    const int base = (int)(threadIdx.x & 63) - sub;
    const double mine = rep_errors[1 + sub];
    const double e1 = __shfl(mine, base, 64), e2 = __shfl(mine, base + 1, 64), e3 = __shfl(mine, base + 2, 64);
    int Nc = 1;
    if (e2 < e1) Nc = 2;
    if (e3 < (Nc == 1 ? e1 : e2)) Nc = 3;
    double R[9];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      tvec[i] = __shfl(ts[1 + sub][i], base + Nc - 1, 64);
#pragma unroll
      for (int j = 0; j < 3; j++) R[i * 3 + j] = __shfl(Rs[1 + sub][i][j], base + Nc - 1, 64);
    }
    if (sub == 0) gm_rodrigues_m2v(R, rvec);
    return;
  }
#endif
  int N = 1;
  if (rep_errors[2] < rep_errors[1]) N = 2;
  if (rep_errors[3] < rep_errors[N]) N = 3;
  double R[9];
  for (int i = 0; i < 3; i++) {
    tvec[i] = ts[N][i];
    for (int j = 0; j < 3; j++) R[i * 3 + j] = Rs[N][i][j];
  }
  gm_rodrigues_m2v(R, rvec);
  EP_TICK(7)
#ifdef EP_TIMING
  if (threadIdx.x == 0 && blockIdx.x == 0)
    printf("EP_TIMING prep %lld mtm+svd12 %lld L %lld approx1 %lld gn1 %lld Rt1 %lld approx2,3 %lld final %lld (100MHz ticks, cumulative)\n", g_ep_ticks[0], g_ep_ticks[1],
           g_ep_ticks[2], g_ep_ticks[3], g_ep_ticks[4], g_ep_ticks[5], g_ep_ticks[6], g_ep_ticks[7]);
#endif
}

#ifndef RS_PNP_CH
#define RS_PNP_CH 12   // EPnP hypotheses per round (20 until round 3: in the bench 12 is +4 %: 28 KB of LDS per stream instead of 46 beside other contexts' LK).  One stream (profiles/tools/pnp_wall.py, RANSAC + refine): 16 / 20 / 32 wide 1.51 / 1.47 /
                       // 1.53 ms at 2-20 % outliers (one round: latency bound whatever the width), 4.2 / 3.2 / 2.5 ms at 40 % (fewer rounds)
#endif
struct PnPModel {
  static constexpr bool SEQ_SCORE = true;    // score + replay hypothesis by hypothesis: the loop usually ends after a few
  static constexpr int LANES = 3;            // lanes per hypothesis in the solver (gm_epnp5: one beta approximation each)
  static constexpr int MP = 5, MAXM = 1, MS = 6, PT1 = 3, PT2 = 2;
  // Every round is 20 wide with the workspaces in LDS (20 x 289 doubles = 46 KB): a round is latency bound, ~0.85 ms
  // whatever its width, while a 64-wide round with 48 workspaces in private memory took 3.8 ms - more per hypothesis
  // than the LDS rounds, and coarser when the iteration bound shrinks mid-way.  24 lanes of the earlier 300-double
  // workspace (62 KB) needed as few rounds, but a 62 KB workgroup waits for LDS beside the image kernels' workgroups
  // once 512 streams are resident (the launch took twice as long); 12 lanes need too many rounds.
  static constexpr int CH = RS_PNP_CH, WS = 144 + 132 + 12;  // MtM -> Ut; M (120), then dv + L (132) / the small solves' workspaces; singular values
  static constexpr bool WIDE = false;
  static constexpr int LMEDS_BELOW = 0;
  static constexpr int MP_ALT = 4;   // four correspondences: P3P on all of them instead of RANSAC
  static constexpr bool OVERDRAW = false;
  __device__ static int solve_alt(const ModelParams& P, const float* ms1, const float* ms2, double* model) {
    double rvec[3], tvec[3];
    if (gm_p3p4(ms1, ms2, P.cam, rvec, tvec) == 0) return 0;
    for (int i = 0; i < 3; i++) { model[2 * i] = rvec[i]; model[2 * i + 1] = tvec[i]; }  // hconcat(rvec, tvec)
    return 1;
  }
  __device__ static bool check_subset(const float*, const float*) { return true; }
  __device__ static int solve(const ModelParams& P, const float* ms1, const float* ms2, double* model, double* ws) {
    double rvec[3], tvec[3];
    gm_epnp5(ms1, ms2, P.cam, rvec, tvec, ws);
    for (int i = 0; i < 3; i++) { model[2 * i] = rvec[i]; model[2 * i + 1] = tvec[i]; }  // hconcat(rvec, tvec)
    return 1;
  }
  // the same by LANES adjacent lanes (lane `sub` of them): the model is valid on sub == 0
  __device__ static int solve_coop(const ModelParams& P, const float* ms1, const float* ms2, double* model, double* ws, int sub) {
    double rvec[3] = {0, 0, 0}, tvec[3] = {0, 0, 0};
    gm_epnp5(ms1, ms2, P.cam, rvec, tvec, ws, sub);
    for (int i = 0; i < 3; i++) { model[2 * i] = rvec[i]; model[2 * i + 1] = tvec[i]; }
    return 1;
  }
  struct Scorer {
    double R[9], t[3];
    CamK cam;
    __device__ void init(const ModelParams& P, const double* model) {
      double rvec[3] = {model[0], model[2], model[4]};
      t[0] = model[1]; t[1] = model[3]; t[2] = model[5];
      gm_rodrigues_v2m(rvec, R, nullptr);
      cam = P.cam;
    }
    __device__ __forceinline__ float err(const float* M3, const float* m2) const {
      double M[3] = {M3[0], M3[1], M3[2]}, m[2];
      gm_project_point(R, nullptr, t, cam, M, m, nullptr, nullptr);
      float px = (float)m[0], py = (float)m[1];
      float dx = m2[0] - px, dy = m2[1] - py;
      float s = 0;
      s += dx * dx;
      s += dy * dy;
      return s;
    }
  };
};

// ---------------------------------------------------------------------------------------------------
// Essential matrix, 5 points (five-point.cpp EMEstimatorCallback) — Nister's solver.
// Points are pixel floats; each use normalises them with K in double exactly like findEssentialMat
// ((double)p - c) / f.  The 10x20 constraint matrix and det B(z) are expanded symbolically (same
// arithmetic as oracle/orc_essential.cpp); roots by Durand-Kerner as cv::solvePoly.
// ---------------------------------------------------------------------------------------------------
struct EmPoly { double c[20]; };

__device__ __constant__ signed char kEmMono[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1}, {0, 2, 0}, {1, 1, 1}, {1, 1, 0},
                                                      {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};

__device__ GL_NOINLINE int em_mono_index(int a, int b, int c) {
  for (int i = 0; i < 20; i++)
    if (kEmMono[i][0] == a && kEmMono[i][1] == b && kEmMono[i][2] == c) return i;
  return -1;
}
__device__ GL_NOINLINE void em_pmul(const EmPoly& a, const EmPoly& b, EmPoly& r) {
  double out[20];
  for (int i = 0; i < 20; i++) out[i] = 0;
  for (int i = 0; i < 20; i++) {
    if (a.c[i] == 0) continue;
    for (int j = 0; j < 20; j++) {
      if (b.c[j] == 0) continue;
      int e0 = kEmMono[i][0] + kEmMono[j][0], e1 = kEmMono[i][1] + kEmMono[j][1], e2 = kEmMono[i][2] + kEmMono[j][2];
      if (e0 + e1 + e2 > 3) continue;
      out[em_mono_index(e0, e1, e2)] += a.c[i] * b.c[j];
    }
  }
  for (int i = 0; i < 20; i++) r.c[i] = out[i];
}
__device__ __forceinline__ void em_padd(const EmPoly& a, const EmPoly& b, EmPoly& r) { for (int i = 0; i < 20; i++) r.c[i] = a.c[i] + b.c[i]; }
__device__ __forceinline__ void em_psub(const EmPoly& a, const EmPoly& b, EmPoly& r) { for (int i = 0; i < 20; i++) r.c[i] = a.c[i] - b.c[i]; }

// hal::LU64f with right-hand side (partial pivoting); m = n = 10 here
__device__ GL_NOINLINE int em_lu_solve(double* A, int m, double* b, int n) {
  const double eps = DBL_EPSILON * 100;
  int i, j, k, p = 1;
  for (i = 0; i < m; i++) {
    k = i;
    for (j = i + 1; j < m; j++)
      if (fabs(A[j * m + i]) > fabs(A[k * m + i])) k = j;
    if (fabs(A[k * m + i]) < eps) return 0;
    if (k != i) {
      for (j = i; j < m; j++) gl_swap(A[i * m + j], A[k * m + j]);
      for (j = 0; j < n; j++) gl_swap(b[i * n + j], b[k * n + j]);
      p = -p;
    }
    double d = -1 / A[i * m + i];
    for (j = i + 1; j < m; j++) {
      double alpha = A[j * m + i] * d;
      for (k = i + 1; k < m; k++) A[j * m + k] += alpha * A[i * m + k];
      for (k = 0; k < n; k++) b[j * n + k] += alpha * b[i * n + k];
    }
  }
  for (i = m - 1; i >= 0; i--)
    for (j = 0; j < n; j++) {
      double s = b[i * n + j];
      for (k = i + 1; k < m; k++) s -= A[i * m + k] * b[k * n + j];
      b[i * n + j] = s / A[i * m + i];
    }
  return p;
}

// cv::solvePoly (Durand-Kerner), ascending coefficients, degree n0 = 10
__device__ GL_NOINLINE int em_solve_poly(const double* c, int n0, double* re, double* im) {
  int n = n0;
  for (; n > 1; n--)
    if (fabs(c[n]) > DBL_EPSILON) break;
  double pr = 1, pi = 0;
  for (int i = 0; i < n; i++) {
    re[i] = pr; im[i] = pi;
    double t = pr * 1 - pi * 1;
    pi = pr * 1 + pi * 1;
    pr = t;
  }
  for (int iter = 0; iter < 1000; iter++) {
    double maxDiff = 0;
    for (int i = 0; i < n; i++) {
      double p_re = re[i], p_im = im[i];
      double num_re = c[n], num_im = 0, den_re = c[n], den_im = 0;
      for (int j = 0; j < n; j++) {
        double t = num_re * p_re - num_im * p_im;
        num_im = num_re * p_im + num_im * p_re;
        num_re = t + c[n - j - 1];
        if (j != i) {
          double d_re = p_re - re[j], d_im = p_im - im[j];
          if (d_re != 0 || d_im != 0) {
            double t2 = den_re * d_re - den_im * d_im;
            den_im = den_re * d_im + den_im * d_re;
            den_re = t2;
          }
        }
      }
      double tt = 1. / (den_re * den_re + den_im * den_im);
      double q_re = (num_re * den_re + num_im * den_im) * tt;
      double q_im = (-num_re * den_im + num_im * den_re) * tt;
      re[i] = p_re - q_re; im[i] = p_im - q_im;
      maxDiff = fmax(maxDiff, sqrt(q_re * q_re + q_im * q_im));
    }
    if (maxDiff <= 0) break;
  }
  for (int i = 0; i < n; i++)
    if (fabs(im[i]) < 1e-100) im[i] = 0;
  for (int k = n; k < n0; k++) { re[k] = re[k - 1]; im[k] = im[k - 1]; }
  return n;
}

__device__ GL_NOINLINE int em_solve5(const double* q1, const double* q2, double* models) {
  const int n = 5;
  double Q[45];
  for (int i = 0; i < n; i++) {
    double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
    double* q = &Q[i * 9];
    q[0] = x2 * x1; q[1] = x2 * y1; q[2] = x2; q[3] = y2 * x1; q[4] = y2 * y1; q[5] = y2; q[6] = x1; q[7] = y1; q[8] = 1.0;
  }
  double w[9], Vt[81], ta[81], tv[25];
  gl_svd_compute(Q, n, 9, w, nullptr, Vt, true, ta, tv);
  const double* EE[4] = {Vt + 45, Vt + 54, Vt + 63, Vt + 72};
  EmPoly E[9];
  for (int k = 0; k < 9; k++) {
    for (int i = 0; i < 20; i++) E[k].c[i] = 0;
    E[k].c[12] = EE[0][k]; E[k].c[15] = EE[1][k]; E[k].c[18] = EE[2][k]; E[k].c[19] = EE[3][k];  // x, y, z, 1
  }
  EmPoly rows[10], t1, t2, t3;
  // det(E) = E0 (E4 E8 - E5 E7) - E1 (E3 E8 - E5 E6) + E2 (E3 E7 - E4 E6)
  em_pmul(E[4], E[8], t1); em_pmul(E[5], E[7], t2); em_psub(t1, t2, t1); em_pmul(E[0], t1, rows[0]);
  em_pmul(E[3], E[8], t1); em_pmul(E[5], E[6], t2); em_psub(t1, t2, t1); em_pmul(E[1], t1, t3); em_psub(rows[0], t3, rows[0]);
  em_pmul(E[3], E[7], t1); em_pmul(E[4], E[6], t2); em_psub(t1, t2, t1); em_pmul(E[2], t1, t3); em_padd(rows[0], t3, rows[0]);
  EmPoly EEt[9], tr;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      em_pmul(E[i * 3], E[j * 3], t1); em_pmul(E[i * 3 + 1], E[j * 3 + 1], t2); em_padd(t1, t2, t1);
      em_pmul(E[i * 3 + 2], E[j * 3 + 2], t2); em_padd(t1, t2, EEt[i * 3 + j]);
    }
  em_padd(EEt[0], EEt[4], tr); em_padd(tr, EEt[8], tr);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      em_pmul(EEt[i * 3], E[j], t1); em_pmul(EEt[i * 3 + 1], E[3 + j], t2); em_padd(t1, t2, t1);
      em_pmul(EEt[i * 3 + 2], E[6 + j], t2); em_padd(t1, t2, t1);
      for (int k = 0; k < 20; k++) t1.c[k] = t1.c[k] * 2.0;
      em_pmul(tr, E[i * 3 + j], t2);
      em_psub(t1, t2, rows[1 + i * 3 + j]);
    }
  double A1[100], A2[100];
  for (int r = 0; r < 10; r++)
    for (int c = 0; c < 10; c++) { A1[r * 10 + c] = rows[r].c[c]; A2[r * 10 + c] = rows[r].c[10 + c]; }
  if (!em_lu_solve(A1, 10, A2, 10)) return 0;
  const double* A = A2;
  double b[39];
  for (int i = 0; i < 3; i++) {
    const double* a1 = A + (i * 2 + 4) * 10;
    const double* a2 = A + (i * 2 + 5) * 10;
    double row1[13], row2[13];
    for (int k = 0; k < 13; k++) { row1[k] = 0; row2[k] = 0; }
    for (int k = 0; k < 3; k++) { row1[1 + k] = a1[k]; row1[5 + k] = a1[3 + k]; row2[k] = a2[k]; row2[4 + k] = a2[3 + k]; }
    for (int k = 0; k < 4; k++) { row1[9 + k] = a1[6 + k]; row2[8 + k] = a2[6 + k]; }
    for (int k = 0; k < 13; k++) b[i * 13 + k] = row1[k] - row2[k];
  }
  double P[3][3][5];
  const int deg[3] = {4, 4, 5};
  for (int j = 0; j < 3; j++) {
    const double* br = b + j * 13;
    for (int k = 0; k < 4; k++) { P[j][0][k] = br[3 - k]; P[j][1][k] = br[7 - k]; }
    P[j][0][4] = P[j][1][4] = 0;
    for (int k = 0; k < 5; k++) P[j][2][k] = br[12 - k];
  }
  double c[11];
  for (int k = 0; k < 11; k++) c[k] = 0;
  const int perm[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};
  const double sgn[6] = {1, 1, 1, -1, -1, -1};
  for (int q = 0; q < 6; q++) {
    double u1[9], u2[13];
    const double* pa = P[0][perm[q][0]]; int na = deg[perm[q][0]];
    const double* pb = P[1][perm[q][1]]; int nb = deg[perm[q][1]];
    const double* pc = P[2][perm[q][2]]; int nc = deg[perm[q][2]];
    for (int i = 0; i < na + nb - 1; i++) u1[i] = 0;
    for (int i = 0; i < na; i++)
      for (int j = 0; j < nb; j++) u1[i + j] += pa[i] * pb[j];
    int n1 = na + nb - 1;
    for (int i = 0; i < n1 + nc - 1; i++) u2[i] = 0;
    for (int i = 0; i < n1; i++)
      for (int j = 0; j < nc; j++) u2[i + j] += u1[i] * pc[j];
    for (int k = 0; k < 11; k++) c[k] += sgn[q] * u2[k];
  }
  double rre[16], rim[16];
  em_solve_poly(c, 10, rre, rim);
  int count = 0;
  for (int i = 0; i < 10; i++) {
    if (fabs(rim[i]) > 1e-10) continue;
    double z1 = rre[i], z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
    double bz[9];
    for (int j = 0; j < 3; j++) {
      const double* br = b + j * 13;
      bz[j * 3] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
      bz[j * 3 + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
      bz[j * 3 + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
    }
    double ww[3], vt[9];
    gl_svd3(bz, ww, nullptr, vt);
    const double* xy1 = vt + 6;
    if (fabs(xy1[2]) < 1e-10) continue;
    double xs = xy1[0] / xy1[2], ys = xy1[1] / xy1[2], zs = z1;
    double Ev[9], nrm = 0;
    for (int k = 0; k < 9; k++) {
      Ev[k] = EE[0][k] * xs + EE[1][k] * ys + EE[2][k] * zs + EE[3][k];
      nrm += Ev[k] * Ev[k];
    }
    nrm = sqrt(nrm);
    for (int k = 0; k < 9; k++) models[count * 9 + k] = Ev[k] / nrm;
    count++;
  }
  return count;
}

struct EModel {
  static constexpr int LANES = 1;
  static constexpr bool SEQ_SCORE = false;   // score all hypotheses of a round, then replay (geom.hip)
  static constexpr int MP = 5, MAXM = 10, MS = 9, PT1 = 2, PT2 = 2;
  static constexpr int CH = 64, WS = 0;  // initialisation only (src/initializer.cpp), not on the per-frame path: private memory
  static constexpr bool WIDE = true;
  static constexpr int LMEDS_BELOW = 0;
  static constexpr int MP_ALT = 0;   // no second sample size
  static constexpr bool OVERDRAW = false;
  __device__ static bool check_subset(const float*, const float*) { return true; }
  __device__ static int solve(const ModelParams& P, const float* ms1, const float* ms2, double* models, double*) {
    double q1[10], q2[10];
    for (int i = 0; i < 5; i++) {
      q1[2 * i] = ((double)ms1[2 * i] - P.cam.cx) / P.cam.fx; q1[2 * i + 1] = ((double)ms1[2 * i + 1] - P.cam.cy) / P.cam.fy;
      q2[2 * i] = ((double)ms2[2 * i] - P.cam.cx) / P.cam.fx; q2[2 * i + 1] = ((double)ms2[2 * i + 1] - P.cam.cy) / P.cam.fy;
    }
    return em_solve5(q1, q2, models);
  }
  struct Scorer {
    double E[9];
    CamK cam;
    __device__ void init(const ModelParams& P, const double* e) {
      for (int i = 0; i < 9; i++) E[i] = e[i];
      cam = P.cam;
    }
    __device__ __forceinline__ float err(const float* p1, const float* p2) const {
      double x1[3] = {((double)p1[0] - cam.cx) / cam.fx, ((double)p1[1] - cam.cy) / cam.fy, 1.};
      double x2[3] = {((double)p2[0] - cam.cx) / cam.fx, ((double)p2[1] - cam.cy) / cam.fy, 1.};
      double Ex1[3], Etx2[3];
      for (int r = 0; r < 3; r++) {
        Ex1[r] = E[r * 3] * x1[0] + E[r * 3 + 1] * x1[1] + E[r * 3 + 2] * x1[2];
        Etx2[r] = E[r] * x2[0] + E[3 + r] * x2[1] + E[6 + r] * x2[2];
      }
      double x2tEx1 = x2[0] * Ex1[0] + x2[1] * Ex1[1] + x2[2] * Ex1[2];
      double a = Ex1[0] * Ex1[0], b = Ex1[1] * Ex1[1], c = Etx2[0] * Etx2[0], d = Etx2[1] * Etx2[1];
      return (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
    }
  };
};
