// oracle/orc_linalg.h — TEST INFRASTRUCTURE ONLY (see orc_common.h).  PARITY UNPINNED.
//
// Restatement of the OpenCV-4.6 core numerics the calib3d path runs on (core/src/lapack.cpp:
// JacobiImpl_, JacobiSVDImpl_, SVBkSbImpl_; core/src/mathfuncs.cpp: solveCubic), i.e. the non-LAPACK,
// non-Eigen code path of OpenCV.  Row-major dense doubles, sizes given at run time.
//
// One deliberate difference: std::hypot is replaced by the explicit scaled formula below so that the
// CPU oracle and the GPU run the same IEEE operation sequence (libm's hypot is not available on the
// device); it differs from glibc's by at most an ulp.
#pragma once
#include "orc_common.h"

namespace orc {

static inline double pm_hypot(double a, double b) {
  a = std::fabs(a); b = std::fabs(b);
  if (a < b) std::swap(a, b);
  if (a == 0.0) return 0.0;
  double t = b / a;
  return a * std::sqrt(1.0 + t * t);
}

// cv::eigen for symmetric matrices — JacobiImpl_ (max-pivot cyclic Jacobi with row/column index caches).
// A: n x n (destroyed), W: n eigenvalues (descending), V: n x n, eigenvectors in ROWS.
static inline void jacobi_eigen(double* A, int n, double* W, double* V) {
  const double eps = DBL_EPSILON;
  int i, j, k, m;
  for (i = 0; i < n; i++) {
    for (j = 0; j < n; j++) V[i * n + j] = 0;
    V[i * n + i] = 1;
  }
  int iters, maxIters = n * n * 30;
  std::vector<int> indR(n), indC(n);
  double mv = 0;
  for (k = 0; k < n; k++) {
    W[k] = A[(n + 1) * k];
    if (k < n - 1) {
      for (m = k + 1, mv = std::fabs(A[n * k + m]), i = k + 2; i < n; i++) {
        double val = std::fabs(A[n * k + i]);
        if (mv < val) mv = val, m = i;
      }
      indR[k] = m;
    }
    if (k > 0) {
      for (m = 0, mv = std::fabs(A[k]), i = 1; i < k; i++) {
        double val = std::fabs(A[n * i + k]);
        if (mv < val) mv = val, m = i;
      }
      indC[k] = m;
    }
  }
  if (n > 1)
    for (iters = 0; iters < maxIters; iters++) {
      for (k = 0, mv = std::fabs(A[indR[0]]), i = 1; i < n - 1; i++) {
        double val = std::fabs(A[n * i + indR[i]]);
        if (mv < val) mv = val, k = i;
      }
      int l = indR[k];
      for (i = 1; i < n; i++) {
        double val = std::fabs(A[n * indC[i] + i]);
        if (mv < val) mv = val, k = indC[i], l = i;
      }
      double p = A[n * k + l];
      if (std::fabs(p) <= eps) break;
      double y = (W[l] - W[k]) * 0.5;
      double t = std::fabs(y) + pm_hypot(p, y);
      double s = pm_hypot(p, t);
      double c = t / s;
      s = p / s;
      t = (p / t) * p;
      if (y < 0) s = -s, t = -t;
      A[n * k + l] = 0;
      W[k] -= t;
      W[l] += t;
      double a0, b0;
#define ORC_ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
      for (i = 0; i < k; i++) ORC_ROT(A[n * i + k], A[n * i + l]);
      for (i = k + 1; i < l; i++) ORC_ROT(A[n * k + i], A[n * i + l]);
      for (i = l + 1; i < n; i++) ORC_ROT(A[n * k + i], A[n * l + i]);
      for (i = 0; i < n; i++) ORC_ROT(V[n * k + i], V[n * l + i]);
#undef ORC_ROT
      for (j = 0; j < 2; j++) {
        int idx = j == 0 ? k : l;
        if (idx < n - 1) {
          for (m = idx + 1, mv = std::fabs(A[n * idx + m]), i = idx + 2; i < n; i++) {
            double val = std::fabs(A[n * idx + i]);
            if (mv < val) mv = val, m = i;
          }
          indR[idx] = m;
        }
        if (idx > 0) {
          for (m = 0, mv = std::fabs(A[idx]), i = 1; i < idx; i++) {
            double val = std::fabs(A[n * i + idx]);
            if (mv < val) mv = val, m = i;
          }
          indC[idx] = m;
        }
      }
    }
  for (k = 0; k < n - 1; k++) {
    m = k;
    for (i = k + 1; i < n; i++)
      if (W[m] < W[i]) m = i;
    if (k != m) {
      std::swap(W[m], W[k]);
      for (i = 0; i < n; i++) std::swap(V[n * m + i], V[n * k + i]);
    }
  }
}

// JacobiSVDImpl_<double>: one-sided Jacobi on At (n rows of length m, row stride astep).
// On return rows 0..n1-1 of At are left singular vectors (rows n..n1-1 completed by the seeded
// Gram-Schmidt fill), W descending, Vt (n x n, stride n) right singular vectors in rows.
static inline void jacobi_svd(double* At, int astep, double* Wout, double* Vt, int m, int n, int n1) {
  const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
  std::vector<double> W(n);
  int i, j, k, iter, max_iter = std::max(m, 30);
  double c, s, sd;
  for (i = 0; i < n; i++) {
    for (k = 0, sd = 0; k < m; k++) {
      double t = At[i * astep + k];
      sd += t * t;
    }
    W[i] = sd;
    if (Vt) {
      for (k = 0; k < n; k++) Vt[i * n + k] = 0;
      Vt[i * n + i] = 1;
    }
  }
  for (iter = 0; iter < max_iter; iter++) {
    bool changed = false;
    for (i = 0; i < n - 1; i++)
      for (j = i + 1; j < n; j++) {
        double *Ai = At + i * astep, *Aj = At + j * astep;
        double a = W[i], p = 0, b = W[j];
        for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
        if (std::fabs(p) <= eps * std::sqrt(a * b)) continue;
        p *= 2;
        double beta = a - b, gamma = pm_hypot(p, beta);
        if (beta < 0) {
          double delta = (gamma - beta) * 0.5;
          s = std::sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = std::sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
        for (k = 0; k < m; k++) {
          double t0 = c * Ai[k] + s * Aj[k];
          double t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0; Aj[k] = t1;
          a += t0 * t0; b += t1 * t1;
        }
        W[i] = a; W[j] = b;
        changed = true;
        if (Vt) {
          double *Vi = Vt + i * n, *Vj = Vt + j * n;
          for (k = 0; k < n; k++) {
            double t0 = c * Vi[k] + s * Vj[k];
            double t1 = -s * Vi[k] + c * Vj[k];
            Vi[k] = t0; Vj[k] = t1;
          }
        }
      }
    if (!changed) break;
  }
  for (i = 0; i < n; i++) {
    for (k = 0, sd = 0; k < m; k++) {
      double t = At[i * astep + k];
      sd += t * t;
    }
    W[i] = std::sqrt(sd);
  }
  for (i = 0; i < n - 1; i++) {
    j = i;
    for (k = i + 1; k < n; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      std::swap(W[i], W[j]);
      if (Vt) {
        for (k = 0; k < m; k++) std::swap(At[i * astep + k], At[j * astep + k]);
        for (k = 0; k < n; k++) std::swap(Vt[i * n + k], Vt[j * n + k]);
      }
    }
  }
  for (i = 0; i < n; i++) Wout[i] = W[i];
  if (!Vt) return;
  RNG rng(0x12345678);
  for (i = 0; i < n1; i++) {
    sd = i < n ? W[i] : 0;
    for (int ii = 0; ii < 100 && sd <= minval; ii++) {
      const double val0 = 1. / m;
      for (k = 0; k < m; k++) {
        double val = (rng.next() & 256) != 0 ? val0 : -val0;
        At[i * astep + k] = val;
      }
      for (iter = 0; iter < 2; iter++) {
        for (j = 0; j < i; j++) {
          sd = 0;
          for (k = 0; k < m; k++) sd += At[i * astep + k] * At[j * astep + k];
          double asum = 0;
          for (k = 0; k < m; k++) {
            double t = At[i * astep + k] - sd * At[j * astep + k];
            At[i * astep + k] = t;
            asum += std::fabs(t);
          }
          asum = asum > eps * 100 ? 1 / asum : 0;
          for (k = 0; k < m; k++) At[i * astep + k] *= asum;
        }
      }
      sd = 0;
      for (k = 0; k < m; k++) {
        double t = At[i * astep + k];
        sd += t * t;
      }
      sd = std::sqrt(sd);
    }
    s = sd > minval ? 1 / sd : 0.;
    for (k = 0; k < m; k++) At[i * astep + k] *= s;
  }
}

// cv::SVD::compute(A (m x n), w, u, vt[, FULL_UV]).  Ut: rows are left singular vectors
// (urows x m where urows = full ? max(m,n) : min(m,n), only meaningful when m >= n);
// for m < n the roles swap exactly as in _SVDcompute.  Outputs:
//   w[min(m,n)], U (m x ucols row-major, may be null), Vt (vrows x n row-major, may be null)
static inline void svd_compute(const double* A, int m, int n, double* w, double* U, double* Vt, bool full_uv) {
  bool at = false;
  int mm = m, nn = n;
  if (mm < nn) { std::swap(mm, nn); at = true; }
  int urows = full_uv ? mm : nn;
  std::vector<double> ta((size_t)urows * mm, 0.0), tv((size_t)nn * nn), tw(nn);
  if (!at) {
    for (int i = 0; i < m; i++)
      for (int j = 0; j < n; j++) ta[(size_t)j * mm + i] = A[i * n + j];  // temp_a = A^T (n x m)
  } else {
    for (int i = 0; i < m; i++)
      for (int j = 0; j < n; j++) ta[(size_t)i * mm + j] = A[i * n + j];  // temp_a = A (m x n) = (nn x mm)
  }
  jacobi_svd(ta.data(), mm, tw.data(), tv.data(), mm, nn, urows);
  for (int i = 0; i < nn; i++) w[i] = tw[i];
  if (!at) {
    // u = temp_u^T (m x urows), vt = temp_v (n x n)
    if (U)
      for (int i = 0; i < urows; i++)
        for (int k = 0; k < mm; k++) U[(size_t)k * urows + i] = ta[(size_t)i * mm + k];
    if (Vt) memcpy(Vt, tv.data(), sizeof(double) * nn * nn);
  } else {
    // u = temp_v^T (m x m), vt = temp_u (urows x n)
    if (U)
      for (int i = 0; i < nn; i++)
        for (int k = 0; k < nn; k++) U[(size_t)k * nn + i] = tv[(size_t)i * nn + k];
    if (Vt) memcpy(Vt, ta.data(), sizeof(double) * urows * mm);
  }
}

// SVBkSbImpl_: x (n x nb) = V diag(1/w) U^T b, with U given as rows-of-Ut (uT = true) and V as rows of Vt.
// b == nullptr -> pseudo-inverse (nb = m).
static inline void svbksb(int m, int n, const double* w, const double* Ut, int ldu, const double* Vt, int ldv,
                          const double* b, int ldb, int nb, double* x, int ldx) {
  const double eps = DBL_EPSILON * 2;
  double threshold = 0;
  int nm = std::min(m, n);
  if (!b) nb = m;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < nb; j++) x[i * ldx + j] = 0;
  for (int i = 0; i < nm; i++) threshold += w[i];
  threshold *= eps;
  std::vector<double> buffer(nb);
  for (int i = 0; i < nm; i++) {
    const double* u = Ut + (size_t)i * ldu;
    const double* v = Vt + (size_t)i * ldv;
    double wi = w[i];
    if (std::fabs(wi) <= threshold) continue;
    wi = 1 / wi;
    if (nb == 1) {
      double s = 0;
      if (b)
        for (int j = 0; j < m; j++) s += u[j] * b[j * ldb];
      else
        s = u[0];
      s *= wi;
      for (int j = 0; j < n; j++) x[j * ldx] = x[j * ldx] + s * v[j];
    } else {
      if (b) {
        for (int j = 0; j < nb; j++) buffer[j] = 0;
        for (int k = 0; k < m; k++) {  // MatrAXPY(m, nb, b, ldb, u, 1, buffer, 0)
          double s = u[k];
          for (int j = 0; j < nb; j++) buffer[j] = buffer[j] + s * b[k * ldb + j];
        }
        for (int j = 0; j < nb; j++) buffer[j] *= wi;
      } else {
        for (int j = 0; j < nb; j++) buffer[j] = u[j] * wi;
      }
      for (int k = 0; k < n; k++) {  // MatrAXPY(n, nb, buffer, 0, v, 1, x, ldx)
        double s = v[k];
        for (int j = 0; j < nb; j++) x[k * ldx + j] = x[k * ldx + j] + s * buffer[j];
      }
    }
  }
}

// cv::solve(A (m x n, m >= n), b (m x 1), x, DECOMP_SVD)
static inline void solve_svd(const double* A, int m, int n, const double* b, double* x) {
  std::vector<double> at((size_t)n * m), w(n), vt((size_t)n * n);
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) at[(size_t)j * m + i] = A[i * n + j];
  jacobi_svd(at.data(), m, w.data(), vt.data(), m, n, n);
  svbksb(m, n, w.data(), at.data(), m, vt.data(), n, b, 1, 1, x, 1);
}

// cv::invert(A (n x n), DECOMP_SVD)
static inline void invert_svd(const double* A, int n, double* Ainv) {
  std::vector<double> at((size_t)n * n), w(n), vt((size_t)n * n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) at[(size_t)j * n + i] = A[i * n + j];
  jacobi_svd(at.data(), n, w.data(), vt.data(), n, n, n);
  svbksb(n, n, w.data(), at.data(), n, vt.data(), n, nullptr, 0, n, Ainv, n);
}

// cv::solveCubic for 4 double coefficients; returns the root count (roots in x[0..2]).
static inline int solve_cubic(const double c[4], double x[3]) {
  double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
  double x0 = 0., x1 = 0., x2 = 0.;
  int n = 0;
  if (a0 == 0) {
    if (a1 == 0) {
      if (a2 == 0) n = a3 == 0 ? -1 : 0;
      else { x0 = -a3 / a2; n = 1; }
    } else {
      double d = a2 * a2 - 4 * a1 * a3;
      if (d >= 0) {
        d = std::sqrt(d);
        double q1 = (-a2 + d) * 0.5;
        double q2 = (a2 + d) * -0.5;
        if (std::fabs(q1) > std::fabs(q2)) { x0 = q1 / a1; x1 = a3 / q1; }
        else { x0 = q2 / a1; x1 = a3 / q2; }
        n = d > 0 ? 2 : 1;
      }
    }
  } else {
    a0 = 1. / a0;
    a1 *= a0; a2 *= a0; a3 *= a0;
    double Q = (a1 * a1 - 3 * a2) * (1. / 9);
    double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
    double Qcubed = Q * Q * Q;
    double d = Qcubed - R * R;
    if (d > 0) {
      double theta = std::acos(R / std::sqrt(Qcubed));
      double sqrtQ = std::sqrt(Q);
      double t0 = -2 * sqrtQ;
      double t1 = theta * (1. / 3);
      double t2 = a1 * (1. / 3);
      x0 = t0 * std::cos(t1) - t2;
      x1 = t0 * std::cos(t1 + (2. * M_PI / 3)) - t2;
      x2 = t0 * std::cos(t1 + (4. * M_PI / 3)) - t2;
      n = 3;
    } else if (d == 0) {
      if (R >= 0) {
        x0 = -2 * std::pow(R, 1. / 3) - a1 / 3;
        x1 = std::pow(R, 1. / 3) - a1 / 3;
      } else {
        x0 = 2 * std::pow(-R, 1. / 3) - a1 / 3;
        x1 = -std::pow(-R, 1. / 3) - a1 / 3;
      }
      x2 = 0;
      n = x0 == x1 ? 1 : 2;
      x1 = x0 == x1 ? 0 : x1;
    } else {
      double e;
      d = std::sqrt(-d);
      e = std::pow(d + std::fabs(R), 1. / 3);
      if (R > 0) e = -e;
      x0 = (e + Q / e) - a1 * (1. / 3);
      n = 1;
    }
  }
  x[0] = x0; x[1] = x1; x[2] = x2;
  return n;
}

static inline double det3(const double* M) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
static inline void mat3mul(const double* A, const double* B, double* C) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  memcpy(C, t, sizeof(t));
}

}  // namespace orc
