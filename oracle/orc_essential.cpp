// oracle/orc_essential.cpp — TEST INFRASTRUCTURE ONLY (see orc_common.h).  PARITY UNPINNED.
//
// CPU restatement of cv::findEssentialMat(p1, p2, K, RANSAC, 0.99, 1.0, mask) as called by the reference
// at src/initializer.cpp:228-229 (OpenCV 4.6 calib3d/src/five-point.cpp): points
// normalised by K in double, threshold /= (fx+fy)/2, RANSAC with 5-point samples (max 1000 iterations),
// Nister's solver (null space of the 5x9 epipolar system, 10 cubic constraints, elimination to a
// 10th-degree polynomial in z, Durand-Kerner roots as cv::solvePoly), Sampson error in float.
//
// Where OpenCV uses machine-generated expansions this file derives the same quantities symbolically:
//   * getCoeffMat (10x20 constraint matrix): polynomial expansion of det(E) and 2EE'E - tr(EE')E in the
//     monomial order x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1;
//   * the 11 coefficients of det B(z): polynomial products.
// Both agree with OpenCV's expressions up to floating-point rounding (row order / association differ).
// cv::solvePoly's repeated-root special case is not restated.
#include "orc_common.h"
#include "orc_linalg.h"
#include "orc_ransac.h"
#include "mvo_oracle.h"

namespace orc {

// monomial order (exponents of x, y, z)
static const int kMono[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1}, {0, 2, 0}, {1, 1, 1}, {1, 1, 0},
                                 {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
static int mono_index(int a, int b, int c) {
  for (int i = 0; i < 20; i++)
    if (kMono[i][0] == a && kMono[i][1] == b && kMono[i][2] == c) return i;
  return -1;
}
struct Poly { double c[20]; };
static Poly pzero() { Poly p; memset(p.c, 0, sizeof(p.c)); return p; }
static Poly padd(const Poly& a, const Poly& b) { Poly r; for (int i = 0; i < 20; i++) r.c[i] = a.c[i] + b.c[i]; return r; }
static Poly psub(const Poly& a, const Poly& b) { Poly r; for (int i = 0; i < 20; i++) r.c[i] = a.c[i] - b.c[i]; return r; }
static Poly pscale(const Poly& a, double s) { Poly r; for (int i = 0; i < 20; i++) r.c[i] = a.c[i] * s; return r; }
static Poly pmul(const Poly& a, const Poly& b) {
  static int tab[20][20];
  static bool init = false;
  if (!init) {
    for (int i = 0; i < 20; i++)
      for (int j = 0; j < 20; j++) {
        int e0 = kMono[i][0] + kMono[j][0], e1 = kMono[i][1] + kMono[j][1], e2 = kMono[i][2] + kMono[j][2];
        tab[i][j] = (e0 + e1 + e2 <= 3) ? mono_index(e0, e1, e2) : -1;
      }
    init = true;
  }
  Poly r = pzero();
  for (int i = 0; i < 20; i++) {
    if (a.c[i] == 0) continue;
    for (int j = 0; j < 20; j++) {
      if (b.c[j] == 0 || tab[i][j] < 0) continue;
      r.c[tab[i][j]] += a.c[i] * b.c[j];
    }
  }
  return r;
}

// hal::LU64f with right-hand side (Gaussian elimination, partial pivoting); returns 0 when singular.
static int lu_solve(double* A, int m, double* b, int n) {
  const double eps = DBL_EPSILON * 100;
  int i, j, k, p = 1;
  for (i = 0; i < m; i++) {
    k = i;
    for (j = i + 1; j < m; j++)
      if (std::fabs(A[j * m + i]) > std::fabs(A[k * m + i])) k = j;
    if (std::fabs(A[k * m + i]) < eps) return 0;
    if (k != i) {
      for (j = i; j < m; j++) std::swap(A[i * m + j], A[k * m + j]);
      for (j = 0; j < n; j++) std::swap(b[i * n + j], b[k * n + j]);
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

struct Cplx { double re, im; };
static inline Cplx cmul(Cplx a, Cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
static inline Cplx cadd(Cplx a, Cplx b) { return {a.re + b.re, a.im + b.im}; }
static inline Cplx csub(Cplx a, Cplx b) { return {a.re - b.re, a.im - b.im}; }
static inline Cplx cdiv(Cplx a, Cplx b) {  // cv::Complex operator /
  double t = 1. / (b.re * b.re + b.im * b.im);
  return {(a.re * b.re + a.im * b.im) * t, (-a.re * b.im + a.im * b.re) * t};
}

// cv::solvePoly (Durand-Kerner), coefficients ascending (c[0] constant term); returns the degree used.
static int solve_poly(const double* c, int n0, Cplx* roots, int maxIters = 1000) {
  int n = n0;
  Cplx coeffs[16];
  for (int i = 0; i <= n; i++) coeffs[i] = {c[i], 0};
  for (; n > 1; n--)
    if (std::fabs(coeffs[n].re) + std::fabs(coeffs[n].im) > DBL_EPSILON) break;
  Cplx p = {1, 0}, r = {1, 1};
  for (int i = 0; i < n; i++) { roots[i] = p; p = cmul(p, r); }
  for (int iter = 0; iter < maxIters; iter++) {
    double maxDiff = 0;
    for (int i = 0; i < n; i++) {
      p = roots[i];
      Cplx num = coeffs[n], denom = coeffs[n];
      for (int j = 0; j < n; j++) {
        num = cadd(cmul(num, p), coeffs[n - j - 1]);
        if (j != i) {
          Cplx d = csub(p, roots[j]);
          if (d.re != 0 || d.im != 0) denom = cmul(denom, d);
        }
      }
      num = cdiv(num, denom);
      roots[i] = csub(p, num);
      maxDiff = std::max(maxDiff, std::sqrt(num.re * num.re + num.im * num.im));
    }
    if (maxDiff <= 0) break;
  }
  for (int i = 0; i < n; i++)
    if (std::fabs(roots[i].im) < 1e-100) roots[i].im = 0;
  for (int k = n; k < n0; k++) roots[k] = roots[k - 1];
  return n;
}

struct EssentialCb : RansacCbT<double> {
  EssentialCb() { d1 = 2; d2 = 2; model_size = 9; max_models = 10; }
  int run_kernel(const double* q1, const double* q2, int n, double* models) const override {
    std::vector<double> Q((size_t)n * 9);
    for (int i = 0; i < n; i++) {
      double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
      double* q = &Q[(size_t)i * 9];
      q[0] = x2 * x1; q[1] = x2 * y1; q[2] = x2; q[3] = y2 * x1; q[4] = y2 * y1; q[5] = y2; q[6] = x1; q[7] = y1; q[8] = 1.0;
    }
    double w[9], Vt[81];
    svd_compute(Q.data(), n, 9, w, nullptr, Vt, true);
    const double* EE[4] = {Vt + 5 * 9, Vt + 6 * 9, Vt + 7 * 9, Vt + 8 * 9};  // X, Y, Z, W
    // E(x,y,z) entries as polynomials
    Poly E[9];
    const int ix = mono_index(1, 0, 0), iy = mono_index(0, 1, 0), iz = mono_index(0, 0, 1), i1 = mono_index(0, 0, 0);
    for (int k = 0; k < 9; k++) {
      E[k] = pzero();
      E[k].c[ix] = EE[0][k]; E[k].c[iy] = EE[1][k]; E[k].c[iz] = EE[2][k]; E[k].c[i1] = EE[3][k];
    }
    Poly rows[10];
    // det(E)
    rows[0] = padd(psub(pmul(E[0], psub(pmul(E[4], E[8]), pmul(E[5], E[7]))), pmul(E[1], psub(pmul(E[3], E[8]), pmul(E[5], E[6])))),
                   pmul(E[2], psub(pmul(E[3], E[7]), pmul(E[4], E[6]))));
    // EEt, trace
    Poly EEt[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        EEt[i * 3 + j] = padd(padd(pmul(E[i * 3], E[j * 3]), pmul(E[i * 3 + 1], E[j * 3 + 1])), pmul(E[i * 3 + 2], E[j * 3 + 2]));
    Poly tr = padd(padd(EEt[0], EEt[4]), EEt[8]);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        Poly s = padd(padd(pmul(EEt[i * 3], E[j]), pmul(EEt[i * 3 + 1], E[3 + j])), pmul(EEt[i * 3 + 2], E[6 + j]));
        rows[1 + i * 3 + j] = psub(pscale(s, 2.0), pmul(tr, E[i * 3 + j]));
      }
    double A1[100], A2[100];
    for (int r = 0; r < 10; r++)
      for (int c = 0; c < 10; c++) { A1[r * 10 + c] = rows[r].c[c]; A2[r * 10 + c] = rows[r].c[10 + c]; }
    if (!lu_solve(A1, 10, A2, 10)) return 0;  // A = A[:, :10]^-1 A[:, 10:]
    const double* A = A2;
    double b[3 * 13];
    for (int i = 0; i < 3; i++) {
      const double* a1 = A + (i * 2 + 4) * 10;
      const double* a2 = A + (i * 2 + 5) * 10;
      double row1[13] = {0}, row2[13] = {0};
      for (int k = 0; k < 3; k++) { row1[1 + k] = a1[k]; row1[5 + k] = a1[3 + k]; row2[k] = a2[k]; row2[4 + k] = a2[3 + k]; }
      for (int k = 0; k < 4; k++) { row1[9 + k] = a1[6 + k]; row2[8 + k] = a2[6 + k]; }
      for (int k = 0; k < 13; k++) b[i * 13 + k] = row1[k] - row2[k];
    }
    // det B(z): B[j][0], B[j][1] degree 3 (coefficients br[0..3], br[4..7], highest first), B[j][2] degree 4 (br[8..12])
    auto polymul = [](const double* a, int na, const double* bb, int nb, double* out) {  // ascending coefficients
      for (int i = 0; i < na + nb - 1; i++) out[i] = 0;
      for (int i = 0; i < na; i++)
        for (int j = 0; j < nb; j++) out[i + j] += a[i] * bb[j];
    };
    double P[3][3][5];
    int deg[3] = {4, 4, 5};
    for (int j = 0; j < 3; j++) {
      const double* br = b + j * 13;
      for (int k = 0; k < 4; k++) { P[j][0][k] = br[3 - k]; P[j][1][k] = br[7 - k]; }
      P[j][0][4] = P[j][1][4] = 0;
      for (int k = 0; k < 5; k++) P[j][2][k] = br[12 - k];
    }
    double c[11] = {0};
    static const int perm[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};
    static const double sgn[6] = {1, 1, 1, -1, -1, -1};
    for (int q = 0; q < 6; q++) {
      double t1[9], t2[13];
      polymul(P[0][perm[q][0]], deg[perm[q][0]], P[1][perm[q][1]], deg[perm[q][1]], t1);
      polymul(t1, deg[perm[q][0]] + deg[perm[q][1]] - 1, P[2][perm[q][2]], deg[perm[q][2]], t2);
      for (int k = 0; k < 11; k++) c[k] += sgn[q] * t2[k];
    }
    Cplx roots[16];
    solve_poly(c, 10, roots);
    int count = 0;
    for (int i = 0; i < 10; i++) {
      if (std::fabs(roots[i].im) > 1e-10) continue;
      double z1 = roots[i].re, z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
      double bz[9];
      for (int j = 0; j < 3; j++) {
        const double* br = b + j * 13;
        bz[j * 3] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
        bz[j * 3 + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
        bz[j * 3 + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
      }
      double ww[3], vt[9];
      svd_compute(bz, 3, 3, ww, nullptr, vt, false);  // SVD::solveZ: last row of Vt
      const double* xy1 = vt + 6;
      if (std::fabs(xy1[2]) < 1e-10) continue;
      double xs = xy1[0] / xy1[2], ys = xy1[1] / xy1[2], zs = z1;
      double Ev[9], nrm = 0;
      for (int k = 0; k < 9; k++) {
        Ev[k] = EE[0][k] * xs + EE[1][k] * ys + EE[2][k] * zs + EE[3][k];
        nrm += Ev[k] * Ev[k];
      }
      nrm = std::sqrt(nrm);
      for (int k = 0; k < 9; k++) models[count * 9 + k] = Ev[k] / nrm;
      count++;
    }
    return count;
  }
  void compute_error(const double* x1p, const double* x2p, int n, const double* E, float* err) const override {
    for (int i = 0; i < n; i++) {
      double x1[3] = {x1p[2 * i], x1p[2 * i + 1], 1.}, x2[3] = {x2p[2 * i], x2p[2 * i + 1], 1.};
      double Ex1[3], Etx2[3];
      for (int r = 0; r < 3; r++) {
        Ex1[r] = E[r * 3] * x1[0] + E[r * 3 + 1] * x1[1] + E[r * 3 + 2] * x1[2];
        Etx2[r] = E[r] * x2[0] + E[3 + r] * x2[1] + E[6 + r] * x2[2];
      }
      double x2tEx1 = x2[0] * Ex1[0] + x2[1] * Ex1[1] + x2[2] * Ex1[2];
      double a = Ex1[0] * Ex1[0], b = Ex1[1] * Ex1[1], c = Etx2[0] * Etx2[0], d = Etx2[1] * Etx2[1];
      err[i] = (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
    }
  }
};

}  // namespace orc

using namespace orc;

extern "C" int orc_e5_kernel(const double* q1, const double* q2, int n, double* models) {
  EssentialCb cb;
  return cb.run_kernel(q1, q2, n, models);
}

extern "C" int orc_find_essential_ransac(const float* p1, const float* p2, int n, const double* K, double prob, double thr,
                                         int max_iters, unsigned char* mask, double* E, int* stats) {
  if (n < 5) return -1;
  double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  std::vector<double> x1(2 * n), x2(2 * n);
  for (int i = 0; i < n; i++) {
    x1[2 * i] = (p1[2 * i] - cx) / fx; x1[2 * i + 1] = (p1[2 * i + 1] - cy) / fy;
    x2[2 * i] = (p2[2 * i] - cx) / fx; x2[2 * i + 1] = (p2[2 * i + 1] - cy) / fy;
  }
  thr /= (fx + fy) / 2;
  EssentialCb cb;
  RansacStats st;
  bool ok;
  if (n == 5) {
    double models[90];
    int nm = cb.run_kernel(x1.data(), x2.data(), n, models);
    ok = nm > 0;
    if (ok) memcpy(E, models, 9 * sizeof(double));  // bestModel = first 3 rows is NOT what OpenCV does (it copies all); see note
    memset(mask, 1, n);
  } else {
    ok = ransac_run<double>(cb, x1.data(), x2.data(), n, 5, thr, prob, max_iters, E, mask, &st);
  }
  if (stats) { stats[0] = st.iters_run; stats[1] = st.niters_final; stats[2] = st.hyp_models; }
  if (!ok) { memset(mask, 0, n); return 0; }
  int c = 0;
  for (int i = 0; i < n; i++) c += mask[i] != 0;
  return c;
}
