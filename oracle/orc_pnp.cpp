// oracle/orc_pnp.cpp — TEST INFRASTRUCTURE ONLY (see orc_common.h).  PARITY UNPINNED.
//
// CPU restatement of cv::solvePnPRansac(obj, img, K, d, rvec, tvec, false, 100, 8.0, 0.99, inliers)
// as called by the reference at src/tracker.cpp:309, plus cv::Rodrigues (src/tracker.cpp:315).
// OpenCV 4.6 calib3d semantics per SURVEY.md A.7:
//   RANSAC (5-point samples, SOLVEPNP_EPNP kernel, reprojection error in float) -> inlier set ->
//   solvePnP(SOLVEPNP_ITERATIVE): DLT / planar-homography initialisation + CvLevMarq (<= 20 iterations).
// Files followed: calib3d/src/{solvepnp,epnp,calibration,compat_ptsetreg,undistort.dispatch}.cpp.
// Restriction: distortion coefficients must be zero (the synthetic configs; rectified input).  With
// d = 0 cvUndistortPoints is exactly (u-cx)*(1/fx) and cvProjectPoints2 is exactly x*fx+cx.
#include "orc_common.h"
#include "orc_linalg.h"
#include "orc_ransac.h"
#include "mvo_oracle.h"

namespace orc {

// ---- cv::Rodrigues -----------------------------------------------------------------------------------
// vector -> matrix, optional jacobian dR/dr (3 x 9, row i = d vec(R) / d r_i) as cvRodrigues2 lays it out.
static void rodrigues_v2m(const double r_[3], double R[9], double* J /* 27 or null */) {
  double rx = r_[0], ry = r_[1], rz = r_[2];
  double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
  if (theta < DBL_EPSILON) {
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1 : 0;
    if (J) {
      memset(J, 0, 27 * sizeof(double));
      J[5] = J[15] = J[19] = -1;
      J[7] = J[11] = J[21] = 1;
    }
    return;
  }
  double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c;
  double itheta = theta ? 1. / theta : 0.;
  rx *= itheta; ry *= itheta; rz *= itheta;
  double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
  double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
  static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
  if (J) {
    const double drrt[] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0, 0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                           0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
    const double d_r_x_[] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; i++) {
      double ri = i == 0 ? rx : i == 1 ? ry : rz;
      double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
      double a3 = (c - s * itheta) * ri, a4 = s * itheta;
      for (int k = 0; k < 9; k++)
        J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
    }
  }
}

// matrix -> vector (no jacobian needed on this path)
static void rodrigues_m2v(const double Rin[9], double r[3]) {
  for (int i = 0; i < 9; i++)
    if (!(Rin[i] > -100 && Rin[i] < 100)) { r[0] = r[1] = r[2] = 0; return; }  // checkRange
  double w[3], U[9], Vt[9], R[9];
  svd_compute(Rin, 3, 3, w, U, Vt, false);
  mat3mul(U, Vt, R);
  double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
  double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  double theta = std::acos(c);
  if (s < 1e-5) {
    double t;
    if (c > 0) rx = ry = rz = 0;
    else {
      t = (R[0] + 1) * 0.5;
      rx = std::sqrt(std::max(t, 0.));
      t = (R[4] + 1) * 0.5;
      ry = std::sqrt(std::max(t, 0.)) * (R[1] < 0 ? -1. : 1.);
      t = (R[8] + 1) * 0.5;
      rz = std::sqrt(std::max(t, 0.)) * (R[2] < 0 ? -1. : 1.);
      if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
      theta /= std::sqrt(rx * rx + ry * ry + rz * rz);
      rx *= theta; ry *= theta; rz *= theta;
    }
  } else {
    double vth = 1 / (2 * s);
    vth *= theta;
    rx *= vth; ry *= vth; rz *= vth;
  }
  r[0] = rx; r[1] = ry; r[2] = rz;
}

// ---- cvProjectPoints2: m = K * distort([R|t] M), optional d m/d r (2x3) and d m/d t (2x3) -------------
// Distortion model: the 5-coefficient plumb-bob set the reference forwards from sensor_msgs/CameraInfo::d
// (k1, k2, p1, p2, k3); OpenCV's rational / thin-prism / tilt terms (k4..k6, s1..s4, tau) are zero.
struct Cam {
  double fx, fy, cx, cy;
  double k[5] = {0, 0, 0, 0, 0};
  bool dist = false;
};

static Cam make_cam(const double* K, const double* d) {
  Cam c;
  c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
  for (int i = 0; i < 5; i++) { c.k[i] = d ? d[i] : 0.0; c.dist = c.dist || c.k[i] != 0.0; }
  return c;
}

// cvUndistortPointsInternal for one pixel, identity R, no P: 5 fixed-point iterations (TermCriteria COUNT 5)
static void undistort_point(const Cam& cam, double u, double v, double& xo, double& yo) {
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

static void project_point(const double R[9], const double dRdr[27], const double t[3], const Cam& cam, const double M[3],
                          double m[2], double* dpdr /* 6 */, double* dpdt /* 6 */) {
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
    double dxdt[] = {z, 0, -x * z}, dydt[] = {0, z, -y * z};
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
    double dx0dr[] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11],
                      X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
    double dy0dr[] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14],
                      X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
    double dz0dr[] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17],
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

// ---- EPnP (calib3d/src/epnp.cpp) -------------------------------------------------------------------
struct Epnp {
  double fu, fv, uc, vc;
  int n;
  std::vector<double> pws, us, alphas, pcs;
  double cws[4][3], ccs[4][3];

  static double dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
  static double dist2(const double* p1, const double* p2) {
    return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
  }

  void choose_control_points() {
    cws[0][0] = cws[0][1] = cws[0][2] = 0;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < 3; j++) cws[0][j] += pws[3 * i + j];
    for (int j = 0; j < 3; j++) cws[0][j] /= n;
    std::vector<double> pw0(3 * n);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < 3; j++) pw0[3 * i + j] = pws[3 * i + j] - cws[0][j];
    double pw0tpw0[9] = {0}, dc[3], uct[9];
    // cvMulTransposed(PW0, &PW0tPW0, 1): A^T A, accumulated over rows
    for (int a = 0; a < 3; a++)
      for (int b = a; b < 3; b++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += pw0[3 * i + a] * pw0[3 * i + b];
        pw0tpw0[a * 3 + b] = pw0tpw0[b * 3 + a] = s;
      }
    // cvSVD(&PW0tPW0, &DC, &UCt, 0, MODIFY_A | U_T): UCt rows = left singular vectors
    double U[9];
    svd_compute(pw0tpw0, 3, 3, dc, U, nullptr, false);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) uct[i * 3 + j] = U[j * 3 + i];
    for (int i = 1; i < 4; i++) {
      double k = std::sqrt(dc[i - 1] / n);
      for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * uct[3 * (i - 1) + j];
    }
  }

  void compute_barycentric_coordinates() {
    double cc[9], cc_inv[9];
    for (int i = 0; i < 3; i++)
      for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
    invert_svd(cc, 3, cc_inv);
    double* ci = cc_inv;
    for (int i = 0; i < n; i++) {
      double* pi = &pws[3 * i];
      double* a = &alphas[4 * i];
      for (int j = 0; j < 3; j++)
        a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) + ci[3 * j + 2] * (pi[2] - cws[0][2]);
      a[0] = 1.0f - a[1] - a[2] - a[3];
    }
  }

  void compute_ccs(const double* betas, const double* ut) {
    for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0f;
    for (int i = 0; i < 4; i++) {
      const double* v = ut + 12 * (11 - i);
      for (int j = 0; j < 4; j++)
        for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
    }
  }
  void compute_pcs() {
    for (int i = 0; i < n; i++) {
      double* a = &alphas[4 * i];
      double* pc = &pcs[3 * i];
      for (int j = 0; j < 3; j++) pc[j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
    }
  }
  void solve_for_sign() {
    if (pcs[2] < 0.0) {
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 3; j++) ccs[i][j] = -ccs[i][j];
      for (int i = 0; i < n; i++) {
        pcs[3 * i] = -pcs[3 * i]; pcs[3 * i + 1] = -pcs[3 * i + 1]; pcs[3 * i + 2] = -pcs[3 * i + 2];
      }
    }
  }
  void estimate_R_and_t(double R[3][3], double t[3]) {
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
      for (int j = 0; j < 3; j++) { pc0[j] += pcs[3 * i + j]; pw0[j] += pws[3 * i + j]; }
    for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    double abt[9] = {0}, abt_d[3], abt_u[9], abt_vt[9], abt_v[9];
    for (int i = 0; i < n; i++) {
      double* pc = &pcs[3 * i];
      double* pw = &pws[3 * i];
      for (int j = 0; j < 3; j++) {
        abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
        abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
        abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
      }
    }
    // cvSVD(&ABt, &ABt_D, &ABt_U, &ABt_V, MODIFY_A): U as is, V (not transposed)
    svd_compute(abt, 3, 3, abt_d, abt_u, abt_vt, false);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) abt_v[i * 3 + j] = abt_vt[j * 3 + i];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R[i][j] = dot(abt_u + 3 * i, abt_v + 3 * j);
    const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                       R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
    if (det < 0) { R[2][0] = -R[2][0]; R[2][1] = -R[2][1]; R[2][2] = -R[2][2]; }
    t[0] = pc0[0] - dot(R[0], pw0);
    t[1] = pc0[1] - dot(R[1], pw0);
    t[2] = pc0[2] - dot(R[2], pw0);
  }
  double reprojection_error(const double R[3][3], const double t[3]) {
    double sum2 = 0.0;
    for (int i = 0; i < n; i++) {
      double* pw = &pws[3 * i];
      double Xc = dot(R[0], pw) + t[0];
      double Yc = dot(R[1], pw) + t[1];
      double inv_Zc = 1.0 / (dot(R[2], pw) + t[2]);
      double ue = uc + fu * Xc * inv_Zc;
      double ve = vc + fv * Yc * inv_Zc;
      double u = us[2 * i], v = us[2 * i + 1];
      sum2 += std::sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / n;
  }
  double compute_R_and_t(const double* ut, const double* betas, double R[3][3], double t[3]) {
    compute_ccs(betas, ut);
    compute_pcs();
    solve_for_sign();
    estimate_R_and_t(R, t);
    return reprojection_error(R, t);
  }
  static void compute_L_6x10(const double* ut, double* l_6x10) {
    const double* v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
      int a = 0, b = 1;
      for (int j = 0; j < 6; j++) {
        dv[i][j][0] = v[i][3 * a] - v[i][3 * b];
        dv[i][j][1] = v[i][3 * a + 1] - v[i][3 * b + 1];
        dv[i][j][2] = v[i][3 * a + 2] - v[i][3 * b + 2];
        b++;
        if (b > 3) { a++; b = a + 1; }
      }
    }
    for (int i = 0; i < 6; i++) {
      double* row = l_6x10 + 10 * i;
      row[0] = dot(dv[0][i], dv[0][i]);
      row[1] = 2.0f * dot(dv[0][i], dv[1][i]);
      row[2] = dot(dv[1][i], dv[1][i]);
      row[3] = 2.0f * dot(dv[0][i], dv[2][i]);
      row[4] = 2.0f * dot(dv[1][i], dv[2][i]);
      row[5] = dot(dv[2][i], dv[2][i]);
      row[6] = 2.0f * dot(dv[0][i], dv[3][i]);
      row[7] = 2.0f * dot(dv[1][i], dv[3][i]);
      row[8] = 2.0f * dot(dv[2][i], dv[3][i]);
      row[9] = dot(dv[3][i], dv[3][i]);
    }
  }
  void compute_rho(double* rho) {
    rho[0] = dist2(cws[0], cws[1]); rho[1] = dist2(cws[0], cws[2]); rho[2] = dist2(cws[0], cws[3]);
    rho[3] = dist2(cws[1], cws[2]); rho[4] = dist2(cws[1], cws[3]); rho[5] = dist2(cws[2], cws[3]);
  }
  static void find_betas_approx_1(const double* L, const double* rho, double* betas) {
    double l[6 * 4], b4[4];
    for (int i = 0; i < 6; i++) { l[i * 4] = L[i * 10]; l[i * 4 + 1] = L[i * 10 + 1]; l[i * 4 + 2] = L[i * 10 + 3]; l[i * 4 + 3] = L[i * 10 + 6]; }
    solve_svd(l, 6, 4, rho, b4);
    if (b4[0] < 0) {
      betas[0] = std::sqrt(-b4[0]); betas[1] = -b4[1] / betas[0]; betas[2] = -b4[2] / betas[0]; betas[3] = -b4[3] / betas[0];
    } else {
      betas[0] = std::sqrt(b4[0]); betas[1] = b4[1] / betas[0]; betas[2] = b4[2] / betas[0]; betas[3] = b4[3] / betas[0];
    }
  }
  static void find_betas_approx_2(const double* L, const double* rho, double* betas) {
    double l[6 * 3], b3[3];
    for (int i = 0; i < 6; i++) { l[i * 3] = L[i * 10]; l[i * 3 + 1] = L[i * 10 + 1]; l[i * 3 + 2] = L[i * 10 + 2]; }
    solve_svd(l, 6, 3, rho, b3);
    if (b3[0] < 0) { betas[0] = std::sqrt(-b3[0]); betas[1] = (b3[2] < 0) ? std::sqrt(-b3[2]) : 0.0; }
    else { betas[0] = std::sqrt(b3[0]); betas[1] = (b3[2] > 0) ? std::sqrt(b3[2]) : 0.0; }
    if (b3[1] < 0) betas[0] = -betas[0];
    betas[2] = 0.0; betas[3] = 0.0;
  }
  static void find_betas_approx_3(const double* L, const double* rho, double* betas) {
    double l[6 * 5], b5[5];
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 5; j++) l[i * 5 + j] = L[i * 10 + j];
    solve_svd(l, 6, 5, rho, b5);
    if (b5[0] < 0) { betas[0] = std::sqrt(-b5[0]); betas[1] = (b5[2] < 0) ? std::sqrt(-b5[2]) : 0.0; }
    else { betas[0] = std::sqrt(b5[0]); betas[1] = (b5[2] > 0) ? std::sqrt(b5[2]) : 0.0; }
    if (b5[1] < 0) betas[0] = -betas[0];
    betas[2] = b5[3] / betas[0];
    betas[3] = 0.0;
  }
  static void compute_A_and_b_gauss_newton(const double* l_6x10, const double* rho, const double betas[4], double* A, double* b) {
    for (int i = 0; i < 6; i++) {
      const double* rowL = l_6x10 + i * 10;
      double* rowA = A + i * 4;
      rowA[0] = 2 * rowL[0] * betas[0] + rowL[1] * betas[1] + rowL[3] * betas[2] + rowL[6] * betas[3];
      rowA[1] = rowL[1] * betas[0] + 2 * rowL[2] * betas[1] + rowL[4] * betas[2] + rowL[7] * betas[3];
      rowA[2] = rowL[3] * betas[0] + rowL[4] * betas[1] + 2 * rowL[5] * betas[2] + rowL[8] * betas[3];
      rowA[3] = rowL[6] * betas[0] + rowL[7] * betas[1] + rowL[8] * betas[2] + 2 * rowL[9] * betas[3];
      b[i] = rho[i] - (rowL[0] * betas[0] * betas[0] + rowL[1] * betas[0] * betas[1] + rowL[2] * betas[1] * betas[1] +
                       rowL[3] * betas[0] * betas[2] + rowL[4] * betas[1] * betas[2] + rowL[5] * betas[2] * betas[2] +
                       rowL[6] * betas[0] * betas[3] + rowL[7] * betas[1] * betas[3] + rowL[8] * betas[2] * betas[3] +
                       rowL[9] * betas[3] * betas[3]);
    }
  }
  // Householder QR exactly as epnp.cpp::qr_solve (including its pivot-scan quirk); x is left untouched on a singular column.
  static void qr_solve(double* pA, double* pb, double* pX, int nr, int nc) {
    double A1[6], A2[6];
    double* ppAkk = pA;
    for (int k = 0; k < nc; k++) {
      double *ppAik1 = ppAkk, eta = std::fabs(*ppAik1);
      for (int i = k + 1; i < nr; i++) {
        double elt = std::fabs(*ppAik1);
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
      double sigma = std::sqrt(sum2);
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
  static void gauss_newton(const double* L, const double* rho, double betas[4]) {
    double a[24], b[6], x[4] = {0, 0, 0, 0};
    for (int k = 0; k < 5; k++) {
      compute_A_and_b_gauss_newton(L, rho, betas, a, b);
      qr_solve(a, b, x, 6, 4);
      for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
  }

  void compute_pose(double R[9], double t[3]) {
    choose_control_points();
    compute_barycentric_coordinates();
    std::vector<double> M((size_t)2 * n * 12);
    for (int i = 0; i < n; i++) {
      const double* as = &alphas[4 * i];
      double u = us[2 * i], v = us[2 * i + 1];
      double* M1 = &M[(size_t)(2 * i) * 12];
      double* M2 = M1 + 12;
      for (int k = 0; k < 4; k++) {
        M1[3 * k] = as[k] * fu; M1[3 * k + 1] = 0.0; M1[3 * k + 2] = as[k] * (uc - u);
        M2[3 * k] = 0.0; M2[3 * k + 1] = as[k] * fv; M2[3 * k + 2] = as[k] * (vc - v);
      }
    }
    double mtm[144], d[12], ut[144], U[144];
    for (int a = 0; a < 12; a++)
      for (int b = a; b < 12; b++) {
        double s = 0;
        for (int i = 0; i < 2 * n; i++) s += M[(size_t)i * 12 + a] * M[(size_t)i * 12 + b];
        mtm[a * 12 + b] = mtm[b * 12 + a] = s;
      }
    svd_compute(mtm, 12, 12, d, U, nullptr, false);
    for (int i = 0; i < 12; i++)
      for (int j = 0; j < 12; j++) ut[i * 12 + j] = U[j * 12 + i];
    double l_6x10[60], rho[6];
    compute_L_6x10(ut, l_6x10);
    compute_rho(rho);
    double Betas[4][4] = {{0}}, rep_errors[4] = {0};
    double Rs[4][3][3], ts[4][3];
    find_betas_approx_1(l_6x10, rho, Betas[1]);
    gauss_newton(l_6x10, rho, Betas[1]);
    rep_errors[1] = compute_R_and_t(ut, Betas[1], Rs[1], ts[1]);
    find_betas_approx_2(l_6x10, rho, Betas[2]);
    gauss_newton(l_6x10, rho, Betas[2]);
    rep_errors[2] = compute_R_and_t(ut, Betas[2], Rs[2], ts[2]);
    find_betas_approx_3(l_6x10, rho, Betas[3]);
    gauss_newton(l_6x10, rho, Betas[3]);
    rep_errors[3] = compute_R_and_t(ut, Betas[3], Rs[3], ts[3]);
    int N = 1;
    if (rep_errors[2] < rep_errors[1]) N = 2;
    if (rep_errors[3] < rep_errors[N]) N = 3;
    for (int i = 0; i < 3; i++) {
      t[i] = ts[N][i];
      for (int j = 0; j < 3; j++) R[i * 3 + j] = Rs[N][i][j];
    }
  }
};

// solvePnP(SOLVEPNP_EPNP) on float sample points: undistortPoints -> float, epnp, Rodrigues.
static void solve_pnp_epnp(const float* obj, const float* img, int count, const Cam& cam, double rvec[3], double tvec[3]) {
  Epnp e;
  e.fu = cam.fx; e.fv = cam.fy; e.uc = cam.cx; e.vc = cam.cy;
  e.n = count;
  e.pws.resize(3 * count); e.us.resize(2 * count); e.alphas.resize(4 * count); e.pcs.resize(3 * count);
  for (int i = 0; i < count; i++) {
    e.pws[3 * i] = obj[3 * i]; e.pws[3 * i + 1] = obj[3 * i + 1]; e.pws[3 * i + 2] = obj[3 * i + 2];
    // cvUndistortPoints writes CV_32FC2; epnp::init_points re-applies fu, uc in double
    double xu, yu;
    undistort_point(cam, (double)img[2 * i], (double)img[2 * i + 1], xu, yu);
    float xn = (float)xu, yn = (float)yu;
    e.us[2 * i] = xn * e.fu + e.uc;
    e.us[2 * i + 1] = yn * e.fv + e.vc;
  }
  double R[9];
  e.compute_pose(R, tvec);
  rodrigues_m2v(R, rvec);
}

struct PnPCb : RansacCb {
  Cam cam;
  PnPCb() { d1 = 3; d2 = 2; model_size = 6; }
  int run_kernel(const float* m1, const float* m2, int count, double* model) const override {
    double rvec[3], tvec[3];
    solve_pnp_epnp(m1, m2, count, cam, rvec, tvec);
    // hconcat(rvec, tvec) -> 3x2 row-major: [r0 t0; r1 t1; r2 t2]
    for (int i = 0; i < 3; i++) { model[2 * i] = rvec[i]; model[2 * i + 1] = tvec[i]; }
    return 1;
  }
  void compute_error(const float* m1, const float* m2, int count, const double* model, float* err) const override {
    double rvec[3] = {model[0], model[2], model[4]}, tvec[3] = {model[1], model[3], model[5]};
    double R[9];
    rodrigues_v2m(rvec, R, nullptr);
    for (int i = 0; i < count; i++) {
      double M[3] = {m1[3 * i], m1[3 * i + 1], m1[3 * i + 2]}, m[2];
      project_point(R, nullptr, tvec, cam, M, m, nullptr, nullptr);
      float px = (float)m[0], py = (float)m[1];  // projpoints is CV_32F
      float dx = m2[2 * i] - px, dy = m2[2 * i + 1] - py;
      float s = 0;
      s += dx * dx;
      s += dy * dy;
      err[i] = s;
    }
  }
};

// CvLevMarq (calib3d/src/compat_ptsetreg.cpp) as driven by cvFindExtrinsicCameraParams2: 6 params, J + err mode.
static void lm_refine(const std::vector<double>& M, const std::vector<double>& m, int count, const Cam& cam, double param[6]) {
  const int max_iter = 20;
  const double epsilon = FLT_EPSILON;
  enum { DONE = 0, STARTED = 1, CALC_J = 2, CHECK_ERR = 3 };
  int state = STARTED, iters = 0, lambdaLg10 = -3;
  double prevParam[6], JtJ[36], JtErr[6], prevErrNorm = DBL_MAX, errNorm = 0;
  std::vector<double> J((size_t)2 * count * 6), err((size_t)2 * count);
  auto step = [&]() {
    const double LOG10 = std::log(10.);
    double lambda = std::exp(lambdaLg10 * LOG10);
    double A[36], x[6];
    memcpy(A, JtJ, sizeof(A));
    for (int i = 0; i < 6; i++) A[i * 6 + i] *= 1. + lambda;
    solve_svd(A, 6, 6, JtErr, x);
    for (int i = 0; i < 6; i++) param[i] = prevParam[i] - x[i];
  };
  auto eval = [&](bool withJ) {
    double R[9], dRdr[27];
    rodrigues_v2m(param, R, withJ ? dRdr : nullptr);
    for (int i = 0; i < count; i++) {
      double mm[2], dpdr[6], dpdt[6];
      project_point(R, withJ ? dRdr : nullptr, param + 3, cam, &M[3 * i], mm, withJ ? dpdr : nullptr, withJ ? dpdt : nullptr);
      err[2 * i] = mm[0] - m[2 * i];
      err[2 * i + 1] = mm[1] - m[2 * i + 1];
      if (withJ)
        for (int r = 0; r < 2; r++)
          for (int j = 0; j < 3; j++) {
            J[(size_t)(2 * i + r) * 6 + j] = dpdr[r * 3 + j];
            J[(size_t)(2 * i + r) * 6 + 3 + j] = dpdt[r * 3 + j];
          }
    }
  };
  auto norm2 = [&](const std::vector<double>& v) { double s = 0; for (double x : v) s += x * x; return std::sqrt(s); };
  for (;;) {
    bool wantJ = false, wantErr = false, proceed;
    // CvLevMarq::update
    if (state == DONE) { proceed = false; }
    else if (state == STARTED) { wantJ = wantErr = true; state = CALC_J; proceed = true; }
    else if (state == CALC_J) {
      for (int a = 0; a < 6; a++)
        for (int b = a; b < 6; b++) {
          double s = 0;
          for (int i = 0; i < 2 * count; i++) s += J[(size_t)i * 6 + a] * J[(size_t)i * 6 + b];
          JtJ[a * 6 + b] = JtJ[b * 6 + a] = s;
        }
      for (int a = 0; a < 6; a++) {
        double s = 0;
        for (int i = 0; i < 2 * count; i++) s += J[(size_t)i * 6 + a] * err[i];
        JtErr[a] = s;
      }
      memcpy(prevParam, param, sizeof(prevParam));
      step();
      if (iters == 0) prevErrNorm = norm2(err);
      wantErr = true;
      state = CHECK_ERR;
      proceed = true;
    } else {  // CHECK_ERR
      errNorm = norm2(err);
      bool handled = false;
      if (errNorm > prevErrNorm) {
        if (++lambdaLg10 <= 16) {
          step();
          wantErr = true;
          state = CHECK_ERR;
          proceed = true;
          handled = true;
        }
      }
      if (!handled) {
        lambdaLg10 = std::max(lambdaLg10 - 1, -16);
        double dn = 0, pn = 0;
        for (int i = 0; i < 6; i++) { dn += (param[i] - prevParam[i]) * (param[i] - prevParam[i]); pn += prevParam[i] * prevParam[i]; }
        // cvNorm(param, prevParam, CV_RELATIVE_L2) = ||param - prevParam|| / ||prevParam||
        double rel = std::sqrt(dn) / (std::sqrt(pn) + DBL_EPSILON);
        if (++iters >= max_iter || rel < epsilon) { state = DONE; proceed = true; }
        else { prevErrNorm = errNorm; wantJ = wantErr = true; state = CALC_J; proceed = true; }
      }
    }
    if (!proceed || !wantErr) break;
    eval(wantJ);
  }
}

// cvFindExtrinsicCameraParams2 (useExtrinsicGuess = false).  Returns 0 on the "DLT needs >= 6 points" exception.
static int solve_pnp_iterative(const std::vector<double>& M, const std::vector<double>& m, int count, const Cam& cam,
                               double rvec[3], double tvec[3]);

}  // namespace orc

#include "orc_pnp_iterative.inc"
#include "orc_p3p.inc"

using namespace orc;

extern "C" int orc_rodrigues_v2m(const double* r, double* R) { rodrigues_v2m(r, R, nullptr); return 0; }
extern "C" int orc_rodrigues_m2v(const double* R, double* r) { rodrigues_m2v(R, r); return 0; }

extern "C" int orc_epnp(const float* obj, const float* img, int n, const double* K, double* rvec, double* tvec) {
  Cam cam = make_cam(K, nullptr);
  solve_pnp_epnp(obj, img, n, cam, rvec, tvec);
  return 0;
}

// solvePnP(SOLVEPNP_P3P) on four correspondences; returns the number of P3P solutions, the best one in rvec / tvec
extern "C" int orc_solve_p3p4(const float* obj, const float* img, const double* K, const double* d, double* rvec, double* tvec) {
  Cam cam = make_cam(K, d);
  return solve_pnp_p3p4(obj, img, cam, rvec, tvec);
}

// real roots of a x^4 + b x^3 + c x^2 + d x + e (polynom_solver.cpp solve_deg4); returns their number
extern "C" int orc_solve_deg4(double a, double b, double c, double d, double e, double* roots) {
  return p3_solve_deg4(a, b, c, d, e, roots[0], roots[1], roots[2], roots[3]);
}

extern "C" int orc_solve_pnp_ransac(const float* obj, const float* img, int n, const double* K, const double* d, int iters,
                                    float reproj_err, double confidence, double* rvec, double* tvec, int* inlier_idx,
                                    int* n_inliers, int* stats) {
  if (n < 4) return -1;
  *n_inliers = 0;
  Cam cam = make_cam(K, d);
  if (n == 4) {  // model_points == npoints == 4: solvePnP(SOLVEPNP_P3P) on all four, no RANSAC, no refinement
    if (solve_pnp_p3p4(obj, img, cam, rvec, tvec) == 0) return 0;
    for (int i = 0; i < n; i++) inlier_idx[i] = i;
    *n_inliers = n;
    return 1;
  }
  PnPCb cb;
  cb.cam = cam;
  if (n == 5) {
    double model[6];
    cb.run_kernel(obj, img, n, model);
    for (int i = 0; i < 3; i++) { rvec[i] = model[2 * i]; tvec[i] = model[2 * i + 1]; }
    for (int i = 0; i < n; i++) inlier_idx[i] = i;
    *n_inliers = n;
    return 1;
  }
  std::vector<unsigned char> mask(n);
  double model[6];
  RansacStats st;
  bool ok = ransac_run(cb, obj, img, n, 5, (double)reproj_err, confidence, iters, model, mask.data(), &st);
  if (stats) { stats[0] = st.iters_run; stats[1] = st.niters_final; stats[2] = st.hyp_models; }
  if (!ok) return 0;
  std::vector<double> Mi, mi;
  int cnt = 0;
  for (int i = 0; i < n; i++)
    if (mask[i]) {
      Mi.push_back(obj[3 * i]); Mi.push_back(obj[3 * i + 1]); Mi.push_back(obj[3 * i + 2]);
      mi.push_back(img[2 * i]); mi.push_back(img[2 * i + 1]);
      cnt++;
    }
  double r[3], t[3];
  int res = solve_pnp_iterative(Mi, mi, cnt, cam, r, t);
  if (res == 0) {  // DLT exception with exactly 5 points: keep the minimal-sample model
    if (cnt == 5) { for (int i = 0; i < 3; i++) { r[i] = model[2 * i]; t[i] = model[2 * i + 1]; } }
    else return -4;
  }
  for (int i = 0; i < 3; i++) { rvec[i] = r[i]; tvec[i] = t[i]; }
  int k = 0;
  for (int i = 0; i < n; i++)
    if (mask[i]) inlier_idx[k++] = i;
  *n_inliers = k;
  return 1;
}
