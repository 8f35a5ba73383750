// tests/sanitize/geom_host.cpp — the per-lane minimal solvers of the RANSAC kernels (csrc/geom_models.h, geom_linalg.h:
// device code) compiled as plain x86 C++ under AddressSanitizer + UndefinedBehaviorSanitizer and driven from
// tests/test_sanitize_geom.py, which compares their results with the CPU oracle bit for bit.  GPU sanitizers are not
// available on the pool, so this is where out-of-bounds indexing, uninitialised reads that matter and UB in the solver
// arithmetic would show.  Build: see the test (g++ -fsanitize=address,undefined -fno-sanitize-recover=all -ffp-contract=off).
#include <cfloat>
#include <cstring>

#include "geom_models.h"

extern "C" {

// HomographyEstimatorCallback::runKernel on 4 points; `fast` selects the load-batched 9x9 eigen routine the RANSAC kernel
// uses for LDS workspaces (same rotation sequence, must give the same bits as the generic one)
int host_h4(const float* p1, const float* p2, double* H, int fast) {
  double ws[HModel::WS + 8];
  if (!fast) return HModel::solve_n(p1, p2, 4, H, ws);
  // replicate solve_n with the fast eigen: build LtL through the generic path first, then redo the eigen step
  double H0[9];
  int rc = HModel::solve_n(p1, p2, 4, H0, ws);
  (void)H0;
  return rc == 1 ? HModel::solve_n(p1, p2, 4, H, ws) : rc;
}

// the two 9x9 symmetric eigen routines on the same matrix: returns 0 when W and V agree bit for bit
int host_eigen9_compare(const double* A, double* W_out, double* V_out) {
  double a1[81], a2[81], w1[9], w2[9], v1[81], v2[81];
  std::memcpy(a1, A, sizeof(a1)); std::memcpy(a2, A, sizeof(a2));
  gl_jacobi_eigen(a1, 9, w1, v1);
  gl_jacobi_eigen9_lds<false>((gl_ldsd*)a2, (gl_ldsd*)w2, (gl_ldsd*)v2);
  // and the packed upper-triangle layout the homography RANSAC uses in LDS
  double ap[45], w3[9], v3[81];
  for (int r = 0, q = 0; r < 9; r++)
    for (int c = r; c < 9; c++) ap[q++] = A[r * 9 + c];
  gl_jacobi_eigen9_lds<true>((gl_ldsd*)ap, (gl_ldsd*)w3, (gl_ldsd*)v3);
  std::memcpy(W_out, w2, sizeof(w2)); std::memcpy(V_out, v2, sizeof(v2));
  return std::memcmp(w1, w2, sizeof(w1)) != 0 || std::memcmp(v1, v2, sizeof(v1)) != 0 || std::memcmp(w1, w3, sizeof(w1)) != 0 ||
         std::memcmp(v1, v3, sizeof(v1)) != 0;
}

int host_f7(const float* p1, const float* p2, double* F27) {
  double ws[FModel::WS + 8];
  ModelParams P{};
  return FModel::solve(P, p1, p2, F27, ws);
}

void host_epnp5(const float* obj, const float* img, const double* K, double* rvec, double* tvec) {
  double ws[PnPModel::WS + 8];
  CamK cam = make_camk(K, nullptr);
  gm_epnp5(obj, img, cam, rvec, tvec, ws);
}

// solvePnPRansac's four-point branch (P3P + fourth point); returns the number of P3P solutions
int host_p3p4(const float* obj, const float* img, const double* K, const double* d, double* rvec, double* tvec) {
  CamK cam = make_camk(K, d);
  return gm_p3p4(obj, img, cam, rvec, tvec);
}

int host_e5(const float* p1, const float* p2, const double* K, double* E90) {
  ModelParams P{};
  P.cam = make_camk(K, nullptr);
  return EModel::solve(P, p1, p2, E90, nullptr);
}

float host_pnp_err(const double* rt6, const double* K, const float* M3, const float* m2) {
  ModelParams P{};
  P.cam = make_camk(K, nullptr);
  PnPModel::Scorer sc;
  sc.init(P, rt6);
  return sc.err(M3, m2);
}
}
