// csrc/geom_linalg.h — per-lane dense linear algebra for the RANSAC solvers (device code).
//
// Every lane of a wavefront solves its own minimal problem, so these are plain sequential routines on
// small row-major double matrices held in per-lane (private) arrays.  They follow the OpenCV numerics the
// reference's calib3d calls run on (core/src/lapack.cpp JacobiImpl_ / JacobiSVDImpl_ / SVBkSbImpl_ / LUImpl,
// mathfuncs.cpp solveCubic) operation by operation, with -ffp-contract=off, so that a lane reproduces the
// CPU sequence bit for bit wherever no libm call is involved (hypot is the explicit scaled form).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>

#define GL_MAXN 12

// The dense kernels below are deliberately NOT inlined: a minimal solver calls the Jacobi SVD a dozen
// times, and inlining every copy produced single functions > 128 KB whose long-branch expansion
// (s_setpc through a scratch SGPR pair) corrupted control flow on gfx950 / ROCm 7.2.
#define GL_NOINLINE __attribute__((noinline))


__device__ __forceinline__ void gl_swap(double& a, double& b) { double t = a; a = b; b = t; }

__device__ __forceinline__ double gl_hypot(double a, double b) {
  a = fabs(a); b = fabs(b);
  if (a < b) gl_swap(a, b);
  if (a == 0.0) return 0.0;
  double t = b / a;
  return a * sqrt(1.0 + t * t);
}

// cv::RNG (64-bit multiply-with-carry)
struct GlRng {
  unsigned long long state;
  __device__ explicit GlRng(unsigned long long s) : state(s ? s : 0xffffffffULL) {}
  __device__ __forceinline__ unsigned next() {
    state = (unsigned long long)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
    return (unsigned)state;
  }
  __device__ __forceinline__ int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// cv::eigen (symmetric) — JacobiImpl_.  A n x n destroyed; W descending; V eigenvectors in rows.
__device__ GL_NOINLINE void gl_jacobi_eigen(double* A, int n, double* W, double* V) {
  const double eps = DBL_EPSILON;
  int i, j, k, m;
  int indR[GL_MAXN], indC[GL_MAXN];
  for (i = 0; i < n; i++) {
    for (j = 0; j < n; j++) V[i * n + j] = 0;
    V[i * n + i] = 1;
  }
  int iters, maxIters = n * n * 30;
  double mv = 0;
  for (k = 0; k < n; k++) {
    W[k] = A[(n + 1) * k];
    if (k < n - 1) {
      for (m = k + 1, mv = fabs(A[n * k + m]), i = k + 2; i < n; i++) {
        double val = fabs(A[n * k + i]);
        if (mv < val) mv = val, m = i;
      }
      indR[k] = m;
    }
    if (k > 0) {
      for (m = 0, mv = fabs(A[k]), i = 1; i < k; i++) {
        double val = fabs(A[n * i + k]);
        if (mv < val) mv = val, m = i;
      }
      indC[k] = m;
    }
  }
  if (n > 1)
    for (iters = 0; iters < maxIters; iters++) {
      for (k = 0, mv = fabs(A[indR[0]]), i = 1; i < n - 1; i++) {
        double val = fabs(A[n * i + indR[i]]);
        if (mv < val) mv = val, k = i;
      }
      int l = indR[k];
      for (i = 1; i < n; i++) {
        double val = fabs(A[n * indC[i] + i]);
        if (mv < val) mv = val, k = indC[i], l = i;
      }
      double p = A[n * k + l];
      if (fabs(p) <= eps) break;
      double y = (W[l] - W[k]) * 0.5;
      double t = fabs(y) + gl_hypot(p, y);
      double s = gl_hypot(p, t);
      double c = t / s;
      s = p / s;
      t = (p / t) * p;
      if (y < 0) s = -s, t = -t;
      A[n * k + l] = 0;
      W[k] -= t;
      W[l] += t;
      double a0, b0;
#define GL_ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
      for (i = 0; i < k; i++) GL_ROT(A[n * i + k], A[n * i + l]);
      for (i = k + 1; i < l; i++) GL_ROT(A[n * k + i], A[n * i + l]);
      for (i = l + 1; i < n; i++) GL_ROT(A[n * k + i], A[n * l + i]);
      for (i = 0; i < n; i++) GL_ROT(V[n * k + i], V[n * l + i]);
#undef GL_ROT
      for (j = 0; j < 2; j++) {
        int idx = j == 0 ? k : l;
        if (idx < n - 1) {
          for (m = idx + 1, mv = fabs(A[n * idx + m]), i = idx + 2; i < n; i++) {
            double val = fabs(A[n * idx + i]);
            if (mv < val) mv = val, m = i;
          }
          indR[idx] = m;
        }
        if (idx > 0) {
          for (m = 0, mv = fabs(A[idx]), i = 1; i < idx; i++) {
            double val = fabs(A[n * i + idx]);
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
      gl_swap(W[m], W[k]);
      for (i = 0; i < n; i++) gl_swap(V[n * m + i], V[n * k + i]);
    }
  }
}

// cv::eigen of a 9 x 9 symmetric matrix held in LDS (the homography minimal solver's LtL) - the same JacobiImpl_ rotation
// sequence as gl_jacobi_eigen, value for value, restructured for one lane's instruction-level parallelism: a rotation
// touches 7 element pairs of A and 9 of V that are independent of each other, so all 32 operands are requested before the
// first is used (one LDS latency per rotation instead of sixteen dependent ones), the pivot search reads its 16
// candidates the same way, and the indR / indC entries of the two rotated rows come from the freshly rotated values in
// registers - they ARE rows / columns k and l - instead of being re-read.  Loops run over the static range 0..8 with
// predicates, so the per-lane index arrays stay in registers (no scratch).  A: upper triangle used, destroyed.
// PACKED: A holds only the upper triangle, row after row (45 doubles: entry (r, c), r <= c, at r (17 - r) / 2 + c) - the
// routine never touches the lower one - which lets the homography RANSAC keep 96 instead of 64 workspaces in LDS.
typedef __attribute__((address_space(3))) double gl_ldsd;
template <bool PACKED>
__device__ __forceinline__ int gl_tri9(int r) { return PACKED ? (r * (17 - r)) >> 1 : 9 * r; }   // offset of row r; entry (r, c) at gl_tri9(r) + c
template <bool PACKED>
__device__ GL_NOINLINE void gl_jacobi_eigen9_lds(gl_ldsd* A, gl_ldsd* W, gl_ldsd* V) {
  constexpr int n = 9;
  const double eps = DBL_EPSILON;
  int indR[n], indC[n];
#pragma unroll
  for (int i = 0; i < n * n; i++) V[i] = 0;
#pragma unroll
  for (int i = 0; i < n; i++) V[i * n + i] = 1;
#pragma unroll
  for (int k = 0; k < n; k++) {
    W[k] = A[gl_tri9<PACKED>(k) + k];
    indR[k] = 0; indC[k] = 0;
    if (k < n - 1) {
      int m = k + 1;
      double mv = fabs(A[gl_tri9<PACKED>(k) + m]);
#pragma unroll
      for (int i = k + 2; i < n; i++) { const double val = fabs(A[gl_tri9<PACKED>(k) + i]); if (mv < val) mv = val, m = i; }
      indR[k] = m;
    }
    if (k > 0) {
      int m = 0;
      double mv = fabs(A[k]);
#pragma unroll
      for (int i = 1; i < k; i++) { const double val = fabs(A[gl_tri9<PACKED>(i) + k]); if (mv < val) mv = val, m = i; }
      indC[k] = m;
    }
  }
  const int maxIters = n * n * 30;
  int iters = 0;
#pragma unroll 1
  for (; iters < maxIters; iters++) {
    // ---- pivot: largest |A[i][indR[i]]|, then |A[indC[i]][i]|, first maximum wins (strict <).  Everything below is
    // written with selects: a branch per predicate made the loop body ~1500 instructions long ------------------------
    double pr[n - 1], pc[n - 1];
#pragma unroll
    for (int i = 0; i < n - 1; i++) pr[i] = A[gl_tri9<PACKED>(i) + indR[i]];
#pragma unroll
    for (int i = 1; i < n; i++) pc[i - 1] = A[gl_tri9<PACKED>(indC[i]) + i];
    int k = 0, l = indR[0];
    double p = pr[0], mv = fabs(pr[0]);
#pragma unroll
    for (int i = 1; i < n - 1; i++) {
      const double val = fabs(pr[i]);
      const bool tk = mv < val;
      mv = tk ? val : mv; k = tk ? i : k; l = tk ? indR[i] : l; p = tk ? pr[i] : p;
    }
#pragma unroll
    for (int i = 1; i < n; i++) {
      const double val = fabs(pc[i - 1]);
      const bool tk = mv < val;
      mv = tk ? val : mv; k = tk ? indC[i] : k; l = tk ? i : l; p = tk ? pc[i - 1] : p;
    }
    if (fabs(p) <= eps) break;
    const double Wk = W[k], Wl = W[l];
    // ---- the 16 independent element pairs: (A[up(i,k)], A[up(i,l)]) and (V[k][i], V[l][i]); loads first ---------------
    int a0i[n], a1i[n];
    double a0[n], a1[n], v0[n], v1[n];
    const int tk = gl_tri9<PACKED>(k), tl = gl_tri9<PACKED>(l);
#pragma unroll
    for (int i = 0; i < n; i++) {
      a0i[i] = i < k ? gl_tri9<PACKED>(i) + k : tk + i;
      a1i[i] = i < l ? gl_tri9<PACKED>(i) + l : tl + i;
      a0[i] = A[a0i[i]]; a1[i] = A[a1i[i]];
    }
#pragma unroll
    for (int i = 0; i < n; i++) { v0[i] = V[n * k + i]; v1[i] = V[n * l + i]; }
    const double y = (Wl - Wk) * 0.5;
    double t = fabs(y) + gl_hypot(p, y);
    double s = gl_hypot(p, t);
    const double c = t / s;
    s = p / s;
    t = (p / t) * p;
    const bool neg = y < 0;
    s = neg ? -s : s; t = neg ? -t : t;
    W[k] = Wk - t;
    W[l] = Wl + t;
    // i == k addresses (A[k][k], A[k][l]) and i == l addresses (A[k][l], A[l][l]): the diagonal entries are dead (W holds
    // the diagonal) and keep their value, A[k][l] becomes 0 - so every store is unconditional
#pragma unroll
    for (int i = 0; i < n; i++) {
      const double x0 = a0[i], x1 = a1[i];
      const double r0 = x0 * c - x1 * s, r1 = x0 * s + x1 * c;
      a0[i] = i == k ? x0 : (i == l ? 0.0 : r0);
      a1[i] = i == k ? 0.0 : (i == l ? x1 : r1);
      A[a0i[i]] = a0[i]; A[a1i[i]] = a1[i];
    }
#pragma unroll
    for (int i = 0; i < n; i++) {
      const double x0 = v0[i], x1 = v1[i];
      V[n * k + i] = x0 * c - x1 * s; V[n * l + i] = x0 * s + x1 * c;
    }
    // ---- indR / indC of rows k and l from the values just stored: row k = a0[i > k] (0 at i = l), column k = a0[i < k],
    // row l = a1[i > l], column l = a1[i < l] (0 at i = k).  "first element initialises, then strict <" as the scans do
    {
      int mRk = 0, mCk = 0, mRl = 0, mCl = 0;
      double vRk = 0, vCk = 0, vRl = 0, vCl = 0;
      bool hRk = false, hCk = false, hRl = false, hCl = false;
#pragma unroll
      for (int i = 0; i < n; i++) {
        const double f0 = fabs(a0[i]), f1 = fabs(a1[i]);
        const bool gk = i > k, lk = i < k, gl = i > l, ll = i < l;
        const bool tRk = gk & (!hRk | (vRk < f0)), tCk = lk & (!hCk | (vCk < f0));
        const bool tRl = gl & (!hRl | (vRl < f1)), tCl = ll & (!hCl | (vCl < f1));
        vRk = tRk ? f0 : vRk; mRk = tRk ? i : mRk; hRk |= gk;
        vCk = tCk ? f0 : vCk; mCk = tCk ? i : mCk; hCk |= lk;
        vRl = tRl ? f1 : vRl; mRl = tRl ? i : mRl; hRl |= gl;
        vCl = tCl ? f1 : vCl; mCl = tCl ? i : mCl; hCl |= ll;
      }
#pragma unroll
      for (int i = 0; i < n; i++) {
        indR[i] = (i == k && k < n - 1) ? mRk : ((i == l && l < n - 1) ? mRl : indR[i]);
        indC[i] = (i == k && k > 0) ? mCk : (i == l ? mCl : indC[i]);
      }
    }
  }
#ifdef RS_TIMING
  if (threadIdx.x == 0 && blockIdx.x == 0) printf("eigen9 rotations %d\n", iters);
#endif
  // sort eigenvalues descending, rows of V follow (selection sort as in JacobiImpl_)
#pragma unroll 1
  for (int k = 0; k < n - 1; k++) {
    int m = k;
    for (int i = k + 1; i < n; i++)
      if (W[m] < W[i]) m = i;
    if (k != m) {
      const double wk = W[k]; W[k] = W[m]; W[m] = wk;
      for (int i = 0; i < n; i++) { const double t = V[n * m + i]; V[n * m + i] = V[n * k + i]; V[n * k + i] = t; }
    }
  }
}

// Second half of JacobiSVDImpl_: singular values, descending sort (rows of At / Vt follow), unit left vectors, and the
// seeded Gram-Schmidt completion for vanishing singular values.  W: squared row norms are NOT needed, it is recomputed.
// want_u with Vt == nullptr: the caller only reads the left vectors (EPnP's cvSVD(MtM, D, Ut, 0)).  The reference
// still carries V along there, but no value of At or W ever depends on V, so leaving it out changes no result.
__device__ GL_NOINLINE void gl_jacobi_svd_tail(double* At, int astep, double* W, double* Wout, double* Vt, int m, int n, int n1, bool want_u) {
  const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
  int i, j, k, iter;
  double s, sd;
  for (i = 0; i < n; i++) {
    for (k = 0, sd = 0; k < m; k++) {
      double t = At[i * astep + k];
      sd += t * t;
    }
    W[i] = sqrt(sd);
  }
  for (i = 0; i < n - 1; i++) {
    j = i;
    for (k = i + 1; k < n; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      gl_swap(W[i], W[j]);
      if (want_u)
        for (k = 0; k < m; k++) gl_swap(At[i * astep + k], At[j * astep + k]);
      if (Vt)
        for (k = 0; k < n; k++) gl_swap(Vt[i * n + k], Vt[j * n + k]);
    }
  }
  for (i = 0; i < n; i++) Wout[i] = W[i];
  if (!want_u) return;
  GlRng rng(0x12345678);
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
            asum += fabs(t);
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
      sd = sqrt(sd);
    }
    s = sd > minval ? 1 / sd : 0.;
    for (k = 0; k < m; k++) At[i * astep + k] *= s;
  }
}

// JacobiSVDImpl_<double>: At has n rows of length m (stride astep); rows 0..n1-1 become left singular
// vectors (the rows beyond n are the seeded Gram-Schmidt completion); Vt n x n.
__device__ GL_NOINLINE void gl_jacobi_svd(double* At, int astep, double* Wout, double* Vt, int m, int n, int n1, bool want_u) {
  const double eps = DBL_EPSILON * 10;
  double W[GL_MAXN];
  int i, j, k, iter, max_iter = m > 30 ? m : 30;
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
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        double beta = a - b, gamma = gl_hypot(p, beta);
        if (beta < 0) {
          double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
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
  gl_jacobi_svd_tail(At, astep, W, Wout, Vt, m, n, n1, want_u);
}

// The 12 x 12 case of JacobiSVDImpl_ (EPnP's M^T M, the DLT of cvFindExtrinsicCameraParams2) with the matrices in LDS:
// same operation sequence as gl_jacobi_svd, but the inner k loops are unrolled (12 independent ds_read_b64 in flight
// instead of one flat load per element) and row i of At / Vt stays in registers while j runs.  P is a pointer into
// address space 3.  W (12 doubles, LDS) doubles as the singular-value output.
typedef __attribute__((address_space(3))) double gl_lds_double;
// true when a generic pointer lies in the LDS aperture (device pass only; the host pass never runs this code)
__device__ __forceinline__ bool gl_is_lds(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_is_shared(p);
#elif defined(GL_TEST_FORCE_LDS_PATH)   // x86 sanitizer harness (tests/sanitize): take the LDS-specialised routines too
  (void)p;
  return true;
#else
  (void)p;
  return false;
#endif
}
// WITH_V false: left vectors only (Vt unused), see gl_jacobi_svd_tail.
template <bool WITH_V>
__device__ inline void gl_jacobi_svd12_lds(gl_lds_double* At, gl_lds_double* W, gl_lds_double* Vt) {
  const double eps = DBL_EPSILON * 10;
  for (int i = 0; i < 12; i++) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) { double t = At[i * 12 + k]; sd += t * t; }
    W[i] = sd;
    if (WITH_V) {
#pragma unroll
      for (int k = 0; k < 12; k++) Vt[i * 12 + k] = 0;
      Vt[i * 12 + i] = 1;
    }
  }
  for (int iter = 0; iter < 30; iter++) {
    bool changed = false;
    for (int i = 0; i < 11; i++) {
      double Ai[12], Vi[12];
#pragma unroll
      for (int k = 0; k < 12; k++) { Ai[k] = At[i * 12 + k]; Vi[k] = WITH_V ? Vt[i * 12 + k] : 0.0; }
      double a = W[i];
      bool touched = false;
      for (int j = i + 1; j < 12; j++) {
        double Aj[12];
#pragma unroll
        for (int k = 0; k < 12; k++) Aj[k] = At[j * 12 + k];
        double p = 0, b = W[j];
#pragma unroll
        for (int k = 0; k < 12; k++) p += Ai[k] * Aj[k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        double beta = a - b, gamma = gl_hypot(p, beta), c, s;
        if (beta < 0) {
          double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) {
          double t0 = c * Ai[k] + s * Aj[k];
          double t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0; At[j * 12 + k] = t1;
          a += t0 * t0; b += t1 * t1;
        }
        W[j] = b;
        touched = true;
        if (WITH_V) {
#pragma unroll
          for (int k = 0; k < 12; k++) {
            double vj = Vt[j * 12 + k];
            double t0 = c * Vi[k] + s * vj;
            double t1 = -s * Vi[k] + c * vj;
            Vi[k] = t0; Vt[j * 12 + k] = t1;
          }
        }
      }
      if (touched) {
#pragma unroll
        for (int k = 0; k < 12; k++) { At[i * 12 + k] = Ai[k]; if (WITH_V) Vt[i * 12 + k] = Vi[k]; }
        W[i] = a;
        changed = true;
      }
    }
    if (!changed) break;
  }
  gl_jacobi_svd_tail((double*)At, 12, (double*)W, (double*)W, WITH_V ? (double*)Vt : nullptr, 12, 12, 12, true);
}

// JacobiSVDImpl_<double> for compile-time sizes, everything in registers: the same operation sequence as
// gl_jacobi_svd (every loop unrolled, the data-dependent row swaps of the final sort written as predicated swaps over
// static indices).  Only the main path: when a singular value is <= DBL_MIN the reference completes the basis with a
// seeded Gram-Schmidt; that case returns false and the caller re-runs the general routine.
// NEED_U = false: the caller reads W and Vt only.  The completion branch rewrites rows of At (the left vectors) and
// nothing else, so it is skipped and the call always succeeds - rank-deficient systems (e.g. triangulation with a zero
// baseline) stay on the register path.
template <int M, int N, bool NEED_U = true>
__device__ __forceinline__ bool gl_jacobi_svd_fixed(double (&At)[N * M], double (&W)[N], double (&Vt)[N * N]) {
  const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
  const int max_iter = M > 30 ? M : 30;
#pragma unroll
  for (int i = 0; i < N; i++) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < M; k++) { double t = At[i * M + k]; sd += t * t; }
    W[i] = sd;
#pragma unroll
    for (int k = 0; k < N; k++) Vt[i * N + k] = 0;
    Vt[i * N + i] = 1;
  }
  for (int iter = 0; iter < max_iter; iter++) {
    bool changed = false;
#pragma unroll
    for (int i = 0; i < N - 1; i++)
#pragma unroll
      for (int j = i + 1; j < N; j++) {
        double a = W[i], p = 0, b = W[j];
#pragma unroll
        for (int k = 0; k < M; k++) p += At[i * M + k] * At[j * M + k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        double beta = a - b, gamma = gl_hypot(p, beta), c, s;
        if (beta < 0) {
          double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
#pragma unroll
        for (int k = 0; k < M; k++) {
          double t0 = c * At[i * M + k] + s * At[j * M + k];
          double t1 = -s * At[i * M + k] + c * At[j * M + k];
          At[i * M + k] = t0; At[j * M + k] = t1;
          a += t0 * t0; b += t1 * t1;
        }
        W[i] = a; W[j] = b;
        changed = true;
#pragma unroll
        for (int k = 0; k < N; k++) {
          double t0 = c * Vt[i * N + k] + s * Vt[j * N + k];
          double t1 = -s * Vt[i * N + k] + c * Vt[j * N + k];
          Vt[i * N + k] = t0; Vt[j * N + k] = t1;
        }
      }
    if (!changed) break;
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < M; k++) { double t = At[i * M + k]; sd += t * t; }
    W[i] = sqrt(sd);
  }
#pragma unroll
  for (int i = 0; i < N - 1; i++) {
    int j = i;
    double wj = W[i];
#pragma unroll
    for (int k = i + 1; k < N; k++)
      if (wj < W[k]) { j = k; wj = W[k]; }
#pragma unroll
    for (int q = i + 1; q < N; q++)
      if (j == q) {
        gl_swap(W[i], W[q]);
#pragma unroll
        for (int k = 0; k < M; k++) gl_swap(At[i * M + k], At[q * M + k]);
#pragma unroll
        for (int k = 0; k < N; k++) gl_swap(Vt[i * N + k], Vt[q * N + k]);
      }
  }
  if (!NEED_U) return true;
  bool ok = true;
#pragma unroll
  for (int i = 0; i < N; i++) ok = ok && W[i] > minval;
  if (!ok) return false;
#pragma unroll
  for (int i = 0; i < N; i++) {
    double s = 1 / W[i];
#pragma unroll
    for (int k = 0; k < M; k++) At[i * M + k] *= s;
  }
  return true;
}

// cv::SVD::compute(A (m x n)).  Scratch buffers ta (max(m,n)^2 when full_uv, else min*max) and tv (min^2)
// are supplied by the caller.  U (m x ucols) and Vt (vrows x n) may be null.
__device__ GL_NOINLINE void gl_svd_compute(const double* A, int m, int n, double* w, double* U, double* Vt, bool full_uv,
                                      double* ta, double* tv) {
  bool at = false;
  int mm = m, nn = n;
  if (mm < nn) { int t = mm; mm = nn; nn = t; at = true; }
  int urows = full_uv ? mm : nn;
  for (int i = 0; i < urows * mm; i++) ta[i] = 0.0;
  if (!at) {
    for (int i = 0; i < m; i++)
      for (int j = 0; j < n; j++) ta[j * mm + i] = A[i * n + j];
  } else {
    for (int i = 0; i < m; i++)
      for (int j = 0; j < n; j++) ta[i * mm + j] = A[i * n + j];
  }
  gl_jacobi_svd(ta, mm, w, tv, mm, nn, urows, tv != nullptr);
  if (!at) {
    if (U)
      for (int i = 0; i < urows; i++)
        for (int k = 0; k < mm; k++) U[k * urows + i] = ta[i * mm + k];
    if (Vt)
      for (int i = 0; i < nn * nn; i++) Vt[i] = tv[i];
  } else {
    if (U)
      for (int i = 0; i < nn; i++)
        for (int k = 0; k < nn; k++) U[k * nn + i] = tv[i * nn + k];
    if (Vt)
      for (int i = 0; i < urows * mm; i++) Vt[i] = ta[i];
  }
}

// SVBkSbImpl_ (uT = vT = true).  b == nullptr -> pseudo-inverse (nb = m).  buffer: nb doubles.
__device__ GL_NOINLINE void gl_svbksb(int m, int n, const double* w, const double* Ut, int ldu, const double* Vt, int ldv,
                                 const double* b, int ldb, int nb, double* x, int ldx, double* buffer) {
  const double eps = DBL_EPSILON * 2;
  double threshold = 0;
  int nm = m < n ? m : n;
  if (!b) nb = m;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < nb; j++) x[i * ldx + j] = 0;
  for (int i = 0; i < nm; i++) threshold += w[i];
  threshold *= eps;
  for (int i = 0; i < nm; i++) {
    const double* u = Ut + i * ldu;
    const double* v = Vt + i * ldv;
    double wi = w[i];
    if (fabs(wi) <= threshold) continue;
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
        for (int k = 0; k < m; k++) {
          double s = u[k];
          for (int j = 0; j < nb; j++) buffer[j] = buffer[j] + s * b[k * ldb + j];
        }
        for (int j = 0; j < nb; j++) buffer[j] *= wi;
      } else {
        for (int j = 0; j < nb; j++) buffer[j] = u[j] * wi;
      }
      for (int k = 0; k < n; k++) {
        double s = v[k];
        for (int j = 0; j < nb; j++) x[k * ldx + j] = x[k * ldx + j] + s * buffer[j];
      }
    }
  }
}

// cv::solve(A (M x N), b, x, DECOMP_SVD) with the factorisation in registers (same sequence as gl_solve_svd: Jacobi SVD of
// A^T, then SVBkSb's single right-hand-side branch).  Returns false for the rank-deficient completion case, which the
// caller hands to gl_solve_svd_ws.
template <int M, int N>
__device__ __forceinline__ bool gl_solve_svd_fixed(const double* A, const double* b, double* x) {
  double at[N * M], w[N], vt[N * N];
#pragma unroll
  for (int i = 0; i < M; i++)
#pragma unroll
    for (int j = 0; j < N; j++) at[j * M + i] = A[i * N + j];
  if (!gl_jacobi_svd_fixed<M, N>(at, w, vt)) return false;
  const double eps = DBL_EPSILON * 2;
  double threshold = 0;
#pragma unroll
  for (int i = 0; i < N; i++) threshold += w[i];
  threshold *= eps;
#pragma unroll
  for (int j = 0; j < N; j++) x[j] = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    double wi = w[i];
    if (fabs(wi) <= threshold) continue;
    wi = 1 / wi;
    double sacc = 0;
#pragma unroll
    for (int j = 0; j < M; j++) sacc += at[i * M + j] * b[j];
    sacc *= wi;
#pragma unroll
    for (int j = 0; j < N; j++) x[j] = x[j] + sacc * vt[i * N + j];
  }
  return true;
}

// cv::solve(A (m x n, m >= n <= 6), b, x, DECOMP_SVD) with caller-supplied workspaces (e.g. in LDS): at >= m*n, vt >= n*n
__device__ inline void gl_solve_svd_ws(const double* A, int m, int n, const double* b, double* x, double* at, double* vt) {
  double w[6], buf[1];
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) at[j * m + i] = A[i * n + j];
  gl_jacobi_svd(at, m, w, vt, m, n, n, true);
  gl_svbksb(m, n, w, at, m, vt, n, b, 1, 1, x, 1, buf);
}

// cv::solve(A (m x n, m >= n <= 6), b, x, DECOMP_SVD)
__device__ inline void gl_solve_svd(const double* A, int m, int n, const double* b, double* x) {
  double at[36], w[6], vt[36], buf[1];
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) at[j * m + i] = A[i * n + j];
  gl_jacobi_svd(at, m, w, vt, m, n, n, true);
  gl_svbksb(m, n, w, at, m, vt, n, b, 1, 1, x, 1, buf);
}

// cv::invert(A 3x3, DECOMP_SVD)
__device__ GL_NOINLINE void gl_invert3_svd_general(const double* A, double* Ainv) {
  double at[9], w[3], vt[9], buf[3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) at[j * 3 + i] = A[i * 3 + j];
  gl_jacobi_svd(at, 3, w, vt, 3, 3, 3, true);
  gl_svbksb(3, 3, w, at, 3, vt, 3, nullptr, 0, 3, Ainv, 3, buf);
}
__device__ inline void gl_invert3_svd(const double* A, double* Ainv) {
  double at[9], w[3], vt[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) at[j * 3 + i] = A[i * 3 + j];
  if (!gl_jacobi_svd_fixed<3, 3>(at, w, vt)) { gl_invert3_svd_general(A, Ainv); return; }
  // SVBkSb with b = identity (pseudo-inverse), same accumulation order as gl_svbksb's nb > 1 branch
  const double eps = DBL_EPSILON * 2;
  double threshold = (w[0] + w[1] + w[2]) * eps;
#pragma unroll
  for (int k = 0; k < 9; k++) Ainv[k] = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double wi = w[i];
    if (fabs(wi) <= threshold) continue;
    wi = 1 / wi;
    double buf[3];
#pragma unroll
    for (int j = 0; j < 3; j++) buf[j] = at[i * 3 + j] * wi;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double sv = vt[i * 3 + k];
#pragma unroll
      for (int j = 0; j < 3; j++) Ainv[k * 3 + j] = Ainv[k * 3 + j] + sv * buf[j];
    }
  }
}

// SVD of a 3x3 (U, w, Vt); U / Vt may be null
__device__ GL_NOINLINE void gl_svd3_general(const double* A, double* w, double* U, double* Vt) {
  double ta[9], tv[9];
  gl_svd_compute(A, 3, 3, w, U, Vt, false, ta, tv);
}
__device__ inline void gl_svd3(const double* A, double* w, double* U, double* Vt) {
  double ta[9], ww[3], tv[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) ta[j * 3 + i] = A[i * 3 + j];
  if (!gl_jacobi_svd_fixed<3, 3>(ta, ww, tv)) { gl_svd3_general(A, w, U, Vt); return; }
#pragma unroll
  for (int i = 0; i < 3; i++) w[i] = ww[i];
  if (U) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int k = 0; k < 3; k++) U[k * 3 + i] = ta[i * 3 + k];
  }
  if (Vt) {
#pragma unroll
    for (int i = 0; i < 9; i++) Vt[i] = tv[i];
  }
}

// cv::solveCubic (4 double coefficients)
__device__ GL_NOINLINE int gl_solve_cubic(const double c[4], double x[3]) {
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
        d = sqrt(d);
        double q1 = (-a2 + d) * 0.5;
        double q2 = (a2 + d) * -0.5;
        if (fabs(q1) > fabs(q2)) { x0 = q1 / a1; x1 = a3 / q1; }
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
      double theta = acos(R / sqrt(Qcubed));
      double sqrtQ = sqrt(Q);
      double t0 = -2 * sqrtQ;
      double t1 = theta * (1. / 3);
      double t2 = a1 * (1. / 3);
      x0 = t0 * cos(t1) - t2;
      x1 = t0 * cos(t1 + (2. * M_PI / 3)) - t2;
      x2 = t0 * cos(t1 + (4. * M_PI / 3)) - t2;
      n = 3;
    } else if (d == 0) {
      if (R >= 0) {
        x0 = -2 * pow(R, 1. / 3) - a1 / 3;
        x1 = pow(R, 1. / 3) - a1 / 3;
      } else {
        x0 = 2 * pow(-R, 1. / 3) - a1 / 3;
        x1 = -pow(-R, 1. / 3) - a1 / 3;
      }
      x2 = 0;
      n = x0 == x1 ? 1 : 2;
      x1 = x0 == x1 ? 0 : x1;
    } else {
      double e;
      d = sqrt(-d);
      e = pow(d + fabs(R), 1. / 3);
      if (R > 0) e = -e;
      x0 = (e + Q / e) - a1 * (1. / 3);
      n = 1;
    }
  }
  x[0] = x0; x[1] = x1; x[2] = x2;
  return n;
}

__device__ __forceinline__ double gl_det3(const double* M) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
__device__ inline void gl_mat3mul(const double* A, const double* B, double* C) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  for (int i = 0; i < 9; i++) C[i] = t[i];
}

// RANSACUpdateNumIters (ptsetreg.cpp)
__device__ GL_NOINLINE int gl_ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
  p = fmax(p, 0.); p = fmin(p, 1.);
  ep = fmax(ep, 0.); ep = fmin(ep, 1.);
  double num = fmax(1. - p, DBL_MIN);
  double denom = 1. - pow(1. - ep, (double)modelPoints);
  if (denom < DBL_MIN) return 0;
  num = log(num);
  denom = log(denom);
  return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : __double2int_rn(num / denom);
}

// haveCollinearPoints<float>: only the last point against earlier pairs
__device__ GL_NOINLINE bool gl_have_collinear(const float* pts, int count) {
  int i = count - 1;
  for (int j = 0; j < i; j++) {
    double dx1 = pts[2 * j] - pts[2 * i];
    double dy1 = pts[2 * j + 1] - pts[2 * i + 1];
    for (int k = 0; k < j; k++) {
      double dx2 = pts[2 * k] - pts[2 * i];
      double dy2 = pts[2 * k + 1] - pts[2 * i + 1];
      if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
    }
  }
  return false;
}
