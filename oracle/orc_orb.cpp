// oracle/orc_orb.cpp — TEST INFRASTRUCTURE ONLY (see orc_common.h header).  PARITY UNPINNED.
//
// CPU restatement of cv::ORB::create(nfeatures)->detectAndCompute(img, noArray(), kps, desc) with
// OpenCV-4.6 defaults (scaleFactor 1.2f, nlevels 8, edgeThreshold 31, firstLevel 0, WTA_K 2,
// HARRIS_SCORE, patchSize 31, fastThreshold 20), as called by the reference at
// src/feature_processor.cpp:5-23 (via Frame::extract_observations, src/frame.cpp:8-17).
// Follows SURVEY.md Appendix A.1 (features2d/src/{orb,fast,fast_score,keypoint}.cpp,
// imgproc/src/{resize,smooth,filter}.cpp, core/src/mathfuncs_core.simd.hpp).
//
// One reasoned departure from SURVEY A.1.7 ([M]-tagged there): ORB blurs each level *in place on a
// sub-matrix* of the packed pyramid with borderType REFLECT_101 (no BORDER_ISOLATED).  cv::GaussianBlur
// takes its bit-exact fixed-point branch only when `(borderType & BORDER_ISOLATED) || !src.isSubmatrix()`;
// that test fails here, so the call falls through to sepFilter2D with the float kernel converted to
// 8-bit fixed point by cvRound (taps [18,34,49,55,49,34,18], sum 257) and a (sum+2^15)>>16 saturating
// column pass.  blur_mode 0 (default) = that path; blur_mode 1 = the error-diffused bit-exact kernel
// [18,34,48,56,48,34,18] of the fixed-point branch.
#include "orc_common.h"
#include "mvo_oracle.h"

namespace orc {

static const int kPattern[256 * 4] = {
#include "orb_pattern_31.inc"
};

struct Lvl {
  int w = 0, h = 0;
  float scale = 1.f;
  std::vector<u8> d;       // un-blurred
  std::vector<u8> blur;    // blurred (filled lazily)
  inline int at(int x, int y) const { return d[(size_t)reflect101(y, h) * w + reflect101(x, w)]; }
};

// ---- FAST-9/16 (features2d/src/fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>) ---------------
static const int kCircle[16][2] = {{0, 3},  {1, 3},  {2, 2},  {3, 1},  {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                   {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int corner_score16(const u8* img, int stride, int x, int y, int threshold) {
  const int K = 8, N = K * 3 + 1;
  int v = img[(size_t)y * stride + x];
  short d[N];
  for (int k = 0; k < N; k++) {
    const int* o = kCircle[k & 15];
    d[k] = (short)(v - img[(size_t)(y + o[1]) * stride + x + o[0]]);
  }
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = std::min((int)d[k + 1], (int)d[k + 2]);
    a = std::min(a, (int)d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, (int)d[k + 4]);
    a = std::min(a, (int)d[k + 5]);
    a = std::min(a, (int)d[k + 6]);
    a = std::min(a, (int)d[k + 7]);
    a = std::min(a, (int)d[k + 8]);
    a0 = std::max(a0, std::min(a, (int)d[k]));
    a0 = std::max(a0, std::min(a, (int)d[k + 9]));
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = std::max((int)d[k + 1], (int)d[k + 2]);
    b = std::max(b, (int)d[k + 3]);
    b = std::max(b, (int)d[k + 4]);
    b = std::max(b, (int)d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, (int)d[k + 6]);
    b = std::max(b, (int)d[k + 7]);
    b = std::max(b, (int)d[k + 8]);
    b0 = std::min(b0, std::max(b, (int)d[k]));
    b0 = std::min(b0, std::max(b, (int)d[k + 9]));
  }
  return -b0 - 1;
}

struct FastKp { int x, y, score; };

// Row-major list of NMS survivors: corner iff >= 9 contiguous circle pixels brighter than v+t or
// darker than v-t; kept iff its score is strictly greater than all 8 neighbours' scores.
static void fast9_nms(const u8* img, int w, int h, int stride, int threshold, std::vector<FastKp>& out) {
  out.clear();
  if (w < 7 || h < 7) return;
  std::vector<u8> score((size_t)w * h, 0);
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      int v = img[(size_t)y * stride + x];
      int vb = v + threshold, vd = v - threshold;
      int cb = 0, cd = 0;
      bool corner = false;
      for (int k = 0; k < 25 && !corner; k++) {
        const int* o = kCircle[k & 15];
        int p = img[(size_t)(y + o[1]) * stride + x + o[0]];
        if (p > vb) { if (++cb > 8) corner = true; } else cb = 0;
        if (p < vd) { if (++cd > 8) corner = true; } else cd = 0;
      }
      if (corner) score[(size_t)y * w + x] = (u8)corner_score16(img, stride, x, y, threshold);
    }
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      int s = score[(size_t)y * w + x];
      if (!s) continue;
      const u8* p = &score[(size_t)y * w + x];
      if (s > p[-1] && s > p[1] && s > p[-w - 1] && s > p[-w] && s > p[-w + 1] && s > p[w - 1] &&
          s > p[w] && s > p[w + 1])
        out.push_back({x, y, s});
    }
}

// ---- resize INTER_LINEAR_EXACT (imgproc/src/resize.cpp resize_bitExact, ufixedpoint16 path) -------
struct Lin { std::vector<int> ofs, c0, c1; int mn, mx; };
static void lin_coeffs(int srcsize, int dstsize, Lin& L) {
  double inv_scale = (double)dstsize / srcsize;
  double scale = 1.0 / inv_scale;
  L.ofs.assign(dstsize, 0); L.c0.assign(dstsize, 256); L.c1.assign(dstsize, 0);
  L.mn = 0; L.mx = dstsize;
  for (int v = 0; v < dstsize; v++) {
    double fval = scale * ((double)v + 0.5) - 0.5;
    int ival = cv_floor(fval);
    if (ival >= 0 && srcsize > 1) {
      if (ival < srcsize - 1) {
        L.ofs[v] = ival;
        L.c1[v] = cv_round((fval - (double)ival) * 256.0);
        L.c0[v] = 256 - L.c1[v];
      } else {
        L.ofs[v] = srcsize - 1;
        L.mx = std::min(L.mx, v);
      }
    } else {
      L.mn = std::max(L.mn, v + 1);
    }
  }
}

static void resize_linear_exact(const Lvl& s, Lvl& o) {
  Lin X, Y;
  lin_coeffs(s.w, o.w, X);
  lin_coeffs(s.h, o.h, Y);
  o.d.resize((size_t)o.w * o.h);
  auto hline = [&](int sy, std::vector<unsigned>& row) {  // 8.8 fixed point
    const u8* sp = &s.d[(size_t)sy * s.w];
    for (int x = 0; x < o.w; x++) {
      if (x < X.mn) row[x] = (unsigned)sp[0] << 8;
      else if (x >= X.mx) row[x] = (unsigned)sp[s.w - 1] << 8;
      else row[x] = (unsigned)X.c0[x] * sp[X.ofs[x]] + (unsigned)X.c1[x] * sp[X.ofs[x] + 1];
    }
  };
  std::vector<unsigned> r0(o.w), r1(o.w);
  for (int y = 0; y < o.h; y++) {
    u8* dp = &o.d[(size_t)y * o.w];
    if (y < Y.mn || y >= Y.mx) {
      hline(y < Y.mn ? 0 : s.h - 1, r0);
      for (int x = 0; x < o.w; x++) dp[x] = (u8)std::min(255u, (r0[x] + 128) >> 8);
    } else {
      hline(Y.ofs[y], r0);
      hline(Y.ofs[y] + 1, r1);
      for (int x = 0; x < o.w; x++) {
        unsigned v = r0[x] * (unsigned)Y.c0[y] + r1[x] * (unsigned)Y.c1[y];
        dp[x] = (u8)std::min(255u, (v + 32768u) >> 16);
      }
    }
  }
}

// ---- 7x7 sigma=2 Gaussian as ORB gets it (see file header) ---------------------------------------
static void blur7(Lvl& L, int mode) {
  static const int k0[7] = {18, 34, 49, 55, 49, 34, 18};
  static const int k1[7] = {18, 34, 48, 56, 48, 34, 18};
  const int* k = mode ? k1 : k0;
  L.blur.resize((size_t)L.w * L.h);
  std::vector<int> tmp((size_t)L.w * (L.h + 6));
  for (int y = -3; y < L.h + 3; y++)
    for (int x = 0; x < L.w; x++) {
      int s = 0;
      for (int t = 0; t < 7; t++) s += k[t] * L.at(x + t - 3, y);
      tmp[(size_t)(y + 3) * L.w + x] = s;
    }
  for (int y = 0; y < L.h; y++)
    for (int x = 0; x < L.w; x++) {
      int s = 0;
      for (int t = 0; t < 7; t++) s += k[t] * tmp[(size_t)(y + t) * L.w + x];
      L.blur[(size_t)y * L.w + x] = (u8)std::min(255, (s + 32768) >> 16);
    }
}

// ---- fastAtan2 (core mathfuncs_core.simd.hpp atan_f32), degrees in [0,360) -----------------------
static float fast_atan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
  static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
  static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
  static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
  float ax = std::abs(x), ay = std::abs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

struct KP { float x, y, size, angle, response; int octave, class_id; };

// KeyPointsFilter::retainBest — libstdc++ nth_element/partition order is part of the contract.
static void retain_best(std::vector<KP>& k, int n) {
  if (n >= 0 && k.size() > (size_t)n) {
    if (n == 0) { k.clear(); return; }
    std::nth_element(k.begin(), k.begin() + n - 1, k.end(),
                     [](const KP& a, const KP& b) { return a.response > b.response; });
    float amb = k[n - 1].response;
    auto e = std::partition(k.begin() + n, k.end(), [amb](const KP& a) { return a.response >= amb; });
    k.resize(e - k.begin());
  }
}

// Test hook: retainBest on bare responses, returning the original indices in the order the vector is left in.
// depth_limit >= 0 calls libstdc++'s own __introselect with that limit (instead of 2*floor(log2 n)) so that the
// heap-select branch can be exercised; the rest is nth_element's body verbatim in behaviour.
extern "C" int orc_retain_best(const float* resp, int n, int n_keep, int depth_limit, int* out_idx) {
  struct RI { float response; int idx; };
  std::vector<RI> k(n);
  for (int i = 0; i < n; i++) k[i] = {resp[i], i};
  auto gt = [](const RI& a, const RI& b) { return a.response > b.response; };
  if (n_keep >= 0 && k.size() > (size_t)n_keep) {
    if (n_keep == 0) k.clear();
    else {
      if (depth_limit < 0) std::nth_element(k.begin(), k.begin() + n_keep - 1, k.end(), gt);
      else std::__introselect(k.begin(), k.begin() + n_keep - 1, k.end(), (long)depth_limit, __gnu_cxx::__ops::__iter_comp_iter(gt));
      float amb = k[n_keep - 1].response;
      auto e = std::partition(k.begin() + n_keep, k.end(), [amb](const RI& a) { return a.response >= amb; });
      k.resize(e - k.begin());
    }
  }
  for (size_t i = 0; i < k.size(); i++) out_idx[i] = k[i].idx;
  return (int)k.size();
}

static void orb_level_sizes(int w, int h, int nlevels, std::vector<Lvl>& L) {
  L.resize(nlevels);
  double sf = (double)1.2f;
  for (int l = 0; l < nlevels; l++) {
    float scale = (float)std::pow(sf, (double)l);
    float inv = 1.0f / scale;
    L[l].scale = scale;
    L[l].w = cv_round(w * inv);
    L[l].h = cv_round(h * inv);
  }
}

static void features_per_level(int nfeatures, int nlevels, std::vector<int>& q) {
  q.resize(nlevels);
  float factor = (float)(1.0 / (double)1.2f);
  float nd = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; l++) {
    q[l] = cv_round(nd);
    sum += q[l];
    nd *= factor;
  }
  q[nlevels - 1] = std::max(nfeatures - sum, 0);
}

static void umax_table(std::vector<int>& umax) {
  const int half = 15;
  umax.assign(half + 2, 0);
  int v, v0, vmax = cv_floor(half * std::sqrt(2.f) / 2 + 1);
  int vmin = cv_ceil(half * std::sqrt(2.f) / 2);
  for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt((double)half * half - v * v));
  for (v = half, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}

}  // namespace orc

using namespace orc;

extern "C" int orc_fast9_nms(const unsigned char* img, int w, int h, int stride, int threshold,
                             int* xys, int cap) {
  std::vector<FastKp> k;
  fast9_nms(img, w, h, stride, threshold, k);
  int n = (int)std::min<size_t>(k.size(), (size_t)cap);
  for (int i = 0; i < n; i++) { xys[3 * i] = k[i].x; xys[3 * i + 1] = k[i].y; xys[3 * i + 2] = k[i].score; }
  return (int)k.size();
}

extern "C" int orc_orb_level_info(int w, int h, int nfeatures, int* lw, int* lh, float* scale, int* quota) {
  std::vector<Lvl> L; std::vector<int> q;
  orb_level_sizes(w, h, 8, L);
  features_per_level(nfeatures, 8, q);
  for (int l = 0; l < 8; l++) { lw[l] = L[l].w; lh[l] = L[l].h; scale[l] = L[l].scale; quota[l] = q[l]; }
  return 8;
}

extern "C" int orc_resize_linear_exact(const unsigned char* src, int sw, int sh, int sstride,
                                       unsigned char* dst, int dw, int dh, int dstride) {
  Lvl s, o;
  s.w = sw; s.h = sh; s.d.resize((size_t)sw * sh);
  for (int y = 0; y < sh; y++) memcpy(&s.d[(size_t)y * sw], src + (size_t)y * sstride, sw);
  o.w = dw; o.h = dh;
  resize_linear_exact(s, o);
  for (int y = 0; y < dh; y++) memcpy(dst + (size_t)y * dstride, &o.d[(size_t)y * dw], dw);
  return 0;
}

extern "C" int orc_gauss7(const unsigned char* src, int w, int h, int stride, unsigned char* dst,
                          int dstride, int mode) {
  Lvl s;
  s.w = w; s.h = h; s.d.resize((size_t)w * h);
  for (int y = 0; y < h; y++) memcpy(&s.d[(size_t)y * w], src + (size_t)y * stride, w);
  blur7(s, mode);
  for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * dstride, &s.blur[(size_t)y * w], w);
  return 0;
}

extern "C" float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }

// channels: 1 (mono8) or 3 (BGR8, converted with cvtColor's 15-bit fixed-point weights
// BY15=3735, GY15=19235, RY15=9798 of OpenCV 4.x color_rgb.simd.hpp RGB2Gray<uchar>).
extern "C" int orc_orb_detect_and_compute(const unsigned char* img, int w, int h, int stride,
                                          int channels, int nfeatures, int fast_threshold,
                                          int blur_mode, orc_keypoint* out_kps,
                                          unsigned char* out_desc, int cap) {
  const int nlevels = 8, edge = 31, patch = 31, half = 15;
  std::vector<Lvl> L;
  orb_level_sizes(w, h, nlevels, L);
  // gray
  L[0].d.resize((size_t)w * h);
  for (int y = 0; y < h; y++) {
    const u8* sp = img + (size_t)y * stride;
    u8* dp = &L[0].d[(size_t)y * w];
    if (channels == 1) memcpy(dp, sp, w);
    else
      for (int x = 0; x < w; x++)
        dp[x] = (u8)((sp[3 * x] * 3735 + sp[3 * x + 1] * 19235 + sp[3 * x + 2] * 9798 + (1 << 14)) >> 15);
  }
  for (int l = 1; l < nlevels; l++) resize_linear_exact(L[l - 1], L[l]);

  std::vector<int> quota, umax;
  features_per_level(nfeatures, nlevels, quota);
  umax_table(umax);

  std::vector<KP> all;
  std::vector<int> counters(nlevels);
  std::vector<FastKp> fk;
  std::vector<KP> kps;
  for (int l = 0; l < nlevels; l++) {
    const Lvl& V = L[l];
    fast9_nms(V.d.data(), V.w, V.h, V.w, fast_threshold, fk);
    kps.clear();
    // runByImageBorder(edgeThreshold)
    if (!(V.h <= edge * 2 || V.w <= edge * 2))
      for (auto& f : fk)
        if (f.x >= edge && f.x < V.w - edge && f.y >= edge && f.y < V.h - edge)
          kps.push_back({(float)f.x, (float)f.y, 7.f, -1.f, (float)f.score, 0, -1});
    retain_best(kps, 2 * quota[l]);
    counters[l] = (int)kps.size();
    for (auto& k : kps) { k.octave = l; k.size = patch * V.scale; }
    all.insert(all.end(), kps.begin(), kps.end());
  }
  if (all.empty()) return 0;

  // HarrisResponses(blockSize 7, k 0.04)
  {
    const int bs = 7, r = bs / 2;
    float scale = 1.f / ((1 << 2) * bs * 255.f);
    float scale_sq_sq = scale * scale * scale * scale;
    const float harris_k = 0.04f;
    for (auto& k : all) {
      const Lvl& V = L[k.octave];
      int x0 = cv_round(k.x), y0 = cv_round(k.y);
      int a = 0, b = 0, c = 0;
      for (int i = 0; i < bs; i++)
        for (int j = 0; j < bs; j++) {
          int X = x0 - r + j, Y = y0 - r + i;
          int Ix = (V.at(X + 1, Y) - V.at(X - 1, Y)) * 2 + (V.at(X + 1, Y - 1) - V.at(X - 1, Y - 1)) +
                   (V.at(X + 1, Y + 1) - V.at(X - 1, Y + 1));
          int Iy = (V.at(X, Y + 1) - V.at(X, Y - 1)) * 2 + (V.at(X - 1, Y + 1) - V.at(X - 1, Y - 1)) +
                   (V.at(X + 1, Y + 1) - V.at(X + 1, Y - 1));
          a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
      k.response = ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
    }
  }
  std::vector<KP> fin;
  {
    int off = 0;
    for (int l = 0; l < nlevels; l++) {
      kps.assign(all.begin() + off, all.begin() + off + counters[l]);
      off += counters[l];
      retain_best(kps, quota[l]);
      fin.insert(fin.end(), kps.begin(), kps.end());
    }
  }
  // ICAngles on the un-blurred levels
  for (auto& k : fin) {
    const Lvl& V = L[k.octave];
    int cx = cv_round(k.x), cy = cv_round(k.y);
    int m01 = 0, m10 = 0;
    for (int u = -half; u <= half; ++u) m10 += u * V.at(cx + u, cy);
    for (int v = 1; v <= half; ++v) {
      int vsum = 0, d = umax[v];
      for (int u = -d; u <= d; ++u) {
        int vp = V.at(cx + u, cy + v), vm = V.at(cx + u, cy - v);
        vsum += (vp - vm);
        m10 += u * (vp + vm);
      }
      m01 += v * vsum;
    }
    k.angle = fast_atan2((float)m01, (float)m10);
  }
  for (auto& k : fin) { float s = L[k.octave].scale; k.x *= s; k.y *= s; }

  // descriptors on blurred levels
  for (int l = 0; l < nlevels; l++) blur7(L[l], blur_mode);
  int n = (int)fin.size();
  int nout = std::min(n, cap);
  for (int j = 0; j < nout; j++) {
    const KP& k = fin[j];
    const Lvl& V = L[k.octave];
    float scale = 1.f / V.scale;
    float angle = k.angle;
    angle *= (float)(M_PI / 180.f);
    float a = (float)std::cos(angle), b = (float)std::sin(angle);
    int cx = cv_round(k.x * scale), cy = cv_round(k.y * scale);
    const int* pat = kPattern;
    u8* desc = out_desc + (size_t)j * 32;
    for (int i = 0; i < 32; i++, pat += 32) {
      int val = 0;
      for (int t = 0; t < 8; t++) {
        const int* p = pat + 4 * t;
        float x0 = p[0] * a - p[1] * b, y0 = p[0] * b + p[1] * a;
        float x1 = p[2] * a - p[3] * b, y1 = p[2] * b + p[3] * a;
        int t0 = V.blur[(size_t)reflect101(cy + cv_round(y0), V.h) * V.w + reflect101(cx + cv_round(x0), V.w)];
        int t1 = V.blur[(size_t)reflect101(cy + cv_round(y1), V.h) * V.w + reflect101(cx + cv_round(x1), V.w)];
        val |= (t0 < t1) << t;
      }
      desc[i] = (u8)val;
    }
    out_kps[j] = {k.x, k.y, k.size, k.angle, k.response, k.octave, k.class_id};
  }
  return n;
}
