// oracle/orc_lk.cpp — TEST INFRASTRUCTURE ONLY (see orc_common.h header).  PARITY UNPINNED.
//
// CPU restatement of cv::calcOpticalFlowPyrLK with all-default arguments, as called by the
// reference at src/tracker.cpp:68-69 (winSize 21x21, maxLevel 3, criteria COUNT+EPS(30,0.01),
// flags 0, minEigThreshold 1e-4), and of cv::buildOpticalFlowPyramid / cv::pyrDown that it uses.
// Follows SURVEY.md Appendix A.3 (OpenCV video/src/lkpyramid.cpp, imgproc/src/pyramids.cpp).
//
// One documented refinement: OpenCV accumulates the normal-equation sums (A11,A12,A22,b1,b2) in
// float, in a build-dependent SIMD lane order, so its results are not bit-reproducible between
// OpenCV builds.  All summands are integers; this oracle accumulates them exactly (int64) and
// converts once, which is the order-independent value every OpenCV build approximates to ~1e-7
// relative.  The HIP kernel does the same, which makes LK parity bit-exact instead of tolerance-based.
//
// Channel semantics: the reference converts every image to BGR8 (src/mono_vo.cpp:94) so LK runs on
// 3 identical channels.  With identical channels every integer sum is exactly cn x the 1-channel sum;
// `cn` reproduces that (it matters for minEig, whose denominator has no cn — SURVEY A.3.6).
// True-colour input (orc_lk_track_planes): OpenCV treats a window row of a cn-channel image as cn * winSize.width interleaved
// elements - pyrDown, the Scharr derivative and the bilinear template are per channel, and the five normal-equation sums and
// the error sum simply run over all of them (lkpyramid.cpp: `x < winSize.width * cn`); the scale constants keep cn = 3.  So the
// planes are processed like independent images and their integer sums added before the one conversion to float.
#include "orc_common.h"
#include "mvo_oracle.h"

namespace orc {

struct Img {
  int w = 0, h = 0;
  std::vector<u8> d;
  inline int at(int x, int y) const { return d[(size_t)reflect101(y, h) * w + reflect101(x, w)]; }
};

// cv::pyrDown, 8-bit, BORDER_REFLECT_101: separable [1 4 6 4 1], (sum + 128) >> 8.
static void pyr_down(const Img& s, Img& o) {
  o.w = (s.w + 1) / 2;
  o.h = (s.h + 1) / 2;
  o.d.resize((size_t)o.w * o.h);
  std::vector<int> rows[5];
  for (auto& r : rows) r.resize(o.w);
  for (int y = 0; y < o.h; y++) {
    for (int k = 0; k < 5; k++) {
      int sy = reflect101(2 * y - 2 + k, s.h);
      const u8* sp = &s.d[(size_t)sy * s.w];
      for (int x = 0; x < o.w; x++) {
        int x0 = reflect101(2 * x - 2, s.w), x1 = reflect101(2 * x - 1, s.w), x2 = 2 * x;
        int x3 = reflect101(2 * x + 1, s.w), x4 = reflect101(2 * x + 2, s.w);
        rows[k][x] = sp[x2] * 6 + (sp[x1] + sp[x3]) * 4 + sp[x0] + sp[x4];
      }
    }
    for (int x = 0; x < o.w; x++) {
      int v = rows[2][x] * 6 + (rows[1][x] + rows[3][x]) * 4 + rows[0][x] + rows[4][x];
      o.d[(size_t)y * o.w + x] = (u8)((v + 128) >> 8);
    }
  }
}

// calcSharrDeriv: interleaved (dx,dy) int16 per pixel; reflect-101 at the image edge.
static void scharr_deriv(const Img& s, std::vector<short>& dxy) {
  int rows = s.h, cols = s.w;
  dxy.assign((size_t)rows * cols * 2, 0);
  std::vector<int> t0(cols + 2), t1(cols + 2);
  for (int y = 0; y < rows; y++) {
    const u8* r0 = &s.d[(size_t)(y > 0 ? y - 1 : rows > 1 ? 1 : 0) * cols];
    const u8* r1 = &s.d[(size_t)y * cols];
    const u8* r2 = &s.d[(size_t)(y < rows - 1 ? y + 1 : rows > 1 ? rows - 2 : 0) * cols];
    int* a = t0.data() + 1;
    int* b = t1.data() + 1;
    for (int x = 0; x < cols; x++) {
      a[x] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
      b[x] = r2[x] - r0[x];
    }
    int x0 = cols > 1 ? 1 : 0, x1 = cols > 1 ? cols - 2 : 0;
    a[-1] = a[x0]; a[cols] = a[x1];
    b[-1] = b[x0]; b[cols] = b[x1];
    for (int x = 0; x < cols; x++) {
      dxy[((size_t)y * cols + x) * 2 + 0] = (short)(a[x + 1] - a[x - 1]);
      dxy[((size_t)y * cols + x) * 2 + 1] = (short)((b[x + 1] + b[x - 1]) * 3 + b[x] * 10);
    }
  }
}

static inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

}  // namespace orc

using namespace orc;

extern "C" int orc_pyrdown(const unsigned char* src, int w, int h, int stride, unsigned char* dst,
                           int dstride) {
  Img s, o;
  s.w = w; s.h = h; s.d.resize((size_t)w * h);
  for (int y = 0; y < h; y++) memcpy(&s.d[(size_t)y * w], src + (size_t)y * stride, w);
  pyr_down(s, o);
  for (int y = 0; y < o.h; y++) memcpy(dst + (size_t)y * dstride, &o.d[(size_t)y * o.w], o.w);
  return 0;
}

// `np` planes per image (1: one plane whose sums are scaled by cn, the replicated-mono case; 3: the channels of a true-colour
// image, cn = 3 in the scale constants only).  plane c of an image: base + c * plane_step, pixels pixel_step bytes apart.
static int lk_track_core(const unsigned char* prev, const unsigned char* next, int w, int h, int stride, int pixel_step,
                         int plane_step, int np, int cn, const float* prev_pts, int n, float* next_pts,
                         unsigned char* status, float* err, int win, int max_level,
                         int max_count, double epsilon, double min_eig_thr) {
  if (cn < 1) cn = 1;
  const int mult = np == 1 ? cn : 1;   // the factor that stands in for the channels that are not there
  // buildOpticalFlowPyramid: stop when the next level would be <= winSize in either dimension.
  std::vector<std::vector<Img>> PP(np, std::vector<Img>(1)), NN(np, std::vector<Img>(1));
  int levels = 0;
  for (int c = 0; c < np; c++) {
    std::vector<Img>& P = PP[c];
    std::vector<Img>& N = NN[c];
    P[0].w = N[0].w = w; P[0].h = N[0].h = h;
    P[0].d.resize((size_t)w * h); N[0].d.resize((size_t)w * h);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        P[0].d[(size_t)y * w + x] = prev[(size_t)y * stride + (size_t)x * pixel_step + (size_t)c * plane_step];
        N[0].d[(size_t)y * w + x] = next[(size_t)y * stride + (size_t)x * pixel_step + (size_t)c * plane_step];
      }
    levels = 0;
    for (int l = 1; l <= max_level; l++) {
      int sw = (P[l - 1].w + 1) / 2, sh = (P[l - 1].h + 1) / 2;
      if (sw <= win || sh <= win) break;
      P.emplace_back(); N.emplace_back();
      pyr_down(P[l - 1], P[l]);
      pyr_down(N[l - 1], N[l]);
      levels = l;
    }
  }
  // criteria clamp + square (calcOpticalFlowPyrLK prologue)
  max_count = std::min(std::max(max_count, 0), 100);
  epsilon = std::min(std::max(epsilon, 0.), 10.);
  epsilon *= epsilon;

  for (int i = 0; i < n; i++) { status[i] = 1; err[i] = 0.f; }
  const int W_BITS = 14;
  const float FLT_SCALE = 1.f / (1 << 20);
  const float half = (win - 1) * 0.5f;
  std::vector<short> Iw((size_t)np * win * win), dIw((size_t)np * win * win * 2);

  for (int level = levels; level >= 0; level--) {
    const Img& I = PP[0][level];   // geometry (all planes share it)
    const Img& J = NN[0][level];
    std::vector<std::vector<short>> dIs(np);
    for (int c = 0; c < np; c++) scharr_deriv(PP[c][level], dIs[c]);
    auto deriv = [&](int ch, int x, int y, int c) -> int {  // zero (BORDER_CONSTANT) outside the image
      if ((unsigned)x >= (unsigned)I.w || (unsigned)y >= (unsigned)I.h) return 0;
      return dIs[ch][((size_t)y * I.w + x) * 2 + c];
    };
    for (int p = 0; p < n; p++) {
      float px = prev_pts[2 * p] * (float)(1. / (1 << level));
      float py = prev_pts[2 * p + 1] * (float)(1. / (1 << level));
      float nx, ny;
      if (level == levels) { nx = px; ny = py; }
      else { nx = next_pts[2 * p] * 2.f; ny = next_pts[2 * p + 1] * 2.f; }
      next_pts[2 * p] = nx; next_pts[2 * p + 1] = ny;

      px -= half; py -= half;
      int ipx = cv_floor(px), ipy = cv_floor(py);
      if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
        if (level == 0) { status[p] = 0; err[p] = 0; }
        continue;
      }
      float a = px - ipx, b = py - ipy;
      int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
      int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
      int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
      int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      int64_t sA11 = 0, sA12 = 0, sA22 = 0;
      for (int ch = 0; ch < np; ch++) {
        const Img& Ic = PP[ch][level];
        const size_t o = (size_t)ch * win * win;
        for (int y = 0; y < win; y++)
          for (int x = 0; x < win; x++) {
            int X = ipx + x, Y = ipy + y;
            int ival = descale(Ic.at(X, Y) * iw00 + Ic.at(X + 1, Y) * iw01 + Ic.at(X, Y + 1) * iw10 +
                               Ic.at(X + 1, Y + 1) * iw11, W_BITS - 5);
            int ixval = descale(deriv(ch, X, Y, 0) * iw00 + deriv(ch, X + 1, Y, 0) * iw01 +
                                deriv(ch, X, Y + 1, 0) * iw10 + deriv(ch, X + 1, Y + 1, 0) * iw11, W_BITS);
            int iyval = descale(deriv(ch, X, Y, 1) * iw00 + deriv(ch, X + 1, Y, 1) * iw01 +
                                deriv(ch, X, Y + 1, 1) * iw10 + deriv(ch, X + 1, Y + 1, 1) * iw11, W_BITS);
            Iw[o + y * win + x] = (short)ival;
            dIw[(o + y * win + x) * 2] = (short)ixval;
            dIw[(o + y * win + x) * 2 + 1] = (short)iyval;
            sA11 += (int64_t)ixval * ixval;
            sA12 += (int64_t)ixval * iyval;
            sA22 += (int64_t)iyval * iyval;
          }
      }
      float A11 = (float)(sA11 * mult) * FLT_SCALE;
      float A12 = (float)(sA12 * mult) * FLT_SCALE;
      float A22 = (float)(sA22 * mult) * FLT_SCALE;
      float D = A11 * A22 - A12 * A12;
      float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                     (float)(2 * win * win);
      if (minEig < min_eig_thr || D < FLT_EPSILON) {
        if (level == 0) status[p] = 0;
        continue;
      }
      D = 1.f / D;
      nx -= half; ny -= half;
      float pdx = 0.f, pdy = 0.f;
      for (int j = 0; j < max_count; j++) {
        int inx = cv_floor(nx), iny = cv_floor(ny);
        if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
          if (level == 0) status[p] = 0;
          break;
        }
        a = nx - inx; b = ny - iny;
        iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
        iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
        iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        int64_t sb1 = 0, sb2 = 0;
        for (int ch = 0; ch < np; ch++) {
          const Img& Jc = NN[ch][level];
          const size_t o = (size_t)ch * win * win;
          for (int y = 0; y < win; y++)
            for (int x = 0; x < win; x++) {
              int X = inx + x, Y = iny + y;
              int diff = descale(Jc.at(X, Y) * iw00 + Jc.at(X + 1, Y) * iw01 + Jc.at(X, Y + 1) * iw10 +
                                 Jc.at(X + 1, Y + 1) * iw11, W_BITS - 5) - Iw[o + y * win + x];
              sb1 += (int64_t)diff * dIw[(o + y * win + x) * 2];
              sb2 += (int64_t)diff * dIw[(o + y * win + x) * 2 + 1];
            }
        }
        float b1 = (float)(sb1 * mult) * FLT_SCALE;
        float b2 = (float)(sb2 * mult) * FLT_SCALE;
        float dx = (float)((A12 * b2 - A22 * b1) * D);
        float dy = (float)((A12 * b1 - A11 * b2) * D);
        nx += dx; ny += dy;
        next_pts[2 * p] = nx + half; next_pts[2 * p + 1] = ny + half;
        if ((double)dx * dx + (double)dy * dy <= epsilon) break;
        if (j > 0 && std::abs(dx + pdx) < 0.01 && std::abs(dy + pdy) < 0.01) {
          next_pts[2 * p] -= dx * 0.5f;
          next_pts[2 * p + 1] -= dy * 0.5f;
          break;
        }
        pdx = dx; pdy = dy;
      }
      if (status[p] && level == 0) {
        float ex = next_pts[2 * p] - half, ey = next_pts[2 * p + 1] - half;
        int inx = cv_floor(ex), iny = cv_floor(ey);
        if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
          status[p] = 0;
          continue;
        }
        float aa = ex - inx, bb = ey - iny;
        iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
        iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
        iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        int64_t se = 0;
        for (int ch = 0; ch < np; ch++) {
          const Img& Jc = NN[ch][level];
          const size_t o = (size_t)ch * win * win;
          for (int y = 0; y < win; y++)
            for (int x = 0; x < win; x++) {
              int X = inx + x, Y = iny + y;
              int diff = descale(Jc.at(X, Y) * iw00 + Jc.at(X + 1, Y) * iw01 + Jc.at(X, Y + 1) * iw10 +
                                 Jc.at(X + 1, Y + 1) * iw11, W_BITS - 5) - Iw[o + y * win + x];
              se += std::abs(diff);
            }
        }
        float errval = (float)(se * mult);
        err[p] = errval * 1.f / (32 * win * cn * win);
      }
    }
  }
  return levels;
}

extern "C" int orc_lk_track(const unsigned char* prev, const unsigned char* next, int w, int h,
                            int stride, int cn, const float* prev_pts, int n, float* next_pts,
                            unsigned char* status, float* err, int win, int max_level,
                            int max_count, double epsilon, double min_eig_thr) {
  return lk_track_core(prev, next, w, h, stride, 1, 0, 1, cn, prev_pts, n, next_pts, status, err, win, max_level, max_count, epsilon,
                       min_eig_thr);
}

// True-colour images, `bpp` (3 or 4) interleaved bytes per pixel of which the first three are tracked (the reference's BGR8;
// the channel order does not matter to the sums): calcOpticalFlowPyrLK on a CV_8UC3 pair.
extern "C" int orc_lk_track_color(const unsigned char* prev, const unsigned char* next, int w, int h, int stride, int bpp,
                                  const float* prev_pts, int n, float* next_pts, unsigned char* status, float* err, int win,
                                  int max_level, int max_count, double epsilon, double min_eig_thr) {
  return lk_track_core(prev, next, w, h, stride, bpp, 1, 3, 3, prev_pts, n, next_pts, status, err, win, max_level, max_count, epsilon,
                       min_eig_thr);
}
