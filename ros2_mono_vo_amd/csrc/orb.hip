// csrc/orb.hip — ORB detect + describe for gfx950 (replaces cv::ORB::detectAndCompute at reference
// src/feature_processor.cpp:19-23; semantics per SURVEY.md Appendix A.1 and oracle/orc_orb.cpp).
//
// Device pipeline, every launch covering all slots of the batch:
//   resize_exact_kernel   8-level pyramid, INTER_LINEAR_EXACT 8.8 x 8.8 fixed point (level l from l-1), table driven
//   fast_nms_kernel       FAST-9/16 + cornerScore + strict 3x3 NMS fused per 64x32 tile: packed 4-point pre-test,
//                         ballot-compacted candidate list in LDS, full score for candidates only; the survivors inside
//                         the edge band leave as (x, score) records per 64-pixel row segment + segment / row counts
//                         (no dense score map is written)
//   scan_rows/scan_slots  exclusive scans (rows -> levels -> slots) so candidates of the whole batch are
//                         one dense array
//   nms_rows_kernel       ORDERED (row-major) emit of the records: half a wavefront per image row, one lane per segment
//   orb_select_kernel     (orb_select.hip) KeyPointsFilter::retainBest twice per level in libstdc++'s
//                         nth_element/partition element order - that permutation IS OpenCV's key-point order -
//                         with the Harris responses of the first pass's survivors computed in between
//   blur7_kernel          7x7 Gaussian in the 8-bit fixed point ORB gets: v_dot4_u32_u8 rows, v_dot2_u32_u16 columns
//   ic_angle_kernel       intensity-centroid orientation, half a wavefront per key-point, exact int moments
//   brief_kernel          rotated BRIEF, 32 lanes per key-point (one descriptor byte per lane)
#include "mvo_internal.h"

#include <algorithm>
#include <cmath>

static const int kOrbPattern31[256 * 4] = {
#include "orb_pattern_31.inc"
};

#define ORB_EDGE 31
#define ORB_PATCH 31
#define ORB_HALF 15

// ---------------------------------------------------------------------------------------------------
// geometry (host) — OpenCV's float-scale rounding (orb.cpp getScale / detectAndCompute / computeKeyPoints)
// ---------------------------------------------------------------------------------------------------
static void orb_geometry(int w, int h, int nfeatures, int edge, OrbGeom& G) {
  G.nlevels = MVO_ORB_LEVELS;
  G.edge = edge;
  double sf = (double)1.2f;
  size_t off = 0;
  int rows = 0;
  for (int l = 0; l < MVO_ORB_LEVELS; l++) {
    float scale = (float)std::pow(sf, (double)l);
    float inv = 1.0f / scale;
    G.scale[l] = scale;
    G.w[l] = (int)std::lrintf(w * inv);
    G.h[l] = (int)std::lrintf(h * inv);
    G.pitch[l] = align_up(G.w[l], 64);
    G.off[l] = off;
    off += (size_t)G.pitch[l] * G.h[l];
    off = (off + 255) & ~(size_t)255;
    G.row0[l] = rows;
    bool too_small = (G.h[l] <= edge * 2 || G.w[l] <= edge * 2);
    rows += too_small ? 0 : G.h[l] - 2 * edge;
  }
  G.row0[MVO_ORB_LEVELS] = rows;
  G.slot_stride = off;
  float factor = (float)(1.0 / sf);
  float nd = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)MVO_ORB_LEVELS));
  int sum = 0;
  for (int l = 0; l < MVO_ORB_LEVELS - 1; l++) {
    G.quota[l] = (int)std::lrintf(nd);
    sum += G.quota[l];
    nd *= factor;
  }
  G.quota[MVO_ORB_LEVELS - 1] = std::max(nfeatures - sum, 0);
}

// ---------------------------------------------------------------------------------------------------
// resize INTER_LINEAR_EXACT (one level from the previous one), table driven
// ---------------------------------------------------------------------------------------------------
// OpenCV (resize.cpp, resize_bitExact / interpolationLinear) maps destination coordinate v to
//   fval = scale * (v + 0.5) - 0.5,  ival = floor(fval),  c1 = round((fval - ival) * 256), c0 = 256 - c1
// with scale = 1 / ((double)dst / src), clamps to the first / last source sample outside [0, src-1), accumulates
// the horizontal pass in 8.8 fixed point and rounds the vertical pass as (v + 2^15) >> 16 (rows that clamp:
// (h + 128) >> 8, which is the same expression with weights (256, 0)).  The coordinate arithmetic is per
// axis and per level only, so the host tabulates it once per geometry: entry = offset | c1 << 16 with the clamps
// folded into (offset, c1) pairs that keep offset + 1 inside the source.
static void lin_table(int src, int dst, std::vector<unsigned>& tab) {
  const double scale = 1.0 / ((double)dst / src);
  tab.resize(dst);
  for (int v = 0; v < dst; v++) {
    double fval = scale * ((double)v + 0.5) - 0.5;
    int ival = (int)std::floor(fval);
    int ofs, c1;
    if (ival >= 0 && src > 1) {
      if (ival < src - 1) { ofs = ival; c1 = (int)std::lrint((fval - (double)ival) * 256.0); }
      else { ofs = src - 2; c1 = 256; }        // right clamp: src[last] << 8
    } else { ofs = 0; c1 = 0; }                // left clamp (or a 1-sample source): src[0] << 8
    if (src == 1) { ofs = 0; c1 = 0; }
    tab[v] = (unsigned)ofs | ((unsigned)c1 << 16);
  }
}

#define RZ_W 64
#define RZ_H 32
#define RZ_SP 96   // source tile pitch: 64 * 1.2 + taps + alignment slack (host checks the geometry fits)
#define RZ_SH 42   // 32 * 1.2 + taps + slack

// first source sample of destination coordinate v (the table's offset field, recomputed so that the tile loads do not
// have to wait for a table load): same IEEE double expression as lin_table
__device__ __forceinline__ int lin_ofs(double scale, int src, int v) {
  double fval = scale * ((double)v + 0.5) - 0.5;
  int ival = d_cv_floor(fval);
  if (ival >= 0 && src > 1) return ival < src - 1 ? ival : src - 2;
  return 0;
}

struct ResizeArgs {
  const u8* pyr; size_t slot_stride, soff; int spitch, sw, sh; size_t doff; int dpitch, dw, dh; double scale_x, scale_y;
  const unsigned* xtab; const unsigned* ytab; TileGrid tg;
  const int* nact;   // optional device count of active slots (<= tg.gz); the launch is then a strided loop over the tiles
};

__device__ __forceinline__ void resize_exact_tile(const ResizeArgs& A, const TileGrid& tg, unsigned b, unsigned* s_src) {
  const u8* __restrict__ pyr = A.pyr;
  const size_t slot_stride = A.slot_stride, soff = A.soff, doff = A.doff;
  const int spitch = A.spitch, sw = A.sw, sh = A.sh, dpitch = A.dpitch, dw = A.dw, dh = A.dh;
  const double scale_x = A.scale_x, scale_y = A.scale_y;
  const unsigned* __restrict__ xtab = A.xtab;
  const unsigned* __restrict__ ytab = A.ytab;
  int bx, by, bz;
  if (!xcd_tile_b(tg, b, bx, by, bz)) return;
  const u8* sp = pyr + (size_t)bz * slot_stride + soff;
  u8* dp = const_cast<u8*>(pyr) + (size_t)bz * slot_stride + doff;
  const int x0 = bx * RZ_W, y0 = by * RZ_H, tid = threadIdx.x;
  const int x1 = min(x0 + RZ_W, dw) - 1, y1 = min(y0 + RZ_H, dh) - 1;
  const int sxa = lin_ofs(scale_x, sw, x0) & ~3;           // tile origin, dword aligned
  const int sxe = lin_ofs(scale_x, sw, x1) + 2;            // one past the last source column needed
  const int sy0 = lin_ofs(scale_y, sh, y0), sy1 = lin_ofs(scale_y, sh, y1) + 1;
  const int ndw = (sxe - sxa + 3) >> 2, nrow = sy1 - sy0 + 1;
  // 4 horizontally adjacent outputs on 2 rows (row, row + 16) per lane; table entries requested before the tile
  const int row = tid >> 4, c4 = (tid & 15) * 4;
  const int x = x0 + c4;
  const bool live = x < dw;
  uint4 xe = make_uint4(0, 0, 0, 0);
  if (live) xe = *(const uint4*)(xtab + x);               // tables are padded to a multiple of 4 entries
  unsigned ye[2];
#pragma unroll
  for (int q = 0; q < 2; q++) { int y = y0 + row + 16 * q; ye[q] = (live && y < dh) ? ytab[y] : 0u; }
  {
    // 32 lanes per tile row (24 dwords at most), 8 rows per pass: no division, 32-bit multiplies kept off the path
    // (v_mul_lo_u32 / v_mad_u64_u32 are quarter rate on CDNA)
    // rows come in as 16-byte loads, 8 lanes per row (a lane address costs the same for 4 or 16 bytes); a group that
    // would run past the source pitch falls back to dwords
    const int k = tid & 7;
    const u8* gp = sp + (size_t)__umul24(sy0 + (tid >> 3), spitch) + sxa + 16 * k;
    unsigned* lp = s_src + __umul24(tid >> 3, RZ_SP / 4) + 4 * k;
    const bool wide = sxa + 16 * k + 16 <= spitch;
    if (4 * k < ndw)
      for (int ty = tid >> 3; ty < nrow; ty += 32, gp += 32 * (size_t)spitch, lp += 32 * (RZ_SP / 4)) {
        if (wide) { const uint4 v = *(const uint4*)gp; lp[0] = v.x; lp[1] = v.y; lp[2] = v.z; lp[3] = v.w; }
        else
          for (int j = 0; j < 4 && 4 * k + j < ndw; j++) lp[j] = *(const unsigned*)(gp + 4 * j);
      }
  }
  __syncthreads();
  if (!live) return;
  const unsigned xes[4] = {xe.x, xe.y, xe.z, xe.w};
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int y = y0 + row + 16 * q;
    if (y >= dh) break;
    const unsigned cy1 = ye[q] >> 16, cy0 = 256u - cy1;
    const u8* r0 = (const u8*)s_src + __umul24((ye[q] & 0xFFFFu) - sy0, RZ_SP) - sxa;
    const u8* r1 = r0 + RZ_SP;
    unsigned out = 0;
    const int nvalid = min(4, dw - x);   // 4 except in the last lanes of the right-most tile
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (j < nvalid) {
        const unsigned ox = xes[j] & 0xFFFFu, cx1 = xes[j] >> 16, cx0 = 256u - cx1;
        unsigned h0 = __umul24(cx0, r0[ox]) + __umul24(cx1, r0[ox + 1]);   // < 2^16
        unsigned h1 = __umul24(cx0, r1[ox]) + __umul24(cx1, r1[ox + 1]);
        unsigned v = __umul24(h0, cy0) + __umul24(h1, cy1);               // weights sum to 2^16: v <= 255 * 2^16
        out |= ((v + 32768u) >> 16) << (8 * j);                           // <= 255, no clamp needed
      }
    }
    *(unsigned*)(dp + (size_t)__umul24(y, dpitch) + x) = out;  // x % 4 == 0 and pitch % 64 == 0: the padding columns take zeros
  }
}

__global__ __launch_bounds__(256) void resize_exact_kernel(ResizeArgs A) {
  __shared__ unsigned s_src[RZ_SH * RZ_SP / 4];
  TileGrid tg = A.tg;
  if (A.nact) tg.gz = min(tg.gz, max(*A.nact, 0));
  const unsigned nb = xcd_grid_blocks_dev(tg);
  for (unsigned b = blockIdx.x; b < nb; b += gridDim.x) {
    resize_exact_tile(A, tg, b, s_src);
    __syncthreads();   // the next tile reuses the LDS tile
  }
}

// ---------------------------------------------------------------------------------------------------
// FAST-9/16 + 3x3 NMS, fused: the output map holds the score of NMS survivors and 0 elsewhere
// ---------------------------------------------------------------------------------------------------
#define FT_W 64
#define FT_H 32
#define FT_PR (FT_H + 8)   // pixel tile rows: 1 (NMS halo) + 3 (circle radius) on each side
#define FT_P 76            // tile pitch in bytes; tile column c is image x0 - 4 + c, so 4-pixel groups are dword aligned
#define FT_PD (FT_P / 4)
#define FT_SR (FT_H + 2)   // score tile rows (NMS halo 1); same column indexing and pitch as the pixel tile
#define FT_NG 18           // 4-pixel groups per tile row

__device__ __forceinline__ bool has9(unsigned m) {
  unsigned m2 = m | (m << 16);
  unsigned x = m2 & (m2 >> 1);
  x &= x >> 2;
  x &= x >> 4;
  x &= m2 >> 8;
  return (x & 0xFFFFu) != 0;
}

// differences centre - circle pixel, circle offsets (dx,dy) k=0..15 as in fast_score.cpp makeOffsets(16)
__device__ __forceinline__ void fast_ring(const u8* c, int (&d)[25]) {
  const int v = c[0];
  d[0] = v - c[3 * FT_P + 0];   d[1] = v - c[3 * FT_P + 1];   d[2] = v - c[2 * FT_P + 2];
  d[3] = v - c[1 * FT_P + 3];   d[4] = v - c[3];              d[5] = v - c[-1 * FT_P + 3];
  d[6] = v - c[-2 * FT_P + 2];  d[7] = v - c[-3 * FT_P + 1];  d[8] = v - c[-3 * FT_P + 0];
  d[9] = v - c[-3 * FT_P - 1];  d[10] = v - c[-2 * FT_P - 2]; d[11] = v - c[-1 * FT_P - 3];
  d[12] = v - c[-3];            d[13] = v - c[1 * FT_P - 3];  d[14] = v - c[2 * FT_P - 2];
  d[15] = v - c[3 * FT_P - 1];
}

// FAST-9: an arc of 9 contiguous circle pixels all darker than v - t or all brighter than v + t.  The two 16-bit
// masks are gathered sign bit by sign bit: p_k < v - t  <=>  (p_k - (v - t)) < 0, p_k > v + t  <=>  ((v + t) - p_k) < 0,
// and v_alignbit_b32(acc, x, 31) = acc << 1 | sign(x) appends one bit per instruction (k = 15 first, so pixel k ends
// at bit k) - no compare / select pairs.
__device__ __forceinline__ bool fast_is_corner(const u8* c, int t) {
  const int v = c[0];
  const int lo = v - t, hi = v + t;
  // circle offsets (dx,dy) k=0..15 as in fast_score.cpp makeOffsets(16)
  const int p[16] = {c[3 * FT_P + 0],  c[3 * FT_P + 1],  c[2 * FT_P + 2],  c[1 * FT_P + 3], c[3],  c[-1 * FT_P + 3], c[-2 * FT_P + 2],
                     c[-3 * FT_P + 1], c[-3 * FT_P + 0], c[-3 * FT_P - 1], c[-2 * FT_P - 2], c[-1 * FT_P - 3], c[-3], c[1 * FT_P - 3],
                     c[2 * FT_P - 2],  c[3 * FT_P - 1]};
  unsigned mdark = 0, mbright = 0;
#pragma unroll
  for (int k = 15; k >= 0; k--) {
    mdark = __builtin_amdgcn_alignbit(mdark, (unsigned)(p[k] - lo), 31);
    mbright = __builtin_amdgcn_alignbit(mbright, (unsigned)(hi - p[k]), 31);
  }
  return has9(mdark) || has9(mbright);
}

// cornerScore<16> of a pixel that passed fast_is_corner: max(t, max_arc min d, max_arc min(-d)) - 1 over the 16 arcs
// of 9 contiguous pixels
__device__ __forceinline__ int fast_corner_score(const u8* c, int t) {
  int d[25];
  fast_ring(c, d);
#pragma unroll
  for (int k = 16; k < 25; k++) d[k] = d[k - 16];
  int mn2[24], mx2[24], mn4[22], mx4[22], mn8[18], mx8[18];
#pragma unroll
  for (int k = 0; k < 24; k++) { mn2[k] = min(d[k], d[k + 1]); mx2[k] = max(d[k], d[k + 1]); }
#pragma unroll
  for (int k = 0; k < 22; k++) { mn4[k] = min(mn2[k], mn2[k + 2]); mx4[k] = max(mx2[k], mx2[k + 2]); }
#pragma unroll
  for (int k = 0; k < 18; k++) { mn8[k] = min(mn4[k], mn4[k + 4]); mx8[k] = max(mx4[k], mx4[k + 4]); }
  int a0 = t, b0 = -t;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    a0 = max(a0, min(mn8[k], d[k + 8]));
    b0 = min(b0, max(mx8[k], d[k + 8]));
  }
  return max(a0, -b0) - 1;
}

typedef unsigned short ft_us2 __attribute__((ext_vector_type(2)));
typedef short ft_s2 __attribute__((ext_vector_type(2)));
// bytes (0,1) / (2,3) of a dword as two zero-extended 16-bit lanes (v_perm_b32)
__device__ __forceinline__ ft_us2 ft_lo(unsigned d) { return __builtin_bit_cast(ft_us2, __builtin_amdgcn_perm(0u, d, 0x0c010c00u)); }
__device__ __forceinline__ ft_us2 ft_hi(unsigned d) { return __builtin_bit_cast(ft_us2, __builtin_amdgcn_perm(0u, d, 0x0c030c02u)); }
__device__ __forceinline__ ft_us2 ft_max(ft_us2 a, ft_us2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ ft_us2 ft_min(ft_us2 a, ft_us2 b) { return __builtin_elementwise_min(a, b); }

// sign bits (bit 15 / 31) clear where a pixel passes the 4-point pre-test of FAST-9: a 9-arc on the 16-circle
// contains one end of each antipodal pair, so a corner needs (p0 or p8) and (p4 or p12) brighter than v + t, or
// both darker than v - t.  Two pixels per call, packed 16-bit arithmetic.
__device__ __forceinline__ unsigned ft_pretest2(ft_us2 v, ft_us2 p0, ft_us2 p8, ft_us2 p4, ft_us2 p12, ft_s2 t1) {
  ft_us2 M = ft_min(ft_max(p0, p8), ft_max(p4, p12));   // bright side: M > v + t
  ft_us2 m = ft_max(ft_min(p0, p8), ft_min(p4, p12));   // dark side:   m < v - t
  ft_s2 e = __builtin_bit_cast(ft_s2, M) - __builtin_bit_cast(ft_s2, v) - t1;   // >= 0  <=>  M - v >= t + 1
  ft_s2 f = __builtin_bit_cast(ft_s2, v) - __builtin_bit_cast(ft_s2, m) - t1;   // >= 0  <=>  v - m >= t + 1
  return __builtin_bit_cast(unsigned, e) & __builtin_bit_cast(unsigned, f);
}

struct FastArgs {
  const u8* pyr; u8* score;
  size_t slot_stride, off;
  int w, h, pitch, threshold;
  int lo;        // scores are only needed for x in [lo, w-lo), y in [lo, h-lo) (lo = max(3, edge-1): the NMS of the rows /
                 // columns the compaction scans needs one ring of neighbours); everything else is written as 0
  int edge;      // the compaction scans rows [edge, h-edge) x columns [edge, w-edge): survivors there are counted per row
  int* row_cnt;  // [slots][max_rows], zeroed before the launch; this level's rows start at row_first
  int max_rows, row_first;
  u8* seg_cnt;   // [slots][max_rows][seg_per_row], zeroed before the launch: survivors per 64-pixel row segment
  int seg_per_row;
  TileGrid tg;
  const int* nact;   // optional device count of active slots (see ResizeArgs)
};

struct FastLds {
  unsigned px[FT_PR * FT_PD];
  unsigned sc[FT_SR * FT_PD];
  unsigned short list[FT_SR * (FT_W + 2) + 64];
  unsigned short list2[FT_SR * (FT_W + 2) + 64];
  unsigned rowmask[FT_H][2];
  int n, n2;
};

// FAST-9/16 + cornerScore + strict 3x3 NMS for one 64x32 tile, plus the per-row survivor counts of the ordered
// compaction.  Phase 1 rejects with the 4-point pre-test, 4 pixels per lane from aligned LDS dwords (about 5 % pass);
// phase 2a runs the 16-pixel arc test on the compacted candidates and compacts the corners (about 2 % of the pixels),
// phase 2b scores those, phase 3 is the NMS of the corners (list driven, not a pass over the tile), phase 4 writes the
// survivors of each tile row in x order.
__device__ __forceinline__ void fast_nms_tile(const FastArgs& A, const TileGrid& tg, unsigned b, FastLds& L) {
  unsigned* s_px = L.px;
  unsigned* s_sc = L.sc;
  unsigned short* s_list = L.list;
  unsigned short* s_list2 = L.list2;
  unsigned (*s_rowmask)[2] = L.rowmask;
  int& s_n = L.n;
  int& s_n2 = L.n2;
  int bx, by, bz;
  if (!xcd_tile_b(tg, b, bx, by, bz)) return;
  const u8* sp = A.pyr + (size_t)bz * A.slot_stride + A.off;
  u8* dp = A.score + (size_t)bz * A.slot_stride + A.off;
  const int w = A.w, h = A.h, pitch = A.pitch, lo = A.lo;
  const int x0 = bx * FT_W, y0 = by * FT_H;
  const int tid = threadIdx.x, lane = tid & 63;
  // tiles that cannot contain a needed score
  const bool dead = x0 + FT_W + 1 <= lo || x0 - 1 >= w - lo || y0 + FT_H + 1 <= lo || y0 - 1 >= h - lo;
  if (dead) return;  // seg_cnt / row_cnt are zeroed before the launch
  if (tid == 0) { s_n = 0; s_n2 = 0; }
  if (tid < 2 * FT_H) s_rowmask[tid >> 1][tid & 1] = 0u;
  // ---- phase 0: pixel tile rows y0-4 .. y0+35, columns x0-4 .. x0+67 ----------------------------------------------------
  // 72 bytes per row as four 16-byte loads and one 8-byte load (a lane address costs the same for 4 or 16 bytes).  Border
  // tiles take the same path: rows are clamped into the image, the left-most tile starts at column 0 one dword further
  // into the LDS row, and nothing is read past the row pitch.  Halo bytes outside the image then hold arbitrary data,
  // which no needed score depends on: scores are computed for [lo, w - lo) x [lo, h - lo) with lo >= 3 = circle radius.
  if (tid < FT_PR * 5) {
    const int ty = tid / 5, k = tid - ty * 5;
    const int gy = min(max(y0 - 4 + ty, 0), h - 1);
    const int sh1 = x0 == 0 ? 1 : 0;
    const int gx = x0 - 4 + 16 * k + 4 * sh1;
    const u8* gp = sp + (size_t)__umul24(gy, pitch) + gx;
    unsigned* lp = s_px + __umul24(ty, FT_PD) + 4 * k + sh1;
    if (k < 4) { const uint4 v = *(const uint4*)gp; lp[0] = v.x; lp[1] = v.y; lp[2] = v.z; lp[3] = v.w; }
    else if (gx + 8 <= pitch) { const uint2 v = *(const uint2*)gp; lp[0] = v.x; lp[1] = v.y; }
    else { lp[0] = gx + 4 <= pitch ? *(const unsigned*)gp : 0u; lp[1] = 0u; }
  }
  for (int i = tid; i < FT_SR * FT_PD; i += 256) s_sc[i] = 0u;
  __syncthreads();
  // ---- phase 1: pre-test.  Scores are needed on rows y0-1 .. y0+32, columns x0-1 .. x0+64 (the tile plus the NMS ring),
  // clipped to [lo, w-lo) x [lo, h-lo).  The 32 x 64 core goes as 512 items of 4 pixels from aligned dwords (two full
  // passes), the 196 ring pixels one per lane; tiles whose ring lies inside the band skip every clipping test. --------
  {
    const ft_s2 t1 = {(short)(A.threshold + 1), (short)(A.threshold + 1)};
    const bool inner = x0 - 1 >= lo && x0 + FT_W + 1 <= w - lo && y0 - 1 >= lo && y0 + FT_H + 1 <= h - lo;  // block-uniform
    const unsigned long long below = (1ull << lane) - 1;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const int it = pass * 256 + tid;
      const int r = it >> 4, gi = it & 15;
      const int sy = r + 1, g = gi + 1;                 // score-tile row, dword group of the pixel tile
      const int gy = y0 + r, gx = x0 + 4 * gi;
      unsigned cmask = 0;                               // bit j: pixel j of the group is a candidate
      if (inner || (gy >= lo && gy < h - lo && gx + 3 >= lo && gx < w - lo)) {
        const unsigned* c = s_px + __umul24(sy + 3, FT_PD) + g;
        const unsigned Dl = c[-1], Dc = c[0], Dr = c[1], Du = c[-3 * FT_PD], Dd = c[3 * FT_PD];
        const unsigned P4 = __builtin_amdgcn_alignbyte(Dr, Dc, 3);    // x + 3
        const unsigned P12 = __builtin_amdgcn_alignbyte(Dc, Dl, 1);   // x - 3
        unsigned s01 = ft_pretest2(ft_lo(Dc), ft_lo(Dd), ft_lo(Du), ft_lo(P4), ft_lo(P12), t1);
        unsigned s23 = ft_pretest2(ft_hi(Dc), ft_hi(Dd), ft_hi(Du), ft_hi(P4), ft_hi(P12), t1);
        cmask = (~s01 >> 15 & 1u) | (~s01 >> 30 & 2u) | (~s23 >> 13 & 4u) | (~s23 >> 28 & 8u);
        if (!inner) {
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (gx + j < lo || gx + j >= w - lo) cmask &= ~(1u << j);
        }
      }
      // candidates of the wave -> list: per-lane counts, inclusive wave prefix by DPP adds (rows of 16, then the two
      // row broadcasts), one LDS atomic per wave and pass; the order of the list does not matter
      const int nc = __popc(cmask);
      int inc = nc;
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, true);    // row_shr:1
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, true);    // row_shr:2
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, true);    // row_shr:4
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, true);    // row_shr:8
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
      const int total = __builtin_amdgcn_readlane(inc, 63);
      if (total == 0) continue;                                               // wave-uniform
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_n, total);
      base = __builtin_amdgcn_readfirstlane(base);
      int pos = base + inc - nc;
      const unsigned code = (unsigned)(sy << 7) | (unsigned)(4 * g);
      if (cmask & 1u) s_list[pos++] = (unsigned short)(code);
      if (cmask & 2u) s_list[pos++] = (unsigned short)(code + 1);
      if (cmask & 4u) s_list[pos++] = (unsigned short)(code + 2);
      if (cmask & 8u) s_list[pos] = (unsigned short)(code + 3);
    }
    {
      // the ring: rows sy = 0 and FT_SR-1 (66 columns each), columns c = 3 and 68 of rows 1 .. FT_H
      int sy, c;
      if (tid < 66) { sy = 0; c = 3 + tid; }
      else if (tid < 132) { sy = FT_SR - 1; c = 3 + (tid - 66); }
      else if (tid < 132 + FT_H) { sy = 1 + (tid - 132); c = 3; }
      else { sy = 1 + ((tid - 132 - FT_H) & (FT_H - 1)); c = FT_W + 4; }
      const int gy = y0 - 1 + sy, gx = x0 - 4 + c;
      bool cand = false;
      if (tid < 132 + 2 * FT_H && (inner || (gy >= lo && gy < h - lo && gx >= lo && gx < w - lo))) {
        const u8* q = (const u8*)s_px + __umul24(sy + 3, FT_P) + c;
        const int v = q[0], p0 = q[3 * FT_P], p8 = q[-3 * FT_P], p4 = q[3], p12 = q[-3];
        const int M = min(max(p0, p8), max(p4, p12)), m = max(min(p0, p8), min(p4, p12));
        cand = M - v > A.threshold || v - m > A.threshold;
      }
      const unsigned long long mb = __ballot(cand);
      int base = 0;
      if (lane == 0 && mb) base = atomicAdd(&s_n, __popcll(mb));
      base = __shfl(base, 0, 64);
      if (cand) s_list[base + __popcll(mb & below)] = (unsigned short)((sy << 7) | c);
    }
  }
  __syncthreads();
  // ---- phase 2a: the 16-pixel arc test for the candidates (about 5 % of the pixels); corners are compacted again -------
  {
    const int n = s_n;
    const u8* pb = (const u8*)s_px;
    for (int k0 = 0; k0 < n; k0 += 256) {
      const int k = k0 + tid;
      int code = 0;
      bool corner = false;
      if (k < n) {
        code = s_list[k];
        corner = fast_is_corner(pb + __umul24((code >> 7) + 3, FT_P) + (code & 127), A.threshold);
      }
      const unsigned long long m = __ballot(corner);
      int base = 0;
      if (lane == 0 && m) base = atomicAdd(&s_n2, __popcll(m));
      base = __shfl(base, 0, 64);
      if (corner) s_list2[base + __popcll(m & ((1ull << lane) - 1))] = (unsigned short)code;
    }
  }
  __syncthreads();
  // ---- phase 2b: cornerScore for the corners only (about 2 % of the pixels: usually a single dense wavefront) ---------
  const int n2 = s_n2;
  {
    const u8* pb = (const u8*)s_px;
    u8* sb = (u8*)s_sc;
    for (int k = tid; k < n2; k += 256) {
      const int code = s_list2[k];
      const int sy = code >> 7, c = code & 127;
      sb[__umul24(sy, FT_P) + c] = (u8)fast_corner_score(pb + __umul24(sy + 3, FT_P) + c, A.threshold);
    }
  }
  __syncthreads();
  // ---- phase 3: strict 3x3 NMS of the corners inside the tile and the compaction band; survivors as row bit masks ------
  {
    const u8* sb = (const u8*)s_sc;
    for (int k = tid; k < n2; k += 256) {
      const int code = s_list2[k];
      const int sy = code >> 7, c = code & 127;
      const int r = sy - 1, x = c - 4;                       // tile coordinates
      const int gy = y0 + r, gx = x0 + x;
      if (r < 0 || r >= FT_H || x < 0 || x >= FT_W) continue;  // halo corners only serve as neighbours
      if (gy < A.edge || gy >= h - A.edge || gx < A.edge || gx >= w - A.edge) continue;
      const u8* q = sb + __umul24(sy, FT_P) + c;
      const int v = q[0];
      int nb = max(max((int)q[-1], (int)q[1]), max((int)q[-FT_P], (int)q[FT_P]));
      nb = max(nb, max(max((int)q[-FT_P - 1], (int)q[-FT_P + 1]), max((int)q[FT_P - 1], (int)q[FT_P + 1])));
      if (v > nb) atomicOr(&s_rowmask[r][x >> 5], 1u << (x & 31));
    }
  }
  __syncthreads();
  // ---- phase 4: per tile row, survivors in x order as (x in tile, score) records at the head of this row segment's 64
  // bytes of the score plane - strict NMS leaves at most 32 per 64 pixels - with their count in seg_cnt: the dense map is
  // never written, and the emit pass reads counts and records only.
  if (tid < FT_H) {
    const int r = tid, gy = y0 + r;
    unsigned long long m = (unsigned long long)s_rowmask[r][0] | ((unsigned long long)s_rowmask[r][1] << 32);
    const int cnt = __popcll(m);
    if (cnt) {
      const u8* sb = (const u8*)s_sc + __umul24(r + 1, FT_P) + 4;
      unsigned short* rec = (unsigned short*)(dp + (size_t)__umul24(gy, pitch) + x0);
      while (m) {
        const int x = __ffsll((long long)m) - 1;
        m &= m - 1;
        *rec++ = (unsigned short)(((unsigned)x << 8) | sb[x]);
      }
      const size_t rowi = (size_t)bz * A.max_rows + A.row_first + (gy - A.edge);
      A.seg_cnt[rowi * A.seg_per_row + bx] = (u8)cnt;
      atomicAdd(A.row_cnt + rowi, cnt);
    }
  }
}

__global__ __launch_bounds__(256) void fast_nms_kernel(FastArgs A) {
  __shared__ FastLds L;
  TileGrid tg = A.tg;
  if (A.nact) tg.gz = min(tg.gz, max(*A.nact, 0));
  const unsigned nb = xcd_grid_blocks_dev(tg);
  for (unsigned b = blockIdx.x; b < nb; b += gridDim.x) {
    fast_nms_tile(A, tg, b, L);
    __syncthreads();   // the next tile reuses the LDS tile, lists and counters
  }
}

// ---------------------------------------------------------------------------------------------------
// ordered emit of the NMS survivors (row-major per level): one wavefront per image row, one lane per 64-pixel
// segment; counts from seg_cnt, offsets from scan_rows / scan_slots, records from the head of each segment
// ---------------------------------------------------------------------------------------------------
// RPW rows per wavefront: 2 (one per 32-lane half) when a row has at most 32 segments, else 1.
template <int RPW>
__global__ __launch_bounds__(256) void nms_rows_kernel(const u8* __restrict__ score, const u8* __restrict__ seg_cnt, int seg_per_row,
                                                       OrbGeom G, const int* __restrict__ row_off, const int* __restrict__ slot_base,
                                                       int max_rows, unsigned short* __restrict__ cx, unsigned short* __restrict__ cy,
                                                       u8* __restrict__ cs, u8* __restrict__ cl, int* __restrict__ cslot, int cand_cap,
                                                       int row_blocks, int nslots, const int* __restrict__ nact) {
  constexpr int LPR = 64 / RPW;  // lanes per row
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane & (LPR - 1);
  // work item = (row block, slot); a strided loop when the slot count lives on the device
  const int nslots_act = nact ? min(max(*nact, 0), nslots) : nslots;
  const unsigned total = (unsigned)row_blocks * (unsigned)nslots_act;
  for (unsigned wi = blockIdx.x; wi < total; wi += gridDim.x) {
  const int slot = (int)(wi / (unsigned)row_blocks);
  const int rb = (int)(wi - (unsigned)slot * (unsigned)row_blocks);
  const int row = (rb * 4 + wave) * RPW + lane / LPR;
  const bool live = row < G.row0[G.nlevels];
  int l = 0;
#pragma unroll
  for (int k = 1; k < MVO_ORB_LEVELS; k++) l += (row >= G.row0[k]);
  l = live ? l : 0;
  const int pitch = G.pitch[l];
  const int y = G.edge + (row - G.row0[l]);
  const int nseg = (G.w[l] + 63) >> 6;
  const size_t rowi = (size_t)slot * max_rows + (live ? row : 0);
  const int cnt = (live && sub < nseg) ? seg_cnt[rowi * seg_per_row + sub] : 0;
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < LPR; d <<= 1) {
    int t = __shfl_up(incl, d, LPR);
    if (sub >= d) incl += t;
  }
  if (cnt == 0) continue;
  int o = slot_base[slot] + row_off[rowi] + incl - cnt;
  const unsigned short* rec = (const unsigned short*)(score + (size_t)slot * G.slot_stride + G.off[l] + (size_t)y * pitch + sub * 64);
  for (int k = 0; k < cnt; k++, o++) {
    if (o >= cand_cap) break;
    const unsigned r = rec[k];
    cx[o] = (unsigned short)(sub * 64 + (r >> 8)); cy[o] = (unsigned short)y; cs[o] = (u8)(r & 0xFFu); cl[o] = (u8)l; cslot[o] = slot;
  }
  }
}

static void nms_rows_launch(mvo_ctx* ctx, const OrbGeom& G, int nrows, int nslots, const int* d_nact = nullptr) {
  OrbState* o = ctx->orb;
  if (nrows <= 0) return;
  if (o->seg_per_row <= 32) {
    const int rb = (nrows + 7) / 8;
    const unsigned total = (unsigned)rb * nslots;
    hipLaunchKernelGGL(nms_rows_kernel<2>, dim3(d_nact ? persist_grid(total) : total), dim3(256), 0, ctx->stream, o->d_score, o->d_seg_cnt,
                       o->seg_per_row, G, o->d_row_off, o->d_slot_base, o->max_rows, o->d_cx, o->d_cy, o->d_cs, o->d_cl, o->d_cslot,
                       o->cand_cap, rb, nslots, d_nact);
  } else {
    const int rb = (nrows + 3) / 4;
    const unsigned total = (unsigned)rb * nslots;
    hipLaunchKernelGGL(nms_rows_kernel<1>, dim3(d_nact ? persist_grid(total) : total), dim3(256), 0, ctx->stream, o->d_score, o->d_seg_cnt,
                       o->seg_per_row, G, o->d_row_off, o->d_slot_base, o->max_rows, o->d_cx, o->d_cy, o->d_cs, o->d_cl, o->d_cslot,
                       o->cand_cap, rb, nslots, d_nact);
  }
}

// zero the per-row / per-segment survivor counts of the first *nact slots (16-byte stores; the ranges are rounded up, which
// only touches counts of inactive slots - or the 16 bytes of slack behind the last slot)
__global__ __launch_bounds__(256) void orb_zero_counts_kernel(uint4* __restrict__ a, size_t a_bytes_per_slot, uint4* __restrict__ b,
                                                              size_t b_bytes_per_slot, const int* __restrict__ nact, int nslots) {
  const int n = min(max(*nact, 0), nslots);
  const size_t na = ((size_t)n * a_bytes_per_slot + 15) / 16, nb = ((size_t)n * b_bytes_per_slot + 15) / 16;
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < na; i += (size_t)gridDim.x * 256) a[i] = z;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (size_t)gridDim.x * 256) b[i] = z;
}

// one block per slot: exclusive scan of the row counts (rows of all levels in order), per-level totals.
__global__ __launch_bounds__(1024) void scan_rows_kernel(const int* __restrict__ row_cnt, int* __restrict__ row_off,
                                                         OrbGeom G, int max_rows, int* __restrict__ lvl_cnt,
                                                         int* __restrict__ slot_tot, const int* __restrict__ nact) {
  __shared__ int s_w[16];
  __shared__ int s_run;
  const int slot = blockIdx.x;
  if (nact && slot >= *nact) return;   // block-uniform
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nrows = G.row0[G.nlevels];
  if (threadIdx.x == 0) s_run = 0;
  __syncthreads();
  for (int r0 = 0; r0 < nrows; r0 += 1024) {
    int r = r0 + threadIdx.x;
    int v = r < nrows ? row_cnt[(size_t)slot * max_rows + r] : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int off = s_run;
    for (int k = 0; k < wave; k++) off += s_w[k];
    if (r < nrows) row_off[(size_t)slot * max_rows + r] = off + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int k = 0; k < 16; k++) t += s_w[k];
      s_run += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) slot_tot[slot] = s_run;
  // per-level totals = off[row0[l+1]] - off[row0[l]]
  if (threadIdx.x < G.nlevels) {
    int l = threadIdx.x;
    int a = G.row0[l], b = G.row0[l + 1];
    int oa = a < nrows ? row_off[(size_t)slot * max_rows + a] : s_run;
    int ob = b < nrows ? row_off[(size_t)slot * max_rows + b] : s_run;
    if (a == b) { oa = 0; ob = 0; }
    lvl_cnt[slot * MVO_ORB_LEVELS + l] = ob - oa;
  }
}

__global__ void scan_slots_kernel(const int* __restrict__ slot_tot, int* __restrict__ slot_base, int nslots, const int* __restrict__ nact) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (nact) nslots = min(max(*nact, 0), nslots);
    int run = 0;
    for (int s = 0; s < nslots; s++) { slot_base[s] = run; run += slot_tot[s]; }
    slot_base[nslots] = run;
  }
}

// ---------------------------------------------------------------------------------------------------
// IC angle: one wavefront per selected key-point
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float d_fast_atan2(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
  const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
  const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
  const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
  const float eps = (float)2.2204460492503131e-16;
  float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// Half a wavefront per key-point, one lane per patch row dy = lane - 15.  The lane fetches its row x0-16 .. x0+19 as
// nine (misaligned) dwords, masks the bytes outside the disc (|u| <= umax[|dy|]; masks tabulated per |dy|) and gets
// the row sum and the (u + 16)-weighted row sum with two v_dot4_u32_u8 per dword.  Integer moments: any summation
// order gives OpenCV's m_10 / m_01 (orb.cpp IC_Angle).
// What rBRIEF needs of a key-point, prepared where its angle is computed: the byte offset of its centre in the blurred
// pyramid, the level's pitch and the rotation - one 32-byte record instead of the sel -> cl / cslot / kp load chain.
struct __attribute__((aligned(16))) BriefRec {
  unsigned long long off;
  int pitch;
  float a, b;   // cos, sin of the orientation, float(cos(double)) like OpenCV's computeOrbDescriptors
  int pad[3];
};

// `nsel_dev` (optional): the key-point count lives on the device (kp_base[*nact], clamped to nsel = capacity) and the
// launch is a strided loop over groups of 8 key-points.
__device__ __forceinline__ int orb_nsel(int nsel, const int* __restrict__ kp_base, const int* __restrict__ nact) {
  return kp_base ? min(max(kp_base[max(*nact, 0)], 0), nsel) : nsel;
}

__global__ __launch_bounds__(256) void ic_angle_kernel(const u8* __restrict__ pyr, OrbGeom G, const int* __restrict__ sel,
                                                       int nsel, const unsigned short* __restrict__ cx,
                                                       const unsigned short* __restrict__ cy, const u8* __restrict__ cl,
                                                       const int* __restrict__ cslot, const float* __restrict__ ch,
                                                       const unsigned* __restrict__ icmask /* [16][9] */, mvo_keypoint* __restrict__ kp,
                                                       const int* __restrict__ kp_base, const int* __restrict__ nact) {
  nsel = orb_nsel(nsel, kp_base, nact);
  const int lane = threadIdx.x & 63, sub = lane & 31;
  // Key-points are dense per slot, and a slot's pyramid (2.9 MB at 720p) fits the 4 MB L2 of an XCD: the groups of eight go
  // to the XCDs in contiguous eighths of the list (same walk as xcd_tile_b), so a patch row's 128-byte lines are fetched from
  // HBM once instead of by every XCD that happens to hold a key-point of the slot.
  const unsigned ngrp = ((unsigned)nsel + 7u) / 8u, gper = (ngrp + 7u) / 8u;
  for (unsigned b = blockIdx.x; b < 8u * gper; b += gridDim.x) {
  const unsigned grp = (b & 7u) * gper + (b >> 3);
  if (grp >= ngrp) continue;
  const int k = (int)grp * 8 + (threadIdx.x >> 5);
  const bool live = k < nsel;
  int m10 = 0, m01 = 0, ci = 0, l = 0, x0 = 0, y0 = 0;
  if (live) {
    ci = sel[k];
    l = cl[ci]; x0 = cx[ci]; y0 = cy[ci];
    if (sub < 31) {
      const int dy = sub - ORB_HALF, ady = dy < 0 ? -dy : dy;
      const int pitch = G.pitch[l];
      // bytes x0-16 .. x0+19 of the row: three aligned 16-byte loads (one lane address each on the texture path instead
      // of nine), re-phased with v_alignbyte
      const u8* base = pyr + (size_t)cslot[ci] * G.slot_stride + G.off[l] + (size_t)__mul24(y0 + dy, pitch) + x0 - 16;
      const int sh = (int)((size_t)base & 3);
      const uint4* q4 = (const uint4*)(base - sh);
      const uint4 v0 = q4[0], v1 = q4[1], v2 = q4[2];
      const unsigned w[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
      const unsigned* mk = icmask + ady * 9;
      unsigned rs = 0, ws = 0;
#pragma unroll
      for (int j = 0; j < 9; j++) {
        const unsigned d = __builtin_amdgcn_alignbyte(w[j + 1], w[j], sh) & mk[j];
        const unsigned wt = (unsigned)(4 * j) * 0x01010101u + 0x03020100u;   // weights u + 16 = byte index 4j .. 4j+3
        rs = __builtin_amdgcn_udot4(d, 0x01010101u, rs, false);
        ws = __builtin_amdgcn_udot4(d, wt, ws, false);
      }
      m10 = (int)ws - 16 * (int)rs;
      m01 = dy * (int)rs;
    }
  }
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) { m10 += __shfl_xor(m10, m, 64); m01 += __shfl_xor(m01, m, 64); }
  if (live && sub == 0) {
    float sf = G.scale[l];
    mvo_keypoint o;
    o.x = (float)x0 * sf;
    o.y = (float)y0 * sf;
    o.size = ORB_PATCH * sf;
    o.angle = d_fast_atan2((float)m01, (float)m10);
    o.response = ch[ci];
    o.octave = l;
    o.class_id = -1;
    kp[k] = o;
  }
  }
}

// ---------------------------------------------------------------------------------------------------
// 7x7 Gaussian as ORB gets it (see oracle/orc_orb.cpp header for the OpenCV dispatch argument)
// ---------------------------------------------------------------------------------------------------
#define BL_W 64
#define BL_H 32    // 32-row tiles: the tile-latency floor is ~20 % lower than with 16 rows (profiles/microbench/tile_load.hip)
#define BL_SP 72   // source tile pitch: 64 + 3 + 3 columns, 4-byte aligned origin at x0 - 4
#define BL_HP 42   // transposed row-sum pitch in u16 (38 rows + pad; 21 dwords: odd, so column reads spread over banks)
struct BlurTaps { int k[7]; };
typedef unsigned short bl_ushort2 __attribute__((ext_vector_type(2)));

// Separable 7-tap blur, all-integer: horizontal taps by v_alignbyte + 2 x v_dot4_u32_u8 per pixel (4 pixels per lane),
// row sums (<= 257*255, fit u16) stored transposed in LDS, vertical taps by 4 x v_dot2_u32_u16, (s + 2^15) >> 16.
struct BlurArgs {
  const u8* pyr; u8* out; size_t slot_stride, off; int w, h, pitch; BlurTaps T; TileGrid tg;
  const int* nact;   // optional device count of active slots (see ResizeArgs)
};

__device__ __forceinline__ void blur7_tile(const BlurArgs& A, const TileGrid& tg, unsigned b, unsigned* s_src, unsigned short* s_h) {
  const u8* __restrict__ pyr = A.pyr;
  u8* __restrict__ out = A.out;
  const size_t slot_stride = A.slot_stride, off = A.off;
  const int w = A.w, h = A.h, pitch = A.pitch;
  const BlurTaps& T = A.T;
  int bx, by, bz;
  if (!xcd_tile_b(tg, b, bx, by, bz)) return;
  const u8* sp = pyr + (size_t)bz * slot_stride + off;
  u8* dp = out + (size_t)bz * slot_stride + off;
  const int x0 = bx * BL_W, y0 = by * BL_H;
  const int tid = threadIdx.x;
  // ---- stage 1: source tile rows y0-3 .. y0+BL_H+2, bytes x0-4 .. x0+67 ------------------------------------
  // 72 bytes per row as four 16-byte loads and one 8-byte load (a lane address costs the same for 4 or 16 bytes).  Border
  // tiles: rows by reflect-101; the left-most tile starts at column 0 one dword further into the LDS row; nothing is
  // read past the row pitch; then the (at most three) reflected columns a stored output needs on either side are
  // copied inside LDS.  Columns further out feed only outputs that are not stored.
  const bool tiny = w < 8 || h < 8;
  if (!tiny) {
    if (tid < (BL_H + 6) * 5) {
      const int ty = tid / 5, k = tid - ty * 5;
      const int gy = d_reflect101(y0 - 3 + ty, h);
      const int sh1 = x0 == 0 ? 1 : 0;
      const int gx = x0 - 4 + 16 * k + 4 * sh1;
      const u8* gp = sp + (size_t)gy * pitch + gx;
      unsigned* lp = s_src + ty * (BL_SP / 4) + 4 * k + sh1;
      if (k < 4) { const uint4 v = *(const uint4*)gp; lp[0] = v.x; lp[1] = v.y; lp[2] = v.z; lp[3] = v.w; }
      else if (sh1) { lp[0] = gx + 4 <= pitch ? *(const unsigned*)gp : 0u; }   // columns 64..67; the row has 18 dwords
      else if (gx + 8 <= pitch) { const uint2 v = *(const uint2*)gp; lp[0] = v.x; lp[1] = v.y; }
      else { lp[0] = gx + 4 <= pitch ? *(const unsigned*)gp : 0u; lp[1] = 0u; }
    }
    if (x0 == 0 || x0 + 67 > w) {   // block-uniform
      __syncthreads();
      if (tid < BL_H + 6) {
        u8* rowb = (u8*)s_src + tid * BL_SP;
        if (x0 == 0) { rowb[1] = rowb[7]; rowb[2] = rowb[6]; rowb[3] = rowb[5]; }   // x = -3, -2, -1 <- 3, 2, 1
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const int x = w + j;                                     // <- w - 2 - j
          if (x <= x0 + 67 && x >= x0) rowb[x - x0 + 4] = rowb[w - 2 - j - x0 + 4];
        }
      }
    }
  } else {
    u8* sb = (u8*)s_src;
    for (int i = tid; i < (BL_H + 6) * BL_SP; i += 256) {
      int ty = i / BL_SP, tx = i - ty * BL_SP;
      int gx = d_reflect101(x0 - 4 + tx, w), gy = d_reflect101(y0 - 3 + ty, h);
      sb[i] = sp[(size_t)gy * pitch + gx];
    }
  }
  __syncthreads();
  // ---- stage 2: horizontal pass, 4 outputs per work item ------------------------------------------------------------
  const unsigned K0 = (unsigned)T.k[0] | ((unsigned)T.k[1] << 8) | ((unsigned)T.k[2] << 16) | ((unsigned)T.k[3] << 24);
  const unsigned K1 = (unsigned)T.k[4] | ((unsigned)T.k[5] << 8) | ((unsigned)T.k[6] << 16);
  for (int i = tid; i < (BL_H + 6) * 16; i += 256) {
    int ty = i >> 4, c = i & 15;
    const unsigned* r = &s_src[ty * (BL_SP / 4) + c];
    unsigned d0 = r[0], d1 = r[1], d2 = r[2];  // bytes 4c .. 4c+11 of the tile row; output j uses bytes 4c+1+j .. 4c+7+j
    unsigned lo0 = __builtin_amdgcn_alignbyte(d1, d0, 1), hi0 = __builtin_amdgcn_alignbyte(d2, d1, 1);
    unsigned lo1 = __builtin_amdgcn_alignbyte(d1, d0, 2), hi1 = __builtin_amdgcn_alignbyte(d2, d1, 2);
    unsigned lo2 = __builtin_amdgcn_alignbyte(d1, d0, 3), hi2 = __builtin_amdgcn_alignbyte(d2, d1, 3);
    unsigned v0 = __builtin_amdgcn_udot4(lo0, K0, __builtin_amdgcn_udot4(hi0, K1, 0u, false), false);
    unsigned v1 = __builtin_amdgcn_udot4(lo1, K0, __builtin_amdgcn_udot4(hi1, K1, 0u, false), false);
    unsigned v2 = __builtin_amdgcn_udot4(lo2, K0, __builtin_amdgcn_udot4(hi2, K1, 0u, false), false);
    unsigned v3 = __builtin_amdgcn_udot4(d1, K0, __builtin_amdgcn_udot4(d2, K1, 0u, false), false);
    unsigned short* hcol = &s_h[(4 * c) * BL_HP + ty];
    hcol[0] = (unsigned short)v0; hcol[BL_HP] = (unsigned short)v1; hcol[2 * BL_HP] = (unsigned short)v2; hcol[3 * BL_HP] = (unsigned short)v3;
  }
  __syncthreads();
  // ---- stage 3: vertical pass, lane = column, 2 x 4 output rows per lane ------------------------------------------------
  const unsigned T01 = (unsigned)T.k[0] | ((unsigned)T.k[1] << 16), T23 = (unsigned)T.k[2] | ((unsigned)T.k[3] << 16);
  const unsigned T45 = (unsigned)T.k[4] | ((unsigned)T.k[5] << 16), T6 = (unsigned)T.k[6];
#pragma unroll
  for (int g = 0; g < BL_H / 16; g++) {
    const int x = tid & 63, r0 = (tid >> 6) * (BL_H / 4) + 4 * g;
    const unsigned* hc = (const unsigned*)&s_h[x * BL_HP + r0];  // rows r0 .. r0+9 (r0 % 4 == 0 -> dword aligned)
    unsigned e0 = hc[0], e1 = hc[1], e2 = hc[2], e3 = hc[3], e4 = hc[4];
    // odd rows start one u16 later: re-pair with a 2-byte funnel shift
    unsigned o0 = __builtin_amdgcn_alignbyte(e1, e0, 2), o1 = __builtin_amdgcn_alignbyte(e2, e1, 2), o2 = __builtin_amdgcn_alignbyte(e3, e2, 2),
             o3 = __builtin_amdgcn_alignbyte(e4, e3, 2);
#define BL_D2(a, b, c) __builtin_amdgcn_udot2(__builtin_bit_cast(bl_ushort2, (unsigned)(a)), __builtin_bit_cast(bl_ushort2, (unsigned)(b)), (c), false)
    unsigned s0 = BL_D2(e0, T01, BL_D2(e1, T23, BL_D2(e2, T45, BL_D2(e3, T6, 0u))));
    unsigned s1 = BL_D2(o0, T01, BL_D2(o1, T23, BL_D2(o2, T45, BL_D2(o3, T6, 0u))));
    unsigned s2 = BL_D2(e1, T01, BL_D2(e2, T23, BL_D2(e3, T45, BL_D2(e4, T6, 0u))));
    unsigned s3 = BL_D2(o1, T01, BL_D2(o2, T23, BL_D2(o3, T45, BL_D2(e4 >> 16, T6, 0u))));
#undef BL_D2
    const int gx = x0 + x;
    if (gx < w) {
      unsigned sv[4] = {s0, s1, s2, s3};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        int gy = y0 + r0 + j;
        if (gy < h) dp[(size_t)__umul24(gy, pitch) + gx] = (u8)min(255u, (sv[j] + 32768u) >> 16);
      }
    }
  }
}

__global__ __launch_bounds__(256) void blur7_kernel(BlurArgs A) {
  __shared__ unsigned s_src[(BL_H + 6) * BL_SP / 4];
  __shared__ unsigned short s_h[BL_W * BL_HP];
  TileGrid tg = A.tg;
  if (A.nact) tg.gz = min(tg.gz, max(*A.nact, 0));
  const unsigned nb = xcd_grid_blocks_dev(tg);
  for (unsigned b = blockIdx.x; b < nb; b += gridDim.x) {
    blur7_tile(A, tg, b, s_src, s_h);
    __syncthreads();   // the next tile reuses both LDS tiles
  }
}

// ---------------------------------------------------------------------------------------------------
// rotated BRIEF: 2 key-points per wavefront, one descriptor byte per lane
// ---------------------------------------------------------------------------------------------------
// One lane per key-point: the records rBRIEF works from (64 sin / cos pairs per wavefront instead of one per half-wave).
__global__ __launch_bounds__(256) void brief_rec_kernel(OrbGeom G, const int* __restrict__ sel, int nsel, const u8* __restrict__ cl,
                                                        const int* __restrict__ cslot, const mvo_keypoint* __restrict__ kp,
                                                        BriefRec* __restrict__ rec, const int* __restrict__ kp_base,
                                                        const int* __restrict__ nact) {
  nsel = orb_nsel(nsel, kp_base, nact);
  for (int k = blockIdx.x * 256 + threadIdx.x; k < nsel; k += gridDim.x * 256) {
  const int ci = sel[k], l = cl[ci];
  const mvo_keypoint kpt = kp[k];
  // computeOrbDescriptors: centre = cvRound(kpt.pt * (1 / scale)) in the level image, angle in radians as float
  const float scale = 1.f / G.scale[l];
  const int ccx = d_cv_round(kpt.x * scale), ccy = d_cv_round(kpt.y * scale);
  float angle = kpt.angle;
  angle *= (float)(M_PI / 180.f);
  BriefRec r;
  r.off = (unsigned long long)((size_t)cslot[ci] * G.slot_stride + G.off[l] + (size_t)ccy * G.pitch[l] + ccx);
  r.pitch = G.pitch[l];
  r.a = (float)cos((double)angle);
  r.b = (float)sin((double)angle);
  r.pad[0] = r.pad[1] = r.pad[2] = 0;
  rec[k] = r;
  }
}

// 32 lanes per key-point, one descriptor byte per lane.  The rotated pattern stays within +-19 pixels of the centre
// (|p| <= 13 * sqrt(2)), so the 39 x 48-byte neighbourhood is staged in LDS with 16-byte loads first: the texture path
// serves about one lane address per cycle and CU, and 512 byte gathers per key-point made the kernel bound by it
// (117 wide loads now); the 16 taps of a lane are LDS byte reads.
#define BR_R 19
#define BR_ROWS (2 * BR_R + 1)
#define BR_LP 64   // LDS row pitch in bytes: 48 staged + pad, keeps the 16-byte stores aligned
__global__ __launch_bounds__(256) void brief_kernel(const u8* __restrict__ blur, const BriefRec* __restrict__ rec, int nsel,
                                                    const char4* __restrict__ pattern, u8* __restrict__ desc,
                                                    const int* __restrict__ kp_base, const int* __restrict__ nact) {
  __shared__ char4 s_pat[256];
  __shared__ uint4 s_patch[8][BR_ROWS * BR_LP / 16];
  nsel = orb_nsel(nsel, kp_base, nact);
  s_pat[threadIdx.x] = pattern[threadIdx.x];
  const int h = threadIdx.x >> 5, byte = threadIdx.x & 31;
  const unsigned ngrp = ((unsigned)nsel + 7u) / 8u, gper = (ngrp + 7u) / 8u;   // XCD-contiguous walk: see ic_angle_kernel
  for (unsigned b = blockIdx.x; b < 8u * gper; b += gridDim.x) {                // block-uniform trip count
  const unsigned grp = (b & 7u) * gper + (b >> 3);
  if (grp >= ngrp) continue;                                                    // block-uniform
  const int k = (int)grp * 8 + h;
  const bool live = k < nsel;
  BriefRec r;
  r.off = 0; r.pitch = 0; r.a = 0.f; r.b = 0.f;
  int shift = 0;
  if (live) {
    r = rec[k];
    const u8* base = blur + r.off - (size_t)BR_R * r.pitch - BR_R;   // key-points keep 31 px from the border: in range
    shift = (int)((size_t)base & 3);
    const u8* ab = base - shift;
    for (int i = byte; i < BR_ROWS * 3; i += 32) {
      const int row = i / 3, q = i - row * 3;
      s_patch[h][row * (BR_LP / 16) + q] = *(const uint4*)(ab + (size_t)row * r.pitch + 16 * q);
    }
  }
  __syncthreads();
  if (live) {
  const float a = r.a, b = r.b;
  const u8* c = (const u8*)s_patch[h] + BR_R * BR_LP + BR_R + shift;
  int val = 0;
#pragma unroll
  for (int t = 0; t < 8; t++) {
    char4 p = s_pat[byte * 8 + t];
    float x0 = p.x * a - p.y * b, y0 = p.x * b + p.y * a;
    float x1 = p.z * a - p.w * b, y1 = p.z * b + p.w * a;
    int t0 = c[d_cv_round(y0) * BR_LP + d_cv_round(x0)];
    int t1 = c[d_cv_round(y1) * BR_LP + d_cv_round(x1)];
    val |= (t0 < t1) << t;
  }
  desc[(size_t)k * 32 + byte] = (u8)val;
  }
  __syncthreads();   // the next group of key-points reuses the patches
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
int orb_state_create(mvo_ctx* ctx) {
  OrbState* o = new OrbState();
  ctx->orb = o;
  OrbGeom G;
  orb_geometry(ctx->maxw, ctx->maxh, ctx->cfg.nfeatures, 3, G);  // edge 3 = capacity for mvo_fast9_nms too
  o->slot_bytes = G.slot_stride;
  o->max_rows = G.row0[MVO_ORB_LEVELS];
  size_t tot = o->slot_bytes * ctx->B;
  MVO_HIP(hipMalloc(&o->d_pyr, tot));
  MVO_HIP(hipMalloc(&o->d_score, tot));
  MVO_HIP(hipMalloc(&o->d_blur, tot));
  MVO_HIP(hipMalloc(&o->d_row_cnt, (size_t)ctx->B * o->max_rows * sizeof(int) + 16));
  o->seg_per_row = (ctx->maxw + 63) / 64;
  if (o->seg_per_row > 64) { ctx->set_error("max_width above 4096 is not supported by the ORB emit pass"); return MVO_E_ARG; }
  MVO_HIP(hipMalloc(&o->d_seg_cnt, (size_t)ctx->B * o->max_rows * o->seg_per_row + 16));
  MVO_HIP(hipMalloc(&o->d_row_off, (size_t)ctx->B * o->max_rows * sizeof(int)));
  MVO_HIP(hipMalloc(&o->d_lvl_cnt, (size_t)ctx->B * MVO_ORB_LEVELS * sizeof(int)));
  MVO_HIP(hipMalloc(&o->d_slot_tot, (size_t)ctx->B * sizeof(int)));
  MVO_HIP(hipMalloc(&o->d_slot_base, (size_t)(ctx->B + 1) * sizeof(int)));
  // candidates: NMS leaves at most one survivor per 2x2 block; budget ~1/24 of the pyramid pixels.
  size_t pix = 0;
  for (int l = 0; l < MVO_ORB_LEVELS; l++) pix += (size_t)G.w[l] * G.h[l];
  size_t cap = std::max<size_t>(pix / 24, 16384) * ctx->B;
  if (cap > (size_t)1 << 28) cap = (size_t)1 << 28;
  o->cand_cap = (int)cap;
  MVO_HIP(hipMalloc(&o->d_cx, cap * sizeof(unsigned short)));
  MVO_HIP(hipMalloc(&o->d_cy, cap * sizeof(unsigned short)));
  MVO_HIP(hipMalloc(&o->d_cs, cap));
  MVO_HIP(hipMalloc(&o->d_cl, cap));
  MVO_HIP(hipMalloc(&o->d_cslot, cap * sizeof(int)));
  MVO_HIP(hipMalloc(&o->d_ch, cap * sizeof(float)));
  o->rtab_cap = (size_t)(ctx->maxw + ctx->maxh + 72) * MVO_ORB_LEVELS;
  MVO_HIP(hipMalloc(&o->d_rtab, o->rtab_cap * sizeof(unsigned)));
  MVO_HIP(hipMalloc(&o->d_wk, cap * sizeof(uint2)));
  MVO_HIP(hipMalloc(&o->d_stl, cap * sizeof(uint2)));
  MVO_HIP(hipMalloc(&o->d_str, cap * sizeof(uint2)));
  MVO_HIP(hipMalloc(&o->d_kept, (size_t)ctx->B * MVO_ORB_LEVELS * sizeof(int)));
  MVO_HIP(hipMalloc(&o->d_kp_base, (size_t)(ctx->B + 1) * sizeof(int)));
  o->kp_cap = ctx->maxpts * ctx->B;
  MVO_HIP(hipMalloc(&o->d_sel, (size_t)o->kp_cap * sizeof(int)));
  MVO_HIP(hipMalloc(&o->d_kp, (size_t)o->kp_cap * sizeof(mvo_keypoint)));
  MVO_HIP(hipMalloc(&o->d_brec, (size_t)o->kp_cap * sizeof(BriefRec)));
  MVO_HIP(hipMalloc(&o->d_desc, (size_t)o->kp_cap * 32));
  MVO_HIP(hipMalloc(&o->d_pattern, 256 * sizeof(char4)));
  {
    const int* src = ctx->cfg.orb_pattern ? ctx->cfg.orb_pattern : kOrbPattern31;
    char4 pat[256];
    for (int i = 0; i < 256; i++) pat[i] = make_char4((char)src[4 * i], (char)src[4 * i + 1], (char)src[4 * i + 2], (char)src[4 * i + 3]);
    MVO_HIP(hipMemcpy(o->d_pattern, pat, sizeof(pat), hipMemcpyHostToDevice));
    // orb.cpp computeKeyPoints: u_max of the half-patch-15 disc
    int umax[32] = {0};
    const int half = ORB_HALF;
    int v, v0, vmax = (int)std::floor(half * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(half * std::sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v) umax[v] = (int)std::lrint(std::sqrt((double)half * half - v * v));
    for (v = half, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
    // IC-angle row masks: row |dy| keeps bytes u + 16 of the 36-byte window x0-16 .. x0+19 with |u| <= umax[|dy|]
    unsigned icm[16 * 9];
    for (int v = 0; v < 16; v++)
      for (int j = 0; j < 9; j++) {
        unsigned m = 0;
        for (int b = 0; b < 4; b++) {
          int u = 4 * j + b - 16;
          if (u >= -umax[v] && u <= umax[v]) m |= 0xFFu << (8 * b);
        }
        icm[v * 9 + j] = m;
      }
    MVO_HIP(hipMalloc(&o->d_icmask, sizeof(icm)));
    MVO_HIP(hipMemcpy(o->d_icmask, icm, sizeof(icm), hipMemcpyHostToDevice));
  }
  MVO_HIP(hipHostMalloc(&o->h_counts, (size_t)(ctx->B * (MVO_ORB_LEVELS + 2) + 2) * sizeof(int), hipHostMallocDefault));
  MVO_HIP(hipHostMalloc(&o->h_kp, (size_t)o->kp_cap * sizeof(mvo_keypoint), hipHostMallocDefault));
  MVO_HIP(hipHostMalloc(&o->h_desc, (size_t)o->kp_cap * 32, hipHostMallocDefault));
  MVO_HIP(hipEventCreateWithFlags(&o->ev_counts, hipEventDisableTiming));
  MVO_HIP(hipEventCreateWithFlags(&o->ev_cand, hipEventDisableTiming));
  return MVO_OK;
}

void orb_state_destroy(mvo_ctx* ctx) {
  OrbState* o = ctx->orb;
  if (!o) return;
  void* dev[] = {o->d_pyr, o->d_score, o->d_blur, o->d_seg_cnt, o->d_row_cnt, o->d_row_off, o->d_lvl_cnt, o->d_slot_tot,
                 o->d_slot_base, o->d_cx, o->d_cy, o->d_cs, o->d_cl, o->d_cslot, o->d_ch, o->d_sel, o->d_kp, o->d_brec,
                 o->d_desc, o->d_pattern, o->d_wk, o->d_stl, o->d_str, o->d_kept, o->d_kp_base, o->d_rtab, o->d_icmask};
  for (void* p : dev) (void)hipFree(p);
  void* hst[] = {o->h_counts, o->h_kp, o->h_desc};
  for (void* p : hst)
    if (p) (void)hipHostFree(p);
  if (o->ev_counts) (void)hipEventDestroy(o->ev_counts);
  if (o->ev_cand) (void)hipEventDestroy(o->ev_cand);
  delete o;
  ctx->orb = nullptr;
}

static void fast_nms_launch(mvo_ctx* ctx, const OrbGeom& G, int l, int nslots, int threshold, int lo, const int* d_nact = nullptr) {
  OrbState* o = ctx->orb;
  FastArgs A;
  A.nact = d_nact;
  A.pyr = o->d_pyr; A.score = o->d_score; A.slot_stride = G.slot_stride; A.off = G.off[l];
  A.w = G.w[l]; A.h = G.h[l]; A.pitch = G.pitch[l]; A.threshold = threshold; A.lo = lo;
  // a level with no compaction rows (smaller than the edge band) must not count anything
  A.edge = G.row0[l + 1] > G.row0[l] ? G.edge : (1 << 20);
  A.row_cnt = o->d_row_cnt; A.max_rows = o->max_rows; A.row_first = G.row0[l];
  A.seg_cnt = o->d_seg_cnt; A.seg_per_row = o->seg_per_row;
  A.tg = TileGrid{(G.w[l] + FT_W - 1) / FT_W, (G.h[l] + FT_H - 1) / FT_H, nslots};
  hipLaunchKernelGGL(fast_nms_kernel, dim3(d_nact ? persist_grid(xcd_grid_blocks(A.tg)) : xcd_grid_blocks(A.tg)), dim3(256), 0, ctx->stream, A);
}

// (Re)build the per-level resize tables when the frame geometry changes.
static int orb_resize_tables(mvo_ctx* ctx, const OrbGeom& G) {
  OrbState* o = ctx->orb;
  if (o->rtab_w == G.w[0] && o->rtab_h == G.h[0]) return MVO_OK;
  std::vector<unsigned> all, t;
  for (int l = 1; l < G.nlevels; l++) {
    lin_table(G.w[l - 1], G.w[l], t);
    while (all.size() & 3) all.push_back(0);  // 16-byte aligned: the kernel fetches 4 entries per load
    o->rtab_x[l] = (int)all.size();
    all.insert(all.end(), t.begin(), t.end());
    while (all.size() & 3) all.push_back(0);
    // a 64-wide destination tile must fit the LDS source tile
    for (int x0 = 0; x0 < G.w[l]; x0 += RZ_W) {
      int x1 = std::min(x0 + RZ_W, G.w[l]) - 1;
      int span = (int)(t[x1] & 0xFFFFu) + 2 - ((int)(t[x0] & 0xFFFFu) & ~3);
      if (span > RZ_SP) { ctx->set_error("ORB resize: source tile wider than the LDS tile"); return MVO_E_ARG; }
    }
    lin_table(G.h[l - 1], G.h[l], t);
    o->rtab_y[l] = (int)all.size();
    all.insert(all.end(), t.begin(), t.end());
    for (int y0 = 0; y0 < G.h[l]; y0 += RZ_H) {
      int y1 = std::min(y0 + RZ_H, G.h[l]) - 1;
      int span = (int)(t[y1] & 0xFFFFu) + 2 - (int)(t[y0] & 0xFFFFu);
      if (span > RZ_SH) { ctx->set_error("ORB resize: source tile taller than the LDS tile"); return MVO_E_ARG; }
    }
  }
  if (all.size() > o->rtab_cap) { ctx->set_error("ORB resize tables exceed their capacity"); return MVO_E_CAPACITY; }
  // the previous tables may still be in use by queued kernels
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  MVO_HIP(hipMemcpy(o->d_rtab, all.data(), all.size() * sizeof(unsigned), hipMemcpyHostToDevice));
  o->rtab_w = G.w[0]; o->rtab_h = G.h[0];
  return MVO_OK;
}

// Stage 1 (device): pyramid (level 0 must already be resident in d_pyr), FAST, NMS, compaction, Harris.
// `d_nact` (optional): device count of active slots, <= nslots; every launch is then sized for nslots on the host and
// reads the real count on the device (no host wait anywhere).
static int orb_detect_device(mvo_ctx* ctx, const OrbGeom& G, int nslots, hipEvent_t before_fast, const int* d_nact = nullptr) {
  OrbState* o = ctx->orb;
  hipStream_t st = ctx->stream;
  int rc = orb_resize_tables(ctx, G);
  if (rc) return rc;
  for (int l = 1; l < G.nlevels; l++) {
    ResizeArgs R;
    R.pyr = o->d_pyr; R.slot_stride = G.slot_stride; R.soff = G.off[l - 1]; R.spitch = G.pitch[l - 1]; R.sw = G.w[l - 1]; R.sh = G.h[l - 1];
    R.doff = G.off[l]; R.dpitch = G.pitch[l]; R.dw = G.w[l]; R.dh = G.h[l];
    R.scale_x = 1.0 / ((double)G.w[l] / G.w[l - 1]); R.scale_y = 1.0 / ((double)G.h[l] / G.h[l - 1]);
    R.xtab = o->d_rtab + o->rtab_x[l]; R.ytab = o->d_rtab + o->rtab_y[l];
    R.tg = TileGrid{(G.w[l] + RZ_W - 1) / RZ_W, (G.h[l] + RZ_H - 1) / RZ_H, nslots};
    R.nact = d_nact;
    hipLaunchKernelGGL(resize_exact_kernel, dim3(d_nact ? persist_grid(xcd_grid_blocks(R.tg)) : xcd_grid_blocks(R.tg)), dim3(256), 0, st, R);
  }
  // The caller may hold the wide FAST kernels back until some other stream's work is through (the pipeline's LK kernel
  // heads its critical chain and would otherwise share the CUs with FAST half and half).
  if (before_fast) MVO_HIP(hipStreamWaitEvent(st, before_fast, 0));
  // per-row survivor counts are accumulated by the FAST/NMS kernel itself (phase 3)
  if (d_nact) {
    hipLaunchKernelGGL(orb_zero_counts_kernel, dim3(256), dim3(256), 0, st, (uint4*)o->d_row_cnt, (size_t)o->max_rows * sizeof(int),
                       (uint4*)o->d_seg_cnt, (size_t)o->max_rows * o->seg_per_row, d_nact, nslots);
  } else {
    MVO_HIP(hipMemsetAsync(o->d_row_cnt, 0, (size_t)nslots * o->max_rows * sizeof(int), st));
    MVO_HIP(hipMemsetAsync(o->d_seg_cnt, 0, (size_t)nslots * o->max_rows * o->seg_per_row, st));
  }
  for (int l = 0; l < G.nlevels; l++) fast_nms_launch(ctx, G, l, nslots, ctx->cfg.fast_threshold, std::max(3, G.edge - 1), d_nact);
  int nrows = G.row0[G.nlevels];
  hipLaunchKernelGGL(scan_rows_kernel, dim3(nslots), dim3(1024), 0, st, o->d_row_cnt, o->d_row_off, G, o->max_rows,
                     o->d_lvl_cnt, o->d_slot_tot, d_nact);
  hipLaunchKernelGGL(scan_slots_kernel, dim3(1), dim3(64), 0, st, o->d_slot_tot, o->d_slot_base, nslots, d_nact);
  nms_rows_launch(ctx, G, nrows, nslots, d_nact);
  return MVO_OK;
}

static void orb_geom_for(mvo_ctx* ctx, int w, int h, OrbGeom& G) {
  orb_geometry(w, h, ctx->cfg.nfeatures, ORB_EDGE, G);
  G.slot_stride = ctx->orb->slot_bytes;
}

// Phase 1.  Level 0 of each slot must be resident in orb->d_pyr.  h_counts: [B][8] lvl counts, then [B+1] slot bases.
int orb_detect_enqueue(mvo_ctx* ctx, int w, int h, int nslots, hipEvent_t before_fast) {
  OrbState* o = ctx->orb;
  OrbGeom G;
  orb_geom_for(ctx, w, h, G);
  int rc;
  { ProfScope ps(ctx, "orb_detect"); rc = orb_detect_device(ctx, G, nslots, before_fast); }
  if (rc) return rc;
  MVO_HIP(hipMemcpyAsync(o->h_counts, o->d_lvl_cnt, (size_t)nslots * MVO_ORB_LEVELS * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipMemcpyAsync(o->h_counts + (size_t)nslots * MVO_ORB_LEVELS, o->d_slot_base, (size_t)(nslots + 1) * sizeof(int),
                         hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipEventRecord(o->ev_counts, ctx->stream));
  return MVO_OK;
}

static void orb_blur_enqueue(mvo_ctx* ctx, const OrbGeom& G, int nslots, const int* d_nact = nullptr) {
  OrbState* o = ctx->orb;
  BlurArgs A;
  static const int k0[7] = {18, 34, 49, 55, 49, 34, 18};
  static const int k1[7] = {18, 34, 48, 56, 48, 34, 18};
  for (int i = 0; i < 7; i++) A.T.k[i] = ctx->cfg.orb_blur_mode ? k1[i] : k0[i];
  A.pyr = o->d_pyr; A.out = o->d_blur; A.slot_stride = G.slot_stride; A.nact = d_nact;
  for (int l = 0; l < G.nlevels; l++) {
    A.off = G.off[l]; A.w = G.w[l]; A.h = G.h[l]; A.pitch = G.pitch[l];
    A.tg = TileGrid{(G.w[l] + BL_W - 1) / BL_W, (G.h[l] + BL_H - 1) / BL_H, nslots};
    hipLaunchKernelGGL(blur7_kernel, dim3(d_nact ? persist_grid(xcd_grid_blocks(A.tg)) : xcd_grid_blocks(A.tg)), dim3(256), 0, ctx->stream, A);
  }
}

// Phase 2.  kp_base[s]..kp_base[s+1] is slot s's range in the dense selection (d_sel).
int orb_select(mvo_ctx* ctx, int w, int h, int nslots, bool describe, std::vector<int>& kp_base) {
  OrbState* o = ctx->orb;
  hipStream_t st = ctx->stream;
  OrbGeom G;
  orb_geom_for(ctx, w, h, G);
  const int* sbase = o->h_counts + (size_t)nslots * MVO_ORB_LEVELS;
  kp_base.assign(nslots + 1, 0);
  int* h_kpb = o->h_counts + (size_t)nslots * MVO_ORB_LEVELS + nslots + 1;
  {
    // OpenCV's two retainBest passes per level (2*quota by FAST score, quota by Harris), in libstdc++'s element order;
    // the Harris responses of the first pass's survivors are computed in between, inside the same kernel
    ProfScope ps(ctx, "orb_select");
    int rc = orb_select_device(ctx, G, nslots, nullptr);
    if (rc) return rc;
  }
  MVO_HIP(hipMemcpyAsync(h_kpb, o->d_kp_base, (size_t)(nslots + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
  MVO_HIP(hipEventRecord(o->ev_cand, st));
  // the blurred pyramid does not depend on the selection: queue it before waiting for the counts
  if (describe) { ProfScope ps(ctx, "orb_blur"); orb_blur_enqueue(ctx, G, nslots); }
  MVO_HIP(hipEventSynchronize(o->ev_cand));  // also covers the candidate counts queued by orb_detect_enqueue
  if (sbase[nslots] > o->cand_cap) { ctx->set_error("ORB candidate capacity exceeded"); return MVO_E_CAPACITY; }
  for (int s = 0; s <= nslots; s++) kp_base[s] = h_kpb[s];
  if (kp_base[nslots] > o->kp_cap) { ctx->set_error("ORB key-point capacity exceeded"); return MVO_E_CAPACITY; }
  return MVO_OK;
}

// Phase 3.  Outputs stay in d_kp / d_desc (dense over slots); `to_host` mirrors them into h_kp / h_desc and waits.
int orb_describe_enqueue(mvo_ctx* ctx, int w, int h, int nslots, bool describe, bool to_host, const std::vector<int>& kp_base) {
  OrbState* o = ctx->orb;
  hipStream_t st = ctx->stream;
  OrbGeom G;
  orb_geom_for(ctx, w, h, G);
  const int nsel = kp_base.empty() ? 0 : kp_base[nslots];
  if (nsel == 0) return MVO_OK;
  {
    ProfScope ps(ctx, "orb_describe");
    const int* nul = nullptr;
    hipLaunchKernelGGL(ic_angle_kernel, dim3(8 * (((nsel + 7) / 8 + 7) / 8)), dim3(256), 0, st, o->d_pyr, G, o->d_sel, nsel, o->d_cx, o->d_cy,
                       o->d_cl, o->d_cslot, o->d_ch, o->d_icmask, o->d_kp, nul, nul);
    if (describe) {
      hipLaunchKernelGGL(brief_rec_kernel, dim3((nsel + 255) / 256), dim3(256), 0, st, G, o->d_sel, nsel, o->d_cl, o->d_cslot, o->d_kp,
                         (BriefRec*)o->d_brec, nul, nul);
      hipLaunchKernelGGL(brief_kernel, dim3(8 * (((nsel + 7) / 8 + 7) / 8)), dim3(256), 0, st, o->d_blur,
                         (const BriefRec*)o->d_brec, nsel, o->d_pattern, o->d_desc, nul, nul);
    }
  }
  if (to_host) {
    if (describe) MVO_HIP(hipMemcpyAsync(o->h_desc, o->d_desc, (size_t)nsel * 32, hipMemcpyDeviceToHost, st));
    MVO_HIP(hipMemcpyAsync(o->h_kp, o->d_kp, (size_t)nsel * sizeof(mvo_keypoint), hipMemcpyDeviceToHost, st));
    MVO_HIP(hipStreamSynchronize(st));
  }
  return MVO_OK;
}

// Device-driven detect + describe of the first *d_nact slots of the ORB pyramid (level 0 resident), for the frame-batch
// tracker: every launch is sized for `max_slots` on the host and reads the real counts on the device, nothing waits for
// the host.  Outputs: d_kp / d_desc (dense), d_kp_base[0 .. *d_nact].  Capacity overruns are clamped on the device; the
// caller checks d_slot_base[*d_nact] / d_kp_base[*d_nact] against cand_cap / kp_cap where it reports results.
int orb_run_device(mvo_ctx* ctx, int w, int h, int max_slots, const int* d_nact) {
  OrbState* o = ctx->orb;
  hipStream_t st = ctx->stream;
  OrbGeom G;
  orb_geom_for(ctx, w, h, G);
  int rc;
  { ProfScope ps(ctx, "orb_detect"); rc = orb_detect_device(ctx, G, max_slots, nullptr, d_nact); }
  if (rc) return rc;
  { ProfScope ps(ctx, "orb_select"); if ((rc = orb_select_device(ctx, G, max_slots, d_nact))) return rc; }
  { ProfScope ps(ctx, "orb_blur"); orb_blur_enqueue(ctx, G, max_slots, d_nact); }
  {
    ProfScope ps(ctx, "orb_describe");
    const int cap = o->kp_cap;
    hipLaunchKernelGGL(ic_angle_kernel, dim3(persist_grid(8 * (((cap + 7) / 8 + 7) / 8))), dim3(256), 0, st, o->d_pyr, G, o->d_sel, cap, o->d_cx, o->d_cy,
                       o->d_cl, o->d_cslot, o->d_ch, o->d_icmask, o->d_kp, (const int*)o->d_kp_base, d_nact);
    hipLaunchKernelGGL(brief_rec_kernel, dim3(persist_grid((cap + 255) / 256)), dim3(256), 0, st, G, o->d_sel, cap, o->d_cl, o->d_cslot,
                       o->d_kp, (BriefRec*)o->d_brec, (const int*)o->d_kp_base, d_nact);
    hipLaunchKernelGGL(brief_kernel, dim3(persist_grid(8 * (((cap + 7) / 8 + 7) / 8))), dim3(256), 0, st, o->d_blur, (const BriefRec*)o->d_brec, cap,
                       o->d_pattern, o->d_desc, (const int*)o->d_kp_base, d_nact);
  }
  return MVO_OK;
}

// Full batched detect (+ optional describe); outputs also land in h_kp / h_desc.
int orb_run(mvo_ctx* ctx, int w, int h, int nslots, bool describe, std::vector<int>& kp_base) {
  int rc;
  if ((rc = orb_detect_enqueue(ctx, w, h, nslots))) return rc;
  if ((rc = orb_select(ctx, w, h, nslots, describe, kp_base))) return rc;
  return orb_describe_enqueue(ctx, w, h, nslots, describe, true, kp_base);
}

static int orb_upload(mvo_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels, int slot) {
  OrbState* o = ctx->orb;
  return upload_gray(ctx, img, w, h, stride, channels, o->d_pyr + (size_t)slot * o->slot_bytes, align_up(w, 64), slot);
}

static int orb_api(mvo_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels, mvo_keypoint* kps,
                   uint8_t* desc, int cap, int* n, bool describe) {
  if (!ctx || !img || !n || w < 1 || h < 1 || w > ctx->maxw || h > ctx->maxh) return MVO_E_ARG;
  *n = 0;
  int rc = orb_upload(ctx, img, w, h, stride, channels, 0);
  if (rc) return rc;
  std::vector<int> base;
  if ((rc = orb_run(ctx, w, h, 1, describe, base))) return rc;
  int cnt = base[1];
  *n = cnt;
  int m = cnt < cap ? cnt : cap;
  if (m > 0) {
    if (kps) memcpy(kps, ctx->orb->h_kp, (size_t)m * sizeof(mvo_keypoint));
    if (describe && desc) memcpy(desc, ctx->orb->h_desc, (size_t)m * 32);
  }
  return cnt > cap ? MVO_E_CAPACITY : MVO_OK;
}

extern "C" int mvo_orb_detect_and_compute(mvo_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels,
                                          mvo_keypoint* kps, uint8_t* desc, int cap, int* n) {
  return orb_api(ctx, img, w, h, stride, channels, kps, desc, cap, n, true);
}

extern "C" int mvo_orb_detect(mvo_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels,
                              mvo_keypoint* kps, int cap, int* n) {
  return orb_api(ctx, img, w, h, stride, channels, kps, nullptr, cap, n, false);
}

extern "C" int mvo_fast9_nms(mvo_ctx* ctx, const uint8_t* img, int w, int h, int stride, int threshold, int* xys,
                             int cap, int* n) {
  if (!ctx || !img || !n || w < 7 || h < 7 || w > ctx->maxw || h > ctx->maxh) return MVO_E_ARG;
  OrbState* o = ctx->orb;
  hipStream_t st = ctx->stream;
  *n = 0;
  int rc = orb_upload(ctx, img, w, h, stride, 1, 0);
  if (rc) return rc;
  // single level, no edge cull: rows/cols 3..dim-4 exactly as cv::FAST
  OrbGeom G;
  orb_geometry(w, h, ctx->cfg.nfeatures, 3, G);
  G.slot_stride = o->slot_bytes;
  G.nlevels = 1;
  G.row0[1] = (h <= 6 || w <= 6) ? 0 : h - 6;
  for (int l = 2; l <= MVO_ORB_LEVELS; l++) G.row0[l] = G.row0[1];
  MVO_HIP(hipMemsetAsync(o->d_row_cnt, 0, (size_t)o->max_rows * sizeof(int), st));
  MVO_HIP(hipMemsetAsync(o->d_seg_cnt, 0, (size_t)o->max_rows * o->seg_per_row, st));
  fast_nms_launch(ctx, G, 0, 1, threshold, 3);
  int nrows = G.row0[1];
  hipLaunchKernelGGL(scan_rows_kernel, dim3(1), dim3(1024), 0, st, o->d_row_cnt, o->d_row_off, G, o->max_rows, o->d_lvl_cnt,
                     o->d_slot_tot, (const int*)nullptr);
  hipLaunchKernelGGL(scan_slots_kernel, dim3(1), dim3(64), 0, st, o->d_slot_tot, o->d_slot_base, 1, (const int*)nullptr);
  nms_rows_launch(ctx, G, nrows, 1);
  int* hn = (int*)ctx->h_pin;
  MVO_HIP(hipMemcpyAsync(hn, o->d_slot_tot, sizeof(int), hipMemcpyDeviceToHost, st));
  MVO_HIP(hipStreamSynchronize(st));
  int total = hn[0];
  *n = total;
  if (total > o->cand_cap) { ctx->set_error("FAST candidate capacity exceeded"); return MVO_E_CAPACITY; }
  int m = total < cap ? total : cap;
  if (m > 0) {
    std::vector<unsigned short> hx(m), hy(m);
    std::vector<u8> hs(m);
    MVO_HIP(hipMemcpy(hx.data(), o->d_cx, (size_t)m * 2, hipMemcpyDeviceToHost));
    MVO_HIP(hipMemcpy(hy.data(), o->d_cy, (size_t)m * 2, hipMemcpyDeviceToHost));
    MVO_HIP(hipMemcpy(hs.data(), o->d_cs, (size_t)m, hipMemcpyDeviceToHost));
    for (int i = 0; i < m; i++) { xys[3 * i] = hx[i]; xys[3 * i + 1] = hy[i]; xys[3 * i + 2] = hs[i]; }
  }
  return total > cap ? MVO_E_CAPACITY : MVO_OK;
}
