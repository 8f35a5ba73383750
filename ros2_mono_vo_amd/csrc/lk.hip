// csrc/lk.hip — pyramidal Lucas-Kanade optical flow for gfx950 (replaces cv::calcOpticalFlowPyrLK at
// reference src/tracker.cpp:68-69; semantics per SURVEY.md Appendix A.3).
//
// Kernels
//   pyrdown_kernel : cv::pyrDown ([1 4 6 4 1]^2, (sum+128)>>8, reflect-101), LDS-tiled, one launch per level
//                    covering every slot of the batch.
//   lk_track_kernel: ONE WAVEFRONT PER TRACKED POINT, all pyramid levels inside one launch.  Per level the
//                    wave stages the 24x24 source neighbourhood of the previous image in LDS, derives the
//                    22x22 Scharr field there, and keeps its 21x21 fixed-point template (I, Ix, Iy) in
//                    registers: 63 lanes x 7 horizontally adjacent pixels.  The search window of the next
//                    image is staged as a 32x32 LDS tile (aligned dword loads; re-fetched only when the window
//                    leaves it).  The inner loop reads two unaligned 8-byte spans as dwords + v_alignbyte,
//                    packs pixel pairs with v_perm and evaluates the 14-bit bilinear sum with v_dot2c_i32_i16.
//                    Normal-equation sums are exact integers: int32 per lane, DPP wave reduction of a 16/16
//                    split, so results are independent of summation order and bit-identical to the oracle.
#include "mvo_internal.h"

// ---------------------------------------------------------------------------------------------------
// pyrDown
// ---------------------------------------------------------------------------------------------------
#define PD_TW 64
#define PD_TH 16
#define PD_SR (2 * PD_TH + 3)   // source tile rows
#define PD_SD 35                // source tile pitch in dwords: 136 bytes (2*64 + 8, origin 2*dx0 - 4) + pad, odd
#define PD_HD 33                // row-sum tile pitch in dwords (32 packed u16 pairs + pad)
typedef unsigned short pd_us2 __attribute__((ext_vector_type(2)));

// cv::pyrDown (5x5 [1 4 6 4 1]^2 / 256, BORDER_REFLECT_101): source tile by dwords, horizontal taps with v_dot4_u32_u8
// (two outputs per item), vertical taps on packed u16 pairs (the 16 * 4080 maximum fits 16 bits), (v + 128) >> 8.
__global__ __launch_bounds__(256) void pyrdown_kernel(ImgSet src, ImgSet dst, TileGrid tg) {
  __shared__ unsigned s_src[PD_SR * PD_SD];
  __shared__ unsigned s_h[PD_SR * PD_HD];
  int bx, by, bz;
  if (!xcd_tile(tg, bx, by, bz)) return;
  const u8* sp = src.slot(bz);
  u8* dp = dst.slot(bz);
  const int dx0 = bx * PD_TW, dy0 = by * PD_TH, tid = threadIdx.x;
  const int sx0 = 2 * dx0 - 4, sy0 = 2 * dy0 - 2;   // tile origin (dword aligned in x)
  // 136 bytes per row as eight 16-byte loads and one 8-byte load (a lane address costs the same for 4 or 16 bytes).
  // Border tiles: rows by reflect-101; the left-most tile starts at column 0 one dword further into the LDS row;
  // nothing is read past the row pitch; then the two reflected columns the 5-tap filter needs on either side are
  // copied inside LDS (outputs stop at 2 * dst.w - 1 <= src.w, so at most columns -2, -1 and w, w + 1).
  const bool tiny = src.w < 8 || src.h < 8;
  if (!tiny) {
    const int sh1 = sx0 < 0 ? 1 : 0;
    for (int i = tid; i < PD_SR * 9; i += 256) {
      const int ty = __umul24(i, 7282) >> 16, k = i - ty * 9;   // i / 9 for i < 2^12
      const int gy = d_reflect101(sy0 + ty, src.h);
      const int gx = sx0 + 16 * k + 4 * sh1;
      const u8* gp = sp + (size_t)__umul24(gy, src.pitch) + gx;
      unsigned* lp = s_src + ty * PD_SD + 4 * k + sh1;
      if (gx + 16 <= src.pitch && k < 8) { const uint4 v = *(const uint4*)gp; lp[0] = v.x; lp[1] = v.y; lp[2] = v.z; lp[3] = v.w; }
      else {
        const int nd = k < 8 ? 4 : 2;
        for (int j = 0; j < nd; j++) lp[j] = gx + 4 * j + 4 <= src.pitch ? *(const unsigned*)(gp + 4 * j) : 0u;
      }
    }
    if (sx0 < 0 || sx0 + 136 > src.w) {   // block-uniform
      __syncthreads();
      if (tid < PD_SR) {
        u8* rowb = (u8*)s_src + tid * (PD_SD * 4);
        if (sx0 < 0) { rowb[2] = rowb[6]; rowb[3] = rowb[5]; }   // x = -2, -1 <- 2, 1
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int x = src.w + j;                                // <- w - 2 - j
          if (x < sx0 + 136 && x >= sx0) rowb[x - sx0] = rowb[src.w - 2 - j - sx0];
        }
      }
    }
  } else {
    u8* sb = (u8*)s_src;
    for (int i = tid; i < PD_SR * 136; i += 256) {
      const int ty = i / 136, tx = i - ty * 136;
      const int sx = d_reflect101(sx0 + tx, src.w), sy = d_reflect101(sy0 + ty, src.h);
      sb[ty * (PD_SD * 4) + tx] = sp[(size_t)__umul24(sy, src.pitch) + sx];
    }
  }
  __syncthreads();
  // horizontal: item = (row, pair m) -> outputs 2m, 2m+1 from tile bytes 4m+2 .. 4m+8
  for (int i = tid; i < PD_SR * 32; i += 256) {
    const int ty = i >> 5, m = i & 31;
    const unsigned* r = s_src + ty * PD_SD + m;
    const unsigned d0 = r[0], d1 = r[1], d2 = r[2];
    unsigned o0 = __builtin_amdgcn_udot4(d0, 0x04010000u, __builtin_amdgcn_udot4(d1, 0x00010406u, 0u, false), false);
    unsigned o1 = __builtin_amdgcn_udot4(d1, 0x04060401u, d2 & 0xFFu, false);
    s_h[ty * PD_HD + m] = o0 | (o1 << 16);
  }
  __syncthreads();
  // vertical: one item per thread = 4 outputs (two packed pairs) of one row
  {
    const int ty = tid >> 4, q = tid & 15;
    const int x = dx0 + 4 * q, y = dy0 + ty;
    if (x < dst.w && y < dst.h) {
      const unsigned* c = s_h + (2 * ty) * PD_HD + 2 * q;
      unsigned out = 0;
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const pd_us2 h0 = __builtin_bit_cast(pd_us2, c[half]), h1 = __builtin_bit_cast(pd_us2, c[PD_HD + half]),
                     h2 = __builtin_bit_cast(pd_us2, c[2 * PD_HD + half]), h3 = __builtin_bit_cast(pd_us2, c[3 * PD_HD + half]),
                     h4 = __builtin_bit_cast(pd_us2, c[4 * PD_HD + half]);
        const pd_us2 six = {6, 6}, four = {4, 4}, rnd = {128, 128};
        pd_us2 v = h2 * six + (h1 + h3) * four + h0 + h4 + rnd;
        out |= ((unsigned)(v.x >> 8) | ((unsigned)(v.y >> 8) << 8)) << (16 * half);
      }
      *(unsigned*)(dp + (size_t)__umul24(y, dst.pitch) + x) = out;   // x % 4 == 0, pitch % 64 == 0
    }
  }
}

static void launch_pyrdown(mvo_ctx* ctx, const ImgSet& s, const ImgSet& d, int nslots, hipStream_t st) {
  TileGrid tg{(d.w + PD_TW - 1) / PD_TW, (d.h + PD_TH - 1) / PD_TH, nslots};
  hipLaunchKernelGGL(pyrdown_kernel, dim3(xcd_grid_blocks(tg)), dim3(256), 0, st, s, d, tg);
}

// ---------------------------------------------------------------------------------------------------
// LK tracker
// ---------------------------------------------------------------------------------------------------
struct LkLevelDesc {
  const u8* I;  // previous image, level l (slot 0)
  const u8* J;  // next image
  int w, h, pitch;
};
struct LkArgs {
  LkLevelDesc lv[MVO_LK_MAX_LEVELS];
  size_t slot_stride;  // bytes between slots in both pyramid sets
  int nlevels;         // levels in use (maxLevel + 1)
  const float* prev_pts;
  float* next_pts;
  u8* status;
  float* err;
  const int* npts;  // [B]
  int maxpts;
  int cn;  // channel-count semantics (see oracle/orc_lk.cpp header)
  int max_count;
  double eps2;
  double min_eig;
  // device-driven launch (frame-batch tracker): the points of all slots form one dense work list,
  // work_slot[w] = slot of item w, pt_base[slot] = first item of the slot, pt_base[nslots] = item count; persistent
  // wavefronts claim LK_CHUNK items at a time from the eight counters work_ctr[0..8) (one per XCD part of the list, zeroed
  // before the launch).  Null: one wavefront per
  // (blockIdx.x * 4 + wave, blockIdx.y) with the host-sized grid.
  const int* work_slot;
  const int* pt_base;
  int* work_ctr;
  int nslots;
};
#ifndef LK_CHUNK
#define LK_CHUNK 4
#endif
#ifndef LK_PARTS
#define LK_PARTS 1   // parts of the work list: 8 = one per XCD (5.6x less HBM traffic, but 9-15 % slower: see DESIGN.md)
#endif

#define LK_WIN 21
#define LK_IT 24          // I tile edge (WIN + 1 bilinear + 2 Scharr halo)
#define LK_IP 36          // I tile pitch: two 16-byte loads cover 24 bytes at any 4-byte phase; 9 dwords (odd: rows spread over banks)
#define LK_DT 22          // derivative tile edge
#define LK_JT 32          // J tile edge
#define LK_JP 52          // J tile pitch: three 16-byte loads cover 32 bytes at any 4-byte phase; 13 dwords
#define LK_JSLACK ((LK_JT - (LK_WIN + 1)) / 2)

typedef short lk_short2 __attribute__((ext_vector_type(2)));

// Wave-wide integer sum with DPP adds (no LDS traffic, ~6 dependent VALU ops) -> value in every lane.
__device__ __forceinline__ int wave_sum_i32_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);  // row_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);  // row_bcast:15 -> rows 1,3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);  // row_bcast:31 -> rows 2,3
  return __builtin_amdgcn_readlane(v, 63);
}

// Exact 64-bit wave sum of int32 partials: split into a signed high part and an unsigned 16-bit low part,
// each of which sums without overflow in 32 bits across 64 lanes.
__device__ __forceinline__ long long wave_sum_exact(int v) {
  int lo = v & 0xFFFF, hi = v >> 16;
  return (long long)wave_sum_i32_dpp(hi) * 65536LL + (long long)wave_sum_i32_dpp(lo);
}

// The same sum when every lane's partial is known to be small enough that the first PLAIN steps of the butterfly cannot
// overflow: |v| < 2^28 lets groups of 8 lanes (3 steps) be summed in int32, |v| < 2^27 groups of 16 (4 steps); only
// the remaining steps run on the hi / lo halves.  Bounds used by the callers: |J - I| <= 8160 and |Ix|, |Iy| <= 4080
// (u8 image, 14-bit weights, 5 extra fractional bits; Scharr taps sum to 16), 7 pixels per lane.
// The callers turn the sum into a float: (float)(sum * cn).  hi * 65536 + lo is exact in a double (|sum| < 2^47) and so is
// the product with cn, and v_cvt_f32_f64 rounds to nearest even exactly like the int64 -> float conversion: same float,
// five VALU instructions instead of the ~25 SALU instructions of the 64-bit integer path.
template <int PLAIN>
__device__ __forceinline__ double wave_sum_exact_bounded(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);  // row_half_mirror: groups of 8
  if (PLAIN >= 4) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);  // row_mirror: groups of 16
  int lo = v & 0xFFFF, hi = v >> 16;
  if (PLAIN < 4) {
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true);
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true);
  }
  lo += __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, true);  // row_bcast:15
  hi += __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, true);
  lo += __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xC, 0xF, true);  // row_bcast:31
  hi += __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xC, 0xF, true);
  return (double)__builtin_amdgcn_readlane(hi, 63) * 65536.0 + (double)__builtin_amdgcn_readlane(lo, 63);
}

__device__ __forceinline__ void lk_weights(float a, float b, int& w00, int& w01, int& w10, int& w11) {
  const int W_BITS = 14;
  w00 = d_cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
  w01 = d_cv_round(a * (1.f - b) * (1 << W_BITS));
  w10 = d_cv_round((1.f - a) * b * (1 << W_BITS));
  w11 = (1 << W_BITS) - w00 - w01 - w10;
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

struct LkWaveLds {
  unsigned it[LK_IT * LK_IP / 4];   // previous-image neighbourhood (rows of LK_IP bytes)
  short2 dt[LK_DT * LK_DT];         // Scharr (dx, dy)
  unsigned jt[LK_JT * LK_JP / 4];   // next-image search tile (rows of LK_JP bytes)
};

// Stage ROWS x TW bytes of the image starting at (x0, y0) into LDS rows of PD dwords.  Fast path: the tile lies inside
// the image -> NX4 aligned 16-byte loads per row (a lane address on the texture path costs the same for 4 or 16
// bytes), the tile starts `shift` bytes into each LDS row.  Slow path (image border): per-byte reflect-101.  Returns
// the byte shift (wave-uniform).
template <int ROWS, int TW, int PD, int NX4>
__device__ __forceinline__ int lk_load_tile(unsigned* lds, const u8* __restrict__ img, int w, int h, int pitch, int x0, int y0, int lane) {
  const int xa = x0 & ~3;
  const bool inside = x0 >= 0 && y0 >= 0 && x0 + TW <= w && y0 + ROWS <= h && xa + 16 * NX4 <= pitch;
  if (inside) {
    const u8* base = img + (size_t)y0 * pitch + xa;
    for (int i = lane; i < ROWS * NX4; i += 64) {
      const int row = i / NX4, k = i - row * NX4;
      const uint4 v = *(const uint4*)(base + (size_t)row * pitch + 16 * k);
      unsigned* d = lds + row * PD + 4 * k;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    return x0 - xa;
  }
  u8* lb = (u8*)lds;
  for (int i = lane; i < ROWS * TW; i += 64) {
    int row = i / TW, col = i - row * TW;
    int gx = d_reflect101(x0 + col, w), gy = d_reflect101(y0 + row, h);
    lb[row * (PD * 4) + col] = img[(size_t)gy * pitch + gx];
  }
  return 0;
}

// One lane's 7 pixels of (bilinear J - I) against (Ix, Iy) or |.|: two unaligned 8-byte row spans are read
// as 3 dwords each and re-aligned with v_alignbyte.  v_perm builds the eight VERTICAL pairs V_k = (row0[k], row1[k])
// as 16-bit lanes - neighbouring pixels share them - and two v_dot2c_i32_i16 per pixel, V_k . (w00, w10) +
// V_k+1 . (w01, w11), evaluate the 4-tap fixed-point bilinear sum exactly.  Wa = w00 | w10 << 16, Wb = w01 | w11 << 16.
// (a.lo * b.lo + a.hi * b.hi) + c on packed signed 16-bit pairs, three-operand form (no accumulator move)
__device__ __forceinline__ int lk_dot2(unsigned a, unsigned b, int c) {
  int r;
  asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// low halves of two registers as one packed pair (lo, hi)
__device__ __forceinline__ unsigned lk_pack16(int lo, int hi) { return __builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x05040100u); }

// IxP / IyP: the lane's seven derivative values as packed pairs (0,1) (2,3) (4,5) (6,-): with the differences packed the same
// way one v_dot2_i32_i16 accumulates two pixels of sum(diff * Ix) (|diff| <= 8160, |Ix|, |Iy| <= 4080: all fit 16 bits).
template <bool ERR>
__device__ __forceinline__ void lk_accumulate(const unsigned* jt, int byte_off, bool active, unsigned Wa, unsigned Wb, const int* Iv,
                                              const unsigned* IxP, const unsigned* IyP, int& s1, int& s2) {
  s1 = 0; s2 = 0;
  if (!active) return;
  const int sh = byte_off & 3;
  const unsigned* q = jt + (byte_off >> 2);
  unsigned a0 = q[0], a1 = q[1], a2 = q[2];
  unsigned b0 = q[LK_JP / 4], b1 = q[LK_JP / 4 + 1], b2 = q[LK_JP / 4 + 2];
  unsigned r0lo = __builtin_amdgcn_alignbyte(a1, a0, sh), r0hi = __builtin_amdgcn_alignbyte(a2, a1, sh);
  unsigned r1lo = __builtin_amdgcn_alignbyte(b1, b0, sh), r1hi = __builtin_amdgcn_alignbyte(b2, b1, sh);
  const lk_short2 wb = __builtin_bit_cast(lk_short2, Wb);
  lk_short2 V[8];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const unsigned sel = (unsigned)k | (0x0Cu << 8) | ((unsigned)(4 + k) << 16) | (0x0Cu << 24);   // (src1 byte k, src0 byte k)
    V[k] = __builtin_bit_cast(lk_short2, __builtin_amdgcn_perm(r1lo, r0lo, sel));
    V[4 + k] = __builtin_bit_cast(lk_short2, __builtin_amdgcn_perm(r1hi, r0hi, sel));
  }
  int diff[7];
#pragma unroll
  for (int k = 0; k < 7; k++) {
    int acc = lk_dot2(__builtin_bit_cast(unsigned, V[k]), Wa, Iv[k]);
    acc = __builtin_amdgcn_sdot2(V[k + 1], wb, acc, false);
    diff[k] = acc >> 9;   // Iv[k] = 256 - 512 * I:  ((J + 256) >> 9) - I == (J + 256 - 512 * I) >> 9 exactly
  }
  if (ERR) {
#pragma unroll
    for (int k = 0; k < 7; k++) s1 += abs(diff[k]);
  } else {
    const unsigned d01 = lk_pack16(diff[0], diff[1]), d23 = lk_pack16(diff[2], diff[3]), d45 = lk_pack16(diff[4], diff[5]);
    const unsigned d6 = (unsigned)diff[6] & 0xFFFFu;
    s1 = lk_dot2(d01, IxP[0], lk_dot2(d23, IxP[1], lk_dot2(d45, IxP[2], lk_dot2(d6, IxP[3], 0))));
    s2 = lk_dot2(d01, IyP[0], lk_dot2(d23, IyP[1], lk_dot2(d45, IyP[2], lk_dot2(d6, IyP[3], 0))));
  }
}

__device__ __forceinline__ void lk_track_point(const LkArgs& A, LkWaveLds& S, const int slot, const int p, const int lane) {
  const size_t pidx = (size_t)slot * A.maxpts + p;
  const float ptx = A.prev_pts[2 * pidx], pty = A.prev_pts[2 * pidx + 1];
  const float FLT_SCALE = 1.f / (1 << 20);
  const float half = (LK_WIN - 1) * 0.5f;
  const int r = lane / 3, x0 = (lane - r * 3) * 7;
  const bool active = lane < 63;

  int status = 1;
  float errv = 0.f;
  float sx = 0.f, sy = 0.f;  // nextPts[ptidx] as stored by OpenCV between levels

  for (int level = A.nlevels - 1; level >= 0; level--) {
    const LkLevelDesc lv = A.lv[level];
    const u8* I = lv.I + (size_t)slot * A.slot_stride;
    const u8* J = lv.J + (size_t)slot * A.slot_stride;
    float px = ptx * (float)(1. / (1 << level));
    float py = pty * (float)(1. / (1 << level));
    float nx, ny;
    if (level == A.nlevels - 1) { nx = px; ny = py; }
    else { nx = sx * 2.f; ny = sy * 2.f; }
    sx = nx; sy = ny;
    px -= half; py -= half;
    int ipx = d_cv_floor(px), ipy = d_cv_floor(py);
    if (ipx < -LK_WIN || ipx >= lv.w || ipy < -LK_WIN || ipy >= lv.h) {
      if (level == 0) { status = 0; errv = 0.f; }
      continue;
    }
    // ---- stage the 24x24 neighbourhood of I (origin ipx-1, ipy-1) -------------------------------------
    __builtin_amdgcn_wave_barrier();
    const int shI = lk_load_tile<LK_IT, LK_IT, LK_IP / 4, 2>(S.it, I, lv.w, lv.h, lv.pitch, ipx - 1, ipy - 1, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- Scharr field on the 22x22 window-source positions (zero outside the image) --------------
    // Separable form, 11 pixels per lane (22 rows x 2 halves): with S = 3*(top + bottom) + 10*mid and D = bottom - top
    // per column, dx = S[c+1] - S[c-1] and dy = 3*(D[c-1] + D[c+1]) + 10*D[c].  Three tile rows of 13 bytes come in as
    // 4 aligned dwords each and are re-phased with v_alignbyte.
    if (lane < 2 * LK_DT) {
      const int ty = lane >> 1, hx = lane & 1, tx0 = 11 * hx;
      const int bo = shI + tx0, ph = bo & 3;
      const unsigned* rp = S.it + ty * (LK_IP / 4) + (bo >> 2);
      unsigned col[3][4];   // bytes tx0 .. tx0+12 of tile rows ty, ty+1, ty+2 (byte 12 in col[.][3] bits 0-7)
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const unsigned d0 = rp[q * (LK_IP / 4)], d1 = rp[q * (LK_IP / 4) + 1], d2 = rp[q * (LK_IP / 4) + 2], d3 = rp[q * (LK_IP / 4) + 3];
        col[q][0] = __builtin_amdgcn_alignbyte(d1, d0, ph);
        col[q][1] = __builtin_amdgcn_alignbyte(d2, d1, ph);
        col[q][2] = __builtin_amdgcn_alignbyte(d3, d2, ph);
        col[q][3] = d3 >> (8 * ph);
      }
      // packed 16-bit arithmetic, two columns per register: S <= 4080 and |D| <= 255, |dx|, |dy| <= 4080
      typedef unsigned short lk_us2 __attribute__((ext_vector_type(2)));
      lk_us2 Sp[7], Dp[7];   // columns (2j, 2j+1); column 13 is padding and only reaches the unused output 11
#pragma unroll
      for (int j = 0; j < 7; j++) {
        const unsigned selp = (j & 1) ? 0x0c030c02u : 0x0c010c00u;   // bytes (2, 3) / (0, 1) of the dword as two u16
        const lk_us2 t = __builtin_bit_cast(lk_us2, __builtin_amdgcn_perm(0u, col[0][j >> 1], selp));
        const lk_us2 m = __builtin_bit_cast(lk_us2, __builtin_amdgcn_perm(0u, col[1][j >> 1], selp));
        const lk_us2 b = __builtin_bit_cast(lk_us2, __builtin_amdgcn_perm(0u, col[2][j >> 1], selp));
        const lk_us2 three = {3, 3}, ten = {10, 10};
        Sp[j] = (t + b) * three + m * ten;
        Dp[j] = b - t;   // two's complement in 16 bits
      }
      const int gy = ipy + ty, gx0 = ipx + tx0;
      const bool row_ok = (unsigned)gy < (unsigned)lv.h;
      const bool all_ok = row_ok && gx0 >= 0 && gx0 + 10 < lv.w;
      const bool fast = __all(all_ok) != 0;   // nearly always: the window lies inside the image
      unsigned* dst = (unsigned*)&S.dt[ty * LK_DT + tx0];
#pragma unroll
      for (int j = 0; j < 6; j++) {
        const lk_us2 three = {3, 3}, ten = {10, 10};
        const lk_us2 dxp = Sp[j + 1] - Sp[j];                                    // dx of outputs 2j, 2j+1
        const lk_us2 mid = __builtin_bit_cast(lk_us2, __builtin_amdgcn_alignbyte(__builtin_bit_cast(unsigned, Dp[j + 1]),
                                                                                  __builtin_bit_cast(unsigned, Dp[j]), 2));
        const lk_us2 dyp = (Dp[j] + Dp[j + 1]) * three + mid * ten;              // dy of outputs 2j, 2j+1
        const unsigned X = __builtin_bit_cast(unsigned, dxp), Y = __builtin_bit_cast(unsigned, dyp);
        unsigned e0 = __builtin_amdgcn_perm(Y, X, 0x05040100u);                  // (dx, dy) of output 2j
        unsigned e1 = __builtin_amdgcn_perm(Y, X, 0x07060302u);                  // (dx, dy) of output 2j+1
        if (!fast) {
          if (!(row_ok && (unsigned)(gx0 + 2 * j) < (unsigned)lv.w)) e0 = 0u;
          if (!(row_ok && (unsigned)(gx0 + 2 * j + 1) < (unsigned)lv.w)) e1 = 0u;
        }
        dst[2 * j] = e0;
        if (j < 5) dst[2 * j + 1] = e1;   // output 11 does not exist
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- template in registers + exact A sums ---------------------------------------------------------
    // Same machinery as the iteration: vertical 16-bit pairs (row r, row r + 1) by v_perm, shared by neighbouring
    // pixels, and the 4-tap fixed-point sums by two v_dot2c_i32_i16; the derivative field is interleaved (dx, dy), so
    // the low / high halves of two vertically adjacent entries give the dx / dy pairs.
    int w00, w01, w10, w11;
    lk_weights(px - ipx, py - ipy, w00, w01, w10, w11);
    int Iv[7];
    unsigned IxP[4], IyP[4];   // (Ix, Iy) of the lane's 7 pixels as packed 16-bit pairs (0,1) (2,3) (4,5) (6,-)
    int a11 = 0, a12 = 0, a22 = 0;
    if (active) {
      const lk_short2 wa = __builtin_bit_cast(lk_short2, (unsigned)w00 | ((unsigned)w10 << 16));
      const lk_short2 wb = __builtin_bit_cast(lk_short2, (unsigned)w01 | ((unsigned)w11 << 16));
      {
        const int bo = (r + 1) * LK_IP + x0 + 1 + shI, sh = bo & 3;
        const unsigned* q = S.it + (bo >> 2);
        const unsigned a0 = q[0], a1 = q[1], a2 = q[2];
        const unsigned b0 = q[LK_IP / 4], b1 = q[LK_IP / 4 + 1], b2 = q[LK_IP / 4 + 2];
        const unsigned r0lo = __builtin_amdgcn_alignbyte(a1, a0, sh), r0hi = __builtin_amdgcn_alignbyte(a2, a1, sh);
        const unsigned r1lo = __builtin_amdgcn_alignbyte(b1, b0, sh), r1hi = __builtin_amdgcn_alignbyte(b2, b1, sh);
        lk_short2 V[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned sel = (unsigned)k | (0x0Cu << 8) | ((unsigned)(4 + k) << 16) | (0x0Cu << 24);
          V[k] = __builtin_bit_cast(lk_short2, __builtin_amdgcn_perm(r1lo, r0lo, sel));
          V[4 + k] = __builtin_bit_cast(lk_short2, __builtin_amdgcn_perm(r1hi, r0hi, sel));
        }
#pragma unroll
        for (int i = 0; i < 7; i++) {
          int acc = __builtin_amdgcn_sdot2(V[i], wa, 0, false);
          acc = __builtin_amdgcn_sdot2(V[i + 1], wb, acc, false);
          Iv[i] = 256 - (descale(acc, 14 - 5) << 9);   // kept as the start value of the iteration's accumulator (lk_accumulate)
        }
      }
      const unsigned* d0 = (const unsigned*)&S.dt[r * LK_DT + x0];
      const unsigned* d1 = d0 + LK_DT;
      lk_short2 XV[8], YV[8];   // (dx, dx below), (dy, dy below) at positions x0 .. x0 + 7
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const unsigned e0 = d0[i], e1 = d1[i];
        XV[i] = __builtin_bit_cast(lk_short2, __builtin_amdgcn_perm(e1, e0, 0x05040100u));
        YV[i] = __builtin_bit_cast(lk_short2, __builtin_amdgcn_perm(e1, e0, 0x07060302u));
      }
      int ixv[7], iyv[7];
#pragma unroll
      for (int i = 0; i < 7; i++) {
        int ix = __builtin_amdgcn_sdot2(XV[i], wa, 0, false);
        ixv[i] = descale(__builtin_amdgcn_sdot2(XV[i + 1], wb, ix, false), 14);
        int iy = __builtin_amdgcn_sdot2(YV[i], wa, 0, false);
        iyv[i] = descale(__builtin_amdgcn_sdot2(YV[i + 1], wb, iy, false), 14);
      }
#pragma unroll
      for (int j = 0; j < 3; j++) { IxP[j] = lk_pack16(ixv[2 * j], ixv[2 * j + 1]); IyP[j] = lk_pack16(iyv[2 * j], iyv[2 * j + 1]); }
      IxP[3] = (unsigned)ixv[6] & 0xFFFFu; IyP[3] = (unsigned)iyv[6] & 0xFFFFu;
#pragma unroll
      for (int j = 0; j < 4; j++) {   // sum(ix * ix) etc. two pixels per instruction; exact integers either way
        a11 = lk_dot2(IxP[j], IxP[j], a11); a12 = lk_dot2(IxP[j], IyP[j], a12); a22 = lk_dot2(IyP[j], IyP[j], a22);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 7; i++) Iv[i] = 256;
#pragma unroll
      for (int j = 0; j < 4; j++) { IxP[j] = 0u; IyP[j] = 0u; }
    }
    // per lane: 7 * 4080^2 < 2^27
    const double cnd = (double)A.cn;
    const double sA11 = wave_sum_exact_bounded<4>(a11), sA12 = wave_sum_exact_bounded<4>(a12), sA22 = wave_sum_exact_bounded<4>(a22);
    float A11 = (float)(sA11 * cnd) * FLT_SCALE;
    float A12 = (float)(sA12 * cnd) * FLT_SCALE;
    float A22 = (float)(sA22 * cnd) * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                   (float)(2 * LK_WIN * LK_WIN);
    if ((double)minEig < A.min_eig || D < 1.1920928955078125e-07f) {
      if (level == 0) status = 0;
      continue;
    }
    D = 1.f / D;
    nx -= half; ny -= half;
    float pdx = 0.f, pdy = 0.f;
    int jx0 = 0, jy0 = 0, shJ = 0;
    bool have_tile = false;
    for (int j = 0; j < A.max_count; j++) {
      int inx = d_cv_floor(nx), iny = d_cv_floor(ny);
      if (inx < -LK_WIN || inx >= lv.w || iny < -LK_WIN || iny >= lv.h) {
        if (level == 0) status = 0;
        break;
      }
      int ddx = inx - jx0, ddy = iny - jy0;
      if (!have_tile || ddx < 0 || ddx > LK_JT - (LK_WIN + 1) || ddy < 0 || ddy > LK_JT - (LK_WIN + 1)) {
        jx0 = inx - LK_JSLACK; jy0 = iny - LK_JSLACK;
        __builtin_amdgcn_wave_barrier();
        shJ = lk_load_tile<LK_JT, LK_JT, LK_JP / 4, 3>(S.jt, J, lv.w, lv.h, lv.pitch, jx0, jy0, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        have_tile = true;
        ddx = LK_JSLACK; ddy = LK_JSLACK;
      }
      lk_weights(nx - inx, ny - iny, w00, w01, w10, w11);
      int s1, s2;
      lk_accumulate<false>(S.jt, (r + ddy) * LK_JP + x0 + ddx + shJ, active, (unsigned)w00 | ((unsigned)w10 << 16),
                           (unsigned)w01 | ((unsigned)w11 << 16), Iv, IxP, IyP, s1, s2);
      const double sb1 = wave_sum_exact_bounded<3>(s1), sb2 = wave_sum_exact_bounded<3>(s2);  // per lane: 7 * 8160 * 4080 < 2^28
      float b1 = (float)(sb1 * cnd) * FLT_SCALE;
      float b2 = (float)(sb2 * cnd) * FLT_SCALE;
      float dx = (A12 * b2 - A22 * b1) * D;
      float dy = (A12 * b1 - A11 * b2) * D;
      nx += dx; ny += dy;
      sx = nx + half; sy = ny + half;
      if ((double)dx * dx + (double)dy * dy <= A.eps2) break;
      if (j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
        sx -= dx * 0.5f;
        sy -= dy * 0.5f;
        break;
      }
      pdx = dx; pdy = dy;
    }
    if (status && level == 0) {
      float ex = sx - half, ey = sy - half;
      int inx = d_cv_floor(ex), iny = d_cv_floor(ey);
      if (inx < -LK_WIN || inx >= lv.w || iny < -LK_WIN || iny >= lv.h) {
        status = 0;
      } else {
        int ddx = inx - jx0, ddy = iny - jy0;
        if (!have_tile || ddx < 0 || ddx > LK_JT - (LK_WIN + 1) || ddy < 0 || ddy > LK_JT - (LK_WIN + 1)) {
          jx0 = inx - LK_JSLACK; jy0 = iny - LK_JSLACK;
          __builtin_amdgcn_wave_barrier();
          shJ = lk_load_tile<LK_JT, LK_JT, LK_JP / 4, 3>(S.jt, J, lv.w, lv.h, lv.pitch, jx0, jy0, lane);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          ddx = LK_JSLACK; ddy = LK_JSLACK;
        }
        lk_weights(ex - inx, ey - iny, w00, w01, w10, w11);
        int s1, s2;
        lk_accumulate<true>(S.jt, (r + ddy) * LK_JP + x0 + ddx + shJ, active, (unsigned)w00 | ((unsigned)w10 << 16),
                            (unsigned)w01 | ((unsigned)w11 << 16), Iv, IxP, IyP, s1, s2);
        const double se = (double)wave_sum_i32_dpp(s1);  // 64 * 7 * 8160 fits 32 bits
        float errval = (float)(se * cnd);
        errv = errval * 1.f / (float)(32 * LK_WIN * A.cn * LK_WIN);
      }
    }
  }
  if (lane == 0) {
    A.next_pts[2 * pidx] = sx;
    A.next_pts[2 * pidx + 1] = sy;
    A.status[pidx] = (u8)status;
    A.err[pidx] = errv;
  }
}

// Register budget: 6 wavefronts per SIMD (<= 80 VGPRs, no spills).  The kernel is a chain of dependent instructions per
// wavefront (one tracked point), so resident wavefronts are what hides its latencies.
#ifndef LK_WAVES_PER_EU
#define LK_WAVES_PER_EU 6
#endif
__global__ __launch_bounds__(256, LK_WAVES_PER_EU) void lk_track_kernel(LkArgs A) {
  __shared__ LkWaveLds lds[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  LkWaveLds& S = lds[wave];
  if (A.work_slot) {
    // The list is slot major.  It is cut into 8 contiguous parts, one per XCD (workgroups go to the XCDs round-robin by
    // linear id, so blockIdx.x & 7 is the XCD - a speed hint only): an XCD then walks whole slots and both pyramids of a
    // slot (2.4 MB at 720p) stay in ITS 4 MB L2, instead of every XCD fetching every slot's lines (measured: 4.1x the
    // algorithmic bytes without the cut).  A wavefront whose part has run dry helps with the next ones.
    const int total = min(max(A.pt_base[A.nslots], 0), A.nslots * A.maxpts);
    const int home = blockIdx.x % LK_PARTS;
    for (int k = 0; k < LK_PARTS; k++) {   // every wavefront leaves once all counters have passed their parts
      const int part = (home + k) % LK_PARTS;
      const int lo = (int)((long long)total * part / LK_PARTS), hi = (int)((long long)total * (part + 1) / LK_PARTS);
      for (;;) {
        int w0 = 0;
        if (lane == 0) w0 = atomicAdd(A.work_ctr + part, LK_CHUNK);
        w0 = lo + __builtin_amdgcn_readfirstlane(w0);
        if (w0 >= hi) break;
        const int w1 = min(w0 + LK_CHUNK, hi);
        for (int w = w0; w < w1; w++) {
          const int slot = min(max(A.work_slot[w], 0), A.nslots - 1);
          const int p = w - A.pt_base[slot];
          if (p >= 0 && p < A.maxpts) lk_track_point(A, S, slot, p, lane);   // always true for a consistent list
        }
      }
    }
    return;
  }
  const int slot = blockIdx.y;
  const int p = blockIdx.x * 4 + wave;
  if (p >= min(A.npts[slot], A.maxpts)) return;  // wave-uniform
  lk_track_point(A, S, slot, p, lane);
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static ImgSet lk_imgset(mvo_ctx* ctx, int set, const LkLevels& L, int level) {
  ImgSet s;
  s.base = ctx->lk_mem[set] + ctx->lk_level_off[level];
  s.w = L.w[level]; s.h = L.h[level]; s.pitch = L.pitch[level];
  s.slot_stride = ctx->lk_slot_bytes;
  return s;
}

// Build levels 1.. of pyramid set `set` for `nslots` slots (level 0 already resident).
int lk_build_pyramid(mvo_ctx* ctx, int set, const LkLevels& L, int nslots, hipStream_t st) {
  if (!st) st = ctx->stream;
  for (int l = 1; l < L.n; l++) launch_pyrdown(ctx, lk_imgset(ctx, set, L, l - 1), lk_imgset(ctx, set, L, l), nslots, st);
  return MVO_OK;
}

// Track d_prev_pts -> d_next_pts for `nslots` slots between pyramid sets prev_set and cur_set.
int lk_track_device(mvo_ctx* ctx, int prev_set, int cur_set, const LkLevels& L, int nslots, int max_n, hipStream_t st,
                    const int* d_work_slot, const int* d_pt_base, int* d_work_ctr) {
  if (!st) st = ctx->stream;
  LkArgs A;
  memset(&A, 0, sizeof(A));
  A.work_slot = d_work_slot; A.pt_base = d_pt_base; A.work_ctr = d_work_ctr; A.nslots = nslots;
  for (int l = 0; l < L.n; l++) {
    ImgSet p = lk_imgset(ctx, prev_set, L, l), c = lk_imgset(ctx, cur_set, L, l);
    A.lv[l].I = p.base; A.lv[l].J = c.base;
    A.lv[l].w = L.w[l]; A.lv[l].h = L.h[l]; A.lv[l].pitch = L.pitch[l];
  }
  A.slot_stride = ctx->lk_slot_bytes;
  A.nlevels = L.n;
  A.prev_pts = ctx->d_prev_pts; A.next_pts = ctx->d_next_pts;
  A.status = ctx->d_status; A.err = ctx->d_err; A.npts = ctx->d_npts;
  A.maxpts = ctx->maxpts;
  A.cn = ctx->cfg.lk_channels < 1 ? 1 : ctx->cfg.lk_channels;
  int mc = ctx->cfg.lk_max_count; mc = mc < 0 ? 0 : (mc > 100 ? 100 : mc);
  double eps = ctx->cfg.lk_epsilon; eps = eps < 0 ? 0 : (eps > 10 ? 10 : eps);
  A.max_count = mc;
  A.eps2 = eps * eps;
  A.min_eig = ctx->cfg.lk_min_eig;
  if (d_work_slot) {
    // persistent wavefronts: LK_WAVES_PER_EU workgroups of 4 per CU is what the kernel's registers allow (256 CUs)
    const unsigned want = ((unsigned)nslots * (unsigned)ctx->maxpts + 4 * LK_CHUNK - 1) / (4 * LK_CHUNK);
    const unsigned full = 256u * LK_WAVES_PER_EU;
    hipLaunchKernelGGL(lk_track_kernel, dim3(want < full ? want : full), dim3(256), 0, st, A);
    return MVO_OK;
  }
  if (max_n <= 0) return MVO_OK;
  dim3 grid((max_n + 3) / 4, nslots);
  hipLaunchKernelGGL(lk_track_kernel, grid, dim3(256), 0, st, A);
  return MVO_OK;
}

extern "C" int mvo_pyrdown(mvo_ctx* ctx, const uint8_t* src, int w, int h, int stride, uint8_t* dst,
                           int dstride) {
  if (!ctx || !src || !dst || w < 1 || h < 1 || w > ctx->maxw || h > ctx->maxh) return MVO_E_ARG;
  LkLevels L = lk_levels(w, h, 0, 1);   // win 0: no early stop, a 1 x 1 level is a valid cv::pyrDown result
  ImgSet s = lk_imgset(ctx, 0, L, 0), d = lk_imgset(ctx, 0, L, 1);
  int rc = upload_gray(ctx, src, w, h, stride, 1, s.base, s.pitch, 0);
  if (rc) return rc;
  launch_pyrdown(ctx, s, d, 1, ctx->stream);
  MVO_HIP(hipMemcpy2DAsync(dst, dstride, d.base, d.pitch, d.w, d.h, hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  return MVO_OK;
}

extern "C" int mvo_lk_track(mvo_ctx* ctx, const uint8_t* prev, const uint8_t* next, int w, int h, int stride,
                            int channels, const float* prev_pts, int n, float* next_pts, uint8_t* status,
                            float* err) {
  if (!ctx || !prev || !next || n < 0 || w > ctx->maxw || h > ctx->maxh || w <= 0 || h <= 0) return MVO_E_ARG;
  if (n > ctx->maxpts) { ctx->set_error("mvo_lk_track: n exceeds max_points"); return MVO_E_CAPACITY; }
  if (n == 0) return MVO_OK;
  if (ctx->cfg.lk_win != LK_WIN) { ctx->set_error("only winSize 21x21 is built"); return MVO_E_ARG; }
  LkLevels L = lk_levels(w, h, ctx->cfg.lk_win, ctx->cfg.lk_max_level);
  int rc;
  ImgSet p0 = lk_imgset(ctx, 0, L, 0), c0 = lk_imgset(ctx, 1, L, 0);
  if ((rc = upload_gray(ctx, prev, w, h, stride, channels, p0.base, p0.pitch, 0, true))) return rc;
  if ((rc = upload_gray(ctx, next, w, h, stride, channels, c0.base, c0.pitch, 0, true))) return rc;
  if (channels != 1) {
    int differ = 0;
    if ((rc = color_channels_differ(ctx, &differ))) return rc;
    if (differ) {
      ctx->set_error("mvo_lk_track: true-colour input (channels differ) is not built: LK tracks one plane, which equals the "
                     "reference's 3-channel LK only for mono8 replicated to BGR8; convert to mono8 or pass a replicated image");
      return MVO_E_ARG;
    }
  }
  lk_build_pyramid(ctx, 0, L, 1);
  lk_build_pyramid(ctx, 1, L, 1);
  MVO_HIP(hipMemcpyAsync(ctx->d_prev_pts, prev_pts, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  int* hn = (int*)ctx->h_pin;
  hn[0] = n;
  MVO_HIP(hipMemcpyAsync(ctx->d_npts, hn, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  lk_track_device(ctx, 0, 1, L, 1, n);
  MVO_HIP(hipMemcpyAsync(next_pts, ctx->d_next_pts, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipMemcpyAsync(status, ctx->d_status, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipMemcpyAsync(err, ctx->d_err, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  return MVO_OK;
}
