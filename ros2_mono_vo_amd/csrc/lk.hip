// csrc/lk.hip — pyramidal Lucas-Kanade optical flow for gfx950 (replaces cv::calcOpticalFlowPyrLK at
// reference src/tracker.cpp:68-69; semantics per SURVEY.md Appendix A.3).
//
// Kernels
//   pyrdown_kernel : cv::pyrDown ([1 4 6 4 1]^2, (sum+128)>>8, reflect-101), LDS-tiled, one launch per level
//                    covering every slot of the batch.
//   lk_track_kernel: FOUR TRACKED POINTS PER WAVEFRONT, one per DPP row of 16 lanes, all pyramid levels inside one
//                    launch.  Per level a row stages the 24x24 source neighbourhood of the previous image in LDS, derives
//                    the 22x22 Scharr field there, and keeps its 21x21 fixed-point template (I, Ix, Iy) in registers:
//                    63 segments of 7 horizontally adjacent pixels, four per lane.  The search window of the next image
//                    is staged as a 32x32 LDS tile (aligned 16-byte loads; re-fetched only when the window leaves it).
//                    The inner loop reads two unaligned 8-byte spans as dwords + v_alignbyte, packs pixel pairs with
//                    v_perm and evaluates the 14-bit bilinear sum with v_dot2_i32_i16.  Normal-equation sums are exact
//                    integers: int32 per lane, DPP row reduction of a 16/16 split, so results are independent of
//                    summation order and bit-identical to the oracle.  The per-point scalar work (weights, 2x2 solve,
//                    convergence tests) is done by a row for its point, so one instruction serves four points.
#include "mvo_internal.h"

#include <cstdlib>

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
// pyramid levels 1..3 in ONE kernel
// ---------------------------------------------------------------------------------------------------
// A workgroup owns 16 x 16 pixels of level 3 = 32 x 32 of level 2 = 64 x 64 of level 1 = 128 x 128 of level 0 and reads its
// level-0 neighbourhood ONCE: 149 rows x 160 bytes arrive by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write),
// every level is written to HBM once and nothing is read back.  With o_l the first coordinate of the level-l region
// the tile needs (o3 = 16 T, o2 = 2 o3 - 2, o1 = 2 o2 - 2, level-0 origin 2 o1 - 4 = 128 T - 16: 16-byte aligned), region l
// is 16 / 35 / 73 wide and the source tile of a level starts 4 bytes left of 2 o: pixel pair m of a row needs source bytes
// 4m + 2 .. 4m + 8, i.e. dwords m, m + 1, m + 2 (the same v_dot4 form as pyrdown_kernel).
// A thread owns a pair of output columns and walks down a strip of rows, keeping the five horizontal sums of the
// sliding vertical window in registers (packed u16 pairs): there is no horizontal-pass tile.
// cv::pyrDown's BORDER_REFLECT_101 applies per level to THAT level's image: out-of-image rows of level 0 are reflected at
// load time, out-of-image rows of the level-1 / level-2 tiles by an index map when they are read, out-of-image columns
// (at most two on either side) by a byte fix-up inside LDS; a reflected coordinate always lies inside the 5-tap window
// that asks for it, hence inside the tile.
#define P3_S0_ROWS 149
#define P3_S0_PD 40     // dwords per level-0 tile row: ten 16-byte chunks, lane-linear (LDS-DMA writes 1 KiB per wave-instruction)
#define P3_T1_ROWS 73
#define P3_T1_PD 21     // level-1 tile: byte index = column - o1 + 2 (77 bytes used)
#define P3_T2_ROWS 35
#define P3_T2_PD 11     // level-2 tile: byte index = column - o2 + 2 (37 bytes used)
struct Pyr3Args {
  const u8* src; size_t src_stride; int w0, h0, p0;   // level 0 (e.g. a ring entry), slots src_stride apart
  u8* l1; u8* l2; u8* l3; size_t dst_stride;          // levels 1..3 of the pyramid set
  int w1, h1, p1, w2, h2, p2, w3, h3, p3;
  int lim1, lim2;                                     // 16-byte stores of levels 1 / 2 start below this column (w rounded up to 16: the
                                                      // excess lands in the plane's right border, which lk_border_kernel rewrites)
  TileGrid tg;                                        // tiles of 16 x 16 level-3 pixels x slots
};

__device__ __forceinline__ unsigned pd_hpair(unsigned d0, unsigned d1, unsigned d2) {
  const unsigned o0 = __builtin_amdgcn_udot4(d0, 0x04010000u, __builtin_amdgcn_udot4(d1, 0x00010406u, 0u, false), false);
  const unsigned o1 = __builtin_amdgcn_udot4(d1, 0x04060401u, d2 & 0xFFu, false);
  return o0 | (o1 << 16);
}
// two output pixels (bytes 0 and 1 of the result) from five packed horizontal sums: (6 h2 + 4 (h1 + h3) + h0 + h4 + 128) >> 8
__device__ __forceinline__ unsigned pd_vpair(unsigned a0, unsigned a1, unsigned a2, unsigned a3, unsigned a4) {
  const pd_us2 h0 = __builtin_bit_cast(pd_us2, a0), h1 = __builtin_bit_cast(pd_us2, a1), h2 = __builtin_bit_cast(pd_us2, a2),
               h3 = __builtin_bit_cast(pd_us2, a3), h4 = __builtin_bit_cast(pd_us2, a4);
  const pd_us2 six = {6, 6}, four = {4, 4}, rnd = {128, 128};
  const pd_us2 v = h2 * six + (h1 + h3) * four + h0 + h4 + rnd;
  return (unsigned)(v.x >> 8) | ((unsigned)(v.y >> 8) << 8);
}
// Output rows j0 .. j0 + NR - 1 of pair-column p from a source tile with PD dwords per row; rmap(r) = tile row holding source
// row r of the strip's window (r = 2 j .. 2 j + 4 for output row j); emit(j, two bytes).
template <int NR, int PD, class RowMap, class Emit>
__device__ __forceinline__ void pd_strip(const unsigned* __restrict__ src, int p, int j0, RowMap rmap, Emit emit) {
  unsigned h[2 * NR + 3];
#pragma unroll
  for (int r = 0; r < 2 * NR + 3; r++) {
    const unsigned* q = src + rmap(2 * j0 + r) * PD + p;
    h[r] = pd_hpair(q[0], q[1], q[2]);
    if (r >= 4 && !(r & 1)) emit(j0 + (r - 4) / 2, pd_vpair(h[r - 4], h[r - 3], h[r - 2], h[r - 1], h[r]));
  }
}
// columns -2, -1 and w, w + 1 of a level tile <- their reflect-101 sources (byte index of column c is c - org); one thread per row
__device__ __forceinline__ void pd_fix_cols(u8* row, int org, int w, int nbytes) {
  if (org < 0) { row[-2 - org] = row[2 - org]; row[-1 - org] = row[1 - org]; }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int c = w + j - org;
    if (c >= 0 && c < nbytes) row[c] = row[w - 2 - j - org];
  }
}

__global__ __launch_bounds__(256) void pyr3_kernel(Pyr3Args A) {
  __shared__ __attribute__((aligned(16))) unsigned s0[P3_S0_ROWS * P3_S0_PD];
  __shared__ unsigned t1[P3_T1_ROWS * P3_T1_PD];
  __shared__ unsigned t2[P3_T2_ROWS * P3_T2_PD];
  int bx, by, bz;
  if (!xcd_tile(A.tg, bx, by, bz)) return;
  const int tid = threadIdx.x;
  const u8* sp = A.src + (size_t)bz * A.src_stride;
  const int o1x = 64 * bx - 6, o1y = 64 * by - 6, o2x = 32 * bx - 2, o2y = 32 * by - 2;
  const int x0 = 128 * bx - 16, y0 = 128 * by - 14;   // level-0 tile origin
  // ---- level 0 -> LDS: chunk i (16 bytes) = tile row i / 10, bytes 16 (i % 10) ..; rows by reflect-101, chunks clamped
  // into the row (what a clamped chunk holds is either never used or rewritten by the column fix-up)
  {
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int it = 0; it < (P3_S0_ROWS * 10 + 255) / 256; it++) {
      const int i0 = (it * 4 + wave) * 64, i = i0 + lane;
      if (i < P3_S0_ROWS * 10) {
        const int row = __umul24(i, 6554) >> 16, k = i - row * 10;   // i / 10 for i < 2^14
        const int gy = d_reflect101(y0 + row, A.h0);
        int gx = x0 + 16 * k;
        gx = gx < 0 ? 0 : (gx + 16 > A.p0 ? A.p0 - 16 : gx);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sp + (size_t)__umul24(gy, A.p0) + gx),
                                         (__attribute__((address_space(3))) void*)(s0 + i0 * 4), 16, 0, 0);
      }
    }
  }
  __syncthreads();   // the fence of the barrier waits for the DMA (vmcnt(0))
  if (x0 < 0 || x0 + 4 * P3_S0_PD > A.w0) {   // block-uniform
    if (tid < P3_S0_ROWS) pd_fix_cols((u8*)s0 + tid * (4 * P3_S0_PD), x0, A.w0, 4 * P3_S0_PD);
    __syncthreads();
  }
  // ---- level 1: 37 pair-columns x 6 strips of 13 rows (73 x 73 region) -> t1 --------------------------------------------
  if (tid < 37 * 6) {
    const int seg = __umul24(tid, 1772) >> 16, p = tid - seg * 37;   // tid / 37 for tid < 2^11
    unsigned short* out = (unsigned short*)t1 + 1 + p;               // byte index 2 + 2 p of a row
    pd_strip<13, P3_S0_PD>(s0, p, 13 * seg, [](int r) { return r < P3_S0_ROWS ? r : P3_S0_ROWS - 1; },
                           [&](int j, unsigned v) { if (j < P3_T1_ROWS) out[j * (2 * P3_T1_PD)] = (unsigned short)v; });
  }
  __syncthreads();
  if (o1x < 0 || o1x - 2 + 4 * P3_T1_PD > A.w1) {
    if (tid < P3_T1_ROWS) pd_fix_cols((u8*)t1 + tid * (4 * P3_T1_PD), o1x - 2, A.w1, 4 * P3_T1_PD);
    __syncthreads();
  }
  // ---- level 1 -> HBM (own 64 x 64: 16 bytes per thread); level 2: 18 pair-columns x 14 strips of 3 rows -> t2 -----------------
  {
    const int row = tid >> 2, q = tid & 3;
    const int y = 64 * by + row, x = 64 * bx + 16 * q;
    if (y < A.h1 && x < A.lim1) {
      const unsigned* r = t1 + (6 + row) * P3_T1_PD + 2 + 4 * q;   // byte index 8 = column 64 bx
      *(uint4*)(A.l1 + (size_t)bz * A.dst_stride + (size_t)__umul24(y, A.p1) + x) = make_uint4(r[0], r[1], r[2], r[3]);
    }
  }
  if (tid < 18 * 14) {
    const int seg = __umul24(tid, 3641) >> 16, p = tid - seg * 18;   // tid / 18
    unsigned short* out = (unsigned short*)t2 + 1 + p;
    const int h1 = A.h1;
    pd_strip<3, P3_T1_PD>(t1, p, 3 * seg,
                          [&](int r) { int j = d_reflect101(o1y + r, h1) - o1y; return j < 0 ? 0 : (j < P3_T1_ROWS ? j : P3_T1_ROWS - 1); },
                          [&](int j, unsigned v) { if (j < P3_T2_ROWS) out[j * (2 * P3_T2_PD)] = (unsigned short)v; });
  }
  __syncthreads();
  if (o2x < 0 || o2x - 2 + 4 * P3_T2_PD > A.w2) {
    if (tid < P3_T2_ROWS) pd_fix_cols((u8*)t2 + tid * (4 * P3_T2_PD), o2x - 2, A.w2, 4 * P3_T2_PD);
    __syncthreads();
  }
  // ---- level 2 -> HBM (own 32 x 32); level 3: 8 pair-columns x 16 rows straight to HBM ---------------------------------------
  if (tid < 64) {
    const int row = tid >> 1, q = tid & 1;
    const int y = 32 * by + row, x = 32 * bx + 16 * q;
    if (y < A.h2 && x < A.lim2) {
      const unsigned* r = t2 + (2 + row) * P3_T2_PD + 1 + 4 * q;   // byte index 4 = column 32 bx
      *(uint4*)(A.l2 + (size_t)bz * A.dst_stride + (size_t)__umul24(y, A.p2) + x) = make_uint4(r[0], r[1], r[2], r[3]);
    }
  } else if (tid < 64 + 128) {
    const int t = tid - 64, row = t >> 3, p = t & 7;
    const int y = 16 * by + row, x = 16 * bx + 2 * p;
    const int h2 = A.h2;
    u8* dst = A.l3 + (size_t)bz * A.dst_stride;
    const int w3 = A.w3, h3 = A.h3, p3 = A.p3;
    pd_strip<1, P3_T2_PD>(t2, p, row,
                          [&](int r) { int j = d_reflect101(o2y + r, h2) - o2y; return j < 0 ? 0 : (j < P3_T2_ROWS ? j : P3_T2_ROWS - 1); },
                          [&](int, unsigned v) { if (y < h3 && x < w3) *(unsigned short*)(dst + (size_t)__umul24(y, p3) + x) = (unsigned short)v; });
  }
}


// ---------------------------------------------------------------------------------------------------
// borders of the level planes
// ---------------------------------------------------------------------------------------------------
// One launch after the pyramid of a set is built: levels 1.. of every slot get the reflect-101 border that
// cv::buildOpticalFlowPyramid gives its levels and calcOpticalFlowPyrLK's windows read when they hang over the edge
// (MVO_LK_PAD rows above and below, MVO_LK_PAD columns on the left, MVO_LK_PADR on the right), so that lk_track_kernel's
// 16-byte tile loads never need their per-byte border path at these levels.  A thread writes one dword of the border: the
// top and bottom bands are full plane rows, the side bands 8 + nr dwords per image row, where the right band starts at the
// dword holding column w (its in-image bytes get their own value back).  Reads touch pixels inside the image only
// (reflected coordinates), writes only the border: no ordering between threads is needed.
struct LkBorderArgs {
  u8* img[MVO_LK_MAX_LEVELS - 1];
  int w[MVO_LK_MAX_LEVELS - 1], h[MVO_LK_MAX_LEVELS - 1], pitch[MVO_LK_MAX_LEVELS - 1];
  int first[MVO_LK_MAX_LEVELS];   // first border dword of a level in the launch's linear index; [nlev] = total
  int nlev;
  size_t img_stride;
};

__global__ __launch_bounds__(256) void lk_border_kernel(LkBorderArgs A) {
  int i = blockIdx.x * 256 + threadIdx.x, k = 0;
  if (i >= A.first[A.nlev]) return;
  while (k + 1 < A.nlev && i >= A.first[k + 1]) k++;
  i -= A.first[k];
  const int w = A.w[k], h = A.h[k], pitch = A.pitch[k];
  const int pd = (MVO_LK_PAD + w + MVO_LK_PADR) >> 2;             // dwords per plane row that are written
  const int xr = w & ~3, nr = (w + MVO_LK_PADR - xr) >> 2;        // right band: first column, dwords
  const int band = 2 * MVO_LK_PAD * pd;
  int x0, y;
  if (i < band) {                                                  // top / bottom rows
    const int r = i / pd;
    x0 = -MVO_LK_PAD + 4 * (i - r * pd);
    y = r < MVO_LK_PAD ? r - MVO_LK_PAD : h + (r - MVO_LK_PAD);
  } else {                                                         // left / right columns of the image rows
    const int per = MVO_LK_PAD / 4 + nr, j = i - band;
    y = j / per;
    const int c = j - y * per;
    x0 = c < MVO_LK_PAD / 4 ? -MVO_LK_PAD + 4 * c : xr + 4 * (c - MVO_LK_PAD / 4);
  }
  u8* img = A.img[k] + (size_t)blockIdx.y * A.img_stride;
  const u8* rp = img + (ptrdiff_t)d_reflect101(y, h) * pitch;
  unsigned v = 0;
  if (x0 >= 0 && x0 + 4 <= w) v = *(const unsigned*)(rp + x0);   // three quarters of the threads: a border row above / below the image columns
  else {
#pragma unroll
    for (int b = 0; b < 4; b++) v |= (unsigned)rp[d_reflect101(x0 + b, w)] << (8 * b);
  }
  *(unsigned*)(img + (ptrdiff_t)y * pitch + x0) = v;
}

static void lk_launch_border(mvo_ctx* ctx, int set, const LkLevels& L, int nslots, hipStream_t st);

// ---------------------------------------------------------------------------------------------------
// LK tracker
// ---------------------------------------------------------------------------------------------------
struct LkLevelDesc {
  const u8* I;  // previous image, level l (slot 0)
  const u8* J;  // next image
  size_t stride;  // bytes between slots at this level (both images): level 0 may live in the frame ring, levels 1.. in the pyramid sets
  int w, h, pitch;
  int pad;        // rows / columns of reflect-101 border around both images (0: level 0 in the frame ring; MVO_LK_PAD: a pyramid plane)
};
struct LkArgs {
  LkLevelDesc lv[MVO_LK_MAX_LEVELS];
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
  // work_slot[w] = slot << 16 | point of item w, pt_base[slot] = first item of the slot, pt_base[nslots] = item count; persistent
  // wavefronts claim four items at a time (one per DPP row) from the counter work_ctr[0] (zeroed before the launch); with
  // work_ctr null, workgroup b takes items 4 b .. 4 b + 3 and leaves (the grid covers the list).  work_slot null: points
  // blockIdx.x * 4 .. + 3 of slot blockIdx.y with the host-sized grid.
  const int* work_slot;
  const int* pt_base;
  int* work_ctr;
  int nslots;
};

#define LK_WIN 21
#define LK_IT 24          // I tile edge (WIN + 1 bilinear + 2 Scharr halo)
#define LK_IP 36          // I tile pitch: two 16-byte loads cover 24 bytes at any 4-byte phase; 9 dwords (odd: rows spread over banks)
#define LK_DT 22          // derivative tile edge
#define LK_JT 32          // J tile edge
#define LK_JP 52          // J tile pitch: three 16-byte loads cover 32 bytes at any 4-byte phase; 13 dwords
#define LK_JSLACK ((LK_JT - (LK_WIN + 1)) / 2)

typedef short lk_short2 __attribute__((ext_vector_type(2)));

// ---- four points per wavefront ------------------------------------------------------------------------------------------
// A wavefront tracks FOUR points at once, one per DPP row of 16 lanes.  Everything that is a per-point scalar in the
// algorithm (window origin, bilinear weights, the 2x2 solve, the convergence tests: about half of all instructions when
// a wavefront tracked one point) is computed by the 16 lanes of the point's row, i.e. one instruction serves four points;
// the per-pixel work is unchanged: lane l of a row owns the 7-pixel row segments s = l + 16 k (k < 4, s < 63) of the
// 21x21 window - segment s is row s / 3, columns 7 (s % 3) .. + 6 - and keeps their template in registers.  Sums run
// over a row with four DPP steps (quad_perm, quad_perm, row_half_mirror, row_mirror) and land in every lane of the row.
// Rows take their own branches (level skipped, minEig reject, iteration counts): the loops are wave-uniform with per-row
// predicates, and a row never waits for another except by sharing the instruction stream.
#define LK_G 4    // points per wavefront
#define LK_NS 4   // segments per lane

struct LkGroupLds {
  union {
    struct {
      unsigned it[LK_IT * LK_IP / 4];   // previous-image neighbourhood (rows of LK_IP bytes)
      short2 dt[LK_DT * LK_DT];         // Scharr (dx, dy)
    } s;                                // set-up of a level
    unsigned jt[LK_JT * LK_JP / 4];     // iterations: next-image search tile (rows of LK_JP bytes); the template is in registers by then
  };
  unsigned pad[20];                     // 720 dwords: the four rows of a wavefront start 16 banks apart
};

// sum over the 16 lanes of a DPP row, in every lane of the row
__device__ __forceinline__ int row_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);  // row_mirror
  return v;
}

// Exact row sum of int32 partials whose total may overflow 32 bits: the first PLAIN butterfly steps are known not to
// overflow (bounds at the call sites: |J - I| <= 8160 and |Ix|, |Iy| <= 4080 - u8 image, 14-bit weights, 5 extra fractional
// bits, Scharr taps sum to 16 - 28 pixels per lane), the rest run on a signed high part and an unsigned 16-bit low part.
// hi * 65536 + lo is exact in a double (|sum| < 2^47), so is the product with cn the callers form, and v_cvt_f32_f64 rounds
// to nearest even exactly like the int64 -> float conversion of the reference arithmetic.
template <int PLAIN>
__device__ __forceinline__ double row_sum_exact_bounded(int v) {
  if (PLAIN >= 1) v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
  if (PLAIN >= 2) v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
  int lo = v & 0xFFFF, hi = v >> 16;
  if (PLAIN < 1) {
    lo += __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
    hi += __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
  }
  if (PLAIN < 2) {
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true);
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true);
  }
  lo += __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true);
  hi += __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true);
  lo += __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true);
  hi += __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true);
  return (double)hi * 65536.0 + (double)lo;
}

__device__ __forceinline__ void lk_weights(float a, float b, int& w00, int& w01, int& w10, int& w11) {
  const int W_BITS = 14;
  w00 = d_cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
  w01 = d_cv_round(a * (1.f - b) * (1 << W_BITS));
  w10 = d_cv_round((1.f - a) * b * (1 << W_BITS));
  w11 = (1 << W_BITS) - w00 - w01 - w10;
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// Stage ROWS x TW bytes of the image starting at (x0, y0) into LDS rows of PD dwords with the 16 lanes of a row.  Fast
// path: the tile lies inside the image -> NX4 aligned 16-byte loads per tile row (a lane address on the texture path costs
// the same for 4 or 16 bytes), the tile starts `shift` bytes into each LDS row.  Border path: one dword of a tile row per
// item, the image row by reflect-101, the four columns from two aligned dwords when they lie inside the image and per
// byte by reflect-101 otherwise; the tile then starts at byte 0.  Returns the byte shift (uniform over the row).
// `pad`: rows / columns of reflect-101 border the plane carries around the image (the fast path then covers every tile the
// tracker can ask for: window origins lie in [-21, w) x [-21, h), search tiles 5 further out).
template <int ROWS, int TW, int PD, int NX4>
__device__ __forceinline__ int lk_load_tile16(unsigned* lds, const u8* __restrict__ img, int w, int h, int pitch, int pad, int x0, int y0, int l) {
  const int xa = x0 & ~3;
  const bool inside = x0 >= -pad && y0 >= -pad && x0 + TW <= w + pad && y0 + ROWS <= h + pad && xa + pad + 16 * NX4 <= pitch;
  if (inside) {
    const u8* base = img + (size_t)y0 * pitch + xa;
#pragma unroll
    for (int i0 = 0; i0 < ROWS * NX4; i0 += 16) {
      const int i = i0 + l;
      const int row = i / NX4, k = i - row * NX4;
      const uint4 v = *(const uint4*)(base + (size_t)row * pitch + 16 * k);
      unsigned* d = lds + row * PD + 4 * k;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    return x0 - xa;
  }
  constexpr int ND = TW / 4;
  for (int i = l; i < ROWS * ND; i += 16) {
    const int row = i / ND, dc = i - row * ND;
    const int gy = d_reflect101(y0 + row, h), gx = x0 + 4 * dc;
    const u8* rp = img + (size_t)gy * pitch;
    const int a = gx & ~3;
    unsigned v;
    if (gx >= 0 && gx + 3 < w && a + 8 <= pitch) {
      const unsigned d0 = *(const unsigned*)(rp + a), d1 = *(const unsigned*)(rp + a + 4);
      v = __builtin_amdgcn_alignbyte(d1, d0, gx & 3);
    } else {
      v = 0u;
#pragma unroll
      for (int b = 0; b < 4; b++) v |= (unsigned)rp[d_reflect101(gx + b, w)] << (8 * b);
    }
    lds[row * PD + dc] = v;
  }
  return 0;
}

// (a.lo * b.lo + a.hi * b.hi) + c on packed signed 16-bit pairs, three-operand form (no accumulator move)
__device__ __forceinline__ int lk_dot2(unsigned a, unsigned b, int c) {
  int r;
  asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// low halves of two registers as one packed pair (lo, hi)
__device__ __forceinline__ unsigned lk_pack16(int lo, int hi) { return __builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x05040100u); }

// One segment's 7 values of (bilinear J) - I with 5 fractional bits: two unaligned 8-byte row spans are read as 3 dwords
// each and re-aligned with v_alignbyte.  v_perm builds the eight VERTICAL pairs V_k = (row0[k], row1[k]) as 16-bit lanes -
// neighbouring pixels share them - and two v_dot2 per pixel, V_k . (w00, w10) + V_k+1 . (w01, w11), evaluate the 4-tap
// fixed-point bilinear sum exactly.  Wa = w00 | w10 << 16, Wb = w01 | w11 << 16; Iv[k] = 256 - 512 * I starts the
// accumulator: ((J + 256) >> 9) - I == (J + 256 - 512 * I) >> 9 exactly.
__device__ __forceinline__ void lk_seg_diff(const unsigned* jt, int byte_off, unsigned Wa, unsigned Wb, const int* Iv, int* diff) {
  const int sh = byte_off & 3;
  const unsigned* q = jt + (byte_off >> 2);
  const unsigned a0 = q[0], a1 = q[1], a2 = q[2];
  const unsigned b0 = q[LK_JP / 4], b1 = q[LK_JP / 4 + 1], b2 = q[LK_JP / 4 + 2];
  const unsigned r0lo = __builtin_amdgcn_alignbyte(a1, a0, sh), r0hi = __builtin_amdgcn_alignbyte(a2, a1, sh);
  const unsigned r1lo = __builtin_amdgcn_alignbyte(b1, b0, sh), r1hi = __builtin_amdgcn_alignbyte(b2, b1, sh);
  unsigned V[8];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const unsigned sel = (unsigned)k | (0x0Cu << 8) | ((unsigned)(4 + k) << 16) | (0x0Cu << 24);   // (src1 byte k, src0 byte k)
    V[k] = __builtin_amdgcn_perm(r1lo, r0lo, sel);
    V[4 + k] = __builtin_amdgcn_perm(r1hi, r0hi, sel);
  }
#pragma unroll
  for (int k = 0; k < 7; k++) diff[k] = lk_dot2(V[k + 1], Wb, lk_dot2(V[k], Wa, Iv[k])) >> 9;
}

// `S` is the row's LDS block, `l` the lane within the row, (slot, p) the row's point; a row without a point has valid = false.
// True-colour tracking (COLOUR): rows 0..2 of the wavefront hold the three channel planes of ONE point (image slot `islot` =
// channel, the row's geometry, weights and iteration are the same in all three), and every normal-equation sum is the sum
// over the channels - cv::calcOpticalFlowPyrLK walks a window row as cn * winSize.width interleaved elements.  A row's sum
// is the same exact integer (as a double) in each of its 16 lanes: the three are added (exact below 2^53) and every lane gets
// the total; row 3 idles.
template <bool COLOUR>
__device__ __forceinline__ double lk_over_channels(double v) {
  if (!COLOUR) return v;
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const double a = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double b = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double c = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  return a + b + c;
}

// `slot` indexes the points, `islot` the images (the same except for true colour, where islot is the channel plane).
template <bool COLOUR>
__device__ __forceinline__ void lk_track_group(const LkArgs& A, LkGroupLds& S, const int slot, const int islot, const int p, const bool valid,
                                               const int l) {
  const size_t pidx = valid ? (size_t)slot * A.maxpts + p : 0;
  const float ptx = A.prev_pts[2 * pidx], pty = A.prev_pts[2 * pidx + 1];
  const float FLT_SCALE = 1.f / (1 << 20);
  const float half = (LK_WIN - 1) * 0.5f;
  const double cnd = COLOUR ? 1.0 : (double)A.cn;   // one plane stands for cn identical channels; three planes are the channels
  const bool last_active = l < 15;   // segment 63 (k = 3, l = 15) does not exist

  int status = 1;
  float errv = 0.f;
  float sx = 0.f, sy = 0.f;  // nextPts[ptidx] as stored by OpenCV between levels
#ifdef LK_ITER_STATS   // experiment builds only (tools/build_variant.sh): err = own iterations + 100 * loop trips of the wavefront + 10000 * search-tile loads
  int st_own = 0, st_trips = 0, st_jl = 0;
#endif

  for (int level = A.nlevels - 1; level >= 0; level--) {
    const LkLevelDesc lv = A.lv[level];
    // Row and first column of the lane's segments, from an opaque copy of the lane index: computed here they cost 12
    // instructions per level; computed once per wavefront the compiler hoists every LDS offset derived from them out of the
    // level loop as well and spills them (scratch traffic of ~1.5 KB per point, PMC).
    int lq = l;
    asm volatile("" : "+v"(lq));
    int sr[LK_NS], sc[LK_NS];
#pragma unroll
    for (int k = 0; k < LK_NS; k++) {
      const int s = lq + 16 * k;
      sr[k] = __umul24(s, 21846) >> 16;   // s / 3 for s < 2^15
      sc[k] = (s - 3 * sr[k]) * 7;
    }
    const u8* I = lv.I + (size_t)islot * lv.stride;
    const u8* J = lv.J + (size_t)islot * lv.stride;
    bool go = valid;
    float px = ptx * (float)(1. / (1 << level));
    float py = pty * (float)(1. / (1 << level));
    float nx, ny;
    if (level == A.nlevels - 1) { nx = px; ny = py; }
    else { nx = sx * 2.f; ny = sy * 2.f; }
    sx = nx; sy = ny;
    px -= half; py -= half;
    const int ipx = d_cv_floor(px), ipy = d_cv_floor(py);
    if (ipx < -LK_WIN || ipx >= lv.w || ipy < -LK_WIN || ipy >= lv.h) {
      if (level == 0) { status = 0; errv = 0.f; }
      go = false;
    }
    // ---- stage the 24x24 neighbourhood of I (origin ipx-1, ipy-1) -------------------------------------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the search tile of the previous level shares this memory
    int shI = 0;
    if (go) shI = lk_load_tile16<LK_IT, LK_IT, LK_IP / 4, 2>(S.s.it, I, lv.w, lv.h, lv.pitch, lv.pad, ipx - 1, ipy - 1, l);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- Scharr field on the 22x22 window-source positions (zero outside the image) --------------
    // Separable form, 11 pixels per item (22 rows x 2 halves = 44 items over the 16 lanes): with S = 3*(top + bottom) +
    // 10*mid and D = bottom - top per column, dx = S[c+1] - S[c-1] and dy = 3*(D[c-1] + D[c+1]) + 10*D[c].  Three tile
    // rows of 13 bytes come in as 4 aligned dwords each and are re-phased with v_alignbyte.
    if (go) {
#pragma unroll 1
      for (int k = 0; k < 3; k++) {
        const int item = l + 16 * k;
        if (item < 2 * LK_DT) {
          const int ty = item >> 1, hx = item & 1, tx0 = 11 * hx;
          const int bo = shI + tx0, ph = bo & 3;
          const unsigned* rp = S.s.it + ty * (LK_IP / 4) + (bo >> 2);
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
          const bool all_ok = row_ok && gx0 >= 0 && gx0 + 10 < lv.w;   // nearly always: the window lies inside the image
          unsigned* dst = (unsigned*)&S.s.dt[ty * LK_DT + tx0];
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
            if (!all_ok) {
              if (!(row_ok && (unsigned)(gx0 + 2 * j) < (unsigned)lv.w)) e0 = 0u;
              if (!(row_ok && (unsigned)(gx0 + 2 * j + 1) < (unsigned)lv.w)) e1 = 0u;
            }
            dst[2 * j] = e0;
            if (j < 5) dst[2 * j + 1] = e1;   // output 11 does not exist
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- template in registers + exact A sums ---------------------------------------------------------
    // Same machinery as the iteration: vertical 16-bit pairs (row r, row r + 1) by v_perm, shared by neighbouring
    // pixels, and the 4-tap fixed-point sums by two v_dot2; the derivative field is interleaved (dx, dy), so the low /
    // high halves of two vertically adjacent entries give the dx / dy pairs.
    int Iv[LK_NS][7];
    unsigned IxP[LK_NS][4], IyP[LK_NS][4];   // (Ix, Iy) of a segment's 7 pixels as packed 16-bit pairs (0,1) (2,3) (4,5) (6,-)
    float A11 = 0.f, A12 = 0.f, A22 = 0.f, D = 1.f;
    if (go) {
      int w00, w01, w10, w11;
      lk_weights(px - ipx, py - ipy, w00, w01, w10, w11);
      const unsigned Wa = (unsigned)w00 | ((unsigned)w10 << 16), Wb = (unsigned)w01 | ((unsigned)w11 << 16);
      int a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
      for (int k = 0; k < LK_NS; k++) {
        if (k < LK_NS - 1 || last_active) {
          {
            const int bo = (sr[k] + 1) * LK_IP + sc[k] + 1 + shI, sh = bo & 3;
            const unsigned* q = S.s.it + (bo >> 2);
            const unsigned a0 = q[0], a1 = q[1], a2 = q[2];
            const unsigned b0 = q[LK_IP / 4], b1 = q[LK_IP / 4 + 1], b2 = q[LK_IP / 4 + 2];
            const unsigned r0lo = __builtin_amdgcn_alignbyte(a1, a0, sh), r0hi = __builtin_amdgcn_alignbyte(a2, a1, sh);
            const unsigned r1lo = __builtin_amdgcn_alignbyte(b1, b0, sh), r1hi = __builtin_amdgcn_alignbyte(b2, b1, sh);
            unsigned V[8];
#pragma unroll
            for (int i = 0; i < 4; i++) {
              const unsigned sel = (unsigned)i | (0x0Cu << 8) | ((unsigned)(4 + i) << 16) | (0x0Cu << 24);
              V[i] = __builtin_amdgcn_perm(r1lo, r0lo, sel);
              V[4 + i] = __builtin_amdgcn_perm(r1hi, r0hi, sel);
            }
#pragma unroll
            for (int i = 0; i < 7; i++)   // kept as the start value of the iteration's accumulator (lk_seg_diff)
              // 256 - (descale(sum, 14 - 5) << 9) with the rounding constant as the accumulator's start value: (x >> 9) << 9 == x & ~511
              Iv[k][i] = 256 - (lk_dot2(V[i + 1], Wb, lk_dot2(V[i], Wa, 1 << (14 - 5 - 1))) & ~511);
          }
          const unsigned* d0 = (const unsigned*)&S.s.dt[sr[k] * LK_DT + sc[k]];
          const unsigned* d1 = d0 + LK_DT;
          unsigned XV[8], YV[8];   // (dx, dx below), (dy, dy below) at positions sc .. sc + 7
#pragma unroll
          for (int i = 0; i < 8; i++) {
            const unsigned e0 = d0[i], e1 = d1[i];
            XV[i] = __builtin_amdgcn_perm(e1, e0, 0x05040100u);
            YV[i] = __builtin_amdgcn_perm(e1, e0, 0x07060302u);
          }
          int ixv[7], iyv[7];
#pragma unroll
          for (int i = 0; i < 7; i++) {
            ixv[i] = lk_dot2(XV[i + 1], Wb, lk_dot2(XV[i], Wa, 1 << 13)) >> 14;   // descale(sum, 14), the rounding constant riding in the accumulator
            iyv[i] = lk_dot2(YV[i + 1], Wb, lk_dot2(YV[i], Wa, 1 << 13)) >> 14;
          }
#pragma unroll
          for (int j = 0; j < 3; j++) { IxP[k][j] = lk_pack16(ixv[2 * j], ixv[2 * j + 1]); IyP[k][j] = lk_pack16(iyv[2 * j], iyv[2 * j + 1]); }
          IxP[k][3] = (unsigned)ixv[6] & 0xFFFFu; IyP[k][3] = (unsigned)iyv[6] & 0xFFFFu;
#pragma unroll
          for (int j = 0; j < 4; j++) {   // sum(ix * ix) etc. two pixels per instruction; exact integers either way
            a11 = lk_dot2(IxP[k][j], IxP[k][j], a11); a12 = lk_dot2(IxP[k][j], IyP[k][j], a12); a22 = lk_dot2(IyP[k][j], IyP[k][j], a22);
          }
        }
      }
      // per lane: 28 * 4080^2 < 2^29 -> two plain butterfly steps
      const double sA11 = lk_over_channels<COLOUR>(row_sum_exact_bounded<2>(a11)), sA12 = lk_over_channels<COLOUR>(row_sum_exact_bounded<2>(a12)),
                   sA22 = lk_over_channels<COLOUR>(row_sum_exact_bounded<2>(a22));
      A11 = (float)(sA11 * cnd) * FLT_SCALE;
      A12 = (float)(sA12 * cnd) * FLT_SCALE;
      A22 = (float)(sA22 * cnd) * FLT_SCALE;
      D = A11 * A22 - A12 * A12;
      const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * LK_WIN * LK_WIN);
      if ((double)minEig < A.min_eig || D < 1.1920928955078125e-07f) {
        if (level == 0) status = 0;
        go = false;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the search tile overwrites the set-up tiles from here on
    D = 1.f / D;
    nx -= half; ny -= half;
    float pdx = 0.f, pdy = 0.f;
    int jx0 = 0, jy0 = 0, shJ = 0;
    bool have_tile = false;
    bool iter = go;
    for (int j = 0; j < A.max_count; j++) {
      if (!__any(iter)) break;
#ifdef LK_ITER_STATS
      st_trips++; st_own += iter ? 1 : 0;
#endif
      if (iter) {
        const int inx = d_cv_floor(nx), iny = d_cv_floor(ny);
        if (inx < -LK_WIN || inx >= lv.w || iny < -LK_WIN || iny >= lv.h) {
          if (level == 0) status = 0;
          iter = false;
        } else {
          int ddx = inx - jx0, ddy = iny - jy0;
          if (!have_tile || ddx < 0 || ddx > LK_JT - (LK_WIN + 1) || ddy < 0 || ddy > LK_JT - (LK_WIN + 1)) {
            jx0 = inx - LK_JSLACK; jy0 = iny - LK_JSLACK;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            shJ = lk_load_tile16<LK_JT, LK_JT, LK_JP / 4, 3>(S.jt, J, lv.w, lv.h, lv.pitch, lv.pad, jx0, jy0, l);
#ifdef LK_ITER_STATS
            st_jl++;
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            have_tile = true;
            ddx = LK_JSLACK; ddy = LK_JSLACK;
          }
          int w00, w01, w10, w11;
          lk_weights(nx - inx, ny - iny, w00, w01, w10, w11);
          const unsigned Wa = (unsigned)w00 | ((unsigned)w10 << 16), Wb = (unsigned)w01 | ((unsigned)w11 << 16);
          const int jbase = ddy * LK_JP + ddx + shJ;
          int s1 = 0, s2 = 0;
#pragma unroll
          for (int k = 0; k < LK_NS; k++) {
            if (k < LK_NS - 1 || last_active) {
              int diff[7];
              lk_seg_diff(S.jt, sr[k] * LK_JP + sc[k] + jbase, Wa, Wb, Iv[k], diff);
              // differences packed like the derivatives: one v_dot2 accumulates two pixels of sum(diff * Ix)
              const unsigned d01 = lk_pack16(diff[0], diff[1]), d23 = lk_pack16(diff[2], diff[3]), d45 = lk_pack16(diff[4], diff[5]);
              const unsigned d6 = (unsigned)diff[6] & 0xFFFFu;
              s1 = lk_dot2(d01, IxP[k][0], lk_dot2(d23, IxP[k][1], lk_dot2(d45, IxP[k][2], lk_dot2(d6, IxP[k][3], s1))));
              s2 = lk_dot2(d01, IyP[k][0], lk_dot2(d23, IyP[k][1], lk_dot2(d45, IyP[k][2], lk_dot2(d6, IyP[k][3], s2))));
            }
          }
          // per lane: 28 * 8160 * 4080 < 2^30 -> one plain butterfly step
          const double sb1 = lk_over_channels<COLOUR>(row_sum_exact_bounded<1>(s1)), sb2 = lk_over_channels<COLOUR>(row_sum_exact_bounded<1>(s2));
          const float b1 = (float)(sb1 * cnd) * FLT_SCALE;
          const float b2 = (float)(sb2 * cnd) * FLT_SCALE;
          const float dx = (A12 * b2 - A22 * b1) * D;
          const float dy = (A12 * b1 - A11 * b2) * D;
          nx += dx; ny += dy;
          sx = nx + half; sy = ny + half;
          if ((double)dx * dx + (double)dy * dy <= A.eps2) iter = false;
          else if (j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
            sx -= dx * 0.5f;
            sy -= dy * 0.5f;
            iter = false;
          }
          pdx = dx; pdy = dy;
        }
      }
    }
    if (go && status && level == 0) {
      const float ex = sx - half, ey = sy - half;
      const int inx = d_cv_floor(ex), iny = d_cv_floor(ey);
      if (inx < -LK_WIN || inx >= lv.w || iny < -LK_WIN || iny >= lv.h) {
        status = 0;
      } else {
        int ddx = inx - jx0, ddy = iny - jy0;
        if (!have_tile || ddx < 0 || ddx > LK_JT - (LK_WIN + 1) || ddy < 0 || ddy > LK_JT - (LK_WIN + 1)) {
          jx0 = inx - LK_JSLACK; jy0 = iny - LK_JSLACK;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          shJ = lk_load_tile16<LK_JT, LK_JT, LK_JP / 4, 3>(S.jt, J, lv.w, lv.h, lv.pitch, lv.pad, jx0, jy0, l);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          ddx = LK_JSLACK; ddy = LK_JSLACK;
        }
        int w00, w01, w10, w11;
        lk_weights(ex - inx, ey - iny, w00, w01, w10, w11);
        const unsigned Wa = (unsigned)w00 | ((unsigned)w10 << 16), Wb = (unsigned)w01 | ((unsigned)w11 << 16);
        const int jbase = ddy * LK_JP + ddx + shJ;
        int s1 = 0;
#pragma unroll
        for (int k = 0; k < LK_NS; k++) {
          if (k < LK_NS - 1 || last_active) {
            int diff[7];
            lk_seg_diff(S.jt, sr[k] * LK_JP + sc[k] + jbase, Wa, Wb, Iv[k], diff);
#pragma unroll
            for (int i = 0; i < 7; i++) s1 += abs(diff[i]);
          }
        }
        const double se = lk_over_channels<COLOUR>((double)row_sum_i32(s1));  // 16 * 28 * 8160 fits 32 bits
        const float errval = (float)(se * cnd);
        errv = errval * 1.f / (float)(32 * LK_WIN * A.cn * LK_WIN);
      }
    }
  }
  if (valid && l == 0 && (!COLOUR || islot == 0)) {
    int slot_o = slot;                 // (opaque: the output addresses are formed here, not kept in registers - or scratch - from the top)
    asm volatile("" : "+v"(slot_o));
    const size_t po = (size_t)slot_o * A.maxpts + p;
    A.next_pts[2 * po] = sx;
    A.next_pts[2 * po + 1] = sy;
    A.status[po] = (u8)status;
    A.err[po] = errv;
#ifdef LK_ITER_STATS
    A.err[po] = (float)(st_own + 100 * st_trips + 10000 * st_jl);
#endif
  }
}

// One wavefront per workgroup, 11.25 KB of LDS.  Registers: four templates of 15 registers per lane plus the per-point
// scalars and a segment's working set need ~170 VGPRs -> 3 wavefronts per SIMD (12 points in flight per SIMD).  Measured
// (1 x 256 streams, 508 k points): 3 per SIMD 3.20 ms, 2 per SIMD (205 VGPRs) 3.53 ms, 4 per SIMD (128 VGPRs, spills) 4.64 ms;
// one point per wavefront at 6 per SIMD was 3.90 ms.
#ifndef LK_WAVES_PER_EU
#define LK_WAVES_PER_EU 3
#endif
#ifndef LK_PRIO
#define LK_PRIO 1
#endif
#ifndef LK_RESIDENT_PER_CU
#define LK_RESIDENT_PER_CU (4 * LK_WAVES_PER_EU)
#endif
__global__ __launch_bounds__(64, LK_WAVES_PER_EU) void lk_track_kernel(LkArgs A) {
  __shared__ __attribute__((aligned(16))) LkGroupLds lds[LK_G];
#if LK_PRIO > 0
  __builtin_amdgcn_s_setprio(LK_PRIO);   // issue priority over other contexts' image kernels that share the SIMD (the RANSAC chains use 3)
#endif
  const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
  LkGroupLds& S = lds[g];
  // One inlined copy of lk_track_group serves the three launch forms: persistent wavefronts claiming groups of four items
  // from work_ctr, one workgroup per group of the list (work_ctr null), and the host-sized grid of the per-call API.
  const bool list = A.work_slot != nullptr, persistent = list && A.work_ctr != nullptr;
  const int total = list ? min(max(A.pt_base[A.nslots], 0), A.nslots * A.maxpts) : 0;
  int next = blockIdx.x * LK_G;   // non-persistent list form: this workgroup's group; the LAST workgroup goes on to any groups
                                  // beyond the grid (a caller's bound on the list that turned out too small costs time, not points)
  for (;;) {
    int slot, p;
    bool valid;
    if (list) {
      int w0 = next;
      if (persistent) {
        if (lane == 0) w0 = atomicAdd(A.work_ctr, LK_G);
        w0 = __builtin_amdgcn_readfirstlane(w0);
      }
      if (w0 >= total) return;
      const int w = w0 + g;
      valid = w < total;
      const int item = valid ? A.work_slot[w] : 0;   // slot << 16 | point (the slot's items in an order of the caller's choice)
      slot = min(max(item >> 16, 0), A.nslots - 1);
      p = item & 0xFFFF;
      valid = valid && p < A.maxpts;   // always true for a consistent list
    } else {
      slot = blockIdx.y;
      p = blockIdx.x * LK_G + g;
      valid = p < min(A.npts[slot], A.maxpts);
      if (!__any(valid)) return;
    }
    lk_track_group<false>(A, S, slot, slot, p, valid, l);
    if (!persistent) {
      if (!list || blockIdx.x != gridDim.x - 1) return;
      next += LK_G;
    }
  }
}

// True-colour pair (per-call API): one point per wavefront, its three channel planes on DPP rows 0..2 (image slot = channel).
__global__ __launch_bounds__(64, LK_WAVES_PER_EU) void lk_track_colour_kernel(LkArgs A) {
  __shared__ __attribute__((aligned(16))) LkGroupLds lds[LK_G];
  const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
  const int p = blockIdx.x;
  if (p >= min(A.npts[0], A.maxpts)) return;
  lk_track_group<true>(A, lds[g], 0, g < 3 ? g : 0, p, g < 3, l);
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static ImgSet lk_imgset(mvo_ctx* ctx, int set, const LkLevels& L, int level) {
  ImgSet s;
  s.w = L.w[level]; s.h = L.h[level]; s.pitch = L.pitch[level];
  if (level == 0) {   // the per-call API's own level 0: slot 0 only (the frame-batch tracker reads level 0 in the frame ring)
    s.base = ctx->lk_l0[set];
    s.slot_stride = 0;
  } else {   // pixel (0, 0) of the bordered plane
    s.base = ctx->lk_mem[set] + ctx->lk_level_off[level] + (size_t)MVO_LK_PAD * s.pitch + MVO_LK_PAD;
    s.slot_stride = ctx->lk_slot_bytes;
  }
  return s;
}
static void lk_launch_border(mvo_ctx* ctx, int set, const LkLevels& L, int nslots, hipStream_t st) {
  if (L.n < 2) return;
  LkBorderArgs A;
  memset(&A, 0, sizeof(A));
  int total = 0;
  for (int l = 1; l < L.n; l++) {
    const int k = l - 1;
    ImgSet s = lk_imgset(ctx, set, L, l);
    A.img[k] = s.base; A.w[k] = s.w; A.h[k] = s.h; A.pitch[k] = s.pitch;
    A.first[k] = total;
    const int pd = (MVO_LK_PAD + s.w + MVO_LK_PADR) >> 2, nr = (s.w + MVO_LK_PADR - (s.w & ~3)) >> 2;
    total += 2 * MVO_LK_PAD * pd + s.h * (MVO_LK_PAD / 4 + nr);
  }
  A.first[L.n - 1] = total;
  A.nlev = L.n - 1;
  A.img_stride = ctx->lk_slot_bytes;
  hipLaunchKernelGGL(lk_border_kernel, dim3((total + 255) / 256, nslots), dim3(256), 0, st, A);
}

// Build levels 1.. of pyramid set `set` for `nslots` slots.  Level 0 is `l0` (slots `l0_stride` bytes apart, e.g. an entry of
// the frame ring) or, when null, the set's own level 0 (slot 0 only: the per-call API).
int lk_build_pyramid(mvo_ctx* ctx, int set, const LkLevels& L, int nslots, hipStream_t st, const u8* l0, size_t l0_stride) {
  if (!st) st = ctx->stream;
  static const bool fused = !(getenv("MVO_PYR_FUSED") && atoi(getenv("MVO_PYR_FUSED")) == 0);
  if (fused && L.n == 4 && L.w[3] >= 8 && L.h[3] >= 8) {   // all of levels 1..3 in one launch, level 0 read once
    ImgSet s0 = lk_imgset(ctx, set, L, 0);
    if (l0) { s0.base = const_cast<u8*>(l0); s0.slot_stride = l0_stride; }
    Pyr3Args A;
    A.src = s0.base; A.src_stride = s0.slot_stride; A.w0 = L.w[0]; A.h0 = L.h[0]; A.p0 = L.pitch[0];
    A.l1 = lk_imgset(ctx, set, L, 1).base; A.l2 = lk_imgset(ctx, set, L, 2).base; A.l3 = lk_imgset(ctx, set, L, 3).base;
    A.dst_stride = ctx->lk_slot_bytes;
    A.w1 = L.w[1]; A.h1 = L.h[1]; A.p1 = L.pitch[1]; A.w2 = L.w[2]; A.h2 = L.h[2]; A.p2 = L.pitch[2];
    A.w3 = L.w[3]; A.h3 = L.h[3]; A.p3 = L.pitch[3];
    A.lim1 = align_up(L.w[1], 16); A.lim2 = align_up(L.w[2], 16);
    A.tg = TileGrid{(L.w[3] + 15) / 16, (L.h[3] + 15) / 16, nslots};
    hipLaunchKernelGGL(pyr3_kernel, dim3(xcd_grid_blocks(A.tg)), dim3(256), 0, st, A);
    lk_launch_border(ctx, set, L, nslots, st);
    return MVO_OK;
  }
  for (int l = 1; l < L.n; l++) {
    ImgSet src = lk_imgset(ctx, set, L, l - 1);
    if (l == 1 && l0) { src.base = const_cast<u8*>(l0); src.slot_stride = l0_stride; }
    launch_pyrdown(ctx, src, lk_imgset(ctx, set, L, l), nslots, st);
  }
  lk_launch_border(ctx, set, L, nslots, st);
  return MVO_OK;
}

// Track d_prev_pts -> d_next_pts for `nslots` slots between pyramid sets prev_set and cur_set; `prev_l0` / `cur_l0` (slots
// `l0_stride` bytes apart) replace the sets' level 0 when given.
int lk_track_device(mvo_ctx* ctx, int prev_set, int cur_set, const LkLevels& L, int nslots, int max_n, hipStream_t st,
                    const int* d_work_slot, const int* d_pt_base, int* d_work_ctr, const u8* prev_l0, const u8* cur_l0, size_t l0_stride,
                    int items_bound) {
  if (!st) st = ctx->stream;
  LkArgs A;
  memset(&A, 0, sizeof(A));
  A.work_slot = d_work_slot; A.pt_base = d_pt_base; A.work_ctr = d_work_ctr; A.nslots = nslots;
  for (int l = 0; l < L.n; l++) {
    ImgSet p = lk_imgset(ctx, prev_set, L, l), c = lk_imgset(ctx, cur_set, L, l);
    A.lv[l].I = p.base; A.lv[l].J = c.base; A.lv[l].stride = p.slot_stride;
    A.lv[l].w = L.w[l]; A.lv[l].h = L.h[l]; A.lv[l].pitch = L.pitch[l];
    A.lv[l].pad = l ? MVO_LK_PAD : 0;
  }
  if (prev_l0 && cur_l0) { A.lv[0].I = prev_l0; A.lv[0].J = cur_l0; A.lv[0].stride = l0_stride; }
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
    // persistent single-wavefront workgroups, as many as stay resident (256 CUs x 4 SIMDs x LK_WAVES_PER_EU)
    unsigned items = (unsigned)nslots * (unsigned)ctx->maxpts;
    if (items_bound >= 0 && (unsigned)items_bound < items) items = (unsigned)items_bound;   // the caller knows an upper bound of the list
    const unsigned want = (items + LK_G - 1) / LK_G;
    static const bool persistent = !(getenv("MVO_LK_PERSISTENT") && atoi(getenv("MVO_LK_PERSISTENT")) == 0);
    if (!persistent) {
      A.work_ctr = nullptr;   // at least one workgroup: the last one walks on past the grid should the bound have been too small
      hipLaunchKernelGGL(lk_track_kernel, dim3(want ? want : 1u), dim3(64), 0, st, A);
      return MVO_OK;
    }
    const unsigned full = 256u * LK_RESIDENT_PER_CU;
    hipLaunchKernelGGL(lk_track_kernel, dim3(want < full ? want : full), dim3(64), 0, st, A);
    return MVO_OK;
  }
  if (max_n <= 0) return MVO_OK;
  dim3 grid((max_n + LK_G - 1) / LK_G, nslots);
  hipLaunchKernelGGL(lk_track_kernel, grid, dim3(64), 0, st, A);
  return MVO_OK;
}

// True-colour pair of the per-call API: level 0 = the channel planes in lk_c0, levels 1.. = slots 0..2 of the pyramid sets;
// one single-wavefront workgroup per point (lk_track_colour_kernel).
int lk_track_colour_device(mvo_ctx* ctx, int prev_set, int cur_set, const LkLevels& L, int n) {
  LkArgs A;
  memset(&A, 0, sizeof(A));
  A.nslots = 1;
  for (int l = 0; l < L.n; l++) {
    ImgSet p = lk_imgset(ctx, prev_set, L, l), c = lk_imgset(ctx, cur_set, L, l);
    A.lv[l].I = p.base; A.lv[l].J = c.base; A.lv[l].stride = p.slot_stride;
    A.lv[l].w = L.w[l]; A.lv[l].h = L.h[l]; A.lv[l].pitch = L.pitch[l];
    A.lv[l].pad = l ? MVO_LK_PAD : 0;
  }
  A.lv[0].I = ctx->lk_c0[prev_set]; A.lv[0].J = ctx->lk_c0[cur_set]; A.lv[0].stride = ctx->lk_c0_plane;
  A.nlevels = L.n;
  A.prev_pts = ctx->d_prev_pts; A.next_pts = ctx->d_next_pts;
  A.status = ctx->d_status; A.err = ctx->d_err; A.npts = ctx->d_npts;
  A.maxpts = ctx->maxpts;
  A.cn = 3;
  int mc = ctx->cfg.lk_max_count; mc = mc < 0 ? 0 : (mc > 100 ? 100 : mc);
  double eps = ctx->cfg.lk_epsilon; eps = eps < 0 ? 0 : (eps > 10 ? 10 : eps);
  A.max_count = mc;
  A.eps2 = eps * eps;
  A.min_eig = ctx->cfg.lk_min_eig;
  if (n > 0) hipLaunchKernelGGL(lk_track_colour_kernel, dim3(n), dim3(64), 0, ctx->stream, A);
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

// Levels 1.. of the LK pyramid of one image exactly as the tracker builds them (cv::buildOpticalFlowPyramid's images: pyrDown
// chain, early stop when a level would not exceed winSize): level l goes to levels[l - 1], rows tightly packed.
extern "C" int mvo_build_lk_pyramid(mvo_ctx* ctx, const uint8_t* img, int w, int h, int stride, uint8_t* const* levels, int* n_levels) {
  if (!ctx || !img || !levels || !n_levels || w < 1 || h < 1 || w > ctx->maxw || h > ctx->maxh) return MVO_E_ARG;
  LkLevels L = lk_levels(w, h, ctx->cfg.lk_win, ctx->cfg.lk_max_level);
  ImgSet s = lk_imgset(ctx, 0, L, 0);
  int rc = upload_gray(ctx, img, w, h, stride, 1, s.base, s.pitch, 0);
  if (rc) return rc;
  lk_build_pyramid(ctx, 0, L, 1);
  for (int l = 1; l < L.n; l++) {
    if (!levels[l - 1]) return MVO_E_ARG;
    ImgSet d = lk_imgset(ctx, 0, L, l);
    MVO_HIP(hipMemcpy2DAsync(levels[l - 1], d.w, d.base, d.pitch, d.w, d.h, hipMemcpyDeviceToHost, ctx->stream));
  }
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  *n_levels = L.n;
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
  // A colour image is reduced to gray (what a replicated mono8 source needs: one plane, sums scaled by lk_channels) AND kept as
  // its three channel planes: the reference tracks on the BGR8 image (src/mono_vo.cpp:94 -> src/tracker.cpp:68), so when the
  // channels of either image differ the sums must run over all three.
  if ((rc = upload_gray(ctx, prev, w, h, stride, channels, p0.base, p0.pitch, 0, true))) return rc;
  if (channels != 1) planes_from_stage(ctx, w, h, channels, ctx->lk_c0[0], p0.pitch, ctx->lk_c0_plane);
  if ((rc = upload_gray(ctx, next, w, h, stride, channels, c0.base, c0.pitch, 0, true))) return rc;
  if (channels != 1) planes_from_stage(ctx, w, h, channels, ctx->lk_c0[1], c0.pitch, ctx->lk_c0_plane);
  int differ = 0;
  if (channels != 1 && (rc = color_channels_differ(ctx, &differ))) return rc;
  MVO_HIP(hipMemcpyAsync(ctx->d_prev_pts, prev_pts, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  int* hn = (int*)ctx->h_pin;
  hn[0] = n;
  MVO_HIP(hipMemcpyAsync(ctx->d_npts, hn, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  if (differ) {   // true colour: the channel planes are slots 0..2 of the pyramid sets
    lk_build_pyramid(ctx, 0, L, 3, nullptr, ctx->lk_c0[0], ctx->lk_c0_plane);
    lk_build_pyramid(ctx, 1, L, 3, nullptr, ctx->lk_c0[1], ctx->lk_c0_plane);
    lk_track_colour_device(ctx, 0, 1, L, n);
  } else {
    lk_build_pyramid(ctx, 0, L, 1);
    lk_build_pyramid(ctx, 1, L, 1);
    lk_track_device(ctx, 0, 1, L, 1, n);
  }
  MVO_HIP(hipMemcpyAsync(next_pts, ctx->d_next_pts, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipMemcpyAsync(status, ctx->d_status, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipMemcpyAsync(err, ctx->d_err, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  return MVO_OK;
}
