// csrc/match.hip — brute-force Hamming 2-NN + Lowe ratio for gfx950 (replaces
// cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, 2) + the ratio loop at reference
// src/feature_processor.cpp:25-41; semantics per SURVEY.md Appendix A.2).
//
// hamming_knn2_kernel: the distance matrix is a GEMM and runs on the matrix cores.  With every descriptor bit b mapped to the
// int8 value 1 - 2 b (queries) or 2 b - 1 (trains, i.e. negated), the dot product of a query and a train over the 256 bit
// positions is (#different - #equal) = 2 d - 256, d the Hamming distance: v_mfma_i32_32x32x32_i8 produces 32 trains x 32
// queries per instruction and K-slice of 32 bit positions, eight slices per descriptor.  A wavefront keeps 64 queries
// (two 32-column B operands x 8 slices = 64 VGPRs) for the whole launch; train descriptors stream through LDS in tiles of
// 32, expanded to bytes by a 256-entry table (byte -> 8 bytes) while they are staged, and are read back as the A operand
// with one 16-byte read per lane and slice (rows 272 bytes apart: conflict-free).  A lane then holds, for its query
// column, 16 of the 32 trains of the tile in its accumulator registers (row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)):
// key = d << 16 | trainIdx is one v_lshl_add_u32 from the accumulator - (acc << 15) + ((256 << 15) + trainIdx) - and the
// two smallest keys per query are kept with v_min / v_med3; lanes l and l + 32 merge their halves at the end.  The two
// smallest packed keys are exactly OpenCV's insertion order (smaller distance first, ties -> lower train index first).
// ratio_compact_kernel: ordered (query-order) compaction of the matches that pass the ratio test, with
// the comparison done in double like the reference's `float < double * float`.
#include "mvo_internal.h"

#include <type_traits>

#define MM_TT 32     // trains per LDS tile (rows of the A operand)
#define MM_RP 272    // LDS bytes per expanded train row: 256 + 16
#define MM_INVALID 0xFF000000u   // key base of a padding row: (acc << 15) + this lies in [0xFE800000, 0xFF800000], real keys end at 0x0100FFFF
typedef int mm_v4i __attribute__((ext_vector_type(4)));
typedef int mm_v16i __attribute__((ext_vector_type(16)));

// the four low bits of x as four bytes 0 / 1 (bit i -> byte i): the products b_i 2^(i + 7 j) only meet a byte's bit 0 for i = j
__device__ __forceinline__ unsigned mm_spread4(unsigned x) { return (__umul24(x & 15u, 0x00204081u)) & 0x01010101u; }
__device__ __forceinline__ unsigned mm_med3(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

__global__ __launch_bounds__(256) void hamming_knn2_kernel(const u8* __restrict__ q, const u8* __restrict__ t,
                                                           const int* __restrict__ nq_, const int* __restrict__ nt_,
                                                           unsigned* __restrict__ best, int cap,
                                                           const int* __restrict__ list, const int* __restrict__ nlist) {
  __shared__ uint2 s_lut[256];                                            // byte -> its 8 bits as bytes 0x01 (bit clear) / 0xFF (bit set)
  __shared__ __attribute__((aligned(16))) u8 s_t[2][MM_TT * MM_RP];      // two train tiles (double buffer)
  // `list` (optional): blockIdx.y indexes a device-resident list of *nlist slots (the key-frame slots of this step)
  if (list && (int)blockIdx.y >= *nlist) return;   // block-uniform
  const int slot = list ? list[blockIdx.y] : blockIdx.y;
  const int nq = min(nq_[slot], cap), nt = min(nt_[slot], cap);
  const int q0 = blockIdx.x * 256;
  if (q0 >= nq) return;  // block-uniform
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  {
    unsigned lo = mm_spread4(tid), hi = mm_spread4(tid >> 4);
    lo = ((lo << 8) - lo) | 0x01010101u;   // byte 1 -> 0xFF (-1), byte 0 -> 0x01 (+1)
    hi = ((hi << 8) - hi) | 0x01010101u;
    s_lut[tid] = make_uint2(lo, hi);
  }
  const u8* tp = t + (size_t)slot * cap * 32;
  // staging role of this thread: dword (tid & 7) of train (tid >> 3) of a tile
  const int srow = tid >> 3, sdw = tid & 7;
  unsigned nxt = (srow < nt) ? *(const unsigned*)(tp + (size_t)srow * 32 + 4 * sdw) : 0u;
  __syncthreads();   // the table
  // ---- this wavefront's 64 queries as B operands: column r of tile c is query q0 + 64 wave + 32 c + r; the lane half h
  // holds bit positions 32 kb + 16 h .. + 15 of slice kb (the A operand uses the same split, so the products pair up)
  mm_v4i B[2][8];
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const int qi = q0 + 64 * wave + 32 * c + r;
    const uint4* qp = (const uint4*)(q + ((size_t)slot * cap + (qi < nq ? qi : 0)) * 32);
    const uint4 qa = qp[0], qb = qp[1];
    const unsigned w[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
#pragma unroll
    for (int kb = 0; kb < 8; kb++) {
      const unsigned two = (w[kb] >> (16 * h)) & 0xFFFFu;
      const uint2 e0 = s_lut[two & 0xFFu], e1 = s_lut[two >> 8];
      B[c][kb] = mm_v4i{(int)e0.x, (int)e0.y, (int)e1.x, (int)e1.y};
    }
  }
  // accumulator register g of a lane is train row (g & 3) + 8 (g >> 2) + 4 h of the tile; + 256 << 15: see the header
  unsigned idx[16];
#pragma unroll
  for (int g = 0; g < 16; g++) idx[g] = (256u << 15) + (unsigned)((g & 3) + 8 * (g >> 2) + 4 * h);
  unsigned b1[2] = {0xFFFFFFFFu, 0xFFFFFFFFu}, b2[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
  int buf = 0;
  // one tile: stage it (its dword is in `nxt`), fetch the next one's dword, 16 MFMAs, 32 keys into the two running minima
  auto tile = [&](const int t0, auto last) {
    constexpr bool LAST = decltype(last)::value;   // the tile may hold fewer than MM_TT trains
    {
      // expanded and negated: bit set -> +1, clear -> -1
      u8* row = s_t[buf] + srow * MM_RP + 32 * sdw;
      const uint2 e0 = s_lut[nxt & 0xFFu], e1 = s_lut[(nxt >> 8) & 0xFFu], e2 = s_lut[(nxt >> 16) & 0xFFu], e3 = s_lut[nxt >> 24];
      const unsigned NEG = 0xFEFEFEFEu;   // 0x01 <-> 0xFF
      *(uint4*)row = make_uint4(e0.x ^ NEG, e0.y ^ NEG, e1.x ^ NEG, e1.y ^ NEG);
      *(uint4*)(row + 16) = make_uint4(e2.x ^ NEG, e2.y ^ NEG, e3.x ^ NEG, e3.y ^ NEG);
    }
    if (!LAST) {
      const int tn = t0 + MM_TT + srow;
      nxt = (tn < nt) ? *(const unsigned*)(tp + (size_t)tn * 32 + 4 * sdw) : 0u;   // in flight during the MFMAs below
    }
    __syncthreads();   // tile `buf` is complete; the other buffer's readers finished before the previous barrier
    mm_v16i acc0, acc1;
    {
      const u8* ap = s_t[buf] + r * MM_RP + 16 * h;
      mm_v4i A = *(const mm_v4i*)ap;
      const mm_v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[0][0], z, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[1][0], z, 0, 0, 0);
#pragma unroll
      for (int kb = 1; kb < 8; kb++) {
        A = *(const mm_v4i*)(ap + 32 * kb);
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[0][kb], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[1][kb], acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int g = 0; g < 16; g++) {
      unsigned base = idx[g];   // (256 << 15) + index of the train in accumulator register g
      idx[g] += MM_TT;
      // rows beyond nt hold an all-zero descriptor and must not compete: their keys land above every real one
      if (LAST && (int)(base & 0xFFFFu) >= nt) base = MM_INVALID;
      const unsigned k0 = ((unsigned)acc0[g] << 15) + base, k1 = ((unsigned)acc1[g] << 15) + base;
      b2[0] = mm_med3(b1[0], b2[0], k0); b1[0] = min(b1[0], k0);
      b2[1] = mm_med3(b1[1], b2[1], k1); b1[1] = min(b1[1], k1);
    }
    buf ^= 1;
  };
  int t0 = 0;
  for (; t0 + MM_TT <= nt; t0 += MM_TT) tile(t0, std::false_type{});
  if (t0 < nt) tile(t0, std::true_type{});
  // ---- lanes l and l + 32 saw different trains of the same query: the two smallest of the four keys
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const unsigned o1 = __shfl_xor(b1[c], 32, 64), o2 = __shfl_xor(b2[c], 32, 64);
    unsigned m1 = min(b1[c], o1), m2 = min(max(b1[c], o1), min(b2[c], o2));
    if (m1 >= MM_INVALID - (1u << 24)) m1 = 0xFFFFFFFFu;   // only padding rows: no neighbour
    if (m2 >= MM_INVALID - (1u << 24)) m2 = 0xFFFFFFFFu;
    const int qi = q0 + 64 * wave + 32 * c + r;
    if (h == 0 && qi < nq) {
      const size_t o = ((size_t)slot * cap + qi) * 2;
      best[o] = m1;
      best[o + 1] = m2;
    }
  }
}

__global__ __launch_bounds__(1024) void ratio_compact_kernel(const unsigned* __restrict__ best,
                                                             const int* __restrict__ nq_, double ratio,
                                                             mvo_match* __restrict__ out, int* __restrict__ nout,
                                                             int cap, const int* __restrict__ list, const int* __restrict__ nlist) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  if (list && (int)blockIdx.x >= *nlist) return;   // block-uniform
  const int slot = list ? list[blockIdx.x] : blockIdx.x;
  const int nq = min(nq_[slot], cap);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int q0 = 0; q0 < nq; q0 += 1024) {
    int qi = q0 + threadIdx.x;
    bool keep = false;
    unsigned b1 = 0, b2 = 0;
    if (qi < nq) {
      size_t o = ((size_t)slot * cap + qi) * 2;
      b1 = best[o]; b2 = best[o + 1];
      if (b1 != 0xFFFFFFFFu && b2 != 0xFFFFFFFFu) {  // exactly two valid neighbours (size()==2)
        float d0 = (float)(b1 >> 16), d1 = (float)(b2 >> 16);
        keep = (double)d0 < ratio * (double)d1;
      }
    }
    unsigned long long m = __ballot(keep);
    int pre = __popcll(m & ((1ull << lane) - 1));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; w++) off += s_wave[w];
    if (keep) {
      mvo_match mm;
      mm.query_idx = qi; mm.train_idx = (int)(b1 & 0xFFFF); mm.img_idx = 0; mm.distance = (float)(b1 >> 16);
      out[(size_t)slot * cap + off + pre] = mm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int w = 0; w < 16; w++) tot += s_wave[w];
      s_base += tot;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) nout[slot] = s_base;
}

int match_state_create(mvo_ctx* ctx) {
  MatchState* m = new MatchState();
  ctx->match = m;
  m->cap = ctx->maxpts;
  size_t n = (size_t)ctx->B * m->cap;
  MVO_HIP(hipMalloc(&m->d_q, n * 32));
  MVO_HIP(hipMalloc(&m->d_t, n * 32));
  MVO_HIP(hipMalloc(&m->d_best, n * 2 * sizeof(unsigned)));
  MVO_HIP(hipMalloc(&m->d_out, n * sizeof(mvo_match)));
  MVO_HIP(hipMalloc(&m->d_nout, ctx->B * sizeof(int)));
  MVO_HIP(hipMalloc(&m->d_nq, ctx->B * sizeof(int)));
  MVO_HIP(hipMalloc(&m->d_nt, ctx->B * sizeof(int)));
  return MVO_OK;
}

void match_state_destroy(mvo_ctx* ctx) {
  MatchState* m = ctx->match;
  if (!m) return;
  (void)hipFree(m->d_q); (void)hipFree(m->d_t); (void)hipFree(m->d_best); (void)hipFree(m->d_out);
  (void)hipFree(m->d_nout); (void)hipFree(m->d_nq); (void)hipFree(m->d_nt);
  delete m;
  ctx->match = nullptr;
}

// Device-side matcher over `nslots` slots: d_q/d_t/d_nq/d_nt already resident.
int match_device(mvo_ctx* ctx, int nslots, int max_nq, double ratio, const int* d_list, const int* d_nlist) {
  MatchState* m = ctx->match;
  if (max_nq > 0) {
    dim3 grid((max_nq + 255) / 256, nslots);
    hipLaunchKernelGGL(hamming_knn2_kernel, grid, dim3(256), 0, ctx->stream, m->d_q, m->d_t, m->d_nq, m->d_nt,
                       m->d_best, m->cap, d_list, d_nlist);
  }
  hipLaunchKernelGGL(ratio_compact_kernel, dim3(nslots), dim3(1024), 0, ctx->stream, m->d_best, m->d_nq, ratio,
                     m->d_out, m->d_nout, m->cap, d_list, d_nlist);
  return MVO_OK;
}

extern "C" int mvo_match_knn2_ratio(mvo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt,
                                    double ratio, mvo_match* out, int cap, int* n) {
  if (!ctx || !n || nq < 0 || nt < 0 || (nq && !q) || (nt && !t)) return MVO_E_ARG;
  *n = 0;
  if (nq == 0 || nt == 0) return MVO_OK;  // knnMatch on an empty set yields no matches
  MatchState* m = ctx->match;
  if (nq > m->cap || nt > m->cap || nt > 65535) { ctx->set_error("mvo_match_knn2_ratio: capacity"); return MVO_E_CAPACITY; }
  MVO_HIP(hipMemcpyAsync(m->d_q, q, (size_t)nq * 32, hipMemcpyHostToDevice, ctx->stream));
  MVO_HIP(hipMemcpyAsync(m->d_t, t, (size_t)nt * 32, hipMemcpyHostToDevice, ctx->stream));
  int* hn = (int*)ctx->h_pin;
  hn[0] = nq; hn[1] = nt;
  MVO_HIP(hipMemcpyAsync(m->d_nq, hn, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  MVO_HIP(hipMemcpyAsync(m->d_nt, hn + 1, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  match_device(ctx, 1, nq, ratio);
  MVO_HIP(hipMemcpyAsync(hn + 2, m->d_nout, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  int cnt = hn[2];
  *n = cnt;
  int ncopy = cnt < cap ? cnt : cap;
  if (ncopy > 0) {
    if (!out) return MVO_E_ARG;
    MVO_HIP(hipMemcpyAsync(out, m->d_out, (size_t)ncopy * sizeof(mvo_match), hipMemcpyDeviceToHost, ctx->stream));
    MVO_HIP(hipStreamSynchronize(ctx->stream));
  }
  return cnt > cap ? MVO_E_CAPACITY : MVO_OK;
}
