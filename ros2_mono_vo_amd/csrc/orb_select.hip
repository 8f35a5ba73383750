// csrc/orb_select.hip — cv::KeyPointsFilter::retainBest on the device, with the element ORDER OpenCV
// produces (features2d/src/keypoint.cpp: std::nth_element + std::partition; libstdc++ 11 semantics).
//
// ORB applies retainBest twice per pyramid level (orb.cpp computeKeyPoints: 2*quota by FAST score, then quota
// by Harris response) and the order in which the survivors come out of nth_element's partitions becomes the
// key-point order, i.e. descriptor rows, match indices and RANSAC sample indices downstream.  Replacing the
// selection by a sort would keep the same SET but permute it, so the introselect is reproduced step by step:
//
//   * one workgroup per (camera stream, pyramid level) job, the (response, candidate index) pairs in HBM/L2;
//   * every round of __introselect is one median-of-3 pivot move (lane 0) and one __unguarded_partition, and the
//     Hoare partition is data-parallel once it is written in terms of ranks: with L_k the k-th position from the
//     left where the left scan stops (!comp(x, pivot)) and R_k the k-th position from the right where the right
//     scan stops (!comp(pivot, x)), the sequential loop swaps exactly the pairs (L_k, R_k) with L_k < R_k, all of
//     them read from the untouched part of the array.  So: two block-wide prefix counts, the swapping elements go
//     through a staging buffer by rank, and the cut is L_K if that lies before R_{K-1} (else R_{K-1}), K the
//     number of swaps;
//   * the <= 3 element tail is __insertion_sort on lane 0; the depth-limit branch (__heap_select) is restated
//     sequentially on lane 0 (it needs 2*floor(log2 n) unbalanced rounds and does not occur on real score data);
//   * std::partition(first + n, last, response >= ambiguous) is the same rank-swap with a different predicate.
//
// comp(a, b) = a.response > b.response throughout (KeypointResponseGreater).
#include "mvo_internal.h"

// threads per (stream, level) job.  A job is a chain of barrier-separated passes over its candidates, each thread walking a
// contiguous chunk.  Per key-frame step of 512 streams ALONE (4096 jobs): 128 threads 2.71 ms, 256: 2.50, 512: 2.09, 1024: 2.43 -
// but beside three other contexts the 512-thread workgroup places worse (stage 1.03 against 0.81 ms, bench 99.1 against 101.7 k
// frames/s), so 256 it is; an LDS array of 2048 instead of 4096 entries (twice the workgroups per CU): 3.00 ms.
#ifndef RB_T
#define RB_T 256
#endif

struct RbShared {
  int wsumL[RB_T / 64], wsumR[RB_T / 64];
  int K, LK, bound;
};

__device__ __forceinline__ float rb_resp(const uint2& e) { return __uint_as_float(e.x); }

// exclusive prefix over the block of two ints per thread; totals returned in tL/tR.  2 barriers.
__device__ __forceinline__ void rb_block_scan2(int vl, int vr, int& pl, int& pr, int& tL, int& tR, RbShared& S) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int il = vl, ir = vr;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int a = __shfl_up(il, d, 64), b = __shfl_up(ir, d, 64);
    if (lane >= d) { il += a; ir += b; }
  }
  if (lane == 63) { S.wsumL[wave] = il; S.wsumR[wave] = ir; }
  __syncthreads();
  int bl = 0, br = 0, sl = 0, sr = 0;
#pragma unroll
  for (int w = 0; w < RB_T / 64; w++) {
    if (w < wave) { bl += S.wsumL[w]; br += S.wsumR[w]; }
    sl += S.wsumL[w]; sr += S.wsumR[w];
  }
  pl = bl + il - vl; pr = br + ir - vr;
  tL = sl; tR = sr;
  __syncthreads();
}

// One rank-swap partition of a[lo, hi).
//   HOARE = true : __unguarded_partition around pivot response pv;   returns the cut
//   HOARE = false: std::partition with pred(x) = x.response >= pv;    returns lo + #pred
template <bool HOARE>
__device__ int rb_partition(uint2* a, int lo, int hi, float pv, uint2* stL, uint2* stR, RbShared& S) {
  const int tid = threadIdx.x;
  const int r = hi - lo;
  if (r <= 0) return lo;
  const int c = (r + RB_T - 1) / RB_T;
  const int b = min(hi, lo + tid * c), e = min(hi, b + c);
  auto stop_l = [&](float x) { return HOARE ? !(x > pv) : !(x >= pv); };
  auto stop_r = [&](float x) { return HOARE ? !(pv > x) : (x >= pv); };
  int cl = 0, cr = 0;
  for (int i = b; i < e; i++) { float x = rb_resp(a[i]); cl += stop_l(x); cr += stop_r(x); }
  if (tid == 0) { S.K = 0; S.LK = -1; S.bound = hi; }
  int pl0, pr0, TL, TR;
  rb_block_scan2(cl, cr, pl0, pr0, TL, TR, S);  // (its barriers also publish the S.* resets)
  // ---- stage the elements that will move, by rank -------------------------------------------------------------
  int pl = pl0, pr = pr0, nsw = 0;
  for (int i = b; i < e; i++) {
    uint2 v = a[i];
    float x = rb_resp(v);
    bool sl = stop_l(x), sr = stop_r(x);
    int after = TR - pr - (sr ? 1 : 0);  // right-scan stops strictly after i  (= rank of i from the right if sr)
    if (sl && after > pl) { stL[lo + pl] = v; nsw++; }           // i = L_pl and L_pl < R_pl
    else if (sr && pl > after) { stR[lo + after] = v; }          // i = R_after and L_after < R_after
    pl += sl; pr += sr;
  }
  if (nsw) atomicAdd(&S.K, nsw);
  __syncthreads();
  const int K = S.K;
  pl = pl0; pr = pr0;
  for (int i = b; i < e; i++) {
    float x = rb_resp(a[i]);
    bool sl = stop_l(x), sr = stop_r(x);
    int after = TR - pr - (sr ? 1 : 0);
    if (sl && after > pl) a[i] = stR[lo + pl];
    else if (sr && pl > after) {
      a[i] = stL[lo + after];
      if (HOARE && after == K - 1) S.bound = i;  // R_{K-1}
    }
    if (HOARE && sl && pl == K) S.LK = i;        // L_K (rank K from the left, original array)
    pl += sl; pr += sr;
  }
  __syncthreads();
  int cut;
  if (HOARE) {
    const int bound = S.bound, lk = S.LK;
    cut = (lk >= 0 && lk < bound) ? lk : bound;
  } else {
    cut = lo + TR;
  }
  __syncthreads();  // S.* may be reset by the next call
  return cut;
}

// ---- the sequential corners, lane 0 only ------------------------------------------------------------------------
__device__ __forceinline__ bool rb_gt(const uint2& x, const uint2& y) { return rb_resp(x) > rb_resp(y); }
__device__ __forceinline__ void rb_swap(uint2* a, int i, int j) { uint2 t = a[i]; a[i] = a[j]; a[j] = t; }

__device__ void rb_median_to_first(uint2* a, int res, int ia, int ib, int ic) {
  if (rb_gt(a[ia], a[ib])) {
    if (rb_gt(a[ib], a[ic])) rb_swap(a, res, ib);
    else if (rb_gt(a[ia], a[ic])) rb_swap(a, res, ic);
    else rb_swap(a, res, ia);
  } else if (rb_gt(a[ia], a[ic])) rb_swap(a, res, ia);
  else if (rb_gt(a[ib], a[ic])) rb_swap(a, res, ic);
  else rb_swap(a, res, ib);
}

__device__ void rb_insertion_sort(uint2* a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    uint2 val = a[i];
    if (rb_gt(val, a[first])) {
      for (int j = i; j > first; --j) a[j] = a[j - 1];
      a[first] = val;
    } else {
      int next = i - 1, pos = i;
      while (rb_gt(val, a[next])) { a[pos] = a[next]; pos = next; --next; }
      a[pos] = val;
    }
  }
}

__device__ void rb_push_heap(uint2* f, int hole, int top, uint2 value) {
  int parent = (hole - 1) / 2;
  while (hole > top && rb_gt(f[parent], value)) { f[hole] = f[parent]; hole = parent; parent = (hole - 1) / 2; }
  f[hole] = value;
}
__device__ void rb_adjust_heap(uint2* f, int hole, int len, uint2 value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (rb_gt(f[child], f[child - 1])) child--;
    f[hole] = f[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    f[hole] = f[child - 1];
    hole = child - 1;
  }
  rb_push_heap(f, hole, top, value);
}
__device__ void rb_heap_select(uint2* a, int first, int middle, int last) {
  uint2* f = a + first;
  const int len = middle - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    for (;;) {
      uint2 v = f[parent];
      rb_adjust_heap(f, parent, len, v);
      if (parent == 0) break;
      parent--;
    }
  }
  for (int i = middle; i < last; ++i)
    if (rb_gt(a[i], a[first])) {
      uint2 v = a[i];
      a[i] = a[first];
      rb_adjust_heap(f, 0, len, v);
    }
}

// KeyPointsFilter::retainBest on a[0, cnt): returns the new count (identical in every thread).
// depth_limit < 0: libstdc++'s 2 * floor(log2(cnt)).
__device__ int rb_retain_best(uint2* a, int cnt, int n, uint2* stL, uint2* stR, RbShared& S, int depth_limit) {
  if (!(n >= 0 && cnt > n)) return cnt;
  if (n == 0) return 0;
  const int tid = threadIdx.x;
  int first = 0, last = cnt;
  const int nth = n - 1;
  int depth = depth_limit >= 0 ? depth_limit : 2 * (31 - __clz(cnt));
  bool heap_done = false;
  while (last - first > 3) {
    if (depth == 0) {
      if (tid == 0) { rb_heap_select(a, first, nth + 1, last); rb_swap(a, first, nth); }
      __syncthreads();
      heap_done = true;
      break;
    }
    --depth;
    if (tid == 0) rb_median_to_first(a, first, first + 1, first + (last - first) / 2, last - 1);
    __syncthreads();
    const float pv = rb_resp(a[first]);
    const int cut = rb_partition<true>(a, first + 1, last, pv, stL, stR, S);
    if (cut <= nth) first = cut; else last = cut;
  }
  if (!heap_done) {
    if (tid == 0) rb_insertion_sort(a, first, last);
    __syncthreads();
  }
  const float amb = rb_resp(a[n - 1]);
  __syncthreads();
  return rb_partition<false>(a, n, cnt, amb, stL, stR, S);
}

// ---------------------------------------------------------------------------------------------------
// ORB: both passes for one (stream, level) job
// ---------------------------------------------------------------------------------------------------
// Harris response of one candidate (orb.cpp HarrisResponses, blockSize 7, k 0.04): integer sums over the 7x7 block of
// Sobel-like 3x3 derivatives, i.e. a 9x9 byte footprint, fetched as 9 rows x 3 (unaligned) dwords.  Candidates are
// >= 31 px inside the level, so the footprint never leaves it.
__device__ float orb_harris(const u8* __restrict__ img, int pitch, int x0, int y0) {
  unsigned rw[9][3];
  const u8* p = img + (size_t)__umul24(y0 - 4, pitch) + x0 - 4;
#pragma unroll
  for (int r = 0; r < 9; r++, p += pitch) {
    const unsigned* q = (const unsigned*)p;  // byte-aligned address: the memory pipeline takes misaligned dwords
    rw[r][0] = q[0]; rw[r][1] = q[1]; rw[r][2] = q[2];
  }
  auto px = [&](int r, int k) -> int { return (int)((rw[r][k >> 2] >> (8 * (k & 3))) & 0xFFu); };  // k = dx + 4
  int a = 0, b = 0, c = 0;
#pragma unroll
  for (int dy = -3; dy <= 3; dy++) {
    const int r0 = dy + 3, r1 = dy + 4, r2 = dy + 5;
#pragma unroll
    for (int dx = -3; dx <= 3; dx++) {
      const int k = dx + 4;
      int Ix = (px(r1, k + 1) - px(r1, k - 1)) * 2 + (px(r0, k + 1) - px(r0, k - 1)) + (px(r2, k + 1) - px(r2, k - 1));
      int Iy = (px(r2, k) - px(r0, k)) * 2 + (px(r2, k - 1) - px(r0, k - 1)) + (px(r2, k + 1) - px(r0, k + 1));
      a += __mul24(Ix, Ix); b += __mul24(Iy, Iy); c += __mul24(Ix, Iy);
    }
  }
  const float scale = 1.f / ((1 << 2) * 7 * 255.f);
  const float scale_sq_sq = scale * scale * scale * scale;
  return ((float)a * b - (float)c * c - 0.04f * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

struct OrbSelArgs {
  const u8* cs;        // FAST score of each candidate (dense, slot-major then level then row-major)
  float* ch;           // Harris response, written for the survivors of the first pass (ic_angle reads it back)
  const u8* pyr;       // un-blurred pyramid
  size_t slot_stride;
  size_t lvl_off[MVO_ORB_LEVELS];
  int lvl_pitch[MVO_ORB_LEVELS];
  const unsigned short* cx;
  const unsigned short* cy;
  const int* lvl_cnt;  // [B][8]
  const int* slot_base;  // [B+1]
  uint2* wk; uint2* stL; uint2* stR;  // [cand_cap] each
  int* kept;           // [B][8]
  int nlevels;
  int quota[MVO_ORB_LEVELS];
  int cand_cap;
  const int* nact;     // optional device count of active slots
};

// Levels with at most RB_LDS_CAP candidates (at 720p: all of them, level 0 has ~2-3 k) keep the (response, index) array
// in LDS for both passes: every introselect round is a handful of dependent passes over it, and from LDS a round
// costs a fraction of the L2 round trips (the rank-swap staging buffers stay in global memory).
#ifndef RB_LDS_CAP
#define RB_LDS_CAP 4096
#endif
__global__ __launch_bounds__(RB_T) void orb_select_kernel(OrbSelArgs A) {
  __shared__ RbShared S;
  __shared__ uint2 s_a[RB_LDS_CAP];
  const int slot = blockIdx.y, l = blockIdx.x, tid = threadIdx.x;
  if (A.nact && slot >= *A.nact) return;   // block-uniform
  int off = A.slot_base[slot];
  for (int k = 0; k < l; k++) off += A.lvl_cnt[slot * MVO_ORB_LEVELS + k];
  int cnt = A.lvl_cnt[slot * MVO_ORB_LEVELS + l];
  if (off + cnt > A.cand_cap) cnt = max(0, A.cand_cap - off);  // capacity overrun is reported by the host from the counts
  const bool in_lds = cnt <= RB_LDS_CAP;
  uint2* a = in_lds ? s_a : A.wk + off;
  for (int i = tid; i < cnt; i += RB_T) a[i] = make_uint2(__float_as_uint((float)A.cs[off + i]), (unsigned)(off + i));
  __syncthreads();
  int m = rb_retain_best(a, cnt, 2 * A.quota[l], A.stL + off, A.stR + off, S, -1);
  // OpenCV computes the Harris response of the first pass's survivors only (orb.cpp computeKeyPoints)
  {
    const u8* img = A.pyr + (size_t)slot * A.slot_stride + A.lvl_off[l];
    const int pitch = A.lvl_pitch[l];
    for (int i = tid; i < m; i += RB_T) {
      const unsigned ci = a[i].y;
      const float hr = orb_harris(img, pitch, A.cx[ci], A.cy[ci]);
      A.ch[ci] = hr;
      a[i].x = __float_as_uint(hr);
    }
  }
  __syncthreads();
  m = rb_retain_best(a, m, A.quota[l], A.stL + off, A.stR + off, S, -1);
  if (in_lds) {
    __syncthreads();
    for (int i = tid; i < m; i += RB_T) A.wk[off + i] = a[i];   // the gather pass reads the survivors from wk
  }
  if (tid == 0) A.kept[slot * MVO_ORB_LEVELS + l] = m;
}

// kp_base[s] = sum of kept counts of the slots before s: one block, a chunk of 256 slots per pass (a slot per thread,
// LDS Hillis-Steele scan, running carry)
__global__ __launch_bounds__(256) void orb_sel_scan_kernel(const int* __restrict__ kept, int nslots, int nlevels, int* __restrict__ kp_base,
                                                           const int* __restrict__ nact) {
  __shared__ int s_scan[2][256];
  __shared__ int s_carry;
  const int tid = threadIdx.x;
  if (nact) nslots = min(max(*nact, 0), nslots);
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int s0 = 0; s0 < nslots; s0 += 256) {
    const int s = s0 + tid;
    int own = 0;
    if (s < nslots)
      for (int l = 0; l < nlevels; l++) own += kept[s * MVO_ORB_LEVELS + l];
    int cur = 0;
    s_scan[0][tid] = own;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const int v = s_scan[cur][tid] + (tid >= d ? s_scan[cur][tid - d] : 0);
      s_scan[cur ^ 1][tid] = v;
      cur ^= 1;
      __syncthreads();
    }
    const int incl = s_scan[cur][tid], carry = s_carry;
    if (s < nslots) kp_base[s] = carry + incl - own;
    __syncthreads();
    if (tid == 255) s_carry = carry + incl;
    __syncthreads();
  }
  if (tid == 0) kp_base[nslots] = s_carry;
}

// dense selection list: slot-major, levels in order, each level in retainBest's order
__global__ __launch_bounds__(256) void orb_sel_gather_kernel(const uint2* __restrict__ wk, const int* __restrict__ lvl_cnt,
                                                             const int* __restrict__ slot_base, const int* __restrict__ kept,
                                                             const int* __restrict__ kp_base, int kp_cap, int* __restrict__ sel,
                                                             const int* __restrict__ nact) {
  const int slot = blockIdx.y, l = blockIdx.x;
  if (nact && slot >= *nact) return;
  int off = slot_base[slot], dst = kp_base[slot];
  for (int k = 0; k < l; k++) { off += lvl_cnt[slot * MVO_ORB_LEVELS + k]; dst += kept[slot * MVO_ORB_LEVELS + k]; }
  const int m = kept[slot * MVO_ORB_LEVELS + l];
  for (int i = threadIdx.x; i < m; i += 256)
    if (dst + i < kp_cap) sel[dst + i] = (int)wk[off + i].y;
}

// Enqueue selection for `nslots` slots on ctx->stream; d_kp_base / d_sel are valid afterwards (device side).
int orb_select_device(mvo_ctx* ctx, const OrbGeom& G, int nslots, const int* d_nact) {
  OrbState* o = ctx->orb;
  OrbSelArgs A;
  A.nact = d_nact;
  A.cs = o->d_cs; A.ch = o->d_ch; A.lvl_cnt = o->d_lvl_cnt; A.slot_base = o->d_slot_base;
  A.pyr = o->d_pyr; A.slot_stride = G.slot_stride; A.cx = o->d_cx; A.cy = o->d_cy;
  for (int l = 0; l < MVO_ORB_LEVELS; l++) { A.lvl_off[l] = G.off[l]; A.lvl_pitch[l] = G.pitch[l]; }
  A.wk = o->d_wk; A.stL = o->d_stl; A.stR = o->d_str; A.kept = o->d_kept;
  A.nlevels = G.nlevels;
  for (int l = 0; l < MVO_ORB_LEVELS; l++) A.quota[l] = G.quota[l];
  A.cand_cap = o->cand_cap;
  hipLaunchKernelGGL(orb_select_kernel, dim3(G.nlevels, nslots), dim3(RB_T), 0, ctx->stream, A);
  hipLaunchKernelGGL(orb_sel_scan_kernel, dim3(1), dim3(256), 0, ctx->stream, o->d_kept, nslots, G.nlevels, o->d_kp_base, d_nact);
  hipLaunchKernelGGL(orb_sel_gather_kernel, dim3(G.nlevels, nslots), dim3(256), 0, ctx->stream, o->d_wk, o->d_lvl_cnt, o->d_slot_base,
                     o->d_kept, o->d_kp_base, o->kp_cap, o->d_sel, d_nact);
  return MVO_OK;
}

// ---------------------------------------------------------------------------------------------------
// stand-alone entry: cv::KeyPointsFilter::retainBest on a response array
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RB_T) void retain_best_kernel(const float* __restrict__ resp, int n, int keep, int depth_limit, uint2* wk,
                                                           uint2* stL, uint2* stR, int* __restrict__ out_idx, int* __restrict__ out_n) {
  __shared__ RbShared S;
  for (int i = threadIdx.x; i < n; i += RB_T) wk[i] = make_uint2(__float_as_uint(resp[i]), (unsigned)i);
  __syncthreads();
  int m = rb_retain_best(wk, n, keep, stL, stR, S, depth_limit);
  for (int i = threadIdx.x; i < m; i += RB_T) out_idx[i] = (int)wk[i].y;
  if (threadIdx.x == 0) *out_n = m;
}

extern "C" int mvo_retain_best(mvo_ctx* ctx, const float* responses, int n, int n_keep, int depth_limit, int* out_idx, int* out_n) {
  if (!ctx || !responses || !out_idx || !out_n || n < 0) return MVO_E_ARG;
  OrbState* o = ctx->orb;
  *out_n = 0;
  if (n == 0) return MVO_OK;
  if (n > o->cand_cap) { ctx->set_error("mvo_retain_best: more responses than the candidate capacity"); return MVO_E_CAPACITY; }
  hipStream_t st = ctx->stream;
  // d_ch as the response buffer, d_cslot as the index output (both cand_cap wide), d_kept[0] as the count
  MVO_HIP(hipMemcpyAsync(o->d_ch, responses, (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(retain_best_kernel, dim3(1), dim3(RB_T), 0, st, o->d_ch, n, n_keep, depth_limit, o->d_wk, o->d_stl, o->d_str,
                     o->d_cslot, o->d_kept);
  int* hn = (int*)ctx->h_pin;
  MVO_HIP(hipMemcpyAsync(hn, o->d_kept, sizeof(int), hipMemcpyDeviceToHost, st));
  MVO_HIP(hipStreamSynchronize(st));
  int m = hn[0];
  *out_n = m;
  if (m > 0) {
    MVO_HIP(hipMemcpyAsync(out_idx, o->d_cslot, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, st));
    MVO_HIP(hipStreamSynchronize(st));
  }
  return MVO_OK;
}
