// csrc/match.hip — brute-force Hamming 2-NN + Lowe ratio for gfx950 (replaces
// cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, 2) + the ratio loop at reference
// src/feature_processor.cpp:25-41; semantics per SURVEY.md Appendix A.2).
//
// hamming_knn2_kernel: one query descriptor per lane (8 dwords in registers); train descriptors are
// streamed through LDS in tiles and read back as wave-wide broadcasts (conflict-free), XOR + v_bcnt
// accumulate.  Each (distance, trainIdx) pair is packed as (d << 16 | j); the two smallest packed keys
// are exactly OpenCV's insertion order (smaller distance first, ties -> lower train index first).
// ratio_compact_kernel: ordered (query-order) compaction of the matches that pass the ratio test, with
// the comparison done in double like the reference's `float < double * float`.
#include "mvo_internal.h"

#define MT_TILE 256  // train rows per LDS tile (8 KB)

__global__ __launch_bounds__(256) void hamming_knn2_kernel(const u8* __restrict__ q, const u8* __restrict__ t,
                                                           const int* __restrict__ nq_, const int* __restrict__ nt_,
                                                           unsigned* __restrict__ best, int cap,
                                                           const int* __restrict__ list, const int* __restrict__ nlist) {
  __shared__ uint4 s_t[MT_TILE * 2];
  // `list` (optional): blockIdx.y indexes a device-resident list of *nlist slots (the key-frame slots of this step)
  if (list && (int)blockIdx.y >= *nlist) return;   // block-uniform
  const int slot = list ? list[blockIdx.y] : blockIdx.y;
  const int nq = min(nq_[slot], cap), nt = min(nt_[slot], cap);
  const int qi = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= nq) return;  // block-uniform
  const uint4* qp = (const uint4*)(q + ((size_t)slot * cap + (qi < nq ? qi : 0)) * 32);
  const uint4 qa = qp[0], qb = qp[1];
  const uint4* tp = (const uint4*)(t + (size_t)slot * cap * 32);
  unsigned b1 = 0xFFFFFFFFu, b2 = 0xFFFFFFFFu;
  for (int t0 = 0; t0 < nt; t0 += MT_TILE) {
    int cnt = min(MT_TILE, nt - t0);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt * 2; i += 256) s_t[i] = tp[(size_t)t0 * 2 + i];
    __syncthreads();
    for (int j = 0; j < cnt; j++) {
      uint4 ta = s_t[2 * j], tb = s_t[2 * j + 1];
      unsigned d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                   __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
      unsigned key = (d << 16) | (unsigned)(t0 + j);
      b2 = min(b2, max(b1, key));
      b1 = min(b1, key);
    }
  }
  if (qi < nq) {
    size_t o = ((size_t)slot * cap + qi) * 2;
    best[o] = b1;
    best[o + 1] = b2;
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
