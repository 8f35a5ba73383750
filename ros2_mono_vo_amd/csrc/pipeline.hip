// csrc/pipeline.hip — frame-batch mode: B independent camera streams resident in one context, one
// kernel launch per stage for all of them; plus the HIP-event stage timers.
//
// The step mirrors the reference's steady-state Tracker::update (src/tracker.cpp:274-333) in its worst
// case, where the key-frame branch (has_parallax -> add_new_keyframe) runs on every frame:
//   LK(prev->cur) -> status/err filter -> solvePnPRansac -> findHomography + findFundamentalMat ->
//   ORB(cur) -> knn2+ratio vs the last key-frame -> triangulate.
// All per-stream state (previous pyramid, tracked points, landmarks, key-frame descriptors) stays in HBM.
#include "mvo_internal.h"

#include <dlfcn.h>

#include <map>

// ---------------------------------------------------------------------------------------------------
// stage timers
// ---------------------------------------------------------------------------------------------------
struct Prof {
  bool on = false;
  struct Pending { std::string name; hipEvent_t a, b; hipStream_t st; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> pool;
  std::map<std::string, std::pair<double, int>> acc;
  std::vector<Pending> open;
};

// roctx ranges around the enqueue of every stage (tracing), resolved at run time so the library has no hard
// dependency on the tracer: MVO_ROCTX=1 turns them on; `rocprofv3 --marker-trace` shows them beside the kernel trace.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* e = getenv("MVO_ROCTX");
    if (!e || atoi(e) == 0) return;
    void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
    pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
static Roctx& roctx() { static Roctx r; return r; }

static hipEvent_t prof_event(Prof* p) {
  if (!p->pool.empty()) { hipEvent_t e = p->pool.back(); p->pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

void prof_begin(mvo_ctx* ctx, const char* name, hipStream_t st) {
  if (roctx().push) roctx().push(name);
  Prof* p = ctx->prof;
  if (!p || !p->on) return;
  Prof::Pending q;
  q.name = name; q.a = prof_event(p); q.b = prof_event(p); q.st = st ? st : ctx->stream;
  (void)hipEventRecord(q.a, q.st);
  p->open.push_back(q);
}

void prof_end(mvo_ctx* ctx) {
  if (roctx().pop) roctx().pop();
  Prof* p = ctx->prof;
  if (!p || !p->on || p->open.empty()) return;
  Prof::Pending q = p->open.back();
  p->open.pop_back();
  (void)hipEventRecord(q.b, q.st);
  p->pending.push_back(q);
}

static void prof_collect(mvo_ctx* ctx) {
  Prof* p = ctx->prof;
  if (!p) return;
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& q : p->pending) {
    (void)hipEventSynchronize(q.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, q.a, q.b) == hipSuccess) {
      auto& e = p->acc[q.name];
      e.first += ms; e.second += 1;
    }
    p->pool.push_back(q.a); p->pool.push_back(q.b);
  }
  p->pending.clear();
}

extern "C" int mvo_profile_enable(mvo_ctx* ctx, int on) {
  if (!ctx) return MVO_E_ARG;
  if (!ctx->prof) ctx->prof = new Prof();
  prof_collect(ctx);
  ctx->prof->on = on != 0;
  return MVO_OK;
}

extern "C" int mvo_profile_read(mvo_ctx* ctx, const char* name, double* total_ms, int* launches) {
  if (!ctx || !name || !ctx->prof) return MVO_E_ARG;
  prof_collect(ctx);
  auto it = ctx->prof->acc.find(name);
  if (total_ms) *total_ms = it == ctx->prof->acc.end() ? 0.0 : it->second.first;
  if (launches) *launches = it == ctx->prof->acc.end() ? 0 : it->second.second;
  return MVO_OK;
}

extern "C" int mvo_profile_reset(mvo_ctx* ctx) {
  if (!ctx || !ctx->prof) return MVO_E_ARG;
  prof_collect(ctx);
  ctx->prof->acc.clear();
  return MVO_OK;
}

// ---------------------------------------------------------------------------------------------------
// pipeline state
// ---------------------------------------------------------------------------------------------------
// struct PipeState: mvo_internal.h (shared with track.hip)

int pipe_state_create(mvo_ctx* ctx) {
  PipeState* p = new PipeState();
  ctx->pipe = p;
  if (!ctx->prof) ctx->prof = new Prof();
  p->ring = ctx->cfg.ring_frames;
  if (p->ring <= 0) return MVO_OK;
  p->pitch = align_up(ctx->maxw, 64);
  p->frame_bytes = (size_t)p->pitch * ctx->maxh;
  size_t np = (size_t)ctx->B * ctx->maxpts;
  MVO_HIP(hipMalloc(&p->d_ring, p->frame_bytes * ctx->B * p->ring));
  MVO_HIP(hipMalloc(&p->d_lm, np * 3 * sizeof(float)));
  MVO_HIP(hipMalloc(&p->d_kf_pts, np * 2 * sizeof(float)));
  MVO_HIP(hipMalloc(&p->d_cur_pts, np * 2 * sizeof(float)));
  MVO_HIP(hipMalloc(&p->d_cur_lm, np * 3 * sizeof(float)));
  MVO_HIP(hipMalloc(&p->d_cur_kf, np * 2 * sizeof(float)));
  MVO_HIP(hipMalloc(&p->d_ncur, ctx->B * sizeof(int)));
  MVO_HIP(hipMalloc(&p->d_kp_xy, np * 2 * sizeof(float)));
  MVO_HIP(hipMalloc(&p->d_kfkp_xy, np * 2 * sizeof(float)));
  MVO_HIP(hipMalloc(&p->d_kf_has, np));
  MVO_HIP(hipMalloc(&p->d_kf_lm, np * 3 * sizeof(float)));
  MVO_HIP(hipMalloc(&p->d_cur_has, np));
  MVO_HIP(hipMalloc(&p->d_cur_lmk, np * 3 * sizeof(float)));
  MVO_HIP(hipMalloc(&p->d_winner, np * sizeof(int)));
  MVO_HIP(hipMalloc(&p->d_tri, np * 3 * sizeof(float)));
  MVO_HIP(hipMalloc(&p->d_tri_ok, np));
  MVO_HIP(hipMalloc(&p->d_kf_pose, (size_t)ctx->B * 8 * sizeof(double)));
  MVO_HIP(hipMalloc(&p->d_ntri, ctx->B * sizeof(int)));
  // the side streams of the stage-mask step (mvo_batch_step) are created on first use: HIP maps streams onto a few
  // hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and idle streams of several contexts were sharing - and
  // serialising - the queues the device-driven step of other contexts runs on
  MVO_HIP(hipEventCreateWithFlags(&p->ev_frame, hipEventDisableTiming));
  MVO_HIP(hipEventCreateWithFlags(&p->ev_lktrack, hipEventDisableTiming));
  MVO_HIP(hipEventCreateWithFlags(&p->ev_lk, hipEventDisableTiming));
  MVO_HIP(hipEventCreateWithFlags(&p->ev_pnp, hipEventDisableTiming));
  MVO_HIP(hipEventCreateWithFlags(&p->ev_hf, hipEventDisableTiming));
  MVO_HIP(hipMemsetAsync(p->d_kf_has, 0, np, ctx->stream));
  MVO_HIP(hipMemsetAsync(p->d_kf_lm, 0, np * 3 * sizeof(float), ctx->stream));
  MVO_HIP(hipMemsetAsync(p->d_kf_pose, 0, (size_t)ctx->B * 8 * sizeof(double), ctx->stream));
  MVO_HIP(hipHostMalloc(&p->h_ints, (size_t)ctx->B * 64 * sizeof(int), hipHostMallocDefault));
  MVO_HIP(hipMemsetAsync(p->d_lm, 0, np * 3 * sizeof(float), ctx->stream));
  MVO_HIP(hipMemsetAsync(p->d_kf_pts, 0, np * 2 * sizeof(float), ctx->stream));
  return trk_ring_events(ctx);   // the upload stream right behind the compute stream (see there)
}

void pipe_state_destroy(mvo_ctx* ctx) {
  PipeState* p = ctx->pipe;
  if (p) {
    trk_destroy(ctx);
    void* dev[] = {p->d_ring, p->d_lm, p->d_kf_pts, p->d_cur_pts, p->d_cur_lm, p->d_cur_kf, p->d_ncur, p->d_kp_xy, p->d_kfkp_xy,
                   p->d_kf_has, p->d_kf_lm, p->d_cur_has, p->d_cur_lmk, p->d_winner, p->d_tri, p->d_tri_ok, p->d_kf_pose, p->d_ntri};
    for (void* q : dev) (void)hipFree(q);
    if (p->h_ints) (void)hipHostFree(p->h_ints);
    if (p->s_lk) { (void)hipStreamSynchronize(p->s_lk); (void)hipStreamDestroy(p->s_lk); }
    if (p->ev_frame) (void)hipEventDestroy(p->ev_frame);
    if (p->ev_lktrack) (void)hipEventDestroy(p->ev_lktrack);
    if (p->s_pnp) { (void)hipStreamSynchronize(p->s_pnp); (void)hipStreamDestroy(p->s_pnp); }
    if (p->s_hf) { (void)hipStreamSynchronize(p->s_hf); (void)hipStreamDestroy(p->s_hf); }
    if (p->ev_lk) (void)hipEventDestroy(p->ev_lk);
    if (p->ev_pnp) (void)hipEventDestroy(p->ev_pnp);
    if (p->ev_hf) (void)hipEventDestroy(p->ev_hf);
    delete p;
    ctx->pipe = nullptr;
  }
  if (ctx->prof) {
    prof_collect(ctx);
    for (auto e : ctx->prof->pool) (void)hipEventDestroy(e);
    delete ctx->prof;
    ctx->prof = nullptr;
  }
}

// ---------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------
// ring frame -> LK "cur" level 0 and ORB level 0 (read once, write twice; 16 B per lane)
__global__ __launch_bounds__(256) void ring_to_level0_kernel(const u8* __restrict__ ring, size_t ring_slot_stride, int pitch,
                                                             int h, u8* __restrict__ lk0, size_t lk_slot_stride,
                                                             u8* __restrict__ orb0, size_t orb_slot_stride) {
  const int slot = blockIdx.y;
  const size_t n16 = (size_t)pitch * h / 16;
  const uint4* s = (const uint4*)(ring + (size_t)slot * ring_slot_stride);
  uint4* a = (uint4*)(lk0 + (size_t)slot * lk_slot_stride);
  uint4* b = (uint4*)(orb0 + (size_t)slot * orb_slot_stride);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    uint4 v = s[i];
    a[i] = v;
    b[i] = v;
  }
}

// Tracker::track_frame_with_optical_flow's keep rule (src/tracker.cpp:70-77): status && err < thresh,
// order preserved.  One block per slot.
__global__ __launch_bounds__(1024) void lk_filter_compact_kernel(const float* __restrict__ next_pts, const u8* __restrict__ status,
                                                                 const float* __restrict__ err, const int* __restrict__ npts,
                                                                 const float* __restrict__ lm, const float* __restrict__ kf,
                                                                 float thresh, int maxpts, float* __restrict__ o_pts,
                                                                 float* __restrict__ o_lm, float* __restrict__ o_kf,
                                                                 int* __restrict__ o_n) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int slot = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = npts[slot];
  const size_t b = (size_t)slot * maxpts;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {
    int i = i0 + threadIdx.x;
    bool keep = i < n && status[b + i] && err[b + i] < thresh;
    unsigned long long m = __ballot(keep);
    int pre = __popcll(m & ((1ull << lane) - 1));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; w++) off += s_wave[w];
    if (keep) {
      size_t o = b + off + pre;
      o_pts[2 * o] = next_pts[2 * (b + i)]; o_pts[2 * o + 1] = next_pts[2 * (b + i) + 1];
      o_lm[3 * o] = lm[3 * (b + i)]; o_lm[3 * o + 1] = lm[3 * (b + i) + 1]; o_lm[3 * o + 2] = lm[3 * (b + i) + 2];
      o_kf[2 * o] = kf[2 * (b + i)]; o_kf[2 * o + 1] = kf[2 * (b + i) + 1];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < 16; w++) t += s_wave[w];
      s_base += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) o_n[slot] = s_base;
}

void lk_filter_compact_launch(mvo_ctx* ctx, hipStream_t st) {
  PipeState* p = ctx->pipe;
  hipLaunchKernelGGL(lk_filter_compact_kernel, dim3(ctx->B), dim3(1024), 0, st, ctx->d_next_pts, ctx->d_status, ctx->d_err, ctx->d_npts,
                     p->d_lm, p->d_kf_pts, ctx->cfg.tracking_error_thresh, ctx->maxpts, p->d_cur_pts, p->d_cur_lm, p->d_cur_kf, p->d_ncur);
}

// dense ORB output (all slots back to back) -> per-slot matcher / track layout
__global__ __launch_bounds__(256) void scatter_kp_kernel(const mvo_keypoint* __restrict__ kp, const u8* __restrict__ desc,
                                                         const int* __restrict__ kp_base, int maxpts, u8* __restrict__ t_desc,
                                                         int* __restrict__ t_n, float* __restrict__ kp_xy) {
  const int slot = blockIdx.y;
  const int b0 = kp_base[slot], n = kp_base[slot + 1] - b0;
  if (blockIdx.x == 0 && threadIdx.x == 0) t_n[slot] = n;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n * 8; i += gridDim.x * 256) {
    int k = i >> 3, q = i & 7;
    ((unsigned*)t_desc)[((size_t)slot * maxpts + k) * 8 + q] = ((const unsigned*)desc)[(size_t)(b0 + k) * 8 + q];
    if (q == 0) {
      kp_xy[2 * ((size_t)slot * maxpts + k)] = kp[b0 + k].x;
      kp_xy[2 * ((size_t)slot * maxpts + k) + 1] = kp[b0 + k].y;
    }
  }
}


// ---- landmark hand-over of Tracker::add_new_keyframe (src/tracker.cpp:211-227) --------------------------
// Sequential reference semantics: matches are visited in order, so when several matches share a train
// index the LAST valid one decides that key-point's landmark.  Phase 1: winner[t] = max valid match index.
__global__ __launch_bounds__(256) void landmark_winner_kernel(const mvo_match* __restrict__ matches, const int* __restrict__ n_matches,
                                                              const u8* __restrict__ valid, int cap, int* __restrict__ winner) {
  const int slot = blockIdx.y;
  const int n = min(max(n_matches[slot], 0), cap);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const size_t b = (size_t)slot * cap;
  if (valid[b + i]) atomicMax(&winner[b + matches[b + i].train_idx], i);
}

// Phase 2 (one block per slot): per current key-point take the winning match: reuse the key-frame
// observation's landmark if it has one, else the freshly triangulated point; then an ordered compaction
// of the key-points WITH landmarks becomes the next frame's tracks (Frame::get_points_2d(WITH_LANDMARKS)).
__global__ __launch_bounds__(1024) void landmark_assign_kernel(const mvo_match* __restrict__ matches, const int* __restrict__ winner,
                                                               const int* __restrict__ n_kp, const float* __restrict__ kp_xy,
                                                               const u8* __restrict__ kf_has, const float* __restrict__ kf_lm,
                                                               const float* __restrict__ tri, const u8* __restrict__ tri_ok,
                                                               const int* __restrict__ n_matches, int cap,
                                                               u8* __restrict__ cur_has, float* __restrict__ cur_lm,
                                                               float* __restrict__ trk_xy, float* __restrict__ trk_lm,
                                                               float* __restrict__ trk_kf, int* __restrict__ n_trk, int* __restrict__ n_tri) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  __shared__ int s_tri;
  const int slot = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = min(max(n_kp[slot], 0), cap);
  const size_t b = (size_t)slot * cap;
  if (threadIdx.x == 0) { s_base = 0; s_tri = 0; }
  __syncthreads();
  int tri_cnt = 0;
  const int nm = min(max(n_matches[slot], 0), cap);
  for (int i = threadIdx.x; i < nm; i += 1024) tri_cnt += tri_ok[b + i] ? 1 : 0;
  atomicAdd(&s_tri, tri_cnt);
  for (int t0 = 0; t0 < n; t0 += 1024) {
    int t = t0 + threadIdx.x;
    bool has = false;
    float lx = 0, ly = 0, lz = 0;
    if (t < n) {
      int wi = winner[b + t];
      if (wi >= 0) {
        int q = matches[b + wi].query_idx;
        has = true;
        if (kf_has[b + q]) { lx = kf_lm[3 * (b + q)]; ly = kf_lm[3 * (b + q) + 1]; lz = kf_lm[3 * (b + q) + 2]; }
        else { lx = tri[3 * (b + wi)]; ly = tri[3 * (b + wi) + 1]; lz = tri[3 * (b + wi) + 2]; }
      }
      cur_has[b + t] = has ? 1 : 0;
      cur_lm[3 * (b + t)] = lx; cur_lm[3 * (b + t) + 1] = ly; cur_lm[3 * (b + t) + 2] = lz;
    }
    unsigned long long m = __ballot(has);
    int pre = __popcll(m & ((1ull << lane) - 1));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; w++) off += s_wave[w];
    if (has) {
      size_t o = b + off + pre;
      float x = kp_xy[2 * (b + t)], y = kp_xy[2 * (b + t) + 1];
      trk_xy[2 * o] = x; trk_xy[2 * o + 1] = y;
      trk_kf[2 * o] = x; trk_kf[2 * o + 1] = y;
      trk_lm[3 * o] = lx; trk_lm[3 * o + 1] = ly; trk_lm[3 * o + 2] = lz;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int tt = 0;
      for (int w = 0; w < 16; w++) tt += s_wave[w];
      s_base += tt;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { n_trk[slot] = s_base; n_tri[slot] = s_tri; }
}

// ---------------------------------------------------------------------------------------------------
// API
// ---------------------------------------------------------------------------------------------------
extern "C" int mvo_batch_set_intrinsics(mvo_ctx* ctx, const double K[9], const double d[5]) {
  if (!ctx || !K || !ctx->pipe) return MVO_E_ARG;
  memcpy(ctx->pipe->K, K, 9 * sizeof(double));
  if (d) memcpy(ctx->pipe->dist, d, 5 * sizeof(double));
  return MVO_OK;
}

extern "C" int mvo_batch_preload_frame(mvo_ctx* ctx, int slot, int frame_idx, const uint8_t* img, int w, int h,
                                       int stride, int channels) {
  if (!ctx || !img || !ctx->pipe) return MVO_E_ARG;
  PipeState* p = ctx->pipe;
  if (p->ring <= 0 || slot < 0 || slot >= ctx->B || frame_idx < 0 || frame_idx >= p->ring) return MVO_E_ARG;
  if (w > ctx->maxw || h > ctx->maxh || w < 32 || h < 32) return MVO_E_ARG;
  if (p->w == 0) { p->w = w; p->h = h; p->pitch = align_up(w, 64); }
  if (w != p->w || h != p->h) { ctx->set_error("all ring frames must share one size"); return MVO_E_ARG; }
  u8* dst = p->d_ring + ((size_t)frame_idx * ctx->B + slot) * p->frame_bytes;
  int rc = upload_gray(ctx, img, w, h, stride, channels, dst, p->pitch, slot, true);
  if (rc || channels == 1) return rc;
  // the ring feeds LK: colour frames must be replicated mono8 (see color2gray_kernel)
  int differ = 0;
  if ((rc = color_channels_differ(ctx, &differ))) return rc;
  if (differ) { ctx->set_error("mvo_batch_preload_frame: true-colour frames (channels differ) are not built for the tracker's LK"); return MVO_E_ARG; }
  return MVO_OK;
}

static int pipe_load_frame(mvo_ctx* ctx, int frame_idx, int lk_set) {
  PipeState* p = ctx->pipe;
  OrbState* o = ctx->orb;
  ProfScope ps(ctx, "frame_fanout");
  const u8* src = p->d_ring + (size_t)frame_idx * ctx->B * p->frame_bytes;
  dim3 grid(64, ctx->B);
  hipLaunchKernelGGL(ring_to_level0_kernel, grid, dim3(256), 0, ctx->stream, src, p->frame_bytes, p->pitch, p->h,
                     ctx->lk_mem[lk_set] + ctx->lk_level_off[0], ctx->lk_slot_bytes, o->d_pyr, o->slot_bytes);
  return MVO_OK;
}

// After ORB on the current frame: publish key-points as the matcher's train set and the frame's positions.
static int pipe_publish_orb(mvo_ctx* ctx, const std::vector<int>& kp_base, int* max_n) {
  PipeState* p = ctx->pipe;
  OrbState* o = ctx->orb;
  MatchState* m = ctx->match;
  int B = ctx->B;
  int* hb = p->h_ints + 16 * B;  // staging region of its own (the step's counters live in [0, 5B))
  int mx = 0;
  for (int s = 0; s <= B; s++) hb[s] = kp_base[s];
  for (int s = 0; s < B; s++) {
    int n = kp_base[s + 1] - kp_base[s];
    if (n > ctx->maxpts) { ctx->set_error("key-points exceed max_points"); return MVO_E_CAPACITY; }
    mx = n > mx ? n : mx;
  }
  *max_n = mx;
  // d_slot_base is free again after orb_run: reuse it to carry kp_base
  MVO_HIP(hipMemcpyAsync(o->d_slot_base, hb, (size_t)(B + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  dim3 grid(std::max(1, (mx * 8 + 255) / 256), B);
  hipLaunchKernelGGL(scatter_kp_kernel, grid, dim3(256), 0, ctx->stream, o->d_kp, o->d_desc, o->d_slot_base, ctx->maxpts,
                     m->d_t, m->d_nt, p->d_kp_xy);
  return MVO_OK;
}

// key-frame := current frame: descriptors, key-point positions and per-observation landmarks change sides.
// `all_tracks`: seed mode — every key-point becomes a track (landmarks supplied by mvo_batch_set_landmarks).
static int pipe_promote_keyframe(mvo_ctx* ctx, int max_n, bool all_tracks) {
  PipeState* p = ctx->pipe;
  MatchState* m = ctx->match;
  std::swap(m->d_q, m->d_t);
  std::swap(m->d_nq, m->d_nt);
  std::swap(p->d_kfkp_xy, p->d_kp_xy);
  p->kf_max_n = max_n;
  size_t np = (size_t)ctx->B * ctx->maxpts;
  if (all_tracks) {
    p->trk_max_n = max_n;
    MVO_HIP(hipMemcpyAsync(ctx->d_prev_pts, p->d_kfkp_xy, np * 2 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    MVO_HIP(hipMemcpyAsync(p->d_kf_pts, p->d_kfkp_xy, np * 2 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    MVO_HIP(hipMemcpyAsync(ctx->d_npts, m->d_nq, (size_t)ctx->B * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    MVO_HIP(hipMemsetAsync(p->d_kf_has, 0, np, ctx->stream));
  } else {
    std::swap(p->d_kf_has, p->d_cur_has);
    std::swap(p->d_kf_lm, p->d_cur_lmk);
  }
  return MVO_OK;
}

extern "C" int mvo_batch_seed(mvo_ctx* ctx, int frame_idx, int* n_keypoints) {
  if (!ctx || !ctx->pipe) return MVO_E_ARG;
  PipeState* p = ctx->pipe;
  if (p->ring <= 0 || p->w == 0 || frame_idx < 0 || frame_idx >= p->ring) return MVO_E_ARG;
  int rc;
  LkLevels L = lk_levels(p->w, p->h, ctx->cfg.lk_win, ctx->cfg.lk_max_level);
  ctx->lk_cur = 0;
  if ((rc = trk_wait_upload(ctx, frame_idx))) return rc;
  if ((rc = pipe_load_frame(ctx, frame_idx, ctx->lk_cur))) return rc;
  lk_build_pyramid(ctx, ctx->lk_cur, L, ctx->B);
  std::vector<int> base;
  if ((rc = orb_run(ctx, p->w, p->h, ctx->B, true, base))) return rc;
  int mx = 0;
  if ((rc = pipe_publish_orb(ctx, base, &mx))) return rc;
  if ((rc = pipe_promote_keyframe(ctx, mx, true))) return rc;
  MVO_HIP(hipMemsetAsync(p->d_kf_pose, 0, (size_t)ctx->B * 8 * sizeof(double), ctx->stream));  // T_cw = identity
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  if (n_keypoints)
    for (int s = 0; s < ctx->B; s++) n_keypoints[s] = base[s + 1] - base[s];
  p->seeded = true;
  return trk_reset(ctx);   // per-stream tracker state: TRACKING, tracking_count_from_keyframe_ = 0
}

extern "C" int mvo_batch_get_tracks(mvo_ctx* ctx, int slot, float* pts, int cap, int* n) {
  if (!ctx || !ctx->pipe || !n || slot < 0 || slot >= ctx->B) return MVO_E_ARG;
  int* hn = (int*)ctx->h_pin;
  MVO_HIP(hipMemcpyAsync(hn, ctx->d_npts + slot, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  int cnt = hn[0];
  *n = cnt;
  int m = cnt < cap ? cnt : cap;
  if (m > 0 && pts) {
    MVO_HIP(hipMemcpyAsync(pts, ctx->d_prev_pts + (size_t)slot * ctx->maxpts * 2, (size_t)m * 2 * sizeof(float),
                           hipMemcpyDeviceToHost, ctx->stream));
    MVO_HIP(hipStreamSynchronize(ctx->stream));
  }
  return MVO_OK;
}

// Landmarks of the current tracks of `slot`, in track order.  After mvo_batch_seed the tracks are all
// key-points of the key-frame, so the same positions also become the key-frame observations' landmarks.
extern "C" int mvo_batch_set_landmarks(mvo_ctx* ctx, int slot, const float* xyz, int n) {
  if (!ctx || !ctx->pipe || !xyz || slot < 0 || slot >= ctx->B || n < 0 || n > ctx->maxpts) return MVO_E_ARG;
  PipeState* p = ctx->pipe;
  size_t o = (size_t)slot * ctx->maxpts;
  MVO_HIP(hipMemcpyAsync(p->d_lm + o * 3, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  MVO_HIP(hipMemcpyAsync(p->d_kf_lm + o * 3, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  MVO_HIP(hipMemsetAsync(p->d_kf_has + o, 1, (size_t)n, ctx->stream));
  int rc = trk_output_seed(ctx, slot, p->d_lm + o * 3, n);   // output side enabled: the seed landmarks open the slot's map
  if (rc) return rc;
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  return MVO_OK;
}

static int pipe_side_streams(mvo_ctx* ctx) {
  PipeState* p = ctx->pipe;
  if (p->s_lk) return MVO_OK;
  int prio_lo = 0, prio_hi = 0;  // numerically lower = higher priority
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  MVO_HIP(hipStreamCreateWithPriority(&p->s_lk, hipStreamNonBlocking, prio_hi));
  MVO_HIP(hipStreamCreateWithPriority(&p->s_pnp, hipStreamNonBlocking, prio_hi));
  MVO_HIP(hipStreamCreateWithFlags(&p->s_hf, hipStreamNonBlocking));
  return MVO_OK;
}

static int batch_step_enqueue_and_collect(mvo_ctx* ctx, int frame_idx, unsigned stages, mvo_step_result* out);

// An error return in the middle of a step must not leave work queued on the side streams: the next call would start on
// buffers they still write.  Drain them (and the main stream) before handing the status back.
extern "C" int mvo_batch_step(mvo_ctx* ctx, int frame_idx, unsigned stages, mvo_step_result* out) {
  if (!ctx || !ctx->pipe || !out) return MVO_E_ARG;
  const int rc = batch_step_enqueue_and_collect(ctx, frame_idx, stages, out);
  if (rc != MVO_OK) {
    PipeState* p = ctx->pipe;
    if (p->s_lk) (void)hipStreamSynchronize(p->s_lk);
    if (p->s_pnp) (void)hipStreamSynchronize(p->s_pnp);
    if (p->s_hf) (void)hipStreamSynchronize(p->s_hf);
    (void)hipStreamSynchronize(ctx->stream);
  }
  return rc;
}

static int batch_step_enqueue_and_collect(mvo_ctx* ctx, int frame_idx, unsigned stages, mvo_step_result* out) {
  PipeState* p = ctx->pipe;
  { int rcs = pipe_side_streams(ctx); if (rcs) return rcs; }
  MatchState* m = ctx->match;
  GeomState* g = ctx->geom;
  if (!p->seeded || frame_idx < 0 || frame_idx >= p->ring) { ctx->set_error("mvo_batch_step: not seeded / bad frame"); return MVO_E_ARG; }
  const int B = ctx->B;
  const size_t np = (size_t)B * ctx->maxpts;
  int rc;
  LkLevels L = lk_levels(p->w, p->h, ctx->cfg.lk_win, ctx->cfg.lk_max_level);
  const int prev_set = ctx->lk_cur, cur_set = ctx->lk_cur ^ 1;
  memset(out, 0, sizeof(mvo_step_result) * B);
  // Stream plan (all B camera streams per launch):
  //   main   frame fan-out -> ORB detect -> [host: retainBest | device: blur] -> angle + rBRIEF -> match -> triangulate
  //   s_lk   (waits for the fan-out) pyrDown pyramid -> LK -> status/err filter            -- beside ORB detect
  //   s_pnp  (waits for LK) PnP RANSAC + refine        s_hf  (waits for LK) H RANSAC, F RANSAC
  // so the host-side key-point selection overlaps LK and the RANSAC chains instead of idling the device.
  if ((rc = trk_wait_upload(ctx, frame_idx))) return rc;
  if ((rc = pipe_load_frame(ctx, frame_idx, cur_set))) return rc;
  MVO_HIP(hipEventRecord(p->ev_frame, ctx->stream));
  const bool do_orb = stages & MVO_STAGE_ORB;
  MVO_HIP(hipStreamWaitEvent(p->s_lk, p->ev_frame, 0));
  { ProfScope ps(ctx, "lk_pyramid", p->s_lk); lk_build_pyramid(ctx, cur_set, L, B, p->s_lk); }
  // pinned layout: hb[0..B) n_prev, [B..2B) n_tracked, [2B..3B) n_matches, [3B..4B) n_tri, [4B..5B) n_new_tracks,
  //                hr[0..8B) pnp result, [8B..16B) H result, [16B..24B) F result; hp: [B][8] pose
  int* hb = p->h_ints;
  int* hr = g->h_result;
  double* hp = g->h_model;
  if (stages & MVO_STAGE_LK) {
    { ProfScope ps(ctx, "lk_track", p->s_lk); lk_track_device(ctx, prev_set, cur_set, L, B, p->trk_max_n, p->s_lk); }
    MVO_HIP(hipEventRecord(p->ev_lktrack, p->s_lk));
    // ORB's pyramid runs beside LK; its wide FAST kernels start when LK is through (LK heads the critical chain
    // LK -> PnP RANSAC -> refine -> triangulate, whose later links leave most of the device to ORB anyway)
    if (do_orb && (rc = orb_detect_enqueue(ctx, p->w, p->h, B, p->ev_lktrack))) return rc;
    ProfScope ps(ctx, "lk_filter", p->s_lk);
    hipLaunchKernelGGL(lk_filter_compact_kernel, dim3(B), dim3(1024), 0, p->s_lk, ctx->d_next_pts, ctx->d_status, ctx->d_err,
                       ctx->d_npts, p->d_lm, p->d_kf_pts, ctx->cfg.tracking_error_thresh, ctx->maxpts, p->d_cur_pts,
                       p->d_cur_lm, p->d_cur_kf, p->d_ncur);
    MVO_HIP(hipMemcpyAsync(hb, ctx->d_npts, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, p->s_lk));
    MVO_HIP(hipMemcpyAsync(hb + B, p->d_ncur, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, p->s_lk));
  }
  if (do_orb && !(stages & MVO_STAGE_LK) && (rc = orb_detect_enqueue(ctx, p->w, p->h, B))) return rc;
  MVO_HIP(hipEventRecord(p->ev_lk, p->s_lk));
  const bool do_pnp = (stages & MVO_STAGE_LK) && (stages & MVO_STAGE_PNP);
  const bool do_hf = (stages & MVO_STAGE_LK) && (stages & MVO_STAGE_HF);
  if (do_pnp) {
    MVO_HIP(hipStreamWaitEvent(p->s_pnp, p->ev_lk, 0));
    { ProfScope ps(ctx, "pnp", p->s_pnp);
      geom_pnp(ctx, B, p->d_cur_lm, p->d_cur_pts, p->d_ncur, p->K, p->dist, 100, 8.0f, 0.99, g->d_mask, g->d_model, g->d_result, g->d_inl, g->d_pose,
               p->s_pnp); }
    MVO_HIP(hipMemcpyAsync(hr, g->d_result, (size_t)B * 8 * sizeof(int), hipMemcpyDeviceToHost, p->s_pnp));
    MVO_HIP(hipMemcpyAsync(hp, g->d_pose, (size_t)B * 8 * sizeof(double), hipMemcpyDeviceToHost, p->s_pnp));
    MVO_HIP(hipEventRecord(p->ev_pnp, p->s_pnp));
  }
  if (do_hf) {
    // Tracker::has_parallax: key-frame positions of the tracked landmarks vs their current positions
    MVO_HIP(hipStreamWaitEvent(p->s_hf, p->ev_lk, 0));
    // H / F are only read at the end of the step: let them start after ORB's detect kernels, so that FAST shares its
    // SIMDs with the PnP chain alone
    if (do_orb) MVO_HIP(hipStreamWaitEvent(p->s_hf, ctx->orb->ev_counts, 0));
    { ProfScope ps(ctx, "ransac_h", p->s_hf);
      geom_ransac_h(ctx, B, p->d_cur_kf, p->d_cur_pts, p->d_ncur, ctx->cfg.ransac_reproj_thresh, 2000, 0.995, g->d_mask2, g->d_model2,
                    g->d_result2, p->s_hf); }
    MVO_HIP(hipMemcpyAsync(hr + 8 * B, g->d_result2, (size_t)B * 8 * sizeof(int), hipMemcpyDeviceToHost, p->s_hf));
    { ProfScope ps(ctx, "ransac_f", p->s_hf);
      geom_ransac_f(ctx, B, p->d_cur_kf, p->d_cur_pts, p->d_ncur, ctx->cfg.ransac_reproj_thresh, 1000, 0.99, g->d_mask2, g->d_model2,
                    g->d_result2, p->s_hf); }
    MVO_HIP(hipMemcpyAsync(hr + 16 * B, g->d_result2, (size_t)B * 8 * sizeof(int), hipMemcpyDeviceToHost, p->s_hf));
    MVO_HIP(hipEventRecord(p->ev_hf, p->s_hf));
  }
  std::vector<int> base;
  if (do_orb) {
    if ((rc = orb_select(ctx, p->w, p->h, B, true, base))) return rc;
    if ((rc = orb_describe_enqueue(ctx, p->w, p->h, B, true, false, base))) return rc;
    int mx = 0;
    if ((rc = pipe_publish_orb(ctx, base, &mx))) return rc;
    const bool do_match = stages & MVO_STAGE_MATCH;
    const bool do_tri = do_match && do_pnp && (stages & MVO_STAGE_TRIANG);
    if (do_match) {
      ProfScope ps(ctx, "match");
      match_device(ctx, B, p->kf_max_n, ctx->cfg.lowes_distance_ratio);
      MVO_HIP(hipMemcpyAsync(hb + 2 * B, m->d_nout, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    }
    // everything below rewrites the track arrays LK reads (d_prev_pts, d_lm, d_kf_pts, d_npts)
    MVO_HIP(hipStreamWaitEvent(ctx->stream, p->ev_lk, 0));
    if (do_tri) {
      MVO_HIP(hipStreamWaitEvent(ctx->stream, p->ev_pnp, 0));  // the pose comes from the PnP side stream
      ProfScope ps(ctx, "triangulate");
      int max_m = p->kf_max_n;  // matches <= queries
      geom_triangulate_matches(ctx, B, max_m, m->d_out, m->d_nout, p->d_kfkp_xy, p->d_kp_xy, p->d_kf_pose, g->d_pose, g->d_result, p->K,
                               p->d_tri, p->d_tri_ok);
      MVO_HIP(hipMemsetAsync(p->d_winner, 0xFF, np * sizeof(int), ctx->stream));
      dim3 grid((max_m + 255) / 256, B);
      hipLaunchKernelGGL(landmark_winner_kernel, grid, dim3(256), 0, ctx->stream, m->d_out, m->d_nout, p->d_tri_ok, ctx->maxpts, p->d_winner);
      hipLaunchKernelGGL(landmark_assign_kernel, dim3(B), dim3(1024), 0, ctx->stream, m->d_out, p->d_winner, m->d_nt, p->d_kp_xy,
                         p->d_kf_has, p->d_kf_lm, p->d_tri, p->d_tri_ok, m->d_nout, ctx->maxpts, p->d_cur_has, p->d_cur_lmk,
                         ctx->d_prev_pts, p->d_lm, p->d_kf_pts, ctx->d_npts, p->d_ntri);
      MVO_HIP(hipMemcpyAsync(hb + 3 * B, p->d_ntri, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      MVO_HIP(hipMemcpyAsync(hb + 4 * B, ctx->d_npts, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      MVO_HIP(hipMemcpyAsync(p->d_kf_pose, g->d_pose, (size_t)B * 8 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
      if ((rc = pipe_promote_keyframe(ctx, mx, false))) return rc;
    } else {
      // no triangulation stage: every key-point of the new key-frame becomes a track (landmarks keep their last values)
      if ((rc = pipe_promote_keyframe(ctx, mx, true))) return rc;
    }
  } else if (stages & MVO_STAGE_LK) {
    // no key-frame: survivors become the next frame's tracks (src/tracker.cpp:331)
    MVO_HIP(hipStreamWaitEvent(ctx->stream, p->ev_lk, 0));
    MVO_HIP(hipMemcpyAsync(ctx->d_prev_pts, p->d_cur_pts, np * 2 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    MVO_HIP(hipMemcpyAsync(p->d_lm, p->d_cur_lm, np * 3 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    MVO_HIP(hipMemcpyAsync(p->d_kf_pts, p->d_cur_kf, np * 2 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    MVO_HIP(hipMemcpyAsync(ctx->d_npts, p->d_ncur, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
  }
  // the LK survivors (d_cur_*) must not be overwritten by the next step before the side streams are done
  MVO_HIP(hipStreamSynchronize(p->s_lk));
  if (do_pnp) MVO_HIP(hipStreamSynchronize(p->s_pnp));
  if (do_hf) MVO_HIP(hipStreamSynchronize(p->s_hf));
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  const bool tri_done = (stages & MVO_STAGE_ORB) && (stages & MVO_STAGE_MATCH) && do_pnp && (stages & MVO_STAGE_TRIANG);
  int mx_trk = 0;
  for (int s = 0; s < B; s++) {
    if (stages & MVO_STAGE_LK) { out[s].n_prev = hb[s]; out[s].n_tracked = hb[B + s]; }
    if (do_pnp) {
      out[s].pnp_ok = hr[8 * s] && hr[8 * s + 6];
      out[s].n_pnp_inliers = hr[8 * s + 5];
      for (int k = 0; k < 3; k++) { out[s].rvec[k] = hp[8 * s + k]; out[s].tvec[k] = hp[8 * s + 3 + k]; }
    }
    if (do_hf) {
      out[s].score_h = hr[8 * B + 8 * s] ? hr[8 * B + 8 * s + 1] : 0;
      out[s].score_f = hr[16 * B + 8 * s] ? hr[16 * B + 8 * s + 1] : 0;
    }
    if (stages & MVO_STAGE_ORB) out[s].n_keypoints = base[s + 1] - base[s];
    if ((stages & MVO_STAGE_ORB) && (stages & MVO_STAGE_MATCH)) out[s].n_matches = hb[2 * B + s];
    if (tri_done) { out[s].n_triangulated = hb[3 * B + s]; mx_trk = hb[4 * B + s] > mx_trk ? hb[4 * B + s] : mx_trk; }
    else if (!(stages & MVO_STAGE_ORB)) mx_trk = hb[B + s] > mx_trk ? hb[B + s] : mx_trk;
  }
  if (tri_done || ((stages & MVO_STAGE_LK) && !(stages & MVO_STAGE_ORB))) p->trk_max_n = mx_trk;
  ctx->lk_cur = cur_set;
  return MVO_OK;
}
