// csrc/pipeline.hip — frame-batch mode: B independent camera streams resident in one context, one kernel launch per
// stage for all of them: the frame ring, the seed (the Initializer's hand-over: first key-frame + its
// landmarks), the track accessors, and the HIP-event stage timers.  The per-frame step itself is csrc/track.hip.
// All per-stream state (previous pyramid, tracked points, landmarks, key-frame descriptors) stays in HBM.
#include "mvo_internal.h"
#include "track_policy.h"

#include <dlfcn.h>

#include <cstdlib>

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
// ring frame -> ORB level 0 of every slot (16 B per lane).  LK needs no copy: it reads level 0 in the ring.
__global__ __launch_bounds__(256) void ring_to_orb0_kernel(const u8* __restrict__ ring, size_t ring_slot_stride, int pitch, int h,
                                                           u8* __restrict__ orb0, size_t orb_slot_stride) {
  const int slot = blockIdx.y;
  const size_t n16 = (size_t)pitch * h / 16;
  const uint4* s = (const uint4*)(ring + (size_t)slot * ring_slot_stride);
  uint4* b = (uint4*)(orb0 + (size_t)slot * orb_slot_stride);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) b[i] = s[i];
}

// Tracker::track_frame_with_optical_flow's keep rule (src/tracker.cpp:70-77): status && err < thresh,
// order preserved.  One block of four wavefronts per slot (a 16-wave block has to find a whole CU free at once).
#define LKF_T 256
__global__ __launch_bounds__(LKF_T) void lk_filter_compact_kernel(const float* __restrict__ next_pts, const u8* __restrict__ status,
                                                                 const float* __restrict__ err, const int* __restrict__ npts,
                                                                 const float* __restrict__ lm, const float* __restrict__ kf,
                                                                 float thresh, int maxpts, float* __restrict__ o_pts,
                                                                 float* __restrict__ o_lm, float* __restrict__ o_kf,
                                                                 int* __restrict__ o_n, TrkLostPolicy lost) {
  __shared__ int s_wave[LKF_T / 64];
  __shared__ int s_base;
  const int slot = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = npts[slot];
  const size_t b = (size_t)slot * maxpts;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += LKF_T) {
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
      for (int w = 0; w < LKF_T / 64; w++) t += s_wave[w];
      s_base += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    o_n[slot] = s_base;
    if (lost.state) trk_policy_lost_slot(lost, slot, s_base);
  }
}

void lk_filter_compact_launch(mvo_ctx* ctx, hipStream_t st, const TrkLostPolicy* lost) {
  PipeState* p = ctx->pipe;
  hipLaunchKernelGGL(lk_filter_compact_kernel, dim3(ctx->B), dim3(LKF_T), 0, st, ctx->d_next_pts, ctx->d_status, ctx->d_err, ctx->d_npts,
                     p->d_lm, p->d_kf_pts, ctx->cfg.tracking_error_thresh, ctx->maxpts, p->d_cur_pts, p->d_cur_lm, p->d_cur_kf, p->d_ncur,
                     lost ? *lost : TrkLostPolicy{nullptr, 0, nullptr, nullptr, nullptr});
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
  if (frame_idx == p->prev_entry) p->prev_entry = -1;   // the tracker's previous frame is gone: the next step must be a seed
  u8* dst = p->d_ring + ((size_t)frame_idx * ctx->B + slot) * p->frame_bytes;
  int rc = upload_gray(ctx, img, w, h, stride, channels, dst, p->pitch, slot, true);
  if (rc || channels == 1) return rc;
  // the ring feeds LK: colour frames must be replicated mono8 (see color2gray_kernel)
  int differ = 0;
  if ((rc = color_channels_differ(ctx, &differ))) return rc;
  if (differ) { ctx->set_error("mvo_batch_preload_frame: true-colour frames (channels differ) are not built for the tracker's LK"); return MVO_E_ARG; }
  return MVO_OK;
}

static int pipe_load_frame(mvo_ctx* ctx, int frame_idx) {
  PipeState* p = ctx->pipe;
  OrbState* o = ctx->orb;
  ProfScope ps(ctx, "frame_fanout");
  const u8* src = p->d_ring + (size_t)frame_idx * ctx->B * p->frame_bytes;
  dim3 grid(64, ctx->B);
  hipLaunchKernelGGL(ring_to_orb0_kernel, grid, dim3(256), 0, ctx->stream, src, p->frame_bytes, p->pitch, p->h, o->d_pyr, o->slot_bytes);
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

// Seed: the frame becomes the first key-frame (descriptors and key-point positions change sides) and every key-point a
// track; the landmarks come from mvo_batch_set_landmarks.
static int pipe_promote_seed(mvo_ctx* ctx, int max_n) {
  PipeState* p = ctx->pipe;
  MatchState* m = ctx->match;
  std::swap(m->d_q, m->d_t);
  std::swap(m->d_nq, m->d_nt);
  std::swap(p->d_kfkp_xy, p->d_kp_xy);
  p->kf_max_n = max_n;
  p->trk_max_n = max_n;
  size_t np = (size_t)ctx->B * ctx->maxpts;
  MVO_HIP(hipMemcpyAsync(ctx->d_prev_pts, p->d_kfkp_xy, np * 2 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  MVO_HIP(hipMemcpyAsync(p->d_kf_pts, p->d_kfkp_xy, np * 2 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  MVO_HIP(hipMemcpyAsync(ctx->d_npts, m->d_nq, (size_t)ctx->B * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
  MVO_HIP(hipMemsetAsync(p->d_kf_has, 0, np, ctx->stream));
  return MVO_OK;
}

extern "C" int mvo_batch_seed(mvo_ctx* ctx, int frame_idx, int* n_keypoints) {
  if (!ctx || !ctx->pipe) return MVO_E_ARG;
  PipeState* p = ctx->pipe;
  if (p->ring <= 0 || p->w == 0 || frame_idx < 0 || frame_idx >= p->ring) return MVO_E_ARG;
  int rc;
  LkLevels L = lk_levels(p->w, p->h, ctx->cfg.lk_win, ctx->cfg.lk_max_level);
  if ((rc = trk_reset(ctx))) return rc;   // a step still in flight finishes first; per-stream tracker state: TRACKING, count 0
  ctx->lk_cur = 0;
  if ((rc = trk_wait_upload(ctx, frame_idx))) return rc;
  if ((rc = pipe_load_frame(ctx, frame_idx))) return rc;
  // the seed frame is the first "prev" image: levels 1.. into the pyramid set, level 0 stays where it is (the ring entry)
  lk_build_pyramid(ctx, ctx->lk_cur, L, ctx->B, ctx->stream, p->d_ring + (size_t)frame_idx * ctx->B * p->frame_bytes, p->frame_bytes);
  p->prev_entry = frame_idx;
  std::vector<int> base;
  if ((rc = orb_run(ctx, p->w, p->h, ctx->B, true, base))) return rc;
  int mx = 0;
  if ((rc = pipe_publish_orb(ctx, base, &mx))) return rc;
  if ((rc = pipe_promote_seed(ctx, mx))) return rc;
  MVO_HIP(hipMemsetAsync(p->d_kf_pose, 0, (size_t)ctx->B * 8 * sizeof(double), ctx->stream));  // T_cw = identity
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  if (n_keypoints)
    for (int s = 0; s < ctx->B; s++) n_keypoints[s] = base[s + 1] - base[s];
  p->seeded = true;
  return MVO_OK;
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
