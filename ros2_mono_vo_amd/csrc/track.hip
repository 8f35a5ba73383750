// csrc/track.hip — frame-batch mode as B x Tracker::update: every camera stream (slot) carries its own tracker
// state on the device and takes its own branch of the reference's per-frame step (src/tracker.cpp:274-333):
//
//   LOST is terminal                                                  src/tracker.cpp:277-279
//   LK + status/err filter                                            :58-90, :289
//   fewer than min_tracked_points survivors -> LOST, no pose          :292-296
//   solvePnPRansac + Rodrigues -> pose                                :300-316
//   ++tracking_count_from_keyframe; should_add_keyframe               :318-319, :118-136, :92-116
//     -> has_parallax (findHomography + findFundamentalMat scores)    :320, :237-268
//       -> add_new_keyframe (ORB, match, triangulate, landmarks)      :322, :182-235
//   prev_frame_ = new_frame                                           :331
//
// The step is DEVICE DRIVEN: the per-slot decisions are taken by small policy kernels, the key-frame branch runs over
// a compacted device-resident list of the slots that take it, and every launch is sized on the host for the worst case
// and reads the real counts on the device (strided loops / early exits).  Nothing waits for the host between the first
// and the last launch, so a step is enqueued asynchronously (mvo_batch_track_async) and several contexts interleave on
// one GPU: the one-wavefront-per-stream RANSAC chains of one context run beside the wide LK / ORB kernels of another.
// Ingest is asynchronous as well: mvo_batch_upload_async copies pinned host frames into the device ring on a dedicated
// stream while the previous step computes.
#include "mvo_internal.h"
#include "track_policy.h"

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <mutex>

struct TrackState {
  int* d_state = nullptr;      // [B] MVO_TRACK_*
  int* d_count = nullptr;      // [B] tracking_count_from_keyframe_
  int* d_flags = nullptr;      // [B] MVO_STEP_* of the running step
  int* d_n_pnp = nullptr;      // [B] correspondences fed to PnP (0 = slot sits the stage out)
  int* d_n_hf = nullptr;       // [B] correspondences fed to H / F (0 = no key-frame test this frame)
  int* d_kf_list = nullptr;    // [B] slots that add a key-frame this step, ascending
  int* d_nkf = nullptr;        // [1]
  int* d_pt_base = nullptr;    // [B+1] LK work list: first item of each slot
  int* d_work_slot = nullptr;  // [B*maxpts] LK work list: slot << 16 | point of each item (trk_worklist_sort_kernel)
  int* d_work_ctr = nullptr;   // [8] claim counter of the LK work list in [0] (the XCD-partitioned list of round 2 used all eight)
  int* d_hf_ctr = nullptr;     // [2] slot-queue counters of the H and the F launch (zeroed by slot 0's refine workgroup, trk_policy_keyframe_slot)
  bool res_valid = false;      // h_res holds the results of the step collected last (host-side forecast of the key-frame tests)
  int* d_err = nullptr;        // [1] capacity flags raised on the device
  u8* d_mask_f = nullptr;      // [B][maxpts] F consensus mask (H uses geom->d_mask2)
  double* d_model_f = nullptr; // [B][16]
  int* d_result_f = nullptr;   // [B][8]
  mvo_step_result* d_res = nullptr;  // [B]
  // Up to TRK_DEPTH steps may be in flight (enqueued, not yet collected): the next step is enqueued while the previous one still
  // runs, so the stream never waits for the host between two steps (measured: 3.7 ms median from a step's last kernel to the
  // next step's first one when every step was collected before the next was enqueued).  Results land in a ring of pinned buffers.
  mvo_step_result* h_res[2] = {nullptr, nullptr};  // pinned, [TRK_DEPTH][B]
  int* h_err[2] = {nullptr, nullptr};              // pinned [1] each
  mvo_step_result* last_res = nullptr;             // [B] results of the step collected last (host-side forecast)
  int policy = 0;              // 0: the reference's key-frame policy, 1: key-frame branch on every tracked frame (worst case),
                               // 2: never a key-frame (the always-on part of the step: LK + PnP)
  int npending = 0, head = 0;    // steps in flight; ring index of the oldest
  hipEvent_t ev_done[2] = {nullptr, nullptr};
  hipEvent_t ev_lk = nullptr;   // this context's latest LK launch has finished
  // output side (mvo_batch_enable_output): what MonoVO::image_callback derives from the tracker's result on every frame
  bool out_on = false;
  int map_cap = 0, path_cap = 0;
  float* d_cloud = nullptr;        // [B][map_cap][3] the slot's Map as PointCloud2 payload: (z, -x, -y) float32 per landmark, id order
  int* d_n_cloud = nullptr;        // [B]
  double* d_path = nullptr;        // [B][path_cap][7] nav_msgs/Path poses: position xyz, orientation xyzw
  int* d_n_path = nullptr;         // [B]
  mvo_ros_pose* d_ros = nullptr;   // [B] last_pose_ in REP-103 + tracking_valid_
};

// The LK kernel is the one launch of a step that fills the GPU by itself (persistent wavefronts on every SIMD).  Two of
// them from different contexts cannot run side by side - the later one's wavefronts only find room as the earlier one's
// leave - so they take turns explicitly: a context's LK launch waits for the LK launch enqueued before it, whichever
// context that was (process-wide, one process per GPU).  MVO_LK_TURNS=0 turns the ordering off.
static std::mutex g_lk_mu;
static hipEvent_t g_lk_last_dev[32] = {};   // per device: a process that drives several GPUs orders each one's launches separately
static hipEvent_t& lk_last(const mvo_ctx* ctx) { return g_lk_last_dev[(unsigned)ctx->device % 32u]; }   // ordinal resolved by mvo_create
static bool lk_turns() {
  static const bool on = !(getenv("MVO_LK_TURNS") && atoi(getenv("MVO_LK_TURNS")) == 0);
  return on;
}

#define TRK_DEPTH 2
#define TRK_ERR_KEYPOINTS 1   // a slot's key-points exceeded max_points (clamped)
#define TRK_ERR_CAND 2        // FAST candidates exceeded the candidate capacity (clamped)
#define TRK_ERR_KPCAP 4       // dense key-point capacity exceeded (clamped)
#define TRK_ERR_MAPCAP 8      // a slot's landmark cloud or path exceeded its capacity (entries beyond it dropped)

// ---------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------
// LK work list: exclusive scan of the track counts of the slots that are still tracking (one workgroup).
__global__ __launch_bounds__(1024) void trk_worklist_scan_kernel(const int* __restrict__ state, int* __restrict__ npts, int B, int maxpts,
                                                                 int* __restrict__ pt_base, int* __restrict__ work_ctr,
                                                                 int* __restrict__ flags, mvo_step_result* __restrict__ res) {
  __shared__ int s_w[16];
  __shared__ int s_run;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_run = 0;
  if (threadIdx.x < 8) work_ctr[threadIdx.x] = 0;   // [0]: the claim counter of lk_track_kernel's persistent wavefronts (the rest is spare)
  __syncthreads();
  for (int s0 = 0; s0 < B; s0 += 1024) {
    const int s = s0 + threadIdx.x;
    int v = 0;
    if (s < B) {
      v = state[s] == MVO_TRACK_TRACKING ? min(max(npts[s], 0), maxpts) : 0;
      npts[s] = v;        // a LOST stream feeds nothing to LK (Tracker::update returns at once)
      flags[s] = 0;
      mvo_step_result z;
      memset(&z, 0, sizeof(z));
      z.n_prev = v;
      res[s] = z;
    }
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
    if (s < B) pt_base[s] = off + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int k = 0; k < 16; k++) t += s_w[k];
      s_run += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) pt_base[B] = s_run;
}

// The work list proper: item w = slot << 16 | point.  Every point is tracked on its own (results are written by point index),
// so the ORDER of a slot's items is free - and it matters: a wavefront tracks four consecutive items and iterates until the
// slowest of them has converged, and its four windows come from the same cache lines when the points are neighbours.  Tracks
// are in key-point order (pyramid level, then retainBest's permutation: spatially random), so each slot's items are sorted by
// the Morton code of their 16 x 16-pixel cell (bitonic sort in LDS; up to 8192 points, beyond that the order stays as it is).
#define TRK_SORT_MAX 8192
#ifndef TRK_SORT_SHIFT
#define TRK_SORT_SHIFT 4   // log2 of the cell edge in pixels (16: 1.53 ms per 302 k points; 32: 1.55; 64: 1.56; 128: 1.60; unsorted 1.79)
#endif
__global__ __launch_bounds__(256) void trk_worklist_sort_kernel(const int* __restrict__ pt_base, const float* __restrict__ prev_pts, int maxpts,
                                                                int* __restrict__ work_item) {
  __shared__ unsigned key[TRK_SORT_MAX];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int b = pt_base[s], n = pt_base[s + 1] - b;
  if (n <= 0) return;
  if (n > TRK_SORT_MAX) {
    for (int i = tid; i < n; i += 256) work_item[b + i] = (s << 16) | i;
    return;
  }
  int np2 = 64;
  while (np2 < n) np2 <<= 1;
  const float* pts = prev_pts + (size_t)s * maxpts * 2;
  for (int i = tid; i < np2; i += 256) {
    unsigned k = 0xFFFFFFFFu;   // padding sorts last
    if (i < n) {
      const float x = pts[2 * i], y = pts[2 * i + 1];
      unsigned cx = (unsigned)min(max((int)x, 0), 4095) >> TRK_SORT_SHIFT, cy = (unsigned)min(max((int)y, 0), 4095) >> TRK_SORT_SHIFT;   // <= 8 bits each
      unsigned m = 0;
#pragma unroll
      for (int bit = 0; bit < 8; bit++) m |= ((cx >> bit) & 1u) << (2 * bit) | ((cy >> bit) & 1u) << (2 * bit + 1);
      k = (m << 16) | (unsigned)i;
    }
    key[i] = k;
  }
  __syncthreads();
  for (int size = 2; size <= np2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (np2 >> 1); t += 256) {
        const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1)), hi = lo | stride;
        const unsigned a = key[lo], c = key[hi];
        const bool up = (lo & size) == 0;
        if ((a > c) == up) { key[lo] = c; key[hi] = a; }
      }
      __syncthreads();
    }
  for (int i = tid; i < n; i += 256) work_item[b + i] = (s << 16) | (int)(key[i] & 0xFFFFu);
}

// After H / F: has_parallax (src/tracker.cpp:253-265, divisions unguarded as there) and the ordered list of the slots
// that add a key-frame.  One workgroup.
__global__ __launch_bounds__(1024) void trk_policy_parallax_kernel(const int* __restrict__ n_hf, const int* __restrict__ res_h,
                                                                   const int* __restrict__ res_f, int B, double f_inlier_thresh,
                                                                   double model_score_thresh, int policy, int* __restrict__ flags,
                                                                   mvo_step_result* __restrict__ res, int* __restrict__ kf_list,
                                                                   int* __restrict__ nkf) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int s0 = 0; s0 < B; s0 += 1024) {
    const int s = s0 + threadIdx.x;
    bool kf = false;
    if (s < B && n_hf[s] > 0) {
      const int sh = res_h[8 * s] ? res_h[8 * s + 1] : 0;
      const int sf = res_f[8 * s] ? res_f[8 * s + 1] : 0;
      res[s].score_h = sh; res[s].score_f = sf;
      kf = true;
      if ((double)sf / (double)n_hf[s] < f_inlier_thresh) kf = false;
      else {
        const double model_score = (double)sh / (double)sf;   // inf / NaN as in the reference
        if (model_score > model_score_thresh) kf = false;
      }
      if (policy == 1) kf = true;
      if (kf) flags[s] |= MVO_STEP_KEYFRAME;
    }
    const unsigned long long m = __ballot(kf);
    const int pre = __popcll(m & ((1ull << lane) - 1));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; w++) off += s_wave[w];
    if (kf) kf_list[off + pre] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < 16; w++) t += s_wave[w];
      s_base += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *nkf = s_base;
}

// ORB level 0 of compact ORB slot j <- the ring frame of stream slot kf_list[j] (same image, same pitch)
__global__ __launch_bounds__(256) void trk_gather_orb0_kernel(const u8* __restrict__ lk0, size_t lk_slot_stride, int pitch, int h,
                                                              u8* __restrict__ orb0, size_t orb_slot_stride,
                                                              const int* __restrict__ kf_list, const int* __restrict__ nkf, int chunks) {
  const int n = *nkf;
  const size_t n16 = (size_t)pitch * h / 16;
  const unsigned total = (unsigned)n * (unsigned)chunks;
  for (unsigned wi = blockIdx.x; wi < total; wi += gridDim.x) {
    const int j = (int)(wi / (unsigned)chunks), c = (int)(wi - (unsigned)j * (unsigned)chunks);
    const uint4* s = (const uint4*)(lk0 + (size_t)kf_list[j] * lk_slot_stride);
    uint4* d = (uint4*)(orb0 + (size_t)j * orb_slot_stride);
    for (size_t i = (size_t)c * 256 + threadIdx.x; i < n16; i += (size_t)chunks * 256) d[i] = s[i];
  }
}

// dense ORB output of compact slot j -> matcher train side / key-point positions of stream slot kf_list[j]
__global__ __launch_bounds__(256) void trk_scatter_kp_kernel(const mvo_keypoint* __restrict__ kp, const u8* __restrict__ desc,
                                                             const int* __restrict__ kp_base, const int* __restrict__ kf_list,
                                                             const int* __restrict__ nkf, int maxpts, int kp_cap, u8* __restrict__ t_desc,
                                                             int* __restrict__ t_n, float* __restrict__ kp_xy, int* __restrict__ err,
                                                             mvo_step_result* __restrict__ res) {
  const int j = blockIdx.y;
  if (j >= *nkf) return;
  const int slot = kf_list[j];
  const int b0 = min(kp_base[j], kp_cap);
  int n = min(kp_base[j + 1], kp_cap) - b0;
  if (n > maxpts) { n = maxpts; if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(err, TRK_ERR_KEYPOINTS); }
  if (blockIdx.x == 0 && threadIdx.x == 0) { t_n[slot] = n; res[slot].n_keypoints = n; }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n * 8; i += gridDim.x * 256) {
    const int k = i >> 3, q = i & 7;
    ((unsigned*)t_desc)[((size_t)slot * maxpts + k) * 8 + q] = ((const unsigned*)desc)[(size_t)(b0 + k) * 8 + q];
    if (q == 0) {
      kp_xy[2 * ((size_t)slot * maxpts + k)] = kp[b0 + k].x;
      kp_xy[2 * ((size_t)slot * maxpts + k) + 1] = kp[b0 + k].y;
    }
  }
}

// capacity checks that the host did on counts it had waited for
__global__ void trk_orb_capcheck_kernel(const int* __restrict__ slot_base, const int* __restrict__ kp_base, const int* __restrict__ nkf,
                                        int cand_cap, int kp_cap, int* __restrict__ err) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int n = max(*nkf, 0);
    if (slot_base[n] > cand_cap) atomicOr(err, TRK_ERR_CAND);
    if (kp_base[n] > kp_cap) atomicOr(err, TRK_ERR_KPCAP);
  }
}

// ---- output side (src/mono_vo.cpp:117-152, src/utils.cpp:85-243), per slot on the device --------------------------
// Map::add_landmark of Tracker::add_new_keyframe (src/tracker.cpp:211-227): matches are visited in order; a valid
// (cheirality) match whose key-frame observation has no landmark yet creates one, and Map::get_landmark_points returns
// them in id = creation order (std::map<long, Landmark>).  The slot's cloud is that list already in the wire format of
// points3d_to_pointcloud_msg (src/utils.cpp:229-237): ROS x = cv z, y = -cv x, z = -cv y, three float32 per point.
// Runs before trk_assign_promote_kernel (which overwrites the key-frame's landmark flags), one workgroup per list entry.
__global__ __launch_bounds__(1024) void trk_map_append_kernel(const int* __restrict__ kf_list, const int* __restrict__ nkf,
                                                              const mvo_match* __restrict__ matches, const int* __restrict__ n_matches,
                                                              const u8* __restrict__ tri_ok, const float* __restrict__ tri,
                                                              const u8* __restrict__ kf_has, int cap, float* __restrict__ cloud,
                                                              int* __restrict__ n_cloud, int map_cap, int* __restrict__ err) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  if ((int)blockIdx.x >= *nkf) return;
  const int slot = kf_list[blockIdx.x], lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nm = min(max(n_matches[slot], 0), cap);
  const size_t b = (size_t)slot * cap;
  float* out = cloud + (size_t)slot * map_cap * 3;
  if (threadIdx.x == 0) s_base = n_cloud[slot];
  __syncthreads();
  for (int i0 = 0; i0 < nm; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    const bool add = i < nm && tri_ok[b + i] && !kf_has[b + matches[b + i].query_idx];
    const unsigned long long m = __ballot(add);
    const int pre = __popcll(m & ((1ull << lane) - 1));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; w++) off += s_wave[w];
    if (add) {
      const int o = off + pre;
      if (o < map_cap) { out[3 * o] = tri[3 * (b + i) + 2]; out[3 * o + 1] = -tri[3 * (b + i)]; out[3 * o + 2] = -tri[3 * (b + i) + 1]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int tt = 0;
      for (int w = 0; w < 16; w++) tt += s_wave[w];
      s_base += tt;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (s_base > map_cap) { atomicOr(err, TRK_ERR_MAPCAP); s_base = map_cap; }
    n_cloud[slot] = s_base;
  }
}

// affine3d_to_odometry_msg (src/utils.cpp:85-129): camera pose pose_wc = (R_cw, t_cw)^-1 conjugated into REP-103 with
// M = [0 0 1; -1 0 0; 0 -1 0], quaternion by tf2::Matrix3x3::getRotation + normalize.
__device__ inline void trk_pose_cw_to_ros(const double* pose_cw /* rvec, tvec */, double pos[3], double q[4]) {
  double R[9];
  trk_rodrigues(pose_cw, R);
  const double* t = pose_cw + 3;
  // pose_wc: R_wc = R^T, t_wc = -R^T t
  double Rw[9], tw[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) Rw[3 * i + j] = R[3 * j + i];
    tw[i] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
  }
  // M Rw M^T: row / column permutation with signs; ros axes (x, y, z) = cv (z, -x, -y)
  const int ax[3] = {2, 0, 1};
  const double sg[3] = {1.0, -1.0, -1.0};
  double m[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) m[3 * i + j] = sg[i] * sg[j] * Rw[3 * ax[i] + ax[j]];
  for (int i = 0; i < 3; i++) pos[i] = sg[i] * tw[ax[i]];
  const double trace = m[0] + m[4] + m[8];
  double x, y, z, w;
  if (trace > 0.0) {
    double s = sqrt(trace + 1.0);
    w = s * 0.5;
    s = 0.5 / s;
    x = (m[7] - m[5]) * s; y = (m[2] - m[6]) * s; z = (m[3] - m[1]) * s;
  } else {
    const int i = m[0] < m[4] ? (m[4] < m[8] ? 2 : 1) : (m[0] < m[8] ? 2 : 0);
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    double s = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
    double v[3];
    v[i] = s * 0.5;
    s = 0.5 / s;
    w = (m[3 * k + j] - m[3 * j + k]) * s;
    v[j] = (m[3 * j + i] + m[3 * i + j]) * s;
    v[k] = (m[3 * k + i] + m[3 * i + k]) * s;
    x = v[0]; y = v[1]; z = v[2];
  }
  const double n = sqrt(x * x + y * y + z * z + w * w);
  q[0] = x / n; q[1] = y / n; q[2] = z / n; q[3] = w / n;
}

// The pose bookkeeping of MonoVO::image_callback (src/mono_vo.cpp:119-148) for every slot after its Tracker::update: LOST ->
// tracking_valid_ = false and the last pose is kept; a returned pose becomes last_pose_ (tracking_valid_ = true); while
// tracking is valid one PoseStamped goes onto the path (a frame without a pose repeats the last one, as there).
__global__ __launch_bounds__(256) void trk_output_kernel(const int* __restrict__ state, const int* __restrict__ flags, const double* __restrict__ pose,
                                                         int B, mvo_ros_pose* __restrict__ ros, double* __restrict__ path, int* __restrict__ n_path,
                                                         int path_cap, int* __restrict__ err) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= B) return;
  const int st = state[s];
  mvo_ros_pose r = ros[s];
  if (st == MVO_TRACK_LOST) r.tracking_valid = 0;
  else if (flags[s] & MVO_STEP_POSE) {
    trk_pose_cw_to_ros(pose + 8 * s, r.position, r.orientation);
    r.tracking_valid = 1;
    r.has_pose = 1;
  }
  ros[s] = r;
  if (r.tracking_valid && r.has_pose) {
    const int n = n_path[s];
    if (n < path_cap) {
      double* o = path + ((size_t)s * path_cap + n) * 7;
      for (int k = 0; k < 3; k++) o[k] = r.position[k];
      for (int k = 0; k < 4; k++) o[3 + k] = r.orientation[k];
      n_path[s] = n + 1;
    } else atomicOr(err, TRK_ERR_MAPCAP);
  }
}

// ---- landmark hand-over of Tracker::add_new_keyframe (src/tracker.cpp:211-227) over the key-frame list ------------
// Matches are visited in order in the reference, so when several share a train index the LAST valid one decides that
// key-point's landmark: winner[t] = max valid match index (phase 1), then per key-point take the winner (phase 2).
__global__ __launch_bounds__(256) void trk_winner_clear_kernel(const int* __restrict__ kf_list, const int* __restrict__ nkf,
                                                               const int* __restrict__ n_kp, int cap, int* __restrict__ winner) {
  const int j = blockIdx.y;
  if (j >= *nkf) return;
  const int slot = kf_list[j];
  const int n = min(max(n_kp[slot], 0), cap);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) winner[(size_t)slot * cap + i] = -1;
}

__global__ __launch_bounds__(256) void trk_winner_kernel(const mvo_match* __restrict__ matches, const int* __restrict__ n_matches,
                                                         const u8* __restrict__ valid, int cap, const int* __restrict__ kf_list,
                                                         const int* __restrict__ nkf, int* __restrict__ winner) {
  const int j = blockIdx.y;
  if (j >= *nkf) return;
  const int slot = kf_list[j];
  const int n = min(max(n_matches[slot], 0), cap);
  const size_t b = (size_t)slot * cap;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    if (valid[b + i]) atomicMax(&winner[b + matches[b + i].train_idx], i);
}

// Phase 2 + promotion of the new key-frame, one workgroup per list entry: landmarks of the current key-points, their
// ordered compaction into the next frame's tracks (Frame::get_points_2d(WITH_LANDMARKS)), then current frame -> key-frame
// (descriptors, positions, per-observation landmarks, pose) and tracking_count_from_keyframe_ = 0.
__global__ __launch_bounds__(1024) void trk_assign_promote_kernel(const int* __restrict__ kf_list, const int* __restrict__ nkf,
                                                                  const mvo_match* __restrict__ matches, const int* __restrict__ winner,
                                                                  const int* __restrict__ n_kp, float* __restrict__ kp_xy,
                                                                  u8* __restrict__ kf_has, float* __restrict__ kf_lm,
                                                                  const float* __restrict__ tri, const u8* __restrict__ tri_ok,
                                                                  const int* __restrict__ n_matches, int cap, u8* __restrict__ cur_has,
                                                                  float* __restrict__ cur_lm, float* __restrict__ trk_xy,
                                                                  float* __restrict__ trk_lm, float* __restrict__ trk_kf,
                                                                  int* __restrict__ n_trk, float* __restrict__ kfkp_xy,
                                                                  u8* __restrict__ q_desc, const u8* __restrict__ t_desc,
                                                                  int* __restrict__ n_q, double* __restrict__ kf_pose,
                                                                  const double* __restrict__ pose, int* __restrict__ count,
                                                                  mvo_step_result* __restrict__ res) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  __shared__ int s_tri;
  if ((int)blockIdx.x >= *nkf) return;
  const int slot = kf_list[blockIdx.x], lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = min(max(n_kp[slot], 0), cap);
  const size_t b = (size_t)slot * cap;
  if (threadIdx.x == 0) { s_base = 0; s_tri = 0; }
  __syncthreads();
  int tri_cnt = 0;
  const int nm = min(max(n_matches[slot], 0), cap);
  for (int i = threadIdx.x; i < nm; i += 1024) tri_cnt += tri_ok[b + i] ? 1 : 0;
  if (tri_cnt) atomicAdd(&s_tri, tri_cnt);
  for (int t0 = 0; t0 < n; t0 += 1024) {
    const int t = t0 + threadIdx.x;
    bool has = false;
    float lx = 0, ly = 0, lz = 0;
    if (t < n) {
      const int wi = winner[b + t];
      if (wi >= 0) {
        const int q = matches[b + wi].query_idx;
        has = true;
        if (kf_has[b + q]) { lx = kf_lm[3 * (b + q)]; ly = kf_lm[3 * (b + q) + 1]; lz = kf_lm[3 * (b + q) + 2]; }
        else { lx = tri[3 * (b + wi)]; ly = tri[3 * (b + wi) + 1]; lz = tri[3 * (b + wi) + 2]; }
      }
      cur_has[b + t] = has ? 1 : 0;
      cur_lm[3 * (b + t)] = lx; cur_lm[3 * (b + t) + 1] = ly; cur_lm[3 * (b + t) + 2] = lz;
    }
    const unsigned long long m = __ballot(has);
    const int pre = __popcll(m & ((1ull << lane) - 1));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; w++) off += s_wave[w];
    if (has) {
      const size_t o = b + off + pre;
      const float x = kp_xy[2 * (b + t)], y = kp_xy[2 * (b + t) + 1];
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
  // every read of the old key-frame's landmarks (kf_has / kf_lm) is done: the current frame becomes the key-frame
  __syncthreads();
  for (int t = threadIdx.x; t < n; t += 1024) {
    kf_has[b + t] = cur_has[b + t];
    kf_lm[3 * (b + t)] = cur_lm[3 * (b + t)]; kf_lm[3 * (b + t) + 1] = cur_lm[3 * (b + t) + 1]; kf_lm[3 * (b + t) + 2] = cur_lm[3 * (b + t) + 2];
    kfkp_xy[2 * (b + t)] = kp_xy[2 * (b + t)]; kfkp_xy[2 * (b + t) + 1] = kp_xy[2 * (b + t) + 1];
  }
  {
    const uint4* s = (const uint4*)(t_desc + b * 32);
    uint4* d = (uint4*)(q_desc + b * 32);
    for (int i = threadIdx.x; i < n * 2; i += 1024) d[i] = s[i];
  }
  if (threadIdx.x < 8) kf_pose[8 * slot + threadIdx.x] = pose[8 * slot + threadIdx.x];
  if (threadIdx.x == 0) {
    n_trk[slot] = s_base;
    n_q[slot] = n;
    count[slot] = 0;
    res[slot].n_matches = nm;
    res[slot].n_triangulated = s_tri;
  }
}

// Slots that tracked but added no key-frame: the LK survivors are the next frame's observations (src/tracker.cpp:331);
// then the result records of all slots.
__global__ __launch_bounds__(256) void trk_finalize_kernel(const int* __restrict__ state, const int* __restrict__ count,
                                                           const int* __restrict__ flags, const int* __restrict__ ncur, int maxpts,
                                                           const float* __restrict__ cur_pts, const float* __restrict__ cur_lm,
                                                           const float* __restrict__ cur_kf, float* __restrict__ trk_xy,
                                                           float* __restrict__ trk_lm, float* __restrict__ trk_kf, int* __restrict__ npts,
                                                           mvo_step_result* __restrict__ res) {
  const int slot = blockIdx.x;
  const int f = flags[slot], st = state[slot];
  const size_t b = (size_t)slot * maxpts;
  if (st == MVO_TRACK_TRACKING && (f & (MVO_STEP_POSE | MVO_STEP_PNP_FAILED)) && !(f & MVO_STEP_KEYFRAME)) {
    const int n = min(max(ncur[slot], 0), maxpts);
    for (int i = threadIdx.x; i < 2 * n; i += 256) { trk_xy[2 * b + i] = cur_pts[2 * b + i]; trk_kf[2 * b + i] = cur_kf[2 * b + i]; }
    for (int i = threadIdx.x; i < 3 * n; i += 256) trk_lm[3 * b + i] = cur_lm[3 * b + i];
    if (threadIdx.x == 0) npts[slot] = n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (st != MVO_TRACK_TRACKING) npts[slot] = 0;
    res[slot].state = st;
    res[slot].flags = f;
    res[slot].tracking_count = count[slot];
    res[slot].n_tracks = npts[slot];
  }
}

// ---------------------------------------------------------------------------------------------------
// state
// ---------------------------------------------------------------------------------------------------
static int trk_create(mvo_ctx* ctx) {
  PipeState* p = ctx->pipe;
  if (p->trk) return MVO_OK;
  TrackState* t = new TrackState();
  p->trk = t;
  const int B = ctx->B;
  const size_t np = (size_t)B * ctx->maxpts;
  MVO_HIP(hipMalloc(&t->d_state, B * sizeof(int)));
  MVO_HIP(hipMalloc(&t->d_count, B * sizeof(int)));
  MVO_HIP(hipMalloc(&t->d_flags, B * sizeof(int)));
  MVO_HIP(hipMalloc(&t->d_n_pnp, B * sizeof(int)));
  MVO_HIP(hipMalloc(&t->d_n_hf, B * sizeof(int)));
  MVO_HIP(hipMalloc(&t->d_kf_list, B * sizeof(int)));
  MVO_HIP(hipMalloc(&t->d_nkf, sizeof(int)));
  MVO_HIP(hipMalloc(&t->d_pt_base, (B + 1) * sizeof(int)));
  MVO_HIP(hipMalloc(&t->d_work_slot, np * sizeof(int)));
  MVO_HIP(hipMalloc(&t->d_work_ctr, 8 * sizeof(int)));
  MVO_HIP(hipMalloc(&t->d_err, sizeof(int)));
  MVO_HIP(hipMalloc(&t->d_hf_ctr, 2 * sizeof(int)));
  MVO_HIP(hipMalloc(&t->d_mask_f, np));
  MVO_HIP(hipMalloc(&t->d_model_f, (size_t)B * 16 * sizeof(double)));
  MVO_HIP(hipMalloc(&t->d_result_f, (size_t)B * 8 * sizeof(int)));
  MVO_HIP(hipMalloc(&t->d_res, (size_t)B * sizeof(mvo_step_result)));
  for (int i = 0; i < TRK_DEPTH; i++) {
    MVO_HIP(hipHostMalloc(&t->h_res[i], (size_t)B * sizeof(mvo_step_result), hipHostMallocDefault));
    MVO_HIP(hipHostMalloc(&t->h_err[i], sizeof(int), hipHostMallocDefault));
    *t->h_err[i] = 0;
    MVO_HIP(hipEventCreateWithFlags(&t->ev_done[i], hipEventDisableTiming));
  }
  t->last_res = (mvo_step_result*)calloc((size_t)B, sizeof(mvo_step_result));
  MVO_HIP(hipEventCreateWithFlags(&t->ev_lk, hipEventDisableTiming));
  MVO_HIP(hipMemsetAsync(t->d_state, 0, B * sizeof(int), ctx->stream));
  MVO_HIP(hipMemsetAsync(t->d_count, 0, B * sizeof(int), ctx->stream));
  MVO_HIP(hipMemsetAsync(t->d_err, 0, sizeof(int), ctx->stream));
  MVO_HIP(hipMemsetAsync(t->d_nkf, 0, sizeof(int), ctx->stream));
  MVO_HIP(hipMemsetAsync(t->d_result_f, 0, (size_t)B * 8 * sizeof(int), ctx->stream));
  return MVO_OK;
}

void trk_destroy(mvo_ctx* ctx) {
  PipeState* p = ctx->pipe;
  if (!p) return;
  if (p->s_up && !p->up_shared) { (void)hipStreamSynchronize(p->s_up); (void)hipStreamDestroy(p->s_up); }
  p->s_up = nullptr;
  for (auto e : p->ev_up) (void)hipEventDestroy(e);
  for (auto e : p->ev_rd) (void)hipEventDestroy(e);
  p->ev_up.clear(); p->ev_rd.clear();
  TrackState* t = p->trk;
  if (!t) return;
  void* dev[] = {t->d_state, t->d_count, t->d_flags, t->d_n_pnp, t->d_n_hf, t->d_kf_list, t->d_nkf, t->d_pt_base, t->d_work_slot,
                 t->d_work_ctr, t->d_err, t->d_hf_ctr, t->d_mask_f, t->d_model_f, t->d_result_f, t->d_res};
  for (void* q : dev) (void)hipFree(q);
  void* outb[] = {t->d_cloud, t->d_n_cloud, t->d_path, t->d_n_path, t->d_ros};
  for (void* q : outb) if (q) (void)hipFree(q);
  for (int i = 0; i < TRK_DEPTH; i++) {
    if (t->h_res[i]) (void)hipHostFree(t->h_res[i]);
    if (t->h_err[i]) (void)hipHostFree(t->h_err[i]);
    if (t->ev_done[i]) (void)hipEventDestroy(t->ev_done[i]);
  }
  free(t->last_res);
  if (t->ev_lk) {
    std::lock_guard<std::mutex> lock(g_lk_mu);
    if (lk_last(ctx) == t->ev_lk) lk_last(ctx) = nullptr;
    (void)hipEventDestroy(t->ev_lk);
  }
  delete t;
  p->trk = nullptr;
}

// The upload stream and the ring's events.  Called when the context is created (pipe_state_create): a frame-batch context then
// always occupies two consecutive stream slots - compute, upload - whatever order the caller creates contexts and issues
// uploads in, and HIP's stream i -> hardware queue i mod 4 puts the compute streams of four contexts on queues 0, 2, 0, 2:
// the best layout measured (all four on distinct queues: 52 k instead of 73 k frames/s).
int trk_ring_events(mvo_ctx* ctx) {
  PipeState* p = ctx->pipe;
  if (!p->ev_up.empty()) return MVO_OK;
  if (const char* e = getenv("MVO_UPLOAD_STREAM")) p->up_shared = atoi(e) == 0;   // 0: uploads go on the compute stream
  if (p->up_shared) p->s_up = ctx->stream;
  else MVO_HIP(hipStreamCreateWithFlags(&p->s_up, hipStreamNonBlocking));
  p->ev_up.resize(p->ring); p->ev_rd.resize(p->ring);
  p->up_pending.assign(p->ring, 0); p->rd_pending.assign(p->ring, 0);
  for (int i = 0; i < p->ring; i++) {
    MVO_HIP(hipEventCreateWithFlags(&p->ev_up[i], hipEventDisableTiming));
    MVO_HIP(hipEventCreateWithFlags(&p->ev_rd[i], hipEventDisableTiming));
  }
  return MVO_OK;
}

// ---------------------------------------------------------------------------------------------------
// API
// ---------------------------------------------------------------------------------------------------
extern "C" int mvo_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return MVO_E_ARG;
  return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? MVO_OK : MVO_E_HIP;
}

extern "C" int mvo_host_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? MVO_OK : MVO_E_HIP; }

extern "C" int mvo_batch_set_policy(mvo_ctx* ctx, int policy) {
  if (!ctx || !ctx->pipe || ctx->pipe->ring <= 0 || policy < 0 || policy > 2) return MVO_E_ARG;
  int rc = trk_create(ctx);
  if (rc) return rc;
  ctx->pipe->trk->policy = policy;
  return MVO_OK;
}

// All B frames of ring entry `frame_idx` in one asynchronous copy on the upload stream.  `frames` holds B mono8 images,
// `slot_stride` bytes apart, rows `stride` bytes apart; pinned memory (mvo_host_alloc / hipHostRegister) makes the copy
// overlap the step that is running.  The copy waits for the step that last read this ring entry.
extern "C" int mvo_batch_upload_async(mvo_ctx* ctx, int frame_idx, const uint8_t* frames, int w, int h, int stride, size_t slot_stride) {
  if (!ctx || !ctx->pipe || !frames) return MVO_E_ARG;
  PipeState* p = ctx->pipe;
  if (p->ring <= 0 || frame_idx < 0 || frame_idx >= p->ring) return MVO_E_ARG;
  if (w > ctx->maxw || h > ctx->maxh || w < 32 || h < 32 || stride < w || slot_stride < (size_t)stride * (h - 1) + w) return MVO_E_ARG;
  if (p->w == 0) { p->w = w; p->h = h; p->pitch = align_up(w, 64); }
  if (w != p->w || h != p->h) { ctx->set_error("all ring frames must share one size"); return MVO_E_ARG; }
  int rc = trk_ring_events(ctx);
  if (rc) return rc;
  if (p->rd_pending[frame_idx]) { MVO_HIP(hipStreamWaitEvent(p->s_up, p->ev_rd[frame_idx], 0)); p->rd_pending[frame_idx] = 0; }
  if (frame_idx == p->prev_entry) p->prev_entry = -1;   // the previous frame's image is gone: the next step must be a seed (see mvo_batch_track_async)
  u8* dst = p->d_ring + (size_t)frame_idx * ctx->B * p->frame_bytes;
  if (stride == p->pitch && slot_stride == p->frame_bytes) {
    // one linear copy; the last slot stops at its last pixel (the caller guarantees no more than that to be readable)
    MVO_HIP(hipMemcpyAsync(dst, frames, p->frame_bytes * (ctx->B - 1) + (size_t)stride * (h - 1) + w, hipMemcpyDefault, p->s_up));
  } else if (slot_stride == (size_t)stride * h) {
    // one 2-D copy: the B images are one tall image of B*h rows on the host; on the device the slots are frame_bytes
    // = pitch * maxh apart, which equals pitch * h only when h == maxh
    if ((size_t)p->pitch * h == p->frame_bytes) {
      MVO_HIP(hipMemcpy2DAsync(dst, p->pitch, frames, stride, w, (size_t)h * ctx->B, hipMemcpyDefault, p->s_up));
    } else {
      for (int s = 0; s < ctx->B; s++)
        MVO_HIP(hipMemcpy2DAsync(dst + (size_t)s * p->frame_bytes, p->pitch, frames + (size_t)s * slot_stride, stride, w, h,
                                 hipMemcpyDefault, p->s_up));
    }
  } else {
    for (int s = 0; s < ctx->B; s++)
      MVO_HIP(hipMemcpy2DAsync(dst + (size_t)s * p->frame_bytes, p->pitch, frames + (size_t)s * slot_stride, stride, w, h,
                               hipMemcpyDefault, p->s_up));
  }
  MVO_HIP(hipEventRecord(p->ev_up[frame_idx], p->s_up));
  p->up_pending[frame_idx] = 1;
  return MVO_OK;
}

// Enqueue one tracker step of every slot on ring frame `frame_idx`; returns without waiting for the device.
//
// Level 0 of both LK pyramids is read IN PLACE from the frame ring: "prev" is the entry tracked (or seeded) last, "cur" is
// `frame_idx`; only levels 1.. are built (pyramid sets, ping-pong).  The entry of the previous frame therefore has to stay
// intact until this step's LK launch has run: ev_rd[prev] is recorded right after it, and an asynchronous upload into that
// entry waits for it.  A 2-entry ring works (the upload of frame k+1 overlaps everything of step k behind its LK launch), a
// 3-entry ring overlaps the upload with the whole step.
static int trk_step_enqueue(mvo_ctx* ctx, int frame_idx) {
  PipeState* p = ctx->pipe;
  MatchState* m = ctx->match;
  GeomState* g = ctx->geom;
  OrbState* o = ctx->orb;
  TrackState* t = p->trk;
  int rc;
  const int B = ctx->B, cap = ctx->maxpts;
  hipStream_t st = ctx->stream;
  const mvo_config& c = ctx->cfg;
  LkLevels L = lk_levels(p->w, p->h, c.lk_win, c.lk_max_level);
  const int prev_set = ctx->lk_cur, cur_set = ctx->lk_cur ^ 1;
  const int nb = (B + 255) / 256;
  const int prev_entry = p->prev_entry;
  const size_t entry_bytes = (size_t)B * p->frame_bytes;
  const u8* ring_prev = p->d_ring + (size_t)prev_entry * entry_bytes;
  const u8* ring_cur = p->d_ring + (size_t)frame_idx * entry_bytes;

  if ((rc = trk_wait_upload(ctx, frame_idx))) return rc;
  // ---- LK: levels 1.. of the "cur" pyramid from the ring entry; dense work list over the slots that are tracking -------
  { ProfScope ps(ctx, "lk_pyramid"); lk_build_pyramid(ctx, cur_set, L, B, st, ring_cur, p->frame_bytes); }
  {
    ProfScope ps(ctx, "lk_worklist");
    hipLaunchKernelGGL(trk_worklist_scan_kernel, dim3(1), dim3(1024), 0, st, t->d_state, ctx->d_npts, B, cap, t->d_pt_base, t->d_work_ctr,
                       t->d_flags, t->d_res);
    hipLaunchKernelGGL(trk_worklist_sort_kernel, dim3(B), dim3(256), 0, st, t->d_pt_base, ctx->d_prev_pts, cap, t->d_work_slot);
  }
  // upper bound of the work list, known on the host when the last step has been collected: a slot's track count is its n_tracks
  int items_bound = -1;
  if (t->res_valid && t->npending == 0) {
    long long sum = 0;
    for (int s = 0; s < B; s++) sum += std::min(std::max(t->last_res[s].n_tracks, 0), cap);
    items_bound = (int)std::min<long long>(sum, (long long)B * cap);
  }
  if (lk_turns()) {
    std::lock_guard<std::mutex> lock(g_lk_mu);
    hipEvent_t& last = lk_last(ctx);
    if (last && last != t->ev_lk) MVO_HIP(hipStreamWaitEvent(st, last, 0));
    { ProfScope ps(ctx, "lk_track");
      lk_track_device(ctx, prev_set, cur_set, L, B, cap, st, t->d_work_slot, t->d_pt_base, t->d_work_ctr, ring_prev, ring_cur, p->frame_bytes, items_bound); }
    MVO_HIP(hipEventRecord(t->ev_lk, st));
    last = t->ev_lk;
  } else {
    ProfScope ps(ctx, "lk_track");
    lk_track_device(ctx, prev_set, cur_set, L, B, cap, st, t->d_work_slot, t->d_pt_base, t->d_work_ctr, ring_prev, ring_cur, p->frame_bytes, items_bound);
  }
  MVO_HIP(hipEventRecord(p->ev_rd[prev_entry], st));   // the previous frame's ring entry may be overwritten from here on
  p->rd_pending[prev_entry] = 1;
  {
    ProfScope ps(ctx, "lk_filter");
    // + min_tracked_points -> LOST (src/tracker.cpp:292-296) at the end of each slot's workgroup
    const TrkLostPolicy lp{t->d_state, (long long)c.min_tracked_points, t->d_n_pnp, t->d_flags, t->d_res};
    lk_filter_compact_launch(ctx, st, &lp);
  }
  // ---- PnP on the slots that still track ----------------------------------------------------------------------------------
  {
    // + pose, ++tracking_count_from_keyframe_, should_add_keyframe (src/tracker.cpp:318-319) at the end of each slot's refine
    // workgroup: a separate 5 us kernel for it took 0.3-0.7 ms of stream time beside three other contexts (step anatomy)
    ProfScope ps(ctx, "pnp");
    const TrkKeyframePolicy kp{t->d_state, t->d_count, t->d_n_pnp, p->d_kf_pose, (long long)c.min_observations_before_triangulation,
                               (long long)c.max_tracking_after_keyframe, c.max_translation_from_keyframe, c.max_rotation_from_keyframe, t->policy,
                               t->d_n_hf, t->d_flags, t->d_res, t->d_hf_ctr};
    geom_pnp(ctx, B, p->d_cur_lm, p->d_cur_pts, t->d_n_pnp, p->K, p->dist, 100, 8.0f, 0.99, g->d_mask, g->d_model, g->d_result, g->d_inl,
             g->d_pose, st, &kp);
  }
  // ---- has_parallax on the slots whose key-frame test fired -----------------------------------------------------------
  // Persistent workgroups over a slot queue, as many as the host FORECASTS tests from the results of the step collected last
  // (should_add_keyframe's count / observation rules are known one frame ahead; the motion rule is not: a slot the
  // forecast missed is solved by the same workgroups a little later, never skipped).  The H workgroup needs 138 KB of
  // LDS, i.e. an empty CU: one workgroup per slot cost 1.4 + 1.8 ms of stream time on the ten steps in eleven where no slot
  // of the context tests at all.
  int hf_grid = B;
  if (t->res_valid && t->policy != 1) {
    int fc = 0;
    const long long ahead = 1 + t->npending;   // frames between the results in hand and the step being enqueued
    for (int s = 0; s < B; s++) {
      const mvo_step_result& r = t->last_res[s];
      fc += r.state == MVO_TRACK_TRACKING && ((long long)r.tracking_count + ahead > (long long)c.max_tracking_after_keyframe ||
                                                (long long)r.n_tracks < (long long)c.min_observations_before_triangulation);
    }
    hf_grid = t->policy == 2 ? 8 : fc + 8 + fc / 8;
  }
  { ProfScope ps(ctx, "ransac_h");
    geom_ransac_h(ctx, B, p->d_cur_kf, p->d_cur_pts, t->d_n_hf, c.ransac_reproj_thresh, 2000, 0.995, g->d_mask2, g->d_model2, g->d_result2, st,
                  t->d_hf_ctr, hf_grid); }
  { ProfScope ps(ctx, "ransac_f");
    geom_ransac_f(ctx, B, p->d_cur_kf, p->d_cur_pts, t->d_n_hf, c.ransac_reproj_thresh, 1000, 0.99, t->d_mask_f, t->d_model_f, t->d_result_f, st,
                  t->d_hf_ctr + 1, hf_grid); }
  hipLaunchKernelGGL(trk_policy_parallax_kernel, dim3(1), dim3(1024), 0, st, t->d_n_hf, g->d_result2, t->d_result_f, B, c.f_inlier_thresh,
                     c.model_score_thresh, t->policy, t->d_flags, t->d_res, t->d_kf_list, t->d_nkf);
  // ---- add_new_keyframe over the key-frame list --------------------------------------------------------------------------
  {
    ProfScope ps(ctx, "kf_gather");
    const int chunks = 16;
    hipLaunchKernelGGL(trk_gather_orb0_kernel, dim3(persist_grid((unsigned)B * chunks)), dim3(256), 0, st, ring_cur, p->frame_bytes, p->pitch,
                       p->h, o->d_pyr, o->slot_bytes, t->d_kf_list, t->d_nkf, chunks);
  }
  if ((rc = orb_run_device(ctx, p->w, p->h, B, t->d_nkf))) return rc;
  {
    ProfScope ps(ctx, "kf_scatter");
    hipLaunchKernelGGL(trk_orb_capcheck_kernel, dim3(1), dim3(64), 0, st, o->d_slot_base, o->d_kp_base, t->d_nkf, o->cand_cap, o->kp_cap, t->d_err);
    hipLaunchKernelGGL(trk_scatter_kp_kernel, dim3(16, B), dim3(256), 0, st, o->d_kp, o->d_desc, o->d_kp_base, t->d_kf_list, t->d_nkf, cap,
                       o->kp_cap, m->d_t, m->d_nt, p->d_kp_xy, t->d_err, t->d_res);
  }
  { ProfScope ps(ctx, "match"); match_device(ctx, B, cap, c.lowes_distance_ratio, t->d_kf_list, t->d_nkf); }
  {
    ProfScope ps(ctx, "triangulate");
    geom_triangulate_matches(ctx, B, cap, m->d_out, m->d_nout, p->d_kfkp_xy, p->d_kp_xy, p->d_kf_pose, g->d_pose, g->d_result, p->K, p->d_tri,
                             p->d_tri_ok, t->d_kf_list, t->d_nkf);
    hipLaunchKernelGGL(trk_winner_clear_kernel, dim3(4, B), dim3(256), 0, st, t->d_kf_list, t->d_nkf, m->d_nt, cap, p->d_winner);
    hipLaunchKernelGGL(trk_winner_kernel, dim3(4, B), dim3(256), 0, st, m->d_out, m->d_nout, p->d_tri_ok, cap, t->d_kf_list, t->d_nkf, p->d_winner);
    if (t->out_on)
      hipLaunchKernelGGL(trk_map_append_kernel, dim3(B), dim3(1024), 0, st, t->d_kf_list, t->d_nkf, m->d_out, m->d_nout, p->d_tri_ok, p->d_tri,
                         p->d_kf_has, cap, t->d_cloud, t->d_n_cloud, t->map_cap, t->d_err);
    hipLaunchKernelGGL(trk_assign_promote_kernel, dim3(B), dim3(1024), 0, st, t->d_kf_list, t->d_nkf, m->d_out, p->d_winner, m->d_nt, p->d_kp_xy,
                       p->d_kf_has, p->d_kf_lm, p->d_tri, p->d_tri_ok, m->d_nout, cap, p->d_cur_has, p->d_cur_lmk, ctx->d_prev_pts, p->d_lm,
                       p->d_kf_pts, ctx->d_npts, p->d_kfkp_xy, m->d_q, m->d_t, m->d_nq, p->d_kf_pose, g->d_pose, t->d_count, t->d_res);
  }
  // ---- prev_frame_ = new_frame for the others; results -------------------------------------------------------------------
  hipLaunchKernelGGL(trk_finalize_kernel, dim3(B), dim3(256), 0, st, t->d_state, t->d_count, t->d_flags, p->d_ncur, cap, p->d_cur_pts,
                     p->d_cur_lm, p->d_cur_kf, ctx->d_prev_pts, p->d_lm, p->d_kf_pts, ctx->d_npts, t->d_res);
  if (t->out_on)
    hipLaunchKernelGGL(trk_output_kernel, dim3(nb), dim3(256), 0, st, t->d_state, t->d_flags, g->d_pose, B, t->d_ros, t->d_path, t->d_n_path,
                       t->path_cap, t->d_err);
  const int rb = (t->head + t->npending) % TRK_DEPTH;   // result buffer of this step
  MVO_HIP(hipMemcpyAsync(t->h_res[rb], t->d_res, (size_t)B * sizeof(mvo_step_result), hipMemcpyDeviceToHost, st));
  MVO_HIP(hipMemcpyAsync(t->h_err[rb], t->d_err, sizeof(int), hipMemcpyDeviceToHost, st));
  MVO_HIP(hipEventRecord(t->ev_done[rb], st));
  // this step's own reads of its frame (the key-frame branch's gather) are over at ev_rd[frame_idx]; the next step's LK launch
  // re-records it, since the entry is then the "prev" image
  MVO_HIP(hipEventRecord(p->ev_rd[frame_idx], st));
  p->rd_pending[frame_idx] = 1;
  return MVO_OK;
}

extern "C" int mvo_batch_track_async(mvo_ctx* ctx, int frame_idx) {
  if (!ctx || !ctx->pipe) return MVO_E_ARG;
  PipeState* p = ctx->pipe;
  if (!p->seeded || frame_idx < 0 || frame_idx >= p->ring) { ctx->set_error("mvo_batch_track: not seeded / bad frame"); return MVO_E_ARG; }
  int rc = trk_create(ctx);
  if (rc) return rc;
  TrackState* t = p->trk;
  if (t->npending >= TRK_DEPTH) { ctx->set_error("mvo_batch_track_async: two steps are in flight already: collect the oldest first (mvo_batch_track_wait)"); return MVO_E_ARG; }
  if ((rc = trk_ring_events(ctx))) return rc;
  if (p->prev_entry < 0) {
    ctx->set_error("mvo_batch_track: the ring entry of the previous frame was overwritten before the next step (it is level 0 of the LK "
                   "template): keep it until the following step has been enqueued, or seed again");
    return MVO_E_ARG;
  }
  if (frame_idx == p->prev_entry) {
    ctx->set_error("mvo_batch_track: frame_idx is the ring entry of the previous frame (ring_frames >= 2: put the new frame into another entry)");
    return MVO_E_ARG;
  }
  rc = trk_step_enqueue(ctx, frame_idx);
  if (rc != MVO_OK) {
    // half a step is queued: drain it and refuse further steps until the caller seeds again (the per-stream state is undefined)
    const std::string why = ctx->err;
    (void)hipStreamSynchronize(ctx->stream);
    p->seeded = false;
    p->prev_entry = -1;
    ctx->set_error("mvo_batch_track_async failed mid-step (" + why + "); the tracker state is void: seed again");
    return rc;
  }
  ctx->lk_cur ^= 1;
  p->prev_entry = frame_idx;
  t->npending++;
  return MVO_OK;
}

// 1 when the step enqueued last has finished (mvo_batch_track_wait will not block), 0 while it runs.
extern "C" int mvo_batch_track_poll(mvo_ctx* ctx) {
  if (!ctx || !ctx->pipe || !ctx->pipe->trk || !ctx->pipe->trk->npending) return 1;
  return hipEventQuery(ctx->pipe->trk->ev_done[ctx->pipe->trk->head]) == hipSuccess ? 1 : 0;
}

extern "C" int mvo_batch_track_wait(mvo_ctx* ctx, mvo_step_result* out) {
  if (!ctx || !ctx->pipe || !ctx->pipe->trk) return MVO_E_ARG;
  TrackState* t = ctx->pipe->trk;
  if (!t->npending) { ctx->set_error("mvo_batch_track_wait: no step in flight"); return MVO_E_ARG; }
  const int rb = t->head;
  MVO_HIP(hipEventSynchronize(t->ev_done[rb]));
  t->head = (t->head + 1) % TRK_DEPTH;
  t->npending--;
  t->res_valid = true;
  memcpy(t->last_res, t->h_res[rb], (size_t)ctx->B * sizeof(mvo_step_result));
  if (out) memcpy(out, t->last_res, (size_t)ctx->B * sizeof(mvo_step_result));
  if (*t->h_err[rb]) {
    const int e = *t->h_err[rb];
    ctx->set_error(std::string("mvo_batch_track: device capacity exceeded (") + ((e & TRK_ERR_KEYPOINTS) ? "key-points > max_points " : "") +
                   ((e & TRK_ERR_CAND) ? "FAST candidates " : "") + ((e & TRK_ERR_KPCAP) ? "dense key-points " : "") + ((e & TRK_ERR_MAPCAP) ? "landmark cloud / path capacity" : "") + ")");
    MVO_HIP(hipMemsetAsync(t->d_err, 0, sizeof(int), ctx->stream));
    return MVO_E_CAPACITY;
  }
  return MVO_OK;
}

extern "C" int mvo_batch_track(mvo_ctx* ctx, int frame_idx, mvo_step_result* out) {
  int rc = mvo_batch_track_async(ctx, frame_idx);
  if (rc) return rc;
  return mvo_batch_track_wait(ctx, out);
}

// ---- single-stream convenience forms (SURVEY 8(b) export list) ------------------------------------------------------------
extern "C" int mvo_set_intrinsics(mvo_ctx* ctx, const double K[9], const double d[5]) { return mvo_batch_set_intrinsics(ctx, K, d); }

// Fused Tracker::update for a context of ONE stream (cfg.batch == 1, cfg.ring_frames >= 2): uploads the image (any
// encoding code of this header; colour must be replicated mono8, see mvo_lk_track) into the ring entry after the one used
// last and runs mvo_batch_track on it.  Seed once with mvo_batch_preload_frame(ctx, 0, 0, ...) + mvo_batch_seed(ctx, 0, ...)
// + mvo_batch_set_landmarks (the Initializer's hand-over).
extern "C" int mvo_tracker_step(mvo_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels, mvo_step_result* out) {
  if (!ctx || !ctx->pipe || !img || !out) return MVO_E_ARG;
  PipeState* p = ctx->pipe;
  if (ctx->B != 1 || p->ring < 2) { ctx->set_error("mvo_tracker_step: needs cfg.batch == 1 and cfg.ring_frames >= 2"); return MVO_E_ARG; }
  p->step_entry = (p->step_entry + 1) % p->ring;
  int rc = mvo_batch_preload_frame(ctx, 0, p->step_entry, img, w, h, stride, channels);
  if (rc) return rc;
  return mvo_batch_track(ctx, p->step_entry, out);
}

// ---- output side ---------------------------------------------------------------------------------
__global__ void trk_output_seed_kernel(mvo_ros_pose* __restrict__ ros, int* __restrict__ n_path, int slot, const float* __restrict__ lm, int n,
                                       float* __restrict__ cloud, int* __restrict__ n_cloud, int map_cap, int* __restrict__ err) {
  // the Initializer's hand-over (src/mono_vo.cpp:102-112): last_pose_ = identity, tracking valid, the map holds the seed landmarks
  const int i = blockIdx.x * 256 + threadIdx.x;
  float* out = cloud + (size_t)slot * map_cap * 3;
  if (i < n && i < map_cap) { out[3 * i] = lm[3 * i + 2]; out[3 * i + 1] = -lm[3 * i]; out[3 * i + 2] = -lm[3 * i + 1]; }
  if (i == 0) {
    mvo_ros_pose r;
    r.position[0] = r.position[1] = r.position[2] = 0.0;
    r.orientation[0] = r.orientation[1] = r.orientation[2] = 0.0; r.orientation[3] = 1.0;
    r.tracking_valid = 1; r.has_pose = 1;
    ros[slot] = r;
    n_path[slot] = 0;
    if (n > map_cap) atomicOr(err, TRK_ERR_MAPCAP);
    n_cloud[slot] = n < map_cap ? n : map_cap;
  }
}

extern "C" int mvo_batch_enable_output(mvo_ctx* ctx, int map_capacity, int path_capacity) {
  if (!ctx || !ctx->pipe || ctx->pipe->ring <= 0 || map_capacity < 1 || path_capacity < 1) return MVO_E_ARG;
  int rc = trk_create(ctx);
  if (rc) return rc;
  TrackState* t = ctx->pipe->trk;
  if (t->npending) { ctx->set_error("mvo_batch_enable_output: a step is in flight"); return MVO_E_ARG; }
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  void* old[] = {t->d_cloud, t->d_n_cloud, t->d_path, t->d_n_path, t->d_ros};
  for (void* q : old) if (q) (void)hipFree(q);
  t->d_cloud = nullptr; t->d_n_cloud = nullptr; t->d_path = nullptr; t->d_n_path = nullptr; t->d_ros = nullptr;
  t->out_on = false;
  const size_t B = (size_t)ctx->B;
  MVO_HIP(hipMalloc(&t->d_cloud, B * map_capacity * 3 * sizeof(float)));
  MVO_HIP(hipMalloc(&t->d_n_cloud, B * sizeof(int)));
  MVO_HIP(hipMalloc(&t->d_path, B * path_capacity * 7 * sizeof(double)));
  MVO_HIP(hipMalloc(&t->d_n_path, B * sizeof(int)));
  MVO_HIP(hipMalloc(&t->d_ros, B * sizeof(mvo_ros_pose)));
  MVO_HIP(hipMemsetAsync(t->d_n_cloud, 0, B * sizeof(int), ctx->stream));
  MVO_HIP(hipMemsetAsync(t->d_n_path, 0, B * sizeof(int), ctx->stream));
  MVO_HIP(hipMemsetAsync(t->d_ros, 0, B * sizeof(mvo_ros_pose), ctx->stream));
  t->map_cap = map_capacity; t->path_cap = path_capacity;
  t->out_on = true;
  return MVO_OK;
}

// called by mvo_batch_set_landmarks: the slot's seed landmarks (already resident at d_lm) open its map
int trk_output_seed(mvo_ctx* ctx, int slot, const float* d_lm, int n) {
  if (!ctx->pipe || !ctx->pipe->trk || !ctx->pipe->trk->out_on) return MVO_OK;
  TrackState* t = ctx->pipe->trk;
  hipLaunchKernelGGL(trk_output_seed_kernel, dim3((n + 255) / 256 + 1), dim3(256), 0, ctx->stream, t->d_ros, t->d_n_path, slot, d_lm, n, t->d_cloud,
                     t->d_n_cloud, t->map_cap, t->d_err);
  return MVO_OK;
}

static TrackState* trk_output_state(mvo_ctx* ctx) {
  if (!ctx || !ctx->pipe || !ctx->pipe->trk || !ctx->pipe->trk->out_on) {
    if (ctx) ctx->set_error("output side not enabled (mvo_batch_enable_output)");
    return nullptr;
  }
  return ctx->pipe->trk;
}

extern "C" int mvo_batch_get_odometry(mvo_ctx* ctx, mvo_ros_pose* out) {
  TrackState* t = trk_output_state(ctx);
  if (!t || !out) return MVO_E_ARG;
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  MVO_HIP(hipMemcpy(out, t->d_ros, (size_t)ctx->B * sizeof(mvo_ros_pose), hipMemcpyDeviceToHost));
  return MVO_OK;
}

extern "C" int mvo_batch_get_path(mvo_ctx* ctx, int slot, double* poses, int cap, int* n) {
  TrackState* t = trk_output_state(ctx);
  if (!t || slot < 0 || slot >= ctx->B || !n || cap < 0 || (cap && !poses)) return MVO_E_ARG;
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  int cnt = 0;
  MVO_HIP(hipMemcpy(&cnt, t->d_n_path + slot, sizeof(int), hipMemcpyDeviceToHost));
  *n = cnt;
  if (cnt > cap) { ctx->set_error("mvo_batch_get_path: buffer too small"); return MVO_E_CAPACITY; }
  if (cnt) MVO_HIP(hipMemcpy(poses, t->d_path + (size_t)slot * t->path_cap * 7, (size_t)cnt * 7 * sizeof(double), hipMemcpyDeviceToHost));
  return MVO_OK;
}

extern "C" int mvo_batch_get_pointcloud(mvo_ctx* ctx, int slot, float* data, int cap, int* n) {
  TrackState* t = trk_output_state(ctx);
  if (!t || slot < 0 || slot >= ctx->B || !n || cap < 0 || (cap && !data)) return MVO_E_ARG;
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  int cnt = 0;
  MVO_HIP(hipMemcpy(&cnt, t->d_n_cloud + slot, sizeof(int), hipMemcpyDeviceToHost));
  *n = cnt;
  if (cnt > cap) { ctx->set_error("mvo_batch_get_pointcloud: buffer too small"); return MVO_E_CAPACITY; }
  if (cnt) MVO_HIP(hipMemcpy(data, t->d_cloud + (size_t)slot * t->map_cap * 3, (size_t)cnt * 3 * sizeof(float), hipMemcpyDeviceToHost));
  return MVO_OK;
}

// Tracker state of every slot: MVO_TRACK_* (and tracking_count_from_keyframe_).  Blocks until queued work has finished.
extern "C" int mvo_batch_get_state(mvo_ctx* ctx, int* state, int* tracking_count) {
  if (!ctx || !ctx->pipe) return MVO_E_ARG;
  int rc = trk_create(ctx);
  if (rc) return rc;
  TrackState* t = ctx->pipe->trk;
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  if (state) MVO_HIP(hipMemcpy(state, t->d_state, ctx->B * sizeof(int), hipMemcpyDeviceToHost));
  if (tracking_count) MVO_HIP(hipMemcpy(tracking_count, t->d_count, ctx->B * sizeof(int), hipMemcpyDeviceToHost));
  return MVO_OK;
}

// A step / seed on ctx->stream that reads ring entry `frame_idx` waits for an upload still in flight into it.
int trk_wait_upload(mvo_ctx* ctx, int frame_idx) {
  PipeState* p = ctx->pipe;
  if (!p || p->up_pending.empty() || frame_idx < 0 || frame_idx >= (int)p->up_pending.size()) return MVO_OK;
  if (p->up_pending[frame_idx]) { MVO_HIP(hipStreamWaitEvent(ctx->stream, p->ev_up[frame_idx], 0)); p->up_pending[frame_idx] = 0; }
  return MVO_OK;
}

int trk_sync_upload(mvo_ctx* ctx) {
  PipeState* p = ctx->pipe;
  if (p && p->s_up) MVO_HIP(hipStreamSynchronize(p->s_up));
  return MVO_OK;
}

// Called by mvo_batch_seed: a step still in flight finishes first (its results are dropped); every slot starts TRACKING with
// tracking_count_from_keyframe_ = 0, no capacity flags, and an empty path (the cloud restarts with mvo_batch_set_landmarks).
int trk_reset(mvo_ctx* ctx) {
  if (!ctx->pipe || !ctx->pipe->trk) return MVO_OK;
  TrackState* t = ctx->pipe->trk;
  while (t->npending) { MVO_HIP(hipEventSynchronize(t->ev_done[t->head])); t->head = (t->head + 1) % TRK_DEPTH; t->npending--; }
  t->res_valid = false;
  MVO_HIP(hipMemsetAsync(t->d_state, 0, ctx->B * sizeof(int), ctx->stream));
  MVO_HIP(hipMemsetAsync(t->d_count, 0, ctx->B * sizeof(int), ctx->stream));
  MVO_HIP(hipMemsetAsync(t->d_err, 0, sizeof(int), ctx->stream));
  if (t->out_on) {
    MVO_HIP(hipMemsetAsync(t->d_n_cloud, 0, ctx->B * sizeof(int), ctx->stream));
    MVO_HIP(hipMemsetAsync(t->d_n_path, 0, ctx->B * sizeof(int), ctx->stream));
    MVO_HIP(hipMemsetAsync(t->d_ros, 0, ctx->B * sizeof(mvo_ros_pose), ctx->stream));
  }
  return MVO_OK;
}
