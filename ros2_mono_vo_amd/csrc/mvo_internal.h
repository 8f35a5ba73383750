// csrc/mvo_internal.h — context layout and helpers shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mvo.h"

typedef unsigned char u8;

#define MVO_HIP(call)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      ctx->set_error(std::string(#call) + ": " + hipGetErrorString(e_));                     \
      return MVO_E_HIP;                                                                      \
    }                                                                                        \
  } while (0)

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

// A batch of same-sized mono8 images, one per slot.
struct ImgSet {
  u8* base = nullptr;
  int w = 0, h = 0, pitch = 0;
  size_t slot_stride = 0;  // bytes between slots
  __host__ __device__ inline u8* slot(int s) const { return base + (size_t)s * slot_stride; }
};

#define MVO_LK_MAX_LEVELS 4
#define MVO_LK_PAD 32          // border rows (top, bottom) and columns (left) of a pyramid level plane >= 1
#define MVO_LK_PADR 48         // border columns on the right of a level plane: 32 + the over-read of the 16-byte tile loads
#define MVO_ORB_LEVELS 8

struct mvo_ctx {
  mvo_config cfg;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  void set_error(const std::string& s) { err = s; }

  int B = 1;       // slots
  int device = 0;  // HIP device ordinal (resolved at create)
  int refine_waves = 0;  // PnP refine block: 0 = by batch size, 1 / 4 wavefronts forced (MVO_PNP_REFINE_WAVES, tests)
  int maxw = 0, maxh = 0, maxpts = 0;

  // ---- LK: two pyramid sets (ping-pong "prev"/"cur") --------------------------------------------
  u8* lk_mem[2] = {nullptr, nullptr};   // levels 1.. of every slot, packed per slot; every level plane carries a reflect-101 border
                                        // (MVO_LK_PAD rows / columns, MVO_LK_PADR columns on the right), see lk_levels()
  u8* lk_l0[2] = {nullptr, nullptr};    // level 0 of slot 0 only: the per-call API (mvo_lk_track, mvo_pyrdown).  The frame-batch
                                        // tracker reads level 0 in place from the frame ring and keeps no copy of it.
  u8* lk_c0[2] = {nullptr, nullptr};    // level 0 of a TRUE-COLOUR pair of the per-call API: three channel planes lk_c0_plane bytes apart
                                        // (the channels are "slots" 0..2 of the pyramid set: lk_mem holds max(B, 3) slots)
  size_t lk_c0_plane = 0;
  size_t lk_slot_bytes = 0;
  size_t lk_level_off[MVO_LK_MAX_LEVELS] = {0, 0, 0, 0};   // [0] unused
  int lk_cur = 0;  // index of the "cur" set
  float* d_prev_pts = nullptr;  // [B][maxpts][2]
  float* d_next_pts = nullptr;
  u8* d_status = nullptr;       // [B][maxpts]
  float* d_err = nullptr;       // [B][maxpts]
  int* d_npts = nullptr;        // [B]

  // ---- staging -------------------------------------------------------------------------------------
  int* d_colorflag = nullptr;  // set by colour uploads whose channels differ (LK is built for replicated mono8 only)
  u8* d_stage = nullptr;  // raw upload staging (BGR or strided input), maxw*maxh*3 per slot
  size_t stage_slot_bytes = 0;
  u8* h_pin = nullptr;    // pinned host scratch
  size_t h_pin_bytes = 0;

  // ---- ORB ------------------------------------------------------------------------------------------
  struct OrbState* orb = nullptr;
  // ---- matcher ----------------------------------------------------------------------------------
  struct MatchState* match = nullptr;
  // ---- geometry / RANSAC ------------------------------------------------------------------------
  struct GeomState* geom = nullptr;
  // ---- frame-batch pipeline ---------------------------------------------------------------------
  struct PipeState* pipe = nullptr;
  // ---- stage timers -----------------------------------------------------------------------------
  struct Prof* prof = nullptr;
};

// Stage timers (HIP events on ctx->stream).
void prof_begin(mvo_ctx* ctx, const char* name, hipStream_t st = nullptr);
void prof_end(mvo_ctx* ctx);
struct ProfScope {
  mvo_ctx* c;
  ProfScope(mvo_ctx* ctx, const char* name, hipStream_t st = nullptr) : c(ctx) { prof_begin(c, name, st); }
  ~ProfScope() { prof_end(c); }
};

struct OrbGeom {
  int nlevels;
  int w[MVO_ORB_LEVELS], h[MVO_ORB_LEVELS], pitch[MVO_ORB_LEVELS];
  size_t off[MVO_ORB_LEVELS];  // byte offset of level l inside a slot
  size_t slot_stride;
  float scale[MVO_ORB_LEVELS];
  int quota[MVO_ORB_LEVELS];
  int row0[MVO_ORB_LEVELS + 1];  // first NMS-row index of level l (rows inside the edge band only)
  int edge;
};

struct OrbState {
  // capacity geometry (max_width x max_height)
  size_t slot_bytes = 0;
  u8* d_pyr = nullptr;    // un-blurred pyramid [B][slot_bytes]
  u8* d_score = nullptr;  // FAST score maps, same layout
  u8* d_blur = nullptr;   // blurred pyramid, same layout
  int max_rows = 0;
  int* d_row_cnt = nullptr;   // [B][max_rows]
  u8* d_seg_cnt = nullptr;    // [B][max_rows][seg_per_row] NMS survivors per 64-pixel row segment
  int seg_per_row = 0;
  int* d_row_off = nullptr;   // [B][max_rows]  offset inside the slot's candidate range
  int* d_lvl_cnt = nullptr;   // [B][8]
  int* d_slot_tot = nullptr;  // [B]
  int* d_slot_base = nullptr; // [B+1]
  int cand_cap = 0;           // dense candidate capacity for the whole batch
  unsigned short* d_cx = nullptr;
  unsigned short* d_cy = nullptr;
  u8* d_cs = nullptr;
  u8* d_cl = nullptr;      // level of each candidate
  int* d_cslot = nullptr;  // slot of each candidate
  float* d_ch = nullptr;   // harris
  // INTER_LINEAR_EXACT coordinate tables of the current frame geometry (offset | c1 << 16 per destination column / row)
  unsigned* d_rtab = nullptr;
  size_t rtab_cap = 0;
  int rtab_x[MVO_ORB_LEVELS] = {0}, rtab_y[MVO_ORB_LEVELS] = {0};
  int rtab_w = 0, rtab_h = 0;
  // retainBest on the device (orb_select.hip): (response, index) pairs + two staging buffers, all cand_cap wide
  uint2* d_wk = nullptr;
  uint2* d_stl = nullptr;
  uint2* d_str = nullptr;
  int* d_kept = nullptr;     // [B][8] survivors per (slot, level)
  int* d_kp_base = nullptr;  // [B+1] first selected key-point of each slot in the dense selection
  // final key-points
  int kp_cap = 0;  // dense, whole batch
  int* d_sel = nullptr;  // selected candidate indices
  mvo_keypoint* d_kp = nullptr;
  void* d_brec = nullptr;  // [kp_cap] rBRIEF records (orb.hip BriefRec), written by the IC-angle kernel
  u8* d_desc = nullptr;
  char4* d_pattern = nullptr;
  unsigned* d_icmask = nullptr;  // [16][9] byte masks of the IC-angle disc rows
  // pinned host mirrors
  int* h_counts = nullptr;  // [B][8] level counts + [B+1] candidate bases + [B+1] key-point bases
  mvo_keypoint* h_kp = nullptr;
  u8* h_desc = nullptr;
  hipEvent_t ev_counts = nullptr, ev_cand = nullptr;  // phase boundaries of the split detect (orb_detect_enqueue / orb_select)
};


struct MatchState {
  u8* d_q = nullptr;     // [B][cap][32]
  u8* d_t = nullptr;
  unsigned* d_best = nullptr;  // [B][cap][2] packed keys
  mvo_match* d_out = nullptr;  // [B][cap]
  int* d_nout = nullptr;       // [B]
  int* d_nq = nullptr;         // [B]
  int* d_nt = nullptr;         // [B]
  int cap = 0;
};


struct GeomState {
  float* d_m1 = nullptr;   // [B][maxpts][3]
  float* d_m2 = nullptr;   // [B][maxpts][2]
  int* d_n = nullptr;      // [B]
  u8* d_mask = nullptr;    // [B][maxpts]
  double* d_model = nullptr;  // [B][16]
  int* d_result = nullptr;    // [B][8]
  // PnP refine
  int* d_inl = nullptr;       // [B][maxpts] inlier indices
  double* d_pose = nullptr;   // [B][8] rvec, tvec
  float* d_x3 = nullptr;      // [B][maxpts][3]
  // second set for running H and F side by side in the pipeline
  u8* d_mask2 = nullptr;
  double* d_model2 = nullptr;
  int* d_result2 = nullptr;
  double* d_tmp = nullptr;    // 64 doubles of scratch
  double* h_model = nullptr;  // pinned
  int* h_result = nullptr;
};


// ---- frame-batch pipeline state (pipeline.hip, track.hip) --------------------------------------------
struct PipeState {
  int ring = 0;
  int w = 0, h = 0, pitch = 0;  // geometry of the frames in the ring (fixed by the first preload)
  size_t frame_bytes = 0;       // one slot's frame
  u8* d_ring = nullptr;         // [ring][B][h][pitch]
  float* d_lm = nullptr;        // [B][maxpts][3] landmark of each tracked point
  float* d_kf_pts = nullptr;    // [B][maxpts][2] last key-frame position of each tracked point
  float* d_cur_pts = nullptr;   // compacted survivors of LK
  float* d_cur_lm = nullptr;
  float* d_cur_kf = nullptr;
  int* d_ncur = nullptr;        // [B]
  float* d_kp_xy = nullptr;     // [B][maxpts][2] key-point positions of the current frame (match train side)
  float* d_kfkp_xy = nullptr;   // [B][maxpts][2] key-point positions of the last key-frame (match query side)
  int* h_ints = nullptr;        // pinned scratch
  double K[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double dist[5] = {0, 0, 0, 0, 0};
  bool seeded = false;
  // landmark hand-over (device-resident Frame/KeyFrame bookkeeping, src/tracker.cpp:193-234)
  u8* d_kf_has = nullptr;     // [B][maxpts] key-frame observation has a landmark
  float* d_kf_lm = nullptr;   // [B][maxpts][3]
  u8* d_cur_has = nullptr;
  float* d_cur_lmk = nullptr; // [B][maxpts][3]
  int* d_winner = nullptr;    // [B][maxpts] last valid match per current key-point
  float* d_tri = nullptr;     // [B][maxpts][3] triangulated matches
  u8* d_tri_ok = nullptr;     // [B][maxpts]
  double* d_kf_pose = nullptr;  // [B][8] T_cw of the last key-frame (rvec, tvec)
  int* d_ntri = nullptr;      // [B]
  int trk_max_n = 0;  // per-slot track count bound of the seed (informational)
  int kf_max_n = 0;   // key-frame descriptor count bound of the seed
  struct TrackState* trk = nullptr;  // device-driven per-stream tracker (track.hip)
  // asynchronous ingest (track.hip): uploads run on their own stream; ev_up[f] = frame f of the ring has landed,
  // ev_rd[f] = the step that consumed frame f has read it (a later upload into that ring entry waits for it)
  hipStream_t s_up = nullptr;
  int step_entry = 0;       // ring entry mvo_tracker_step used last (the seed frame is entry 0)
  int prev_entry = -1;      // ring entry of the frame seeded / tracked last: it IS level 0 of the "prev" LK pyramid (read in place)
  bool up_shared = false;   // uploads ride on the compute stream (MVO_UPLOAD_STREAM=0)
  std::vector<hipEvent_t> ev_up, ev_rd;
  std::vector<char> up_pending, rd_pending;
};

// Level geometry helpers (host).
struct LkLevels {
  int n;  // number of levels actually used (maxLevel+1 after the <= winSize early stop)
  int w[MVO_LK_MAX_LEVELS], h[MVO_LK_MAX_LEVELS], pitch[MVO_LK_MAX_LEVELS];
};
LkLevels lk_levels(int w, int h, int win, int max_level);

// Implemented per translation unit; called from mvo_create / mvo_destroy.
int orb_state_create(mvo_ctx* ctx);
void orb_state_destroy(mvo_ctx* ctx);
int match_state_create(mvo_ctx* ctx);
void match_state_destroy(mvo_ctx* ctx);
int geom_state_create(mvo_ctx* ctx);
void geom_state_destroy(mvo_ctx* ctx);
int pipe_state_create(mvo_ctx* ctx);
void pipe_state_destroy(mvo_ctx* ctx);
void trk_destroy(mvo_ctx* ctx);   // track.hip
int trk_reset(mvo_ctx* ctx);
int trk_ring_events(mvo_ctx* ctx);                  // upload stream + ring events (idempotent)
int trk_wait_upload(mvo_ctx* ctx, int frame_idx);   // ctx->stream waits for an asynchronous upload into ring entry frame_idx
int trk_output_seed(mvo_ctx* ctx, int slot, const float* d_lm, int n);   // output side: the seed landmarks open the slot's map
int trk_sync_upload(mvo_ctx* ctx);                  // host waits for the upload stream
struct TrkLostPolicy;
struct TrkKeyframePolicy;
void lk_filter_compact_launch(mvo_ctx* ctx, hipStream_t st, const TrkLostPolicy* lost = nullptr);   // pipeline.hip: status/err filter of all slots (+ the LOST policy)

// device-level stage drivers (all slots per launch)
int lk_build_pyramid(mvo_ctx* ctx, int set, const LkLevels& L, int nslots, hipStream_t st = nullptr, const u8* l0 = nullptr,
                     size_t l0_stride = 0);
int lk_track_colour_device(mvo_ctx* ctx, int prev_set, int cur_set, const LkLevels& L, int n);   // true-colour pair in lk_c0 / slots 0..2
int lk_track_device(mvo_ctx* ctx, int prev_set, int cur_set, const LkLevels& L, int nslots, int max_n, hipStream_t st = nullptr,
                    const int* d_work_slot = nullptr, const int* d_pt_base = nullptr, int* d_work_ctr = nullptr,
                    const u8* prev_l0 = nullptr, const u8* cur_l0 = nullptr, size_t l0_stride = 0, int items_bound = -1);
// ORB in three phases on ctx->stream so that a caller can put other GPU work beside the host-side selection:
//   orb_detect_enqueue  pyramid, FAST+NMS, ordered compaction, async copy of the counts           (no host wait)
//   orb_select          waits for the counts (capacity check, grid size), Harris, OpenCV's two retainBest passes per
//                       level on the device, then the blurred pyramid; returns once the per-slot key-point counts are
//                       on the host (the blur is still running)
//   orb_describe_enqueue  IC angle + rBRIEF for the selection; `to_host` also copies key-points/descriptors back and waits
// orb_run is the three in a row.
int orb_detect_enqueue(mvo_ctx* ctx, int w, int h, int nslots, hipEvent_t before_fast = nullptr);
int orb_select(mvo_ctx* ctx, int w, int h, int nslots, bool describe, std::vector<int>& kp_base);
int orb_describe_enqueue(mvo_ctx* ctx, int w, int h, int nslots, bool describe, bool to_host, const std::vector<int>& kp_base);
int orb_run(mvo_ctx* ctx, int w, int h, int nslots, bool describe, std::vector<int>& kp_base);
int orb_select_device(mvo_ctx* ctx, const OrbGeom& G, int nslots, const int* d_nact = nullptr);  // orb_select.hip: both retainBest passes + dense gather
int orb_run_device(mvo_ctx* ctx, int w, int h, int max_slots, const int* d_nact);  // device-driven detect + describe (no host wait)
// d_list / d_nlist (optional): run only the *d_nlist slots named by the device-resident list (grid sized for nslots)
int match_device(mvo_ctx* ctx, int nslots, int max_nq, double ratio, const int* d_list = nullptr, const int* d_nlist = nullptr);
// work_ctr / grid: see RansacArgs::work_ctr (persistent workgroups over a slot queue); null = one workgroup per slot
int geom_ransac_h(mvo_ctx* ctx, int nslots, const float* p1, const float* p2, const int* d_n, double thr, int max_iters, double conf,
                  u8* mask, double* model, int* result, hipStream_t st, int* work_ctr = nullptr, int grid = 0);
int geom_ransac_f(mvo_ctx* ctx, int nslots, const float* p1, const float* p2, const int* d_n, double thr, int max_iters, double conf,
                  u8* mask, double* model, int* result, hipStream_t st, int* work_ctr = nullptr, int grid = 0);
int geom_pnp(mvo_ctx* ctx, int nslots, const float* obj, const float* img, const int* d_n, const double K[9], const double* dist5, int iters,
             float reproj, double conf, u8* mask, double* model, int* result, int* inl, double* pose, hipStream_t st,
             const TrkKeyframePolicy* kp = nullptr);   // kp: the key-frame policy of the tracker step at the end of each slot's refine
int geom_triangulate_matches(mvo_ctx* ctx, int nslots, int max_matches, const mvo_match* matches, const int* n_matches,
                             const float* kf_xy, const float* cur_xy, const double* kf_pose, const double* cur_pose,
                             const int* pnp_result, const double K[9], float* X3, u8* valid, const int* d_list = nullptr,
                             const int* d_nlist = nullptr);

// Upload a host image (mono8 or BGR8, arbitrary stride) into a device mono8 ImgSet slot (async on ctx->stream).
int upload_gray(mvo_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels, u8* d_dst,
                int dpitch, int slot, bool check_identical = false);
void planes_from_stage(mvo_ctx* ctx, int w, int h, int channels, u8* d_dst, int dpitch, size_t plane);   // channel planes of the image upload_gray staged last (slot 0)
int color_channels_differ(mvo_ctx* ctx, int* differ);   // see color2gray_kernel

// ---- device helpers -------------------------------------------------------------------------------
__device__ __forceinline__ int d_reflect101(int p, int len) {
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    if (p < 0) p = -p;
    else p = 2 * (len - 1) - p;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}
// XCD-aware tile order for the image kernels.  Workgroups are handed to the 8 XCDs round-robin by linear id, and
// each XCD has its own L2, so with a plain (x, y, slot) grid the tiles that share 128-byte lines and halo rows land
// on different L2s and every line is fetched from HBM several times (measured 4x the algorithmic bytes).  Here the
// launch is 1-D and workgroup b works on tile (b % 8) * ceil(n / 8) + b / 8: each XCD walks one contiguous run of
// tiles (whole images), so neighbours hit in its L2.  Placement is a speed hint only; any mapping is correct.
struct TileGrid { int gx, gy, gz; };
static inline unsigned xcd_grid_blocks(const TileGrid& g) {
  unsigned n = (unsigned)g.gx * g.gy * g.gz;
  return ((n + 7) / 8) * 8;
}
// Launch width of the device-driven (count read on the device) image kernels: workgroups loop over the tile list with a
// stride of the grid, so a launch never needs the host to know how many slots are active.  8 workgroups per CU and a
// multiple of 8 (b % 8 = XCD stays fixed over a workgroup's iterations).
#define MVO_PERSIST_BLOCKS 2048u
static inline unsigned persist_grid(unsigned want) { return want < MVO_PERSIST_BLOCKS ? want : MVO_PERSIST_BLOCKS; }
// tile of linear workgroup index b (b = blockIdx.x in a one-tile-per-workgroup launch, the loop variable in a strided one)
__device__ __forceinline__ bool xcd_tile_b(const TileGrid& g, unsigned b, int& tx, int& ty, int& tz) {
  const unsigned n = (unsigned)g.gx * g.gy * g.gz, per = (n + 7) / 8;
  const unsigned t = (b & 7u) * per + (b >> 3);
  if (t >= n) return false;
  const unsigned row = t / g.gx;
  tx = (int)(t - row * g.gx);
  tz = (int)(row / g.gy);
  ty = (int)(row - (unsigned)tz * g.gy);
  return true;
}
__device__ __forceinline__ bool xcd_tile(const TileGrid& g, int& tx, int& ty, int& tz) { return xcd_tile_b(g, blockIdx.x, tx, ty, tz); }
__device__ __forceinline__ unsigned xcd_grid_blocks_dev(const TileGrid& g) {
  const unsigned n = (unsigned)g.gx * g.gy * g.gz;
  return ((n + 7) / 8) * 8;
}
__device__ __forceinline__ int d_cv_round(float v) { return __float2int_rn(v); }
__device__ __forceinline__ int d_cv_round(double v) { return __double2int_rn(v); }
__device__ __forceinline__ int d_cv_floor(float v) { return (int)floorf(v); }
__device__ __forceinline__ int d_cv_floor(double v) { return (int)floor(v); }
