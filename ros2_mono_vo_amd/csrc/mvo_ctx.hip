// csrc/mvo_ctx.hip — context lifetime, configuration defaults, image upload.
#include "mvo_internal.h"
#include <cstdlib>

#include <cmath>

extern "C" const char* mvo_version(void) { return "mvo-hip 0.1 (gfx950)"; }

extern "C" void mvo_config_default(mvo_config* c) {
  memset(c, 0, sizeof(*c));
  c->max_width = 1280;
  c->max_height = 720;
  c->batch = 1;
  c->max_points = 8192;
  c->nfeatures = 1000;  // reference src/mono_vo.cpp:16
  c->fast_threshold = 20;
  c->orb_blur_mode = 0;
  c->lk_channels = 3;  // reference src/mono_vo.cpp:94 (everything becomes BGR8)
  c->lk_win = 21;
  c->lk_max_level = 3;
  c->lk_max_count = 30;
  c->lk_epsilon = 0.01;
  c->lk_min_eig = 1e-4;
  // reference include/mono_vo/tracker.hpp:137-147
  c->tracking_error_thresh = 30.0f;
  c->min_observations_before_triangulation = 100;
  c->min_tracked_points = 10;
  c->max_tracking_after_keyframe = 10;
  c->max_rotation_from_keyframe = M_PI * 15.0 / 180.0;
  c->max_translation_from_keyframe = 1.0;
  c->ransac_reproj_thresh = 1.0;
  c->model_score_thresh = 0.85;
  c->f_inlier_thresh = 0.5;
  c->lowes_distance_ratio = 0.7;
  // reference include/mono_vo/initializer.hpp:109-115
  c->occupancy_grid_div = 50;
  c->kp_distribution_thresh = 0.5;
  c->min_matches_for_init = 100;
  c->init_model_score_thresh = 0.56;
  c->hip_stream = nullptr;
  c->orb_pattern = nullptr;
  c->device = -1;
  c->ring_frames = 0;
}

LkLevels lk_levels(int w, int h, int win, int max_level) {
  LkLevels L;
  L.n = 1;
  L.w[0] = w; L.h[0] = h; L.pitch[0] = align_up(w, 64);
  for (int l = 1; l <= max_level && l < MVO_LK_MAX_LEVELS; l++) {
    int sw = (L.w[l - 1] + 1) / 2, sh = (L.h[l - 1] + 1) / 2;
    if (sw <= win || sh <= win) break;  // buildOpticalFlowPyramid early stop
    // levels >= 1 live in planes with a border: pixel (0, 0) sits MVO_LK_PAD rows and columns into the plane
    L.w[l] = sw; L.h[l] = sh; L.pitch[l] = align_up(sw + MVO_LK_PAD + MVO_LK_PADR, 64);
    L.n = l + 1;
  }
  return L;
}

// colour -> gray as the reference's ingest does it: cv_bridge::toCvShare(msg, BGR8) (src/mono_vo.cpp:94; rgb8 is a
// channel swap, bgra8 / rgba8 drop alpha) followed by OpenCV's cvtColor(BGR2GRAY) inside ORB: RGB2Gray<uchar> with the
// 15-bit weights BY15 = 3735, GY15 = 19235, RY15 = 9798.  `bpp` 3 or 4, `rgb` = red comes first.
// `differ` (optional): set to 1 when any pixel's three channels are not identical.  LK callers need it: the reference runs
// calcOpticalFlowPyrLK on the BGR8 image (src/mono_vo.cpp:94 -> src/tracker.cpp:68), i.e. its sums run over three channels; the
// device tracks one plane and scales the sums by lk_channels, which equals the reference only for replicated mono8.
__global__ void color2gray_kernel(const u8* __restrict__ src, int spitch, u8* __restrict__ dst, int dpitch, int w, int h, int bpp,
                                  int rgb, int* __restrict__ differ) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y;
  if (x >= w) return;
  const u8* p = src + (size_t)y * spitch + bpp * x;
  const int b = rgb ? p[2] : p[0], g = p[1], r = rgb ? p[0] : p[2];
  dst[(size_t)y * dpitch + x] = (u8)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
  if (differ && (b != g || g != r)) *differ = 1;   // benign race: every writer stores the same value
}

// the first three channels of the staged interleaved image as three planes (true-colour LK: the channel order does not matter
// to the sums, alpha is dropped)
__global__ void color2planes_kernel(const u8* __restrict__ src, int spitch, u8* __restrict__ dst, int dpitch, size_t plane, int w, int h,
                                    int bpp) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y;
  if (x >= w) return;
  const u8* p = src + (size_t)y * spitch + bpp * x;
  u8* d = dst + (size_t)y * dpitch + x;
  d[0] = p[0]; d[plane] = p[1]; d[2 * plane] = p[2];
}

// after upload_gray(.., slot 0) of a colour image: its channel planes from the staging buffer
void planes_from_stage(mvo_ctx* ctx, int w, int h, int channels, u8* d_dst, int dpitch, size_t plane) {
  const int bpp = channels < 0 ? -channels : channels;
  dim3 grid((w + 255) / 256, h);
  hipLaunchKernelGGL(color2planes_kernel, grid, dim3(256), 0, ctx->stream, ctx->d_stage, align_up(w * bpp, 64), d_dst, dpitch, plane, w, h, bpp);
}

int upload_gray(mvo_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels, u8* d_dst,
                int dpitch, int slot, bool check_identical) {
  if (channels == 1) {
    MVO_HIP(hipMemcpy2DAsync(d_dst, dpitch, img, stride, w, h, hipMemcpyDefault, ctx->stream));
  } else if (channels == 3 || channels == -3 || channels == 4 || channels == -4) {
    const int bpp = channels < 0 ? -channels : channels;
    u8* st = ctx->d_stage + (size_t)slot * ctx->stage_slot_bytes;
    int spitch = align_up(w * bpp, 64);
    MVO_HIP(hipMemcpy2DAsync(st, spitch, img, stride, (size_t)w * bpp, h, hipMemcpyDefault, ctx->stream));
    dim3 grid((w + 255) / 256, h);
    hipLaunchKernelGGL(color2gray_kernel, grid, dim3(256), 0, ctx->stream, st, spitch, d_dst, dpitch, w, h, bpp, channels < 0 ? 1 : 0,
                       check_identical ? ctx->d_colorflag : nullptr);
  } else {
    ctx->set_error("channels must be 1 (mono8), 3 (BGR8), -3 (RGB8), 4 (BGRA8) or -4 (RGBA8)");
    return MVO_E_ARG;
  }
  return MVO_OK;
}

static int mvo_create_impl(const mvo_config* cfg, mvo_ctx** out) {
  if (!cfg || !out) return MVO_E_ARG;
  if (cfg->max_width < 32 || cfg->max_height < 32 || cfg->batch < 1 || cfg->max_points < 16) return MVO_E_ARG;
  if (cfg->ring_frames > 0 && (cfg->max_points > 65535 || cfg->batch > 32767)) return MVO_E_ARG;   // the LK work list packs slot << 16 | point
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return MVO_E_HIP;  // fail loudly: no CPU fallback
  mvo_ctx* ctx = new mvo_ctx();
  ctx->cfg = *cfg;
  ctx->B = cfg->batch;
  ctx->maxw = cfg->max_width;
  ctx->maxh = cfg->max_height;
  ctx->maxpts = cfg->max_points;
  if (const char* e = getenv("MVO_PNP_REFINE_WAVES")) ctx->refine_waves = atoi(e) == 4 ? 4 : (atoi(e) == 1 ? 1 : 0);
  *out = ctx;
  if (cfg->device >= 0) MVO_HIP(hipSetDevice(cfg->device));
  MVO_HIP(hipGetDevice(&ctx->device));   // the ordinal the context lives on (cfg.device = -1: the caller's current device)
  if (cfg->hip_stream) {
    ctx->stream = (hipStream_t)cfg->hip_stream;
  } else {
    MVO_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
  }
  // LK pyramids
  LkLevels L = lk_levels(ctx->maxw, ctx->maxh, 1, MVO_LK_MAX_LEVELS - 1);  // capacity: all 4 levels
  size_t off = 0;
  for (int l = 1; l < MVO_LK_MAX_LEVELS; l++) {
    ctx->lk_level_off[l] = off;
    if (l < L.n) off += (size_t)L.pitch[l] * (L.h[l] + 2 * MVO_LK_PAD);
    off = (off + 255) & ~(size_t)255;
  }
  ctx->lk_slot_bytes = off + 256;
  for (int i = 0; i < 2; i++) {
    MVO_HIP(hipMalloc(&ctx->lk_mem[i], ctx->lk_slot_bytes * (ctx->B > 3 ? ctx->B : 3)));
    MVO_HIP(hipMalloc(&ctx->lk_l0[i], (size_t)L.pitch[0] * L.h[0] + 256));
    ctx->lk_c0_plane = ((size_t)L.pitch[0] * L.h[0] + 255) & ~(size_t)255;
    MVO_HIP(hipMalloc(&ctx->lk_c0[i], 3 * ctx->lk_c0_plane + 256));
  }
  size_t np = (size_t)ctx->B * ctx->maxpts;
  MVO_HIP(hipMalloc(&ctx->d_prev_pts, np * 2 * sizeof(float)));
  MVO_HIP(hipMalloc(&ctx->d_next_pts, np * 2 * sizeof(float)));
  MVO_HIP(hipMalloc(&ctx->d_status, np));
  MVO_HIP(hipMalloc(&ctx->d_err, np * sizeof(float)));
  MVO_HIP(hipMalloc(&ctx->d_npts, ctx->B * sizeof(int)));
  ctx->stage_slot_bytes = (size_t)align_up(ctx->maxw * 4, 64) * ctx->maxh;
  MVO_HIP(hipMalloc(&ctx->d_stage, ctx->stage_slot_bytes * ctx->B));
  MVO_HIP(hipMalloc(&ctx->d_colorflag, sizeof(int)));
  MVO_HIP(hipMemset(ctx->d_colorflag, 0, sizeof(int)));
  ctx->h_pin_bytes = (size_t)16 << 20;
  MVO_HIP(hipHostMalloc(&ctx->h_pin, ctx->h_pin_bytes, hipHostMallocDefault));
  int rc;
  if ((rc = orb_state_create(ctx)) != MVO_OK) return rc;
  if ((rc = match_state_create(ctx)) != MVO_OK) return rc;
  if ((rc = geom_state_create(ctx)) != MVO_OK) return rc;
  if ((rc = pipe_state_create(ctx)) != MVO_OK) return rc;
  return MVO_OK;
}

// On failure nothing is left behind: the half-built context is destroyed and *out is NULL (the error text of a failed
// create is not retrievable through mvo_last_error; the status code says which class of failure it was).
extern "C" int mvo_create(const mvo_config* cfg, mvo_ctx** out) {
  if (!out) return MVO_E_ARG;
  *out = nullptr;
  mvo_ctx* ctx = nullptr;
  const int rc = mvo_create_impl(cfg, &ctx);
  if (rc != MVO_OK) {
    if (ctx) mvo_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return MVO_OK;
}

extern "C" void mvo_destroy(mvo_ctx* ctx) {
  if (!ctx) return;
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  pipe_state_destroy(ctx);
  geom_state_destroy(ctx);
  match_state_destroy(ctx);
  orb_state_destroy(ctx);
  for (int i = 0; i < 2; i++) { (void)hipFree(ctx->lk_mem[i]); (void)hipFree(ctx->lk_l0[i]); (void)hipFree(ctx->lk_c0[i]); }
  (void)hipFree(ctx->d_prev_pts);
  (void)hipFree(ctx->d_next_pts);
  (void)hipFree(ctx->d_status);
  (void)hipFree(ctx->d_err);
  (void)hipFree(ctx->d_npts);
  (void)hipFree(ctx->d_stage);
  (void)hipFree(ctx->d_colorflag);
  if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

// 1 if a colour upload since the last call had differing channels (waits for the stream); clears the flag
int color_channels_differ(mvo_ctx* ctx, int* differ) {
  int* h = (int*)ctx->h_pin;
  MVO_HIP(hipMemcpyAsync(h, ctx->d_colorflag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipMemsetAsync(ctx->d_colorflag, 0, sizeof(int), ctx->stream));
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  *differ = h[0];
  return MVO_OK;
}

extern "C" const char* mvo_last_error(const mvo_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int mvo_sync(mvo_ctx* ctx) {
  if (!ctx) return MVO_E_ARG;
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  return trk_sync_upload(ctx);   // and the upload stream of the frame-batch ring
}

extern "C" void* mvo_stream(mvo_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
