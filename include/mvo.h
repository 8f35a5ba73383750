/* include/mvo.h — C ABI of libmvo_hip.so: the MI355X-native per-frame VO front-end.
 *
 * Drop-in boundary for the hot path of Tatsuya-2/ros2_mono_vo.  The reference has no FFI layer: its
 * FeatureProcessor / Tracker / Initializer call OpenCV directly.  Each entry point below replaces one
 * of those OpenCV call sites (cited per function as reference file:line) and is what a maintainer
 * would bind from the reference's C++ (see INTEGRATION.md for the shim).  POD only, caller-owned
 * host buffers, int status codes, never throws.  Results follow the OpenCV-4.6 semantics the
 * reference relies on; parity is checked against oracle/ (test-only).
 *
 * Threading: a context is single-threaded (one HIP stream); any number of contexts per process.
 * All calls are synchronous unless stated: outputs are valid in host memory on return.
 */
#ifndef MVO_H_
#define MVO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVO_OK 0
#define MVO_E_ARG 1        /* bad argument / shape (OpenCV would throw cv::Exception) */
#define MVO_E_CAPACITY 2   /* an output or internal buffer is too small */
#define MVO_E_HIP 3        /* HIP runtime error (mvo_last_error has the text) */
#define MVO_E_DEGENERATE 4 /* estimator found no model (OpenCV returns an empty Mat / false) */

typedef struct mvo_ctx mvo_ctx;

/* cv::KeyPoint layout (7 x 4 bytes). */
typedef struct mvo_keypoint {
  float x, y, size, angle, response;
  int octave, class_id;
} mvo_keypoint;

/* cv::DMatch layout. */
typedef struct mvo_match {
  int query_idx, train_idx, img_idx;
  float distance;
} mvo_match;

/* Configuration.  Names and defaults mirror the reference's ROS parameters
 * (include/mono_vo/tracker.hpp:137-147, include/mono_vo/initializer.hpp:109-115) plus the constants
 * the reference hard-codes (nfeatures: src/mono_vo.cpp:16; ORB / LK defaults come from OpenCV). */
typedef struct mvo_config {
  int max_width, max_height; /* largest frame this context will see */
  int batch;                 /* independent camera streams (slots) resident in the context, >= 1 */
  int max_points;            /* capacity per slot for tracked points / correspondences */
  /* ORB (FeatureProcessor ctor, src/feature_processor.cpp:5-10) */
  int nfeatures;      /* 1000 in the reference */
  int fast_threshold; /* 20 */
  int orb_blur_mode;  /* 0: sepFilter2D 8-bit taps (what ORB's in-place sub-matrix blur gets); 1: bit-exact ED taps */
  /* LK (cv::calcOpticalFlowPyrLK defaults, src/tracker.cpp:68-69) */
  int lk_channels; /* 3: the reference converts every frame to BGR8 (src/mono_vo.cpp:94) */
  int lk_win;      /* 21 */
  int lk_max_level; /* 3 */
  int lk_max_count; /* 30 */
  double lk_epsilon; /* 0.01 */
  double lk_min_eig; /* 1e-4 */
  /* Tracker parameters (include/mono_vo/tracker.hpp:137-147) */
  float tracking_error_thresh;                   /* 30.0 */
  int64_t min_observations_before_triangulation; /* 100 */
  int64_t min_tracked_points;                    /* 10 */
  int64_t max_tracking_after_keyframe;           /* 10 */
  double max_rotation_from_keyframe;             /* 15 deg in rad */
  double max_translation_from_keyframe;          /* 1.0 */
  double ransac_reproj_thresh;                   /* 1.0 (H / F) */
  double model_score_thresh;                     /* 0.85 tracker */
  double f_inlier_thresh;                        /* 0.5 */
  double lowes_distance_ratio;                   /* 0.7 */
  /* Initializer parameters (include/mono_vo/initializer.hpp:109-115) */
  int occupancy_grid_div;             /* 50 */
  double kp_distribution_thresh;      /* 0.5 */
  int64_t min_matches_for_init;       /* 100 */
  double init_model_score_thresh;     /* 0.56 */
  /* plumbing */
  void* hip_stream;       /* optional caller-owned hipStream_t; NULL = the context creates one */
  const int* orb_pattern; /* optional 1024 ints replacing the built-in rBRIEF pattern */
  int device;             /* HIP device ordinal, -1 = current */
  int ring_frames;        /* frame-batch mode: frames per slot kept resident in HBM (0 = none) */
} mvo_config;

void mvo_config_default(mvo_config* cfg);
/* On failure *out is NULL and nothing stays allocated. */
int mvo_create(const mvo_config* cfg, mvo_ctx** out);
void mvo_destroy(mvo_ctx* ctx);
const char* mvo_last_error(const mvo_ctx* ctx);
const char* mvo_version(void);
/* Blocks until all work queued on the context's stream has finished. */
int mvo_sync(mvo_ctx* ctx);
/* The hipStream_t the context launches on (for HIP-event timing by the caller). */
void* mvo_stream(mvo_ctx* ctx);

/* ---- a1: FeatureProcessor::detect_and_compute (src/feature_processor.cpp:19-23) ------------------
 * cv::ORB::detectAndCompute(img, noArray(), kps, desc).  `channels` names the sensor_msgs encoding of img everywhere
 * in this header: 1 = mono8, 3 = bgr8, -3 = rgb8, 4 = bgra8, -4 = rgba8 (what cv_bridge::toCvShare(msg, BGR8) at
 * src/mono_vo.cpp:94 accepts); colour is reduced with cvtColor(BGR2GRAY)'s 15-bit weights on the device.
 * Writes min(*n, cap) keypoints and 32-byte descriptors; *n is the full count (may exceed nfeatures
 * on score ties, as in OpenCV).  Key-point order is OpenCV's (KeyPointsFilter::retainBest order). */
int mvo_orb_detect_and_compute(mvo_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels,
                               mvo_keypoint* kps, uint8_t* desc, int cap, int* n);
/* FeatureProcessor::detect (src/feature_processor.cpp:12-17; no callers in the reference). */
int mvo_orb_detect(mvo_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels,
                   mvo_keypoint* kps, int cap, int* n);
/* Building block exposed for parity tests: FAST-9/16 + 3x3 NMS on one mono8 image, row-major
 * (x, y, score) triples, as cv::FAST(img, kps, threshold, true). */
int mvo_fast9_nms(mvo_ctx* ctx, const uint8_t* img, int w, int h, int stride, int threshold, int* xys,
                  int cap, int* n);
/* Building block exposed for parity tests: cv::KeyPointsFilter::retainBest(keypoints, n_keep) on an array of
 * responses (what cv::ORB applies twice per pyramid level inside detectAndCompute, src/feature_processor.cpp:19-23).
 * Writes the ORIGINAL indices of the survivors in the order OpenCV leaves them in the vector (std::nth_element +
 * std::partition of libstdc++ 11); *out_n may exceed n_keep on ties with the n_keep-th response.  out_idx must
 * hold n entries.  depth_limit < 0: introselect's own limit, 2*floor(log2 n); >= 0 overrides it (test hook for
 * the heap-select branch). */
int mvo_retain_best(mvo_ctx* ctx, const float* responses, int n, int n_keep, int depth_limit, int* out_idx, int* out_n);

/* ---- a2: FeatureProcessor::find_matches (src/feature_processor.cpp:25-41) ------------------------
 * BFMatcher(NORM_HAMMING).knnMatch(q, t, 2) + Lowe ratio (m[0].distance < ratio * m[1].distance). */
int mvo_match_knn2_ratio(mvo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio,
                         mvo_match* out, int cap, int* n);

/* ---- a3: Tracker::track_frame_with_optical_flow (src/tracker.cpp:58-90) -------------------------
 * cv::calcOpticalFlowPyrLK(prev, next, prev_pts, next_pts, status, err) with default arguments.
 * `channels`: 1 (mono8), 3 / -3 (BGR8 / RGB8), 4 / -4 (BGRA8 / RGBA8, alpha ignored).  The reference tracks on the BGR8 image
 * (src/mono_vo.cpp:94): a mono8 plane, or a colour image whose three channels are identical everywhere, is tracked as ONE
 * plane with every sum scaled by cfg.lk_channels (exactly what three identical channels give); a colour pair whose channels
 * differ is tracked over its three channel planes (calcOpticalFlowPyrLK on CV_8UC3: window rows of 3 * 21 interleaved elements).
 * The frame-batch tracker (mvo_batch_*) keeps mono8 rings and still refuses true colour. */
int mvo_lk_track(mvo_ctx* ctx, const uint8_t* prev, const uint8_t* next, int w, int h, int stride,
                 int channels, const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err);
/* cv::pyrDown building block (parity tests). dst is ((w+1)/2) x ((h+1)/2). */
int mvo_pyrdown(mvo_ctx* ctx, const uint8_t* src, int w, int h, int stride, uint8_t* dst, int dstride);
/* The image levels of cv::buildOpticalFlowPyramid as calcOpticalFlowPyrLK builds them inside src/tracker.cpp:68 (pyrDown chain
 * to cfg.lk_max_level, stopping early where a level would not exceed winSize), produced by the tracker's own pyramid path
 * (levels 1..3 in one fused launch).  Level l (1-based) is written to levels[l - 1], rows tightly packed ((w_l) bytes per
 * row, w_l = (w_{l-1} + 1) / 2); *n_levels = levels in use including level 0.  Parity tests. */
int mvo_build_lk_pyramid(mvo_ctx* ctx, const uint8_t* img, int w, int h, int stride, uint8_t* const* levels, int* n_levels);

/* ---- a5: Tracker::has_parallax / Initializer::check_parallax (src/tracker.cpp:237-268,
 * src/initializer.cpp:77-110) ----------------------------------------------------------------------
 * cv::findHomography(p1, p2, RANSAC, thr, mask) (max_iters 2000, confidence 0.995) — mask is the
 * RANSAC consensus mask (0/1); H is the consensus model before OpenCV's LM polish (the reference
 * discards H).  Returns MVO_E_DEGENERATE with an all-zero mask when no model is found. */
int mvo_find_homography_ransac(mvo_ctx* ctx, const float* p1, const float* p2, int n, double thr,
                               int max_iters, double confidence, uint8_t* mask, double H[9], int* n_inliers);
/* cv::findFundamentalMat(p1, p2, FM_RANSAC, thr, confidence, mask) (7-point, max_iters 1000).  As in OpenCV: n < 7 ->
 * MVO_E_DEGENERATE (empty result), n == 7 -> the direct 7-point solution, 8 <= n < 15 -> LMedS (300 iterations, thr
 * unused), n >= 15 -> RANSAC. */
int mvo_find_fundamental_ransac(mvo_ctx* ctx, const float* p1, const float* p2, int n, double thr,
                                double confidence, int max_iters, uint8_t* mask, double F[9],
                                int* n_inliers);

/* ---- a4: Tracker::update PnP section (src/tracker.cpp:300-316) -----------------------------------
 * cv::solvePnPRansac(obj, img, K, d, rvec, tvec, false, iters, reproj, conf, inliers).
 * d: NULL / zeros, or the 5 plumb-bob coefficients (k1, k2, p1, p2, k3) of sensor_msgs/CameraInfo::d as the reference
 * forwards them: undistortPoints (5 iterations) in the EPnP kernel and the DLT / planar init, distorted projectPoints in
 * the RANSAC error and the Levenberg-Marquardt refine.  n == 4 takes OpenCV's model_points == npoints path: solvePnP(SOLVEPNP_P3P)
 * on the four correspondences (p3p.cpp: Gao's P3P, the fourth point picks the solution), all four inliers, no refinement;
 * n == 5 likewise returns the EPnP solution as is.  n < 4: MVO_E_ARG (OpenCV asserts). */
int mvo_solve_pnp_ransac(mvo_ctx* ctx, const float* obj, const float* img, int n, const double K[9],
                         const double d[5], int iters, float reproj_err, double confidence, double rvec[3],
                         double tvec[3], int* inlier_idx, int* n_inliers);

/* ---- a6: Initializer pose section (src/initializer.cpp:226-249) ----------------------------------
 * cv::findEssentialMat(p1, p2, K, RANSAC, prob, thr, mask) and cv::recoverPose(E, p1, p2, K, R, t, mask). */
int mvo_find_essential_ransac(mvo_ctx* ctx, const float* p1, const float* p2, int n, const double K[9],
                              double prob, double thr, int max_iters, uint8_t* mask, double E[9],
                              int* n_inliers);
int mvo_recover_pose(mvo_ctx* ctx, const double E[9], const float* p1, const float* p2, int n,
                     const double K[9], double R[9], double t[3], uint8_t* mask_io, int* n_good);

/* ---- a7: triangulation (src/tracker.cpp:138-180, src/initializer.cpp:112-163) --------------------
 * cv::triangulatePoints(P1, P2, p1, p2, X4) + cv::convertPointsFromHomogeneous: X3 is n x 3 float. */
int mvo_triangulate(mvo_ctx* ctx, const double P1[12], const double P2[12], const float* p1,
                    const float* p2, int n, float* X3);

/* ==== frame-batch mode ============================================================
 * `cfg.batch` independent camera streams live in one context; every kernel launch covers all of them.
 * Frames live in a device ring (`cfg.ring_frames` entries of `batch` frames, >= 2 for tracking) so a step starts with its
 * inputs resident in HBM.  One step (mvo_batch_track) is the reference's Tracker::update (src/tracker.cpp:274-333) for
 * every slot at once, each slot on its own branch.
 *
 * Ring contract: LK reads level 0 of both pyramids IN PLACE from the ring - the entry tracked (or seeded) last is the
 * "previous image" of the next step (the reference's prev_frame_.image, src/tracker.cpp:61-68).  Keep it intact until the
 * following step has been enqueued; an asynchronous upload into it then waits on the device for that step's LK launch.
 * Overwriting it earlier is detected: the next mvo_batch_track fails with MVO_E_ARG until mvo_batch_seed runs again. */

/* Tracker state of a slot (mvo_batch_track): TrackerState of include/mono_vo/tracker.hpp:20-24 after the INITIALIZING
 * hand-over.  LOST is terminal (src/tracker.cpp:277-279). */
#define MVO_TRACK_TRACKING 0
#define MVO_TRACK_LOST 1
/* What a slot did on this frame (mvo_step_result.flags). */
#define MVO_STEP_LOST_NOW 1u     /* fewer than min_tracked_points survivors: LOST, no pose     (src/tracker.cpp:292-296) */
#define MVO_STEP_POSE 2u         /* Tracker::update returned a pose (rvec / tvec are T_cw)      (src/tracker.cpp:315-316) */
#define MVO_STEP_KF_CHECKED 4u   /* should_add_keyframe was true: H / F RANSAC ran             (src/tracker.cpp:319-320) */
#define MVO_STEP_KEYFRAME 8u     /* has_parallax was true: add_new_keyframe ran                (src/tracker.cpp:321-322) */
/* solvePnPRansac found no model.  DELIBERATE DIVERGENCE from undefined behaviour in the reference: it ignores the return
 * value (src/tracker.cpp:309-315); OpenCV 4.x creates rvec / tvec (3x1) before RANSAC and, without a model, assigns them
 * from an uninitialised local Mat, so cv::Rodrigues turns whatever bytes are there into the frame's "pose" and the
 * key-frame test runs on it.  Defined here instead: the frame has NO pose (no MVO_STEP_POSE, last pose kept), the stream
 * stays TRACKING, tracking_count_from_keyframe_ is incremented, no key-frame test, and the LK survivors become prev_frame_
 * (src/tracker.cpp:331) so that a later frame can recover.  Parity unpinned (no OpenCV here). */
#define MVO_STEP_PNP_FAILED 16u

typedef struct mvo_step_result {
  int n_prev;         /* points fed to LK */
  int n_tracked;      /* status && err < tracking_error_thresh */
  int pnp_ok, n_pnp_inliers;
  double rvec[3], tvec[3];
  int score_h, score_f;
  int n_keypoints, n_matches, n_triangulated;
  int state;          /* MVO_TRACK_* after this frame */
  unsigned flags;     /* MVO_STEP_* */
  int tracking_count; /* tracking_count_from_keyframe_ after this frame */
  int n_tracks;       /* observations with landmarks carried into the next frame (prev_frame_) */
} mvo_step_result;

int mvo_batch_preload_frame(mvo_ctx* ctx, int slot, int frame_idx, const uint8_t* img, int w, int h, int stride,
                            int channels);
/* ORB on ring frame `frame_idx` of every slot: tracks := all key-points, key-frame descriptors := theirs.
 * landmarks (optional, [batch] pointers to n x 3 floats in key-point order) are filled by `depth_fn`-free
 * callers through mvo_batch_set_landmarks after reading the key-points back with mvo_batch_get_tracks. */
int mvo_batch_seed(mvo_ctx* ctx, int frame_idx, int* n_keypoints /* [batch] */);
int mvo_batch_get_tracks(mvo_ctx* ctx, int slot, float* pts /* cap x 2 */, int cap, int* n);
int mvo_batch_set_landmarks(mvo_ctx* ctx, int slot, const float* xyz /* n x 3 */, int n);
int mvo_batch_set_intrinsics(mvo_ctx* ctx, const double K[9], const double d[5]);

/* ---- frame-batch mode as B x Tracker::update (src/tracker.cpp:274-333), device driven ----------------------------------
 * Every slot carries its own tracker state on the device (state, tracking_count_from_keyframe_, last key-frame, tracks)
 * and takes its own branch per frame: LOST below min_tracked_points (:292-296), should_add_keyframe (:118-136) from the
 * slot's PnP pose, has_parallax (:237-268) from its H / F scores, add_new_keyframe (:182-235) only for the slots that
 * pass - run over a compacted device-resident slot list.  No host wait inside a step: launches are sized for the worst
 * case and read the real counts on the device, so a step can be enqueued asynchronously and several contexts interleave
 * on one GPU.  Seed with mvo_batch_seed + mvo_batch_set_landmarks (the Initializer's hand-over, src/mono_vo.cpp:102-105).
 *   mvo_batch_track_async  enqueue the step on ring frame `frame_idx`; returns at once.  TWO steps may be in flight: enqueue
 *                          frame k+1 while frame k still runs and the device never waits for the host between them
 *                          (a third call before a wait is MVO_E_ARG)
 *   mvo_batch_track_poll   1 when the OLDEST step in flight has finished (or none is in flight), 0 while it runs
 *   mvo_batch_track_wait   block until the oldest step in flight has finished, copy its per-slot results to out[batch] (may be
 *                          NULL); MVO_E_CAPACITY if a device-side capacity was exceeded (results clamped)
 *   mvo_batch_track        both
 *   mvo_batch_set_policy   0: the reference's key-frame policy (default); benchmarking loads (LOST handling
 *                          unchanged): 1 = key-frame branch on every tracked frame (worst case), 2 = never a key-frame
 *                          (the always-on part of the step: LK + PnP)
 *   mvo_batch_get_state    MVO_TRACK_* and tracking_count_from_keyframe_ of every slot (blocks) */
int mvo_batch_track_async(mvo_ctx* ctx, int frame_idx);
int mvo_batch_track_poll(mvo_ctx* ctx);
int mvo_batch_track_wait(mvo_ctx* ctx, mvo_step_result* out /* [batch] */);
int mvo_batch_track(mvo_ctx* ctx, int frame_idx, mvo_step_result* out /* [batch] */);
int mvo_batch_set_policy(mvo_ctx* ctx, int policy);
int mvo_batch_get_state(mvo_ctx* ctx, int* state /* [batch] */, int* tracking_count /* [batch] */);

/* Single-stream forms of the same step (a context with cfg.batch == 1, cfg.ring_frames >= 2): mvo_set_intrinsics is
 * mvo_batch_set_intrinsics; mvo_tracker_step uploads `img` into the next ring entry and runs Tracker::update on it - the
 * fused per-frame call of src/mono_vo.cpp:116.  Seed once: mvo_batch_preload_frame(ctx, 0, 0, ...), mvo_batch_seed(ctx, 0, ...),
 * mvo_batch_set_landmarks(ctx, 0, ...). */
int mvo_set_intrinsics(mvo_ctx* ctx, const double K[9], const double d[5]);
int mvo_tracker_step(mvo_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels, mvo_step_result* out);

/* ---- output side on the device (src/mono_vo.cpp:117-152, src/utils.cpp:85-243) ---------------------------------------------
 * What MonoVO::image_callback derives from Tracker::update's result before anything reaches a ROS topic, kept per slot on
 * the device and updated at the end of every mvo_batch_track step once enabled:
 *   last_pose_ / tracking_valid_ (mono_vo.cpp:119-131) as the REP-103 pose of affine3d_to_odometry_msg /
 *     affine3d_to_transform_stamped_msg (utils.cpp:85-188): pose_wc = (R_cw, t_cw)^-1 conjugated with
 *     M = [0 0 1; -1 0 0; 0 -1 0], orientation by tf2::Matrix3x3::getRotation + normalize (x, y, z, w);
 *   the path (mono_vo.cpp:134-148): one pose appended per frame while tracking is valid;
 *   the Map as sensor_msgs/PointCloud2 payload (points3d_to_pointcloud_msg, utils.cpp:190-243): every landmark in id
 *     order - the seed landmarks of mvo_batch_set_landmarks, then what each add_new_keyframe creates (tracker.cpp:218-223) -
 *     as (z, -x, -y) float32, point_step 12.  width = *n, row_step = 12 * *n, data = the buffer.
 * mvo_batch_enable_output allocates `map_capacity` landmarks and `path_capacity` poses per slot (MVO_E_CAPACITY from
 * mvo_batch_track_wait when a slot would exceed them; the excess is dropped); call it before mvo_batch_set_landmarks.
 * The header stamps / frame ids and the fixed covariances (utils.cpp:131-146) stay with the caller's message objects. */
typedef struct mvo_ros_pose {
  double position[3];     /* REP-103: x forward, y left, z up */
  double orientation[4];  /* x, y, z, w */
  int tracking_valid;     /* tracking_valid_ */
  int has_pose;           /* last_pose_.has_value() */
} mvo_ros_pose;
int mvo_batch_enable_output(mvo_ctx* ctx, int map_capacity, int path_capacity);
int mvo_batch_get_odometry(mvo_ctx* ctx, mvo_ros_pose* out /* [batch] */);
int mvo_batch_get_path(mvo_ctx* ctx, int slot, double* poses /* cap x 7: position xyz, orientation xyzw */, int cap, int* n);
int mvo_batch_get_pointcloud(mvo_ctx* ctx, int slot, float* data /* cap x 3 */, int cap, int* n);

/* ---- asynchronous ingest (src/mono_vo.cpp:92-100: the image the callback hands to the tracker) ---------------------------
 * mvo_batch_upload_async copies all `batch` mono8 frames of ring entry `frame_idx` (images `slot_stride` bytes apart, rows
 * `stride` bytes apart) host -> device on a dedicated upload stream and returns at once; the step that uses the entry
 * waits for the copy on the device, and the copy waits for the last launch that reads the entry (the LK launch of the step
 * AFTER the one the entry was tracked on, see the ring contract above).  With a ring of 2 entries the upload of frame k+1
 * overlaps what step k does behind its LK launch, with >= 3 entries the whole step.  The host buffer must stay valid until the step that consumes it has
 * been enqueued and its wait / poll has returned; pinned memory (mvo_host_alloc, or hipHostRegister'd memory) is what
 * makes the copy asynchronous - pageable memory works but serialises. */
int mvo_batch_upload_async(mvo_ctx* ctx, int frame_idx, const uint8_t* frames, int w, int h, int stride, size_t slot_stride);
int mvo_host_alloc(size_t bytes, void** out);
int mvo_host_free(void* p);

/* Stage timers: HIP events recorded on the context's stream around each kernel group.
 * Names: "lk_pyramid", "lk_track", "orb_detect", "orb_describe", "match", "pnp", "ransac_h", "ransac_f", ... */
int mvo_profile_enable(mvo_ctx* ctx, int on);
int mvo_profile_read(mvo_ctx* ctx, const char* name, double* total_ms, int* launches);
int mvo_profile_reset(mvo_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* MVO_H_ */
