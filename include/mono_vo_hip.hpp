// include/mono_vo_hip.hpp — header-only C++17 binding: the reference's FeatureProcessor / Frame / KeyFrame / Map /
// Initializer / Tracker and the dispatch of MonoVO::image_callback, over the C ABI of include/mvo.h and plain structs
// (no OpenCV, no ROS).  Same names, control flow, defaults and quirks as Tatsuya-2/ros2_mono_vo:
//   FeatureProcessor  src/feature_processor.cpp:5-41       Frame     src/frame.cpp        Landmark  src/landmark.cpp
//   Initializer       src/initializer.cpp:52-313           KeyFrame  src/keyframe.cpp     Map       src/map.cpp
//   Tracker           src/tracker.cpp:58-333               VisualOdometry = src/mono_vo.cpp:83-131 without the ROS shell
// Every cv:: call of the reference is one mvo_* call here (the boundary); the classes hold only bookkeeping.  Deliberate
// host-side deviations: the id counters live in the Map instead of process-global statics (src/landmark.cpp:5,
// src/keyframe.cpp:6) so several streams coexist; logging is dropped; a solvePnPRansac without a model yields a frame
// without a pose where the reference reads an uninitialised rvec (src/tracker.cpp:309-315; mvo.h MVO_STEP_PNP_FAILED).
// The Python mirror (ros2_mono_vo_amd/vo.py) is the same code in the other language; tools/mvo_run.cpp is the harness.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "mvo.h"

namespace mono_vo {

struct Error : std::runtime_error { int code; Error(int c, const std::string& m) : std::runtime_error(m), code(c) {} };

using KeyPoint = mvo_keypoint;   // cv::KeyPoint layout
using DMatch = mvo_match;        // cv::DMatch layout
struct Point2f { float x, y; };
struct Point3f { float x, y, z; };
using Descriptor = std::array<uint8_t, 32>;
using Mat3 = std::array<double, 9>;    // row major
using Affine3d = std::array<double, 16>;   // 4 x 4 row major, like cv::Affine3d::matrix

// A borrowed mono8 / bgr8 image (cv::Mat header without the ownership).
struct Image {
  const uint8_t* data = nullptr;
  int width = 0, height = 0, stride = 0, channels = 1;   // channels: the codes of include/mvo.h
};

inline Affine3d affine_identity() { Affine3d T{}; T[0] = T[5] = T[10] = T[15] = 1; return T; }
inline Affine3d affine(const Mat3& R, const double t[3]) {
  Affine3d T = affine_identity();
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[4 * i + j] = R[3 * i + j]; T[4 * i + 3] = t[i]; }
  return T;
}
inline Affine3d affine_mul(const Affine3d& A, const Affine3d& B) {
  Affine3d C{};
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += A[4 * i + k] * B[4 * k + j]; C[4 * i + j] = s; }
  return C;
}
// cv::Affine3d::inv() is a general 4 x 4 inverse, not a transpose: Gauss-Jordan with partial pivoting
inline Affine3d affine_inv(const Affine3d& T) {
  double a[4][8];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { a[i][j] = T[4 * i + j]; a[i][4 + j] = i == j; }
  for (int c = 0; c < 4; c++) {
    int p = c;
    for (int r = c + 1; r < 4; r++) if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
    if (p != c) for (int j = 0; j < 8; j++) std::swap(a[p][j], a[c][j]);
    const double d = a[c][c];
    for (int j = 0; j < 8; j++) a[c][j] /= d;
    for (int r = 0; r < 4; r++) if (r != c) { const double f = a[r][c]; for (int j = 0; j < 8; j++) a[r][j] -= f * a[c][j]; }
  }
  Affine3d R;
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) R[4 * i + j] = a[i][4 + j];
  return R;
}
// cv::Rodrigues(rvec) (src/tracker.cpp:315): plain host arithmetic, as in the reference
inline Mat3 rodrigues(const double r[3]) {
  const double th = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (th < std::numeric_limits<double>::epsilon()) return {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double k[3] = {r[0] / th, r[1] / th, r[2] / th}, c = std::cos(th), s = std::sin(th);
  const double Kx[9] = {0, -k[2], k[1], k[2], 0, -k[0], -k[1], k[0], 0};
  Mat3 R;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[3 * i + j] = c * (i == j) + (1 - c) * k[i] * k[j] + s * Kx[3 * i + j];
  return R;
}

// ---- the boundary: one method per cv:: call site ---------------------------------------------------------------------
class Backend {
 public:
  explicit Backend(const mvo_config& cfg) {
    const int rc = mvo_create(&cfg, &ctx_);
    if (rc != MVO_OK) throw Error(rc, "mvo_create failed (no HIP device / out of memory?)");
    cap_ = cfg.max_points;
  }
  ~Backend() { mvo_destroy(ctx_); }
  Backend(const Backend&) = delete;
  Backend& operator=(const Backend&) = delete;
  mvo_ctx* ctx() const { return ctx_; }

  void detect_and_compute(const Image& im, std::vector<KeyPoint>& kps, std::vector<Descriptor>& desc) const {   // cv::ORB
    kps.resize(cap_); desc.resize(cap_);
    int n = 0;
    check(mvo_orb_detect_and_compute(ctx_, im.data, im.width, im.height, im.stride, im.channels, kps.data(), desc[0].data(), cap_, &n));
    kps.resize(n); desc.resize(n);
  }
  std::vector<DMatch> find_matches(const std::vector<Descriptor>& q, const std::vector<Descriptor>& t, double ratio) const {
    std::vector<DMatch> out(std::max<size_t>(q.size(), 1));
    int n = 0;
    check(mvo_match_knn2_ratio(ctx_, q.empty() ? nullptr : q[0].data(), (int)q.size(), t.empty() ? nullptr : t[0].data(), (int)t.size(), ratio,
                               out.data(), (int)out.size(), &n));
    out.resize(n);
    return out;
  }
  void lk_track(const Image& prev, const Image& next, const std::vector<Point2f>& pts, std::vector<Point2f>& out, std::vector<uint8_t>& status,
                std::vector<float>& err) const {   // cv::calcOpticalFlowPyrLK
    const int n = (int)pts.size();
    out.assign(n, Point2f{0, 0}); status.assign(n, 0); err.assign(n, 0.f);
    if (n == 0) return;
    check(mvo_lk_track(ctx_, prev.data, next.data, prev.width, prev.height, prev.stride, prev.channels, &pts[0].x, n, &out[0].x, status.data(), err.data()));
  }
  int homography_inliers(const std::vector<Point2f>& a, const std::vector<Point2f>& b, double thr) const {   // cv::findHomography
    std::vector<uint8_t> mask(std::max<size_t>(a.size(), 1));
    double H[9]; int n = 0;
    check(mvo_find_homography_ransac(ctx_, a.empty() ? nullptr : &a[0].x, b.empty() ? nullptr : &b[0].x, (int)a.size(), thr, 2000, 0.995, mask.data(), H, &n), true);
    return n;
  }
  int fundamental_inliers(const std::vector<Point2f>& a, const std::vector<Point2f>& b, double thr) const {   // cv::findFundamentalMat
    std::vector<uint8_t> mask(std::max<size_t>(a.size(), 1));
    double F[9]; int n = 0;
    check(mvo_find_fundamental_ransac(ctx_, a.empty() ? nullptr : &a[0].x, b.empty() ? nullptr : &b[0].x, (int)a.size(), thr, 0.99, 1000, mask.data(), F, &n), true);
    return n;
  }
  bool essential(const std::vector<Point2f>& a, const std::vector<Point2f>& b, const Mat3& K, std::vector<uint8_t>& mask, Mat3& E) const {
    mask.assign(a.size(), 0);
    int n = 0;
    return check(mvo_find_essential_ransac(ctx_, &a[0].x, &b[0].x, (int)a.size(), K.data(), 0.99, 1.0, 1000, mask.data(), E.data(), &n), true) == MVO_OK;
  }
  int recover_pose(const Mat3& E, const std::vector<Point2f>& a, const std::vector<Point2f>& b, const Mat3& K, Mat3& R, double t[3], std::vector<uint8_t>& mask) const {
    int good = 0;
    check(mvo_recover_pose(ctx_, E.data(), &a[0].x, &b[0].x, (int)a.size(), K.data(), R.data(), t, mask.data(), &good));
    return good;
  }
  bool solve_pnp_ransac(const std::vector<Point3f>& obj, const std::vector<Point2f>& img, const Mat3& K, const double d[5], double rvec[3], double tvec[3],
                        int& n_inliers) const {
    std::vector<int> idx(std::max<size_t>(obj.size(), 1));
    n_inliers = 0;
    return check(mvo_solve_pnp_ransac(ctx_, &obj[0].x, &img[0].x, (int)obj.size(), K.data(), d, 100, 8.0f, 0.99, rvec, tvec, idx.data(), &n_inliers), true) == MVO_OK;
  }
  std::vector<Point3f> triangulate(const double P1[12], const double P2[12], const std::vector<Point2f>& a, const std::vector<Point2f>& b) const {
    std::vector<Point3f> X(a.size());
    if (!a.empty()) check(mvo_triangulate(ctx_, P1, P2, &a[0].x, &b[0].x, (int)a.size(), &X[0].x));
    return X;
  }

 private:
  int check(int rc, bool allow_degenerate = false) const {
    if (rc == MVO_OK || (allow_degenerate && rc == MVO_E_DEGENERATE)) return rc;
    throw Error(rc, std::string("mvo: ") + mvo_last_error(ctx_));   // OpenCV would throw cv::Exception here
  }
  mvo_ctx* ctx_ = nullptr;
  int cap_ = 0;
};

// ---- frame-batch mode: B camera streams resident in one context, every one its own Tracker on the device ---------------
// (mvo_batch_* of include/mvo.h; csrc/track.hip).  A slot is what `Tracker tracker_` + `Map map_` are for one camera in
// src/mono_vo.cpp; `track` is Tracker::update for all of them, the output side is image_callback's pose / path / cloud
// bookkeeping (src/mono_vo.cpp:117-152) kept per slot on the device.
class BatchTracker {
 public:
  BatchTracker(int width, int height, int batch, int nfeatures = 1000, int max_points = 4096, int ring_frames = 2) : B_(batch), cap_(max_points) {
    mvo_config cfg;
    mvo_config_default(&cfg);
    cfg.max_width = width; cfg.max_height = height; cfg.batch = batch; cfg.nfeatures = nfeatures; cfg.max_points = max_points;
    cfg.ring_frames = ring_frames;
    const int rc = mvo_create(&cfg, &ctx_);
    if (rc != MVO_OK) throw Error(rc, "mvo_create failed (no HIP device / out of memory?)");
  }
  ~BatchTracker() { mvo_destroy(ctx_); }
  BatchTracker(const BatchTracker&) = delete;
  BatchTracker& operator=(const BatchTracker&) = delete;
  mvo_ctx* ctx() const { return ctx_; }
  int batch() const { return B_; }

  void set_intrinsics(const Mat3& K, const double d[5]) { check(mvo_batch_set_intrinsics(ctx_, K.data(), d)); }
  // one camera's frame into ring entry `entry` (synchronous; any encoding code of mvo.h) ...
  void preload_frame(int slot, int entry, const Image& im) { check(mvo_batch_preload_frame(ctx_, slot, entry, im.data, im.width, im.height, im.stride, im.channels)); }
  // ... or all B mono8 frames of an entry in one asynchronous copy (pinned memory: mvo_host_alloc)
  void upload_async(int entry, const uint8_t* frames, int w, int h, int stride, size_t slot_stride) { check(mvo_batch_upload_async(ctx_, entry, frames, w, h, stride, slot_stride)); }
  // the Initializer's hand-over: ORB on entry `entry`, every key-point a track; then the landmarks per slot
  std::vector<int> seed(int entry) {
    std::vector<int> n(B_);
    check(mvo_batch_seed(ctx_, entry, n.data()));
    return n;
  }
  std::vector<Point2f> tracks(int slot) const {
    std::vector<Point2f> p(cap_);
    int n = 0;
    check(mvo_batch_get_tracks(ctx_, slot, &p[0].x, cap_, &n));
    p.resize(n);
    return p;
  }
  void set_landmarks(int slot, const std::vector<Point3f>& lm) { check(mvo_batch_set_landmarks(ctx_, slot, lm.empty() ? nullptr : &lm[0].x, (int)lm.size())); }
  void set_policy(int policy) { check(mvo_batch_set_policy(ctx_, policy)); }
  // Tracker::update for every slot on ring entry `entry`
  std::vector<mvo_step_result> track(int entry) {
    std::vector<mvo_step_result> r(B_);
    check(mvo_batch_track(ctx_, entry, r.data()));
    return r;
  }
  void track_async(int entry) { check(mvo_batch_track_async(ctx_, entry)); }
  bool poll() const { return mvo_batch_track_poll(ctx_) != 0; }
  std::vector<mvo_step_result> wait() {
    std::vector<mvo_step_result> r(B_);
    check(mvo_batch_track_wait(ctx_, r.data()));
    return r;
  }
  void state(std::vector<int>& st, std::vector<int>& tracking_count) const {
    st.assign(B_, 0); tracking_count.assign(B_, 0);
    check(mvo_batch_get_state(ctx_, st.data(), tracking_count.data()));
  }
  // output side (src/mono_vo.cpp:117-152, src/utils.cpp:85-243); enable before set_landmarks
  void enable_output(int map_capacity, int path_capacity) {
    check(mvo_batch_enable_output(ctx_, map_capacity, path_capacity));
    map_cap_ = map_capacity; path_cap_ = path_capacity;
  }
  std::vector<mvo_ros_pose> odometry() const {
    std::vector<mvo_ros_pose> r(B_);
    check(mvo_batch_get_odometry(ctx_, r.data()));
    return r;
  }
  std::vector<std::array<double, 7>> path(int slot) const {   // position xyz, orientation xyzw per pose
    std::vector<std::array<double, 7>> p(path_cap_);
    int n = 0;
    check(mvo_batch_get_path(ctx_, slot, p.empty() ? nullptr : p[0].data(), path_cap_, &n));
    p.resize(n);
    return p;
  }
  std::vector<Point3f> pointcloud(int slot) const {           // PointCloud2 payload: ROS axes, point_step 12
    std::vector<Point3f> c(map_cap_);
    int n = 0;
    check(mvo_batch_get_pointcloud(ctx_, slot, c.empty() ? nullptr : &c[0].x, map_cap_, &n));
    c.resize(n);
    return c;
  }

 private:
  void check(int rc) const {
    if (rc != MVO_OK) throw Error(rc, std::string("mvo: ") + mvo_last_error(ctx_));
  }
  mvo_ctx* ctx_ = nullptr;
  int B_ = 1, cap_ = 0, map_cap_ = 0, path_cap_ = 0;
};

// ---- src/feature_processor.cpp ------------------------------------------------------------------------------------------
class FeatureProcessor {
 public:
  using Ptr = std::shared_ptr<FeatureProcessor>;
  FeatureProcessor(std::shared_ptr<Backend> b, int num_features = 1000) : backend(std::move(b)), num_features(num_features) {}
  void detect_and_compute(const Image& image, std::vector<KeyPoint>& kps, std::vector<Descriptor>& desc) const { backend->detect_and_compute(image, kps, desc); }
  std::vector<DMatch> find_matches(const std::vector<Descriptor>& d1, const std::vector<Descriptor>& d2, double lowes_distance_ratio) const {
    return backend->find_matches(d1, d2, lowes_distance_ratio);
  }
  std::shared_ptr<Backend> backend;
  int num_features;
};

enum class ObservationFilter { ALL, WITH_LANDMARKS, WITHOUT_LANDMARKS };

// ---- src/frame.cpp: observations as parallel arrays -------------------------------------------------------------------------
struct Frame {
  std::vector<uint8_t> pixels;   // Frame(const cv::Mat&) clones the image
  Image image;
  Affine3d pose_wc = affine_identity();
  std::vector<KeyPoint> kps;
  std::vector<Descriptor> desc;
  std::vector<long> landmark_id;
  bool is_tracked = false;

  Frame() = default;
  explicit Frame(const Image& im) {
    if (im.data) {
      const int bpp = im.channels < 0 ? -im.channels : im.channels;
      pixels.resize((size_t)im.width * bpp * im.height);
      for (int y = 0; y < im.height; y++) std::memcpy(&pixels[(size_t)y * im.width * bpp], im.data + (size_t)y * im.stride, (size_t)im.width * bpp);
      image = Image{pixels.data(), im.width, im.height, im.width * bpp, im.channels};
    }
  }
  Frame(const Frame& o) { *this = o; }
  Frame& operator=(const Frame& o) {
    pixels = o.pixels; image = o.image; image.data = pixels.empty() ? nullptr : pixels.data();
    pose_wc = o.pose_wc; kps = o.kps; desc = o.desc; landmark_id = o.landmark_id; is_tracked = o.is_tracked;
    return *this;
  }
  size_t size() const { return kps.size(); }
  void extract_observations(const FeatureProcessor& fp) {
    std::vector<KeyPoint> k; std::vector<Descriptor> d;
    fp.detect_and_compute(image, k, d);
    kps.insert(kps.end(), k.begin(), k.end());
    desc.insert(desc.end(), d.begin(), d.end());
    landmark_id.insert(landmark_id.end(), k.size(), -1L);
  }
  void clear_observations() { kps.clear(); desc.clear(); landmark_id.clear(); }
  std::vector<Point2f> get_points_2d(ObservationFilter f = ObservationFilter::ALL) const {
    std::vector<Point2f> p;
    for (size_t i = 0; i < kps.size(); i++)
      if (f == ObservationFilter::ALL || (f == ObservationFilter::WITH_LANDMARKS) == (landmark_id[i] != -1)) p.push_back({kps[i].x, kps[i].y});
    return p;
  }
};

struct Landmark { long id; Point3f pose_w; Descriptor descriptor; };

// ---- src/keyframe.cpp: a Frame minus its image plus landmark id -> observation index, built ONCE at construction (ids
// back-filled later by the tracker are not indexed - SURVEY Appendix B #5; duplicates map to the last index) -----------------
struct KeyFrame {
  using Ptr = std::shared_ptr<KeyFrame>;
  long id;
  Affine3d pose_wc;
  std::vector<KeyPoint> kps;
  std::vector<Descriptor> desc;
  std::vector<long> landmark_id;
  std::unordered_map<long, size_t> landmark_id_to_index;
  KeyFrame(long id, const Affine3d& pose) : id(id), pose_wc(pose) {}
  KeyFrame(long id, const Frame& f) : id(id), pose_wc(f.pose_wc), kps(f.kps), desc(f.desc), landmark_id(f.landmark_id) {
    for (size_t i = 0; i < landmark_id.size(); i++) if (landmark_id[i] != -1) landmark_id_to_index[landmark_id[i]] = i;
  }
  std::vector<Point2f> get_points_2d_for_landmarks(const std::vector<long>& ids) const {
    std::vector<Point2f> p;
    for (long l : ids) { auto it = landmark_id_to_index.find(l); if (it != landmark_id_to_index.end()) p.push_back({kps[it->second].x, kps[it->second].y}); }
    return p;
  }
};

// ---- src/map.cpp; owns the id counters (process-global statics in the reference) ---------------------------------------------
class Map {
 public:
  using Ptr = std::shared_ptr<Map>;
  std::map<long, Landmark> landmarks;
  std::map<long, KeyFrame::Ptr> keyframes;
  long last_keyframe_id = -1;
  Landmark new_landmark(const Point3f& p, const Descriptor& d) { return Landmark{next_landmark_id_++, p, d}; }
  void add_landmark(const Landmark& lm) { landmarks.emplace(lm.id, lm); }   // emplace keeps an existing entry
  KeyFrame::Ptr new_keyframe(const Affine3d& pose) { return std::make_shared<KeyFrame>(next_keyframe_id_++, pose); }
  KeyFrame::Ptr new_keyframe(const Frame& f) { return std::make_shared<KeyFrame>(next_keyframe_id_++, f); }
  void add_keyframe(const KeyFrame::Ptr& kf) { keyframes.emplace(kf->id, kf); last_keyframe_id = kf->id; }
  KeyFrame::Ptr get_last_keyframe() const { return keyframes.at(last_keyframe_id); }
  // src/map.cpp:15-31: the 2D-3D gather that feeds solvePnPRansac
  void get_observation_to_landmark_point_correspondences(const Frame& f, std::vector<Point2f>& p2, std::vector<Point3f>& p3) const {
    p2.clear(); p3.clear();
    for (size_t i = 0; i < f.kps.size(); i++)
      if (f.landmark_id[i] != -1) { p2.push_back({f.kps[i].x, f.kps[i].y}); p3.push_back(landmarks.at(f.landmark_id[i]).pose_w); }
  }
 private:
  long next_landmark_id_ = 0, next_keyframe_id_ = 0;
};

struct ParallaxResult { bool ok; int score_h, score_f; };
// Tracker::has_parallax / Initializer::check_parallax (src/tracker.cpp:237-268, src/initializer.cpp:77-110) with their
// unguarded divisions (NaN / inf comparisons fall through)
inline ParallaxResult check_parallax_impl(const Backend& b, const std::vector<Point2f>& p1, const std::vector<Point2f>& p2, double thr, double f_inlier_thresh,
                                          double model_score_thresh) {
  const int sh = b.homography_inliers(p1, p2, thr), sf = b.fundamental_inliers(p1, p2, thr);
  if (static_cast<double>(sf) / static_cast<double>(p1.size()) < f_inlier_thresh) return {false, sh, sf};
  const double model_score = static_cast<double>(sh) / static_cast<double>(sf);
  if (model_score > model_score_thresh) return {false, sh, sf};
  return {true, sh, sf};
}

struct InitializerParams {   // include/mono_vo/initializer.hpp:109-115
  int occupancy_grid_div = 50; double kp_distribution_thresh = 0.5, lowes_distance_ratio = 0.7; long min_matches_for_init = 100;
  double ransac_reproj_thresh = 1.0, f_inlier_thresh = 0.5, model_score_thresh = 0.56;
};
struct TrackerParams {       // include/mono_vo/tracker.hpp:137-147
  float tracking_error_thresh = 30.0f; long min_observations_before_triangulation = 100, min_tracked_points = 10, max_tracking_after_keyframe = 10;
  double max_rotation_from_keyframe = M_PI * 15.0 / 180.0, max_translation_from_keyframe = 1.0, ransac_reproj_thresh = 1.0, model_score_thresh = 0.85,
         f_inlier_thresh = 0.5, lowes_distance_ratio = 0.7;
};

enum class State { OBTAINING_REF, INITIALIZING, INITIALIZED };

// ---- src/initializer.cpp: two-view bootstrap --------------------------------------------------------------------------------------
class Initializer {
 public:
  Initializer(Map::Ptr map, FeatureProcessor::Ptr fp, InitializerParams p = {}) : map_(std::move(map)), fp_(std::move(fp)), p_(p) {}
  bool is_initalized() const { return state_ == State::INITIALIZED; }   // [sic] the reference's spelling
  void reset() { state_ = State::OBTAINING_REF; }
  State state() const { return state_; }

  // src/initializer.cpp:52-75 incl. its quirk: c = x / div can equal grid.cols when W % div != 0 and the unchecked
  // Mat::at then aliases the next row (or the byte after the buffer) - mirrored with a flat index + slack
  bool good_keypoint_distribution(const Frame& f) const {
    const int div = p_.occupancy_grid_div, rows = f.image.height / div, cols = f.image.width / div;
    std::vector<uint8_t> grid((size_t)rows * cols + cols + 1, 0);
    int occupied = 0;
    for (const auto& k : f.kps) {
      const int r = (int)(k.y / (float)div), c = (int)(k.x / (float)div);
      const long idx = (long)r * cols + c;
      if (idx >= 0 && idx < (long)grid.size() && !grid[idx]) { grid[idx] = 1; occupied++; }
    }
    const int total = cols * rows;
    const double occupancy = total ? (double)occupied / total : std::numeric_limits<double>::infinity();
    return occupancy > p_.kp_distribution_thresh;
  }
  bool check_parallax(const std::vector<Point2f>& p1, const std::vector<Point2f>& p2) {
    const auto r = check_parallax_impl(*fp_->backend, p1, p2, p_.ransac_reproj_thresh, p_.f_inlier_thresh, p_.model_score_thresh);
    last_score_h = r.score_h; last_score_f = r.score_f;
    return r.ok;
  }
  // src/initializer.cpp:112-163
  std::vector<Point3f> traingulate_points(const Mat3& K, const Mat3& R, const double t[3], const std::vector<Point2f>& ref, const std::vector<Point2f>& cur,
                                          std::vector<uint8_t>& inliers) const {   // [sic]
    double P1[12], P2[12];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 4; j++) {
        P1[4 * i + j] = j < 3 ? K[3 * i + j] : 0.0;
        double s = 0;
        for (int k = 0; k < 3; k++) s += K[3 * i + k] * (j < 3 ? R[3 * k + j] : t[k]);
        P2[4 * i + j] = s;
      }
    const auto X = fp_->backend->triangulate(P1, P2, ref, cur);
    std::vector<Point3f> out;
    inliers.clear();
    for (const auto& p : X) {
      const double zc = (double)p.x * R[6] + (double)p.y * R[7] + (double)p.z * R[8] + t[2];
      const bool in = p.z > 0 && zc > 0;
      inliers.push_back(in);
      if (in) out.push_back(p);
    }
    return out;
  }
  // src/initializer.cpp:165-313: the reference Frame once initialised, else nullopt
  std::optional<Frame> try_initializing(const Frame& frame, const Mat3& K) {
    if (state_ == State::INITIALIZED) return ref_frame_;
    Frame cur = frame;
    cur.extract_observations(*fp_);
    if (state_ == State::OBTAINING_REF) {
      if (!good_keypoint_distribution(cur)) return std::nullopt;
      ref_frame_ = cur;
      state_ = State::INITIALIZING;
      return std::nullopt;
    }
    const auto good = fp_->find_matches(ref_frame_.desc, cur.desc, p_.lowes_distance_ratio);
    if ((long)good.size() < p_.min_matches_for_init) {
      if (good_keypoint_distribution(cur)) ref_frame_ = cur; else reset();
      return std::nullopt;
    }
    std::vector<Point2f> pr, pc;
    for (const auto& m : good) { pr.push_back({ref_frame_.kps[m.query_idx].x, ref_frame_.kps[m.query_idx].y}); pc.push_back({cur.kps[m.train_idx].x, cur.kps[m.train_idx].y}); }
    if (!check_parallax(pr, pc)) return std::nullopt;
    std::vector<uint8_t> mask;
    Mat3 E{}, R{};
    double t[3];
    fp_->backend->essential(pr, pc, K, mask, E);
    const int num_inliers = fp_->backend->recover_pose(E, pr, pc, K, R, t, mask);
    last_pose_inliers = num_inliers;
    if (num_inliers < 4) return std::nullopt;
    std::vector<Point2f> ri, ci; std::vector<int> qi, ti;
    for (size_t i = 0; i < good.size(); i++) if (mask[i]) { ri.push_back(pr[i]); ci.push_back(pc[i]); qi.push_back(good[i].query_idx); ti.push_back(good[i].train_idx); }
    std::vector<uint8_t> chir;
    const auto pts3d = traingulate_points(K, R, t, ri, ci, chir);
    if (pts3d.size() < 4) { reset(); return std::nullopt; }
    map_->add_keyframe(map_->new_keyframe(affine_identity()));   // origin key-frame: pose only, no observations
    cur.pose_wc = affine_inv(affine(R, t));
    size_t k = 0;
    for (size_t i = 0; i < qi.size(); i++)
      if (chir[i]) {
        const Landmark lm = map_->new_landmark(pts3d[k++], cur.desc[ti[i]]);
        map_->add_landmark(lm);
        cur.landmark_id[ti[i]] = lm.id;
        ref_frame_.landmark_id[qi[i]] = lm.id;
      }
    map_->add_keyframe(map_->new_keyframe(cur));
    ref_frame_ = cur;
    state_ = State::INITIALIZED;
    return ref_frame_;
  }
  int last_score_h = 0, last_score_f = 0, last_pose_inliers = 0;

 private:
  Map::Ptr map_;
  FeatureProcessor::Ptr fp_;
  InitializerParams p_;
  State state_ = State::OBTAINING_REF;
  Frame ref_frame_;
};

enum class TrackerState { INITIALIZING, TRACKING, LOST };

// ---- src/tracker.cpp: per-frame tracking -----------------------------------------------------------------------------------------------
class Tracker {
 public:
  Tracker(Map::Ptr map, FeatureProcessor::Ptr fp, TrackerParams p = {}) : map_(std::move(map)), fp_(std::move(fp)), p_(p) {}
  TrackerState get_state() const { return state_; }
  void reset() { state_ = TrackerState::INITIALIZING; }
  struct Last {
    int n_tracked = 0, n_pnp_inliers = 0, score_h = 0, score_f = 0, n_keypoints = 0, n_matches = 0, n_triangulated = 0;
    bool pnp_ran = false, pnp_ok = false, kf_checked = false, keyframe_added = false;   // what the frame did (mvo_step_result.flags)
  } last;
  long tracking_count_from_keyframe() const { return tracking_count_from_keyframe_; }
  const Frame& prev_frame() const { return prev_frame_; }

  Frame track_frame_with_optical_flow(const Image& new_image) {   // src/tracker.cpp:58-90
    Frame nf(new_image);
    std::vector<Point2f> prev_pts; std::vector<size_t> src;
    for (size_t i = 0; i < prev_frame_.kps.size(); i++) if (prev_frame_.landmark_id[i] != -1) { prev_pts.push_back({prev_frame_.kps[i].x, prev_frame_.kps[i].y}); src.push_back(i); }
    std::vector<Point2f> np_; std::vector<uint8_t> st; std::vector<float> err;
    fp_->backend->lk_track(prev_frame_.image, nf.image, prev_pts, np_, st, err);
    for (size_t i = 0; i < prev_pts.size(); i++)
      if (st[i] && err[i] < p_.tracking_error_thresh) {
        nf.kps.push_back(KeyPoint{np_[i].x, np_[i].y, 1.f, -1.f, 0.f, 0, -1});   // cv::KeyPoint(pt, 1)
        nf.desc.push_back(prev_frame_.desc[src[i]]);
        nf.landmark_id.push_back(prev_frame_.landmark_id[src[i]]);
      }
    nf.is_tracked = true;
    return nf;
  }
  bool has_significant_motion(const Frame& f) const {   // src/tracker.cpp:92-116
    const Affine3d rel = affine_mul(affine_inv(map_->get_last_keyframe()->pose_wc), f.pose_wc);
    const double translation = std::sqrt(rel[3] * rel[3] + rel[7] * rel[7] + rel[11] * rel[11]);
    if (translation > p_.max_translation_from_keyframe) return true;
    const double rotation = std::acos((rel[0] + rel[5] + rel[10] - 1.0) / 2.0);   // NaN outside [-1, 1]: the test is false
    return rotation > p_.max_rotation_from_keyframe;
  }
  bool should_add_keyframe(const Frame& f) const {   // src/tracker.cpp:118-136
    if ((long)f.size() < p_.min_observations_before_triangulation) return true;
    if ((long)tracking_count_from_keyframe_ > p_.max_tracking_after_keyframe) return true;
    return has_significant_motion(f);
  }
  bool has_parallax(const Frame& f) {   // src/tracker.cpp:237-268
    const auto p1 = map_->get_last_keyframe()->get_points_2d_for_landmarks(f.landmark_id);
    const auto p2 = f.get_points_2d();
    const auto r = check_parallax_impl(*fp_->backend, p1, p2, p_.ransac_reproj_thresh, p_.f_inlier_thresh, p_.model_score_thresh);
    last.score_h = r.score_h; last.score_f = r.score_f; last.kf_checked = true;
    return r.ok;
  }
  std::vector<Point3f> triangulate_points(const Affine3d& ref_cw, const Affine3d& cur_cw, const Mat3& K, const std::vector<Point2f>& ref, const std::vector<Point2f>& cur,
                                          std::vector<uint8_t>& inliers) const {   // src/tracker.cpp:138-180
    double P1[12], P2[12];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 4; j++) {
        double a = 0, b = 0;
        for (int k = 0; k < 3; k++) { a += K[3 * i + k] * ref_cw[4 * k + j]; b += K[3 * i + k] * cur_cw[4 * k + j]; }
        P1[4 * i + j] = a; P2[4 * i + j] = b;
      }
    const auto X = fp_->backend->triangulate(P1, P2, ref, cur);
    std::vector<Point3f> out;
    inliers.clear();
    for (const auto& p : X) {
      const float zr = (float)((double)p.x * ref_cw[8] + (double)p.y * ref_cw[9] + (double)p.z * ref_cw[10] + ref_cw[11]);   // Affine3d * Point3f -> Point3f
      const float zc = (float)((double)p.x * cur_cw[8] + (double)p.y * cur_cw[9] + (double)p.z * cur_cw[10] + cur_cw[11]);
      const bool in = zr > 0 && zc > 0;
      inliers.push_back(in);
      if (in) out.push_back(p);
    }
    return out;
  }
  void add_new_keyframe(Frame& f, const Mat3& K) {   // src/tracker.cpp:182-235
    f.clear_observations();
    f.extract_observations(*fp_);
    auto prev_kf = map_->get_last_keyframe();
    const auto good = fp_->find_matches(prev_kf->desc, f.desc, p_.lowes_distance_ratio);
    std::vector<Point2f> pr, pc;
    for (const auto& m : good) { pr.push_back({prev_kf->kps[m.query_idx].x, prev_kf->kps[m.query_idx].y}); pc.push_back({f.kps[m.train_idx].x, f.kps[m.train_idx].y}); }
    std::vector<uint8_t> chir;
    const auto pts3d = triangulate_points(affine_inv(prev_kf->pose_wc), affine_inv(f.pose_wc), K, pr, pc, chir);
    size_t k = 0;
    for (size_t i = 0; i < good.size(); i++)   // sequential: a later match overwrites (Appendix B #7)
      if (chir[i]) {
        const Point3f& p3d = pts3d[k++];
        const long lid = prev_kf->landmark_id[good[i].query_idx];
        if (lid != -1) f.landmark_id[good[i].train_idx] = lid;
        else {
          const Landmark lm = map_->new_landmark(p3d, f.desc[good[i].train_idx]);
          map_->add_landmark(lm);
          f.landmark_id[good[i].train_idx] = lm.id;
          prev_kf->landmark_id[good[i].query_idx] = lm.id;   // NOT added to landmark_id_to_index (Appendix B #5)
        }
      }
    map_->add_keyframe(map_->new_keyframe(f));
    tracking_count_from_keyframe_ = 0;
    last.n_keypoints = (int)f.size(); last.n_matches = (int)good.size(); last.n_triangulated = (int)pts3d.size(); last.keyframe_added = true;
  }
  std::optional<Affine3d> update(const Frame& frame, const Mat3& K, const double d[5]) {   // src/tracker.cpp:274-333
    if (state_ == TrackerState::LOST) return std::nullopt;
    if (state_ == TrackerState::INITIALIZING) { prev_frame_ = frame; state_ = TrackerState::TRACKING; return std::nullopt; }
    Frame nf = track_frame_with_optical_flow(frame.image);
    last = Last{};
    last.n_tracked = (int)nf.size();
    if ((long)nf.size() < p_.min_tracked_points) { state_ = TrackerState::LOST; return std::nullopt; }
    std::vector<Point2f> p2; std::vector<Point3f> p3;
    map_->get_observation_to_landmark_point_correspondences(nf, p2, p3);
    double rvec[3], tvec[3];
    last.pnp_ran = true;
    last.pnp_ok = fp_->backend->solve_pnp_ransac(p3, p2, K, d, rvec, tvec, last.n_pnp_inliers);
    if (!last.pnp_ok) {
      // the reference ignores the return value and reads an uninitialised rvec (undefined pose); defined here as in
      // include/mvo.h (MVO_STEP_PNP_FAILED): no pose for the frame, the count advances, the survivors carry on
      tracking_count_from_keyframe_++;
      prev_frame_ = std::move(nf);
      return std::nullopt;
    }
    nf.pose_wc = affine_inv(affine(rodrigues(rvec), tvec));
    tracking_count_from_keyframe_++;
    if (should_add_keyframe(nf) && has_parallax(nf)) add_new_keyframe(nf, K);
    prev_frame_ = std::move(nf);
    return prev_frame_.pose_wc;
  }

 private:
  Map::Ptr map_;
  FeatureProcessor::Ptr fp_;
  TrackerParams p_;
  TrackerState state_ = TrackerState::INITIALIZING;
  Frame prev_frame_;
  long tracking_count_from_keyframe_ = 0;
};

// ---- the dispatch of MonoVO::image_callback (src/mono_vo.cpp:83-131) without ROS -----------------------------------------------------
class VisualOdometry {
 public:
  VisualOdometry(std::shared_ptr<Backend> backend, const Mat3& K, const double d[5], int nfeatures = 1000)
      : K_(K), map(std::make_shared<Map>()), fp(std::make_shared<FeatureProcessor>(std::move(backend), nfeatures)), initializer(map, fp), tracker(map, fp) {
    for (int i = 0; i < 5; i++) d_[i] = d ? d[i] : 0.0;
  }
  std::optional<Affine3d> process(const Image& image) {
    Frame frame(image);
    if (!initializer.is_initalized()) {
      auto ref = initializer.try_initializing(frame, K_);
      if (ref) { tracker.update(*ref, K_, d_); last_pose = affine_identity(); tracking_valid = true; }
      return std::nullopt;
    }
    auto pose = tracker.update(frame, K_, d_);
    if (tracker.get_state() == TrackerState::LOST) tracking_valid = false;
    else if (pose) { last_pose = *pose; tracking_valid = true; }
    return pose;
  }
  Mat3 K_;
  double d_[5];
  Map::Ptr map;
  FeatureProcessor::Ptr fp;
  Initializer initializer;
  Tracker tracker;
  Affine3d last_pose = affine_identity();
  bool tracking_valid = false;
};

// cv -> ROS (REP-103) position of src/utils.cpp:85-121: p_ros = M p_cv with M = [[0,0,1],[-1,0,0],[0,-1,0]]
inline std::array<double, 3> position_cv_to_ros(const Affine3d& pose_wc) { return {pose_wc[11], -pose_wc[3], -pose_wc[7]}; }

}  // namespace mono_vo
