// tests/cxx/track_over_oracle.cpp - TEST INFRASTRUCTURE ONLY.  One camera stream through include/mono_vo_hip.hpp's Tracker with
// the oracle shim underneath (tests/cxx/mvo_oracle_shim.cpp), seeded like mvo_batch_seed + mvo_batch_set_landmarks (ORB on the
// first frame, every key-point a landmark from the renderer's depth, that frame the first key-frame): prints, per frame, the
// integer fields of mvo_step_result in the order of tests/golden/track_v1.json.
//   track_over_oracle FRAMES.raw DEPTH0.f32 W H N
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mono_vo_hip.hpp"

int main(int argc, char** argv) {
  if (argc != 6) { std::fprintf(stderr, "usage: %s frames.raw depth0.f32 W H N\n", argv[0]); return 2; }
  const int W = std::atoi(argv[3]), H = std::atoi(argv[4]), N = std::atoi(argv[5]);
  std::vector<uint8_t> frames((size_t)W * H * N);
  std::vector<float> depth((size_t)W * H);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(frames.data(), 1, frames.size(), f) != frames.size()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  std::fclose(f);
  f = std::fopen(argv[2], "rb");
  if (!f || std::fread(depth.data(), 4, depth.size(), f) != depth.size()) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
  std::fclose(f);
  using namespace mono_vo;
  mvo_config cfg;
  mvo_config_default(&cfg);
  cfg.max_width = W; cfg.max_height = H; cfg.nfeatures = 1000; cfg.max_points = 8192;
  const Mat3 K = {0.9 * W, 0, W / 2.0, 0, 0.9 * W, H / 2.0, 0, 0, 1};   // synth.default_K
  const double d[5] = {0, 0, 0, 0, 0};
  auto backend = std::make_shared<Backend>(cfg);
  auto map = std::make_shared<Map>();
  auto fp = std::make_shared<FeatureProcessor>(backend, 1000);
  Tracker tracker(map, fp);
  auto image = [&](int k) { return Image{frames.data() + (size_t)k * W * H, W, H, W, 1}; };
  // the Initializer's hand-over with every observation carrying a landmark (tests/track_ref.py seed)
  Frame f0(image(0));
  f0.extract_observations(*fp);
  const float fx = (float)K[0], fy = (float)K[4], cx = (float)K[2], cy = (float)K[5];
  for (size_t i = 0; i < f0.kps.size(); i++) {
    const float x = f0.kps[i].x, y = f0.kps[i].y;
    int xi = (int)std::nearbyint(x), yi = (int)std::nearbyint(y);
    xi = xi < 0 ? 0 : (xi > W - 1 ? W - 1 : xi); yi = yi < 0 ? 0 : (yi > H - 1 ? H - 1 : yi);
    float z = depth[(size_t)yi * W + xi];
    if (!std::isfinite(z)) z = 10.0f;
    const Point3f p{(x - cx) / fx * z, (y - cy) / fy * z, z};
    Landmark lm = map->new_landmark(p, f0.desc[i]);
    map->add_landmark(lm);
    f0.landmark_id[i] = lm.id;
  }
  map->add_keyframe(map->new_keyframe(f0));
  tracker.update(f0, K, d);   // INITIALIZING -> TRACKING with prev_frame_ = the seed frame
  for (int k = 1; k < N; k++) {
    int n_prev = 0;
    for (long l : tracker.prev_frame().landmark_id) n_prev += l != -1;
    int state = 0, flags = 0, n_tracks = 0;
    Tracker::Last L{};
    if (tracker.get_state() == TrackerState::LOST) { state = MVO_TRACK_LOST; n_prev = 0; }
    else {
      const auto pose = tracker.update(Frame(image(k)), K, d);
      L = tracker.last;
      if (tracker.get_state() == TrackerState::LOST) { state = MVO_TRACK_LOST; flags |= MVO_STEP_LOST_NOW; }
      if (pose) flags |= MVO_STEP_POSE;
      if (L.pnp_ran && !L.pnp_ok) flags |= MVO_STEP_PNP_FAILED;
      if (L.kf_checked) flags |= MVO_STEP_KF_CHECKED;
      if (L.keyframe_added) flags |= MVO_STEP_KEYFRAME;
      if (state == MVO_TRACK_TRACKING) for (long l : tracker.prev_frame().landmark_id) n_tracks += l != -1;
    }
    std::printf("%d %d %d %d %d %d %d %d %d %d %d %ld %d\n", n_prev, L.n_tracked, (int)L.pnp_ok, L.n_pnp_inliers, L.score_h, L.score_f, L.n_keypoints,
                L.n_matches, L.n_triangulated, state, flags, tracker.tracking_count_from_keyframe(), n_tracks);
  }
  return 0;
}
