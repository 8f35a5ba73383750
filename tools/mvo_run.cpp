// tools/mvo_run.cpp — the ROS-free harness of the path in C++ (SURVEY.md 8(b) "what calls it"): raw mono8 / bgr8 frames
// through the reference-shaped classes of include/mono_vo_hip.hpp (= the dispatch of MonoVO::image_callback,
// src/mono_vo.cpp:83-131) on the HIP stages, one line per frame in the format of `python -m ros2_mono_vo_amd.mvo_run`.
//   mvo_run --raw FILE --width W --height H [--channels 1|3] [--frames N] [--nfeatures NF] [--intrinsics FX FY CX CY]
//   mvo_run --raw FILE --width W --height H --batch B --frames N
//       frame-batch mode (mono_vo::BatchTracker): B streams, stream s = frames s .. of the file, seeded on its first frame with
//       every key-point on the plane z = 10; one line per stream and frame, then path / cloud / pose of the device's output side
// Build (see __graft_entry__.build): g++ -std=c++17 -O2 -Iinclude tools/mvo_run.cpp -Lros2_mono_vo_amd -lmvo_hip
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mono_vo_hip.hpp"

int main(int argc, char** argv) {
  std::string raw;
  int W = 640, H = 480, ch = 1, frames = 0, nf = 1000, batch = 0;
  double fx = 0, fy = 0, cx = 0, cy = 0;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto need = [&](int n) { if (i + n >= argc) { std::fprintf(stderr, "%s needs %d value(s)\n", a.c_str(), n); std::exit(2); } };
    if (a == "--raw") { need(1); raw = argv[++i]; }
    else if (a == "--width") { need(1); W = std::atoi(argv[++i]); }
    else if (a == "--height") { need(1); H = std::atoi(argv[++i]); }
    else if (a == "--channels") { need(1); ch = std::atoi(argv[++i]); }
    else if (a == "--frames") { need(1); frames = std::atoi(argv[++i]); }
    else if (a == "--nfeatures") { need(1); nf = std::atoi(argv[++i]); }
    else if (a == "--batch") { need(1); batch = std::atoi(argv[++i]); }
    else if (a == "--intrinsics") { need(4); fx = std::atof(argv[++i]); fy = std::atof(argv[++i]); cx = std::atof(argv[++i]); cy = std::atof(argv[++i]); }
    else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  if (raw.empty()) { std::fprintf(stderr, "a frame source is required: --raw FILE --width W --height H\n"); return 2; }
  if (fx == 0) { fx = fy = 0.9 * W; cx = W / 2.0; cy = H / 2.0; }
  const mono_vo::Mat3 K = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
  const double d[5] = {0, 0, 0, 0, 0};
  std::FILE* f = std::fopen(raw.c_str(), "rb");
  if (!f) { std::perror(raw.c_str()); return 1; }
  mvo_config cfg;
  mvo_config_default(&cfg);
  cfg.max_width = W; cfg.max_height = H; cfg.nfeatures = nf; cfg.max_points = 4096;
  try {
    if (batch > 0) {
      // ---- frame-batch mode -----------------------------------------------------------------------------------------------
      if (ch != 1 || frames <= batch) { std::fprintf(stderr, "--batch needs mono8 frames and --frames > batch\n"); return 2; }
      std::vector<uint8_t> all((size_t)W * H * frames);
      if (std::fread(all.data(), 1, all.size(), f) != all.size()) { std::fprintf(stderr, "%s: fewer than %d frames\n", raw.c_str(), frames); return 1; }
      const int steps = frames - batch;   // stream s sees frames s .. s + steps
      mono_vo::BatchTracker bt(W, H, batch, nf, 4096, steps + 1);
      bt.set_intrinsics(K, d);
      bt.enable_output(32768, steps + 1);
      for (int s = 0; s < batch; s++)
        for (int k = 0; k <= steps; k++) bt.preload_frame(s, k, mono_vo::Image{all.data() + (size_t)(s + k) * W * H, W, H, W, 1});
      const std::vector<int> nk = bt.seed(0);
      for (int s = 0; s < batch; s++) {
        const auto xy = bt.tracks(s);
        std::vector<mono_vo::Point3f> lm(xy.size());
        for (size_t i = 0; i < xy.size(); i++) lm[i] = {(float)((xy[i].x - cx) / fx * 10.0), (float)((xy[i].y - cy) / fy * 10.0), 10.f};
        bt.set_landmarks(s, lm);
        std::printf("seed   slot %d  keypoints %d\n", s, nk[s]);
      }
      for (int k = 1; k <= steps; k++) {
        const auto r = bt.track(k);
        for (int s = 0; s < batch; s++)
          std::printf("step %3d slot %d  state=%d flags=%u prev=%d tracked=%d pnp=%d/%d h=%d f=%d kp=%d m=%d tri=%d tracks=%d count=%d  r=(%+.6f,%+.6f,%+.6f) t=(%+.6f,%+.6f,%+.6f)\n",
                      k, s, r[s].state, r[s].flags, r[s].n_prev, r[s].n_tracked, r[s].pnp_ok, r[s].n_pnp_inliers, r[s].score_h, r[s].score_f, r[s].n_keypoints,
                      r[s].n_matches, r[s].n_triangulated, r[s].n_tracks, r[s].tracking_count, r[s].rvec[0], r[s].rvec[1], r[s].rvec[2], r[s].tvec[0],
                      r[s].tvec[1], r[s].tvec[2]);
      }
      const auto odo = bt.odometry();
      for (int s = 0; s < batch; s++) {
        const auto cloud = bt.pointcloud(s);
        const auto path = bt.path(s);
        double cs = 0;
        for (const auto& p : cloud) cs += (double)p.x + 2.0 * (double)p.y + 3.0 * (double)p.z;
        std::printf("output slot %d  valid=%d path=%zu cloud=%zu checksum=%.3f  p_ros=(%+.6f,%+.6f,%+.6f) q=(%+.6f,%+.6f,%+.6f,%+.6f)\n", s, odo[s].tracking_valid,
                    path.size(), cloud.size(), cs, odo[s].position[0], odo[s].position[1], odo[s].position[2], odo[s].orientation[0], odo[s].orientation[1],
                    odo[s].orientation[2], odo[s].orientation[3]);
      }
      std::fclose(f);
      return 0;
    }
    auto backend = std::make_shared<mono_vo::Backend>(cfg);
    mono_vo::VisualOdometry odo(backend, K, d, nf);
    const size_t nbytes = (size_t)W * H * ch;
    std::vector<uint8_t> buf(nbytes);
    int poses = 0;
    static const char* init_names[] = {"OBTAINING_REF", "INITIALIZING", "INITIALIZED"};
    static const char* trk_names[] = {"INITIALIZING", "TRACKING", "LOST"};
    for (int k = 0; frames == 0 || k < frames; k++) {
      const size_t got = std::fread(buf.data(), 1, nbytes, f);
      if (got == 0) break;
      if (got != nbytes) { std::fprintf(stderr, "%s: frame %d is truncated (%zu of %zu bytes)\n", raw.c_str(), k, got, nbytes); return 1; }
      const mono_vo::Image im{buf.data(), W, H, W * ch, ch};
      const auto t0 = std::chrono::steady_clock::now();
      const auto pose = odo.process(im);
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      std::printf("frame %5d  init=%-12s tracker=%-12s %7.2f ms", k, init_names[(int)odo.initializer.state()], trk_names[(int)odo.tracker.get_state()], ms);
      if (pose) {
        const auto p = mono_vo::position_cv_to_ros(*pose);
        std::printf("  tracked=%4d pnp_inliers=%4d  p_ros=(%+.4f,%+.4f,%+.4f)", odo.tracker.last.n_tracked, odo.tracker.last.n_pnp_inliers, p[0], p[1], p[2]);
        poses++;
      }
      std::printf("\n");
      std::fflush(stdout);
    }
    std::printf("key-frames %zu  landmarks %zu  poses %d\n", odo.map->keyframes.size(), odo.map->landmarks.size(), poses);
  } catch (const mono_vo::Error& e) {
    std::fprintf(stderr, "mvo error %d: %s\n", e.code, e.what());
    return 1;
  }
  std::fclose(f);
  return 0;
}
