// include/mvo_shim.hpp — header-only C++17 shim: the reference's FeatureProcessor surface and the cv:: calls inside
// Tracker / Initializer, re-expressed over the C ABI of include/mvo.h.  Compiled only where OpenCV is present
// (the reference's own build); nothing in this repository depends on it.  See INTEGRATION.md.
//
// Each wrapper keeps the reference's argument meaning and error behaviour:
//   * OpenCV throws cv::Exception on bad shapes -> MVO_E_ARG is re-thrown as cv::Exception;
//   * degenerate estimators return empty / zero results, exactly as cv:: does (MVO_E_DEGENERATE).
#pragma once
#include <opencv2/opencv.hpp>

#include <stdexcept>
#include <vector>

#include "mvo.h"

namespace mvo_shim {

inline void check(mvo_ctx* ctx, int rc, bool allow_degenerate = false) {
  if (rc == MVO_OK || (allow_degenerate && rc == MVO_E_DEGENERATE)) return;
  CV_Error(cv::Error::StsError, std::string("mvo: ") + mvo_last_error(ctx));
}

// Replaces cv::ORB::detectAndCompute at reference src/feature_processor.cpp:19-23.
inline void detect_and_compute(mvo_ctx* ctx, const cv::Mat& image, std::vector<cv::KeyPoint>& keypoints, cv::Mat& descriptors) {
  CV_Assert(image.type() == CV_8UC1 || image.type() == CV_8UC3);
  static_assert(sizeof(mvo_keypoint) == sizeof(cv::KeyPoint), "cv::KeyPoint layout");
  // staging buffers live as long as the thread: no 16384-entry allocation per frame (cv::ORB returns at most a little over
  // nfeatures key-points; MVO_E_CAPACITY would say so if a context were configured for more than this)
  constexpr int cap = 16384;
  thread_local std::vector<mvo_keypoint> kp_buf(cap);
  thread_local std::vector<uchar> desc_buf((size_t)cap * 32);
  int n = 0;
  check(ctx, mvo_orb_detect_and_compute(ctx, image.data, image.cols, image.rows, (int)image.step, image.channels(), kp_buf.data(),
                                        desc_buf.data(), cap, &n));
  keypoints.resize(n);
  descriptors.create(n, 32, CV_8U);
  for (int i = 0; i < n; i++) {
    const mvo_keypoint& k = kp_buf[i];
    keypoints[i] = cv::KeyPoint{{k.x, k.y}, k.size, k.angle, k.response, k.octave, k.class_id};
    for (int b = 0; b < 32; b++) descriptors.ptr<uchar>(i)[b] = desc_buf[(size_t)i * 32 + b];
  }
}

// Replaces matcher_.knnMatch + the Lowe loop at reference src/feature_processor.cpp:25-41.
inline std::vector<cv::DMatch> find_matches(mvo_ctx* ctx, const cv::Mat& d1, const cv::Mat& d2, double ratio) {
  static_assert(sizeof(mvo_match) == sizeof(cv::DMatch), "cv::DMatch layout");
  std::vector<cv::DMatch> out(std::max(d1.rows, 1));
  int n = 0;
  check(ctx, mvo_match_knn2_ratio(ctx, d1.data, d1.rows, d2.data, d2.rows, ratio, reinterpret_cast<mvo_match*>(out.data()),
                                  (int)out.size(), &n));
  out.resize(n);
  return out;
}

// Replaces cv::calcOpticalFlowPyrLK at reference src/tracker.cpp:68-69.
inline void calc_optical_flow_pyr_lk(mvo_ctx* ctx, const cv::Mat& prev, const cv::Mat& next, const std::vector<cv::Point2f>& prev_pts,
                                     std::vector<cv::Point2f>& next_pts, std::vector<uchar>& status, std::vector<float>& err) {
  const int n = (int)prev_pts.size();
  next_pts.resize(n); status.resize(n); err.resize(n);
  check(ctx, mvo_lk_track(ctx, prev.data, next.data, prev.cols, prev.rows, (int)prev.step, prev.channels(),
                          reinterpret_cast<const float*>(prev_pts.data()), n, reinterpret_cast<float*>(next_pts.data()), status.data(),
                          err.data()));
}

// Replaces cv::findHomography(pts1, pts2, cv::RANSAC, thr, mask) at reference src/tracker.cpp:243, src/initializer.cpp:82.
inline int find_homography_inliers(mvo_ctx* ctx, const std::vector<cv::Point2f>& p1, const std::vector<cv::Point2f>& p2, double thr,
                                   std::vector<uchar>& mask) {
  mask.assign(p1.size(), 0);
  int n_inl = 0;
  double H[9];
  check(ctx, mvo_find_homography_ransac(ctx, reinterpret_cast<const float*>(p1.data()), reinterpret_cast<const float*>(p2.data()),
                                        (int)p1.size(), thr, 2000, 0.995, mask.data(), H, &n_inl), true);
  return n_inl;
}

// Replaces cv::findFundamentalMat(pts1, pts2, cv::FM_RANSAC, thr, 0.99, mask) at src/tracker.cpp:248, src/initializer.cpp:87.
inline int find_fundamental_inliers(mvo_ctx* ctx, const std::vector<cv::Point2f>& p1, const std::vector<cv::Point2f>& p2, double thr,
                                    std::vector<uchar>& mask) {
  mask.assign(p1.size(), 0);
  int n_inl = 0;
  double F[9];
  check(ctx, mvo_find_fundamental_ransac(ctx, reinterpret_cast<const float*>(p1.data()), reinterpret_cast<const float*>(p2.data()),
                                         (int)p1.size(), thr, 0.99, 1000, mask.data(), F, &n_inl), true);
  return n_inl;
}

// Replaces cv::solvePnPRansac(..., false, 100, 8.0, 0.99, inliers) at reference src/tracker.cpp:309.
inline bool solve_pnp_ransac(mvo_ctx* ctx, const std::vector<cv::Point3f>& obj, const std::vector<cv::Point2f>& img, const cv::Mat& K,
                             const cv::Mat& d, cv::Mat& rvec, cv::Mat& tvec, cv::Mat& inliers) {
  cv::Mat Kd, dd;
  K.convertTo(Kd, CV_64F);
  d.reshape(1, 1).convertTo(dd, CV_64F);
  double d5[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < std::min(5, dd.cols); i++) d5[i] = dd.at<double>(i);
  std::vector<int> idx(obj.size());
  int n_inl = 0;
  rvec.create(3, 1, CV_64F); tvec.create(3, 1, CV_64F);
  int rc = mvo_solve_pnp_ransac(ctx, reinterpret_cast<const float*>(obj.data()), reinterpret_cast<const float*>(img.data()), (int)obj.size(),
                                Kd.ptr<double>(), d5, 100, 8.0f, 0.99, rvec.ptr<double>(), tvec.ptr<double>(), idx.data(), &n_inl);
  check(ctx, rc, true);
  inliers = cv::Mat(n_inl, 1, CV_32S);
  for (int i = 0; i < n_inl; i++) inliers.at<int>(i) = idx[i];
  return rc == MVO_OK;
}

// Replaces cv::triangulatePoints + cv::convertPointsFromHomogeneous at src/tracker.cpp:149-152, src/initializer.cpp:125-131.
inline std::vector<cv::Point3f> triangulate(mvo_ctx* ctx, const cv::Mat& P1, const cv::Mat& P2, const std::vector<cv::Point2f>& p1,
                                            const std::vector<cv::Point2f>& p2) {
  cv::Mat A, B;
  P1.convertTo(A, CV_64F); P2.convertTo(B, CV_64F);
  A = A.clone(); B = B.clone();
  std::vector<cv::Point3f> X(p1.size());
  check(ctx, mvo_triangulate(ctx, A.ptr<double>(), B.ptr<double>(), reinterpret_cast<const float*>(p1.data()),
                             reinterpret_cast<const float*>(p2.data()), (int)p1.size(), reinterpret_cast<float*>(X.data())));
  return X;
}

// Replaces cv::findEssentialMat(p1, p2, K, cv::RANSAC, prob, thr, mask) at reference src/initializer.cpp:228-229.
// Returns the 3x3 essential matrix (empty when no model was found, like OpenCV); mask gets one byte per point.
inline cv::Mat find_essential_mat(mvo_ctx* ctx, const std::vector<cv::Point2f>& p1, const std::vector<cv::Point2f>& p2, const cv::Mat& K,
                                  double prob, double thr, std::vector<uchar>& mask) {
  cv::Mat Kd;
  K.convertTo(Kd, CV_64F); Kd = Kd.clone();
  mask.assign(p1.size(), 0);
  cv::Mat E(3, 3, CV_64F);
  int n_inl = 0;
  int rc = mvo_find_essential_ransac(ctx, reinterpret_cast<const float*>(p1.data()), reinterpret_cast<const float*>(p2.data()), (int)p1.size(),
                                     Kd.ptr<double>(), prob, thr, 1000, mask.data(), E.ptr<double>(), &n_inl);
  check(ctx, rc, true);
  return rc == MVO_OK ? E : cv::Mat();
}

// Replaces cv::recoverPose(E, p1, p2, K, R, t, mask) at reference src/initializer.cpp:236.
inline int recover_pose(mvo_ctx* ctx, const cv::Mat& E, const std::vector<cv::Point2f>& p1, const std::vector<cv::Point2f>& p2, const cv::Mat& K,
                        cv::Mat& R, cv::Mat& t, std::vector<uchar>& mask) {
  cv::Mat Ed, Kd;
  E.convertTo(Ed, CV_64F); K.convertTo(Kd, CV_64F);
  Ed = Ed.clone(); Kd = Kd.clone();
  R.create(3, 3, CV_64F); t.create(3, 1, CV_64F);
  int good = 0;
  check(ctx, mvo_recover_pose(ctx, Ed.ptr<double>(), reinterpret_cast<const float*>(p1.data()), reinterpret_cast<const float*>(p2.data()),
                              (int)p1.size(), Kd.ptr<double>(), R.ptr<double>(), t.ptr<double>(), mask.empty() ? nullptr : mask.data(), &good));
  return good;
}

}  // namespace mvo_shim
