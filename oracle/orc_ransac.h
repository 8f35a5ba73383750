// oracle/orc_ransac.h — TEST INFRASTRUCTURE ONLY (see orc_common.h).  PARITY UNPINNED.
//
// RANSACPointSetRegistrator::run (calib3d/src/ptsetreg.cpp, OpenCV 4.6) restated per SURVEY.md A.4:
// RNG((uint64)-1) fresh per call, getSubset with duplicate-index redraws and checkSubset retries,
// "strictly better" consensus update and RANSACUpdateNumIters adaptive stopping.
#pragma once
#include "orc_common.h"

namespace orc {

static inline int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
  p = std::max(p, 0.); p = std::min(p, 1.);
  ep = std::max(ep, 0.); ep = std::min(ep, 1.);
  double num = std::max(1. - p, DBL_MIN);
  double denom = 1. - std::pow(1. - ep, modelPoints);
  if (denom < DBL_MIN) return 0;
  num = std::log(num);
  denom = std::log(denom);
  return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : cv_round(num / denom);
}

// modelutils: haveCollinearPoints<float> — only the LAST point is tested against earlier pairs.
static inline bool have_collinear_points(const float* pts, int count) {
  int i = count - 1;
  for (int j = 0; j < i; j++) {
    double dx1 = pts[2 * j] - pts[2 * i];
    double dy1 = pts[2 * j + 1] - pts[2 * i + 1];
    for (int k = 0; k < j; k++) {
      double dx2 = pts[2 * k] - pts[2 * i];
      double dy2 = pts[2 * k + 1] - pts[2 * i + 1];
      if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2)))
        return true;
    }
  }
  return false;
}

// Callback: points are rows of d1 / d2 scalars (float for H/F/PnP, double for E).
template <typename T>
struct RansacCbT {
  int d1 = 2, d2 = 2, model_size = 9;
  virtual ~RansacCbT() {}
  int max_models = 3;
  virtual int run_kernel(const T* m1, const T* m2, int count, double* models) const = 0;  // <= max_models
  virtual void compute_error(const T* m1, const T* m2, int count, const double* model, float* err) const = 0;
  virtual bool check_subset(const T*, const T*, int) const { return true; }
};
typedef RansacCbT<float> RansacCb;

struct RansacStats { int iters_run = 0, niters_final = 0, hyp_models = 0; };

// Returns true when a model was found; mask (0/1) and best model filled.
template <typename T>
static inline bool ransac_run(const RansacCbT<T>& cb, const T* m1, const T* m2, int count, int modelPoints,
                              double threshold, double confidence, int maxIters, double* bestModel,
                              unsigned char* bestMaskOut, RansacStats* stats = nullptr) {
  int niters = std::max(maxIters, 1);
  int maxGoodCount = 0;
  RNG rng((uint64_t)-1);
  if (count < modelPoints) return false;
  std::vector<unsigned char> mask(count), bestMask(count, 0);
  std::vector<float> err(count);
  std::vector<double> model((size_t)cb.max_models * cb.model_size);
  if (count == modelPoints) {
    if (cb.run_kernel(m1, m2, count, model.data()) <= 0) return false;
    memcpy(bestModel, model.data(), sizeof(double) * cb.model_size);
    memset(bestMaskOut, 1, count);
    return true;
  }
  std::vector<T> ms1((size_t)modelPoints * cb.d1), ms2((size_t)modelPoints * cb.d2);
  std::vector<int> idx(modelPoints);
  const float t = (float)(threshold * threshold);
  int iter;
  for (iter = 0; iter < niters; iter++) {
    // getSubset(m1, m2, ms1, ms2, rng, 10000)
    bool found = false;
    for (int attempt = 0; attempt < 10000; ++attempt) {
      int i;
      for (i = 0; i < modelPoints; ++i) {
        int idx_i;
        for (idx_i = rng.uniform(0, count); std::find(idx.begin(), idx.begin() + i, idx_i) != idx.begin() + i;
             idx_i = rng.uniform(0, count)) {}
        idx[i] = idx_i;
        for (int k = 0; k < cb.d1; k++) ms1[i * cb.d1 + k] = m1[(size_t)idx_i * cb.d1 + k];
        for (int k = 0; k < cb.d2; k++) ms2[i * cb.d2 + k] = m2[(size_t)idx_i * cb.d2 + k];
      }
      if (cb.check_subset(ms1.data(), ms2.data(), i)) { found = true; break; }
    }
    if (!found) {
      if (iter == 0) return false;
      break;
    }
    int nmodels = cb.run_kernel(ms1.data(), ms2.data(), modelPoints, model.data());
    if (nmodels <= 0) continue;
    for (int i = 0; i < nmodels; i++) {
      const double* model_i = model.data() + (size_t)i * cb.model_size;
      cb.compute_error(m1, m2, count, model_i, err.data());
      int good = 0;
      for (int k = 0; k < count; k++) {
        int f = err[k] <= t;
        mask[k] = (unsigned char)f;
        good += f;
      }
      if (stats) stats->hyp_models++;
      if (good > std::max(maxGoodCount, modelPoints - 1)) {
        std::swap(mask, bestMask);
        memcpy(bestModel, model_i, sizeof(double) * cb.model_size);
        maxGoodCount = good;
        niters = ransac_update_num_iters(confidence, (double)(count - good) / count, modelPoints, niters);
      }
    }
  }
  if (stats) { stats->iters_run = iter; stats->niters_final = niters; }
  if (maxGoodCount > 0) {
    memcpy(bestMaskOut, bestMask.data(), count);
    return true;
  }
  return false;
}

// LMeDSPointSetRegistrator::run (ptsetreg.cpp): fixed iteration count from the assumed outlier ratio 0.45, the model
// with the smallest MEDIAN error wins (strictly smaller), then inliers = err <= sigma^2 with
// sigma = 2.5 * 1.4826 * (1 + 5 / (count - modelPoints)) * sqrt(minMedian), floored at 0.001.
// OpenCV takes the median as element count/2 after std::nth_element over the float bits: an order statistic,
// so any selection gives the same value.
template <typename T>
static inline bool lmeds_run(const RansacCbT<T>& cb, const T* m1, const T* m2, int count, int modelPoints, double confidence,
                             int maxIters, double* bestModel, unsigned char* bestMaskOut, RansacStats* stats = nullptr) {
  const double outlierRatio = 0.45;
  double minMedian = DBL_MAX;
  RNG rng((uint64_t)-1);
  if (count < modelPoints) return false;
  std::vector<float> err(count), srt(count);
  std::vector<double> model((size_t)cb.max_models * cb.model_size);
  if (count == modelPoints) {
    if (cb.run_kernel(m1, m2, count, model.data()) <= 0) return false;
    memcpy(bestModel, model.data(), sizeof(double) * cb.model_size);
    memset(bestMaskOut, 1, count);
    return true;
  }
  int niters = ransac_update_num_iters(confidence, outlierRatio, modelPoints, maxIters);
  niters = std::max(niters, 3);
  std::vector<T> ms1((size_t)modelPoints * cb.d1), ms2((size_t)modelPoints * cb.d2);
  std::vector<int> idx(modelPoints);
  int iter;
  for (iter = 0; iter < niters; iter++) {
    bool found = false;
    for (int attempt = 0; attempt < 10000; ++attempt) {
      int i;
      for (i = 0; i < modelPoints; ++i) {
        int idx_i;
        for (idx_i = rng.uniform(0, count); std::find(idx.begin(), idx.begin() + i, idx_i) != idx.begin() + i;
             idx_i = rng.uniform(0, count)) {}
        idx[i] = idx_i;
        for (int k = 0; k < cb.d1; k++) ms1[i * cb.d1 + k] = m1[(size_t)idx_i * cb.d1 + k];
        for (int k = 0; k < cb.d2; k++) ms2[i * cb.d2 + k] = m2[(size_t)idx_i * cb.d2 + k];
      }
      if (cb.check_subset(ms1.data(), ms2.data(), i)) { found = true; break; }
    }
    if (!found) {
      if (iter == 0) return false;
      break;
    }
    int nmodels = cb.run_kernel(ms1.data(), ms2.data(), modelPoints, model.data());
    if (nmodels <= 0) continue;
    for (int i = 0; i < nmodels; i++) {
      const double* model_i = model.data() + (size_t)i * cb.model_size;
      cb.compute_error(m1, m2, count, model_i, err.data());
      srt = err;
      std::nth_element(srt.begin(), srt.begin() + count / 2, srt.end());
      double median = srt[count / 2];
      if (stats) stats->hyp_models++;
      if (median < minMedian) {
        minMedian = median;
        memcpy(bestModel, model_i, sizeof(double) * cb.model_size);
      }
    }
  }
  if (stats) { stats->iters_run = iter; stats->niters_final = niters; }
  if (minMedian < DBL_MAX) {
    double sigma = 2.5 * 1.4826 * (1 + 5. / (count - modelPoints)) * std::sqrt(minMedian);
    sigma = std::max(sigma, 0.001);
    cb.compute_error(m1, m2, count, bestModel, err.data());
    const float t = (float)(sigma * sigma);
    int good = 0;
    for (int k = 0; k < count; k++) { int f = err[k] <= t; bestMaskOut[k] = (unsigned char)f; good += f; }
    return good >= modelPoints;
  }
  return false;
}

}  // namespace orc
