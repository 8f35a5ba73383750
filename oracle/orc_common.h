// oracle/orc_common.h — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement ("oracle") of the OpenCV-4.6 semantics that Tatsuya-2/ros2_mono_vo
// relies on for its per-frame VO front-end.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this code; the product (libmvo_hip.so) never does.
//
// PARITY UNPINNED: the reference ships no tests / golden vectors and OpenCV itself is not
// available offline, so this restatement is pinned only by first-principles known-answer
// tests and planted-ground-truth geometry tests (tests/test_oracle_*.py).
//
// Shared helpers: OpenCV rounding rules, cv::RNG, border interpolation.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cfloat>
#include <vector>
#include <algorithm>

namespace orc {

typedef unsigned char u8;

// cvRound: round-half-to-even (lrint under the default FE_TONEAREST mode), cvFloor, cvCeil.
static inline int cv_round(double v) { return (int)std::lrint(v); }
static inline int cv_round(float v) { return (int)std::lrintf(v); }
static inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_floor(float v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

// BORDER_REFLECT_101 (cv::borderInterpolate): ... 2 1 | 0 1 2 ... n-2 n-1 | n-2 n-3 ...
static inline int reflect101(int p, int len) {
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    if (p < 0) p = -p;
    else p = 2 * (len - 1) - p;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

// cv::RNG — 64-bit multiply-with-carry (modules/core/include/opencv2/core/operations.hpp).
struct RNG {
  uint64_t state;
  explicit RNG(uint64_t seed = 0xffffffffULL) : state(seed ? seed : 0xffffffffULL) {}
  inline unsigned next() {
    state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
    return (unsigned)state;
  }
  inline int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

}  // namespace orc
