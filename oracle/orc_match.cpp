// oracle/orc_match.cpp — TEST INFRASTRUCTURE ONLY (see orc_common.h header).  PARITY UNPINNED.
//
// CPU restatement of FeatureProcessor::find_matches (reference src/feature_processor.cpp:25-41):
// cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, k=2) followed by Lowe's ratio test.  Follows
// SURVEY.md Appendix A.2 (core/src/batch_distance.cpp: K-insertion with strict comparisons, so ties
// keep the lower train index first; features2d/src/matchers.cpp: invalid neighbours dropped).
#include "orc_common.h"
#include "mvo_oracle.h"
#include <climits>

extern "C" int orc_match_knn2_ratio(const unsigned char* q, int nq, const unsigned char* t, int nt,
                                    double ratio, orc_match* out, int cap) {
  int n = 0;
  if (nq <= 0 || nt <= 0) return 0;
  for (int i = 0; i < nq; i++) {
    int dist[2] = {INT_MAX, INT_MAX}, idx[2] = {-1, -1};
    const uint64_t* a = (const uint64_t*)(q + (size_t)i * 32);
    for (int j = 0; j < nt; j++) {
      uint64_t b[4];
      memcpy(b, t + (size_t)j * 32, 32);
      uint64_t a4[4];
      memcpy(a4, a, 32);
      int d = __builtin_popcountll(a4[0] ^ b[0]) + __builtin_popcountll(a4[1] ^ b[1]) +
              __builtin_popcountll(a4[2] ^ b[2]) + __builtin_popcountll(a4[3] ^ b[3]);
      if (d < dist[1]) {
        int k;
        for (k = 0; k >= 0 && dist[k] > d; k--) { idx[k + 1] = idx[k]; dist[k + 1] = dist[k]; }
        idx[k + 1] = j; dist[k + 1] = d;
      }
    }
    // knnMatch emits only valid neighbours; find_matches requires exactly 2.
    if (idx[0] < 0 || idx[1] < 0) continue;
    float d0 = (float)dist[0], d1 = (float)dist[1];
    if (d0 < ratio * d1) {
      if (n < cap) out[n] = {i, idx[0], 0, d0};
      n++;
    }
  }
  return n;
}
