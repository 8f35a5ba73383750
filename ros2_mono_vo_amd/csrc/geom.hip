// csrc/geom.hip — batched RANSAC engine and two-view / PnP geometry for gfx950.
//
// Replaces, with OpenCV-4.6 semantics:
//   cv::findHomography(RANSAC) / cv::findFundamentalMat(FM_RANSAC)   reference src/tracker.cpp:243,248,
//                                                                     src/initializer.cpp:82,87
//   cv::solvePnPRansac + cv::Rodrigues                               reference src/tracker.cpp:309-316
//   cv::triangulatePoints + convertPointsFromHomogeneous             reference src/tracker.cpp:149-152
//   cv::recoverPose                                                  reference src/initializer.cpp:236
//
// ransac_kernel<Model>: ONE WAVEFRONT PER PROBLEM (camera stream), so a batch of B streams is B single-wave
// workgroups in one launch.  OpenCV's loop is sequential (adaptive iteration count, "strictly better"
// update, data-dependent RNG consumption), so each round does
//   1. lane 0 draws 64 candidate samples from cv::RNG((uint64)-1) in OpenCV's order,
//   2. 64 lanes run checkSubset in parallel; an ordered ballot compaction turns passing candidates into
//      RANSAC iterations (a failing candidate is exactly OpenCV's "retry with the next draws"),
//   3. one hypothesis per lane: minimal solver (4-pt H / 7-pt F / 5-pt EPnP) in private memory,
//   4. every lane scores its own hypothesis over all points (uniform point loads): integer inlier counts,
//   5. lane 0 replays the hypotheses in order applying the consensus update and RANSACUpdateNumIters,
// and stops as soon as iter >= niters — the speculative tail of a round is simply discarded, which is
// unobservable because nothing after the loop reads the RNG.  The consensus mask is then recomputed for
// the winning model by all lanes.  No MFMA: nothing here is a dense contraction.
#include "mvo_internal.h"
#include "geom_models.h"
#include "track_policy.h"

#include <cstdlib>

struct RansacArgs {
  const float* m1;
  const float* m2;
  int stride1, stride2;  // floats between slots
  const int* n;
  double thr, conf;
  int max_iters;
  int cap;  // capacity per slot: counts are clamped to it so a corrupt count can never run away
  ModelParams P;
  u8* mask;
  int mask_stride;
  double* model;  // [B][16]
  int* result;    // [B][8]: ok, n_inliers, iters_run, niters_final, models_scored, [5] (idx != null) entries of the index list
  int* idx;       // [B][idx_stride] ordered list of the consensus set (solvePnPRansac's `inliers`), or null
  int idx_stride;
  // Slot queue (null: workgroup b solves slot b).  The launch is `grid` persistent workgroups that claim slots 0 .. nslots - 1
  // from *work_ctr (zeroed before the launch) and skip the ones without correspondences.  A workgroup is placed only where its
  // LDS fits (H: 138 KB - an empty CU), so on a step where few or no slots need the model a launch of one workgroup per slot
  // spent a millisecond or more just getting its empty workgroups placed and retired (trace: profiles/r03_c_step_timeline.txt).
  int* work_ctr;
  int nslots;
};

// Candidate samples per round = M::CH (16 for H / F / PnP): their solvers keep the dense matrices in a per-lane LDS
// workspace, and 16 lanes of it are 25-37 KB per stream.  RANSAC on this path usually stops after a handful of
// iterations (adaptive niters), so a round of 16 is rarely repeated; the other 48 lanes help with scoring.
// Threads per problem.  One wavefront: the solvers need 200-400 VGPRs, and a 4-wave workgroup parked three idle waves
// of that size on every SIMD of its CU, starving the image kernels (ORB) that run beside it on the main stream.  With
// one wave per stream a launch occupies one SIMD per stream and scoring is a per-lane loop over uniform point loads.
#define RS_T 64
// Register budget: 2 waves per SIMD (<= 256 VGPRs).  The solvers would take up to 420 VGPRs, and one such wave per SIMD
// pushes the image kernels that share the CU (ORB's FAST runs beside the RANSAC chains) down to 1-2 waves per SIMD;
// the chains have slack, the main stream does not.
#ifndef RS_WAVES_PER_EU
#define RS_WAVES_PER_EU 2
#endif
#define RS_NW (RS_T / 64)

// NW wavefronts per problem: every wave solves M::CH hypotheses of a round in its own LDS workspaces, so a round is NW
// times as wide at the same latency.  The easy case (consensus after a handful of iterations) costs the same as with one
// wave; a hard stream - findHomography under true parallax needs hundreds to 2000 iterations - finishes NW times sooner,
// and the batch launch lasts as long as its hardest stream.  Candidate c of a round belongs to wave c / CPW, lane c % CPW.
// The consensus mask as solvePnPRansac's ordered inlier list, by the workgroup that wrote the mask (a separate launch for
// this cost 0.3 ms of stream time per step beside other contexts' kernels).
template <int RS_TT>
__device__ inline void ransac_mask_to_indices(const u8* mask, int count, int* idx, int* s_wave /* [RS_TT / 64] */, int* s_base, int* result) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NWV = RS_TT / 64;
  __syncthreads();   // the mask is complete
  if (tid == 0) *s_base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < count; i0 += RS_TT) {
    const int i = i0 + tid;
    const bool keep = i < count && mask[i];
    const unsigned long long m = __ballot(keep);
    const int pre = __popcll(m & ((1ull << lane) - 1));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int off = *s_base;
    for (int w = 0; w < wave; w++) off += s_wave[w];
    if (keep) idx[off + pre] = i;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < NWV; w++) t += s_wave[w];
      *s_base += t;
    }
    __syncthreads();
  }
  if (tid == 0) result[5] = *s_base;
}

template <class M, int NW>
__device__ __forceinline__ void ransac_slot(const RansacArgs& A, const int slot) {
  constexpr int RS_TT = 64 * NW;
  static_assert(NW == 1 || !M::WIDE, "wide (private-memory) rounds are a single-wave mode");
  // Round width.  The first round (and any round with few iterations left) solves CH0 = M::CH hypotheses, whose dense
  // matrices live in LDS; when more than that many iterations are still owed, the round is 64 wide and lanes >= CH0 keep
  // their matrices in private memory (slower per solve, four times the hypotheses per round) - the tail of a hard
  // problem otherwise gates the whole batch launch.
  constexpr int CH0 = M::CH, RS_CH = (M::WIDE ? 64 : M::CH) * NW;
  // per-lane workspace stride: == 1 (mod 32) doubles, so lane-uniform 8-byte accesses of 16 lanes fall into distinct banks
  constexpr int WSS = M::WS > 0 ? ((M::WS + 30) / 32) * 32 + 1 : 1;
  __shared__ double s_ws[(M::WS > 0 ? CH0 * NW : 1) * WSS];
  // Candidates of a refill / queue of the passing ones still to be solved.  M::OVERDRAW: a refill draws and checks one
  // candidate per THREAD (RS_TT of them) instead of one per solver lane, the passing ones queue up in draw order and a
  // round solves the first RS_CH of the queue: when checkSubset rejects most samples (findHomography on a scene a
  // homography explains only partly: ~70 % rejected) the solver lanes stay full instead of a round solving what
  // happened to pass of its own RS_CH draws.  The candidate ORDER - all that OpenCV's sequential loop observes - is
  // unchanged: passing candidates are consumed first in, first out.
  constexpr int RS_DR = M::OVERDRAW ? RS_TT : RS_CH;
  constexpr int RS_Q = M::OVERDRAW ? RS_CH + RS_DR : RS_CH;
  __shared__ int s_att[RS_DR][M::MP];
  __shared__ int s_idx[RS_Q][M::MP];
  __shared__ int s_qn;   // passing candidates queued
  __shared__ int s_lwave[NW], s_lbase;   // ordered inlier list (A.idx)
  int* const out_idx = A.idx ? A.idx + (size_t)slot * A.idx_stride : nullptr;
  __shared__ double s_models[RS_CH][M::MAXM][M::MS];
  __shared__ int s_nmodels[RS_CH];
  __shared__ int s_cnt[RS_CH][M::MAXM];
  __shared__ double s_best[M::MS];
  __shared__ int s_ctl[8];  // 0: npass, 1: done, 2: maxGood, 3: iter, 4: niters, 5: consec_fail, 6: ok, 7: models scored
  __shared__ int s_wpass[NW];
  __shared__ unsigned long long s_wmask[NW];
  __shared__ unsigned long long s_rng;
  __shared__ double s_minmed;

  // one latency-bound wavefront per stream next to VALU-saturating image kernels: without priority it only gets a
  // round-robin share of the SIMD's issue slots and a hard stream's chain (5 x the instructions) gates the step
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int count = min(max(A.n[slot], 0), A.cap);
  const float* m1 = A.m1 + (size_t)slot * A.stride1;
  const float* m2 = A.m2 + (size_t)slot * A.stride2;
  u8* mask = A.mask + (size_t)slot * A.mask_stride;
  double* out_model = A.model + (size_t)slot * 16;
  int* result = A.result + (size_t)slot * 8;
  const float t = (float)(A.thr * A.thr);

  if constexpr (M::MP_ALT > 0) if (count == M::MP_ALT) {
    // solvePnPRansac with four correspondences: model_points = 4 == npoints -> solvePnP(SOLVEPNP_P3P) on all of them,
    // every point an inlier, no RANSAC (calib3d/src/solvepnp.cpp)
    if (tid == 0) {
      float ms1[M::MP * M::PT1], ms2[M::MP * M::PT2];
      for (int i = 0; i < M::MP_ALT * M::PT1; i++) ms1[i] = m1[i];
      for (int i = 0; i < M::MP_ALT * M::PT2; i++) ms2[i] = m2[i];
      double models[M::MS];
      const int nm = M::solve_alt(A.P, ms1, ms2, models);
      s_ctl[6] = nm > 0;
      if (nm > 0)
        for (int k = 0; k < M::MS; k++) out_model[k] = models[k];
    }
    __syncthreads();
    const int ok = s_ctl[6];
    for (int i = tid; i < count; i += RS_TT) mask[i] = ok ? 1 : 0;
    if (tid == 0) { result[0] = ok; result[1] = ok ? count : 0; result[2] = 1; result[3] = 1; result[4] = 1; }
    if (out_idx) ransac_mask_to_indices<RS_TT>(mask, count, out_idx, s_lwave, &s_lbase, result);
    return;
  }
  if (count < M::MP) {
    for (int i = tid; i < count; i += RS_TT) mask[i] = 0;
    if (tid == 0) { result[0] = 0; result[1] = 0; result[2] = 0; result[3] = 0; result[4] = 0; if (out_idx) result[5] = 0; }
    return;
  }
  // LMeDSPointSetRegistrator instead of RANSAC (findFundamentalMat below 15 points): fixed iteration count from the
  // assumed outlier ratio 0.45, smallest median error wins, inliers by the robust sigma.  Same sampling machinery.
  const bool lmeds = M::LMEDS_BELOW > 0 && count > M::MP && count < M::LMEDS_BELOW;
  if (tid == 0) {
    s_rng = 0xFFFFFFFFFFFFFFFFULL;
    s_ctl[1] = 0; s_ctl[2] = 0; s_ctl[3] = 0; s_ctl[4] = A.max_iters > 1 ? A.max_iters : 1; s_ctl[5] = 0; s_ctl[6] = 0; s_ctl[7] = 0;
    s_qn = 0;
    if (lmeds) { int ni = gl_ransac_update_num_iters(A.conf, 0.45, M::MP, 1000); s_ctl[4] = ni > 3 ? ni : 3; }
    s_minmed = DBL_MAX;
  }
  __syncthreads();

  if (count == M::MP) {
    // a single runKernel on all points, mask all ones
    if (tid == 0) {
      float ms1[M::MP * M::PT1], ms2[M::MP * M::PT2];
      for (int i = 0; i < M::MP * M::PT1; i++) ms1[i] = m1[i];
      for (int i = 0; i < M::MP * M::PT2; i++) ms2[i] = m2[i];
      double models[M::MAXM * M::MS];
      int nm = M::solve(A.P, ms1, ms2, models, s_ws);
      s_ctl[6] = nm > 0;
      if (nm > 0)
        for (int k = 0; k < M::MS; k++) out_model[k] = models[k];
    }
    __syncthreads();
    int ok = s_ctl[6];
    for (int i = tid; i < count; i += RS_TT) mask[i] = ok ? 1 : 0;
    if (tid == 0) { result[0] = ok; result[1] = ok ? count : 0; result[2] = 1; result[3] = 1; result[4] = 1; }
    if (out_idx) ransac_mask_to_indices<RS_TT>(mask, count, out_idx, s_lwave, &s_lbase, result);
    return;
  }

#ifdef RS_TIMING
  long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tq = wall_clock64();
#define RS_TICK(k) { long long tn = wall_clock64(); tm[k] += tn - tq; tq = tn; }
#else
#define RS_TICK(k)
#endif
  for (;;) {
    const int ch = NW > 1 ? RS_CH : ((M::WS > 0 && s_ctl[4] - s_ctl[3] <= CH0) || (M::WS > 0 && s_ctl[3] == 0) ? CH0 : RS_CH);  // uniform
    const int cpw = ch / NW;                                  // hypotheses per wave
    const int cand = lane < cpw ? wave * cpw + lane : -1;     // this thread's hypothesis of the round
    const int want = min(ch, s_ctl[4] - s_ctl[3]);            // hypotheses the round can consume (uniform)
    float ms1[M::MP * M::PT1], ms2[M::MP * M::PT2];
    // refill until the round is full (checkSubset may reject most of a refill); every refill's candidates are accounted for in
    // draw order, and a refill only starts below `want` <= RS_CH queued candidates, so the queue cannot overflow
    for (int rf = 0; rf < (M::OVERDRAW ? 4 : 1) && s_qn < want && !s_ctl[1]; rf++) {   // uniform
      // ---- 1. candidate samples, OpenCV's getSubset draw order (sequential RNG stream, lane 0) ----------
      const int ndraw = M::OVERDRAW ? RS_DR : ch;
      const int q0 = s_qn;
      if (tid == 0) {
        GlRng rng(s_rng);
        for (int a = 0; a < ndraw; a++) {
          for (int i = 0; i < M::MP; ++i) {
            int idx_i;
            for (;;) {
              idx_i = rng.uniform(0, count);
              bool dup = false;
              for (int k = 0; k < i; k++) dup |= (s_att[a][k] == idx_i);
              if (!dup) break;
            }
            s_att[a][i] = idx_i;
          }
        }
        s_rng = rng.state;
      }
      __syncthreads();
      RS_TICK(0)
      // ---- 2. checkSubset in parallel (one candidate per lane / per thread) + ordered compaction over the waves -------
      const int dcand = M::OVERDRAW ? tid : cand;   // candidate d of the refill belongs to wave d / CPW, lane d % CPW
      bool pass = false;
      if (dcand >= 0) {
        for (int i = 0; i < M::MP; i++) {
          int id = s_att[dcand][i];
          for (int k = 0; k < M::PT1; k++) ms1[i * M::PT1 + k] = m1[(size_t)id * M::PT1 + k];
          for (int k = 0; k < M::PT2; k++) ms2[i * M::PT2 + k] = m2[(size_t)id * M::PT2 + k];
        }
        pass = M::check_subset(ms1, ms2);
      }
      const unsigned long long bm = __ballot(pass);
      if (lane == 0) { s_wpass[wave] = __popcll(bm); s_wmask[wave] = bm; }
      __syncthreads();
      int pos = q0 + __popcll(bm & ((1ull << lane) - 1));
      for (int w = 0; w < wave; w++) pos += s_wpass[w];
      if (pass)
        for (int i = 0; i < M::MP; i++) s_idx[pos][i] = s_att[dcand][i];
      if (tid == 0) {
        // OpenCV gives up on an iteration after 10000 consecutive failing attempts: walk the pass bits in order
        int np = 0, run = s_ctl[5];
        bool abort_ = false;
        const int CPW = M::OVERDRAW ? 64 : cpw;  // candidates held by a wave's ballot
        for (int w = 0; w < NW; w++) {
          unsigned long long m = s_wmask[w];
          np += __popcll(m);
          if (m == 0) { run += CPW; abort_ |= run >= 10000; }
          else {
            int lead = __ffsll((long long)m) - 1;
            abort_ |= (run + lead) >= 10000;
            run = __clzll(m) - (64 - CPW);  // failing candidates after the last passing one
          }
        }
        s_qn = q0 + np;
        s_ctl[5] = run;
        if (abort_) s_ctl[1] = 1;
      }
      __syncthreads();
    }
    if (tid < RS_CH) {
      s_nmodels[tid] = 0;
      for (int k = 0; k < M::MAXM; k++) s_cnt[tid][k] = 0;
    }
    __syncthreads();
    if (s_ctl[1]) break;
    const int npass = s_qn;
    RS_TICK(1)
    // ---- 3. minimal solver, one hypothesis per lane; only as many as can still be consumed ---------------
    const int nsolve = min(npass, want);
    int nm = 0;
    if constexpr (M::LANES > 1) {
      // M::LANES adjacent lanes per hypothesis (EPnP: one beta approximation each); hypothesis h on lanes LANES h ..
      static_assert(NW == 1 && !M::WIDE && M::WS > 0 && M::CH * M::LANES <= 64, "cooperative solves are a single-wave mode");
      const int hyp = lane / M::LANES, sub = lane - hyp * M::LANES;
      if (hyp < ch && hyp < nsolve) {
        for (int i = 0; i < M::MP; i++) {
          int id = s_idx[hyp][i];
          for (int k = 0; k < M::PT1; k++) ms1[i * M::PT1 + k] = m1[(size_t)id * M::PT1 + k];
          for (int k = 0; k < M::PT2; k++) ms2[i * M::PT2 + k] = m2[(size_t)id * M::PT2 + k];
        }
        double models[M::MAXM * M::MS];
        nm = M::solve_coop(A.P, ms1, ms2, models, s_ws + hyp * WSS, sub);
        if (sub == 0) {
          s_nmodels[hyp] = nm;
          for (int k = 0; k < M::MS; k++) s_models[hyp][0][k] = models[k];
        }
      }
    } else
    if (cand >= 0 && cand < nsolve) {
      for (int i = 0; i < M::MP; i++) {
        int id = s_idx[cand][i];
        for (int k = 0; k < M::PT1; k++) ms1[i * M::PT1 + k] = m1[(size_t)id * M::PT1 + k];
        for (int k = 0; k < M::PT2; k++) ms2[i * M::PT2 + k] = m2[(size_t)id * M::PT2 + k];
      }
      double models[M::MAXM * M::MS];
      double priv[M::WS > 0 && M::WIDE ? M::WS : 1];  // lanes >= CH0 of a wide round
      nm = M::solve(A.P, ms1, ms2, models, M::WS > 0 ? (!M::WIDE || cand < CH0 ? s_ws + cand * WSS : priv) : s_ws);
      if (nm < 0) nm = 0;
      if (nm > M::MAXM) nm = M::MAXM;
      s_nmodels[cand] = nm;
      for (int q = 0; q < nm; q++)
        for (int k = 0; k < M::MS; k++) s_models[cand][q][k] = models[q * M::MS + k];
    }
    __syncthreads();
    RS_TICK(2)
    // ---- 4 + 5 (M::SEQ_SCORE): score and replay hypothesis by hypothesis --------------------------------------------------
    // solvePnPRansac's loop usually ends after a handful of iterations (RANSACUpdateNumIters on the first good model), so
    // scoring all of a round's hypotheses first - each over every point - is mostly wasted: 188 of 963 us for one stream of
    // 2000 correspondences (RS_TIMING).  Here all threads score ONE hypothesis, lane 0 applies OpenCV's update, and the loop
    // stops where OpenCV's does.  Same counts, same order, same result.
    if constexpr (M::SEQ_SCORE) {
      static_assert(NW == 1 && M::MAXM == 1 && M::LMEDS_BELOW == 0, "sequential scoring is a single-wave, single-model mode");
      for (int h = 0; h < nsolve; h++) {
        if (s_ctl[3] >= s_ctl[4]) break;   // iter >= niters (uniform: read after the fence below)
        if (s_nmodels[h] > 0) {
          typename M::Scorer sc;
          sc.init(A.P, &s_models[h][0][0]);
          int good = 0;
#pragma unroll 1
          for (int i = tid; i < count; i += RS_TT) good += sc.err(m1 + (size_t)i * M::PT1, m2 + (size_t)i * M::PT2) <= t;
          if (good) atomicAdd(&s_cnt[h][0], good);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (tid == 0) {
          int maxGood = s_ctl[2], niters = s_ctl[4];
          if (s_nmodels[h] > 0) {
            const int good = s_cnt[h][0];
            s_ctl[7]++;
            if (good > max(maxGood, M::MP - 1)) {
              for (int k = 0; k < M::MS; k++) s_best[k] = s_models[h][0][k];
              maxGood = good;
              niters = gl_ransac_update_num_iters(A.conf, (double)(count - good) / count, M::MP, niters);
            }
          }
          s_ctl[2] = maxGood; s_ctl[4] = niters; s_ctl[3]++;
          if (s_ctl[3] >= niters) s_ctl[1] = 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();
      RS_TICK(3)
    } else {
    // ---- 4. scoring: hypothesis = tid % RS_CH, point partition = tid / RS_CH; integer inlier counts ------------
      {
        const int parts = RS_TT / ch, hyp = tid % ch, part = tid / ch;
        if (lmeds) {
          if (part == 0 && hyp < nsolve) {
            const int nmh = s_nmodels[hyp];
            for (int q = 0; q < nmh; q++) {
              typename M::Scorer sc;
              sc.init(A.P, &s_models[hyp][q][0]);
              // median = element count/2 of the sorted errors (count < 15: insertion sort in registers / private memory)
              float e[16];
  #pragma unroll 1
              for (int i = 0; i < count; i++) {
                float v = sc.err(m1 + (size_t)i * M::PT1, m2 + (size_t)i * M::PT2);
                int k = i;
                while (k > 0 && e[k - 1] > v) { e[k] = e[k - 1]; --k; }
                e[k] = v;
              }
              s_cnt[hyp][q] = __float_as_int(e[count / 2]);
            }
          }
        } else if (hyp < nsolve && part < parts) {   // lanes past parts * ch (ch not a divisor of 64) sit the scoring out
          const int nmh = s_nmodels[hyp];
          for (int q = 0; q < nmh; q++) {
            typename M::Scorer sc;
            sc.init(A.P, &s_models[hyp][q][0]);
            int good = 0;
  #pragma unroll 1
            for (int i = part; i < count; i += parts) good += sc.err(m1 + (size_t)i * M::PT1, m2 + (size_t)i * M::PT2) <= t;
            atomicAdd(&s_cnt[hyp][q], good);
          }
        }
      }
      __syncthreads();
      RS_TICK(3)
      // ---- 5. ordered replay of OpenCV's consensus update -------------------------------------------------------
      if (tid == 0) {
        int maxGood = s_ctl[2], iter = s_ctl[3], niters = s_ctl[4], scored = s_ctl[7];
        for (int h = 0; h < nsolve && iter < niters; h++, iter++) {
          int nmh = s_nmodels[h];
          for (int q = 0; q < nmh; q++) {
            int good = s_cnt[h][q];
            scored++;
            if (lmeds) {
              const double median = (double)__int_as_float(good);
              if (median < s_minmed) {
                s_minmed = median;
                for (int k = 0; k < M::MS; k++) s_best[k] = s_models[h][q][k];
              }
            } else if (good > max(maxGood, M::MP - 1)) {
              for (int k = 0; k < M::MS; k++) s_best[k] = s_models[h][q][k];
              maxGood = good;
              niters = gl_ransac_update_num_iters(A.conf, (double)(count - good) / count, M::MP, niters);
            }
          }
        }
        s_ctl[2] = maxGood; s_ctl[3] = iter; s_ctl[4] = niters; s_ctl[7] = scored;
        if (iter >= niters) s_ctl[1] = 1;
      }
      __syncthreads();
      RS_TICK(4)
    }
    if (s_ctl[1]) break;
    if (M::OVERDRAW) {   // the candidates solved leave the queue, the others move up (read, barrier, write: ranges overlap)
      const int left = npass - nsolve;
      int keep[RS_Q / RS_TT + 1][M::MP];
      for (int e = tid, j = 0; e < left; e += RS_TT, j++)
        for (int i = 0; i < M::MP; i++) keep[j][i] = s_idx[nsolve + e][i];
      __syncthreads();
      for (int e = tid, j = 0; e < left; e += RS_TT, j++)
        for (int i = 0; i < M::MP; i++) s_idx[e][i] = keep[j][i];
      if (tid == 0) s_qn = left;
      __syncthreads();
    } else {   // a round without over-draw consumes what it drew (or the loop has ended)
      if (tid == 0) s_qn = 0;
      if (NW > 1) __syncthreads();   // the other wavefronts read s_qn at the top of the next round (uniform: NW is a template constant)
    }
  }
  // ---- consensus mask of the winning model ---------------------------------------------------------------------
  if (lmeds) {
    const double minMedian = s_minmed;
    if (minMedian < DBL_MAX) {
      double sigma = 2.5 * 1.4826 * (1 + 5. / (count - M::MP)) * sqrt(minMedian);
      sigma = sigma > 0.001 ? sigma : 0.001;
      const float ts = (float)(sigma * sigma);
      typename M::Scorer sc;
      sc.init(A.P, s_best);
      int good = 0;
#pragma unroll 1
      for (int i = tid; i < count; i += RS_TT) {
        int f = sc.err(m1 + (size_t)i * M::PT1, m2 + (size_t)i * M::PT2) <= ts ? 1 : 0;
        mask[i] = (u8)f;
        good += f;
      }
      if (good) atomicAdd(&s_ctl[2], good);
      if (tid < M::MS) out_model[tid] = s_best[tid];
      __syncthreads();
      good = s_ctl[2];
      if (good < M::MP) for (int i = tid; i < count; i += RS_TT) mask[i] = 0;   // findFundamentalMat returns an empty Mat
      if (tid == 0) { result[0] = good >= M::MP; result[1] = good; result[2] = s_ctl[3]; result[3] = s_ctl[4]; result[4] = s_ctl[7]; }
    } else {
      for (int i = tid; i < count; i += RS_TT) mask[i] = 0;
      if (tid == 0) { result[0] = 0; result[1] = 0; result[2] = s_ctl[3]; result[3] = s_ctl[4]; result[4] = s_ctl[7]; }
    }
    return;
  }
  const int maxGood = s_ctl[2];
  if (maxGood > 0) {
    typename M::Scorer sc;
    sc.init(A.P, s_best);
#pragma unroll 1
    for (int i = tid; i < count; i += RS_TT) mask[i] = sc.err(m1 + (size_t)i * M::PT1, m2 + (size_t)i * M::PT2) <= t ? 1 : 0;
    if (tid < M::MS) out_model[tid] = s_best[tid];
  } else {
    for (int i = tid; i < count; i += RS_TT) mask[i] = 0;
  }
  if (tid == 0) { result[0] = maxGood > 0; result[1] = maxGood; result[2] = s_ctl[3]; result[3] = s_ctl[4]; result[4] = s_ctl[7]; }
  if (out_idx) ransac_mask_to_indices<RS_TT>(mask, count, out_idx, s_lwave, &s_lbase, result);
#ifdef RS_TIMING
  RS_TICK(5)
  if (tid == 0 && slot < 4)
    printf("RS_TIMING MP=%d slot=%d n=%d iters=%d scored=%d | draw %lld check %lld solve %lld score %lld replay %lld mask %lld (100MHz ticks)\n", M::MP, slot, count,
           s_ctl[3], s_ctl[7], tm[0], tm[1], tm[2], tm[3], tm[4], tm[5]);
#endif
}

template <class M, int NW>
__global__ __launch_bounds__(64 * NW, RS_WAVES_PER_EU) void ransac_kernel(RansacArgs A) {
  if (!A.work_ctr) { ransac_slot<M, NW>(A, blockIdx.x); return; }
  __shared__ int s_claim;
  for (;;) {   // every workgroup leaves once the counter has passed the last slot
    __syncthreads();   // the shared state of the slot just solved is dead
    if (threadIdx.x == 0) s_claim = atomicAdd(A.work_ctr, 1);
    __syncthreads();
    const int slot = s_claim;
    if (slot >= A.nslots) return;
    ransac_slot<M, NW>(A, slot);
  }
}

// ---------------------------------------------------------------------------------------------------
// PnP: consensus set -> ordered inlier list; then cvFindExtrinsicCameraParams2 (DLT / planar init + LM)
// ---------------------------------------------------------------------------------------------------
// Block size of the refine.  Throughput mode (many streams resident) runs one wavefront per stream: the kernel is
// register heavy, and a 4-wave workgroup per stream took every SIMD of the chip for itself while ORB's kernels on the
// main stream stalled behind it.  With few streams nothing competes and the chain LK -> RANSAC -> refine IS the step
// latency, so the sums over the inliers run four wavefronts wide (mvo_ctx decides by the batch size).
#ifndef PR_WAVES_PER_EU
#define PR_WAVES_PER_EU 2
#endif
// deterministic block-wide sum of `K` doubles per thread: butterfly inside each wave (every lane ends with the same
// bits), then the wave partials through LDS in a fixed order.  s_red must hold NW*K doubles.
template <int K, int NW>
__device__ inline void block_sum(double* v, double* s_red, double* out /* [K], valid in all threads */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < K; k++) {
    double x = v[k];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    if (NW == 1) out[k] = x;
    else if (lane == 0) s_red[wave * K + k] = x;
  }
  if (NW == 1) return;
  __syncthreads();
  for (int k = 0; k < K; k++) {
    double t = s_red[k];
    for (int w = 1; w < NW; w++) t += s_red[w * K + k];
    out[k] = t;
  }
  __syncthreads();
}

struct PnpRefineArgs {
  const float* obj;  // [B][maxpts][3]
  const float* img;  // [B][maxpts][2]
  int stride_pts;    // points between slots
  const int* inl;    // [B][maxpts]
  int* result;       // [B][8]: [0] ransac ok, [5] n inliers -> [6] refine status
  const double* model;  // [B][16] RANSAC model (3x2 hconcat)
  const int* n;         // [B] correspondences per slot
  double* pose;      // [B][8]
  CamK cam;
  TrkKeyframePolicy kp;   // kp.state == null: none
};

// ---- wave-cooperative one-sided Jacobi SVD ------------------------------------------------------------------------------
// The dense solves of the refine (12 x 12 DLT null vector, 6 x 6 Levenberg-Marquardt step) used to run on lane 0: 0.32 ms and
// 3 x 0.03 ms of a 0.65 ms kernel for one stream (RS_TIMING), in ~1000 spilled registers.  Here a whole wavefront works on the
// matrix in LDS.  JacobiSVDImpl_ (core/src/lapack.cpp; gl_jacobi_svd) visits the row pairs (i, j) of A^T one after the other;
// pairs that share no row commute, so a sweep is re-ordered into the N - 1 rounds of a round-robin tournament, N / 2 disjoint
// pairs each, and four pairs of a round are rotated AT ONCE, one per DPP row of 16 lanes: lane k of a row owns column k of
// the two rows (their dot product and new norms are DPP row sums, the rotation itself is element-wise), and the scalar part
// of a rotation (hypot, two square roots, the divisions - most of its instructions) is issued once for four pairs.  Same
// rotation formulas, skip test and stopping rule as JacobiSVDImpl_; a different pair order, so singular vectors agree with the
// sequential routine to rounding (the refine's contract is 1e-4 on R, t; measured against the oracle <= 1e-9), not bit for bit.
template <int CTRL>
__device__ __forceinline__ double wj_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a DPP row, bit-identical in every lane of the row (each step adds the two halves of a disjoint split)
__device__ __forceinline__ double wj_row_sum(double v) {
  v += wj_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += wj_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += wj_dpp<0x141>(v);   // row_half_mirror
  v += wj_dpp<0x140>(v);   // row_mirror
  return v;
}
#define WJ_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// At: N x N in LDS, row i = row i of A^T (= column i of A), overwritten by sigma_i * u_i^T; W: N doubles (LDS) -> sigma_i^2;
// Vt: N x N (LDS) -> rows v_i^T.  All 64 lanes of ONE wavefront call it; N <= 16.
template <int N>
__device__ __forceinline__ void wave_jacobi_svd(double* At, double* W, double* Vt, const int lane) {
  constexpr int NP = (N + 1) & ~1;   // players of the tournament (odd N: a dummy whose pairings are byes)
  const double eps = DBL_EPSILON * 10;
  const int row = lane >> 4, k = lane & 15;
  const bool col = k < N;
  for (int e = lane; e < N * N; e += 64) Vt[e] = (e / N == e % N) ? 1.0 : 0.0;
  for (int i = row; i < N; i += 4) {
    const double x = col ? At[i * N + k] : 0.0;
    const double sd = wj_row_sum(x * x);
    if (k == 0) W[i] = sd;
  }
  WJ_FENCE();
  for (int iter = 0; iter < 30; iter++) {
    bool changed = false;
#pragma unroll 1
    for (int r = 0; r < NP - 1; r++) {
#pragma unroll 1
      for (int q0 = 0; q0 < NP / 2; q0 += 4) {
        const int q = q0 + row;
        int i = q == 0 ? NP - 1 : (r + q) % (NP - 1);
        int j = q == 0 ? r : (r - q + (NP - 1)) % (NP - 1);
        if (i > j) { const int t = i; i = j; j = t; }
        if (q < NP / 2 && j < N) {   // uniform over the row
          const double ai = col ? At[i * N + k] : 0.0, aj = col ? At[j * N + k] : 0.0;
          double p = wj_row_sum(ai * aj);
          const double a = W[i], b = W[j];
          if (!(fabs(p) <= eps * sqrt(a * b))) {
            p *= 2;
            const double beta = a - b, gamma = gl_hypot(p, beta);
            double c, s;
            if (beta < 0) {
              const double delta = (gamma - beta) * 0.5;
              s = sqrt(delta / gamma);
              c = p / (gamma * s * 2);
            } else {
              c = sqrt((gamma + beta) / (gamma * 2));
              s = p / (gamma * c * 2);
            }
            const double t0 = c * ai + s * aj, t1 = -s * ai + c * aj;
            const double na = wj_row_sum(t0 * t0), nb = wj_row_sum(t1 * t1);
            if (col) {
              At[i * N + k] = t0; At[j * N + k] = t1;
              const double vi = Vt[i * N + k], vj = Vt[j * N + k];
              Vt[i * N + k] = c * vi + s * vj;
              Vt[j * N + k] = -s * vi + c * vj;
            }
            if (k == 0) { W[i] = na; W[j] = nb; }
            changed = true;
          }
        }
        WJ_FENCE();   // the rows rotated by one DPP row are read by another in a later pass
      }
    }
    if (!__any(changed)) break;
  }
}

// One workgroup per stream: PR_T / 64 wavefronts.  Sums over the inlier set are block reductions in a fixed order (the oracle
// sums sequentially, so R, t agree to rounding, not bit for bit); the dense solves are wave_jacobi_svd on wavefront 0.  The
// accumulators of a pass over the points are at most 28 doubles per thread (the DLT normal matrix takes three passes): no
// scratch, <= 128 VGPRs, so four of these wavefronts share a SIMD with anything.
template <int PR_T>
__device__ __forceinline__ void pnp_refine_slot(const PnpRefineArgs& A) {
  constexpr int PR_NW = PR_T / 64;
  __shared__ double s_red[PR_NW * 28];
  __shared__ double s_sh[64];       // broadcast area: Vt of the planarity test / the pose / dR/dr of an evaluation
  __shared__ double s_mat[2 * 144 + 16];
  __shared__ int s_flag[4];
  __builtin_amdgcn_s_setprio(3);   // see ransac_kernel
  const int slot = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* result = A.result + (size_t)slot * 8;
  double* pose = A.pose + (size_t)slot * 8;
  const int ok = result[0];
  const int count = min(max(result[5], 0), A.stride_pts);
  if (!ok) { if (tid == 0) result[6] = 0; return; }
  if (A.n[slot] == 5 || A.n[slot] == 4) {
    // solvePnPRansac with exactly model_points correspondences returns the EPnP (5) / P3P (4) solution as is
    if (tid == 0) {
      const double* model = A.model + (size_t)slot * 16;
      for (int i = 0; i < 3; i++) { pose[i] = model[2 * i]; pose[3 + i] = model[2 * i + 1]; }
      result[6] = 1;
    }
    return;
  }
  const float* obj = A.obj + (size_t)slot * A.stride_pts * 3;
  const float* img = A.img + (size_t)slot * A.stride_pts * 2;
  const int* inl = A.inl + (size_t)slot * A.stride_pts;
  const CamK cam = A.cam;
  auto ptM = [&](int i, double M[3]) { int id = inl[i]; M[0] = obj[3 * id]; M[1] = obj[3 * id + 1]; M[2] = obj[3 * id + 2]; };
  auto ptm = [&](int i, double m[2]) { int id = inl[i]; m[0] = img[2 * id]; m[1] = img[2 * id + 1]; };

#ifdef RS_TIMING
  long long pt[6] = {0, 0, 0, 0, 0, 0};
  long long pq = wall_clock64();
#define PR_TICK(k) { long long tn = wall_clock64(); pt[k] += tn - pq; pq = tn; }
#else
#define PR_TICK(k)
#endif
  double param[6] = {0, 0, 0, 0, 0, 0};
  // ---- Mc, MM ------------------------------------------------------------------------------------------------
  double Mc[3];
  {
    double v[3] = {0, 0, 0};
    for (int i = tid; i < count; i += PR_T) { double M[3]; ptM(i, M); v[0] += M[0]; v[1] += M[1]; v[2] += M[2]; }
    block_sum<3, PR_NW>(v, s_red, Mc);
    for (int j = 0; j < 3; j++) Mc[j] /= count;
  }
  double MM[9];
  {
    double v[6] = {0, 0, 0, 0, 0, 0};
    for (int i = tid; i < count; i += PR_T) {
      double M[3]; ptM(i, M);
      double d0 = M[0] - Mc[0], d1 = M[1] - Mc[1], d2 = M[2] - Mc[2];
      v[0] += d0 * d0; v[1] += d0 * d1; v[2] += d0 * d2; v[3] += d1 * d1; v[4] += d1 * d2; v[5] += d2 * d2;
    }
    double o[6];
    block_sum<6, PR_NW>(v, s_red, o);
    MM[0] = o[0]; MM[1] = MM[3] = o[1]; MM[2] = MM[6] = o[2]; MM[4] = o[3]; MM[5] = MM[7] = o[4]; MM[8] = o[5];
  }
  // lane 0: planarity test (3x3 SVD), broadcast Vt and the flag
  if (tid == 0) {
    double W[3], V[9];
    gl_svd3(MM, W, nullptr, V);
    bool planar = W[2] / W[1] < 1e-3;
    s_flag[0] = planar;
    if (planar) {
      if (V[2] * V[2] + V[5] * V[5] < 1e-10) { for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1 : 0; }
      if (gl_det3(V) < 0) for (int i = 0; i < 9; i++) V[i] *= -1;
    }
    for (int i = 0; i < 9; i++) s_sh[i] = V[i];
  }
  __syncthreads();
  const bool planar = s_flag[0];
  if (planar) {
    double Rt[9], tt[3];
    for (int i = 0; i < 9; i++) Rt[i] = s_sh[i];
    for (int r = 0; r < 3; r++) tt[r] = -(Rt[r * 3] * Mc[0] + Rt[r * 3 + 1] * Mc[1] + Rt[r * 3 + 2] * Mc[2]);
    __syncthreads();
    auto mxy = [&](int i, double& X, double& Y) {
      double M[3]; ptM(i, M);
      X = Rt[0] * M[0] + Rt[1] * M[1] + Rt[2] * M[2] + tt[0];
      Y = Rt[3] * M[0] + Rt[4] * M[1] + Rt[5] * M[2] + tt[1];
    };
    auto mnorm = [&](int i, double& x, double& y) { double m[2]; ptm(i, m); gm_undistort_point(cam, m[0], m[1], x, y); };
    // normalised DLT homography Mxy -> mn (HomographyEstimatorCallback::runKernel on all points)
    double c4[4];
    {
      double v[4] = {0, 0, 0, 0};
      for (int i = tid; i < count; i += PR_T) { double X, Y, x, y; mxy(i, X, Y); mnorm(i, x, y); v[0] += x; v[1] += y; v[2] += X; v[3] += Y; }
      block_sum<4, PR_NW>(v, s_red, c4);
      for (int k = 0; k < 4; k++) c4[k] /= count;
    }
    double s4[4];
    {
      double v[4] = {0, 0, 0, 0};
      for (int i = tid; i < count; i += PR_T) {
        double X, Y, x, y; mxy(i, X, Y); mnorm(i, x, y);
        v[0] += fabs(x - c4[0]); v[1] += fabs(y - c4[1]); v[2] += fabs(X - c4[2]); v[3] += fabs(Y - c4[3]);
      }
      block_sum<4, PR_NW>(v, s_red, s4);
    }
    bool degenerate = fabs(s4[0]) < DBL_EPSILON || fabs(s4[1]) < DBL_EPSILON || fabs(s4[2]) < DBL_EPSILON || fabs(s4[3]) < DBL_EPSILON;
    double smx = count / s4[0], smy = count / s4[1], sMx = count / s4[2], sMy = count / s4[3];
    // LtL (9 x 9 symmetric, 45 sums) in three passes of 15 accumulators: rows 0-1, 2-4, 5-8 of the upper triangle
    double* L = s_mat;
    if (!degenerate) {
#pragma unroll 1
      for (int pass = 0; pass < 3; pass++) {
        const int j0 = pass == 0 ? 0 : (pass == 1 ? 2 : 5), j1 = pass == 0 ? 2 : (pass == 1 ? 5 : 9);
        double acc[18];
        for (int k = 0; k < 18; k++) acc[k] = 0;
        for (int i = tid; i < count; i += PR_T) {
          double X, Y, x, y; mxy(i, X, Y); mnorm(i, x, y);
          x = (x - c4[0]) * smx; y = (y - c4[1]) * smy; X = (X - c4[2]) * sMx; Y = (Y - c4[3]) * sMy;
          const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
          const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
          int q = 0;
#pragma unroll
          for (int j = 0; j < 9; j++)
#pragma unroll
            for (int k = j; k < 9; k++)
              if (j >= j0 && j < j1) { acc[q] += Lx[j] * Lx[k] + Ly[j] * Ly[k]; q++; }
        }
        double o[18];
        block_sum<18, PR_NW>(acc, s_red, o);
        if (tid == 0) {
          int q = 0;
          for (int j = j0; j < j1; j++)
            for (int k = j; k < 9; k++) { L[j * 9 + k] = o[q]; L[k * 9 + j] = o[q]; q++; }
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      double R[9];
      bool okH = !degenerate;
      double h[9];
      if (okH) {
        double* V = s_mat + 144;
        gl_ldsd* Wl = (gl_ldsd*)(s_mat + 2 * 144);   // 9 of the 16 doubles behind the two matrices
        gl_jacobi_eigen9_lds<false>((gl_ldsd*)L, Wl, (gl_ldsd*)V);
        double invHnorm[9] = {1. / smx, 0, c4[0], 0, 1. / smy, c4[1], 0, 0, 1};
        double Hnorm2[9] = {sMx, 0, -c4[2] * sMx, 0, sMy, -c4[3] * sMy, 0, 0, 1};
        double Htemp[9], H0[9];
        gl_mat3mul(invHnorm, V + 72, Htemp);
        gl_mat3mul(Htemp, Hnorm2, H0);
        double s = 1. / H0[8];
        for (int i = 0; i < 9; i++) { h[i] = H0[i] * s; okH = okH && isfinite(h[i]); }
      }
      if (okH) {
        double h1_norm = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]);
        double h2_norm = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
        double s1 = 1. / fmax(h1_norm, DBL_EPSILON), s2 = 1. / fmax(h2_norm, DBL_EPSILON);
        double s3 = 2. / fmax(h1_norm + h2_norm, DBL_EPSILON);
        double t[3] = {h[2] * s3, h[5] * s3, h[8] * s3};
        h[0] *= s1; h[3] *= s1; h[6] *= s1;
        h[1] *= s2; h[4] *= s2; h[7] *= s2;
        h[2] = h[3] * h[7] - h[6] * h[4];
        h[5] = h[6] * h[1] - h[0] * h[7];
        h[8] = h[0] * h[4] - h[3] * h[1];
        double r[3], Hm[9];
        gm_rodrigues_m2v(h, r);
        gm_rodrigues_v2m(r, Hm, nullptr);
        for (int k = 0; k < 3; k++) param[3 + k] = Hm[k * 3] * tt[0] + Hm[k * 3 + 1] * tt[1] + Hm[k * 3 + 2] * tt[2] + t[k];
        gl_mat3mul(Hm, Rt, R);
      } else {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1 : 0;
        param[3] = param[4] = param[5] = 0;
      }
      gm_rodrigues_m2v(R, param);
      for (int i = 0; i < 6; i++) s_sh[i] = param[i];
      s_flag[1] = 1;
    }
    __syncthreads();
  } else {
    if (count < 6) {
      // "DLT algorithm needs at least 6 points": with exactly 5 inliers OpenCV keeps the minimal-sample model
      if (tid == 0) {
        const double* model = A.model + (size_t)slot * 16;
        if (count == 5) {
          for (int i = 0; i < 3; i++) { pose[i] = model[2 * i]; pose[3 + i] = model[2 * i + 1]; }
          result[6] = 1;
        } else {
          result[6] = 0;
        }
      }
      return;
    }
    // DLT normal matrix L^T L (12 x 12) with rows l0 = [P 0 x P], l1 = [0 P y P], P = [M 1]: of its 78 upper entries
    // block (0..3, 4..7) is zero, block (4..7, 4..7) repeats block (0..3, 0..3) term for term, and the others are sums of
    // P_a P_b, P_a (x P_c), P_a (y P_c) and (x P_a)(x P_b) + (y P_a)(y P_b) - 52 sums, in three passes over the points of
    // at most 20 accumulators each (same products, same order over the points as one pass).
    double* L = s_mat;
#pragma unroll 1
    for (int pass = 0; pass < 3; pass++) {
      double acc[20];
      for (int k = 0; k < 20; k++) acc[k] = 0;
      for (int i = tid; i < count; i += PR_T) {
        double M[3], m[2]; ptM(i, M); ptm(i, m);
        double xu, yu;
        gm_undistort_point(cam, m[0], m[1], xu, yu);
        const double x = -xu, y = -yu;
        const double P[4] = {M[0], M[1], M[2], 1.};
        if (pass == 0) {
          const double xP[4] = {x * M[0], x * M[1], x * M[2], x}, yP[4] = {y * M[0], y * M[1], y * M[2], y};
          int q = 0;
#pragma unroll
          for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = a; b < 4; b++) { acc[q] += P[a] * P[b]; acc[10 + q] += xP[a] * xP[b] + yP[a] * yP[b]; q++; }
        } else {
          const double w = pass == 1 ? x : y;
          const double wP[4] = {w * M[0], w * M[1], w * M[2], w};
#pragma unroll
          for (int a = 0; a < 4; a++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[4 * a + c] += P[a] * wP[c];
        }
      }
      double o[20];
      block_sum<20, PR_NW>(acc, s_red, o);
      if (tid == 0) {
        if (pass == 0) {
          int q = 0;
          for (int a = 0; a < 4; a++)
            for (int b = a; b < 4; b++) {
              L[a * 12 + b] = o[q]; L[b * 12 + a] = o[q];
              L[(4 + a) * 12 + 4 + b] = o[q]; L[(4 + b) * 12 + 4 + a] = o[q];
              L[(8 + a) * 12 + 8 + b] = o[10 + q]; L[(8 + b) * 12 + 8 + a] = o[10 + q];
              q++;
            }
          for (int a = 0; a < 4; a++)
            for (int b = 0; b < 4; b++) { L[a * 12 + 4 + b] = 0; L[(4 + b) * 12 + a] = 0; }
        } else {
          const int r0 = pass == 1 ? 0 : 4;
          for (int a = 0; a < 4; a++)
            for (int c = 0; c < 4; c++) { L[(r0 + a) * 12 + 8 + c] = o[4 * a + c]; L[(8 + c) * 12 + r0 + a] = o[4 * a + c]; }
        }
      }
    }
    __syncthreads();
    PR_TICK(0)
    double* LV = s_mat + 144;
    double* LW = s_mat + 288;
    // cvSVD(&_LL, &_LW, 0, &_LV, MODIFY_A + V_T): the right singular vector of the smallest singular value (L^T = L)
    if (wave == 0) wave_jacobi_svd<12>(L, LW, LV, lane);
    __syncthreads();
    if (tid == 0) {
      int imin = 0;
      for (int i = 1; i < 12; i++) if (LW[i] < LW[imin]) imin = i;
      double* RRt = LV + imin * 12;
      double RR[9], ttv[3];
      for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) RR[r * 3 + c] = RRt[r * 4 + c]; ttv[r] = RRt[r * 4 + 3]; }
      if (gl_det3(RR) < 0) { for (int i = 0; i < 9; i++) RR[i] *= -1; for (int i = 0; i < 3; i++) ttv[i] *= -1; }
      double sc = 0;
      for (int i = 0; i < 9; i++) sc += RR[i] * RR[i];
      sc = sqrt(sc);
      int good = fabs(sc) > DBL_EPSILON;
      if (good) {
        double Wd[3], U[9], Vt[9], R[9];
        gl_svd3(RR, Wd, U, Vt);
        gl_mat3mul(U, Vt, R);
        double nR = 0;
        for (int i = 0; i < 9; i++) nR += R[i] * R[i];
        nR = sqrt(nR);
        for (int k = 0; k < 3; k++) param[3 + k] = ttv[k] * (nR / sc);
        gm_rodrigues_m2v(R, param);
      }
      for (int i = 0; i < 6; i++) s_sh[i] = param[i];
      s_flag[1] = good;
    }
    __syncthreads();
  }
  if (!s_flag[1]) { if (tid == 0) result[6] = 0; return; }
  for (int i = 0; i < 6; i++) param[i] = s_sh[i];
  __syncthreads();
  PR_TICK(1)

  // ---- CvLevMarq (J + err mode), state machine replicated by every lane, the 6 x 6 solve by wavefront 0 ---------------------
  enum { DONE = 0, STARTED = 1, CALC_J = 2, CHECK_ERR = 3 };
  int state = STARTED, iters = 0, lambdaLg10 = -3;
  double prevParam[6], prevErrNorm = DBL_MAX, errNorm = 0, curErr2 = 0;
  const int max_iter = 20;
  const double epsilon = FLT_EPSILON;
  // JtJ (upper triangle, 21) and JtErr (6) of the last CALC_J evaluation stay in LDS: s_mat[200 .. 227)
  double* JJ = s_mat + 200;
  auto step = [&]() {
    // cvSolve(JtJ + lambda diag(JtJ), JtErr, DECOMP_SVD): x = sum over sigma_i > threshold of (u_i . b / sigma_i) v_i
    double* Am = s_mat;         // 6 x 6 -> sigma_i u_i^T
    double* Vm = s_mat + 36;    // 6 x 6
    double* Wm = s_mat + 72;    // sigma_i^2
    if (wave == 0) {
      const double LOG10 = log(10.);
      const double lambda = exp(lambdaLg10 * LOG10);
      if (lane < 36) {
        const int a = lane / 6, b = lane % 6;
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        double v = JJ[lo * 6 - lo * (lo - 1) / 2 + (hi - lo)];   // entry (lo, hi) of the packed upper triangle
        if (a == b) v *= 1. + lambda;
        Am[lane] = v;
      }
      WJ_FENCE();
      wave_jacobi_svd<6>(Am, Wm, Vm, lane);
      if (lane == 0) {
        double sig[6], thr = 0;
        for (int i = 0; i < 6; i++) { sig[i] = sqrt(Wm[i]); thr += sig[i]; }
        thr *= DBL_EPSILON * 2;
        double x[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 6; i++) {
          if (fabs(sig[i]) <= thr) continue;
          double d = 0;
          for (int j = 0; j < 6; j++) d += Am[i * 6 + j] * JJ[21 + j];
          d /= Wm[i];                         // (sigma_i u_i . b) / sigma_i^2
          for (int j = 0; j < 6; j++) x[j] += d * Vm[i * 6 + j];
        }
        for (int i = 0; i < 6; i++) s_sh[i] = prevParam[i] - x[i];
      }
    }
    __syncthreads();
    for (int i = 0; i < 6; i++) param[i] = s_sh[i];
    __syncthreads();
    PR_TICK(2)
  };
  auto eval = [&](bool withJ) {
    // R and dR/dr once per evaluation, through LDS (27 + 9 doubles are not worth 72 registers in every lane)
    if (tid == 0) gm_rodrigues_v2m(param, s_sh + 8, withJ ? s_sh + 17 : nullptr);
    __syncthreads();
    double R[9];
    for (int i = 0; i < 9; i++) R[i] = s_sh[8 + i];
    const double* dRdr = s_sh + 17;
    double v[28];
    for (int k = 0; k < 28; k++) v[k] = 0;
    for (int i = tid; i < count; i += PR_T) {
      double M[3], m[2], mm[2], dpdr[6], dpdt[6];
      ptM(i, M); ptm(i, m);
      gm_project_point(R, withJ ? dRdr : nullptr, param + 3, cam, M, mm, withJ ? dpdr : nullptr, withJ ? dpdt : nullptr);
      double e0 = mm[0] - m[0], e1 = mm[1] - m[1];
      v[27] += e0 * e0 + e1 * e1;
      if (withJ) {
        double j0[6] = {dpdr[0], dpdr[1], dpdr[2], dpdt[0], dpdt[1], dpdt[2]};
        double j1[6] = {dpdr[3], dpdr[4], dpdr[5], dpdt[3], dpdt[4], dpdt[5]};
        int q = 0;
        for (int a = 0; a < 6; a++)
          for (int b = a; b < 6; b++) v[q++] += j0[a] * j0[b] + j1[a] * j1[b];
        for (int a = 0; a < 6; a++) v[21 + a] += j0[a] * e0 + j1[a] * e1;
      }
    }
    double o[28];
    if (withJ) {
      block_sum<28, PR_NW>(v, s_red, o);
      if (tid == 0) for (int k = 0; k < 27; k++) JJ[k] = o[k];
      curErr2 = o[27];
    } else {
      block_sum<1, PR_NW>(v + 27, s_red, o);
      curErr2 = o[0];
    }
    __syncthreads();   // JJ is complete / s_sh may be rewritten
    PR_TICK(3)
#ifdef RS_TIMING
    pt[withJ ? 4 : 5] += 1;
#endif
  };
  for (;;) {
    bool wantJ = false, wantErr = false, proceed;
    if (state == DONE) { proceed = false; }
    else if (state == STARTED) { wantJ = wantErr = true; state = CALC_J; proceed = true; }
    else if (state == CALC_J) {
      for (int i = 0; i < 6; i++) prevParam[i] = param[i];
      double errAtJ = sqrt(curErr2);
      step();
      if (iters == 0) prevErrNorm = errAtJ;
      wantErr = true;
      state = CHECK_ERR;
      proceed = true;
    } else {
      errNorm = sqrt(curErr2);
      bool handled = false;
      if (errNorm > prevErrNorm) {
        if (++lambdaLg10 <= 16) {
          step();
          wantErr = true;
          state = CHECK_ERR;
          proceed = true;
          handled = true;
        }
      }
      if (!handled) {
        lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
        double dn = 0, pn = 0;
        for (int i = 0; i < 6; i++) { dn += (param[i] - prevParam[i]) * (param[i] - prevParam[i]); pn += prevParam[i] * prevParam[i]; }
        double rel = sqrt(dn) / (sqrt(pn) + DBL_EPSILON);
        if (++iters >= max_iter || rel < epsilon) { state = DONE; proceed = true; }
        else { prevErrNorm = errNorm; wantJ = wantErr = true; state = CALC_J; proceed = true; }
      }
    }
    if (!proceed || !wantErr) break;
    eval(wantJ);
  }
  if (tid == 0) {
    for (int i = 0; i < 6; i++) pose[i] = param[i];
    result[6] = 1;
  }
#ifdef RS_TIMING
  if (tid == 0 && slot < 2) printf("PR_TIMING slot=%d n=%d planar=%d | sums %lld init-solve %lld lm-solve %lld lm-eval %lld | evalJ %lld evalErr %lld iters %d (100MHz ticks)\n", slot, count, (int)planar, pt[0], pt[1], pt[2], pt[3], pt[4], pt[5], iters);
#endif
}

template <int PR_T>
__global__ __launch_bounds__(PR_T, PR_WAVES_PER_EU) void pnp_refine_kernel(PnpRefineArgs A) {
  pnp_refine_slot<PR_T>(A);
  // the tracker's per-slot decision after PnP, by the thread that wrote this slot's pose and status
  if (A.kp.state && threadIdx.x == 0) trk_policy_keyframe_slot(A.kp, blockIdx.x, A.result + (size_t)blockIdx.x * 8, A.pose + (size_t)blockIdx.x * 8);
}

// ---------------------------------------------------------------------------------------------------
// triangulation (one point per lane) and recoverPose
// ---------------------------------------------------------------------------------------------------
// cv::triangulatePoints for one correspondence: null vector of the 4x4 DLT matrix = last row of Vt.  The 4x4 Jacobi SVD
// runs in registers (gl_jacobi_svd_fixed, V only), rank-deficient systems included.
__device__ inline void gm_triangulate_one(const double* P1, const double* P2, double x1, double y1, double x2, double y2, double X[4]) {
  double A[16], At[16], W[4], Vt[16];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    A[0 * 4 + k] = x1 * P1[8 + k] - P1[0 + k];
    A[1 * 4 + k] = y1 * P1[8 + k] - P1[4 + k];
    A[2 * 4 + k] = x2 * P2[8 + k] - P2[0 + k];
    A[3 * 4 + k] = y2 * P2[8 + k] - P2[4 + k];
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) At[j * 4 + i] = A[i * 4 + j];
  // only Vt is read: no left-vector completion, also when the system is rank deficient (zero baseline)
  gl_jacobi_svd_fixed<4, 4, false>(At, W, Vt);
#pragma unroll
  for (int k = 0; k < 4; k++) X[k] = Vt[12 + k];
}

struct Proj2 { double P1[12], P2[12]; };

__global__ __launch_bounds__(256) void triangulate_kernel(Proj2 P, const float* __restrict__ p1, const float* __restrict__ p2, int n,
                                                          float* __restrict__ X3) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double X[4];
  gm_triangulate_one(P.P1, P.P2, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1], X);
  float xf[4] = {(float)X[0], (float)X[1], (float)X[2], (float)X[3]};
  float scale = xf[3] != 0.f ? 1.f / xf[3] : 1.f;  // convertPointsFromHomogeneous
  X3[3 * i] = xf[0] * scale; X3[3 * i + 1] = xf[1] * scale; X3[3 * i + 2] = xf[2] * scale;
}

struct RecoverArgs {
  double R1[9], R2[9], t[3];
  CamK cam;
  const float* p1;
  const float* p2;
  const u8* mask_in;  // may be null
  int n;
  u8* masks;   // [4][n]
  int* good;   // [4]
};

__global__ __launch_bounds__(256) void recover_pose_kernel(RecoverArgs A) {
  __shared__ int s_good[4];
  if (threadIdx.x < 4) s_good[threadIdx.x] = 0;
  __syncthreads();
  const double dist = 50.0;
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < A.n) {
    double x1 = (A.p1[2 * i] - A.cam.cx) / A.cam.fx, y1 = (A.p1[2 * i + 1] - A.cam.cy) / A.cam.fy;
    double x2 = (A.p2[2 * i] - A.cam.cx) / A.cam.fx, y2 = (A.p2[2 * i + 1] - A.cam.cy) / A.cam.fy;
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int c = 0; c < 4; c++) {
      const double* R = (c & 1) ? A.R2 : A.R1;
      double sg = c < 2 ? 1.0 : -1.0;
      double P[12];
      for (int r = 0; r < 3; r++) {
        for (int k = 0; k < 3; k++) P[r * 4 + k] = R[r * 3 + k];
        P[r * 4 + 3] = A.t[r] * sg;
      }
      double Q[4];
      gm_triangulate_one(P0, P, x1, y1, x2, y2, Q);
      bool m = (Q[2] * Q[3]) > 0;
      double q0 = Q[0] / Q[3], q1 = Q[1] / Q[3], q2 = Q[2] / Q[3];
      m = m && (q2 < dist);
      double z2 = P[8] * q0 + P[9] * q1 + P[10] * q2 + P[11] * 1.0;
      m = m && (z2 > 0) && (z2 < dist);
      if (A.mask_in) m = m && (A.mask_in[i] != 0);
      A.masks[(size_t)c * A.n + i] = m ? 1 : 0;
      if (m) atomicAdd(&s_good[c], 1);
    }
  }
  __syncthreads();
  if (threadIdx.x < 4 && s_good[threadIdx.x]) atomicAdd(&A.good[threadIdx.x], s_good[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
int geom_state_create(mvo_ctx* ctx) {
  GeomState* g = new GeomState();
  ctx->geom = g;
  size_t np = (size_t)ctx->B * ctx->maxpts;
  MVO_HIP(hipMalloc(&g->d_m1, np * 3 * sizeof(float)));
  MVO_HIP(hipMalloc(&g->d_m2, np * 2 * sizeof(float)));
  MVO_HIP(hipMalloc(&g->d_n, ctx->B * sizeof(int)));
  MVO_HIP(hipMalloc(&g->d_mask, np * 4));
  MVO_HIP(hipMalloc(&g->d_mask2, np));
  MVO_HIP(hipMalloc(&g->d_model, (size_t)ctx->B * 16 * sizeof(double)));
  MVO_HIP(hipMalloc(&g->d_model2, (size_t)ctx->B * 16 * sizeof(double)));
  MVO_HIP(hipMalloc(&g->d_result, (size_t)ctx->B * 8 * sizeof(int)));
  MVO_HIP(hipMalloc(&g->d_result2, (size_t)ctx->B * 8 * sizeof(int)));
  MVO_HIP(hipMalloc(&g->d_inl, np * sizeof(int)));
  MVO_HIP(hipMalloc(&g->d_pose, (size_t)ctx->B * 8 * sizeof(double)));
  MVO_HIP(hipMalloc(&g->d_x3, np * 3 * sizeof(float)));
  MVO_HIP(hipMalloc(&g->d_tmp, 64 * sizeof(double)));
  MVO_HIP(hipHostMalloc(&g->h_model, ((size_t)ctx->B * 16 + 64) * sizeof(double) * 3, hipHostMallocDefault));
  MVO_HIP(hipHostMalloc(&g->h_result, ((size_t)ctx->B * 8 + 16) * sizeof(int) * 4, hipHostMallocDefault));
  MVO_HIP(hipMemsetAsync(g->d_result, 0, (size_t)ctx->B * 8 * sizeof(int), ctx->stream));
  MVO_HIP(hipMemsetAsync(g->d_result2, 0, (size_t)ctx->B * 8 * sizeof(int), ctx->stream));
  return MVO_OK;
}

void geom_state_destroy(mvo_ctx* ctx) {
  GeomState* g = ctx->geom;
  if (!g) return;
  void* dev[] = {g->d_m1, g->d_m2, g->d_n, g->d_mask, g->d_mask2, g->d_model, g->d_model2, g->d_result, g->d_result2, g->d_inl,
                 g->d_pose, g->d_x3, g->d_tmp};
  for (void* p : dev) (void)hipFree(p);
  if (g->h_model) (void)hipHostFree(g->h_model);
  if (g->h_result) (void)hipHostFree(g->h_result);
  delete g;
  ctx->geom = nullptr;
}

template <class M, int NW = 1>
static void launch_ransac(mvo_ctx* ctx, hipStream_t st, int nslots, const float* m1, const float* m2, int stride1, int stride2,
                          const int* d_n, double thr, double conf, int max_iters, const ModelParams& P, u8* mask, int mask_stride,
                          double* model, int* result, int* idx = nullptr, int idx_stride = 0, int* work_ctr = nullptr, int grid = 0) {
  RansacArgs A;
  A.idx = idx; A.idx_stride = idx_stride; A.work_ctr = work_ctr; A.nslots = nslots;
  A.m1 = m1; A.m2 = m2; A.stride1 = stride1; A.stride2 = stride2; A.n = d_n;
  A.thr = thr; A.conf = conf; A.max_iters = max_iters; A.cap = ctx->maxpts; A.P = P;
  A.mask = mask; A.mask_stride = mask_stride; A.model = model; A.result = result;
  hipLaunchKernelGGL((ransac_kernel<M, NW>), dim3(work_ctr ? (grid < 1 ? 1 : (grid > nslots ? nslots : grid)) : nslots), dim3(64 * NW), 0, st, A);
}

// Device-level drivers used by the pipeline (inputs already resident, all slots per launch).
int geom_ransac_h(mvo_ctx* ctx, int nslots, const float* p1, const float* p2, const int* d_n, double thr, int max_iters, double conf,
                  u8* mask, double* model, int* result, hipStream_t st, int* work_ctr, int grid) {
  if (!st) st = ctx->stream;
  ModelParams P{};
  launch_ransac<HModel, RS_H_NW>(ctx, st, nslots, p1, p2, ctx->maxpts * 2, ctx->maxpts * 2, d_n, thr, conf, max_iters, P, mask, ctx->maxpts, model, result,
                                 nullptr, 0, work_ctr, grid);
  return MVO_OK;
}
int geom_ransac_f(mvo_ctx* ctx, int nslots, const float* p1, const float* p2, const int* d_n, double thr, int max_iters, double conf,
                  u8* mask, double* model, int* result, hipStream_t st, int* work_ctr, int grid) {
  if (!st) st = ctx->stream;
  ModelParams P{};
  launch_ransac<FModel>(ctx, st, nslots, p1, p2, ctx->maxpts * 2, ctx->maxpts * 2, d_n, thr, conf, max_iters, P, mask, ctx->maxpts, model, result,
                        nullptr, 0, work_ctr, grid);
  return MVO_OK;
}
int geom_pnp(mvo_ctx* ctx, int nslots, const float* obj, const float* img, const int* d_n, const double K[9], const double* dist5, int iters,
             float reproj, double conf, u8* mask, double* model, int* result, int* inl, double* pose, hipStream_t st, const TrkKeyframePolicy* kp) {
  if (!st) st = ctx->stream;
  ModelParams P{};
  P.cam = make_camk(K, dist5);
  {
    ProfScope ps(ctx, "pnp_ransac", st);
    launch_ransac<PnPModel>(ctx, st, nslots, obj, img, ctx->maxpts * 3, ctx->maxpts * 2, d_n, (double)reproj, conf, iters, P, mask, ctx->maxpts,
                            model, result, inl, ctx->maxpts);   // + the ordered inlier list and its length (result[5])
  }
  ProfScope ps2(ctx, "pnp_refine", st);
  PnpRefineArgs R;
  R.obj = obj; R.img = img; R.stride_pts = ctx->maxpts; R.inl = inl; R.result = result; R.model = model; R.n = d_n; R.pose = pose; R.cam = P.cam;
  if (kp) R.kp = *kp; else memset(&R.kp, 0, sizeof(R.kp));
  const bool wide = ctx->refine_waves ? ctx->refine_waves == 4 : nslots <= 64;
  if (wide) hipLaunchKernelGGL(pnp_refine_kernel<256>, dim3(nslots), dim3(256), 0, st, R);
  else hipLaunchKernelGGL(pnp_refine_kernel<64>, dim3(nslots), dim3(64), 0, st, R);
  return MVO_OK;
}


// ---- batched triangulation of key-frame <-> current matches (Tracker::triangulate_points, src/tracker.cpp:138-180) ----
// P = K [R|t] per view from the device-resident poses (rvec, tvec of T_cw); cheirality as the reference:
// p3d (float) transformed by both T_cw in double, rounded to float, z > 0 in both.
struct TriBatchArgs {
  const mvo_match* matches;  // [B][cap]
  const int* n_matches;      // [B]
  const float* kf_xy;        // [B][cap][2]  key-frame key-point positions (match query side)
  const float* cur_xy;       // [B][cap][2]  current key-point positions (match train side)
  const double* kf_pose;     // [B][8] rvec, tvec (T_cw)
  const double* cur_pose;    // [B][8]
  const int* pnp_result;     // [B][8] (slot valid iff [0] && [6]) or null
  CamK cam;
  int cap;
  float* X3;   // [B][cap][3]
  u8* valid;   // [B][cap]
  const int* list;   // optional device list of slots (blockIdx.y indexes it), *nlist entries
  const int* nlist;
};

__global__ __launch_bounds__(256) void triangulate_matches_kernel(TriBatchArgs A) {
  __shared__ double s_P[2][12];
  __shared__ double s_T[2][12];
  if (A.list && (int)blockIdx.y >= *A.nlist) return;   // block-uniform
  const int slot = A.list ? A.list[blockIdx.y] : blockIdx.y;
  const int n = min(max(A.n_matches[slot], 0), A.cap);
  if (blockIdx.x * 256 >= n) return;
  bool pose_ok = true;
  if (A.pnp_result) pose_ok = A.pnp_result[slot * 8] && A.pnp_result[slot * 8 + 6];
  if (threadIdx.x < 2) {
    const double* pose = (threadIdx.x == 0 ? A.kf_pose : A.cur_pose) + (size_t)slot * 8;
    double R[9];
    gm_rodrigues_v2m(pose, R, nullptr);
    double E[12] = {R[0], R[1], R[2], pose[3], R[3], R[4], R[5], pose[4], R[6], R[7], R[8], pose[5]};
    const double Km[9] = {A.cam.fx, 0, A.cam.cx, 0, A.cam.fy, A.cam.cy, 0, 0, 1};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 4; j++) {
        s_P[threadIdx.x][i * 4 + j] = Km[i * 3] * E[j] + Km[i * 3 + 1] * E[4 + j] + Km[i * 3 + 2] * E[8 + j];
        s_T[threadIdx.x][i * 4 + j] = E[i * 4 + j];
      }
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const size_t b = (size_t)slot * A.cap;
  const mvo_match m = A.matches[b + i];
  const float* pr = A.kf_xy + 2 * (b + m.query_idx);
  const float* pc = A.cur_xy + 2 * (b + m.train_idx);
  double X[4];
  gm_triangulate_one(s_P[0], s_P[1], pr[0], pr[1], pc[0], pc[1], X);
  float xf[4] = {(float)X[0], (float)X[1], (float)X[2], (float)X[3]};
  float scale = xf[3] != 0.f ? 1.f / xf[3] : 1.f;
  float px = xf[0] * scale, py = xf[1] * scale, pz = xf[2] * scale;
  // cv::Affine3d * cv::Point3f -> Point3d -> Point3f
  float zr = (float)(s_T[0][8] * (double)px + s_T[0][9] * (double)py + s_T[0][10] * (double)pz + s_T[0][11]);
  float zc = (float)(s_T[1][8] * (double)px + s_T[1][9] * (double)py + s_T[1][10] * (double)pz + s_T[1][11]);
  A.X3[3 * (b + i)] = px; A.X3[3 * (b + i) + 1] = py; A.X3[3 * (b + i) + 2] = pz;
  A.valid[b + i] = (pose_ok && zr > 0 && zc > 0) ? 1 : 0;
}

int geom_triangulate_matches(mvo_ctx* ctx, int nslots, int max_matches, const mvo_match* matches, const int* n_matches,
                             const float* kf_xy, const float* cur_xy, const double* kf_pose, const double* cur_pose,
                             const int* pnp_result, const double K[9], float* X3, u8* valid, const int* d_list, const int* d_nlist) {
  if (max_matches <= 0) return MVO_OK;
  TriBatchArgs A;
  A.list = d_list; A.nlist = d_nlist;
  A.matches = matches; A.n_matches = n_matches; A.kf_xy = kf_xy; A.cur_xy = cur_xy; A.kf_pose = kf_pose; A.cur_pose = cur_pose;
  A.pnp_result = pnp_result; A.cam = CamK{K[0], K[4], K[2], K[5]}; A.cap = ctx->maxpts; A.X3 = X3; A.valid = valid;
  dim3 grid((max_matches + 255) / 256, nslots);
  hipLaunchKernelGGL(triangulate_matches_kernel, grid, dim3(256), 0, ctx->stream, A);
  return MVO_OK;
}

static int upload_pairs(mvo_ctx* ctx, const float* p1, int c1, const float* p2, int c2, int n) {
  GeomState* g = ctx->geom;
  if (n > ctx->maxpts) { ctx->set_error("point count exceeds max_points"); return MVO_E_CAPACITY; }
  if (n > 0) {
    MVO_HIP(hipMemcpyAsync(g->d_m1, p1, (size_t)n * c1 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    MVO_HIP(hipMemcpyAsync(g->d_m2, p2, (size_t)n * c2 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  }
  int* hn = (int*)ctx->h_pin;  // pinned: the async copy must not read a dead stack slot
  hn[0] = n;
  MVO_HIP(hipMemcpyAsync(g->d_n, hn, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  return MVO_OK;
}

extern "C" int mvo_find_homography_ransac(mvo_ctx* ctx, const float* p1, const float* p2, int n, double thr, int max_iters,
                                          double confidence, uint8_t* mask, double H[9], int* n_inliers) {
  if (!ctx || !p1 || !p2 || !mask || n < 0) return MVO_E_ARG;
  if (n < 4) { ctx->set_error("findHomography needs at least 4 correspondences"); return MVO_E_ARG; }
  if (thr <= 0) thr = 3;
  GeomState* g = ctx->geom;
  int rc = upload_pairs(ctx, p1, 2, p2, 2, n);
  if (rc) return rc;
  geom_ransac_h(ctx, 1, g->d_m1, g->d_m2, g->d_n, thr, max_iters, confidence, g->d_mask, g->d_model, g->d_result, nullptr);
  MVO_HIP(hipMemcpyAsync(mask, g->d_mask, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipMemcpyAsync(g->h_model, g->d_model, 9 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipMemcpyAsync(g->h_result, g->d_result, 8 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  if (n_inliers) *n_inliers = g->h_result[0] ? g->h_result[1] : 0;
  if (!g->h_result[0]) return MVO_E_DEGENERATE;
  if (H) memcpy(H, g->h_model, 9 * sizeof(double));
  return MVO_OK;
}

extern "C" int mvo_find_fundamental_ransac(mvo_ctx* ctx, const float* p1, const float* p2, int n, double thr, double confidence,
                                           int max_iters, uint8_t* mask, double F[9], int* n_inliers) {
  if (!ctx || !p1 || !p2 || !mask || n < 0) return MVO_E_ARG;
  if (n_inliers) *n_inliers = 0;
  if (n < 7) return MVO_E_DEGENERATE;  // OpenCV returns an empty Mat, mask untouched
  if (thr <= 0) thr = 3;
  if (confidence < DBL_EPSILON || confidence > 1 - DBL_EPSILON) confidence = 0.99;
  GeomState* g = ctx->geom;
  int rc = upload_pairs(ctx, p1, 2, p2, 2, n);
  if (rc) return rc;
  geom_ransac_f(ctx, 1, g->d_m1, g->d_m2, g->d_n, thr, max_iters, confidence, g->d_mask, g->d_model, g->d_result, nullptr);
  MVO_HIP(hipMemcpyAsync(mask, g->d_mask, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipMemcpyAsync(g->h_model, g->d_model, 9 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipMemcpyAsync(g->h_result, g->d_result, 8 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  if (n_inliers) *n_inliers = g->h_result[0] ? g->h_result[1] : 0;
  if (!g->h_result[0]) return MVO_E_DEGENERATE;
  if (F) memcpy(F, g->h_model, 9 * sizeof(double));
  return MVO_OK;
}

extern "C" int mvo_solve_pnp_ransac(mvo_ctx* ctx, const float* obj, const float* img, int n, const double K[9], const double d[5],
                                    int iters, float reproj_err, double confidence, double rvec[3], double tvec[3], int* inlier_idx,
                                    int* n_inliers) {
  if (!ctx || !obj || !img || !K || !rvec || !tvec || n < 0) return MVO_E_ARG;
  if (n_inliers) *n_inliers = 0;
  if (n < 4) { ctx->set_error("solvePnPRansac needs at least 4 correspondences"); return MVO_E_ARG; }
  GeomState* g = ctx->geom;
  int rc = upload_pairs(ctx, obj, 3, img, 2, n);
  if (rc) return rc;
  geom_pnp(ctx, 1, g->d_m1, g->d_m2, g->d_n, K, d, iters, reproj_err, confidence, g->d_mask, g->d_model, g->d_result, g->d_inl, g->d_pose, nullptr);
  MVO_HIP(hipMemcpyAsync(g->h_model, g->d_pose, 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipMemcpyAsync(g->h_result, g->d_result, 8 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  if (!g->h_result[0] || !g->h_result[6]) return MVO_E_DEGENERATE;
  int cnt = g->h_result[5];
  for (int i = 0; i < 3; i++) { rvec[i] = g->h_model[i]; tvec[i] = g->h_model[3 + i]; }
  if (n_inliers) *n_inliers = cnt;
  if (inlier_idx && cnt > 0) {
    MVO_HIP(hipMemcpyAsync(inlier_idx, g->d_inl, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MVO_HIP(hipStreamSynchronize(ctx->stream));
  }
  return MVO_OK;
}

extern "C" int mvo_triangulate(mvo_ctx* ctx, const double P1[12], const double P2[12], const float* p1, const float* p2, int n,
                               float* X3) {
  if (!ctx || !P1 || !P2 || n < 0 || (n && (!p1 || !p2 || !X3))) return MVO_E_ARG;
  if (n == 0) return MVO_OK;
  GeomState* g = ctx->geom;
  int rc = upload_pairs(ctx, p1, 2, p2, 2, n);
  if (rc) return rc;
  Proj2 P;
  memcpy(P.P1, P1, sizeof(P.P1));
  memcpy(P.P2, P2, sizeof(P.P2));
  hipLaunchKernelGGL(triangulate_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, P, g->d_m1, g->d_m2, n, g->d_x3);
  MVO_HIP(hipMemcpyAsync(X3, g->d_x3, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  return MVO_OK;
}

__global__ void decompose_essential_kernel(const double* __restrict__ E, double* __restrict__ out /* R1[9] R2[9] t[3] */) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double e[9], w[3], U[9], Vt[9];
  for (int i = 0; i < 9; i++) e[i] = E[i];
  gl_svd3(e, w, U, Vt);
  if (gl_det3(U) < 0) for (int i = 0; i < 9; i++) U[i] *= -1.;
  if (gl_det3(Vt) < 0) for (int i = 0; i < 9; i++) Vt[i] *= -1.;
  const double Wm[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
  const double Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
  double tmp[9];
  gl_mat3mul(U, Wm, tmp); gl_mat3mul(tmp, Vt, out);
  gl_mat3mul(U, Wt, tmp); gl_mat3mul(tmp, Vt, out + 9);
  out[18] = U[2]; out[19] = U[5]; out[20] = U[8];
}

extern "C" int mvo_recover_pose(mvo_ctx* ctx, const double E[9], const float* p1, const float* p2, int n, const double K[9],
                                double R[9], double t[3], uint8_t* mask_io, int* n_good) {
  if (!ctx || !E || !p1 || !p2 || !K || !R || !t || n < 1) return MVO_E_ARG;
  GeomState* g = ctx->geom;
  int rc = upload_pairs(ctx, p1, 2, p2, 2, n);
  if (rc) return rc;
  double* dE = g->d_tmp;
  double* dOut = g->d_tmp + 16;
  MVO_HIP(hipMemcpyAsync(dE, E, 9 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(decompose_essential_kernel, dim3(1), dim3(64), 0, ctx->stream, dE, dOut);
  MVO_HIP(hipMemcpyAsync(g->h_model, dOut, 21 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  RecoverArgs A;
  memcpy(A.R1, g->h_model, 9 * sizeof(double));
  memcpy(A.R2, g->h_model + 9, 9 * sizeof(double));
  memcpy(A.t, g->h_model + 18, 3 * sizeof(double));
  A.cam = CamK{K[0], K[4], K[2], K[5]};
  A.p1 = g->d_m1; A.p2 = g->d_m2; A.n = n;
  A.mask_in = nullptr;
  if (mask_io) {
    MVO_HIP(hipMemcpyAsync(g->d_mask2, mask_io, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    A.mask_in = g->d_mask2;
  }
  A.masks = g->d_mask;  // 4 * maxpts bytes
  A.good = g->d_result2;
  MVO_HIP(hipMemsetAsync(g->d_result2, 0, 4 * sizeof(int), ctx->stream));
  hipLaunchKernelGGL(recover_pose_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, A);
  MVO_HIP(hipMemcpyAsync(g->h_result, g->d_result2, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  const int* good = g->h_result;
  int best;
  if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) best = 0;
  else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) best = 1;
  else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) best = 2;
  else best = 3;
  memcpy(R, (best & 1) ? A.R2 : A.R1, 9 * sizeof(double));
  for (int k = 0; k < 3; k++) t[k] = best < 2 ? A.t[k] : -A.t[k];
  if (mask_io) {
    MVO_HIP(hipMemcpyAsync(mask_io, g->d_mask + (size_t)best * n, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    MVO_HIP(hipStreamSynchronize(ctx->stream));
  }
  if (n_good) *n_good = good[best];
  return MVO_OK;
}

extern "C" int mvo_find_essential_ransac(mvo_ctx* ctx, const float* p1, const float* p2, int n, const double K[9], double prob,
                                         double thr, int max_iters, uint8_t* mask, double E[9], int* n_inliers) {
  if (!ctx || !p1 || !p2 || !K || !mask || n < 0) return MVO_E_ARG;
  if (n_inliers) *n_inliers = 0;
  if (n < 5) { ctx->set_error("findEssentialMat needs at least 5 correspondences"); return MVO_E_ARG; }
  GeomState* g = ctx->geom;
  int rc = upload_pairs(ctx, p1, 2, p2, 2, n);
  if (rc) return rc;
  ModelParams P{};
  P.cam = CamK{K[0], K[4], K[2], K[5]};
  double thr_n = thr / ((K[0] + K[4]) / 2);  // findEssentialMat: threshold /= (fx + fy) / 2
  launch_ransac<EModel>(ctx, ctx->stream, 1, g->d_m1, g->d_m2, ctx->maxpts * 2, ctx->maxpts * 2, g->d_n, thr_n, prob, max_iters, P,
                        g->d_mask, ctx->maxpts, g->d_model, g->d_result);
  MVO_HIP(hipMemcpyAsync(mask, g->d_mask, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipMemcpyAsync(g->h_model, g->d_model, 9 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipMemcpyAsync(g->h_result, g->d_result, 8 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MVO_HIP(hipStreamSynchronize(ctx->stream));
  if (n_inliers) *n_inliers = g->h_result[0] ? g->h_result[1] : 0;
  if (!g->h_result[0]) return MVO_E_DEGENERATE;
  if (E) memcpy(E, g->h_model, 9 * sizeof(double));
  return MVO_OK;
}
