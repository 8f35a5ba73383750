// oracle/mvo_oracle.h — TEST INFRASTRUCTURE ONLY.  C entry points of the CPU oracle
// (liborc.so).  See orc_common.h for the scope statement.  PARITY UNPINNED.
#pragma once
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, size, angle, response; int octave, class_id; } orc_keypoint;  // cv::KeyPoint
typedef struct { int query_idx, train_idx, img_idx; float distance; } orc_match;          // cv::DMatch

int orc_pyrdown(const unsigned char* src, int w, int h, int stride, unsigned char* dst, int dstride);
int orc_lk_track(const unsigned char* prev, const unsigned char* next, int w, int h, int stride, int cn,
                 const float* prev_pts, int n, float* next_pts, unsigned char* status, float* err,
                 int win, int max_level, int max_count, double epsilon, double min_eig_thr);

int orc_fast9_nms(const unsigned char* img, int w, int h, int stride, int threshold, int* xys, int cap);
int orc_orb_level_info(int w, int h, int nfeatures, int* lw, int* lh, float* scale, int* quota);
int orc_resize_linear_exact(const unsigned char* src, int sw, int sh, int sstride, unsigned char* dst,
                            int dw, int dh, int dstride);
int orc_gauss7(const unsigned char* src, int w, int h, int stride, unsigned char* dst, int dstride, int mode);
float orc_fast_atan2(float y, float x);
int orc_orb_detect_and_compute(const unsigned char* img, int w, int h, int stride, int channels,
                               int nfeatures, int fast_threshold, int blur_mode, orc_keypoint* out_kps,
                               unsigned char* out_desc, int cap);

int orc_match_knn2_ratio(const unsigned char* q, int nq, const unsigned char* t, int nt, double ratio,
                         orc_match* out, int cap);

#ifdef __cplusplus
}
#endif
