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
/* calcOpticalFlowPyrLK on a true-colour pair (CV_8UC3 / 8UC4 rows of `bpp` interleaved bytes per pixel, first three tracked) */
int orc_lk_track_color(const unsigned char* prev, const unsigned char* next, int w, int h, int stride, int bpp,
                       const float* prev_pts, int n, float* next_pts, unsigned char* status, float* err, int win, int max_level,
                       int max_count, double epsilon, double min_eig_thr);

int orc_fast9_nms(const unsigned char* img, int w, int h, int stride, int threshold, int* xys, int cap);
int orc_retain_best(const float* resp, int n, int n_keep, int depth_limit, int* out_idx);
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

int orc_find_homography_ransac(const float* p1, const float* p2, int n, double thr, int max_iters, double confidence,
                               unsigned char* mask, double* H, int* stats);
int orc_find_fundamental_ransac(const float* p1, const float* p2, int n, double thr, double confidence, int max_iters,
                                unsigned char* mask, double* F, int* stats);
int orc_h4_kernel(const float* p1, const float* p2, int n, double* H);
int orc_f7_kernel(const float* p1, const float* p2, double* F);
void orc_svd(const double* A, int m, int n, double* w, double* U, double* Vt, int full);
void orc_eigen_sym(const double* A, int n, double* W, double* V);
int orc_solve_cubic(const double* c, double* x);
unsigned orc_rng_next(unsigned long long* state);
int orc_ransac_update_num_iters(double p, double ep, int model_points, int max_iters);
int orc_triangulate(const double* P1, const double* P2, const float* p1, const float* p2, int n, float* X3, float* X4);
int orc_recover_pose(const double* E, const float* p1, const float* p2, int n, const double* K, double* R, double* t,
                     unsigned char* mask_io);

int orc_rodrigues_v2m(const double* r, double* R);
int orc_rodrigues_m2v(const double* R, double* r);
int orc_epnp(const float* obj, const float* img, int n, const double* K, double* rvec, double* tvec);
int orc_solve_p3p4(const float* obj, const float* img, const double* K, const double* d, double* rvec, double* tvec);
int orc_solve_deg4(double a, double b, double c, double d, double e, double* roots);
int orc_solve_pnp_ransac(const float* obj, const float* img, int n, const double* K, const double* d, int iters,
                         float reproj_err, double confidence, double* rvec, double* tvec, int* inlier_idx,
                         int* n_inliers, int* stats);

int orc_e5_kernel(const double* q1, const double* q2, int n, double* models);
int orc_find_essential_ransac(const float* p1, const float* p2, int n, const double* K, double prob, double thr,
                              int max_iters, unsigned char* mask, double* E, int* stats);

#ifdef __cplusplus
}
#endif
