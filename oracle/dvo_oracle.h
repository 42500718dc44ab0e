/*
 * dvo_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C + SSE3 intrinsics) of the dense RGB-D alignment hot path of
 * jesusbriales/dvo_slam: dvo::DenseTracker::match() over dvo::core::RgbdImagePyramid.
 * It is the checker the HIP path in dvo_slam_amd/ is compared against.  Nothing in the
 * product path may include, link or call this file: only tests/, __graft_entry__.smoke()
 * and the cpu_baseline leg of bench.py do.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors for this path
 *  and cannot be compiled here (it needs Eigen, OpenCV, Sophus, Boost,
 * none of which are installed), so this restatement is pinned only by its own unit tests
 * (textbook identities, finite differences, hand-computed 8x8 images) and by the committed
 * fixtures it generated itself (tests/golden/).
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 */
#ifndef DVO_ORACLE_H_
#define DVO_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 8

/* how the projection / weight reciprocal is formed */
enum {
  ORC_RCP_SSE = 0,  /* _mm_rcp_ps, as the reference (dense_tracking_impl.cpp:192,700): host specific */
  ORC_RCP_EXACT = 1, /* IEEE division in the same rounding mode: portable, what the HIP path does   */
  ORC_RCP_CLEAN = 2  /* ORC_RCP_EXACT, and additionally the scale estimate uses every residual's own outer product (no Q5)
                        and the log-likelihood counts the last n % 50 residuals (no Q6): SURVEY.md's "CLEAN" oracle, only
                        there to report how far the bug-compatible result is from the intended algorithm */
};

/* In what order the oracle adds up the SAME fp32 terms of the scale estimate (computeScaleSse) and of the normal equations
 * (rankUpdate / b -= J^T W r).  ORC_SUM_REFERENCE is the reference's order: one sequential fp32 accumulator in scan order
 * (dense_tracking_impl.cpp:590-638, math_sse.cpp:117).  The other two re-associate the sum -- what any implementation that
 * does not run the pixels strictly one after the other does -- and exist so that the tests can show how far the reference
 * algorithm's OWN answer moves under a different summation order (the fork criterion of tests/test_gpu_parity.py). */
enum {
  ORC_SUM_REFERENCE = 0,
  ORC_SUM_FP64 = 1,     /* the same fp32 products, accumulated in double */
  ORC_SUM_BLOCKED = 2,  /* fp32 partial sums over blocks of 256 consecutive points, the partial sums added up in fp32 */
  ORC_SUM_BLOCKED_32 = 3,   /* ... blocks of 32 */
  ORC_SUM_BLOCKED_2048 = 4  /* ... blocks of 2048 */
};

/* dense_tracking.h:71-81 */
enum {
  ORC_TERM_ITERATIONS_EXCEEDED = 0,
  ORC_TERM_INCREMENT_TOO_SMALL = 1,
  ORC_TERM_LOGLIKELIHOOD_DECREASED = 2,
  ORC_TERM_TOO_FEW_CONSTRAINTS = 3,
  ORC_TERM_UNSET = -1
};

/* live fields of DenseTracker::Config (dense_tracking.h:42-69, defaults dense_tracking_config.cpp:27-41) */
typedef struct {
  int first_level, last_level;
  int max_iterations_per_level;
  double precision;
  double mu;
  int use_initial_estimate;
  float intensity_derivative_threshold;
  float depth_derivative_threshold;
  int rcp_mode;
  int sum_mode; /* ORC_SUM_* below; not a reference option: test instrumentation (default ORC_SUM_REFERENCE) */
  int ll_guard; /* test instrumentation, default 0 = the reference: computeCompleteDataLogLikelihood multiplies 50 terms
                   (1 + 0.2 r^T P r) in a double before it takes one log (dense_tracking_impl.cpp:413-419); when 50 consecutive
                   residuals all have a Mahalanobis distance above ~7e6 (noise-free synthetic depth gives precisions of 1e9 and
                   more; sensor data never does) that product overflows, the likelihood is -inf and the iteration is rejected.
                   1: the same sum without the overflow (a log is taken early whenever the running product passes 1e200), to tell
                   this artefact apart from everything else when a GPU run and the oracle part ways. */
} orc_config;

/* DenseTracker::IterationStats (dense_tracking.h:83-100) + the linear system of that iteration */
typedef struct {
  int id;
  int valid_constraints;
  double tdist_loglik;       /* = -ll */
  double tdist_mean[2];
  double tdist_precision[4]; /* column-major 2x2 */
  double prior_loglik;
  double increment[6];       /* only valid if has_increment */
  double information[36];    /* A_d, only valid if has_increment */
  double rhs[6];             /* b_d, extra (not in the reference's stats) */
  float scale[4];            /* 2x2 before inversion, extra */
  int has_increment;
  double estimate[16];       /* extra: the transform (column-major 4x4, estimate()) whose float cast the residual stage of this
                                iteration used, dense_tracking.cpp:263 */
  double initial[16];        /* extra: initial() behind :260 of this iteration (what a continuation from here resumes with) */
} orc_iteration_stats;

/* DenseTracker::LevelStats (dense_tracking.h:103-116) */
typedef struct {
  int id;
  int max_valid_pixels;
  int valid_pixels;
  int termination;
  int n_iterations;
  int first_iteration; /* index of this level's first entry in orc_result.iterations */
} orc_level_stats;

/* DenseTracker::Result (dense_tracking.h:125-140) */
typedef struct {
  double T[16];           /* column-major 4x4, Result::Transformation */
  double information[36]; /* Result::Information */
  double loglik;          /* Result::LogLikelihood */
  int n_levels;
  orc_level_stats levels[ORC_MAX_LEVELS];
  int n_iterations;                 /* entries written to iterations[] */
  orc_iteration_stats *iterations;  /* caller provided, may be NULL */
  int iterations_capacity;
  int is_nan;                       /* Result::isNaN() */
} orc_result;

typedef struct orc_pyramid orc_pyramid;

void orc_default_config(orc_config *cfg);

/* RgbdCameraPyramid(w,h,K) + create(intensity, depth) + build(levels): rgbd_image.cpp:141-172,264-296 */
orc_pyramid *orc_pyramid_create(const float *intensity, const float *depth, int width, int height,
                                float fx, float fy, float ox, float oy, int levels);
void orc_pyramid_destroy(orc_pyramid *p);

/* accessors used by the stage-wise parity tests */
int orc_pyramid_levels(const orc_pyramid *p);
void orc_level_size(const orc_pyramid *p, int level, int *w, int *h);
void orc_level_intrinsics(const orc_pyramid *p, int level, float k[4]);
/* plane: 0 I, 1 Z, 2 Ix, 3 Iy, 4 Zx, 5 Zy */
const float *orc_level_plane(orc_pyramid *p, int level, int plane);
/* 48-byte records {x,y,z,1, I,Z,Ix,Iy,Zx,Zy,0,0} of the selected points: point_selection.cpp:89-152 */
int orc_select(orc_pyramid *p, int level, float ti, float td, const float **records);
/* pixel index (y*w+x) of every selected record, same order */
const int *orc_select_index(orc_pyramid *p, int level, float ti, float td);

/*
 * computeResidualsSse (dense_tracking_impl.cpp:133-393) on one level for the float transform T (column-major 4x4).
 * out_points_error: n x 12 floats, out_residuals: n x 2 floats, out_valid: one byte per processed selected point
 * (the Debug=true template), any may be NULL.  Returns n.
 */
int orc_compute_residuals(orc_pyramid *ref, orc_pyramid *cur, int level, float ti, float td, const float *T,
                          int rcp_mode, float *out_points_error, float *out_residuals, unsigned char *out_valid);

/* DenseTracker::match(RgbdImagePyramid&, RgbdImagePyramid&, Result&): dense_tracking.cpp:123-376 */
int orc_match(const orc_config *cfg, orc_pyramid *ref, orc_pyramid *cur, const double *T_init, orc_result *res);

/* TEST INSTRUMENTATION (not a reference entry): the state DenseTracker::match holds at the top of an iteration body
 * (dense_tracking.cpp:259), from which orc_match_from continues the reference's control flow (:247-363 and the level loop around
 * it) exactly as orc_match would.  It exists for the fork criterion of the parity tests (tests/fork_criterion.py): when a
 * free-running GPU match and the oracle take different decisions at some iteration, the oracle is continued from the GPU's OWN
 * state behind that decision and the GPU's remaining iterations are held to that continuation.
 *   level, iteration    the level and the value of `Iteration` (:354) = index of the iteration about to run; 0 = a level start
 *   estimate, initial   the Revertables' current values (:147-150), column-major 4x4, BEFORE the increment x is applied
 *   x                   the increment the iteration applies (:259): the previous iteration's solution (:347), or at a level
 *                       start log(inc) of the last applied increment (:238, Q1)
 *   last_error          `Error` (:304): -ll of the level's last accepted iteration; DBL_MAX at a level start (:210)
 *   precision           the 2x2 precision of the previous iteration of this level (weights of the resumed one, :291), column-major
 *   previous_*          Information / (TDistributionLogLikelihood + PriorLogLikelihood) of that previous iteration: what :369-372
 *                       reads when the resumed iteration of the LAST level is rejected at once */
typedef struct {
  int level, iteration;
  double estimate[16], initial[16];
  double x[6];
  double last_error;
  float precision[4];
  int has_previous;
  double previous_information[36];
  double previous_loglik;
} orc_match_state;
/* res->levels[0] is from->level and counts only the iterations run here (their ids continue at from->iteration).  -4: bad state */
int orc_match_from(const orc_config *cfg, orc_pyramid *ref, orc_pyramid *cur, const orc_match_state *from, orc_result *res);

/* bench.py's many-thread CPU baseline: n_threads threads, each calling orc_match(cfg, ref, curs[k % n_curs]) for `seconds`;
 * returns the number of alignments finished (all threads), *elapsed_s the wall time.  `ref`'s selection must exist already (run
 * one orc_match before): the pyramids are shared read-only. */
long long orc_bench_threads(const orc_config *cfg, orc_pyramid *ref, orc_pyramid *const *curs, int n_curs, int n_threads,
                            double seconds, double *elapsed_s);

/* Frame ingest.
 * depth: SurfacePyramid::convertRawDepthImage / ...Sse (surface_pyramid.cpp:44-105): 0 -> NaN, else (float)raw * scale
 * (one fp32 multiply); callers use scale 1/5000 (benchmark_slam.cpp:77) or 0.001 (camera_dense_tracking.cpp:234).
 * gray: cv::cvtColor(CV_BGR2GRAY) on 8-bit data followed by convertTo(CV_32F) (benchmark_slam.cpp:60-68,
 * camera_dense_tracking.cpp:219-224).  OpenCV is a build-time dependency that is NOT in the reference tree (ROS
 * fuerte/groovy era, OpenCV 2.4.x); restated here is its published 8-bit fixed-point rule
 *   Y = (B*1868 + G*9617 + R*4899 + (1 << 13)) >> 14                        (coefficients 0.114/0.587/0.299 in Q14)
 * Strides are in elements of the respective type (bytes for the image). */
void orc_ingest_depth_u16(const unsigned short *raw, int width, int height, int stride, float scale, float *out);
void orc_ingest_gray_from_bgr8(const unsigned char *bgr, int width, int height, int stride_bytes, float *out);
void orc_ingest_gray_from_gray8(const unsigned char *gray, int width, int height, int stride_bytes, float *out);

/* small pieces exported for unit tests */
void orc_se3_exp(const double xi[6], double T[16]);
void orc_se3_log(const double T[16], double xi[6]);
void orc_jacobian(const float p[3], float Jw[12], float Jz[6]); /* dense_tracking.cpp:448-476, row-major 2x6 */
/* math_sse.cpp:82-207: A(6x6 col-major) += J^T alpha J via the packed 2x2 block accumulator */
void orc_rank_update(const float *J2x6_colmajor, const float *alpha_colmajor, int n, float *A36);
float orc_weights_scale_loglik(const float *residuals, int n, const float *prec_in, int unit_weights, int rcp_mode,
                               float *weights_out, float *scale_out, float *prec_out);
/* one iteration body at a fixed pose and previous precision (dense_tracking.cpp:271-347 minus accept test and solve):
 * returns n; scale / precision column-major 2x2, ll as computeCompleteDataLogLikelihood returns it, A column-major 6x6
 * (without Mu), b (without Mu).  Outputs untouched when n < 6. */
int orc_iteration(orc_pyramid *ref, orc_pyramid *cur, int level, float ti, float td, const float *T, const float prec_in[4],
                  int unit_weights, int rcp_mode, float scale_out[4], float prec_out[4], float *ll_out, float A36[36],
                  float b6[6]);
float orc_host_rcp(float x); /* _mm_rcp_ps lane 0 on this host */
void orc_host_rcp_many(const float *in, float *out, int n);

#ifdef __cplusplus
}
#endif
#endif
