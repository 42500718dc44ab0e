/*
 * dvo_amd.h -- C ABI of the MI355X-native dense RGB-D tracking core.
 *
 * Drop-in boundary for ONE path of jesusbriales/dvo_slam: dvo::DenseTracker::match() over
 * dvo::core::RgbdImagePyramid (dvo_core/include/dvo/dense_tracking.h:39-213,
 * dvo_core/include/dvo/core/rgbd_image.h:127-262).  The reference has no FFI layer (its seam is a
 * C++ class using Eigen / cv::Mat / boost::shared_ptr types), so this header declares the POD
 * interface a binding would use; include/dvo_amd/dense_tracking.hpp re-declares the reference's
 * class / field names on top of it.  Plain pointers and sizes only; no exceptions cross the ABI;
 * every entry point returns a dvo_amd_status.
 *
 * Conventions (same as the reference):
 *  - intensity: float32, 0..255 (benchmark_slam.cpp:60-68); depth: float32 metres, NaN = invalid
 *    (surface_pyramid.cpp:65-105); both row-major, same size; width of every used pyramid level must
 *    be a multiple of 4 (the reference's SSE derivative needs this too: rgbd_image_sse.cpp:258).
 *  - 4x4 transforms are column-major double[16] (Eigen::Affine3d::matrix().data()).
 *  - 6x6 matrices are column-major double[36] (symmetric anyway).
 *  - Result transformation maps current-frame points into the reference frame
 *    (dense_tracking.cpp:371: estimate^-1).
 */
#ifndef DVO_AMD_H_
#define DVO_AMD_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVO_AMD_MAX_LEVELS 8
#define DVO_AMD_ABI_VERSION 3

typedef enum {
  DVO_AMD_OK = 0,
  DVO_AMD_ERR_INVALID_ARGUMENT = 1,
  DVO_AMD_ERR_NO_DEVICE = 2,        /* HIP runtime / GPU not usable: the library never falls back to a CPU path */
  DVO_AMD_ERR_HIP = 3,              /* a HIP call failed; see dvo_amd_last_error() */
  DVO_AMD_ERR_OUT_OF_MEMORY = 4,
  DVO_AMD_ERR_INSANE_CONFIG = 5,    /* Config::IsSane() false (dense_tracking_config.cpp:55-58) */
  DVO_AMD_ERR_TOO_FEW_LEVELS = 6,   /* pyramid holds fewer levels than FirstLevel + 1 */
  DVO_AMD_ERR_CAPACITY = 7,         /* caller-provided iteration array too small */
  DVO_AMD_ERR_DEVICE_MISMATCH = 8,
  DVO_AMD_ERR_NAN_INIT = 9,         /* UseInitialEstimate with a NaN transform (dense_tracking.cpp:139 assert) */
  DVO_AMD_ERR_COMM = 10,
  DVO_AMD_ERR_IO = 11,              /* file cannot be opened / read */
  DVO_AMD_ERR_FORMAT = 12           /* file is not what the reader supports (see the reader's comment) */
} dvo_amd_status;

/* DenseTracker::TerminationCriteria::Enum, dense_tracking.h:71-81 */
typedef enum {
  DVO_AMD_TERM_ITERATIONS_EXCEEDED = 0,
  DVO_AMD_TERM_INCREMENT_TOO_SMALL = 1,
  DVO_AMD_TERM_LOGLIKELIHOOD_DECREASED = 2,
  DVO_AMD_TERM_TOO_FEW_CONSTRAINTS = 3,
  DVO_AMD_TERM_UNSET = -1
} dvo_amd_termination;

/* The live fields of DenseTracker::Config (dense_tracking.h:42-69); defaults dense_tracking_config.cpp:27-41.
 * UseWeighting / UseParallel / InfluenceFunction* / ScaleEstimator* are never read by match() and are not mirrored. */
typedef struct {
  int first_level;                /* FirstLevel, default 3 */
  int last_level;                 /* LastLevel, default 1 */
  int max_iterations_per_level;   /* MaxIterationsPerLevel, default 100 */
  double precision;               /* Precision, default 5e-7 */
  double mu;                      /* Mu, default 0 */
  int use_initial_estimate;       /* UseInitialEstimate, default 0 */
  float intensity_derivative_threshold; /* IntensityDerivativeThreshold, default 0 */
  float depth_derivative_threshold;     /* DepthDerivativeThreshold, default 0 */
  /* NOT a field of the reference (ABI version 3): how a pyramid level is cut into wave segments -- where the fp32 sums of a
   * residual pass are cut.  The pass walks the level's SELECTED pixels in scan order (the points PointSelection keeps, compacted
   * like the reference's own array), a wave takes a run of consecutive points in steps of 64.  Like every other field of this
   * struct it is part of what a result is a function of: two trackers with different values agree to summation noise, not bit for
   * bit; under ONE value match(), the batched forms, the queue, the validator's workers and every band count agree bit for bit
   * (tests/test_determinism.py runs under both).
   *   DVO_AMD_GEOMETRY_THROUGHPUT (default): 640x480 levels 3..0 run 4 / 4 / 10 / 10 steps per wave -- on a level of 64 000
   *     pixels or more a wave segment holds as many points as an image row has pixels (ten steps for a 640-pixel row, twenty for
   *     1280; 16 steps where a row is no whole number of steps): long segments amortise a block's prologue and epilogue, and the
   *     four waves of a block, about a row apart, share the lines they gather -- the most pairs per second;
   *   DVO_AMD_GEOMETRY_LATENCY: 1 / 2 / 2 / 4 -- short segments spread a coarse level over more waves: the shortest single
   *     match() (the reference's default deployment is one match() per frame, dvo_ros/src/camera_dense_tracking.cpp:269), a few
   *     per cent fewer pairs per second in large batches. */
  int segment_geometry;
  int reserved;
} dvo_amd_config;
#define DVO_AMD_GEOMETRY_THROUGHPUT 0
#define DVO_AMD_GEOMETRY_LATENCY 1

/* DenseTracker::IterationStats, dense_tracking.h:83-100 */
typedef struct {
  int id;
  int valid_constraints;
  double tdist_loglik;
  double tdist_mean[2];
  double tdist_precision[4];  /* column-major 2x2 */
  double prior_loglik;
  double increment[6];        /* EstimateIncrement (upsilon, omega); valid if has_increment */
  double information[36];     /* EstimateInformation; valid if has_increment */
  int has_increment;          /* 0 for the iteration a level broke out of (TooFewConstraints / LogLikelihoodDecreased) */
  int reserved;
  /* instrumentation (not in the reference): estimate().matrix() of this iteration, column-major 4x4 -- the transform whose
   * float cast the residual stage used (dense_tracking.cpp:263).  The parity tests replay single iterations from it. */
  double estimate[16];
  double initial[16];         /* likewise initial() of this iteration (dense_tracking.cpp:260,302,346: the prior term) */
} dvo_amd_iteration_stats;

/* DenseTracker::LevelStats, dense_tracking.h:103-116 */
typedef struct {
  int id;
  int max_valid_pixels;
  int valid_pixels;
  int termination;      /* dvo_amd_termination */
  int n_iterations;
  int first_iteration;  /* index of the level's first entry in dvo_amd_result.iterations */
} dvo_amd_level_stats;

/* DenseTracker::Result, dense_tracking.h:125-140 */
typedef struct {
  double transformation[16];
  double information[36];
  double loglik;
  int is_nan;                               /* Result::isNaN(), dense_tracking_config.cpp:96 */
  int n_levels;
  dvo_amd_level_stats levels[DVO_AMD_MAX_LEVELS];
  int n_iterations;                         /* entries written */
  int iterations_capacity;                  /* in: size of iterations[] (0 / NULL: per-iteration stats are dropped) */
  dvo_amd_iteration_stats *iterations;      /* caller-provided */
  /* instrumentation (not in the reference) */
  int n_ticks;                              /* host<->device round trips spent */
  int n_residual_passes;                    /* fused warp+residual+normal-equation launches (incl. discarded speculative ones) */
  double alg_bytes;                         /* 56 B x selected points x residual passes */
  double alg_bytes_discarded;               /* the part of alg_bytes spent on speculative passes whose iteration was rolled back */
} dvo_amd_result;

typedef struct dvo_amd_context dvo_amd_context; /* one DenseTracker instance: one HIP stream + scratch; NOT thread-safe */
typedef struct dvo_amd_pyramid dvo_amd_pyramid; /* one RgbdImagePyramid: refcounted, immutable after create, shareable */

int dvo_amd_abi_version(void);
/* The build this binary is: the first 16 hex digits of the sha256 over the compiler flags and every source file and header of
 * the library (dvo_slam_amd/_build.py: source_id; the Makefile passes the same string).  The Python binding refuses a library
 * whose id is not the hash of the sources next to it, and rebuilds it where hipcc exists.  "unknown" for a hand-made build. */
const char *dvo_amd_build_id(void);
const char *dvo_amd_status_string(int status);
/* text of the most recent HIP failure on the calling thread ("" if none) */
const char *dvo_amd_last_error(void);
int dvo_amd_device_count(void);

/* DenseTracker::Config::Config(), dense_tracking_config.cpp:27-41 */
void dvo_amd_default_config(dvo_amd_config *cfg);

/* DenseTracker::DenseTracker(cfg) / configure(), dense_tracking.cpp:54-97 */
int dvo_amd_context_create(int device, const dvo_amd_config *cfg, dvo_amd_context **out);
void dvo_amd_context_destroy(dvo_amd_context *ctx);
/* the HIP device the context was created on */
int dvo_amd_context_device(const dvo_amd_context *ctx, int *device);

/* The reciprocal the warp stage and the t-distribution weights use (dense_tracking_impl.cpp:192,700: _mm_rcp_ps, a ~12-bit
 * approximation whose bits differ between CPU vendors).
 *   DVO_AMD_RCP_EXACT (default): 1 / z of the projection is the exactly truncated quotient, the weight's reciprocal is within
 *     1 ulp: the same on every machine.
 *   DVO_AMD_RCP_HOST_SSE: both are THIS HOST's _mm_rcp_ps, bit for bit, from a table probed on the host when the mode is
 *     switched on (2^11 or 2^12 entries on the Xeons / EPYCs seen so far): residuals and validity decisions are then
 *     bit-identical to the reference's SSE path as this host runs it, and so is every t-distribution weight: the body of a
 *     pass as 7 rcpps(5 + d) with d formed product by product (:669-700), the last V mod 4 by computeWeight's exact division
 *     (:702-706, Q7 -- a small kernel between the two kernels of a tick finds those pixels once the pass has counted V and
 *     leaves what their exact weights add to the pair sums and the moments).  The one exception: a pair tile-sharded over
 *     SEVERAL GPUs keeps the table's weight for those <= 3 pixels (no rank sees the whole level's counts before the
 *     exchange).  Sums are still taken in this library's order, so whole-match parity stays at tolerance level.  Returns
 *     DVO_AMD_ERR_INVALID_ARGUMENT with a reason in dvo_amd_last_error() if the host's instruction does not have the
 *     structure the table assumes, or under DVO_AMD_ACCUM=valu (the mode is built for the default accumulator only); refused
 *     while pairs are queued.
 * DVO_AMD_RCP=host in the environment makes it the default of every new context. */
#define DVO_AMD_RCP_EXACT 0
#define DVO_AMD_RCP_HOST_SSE 1
int dvo_amd_set_reciprocal_mode(dvo_amd_context *ctx, int mode);
int dvo_amd_get_reciprocal_mode(const dvo_amd_context *ctx, int *mode, int *table_mantissa_bits);
int dvo_amd_configure(dvo_amd_context *ctx, const dvo_amd_config *cfg);
int dvo_amd_get_config(const dvo_amd_context *ctx, dvo_amd_config *cfg);

/*
 * RgbdCameraPyramid(w,h,K).create(intensity, depth) + RgbdImagePyramid::build(levels) + everything match() would build
 * lazily (derivatives, point-cloud rays, gather layout): rgbd_image.cpp:141-172,245-296,419-489,534-543.
 * stride is in floats (>= width).  The host overload copies the two base planes H2D; the device overload reads planes that
 * are already resident in HBM on `device` (no PCIe traffic).  All work is enqueued on an internal stream and complete on return.
 */
int dvo_amd_pyramid_create(int device, const float *intensity, const float *depth, int width, int height, int stride,
                           float fx, float fy, float ox, float oy, int levels, double timestamp, dvo_amd_pyramid **out);
int dvo_amd_pyramid_create_from_device(int device, const float *d_intensity, const float *d_depth, int width, int height,
                                       int stride, float fx, float fy, float ox, float oy, int levels, double timestamp,
                                       dvo_amd_pyramid **out);
/*
 * Frame ingest on the device: the pyramid straight from a raw sensor frame, replacing the host-side
 * cv::cvtColor(CV_BGR2GRAY) + convertTo(CV_32F) (benchmark_slam.cpp:60-68, camera_dense_tracking.cpp:219-229) and
 * SurfacePyramid::convertRawDepthImageSse (surface_pyramid.cpp:65-105) in front of RgbdCameraPyramid::create.
 *   image : uint8, `channels` = 1 (gray) or 3 (B,G,R interleaved); image_stride_bytes >= width * channels
 *   depth : uint16, 0 = no measurement -> NaN, else (float)raw * depth_scale (1/5000 for TUM PNGs, 0.001 for OpenNI)
 *   on_device != 0: both pointers are device memory on `device` (no PCIe traffic); else host memory (5 B/px cross PCIe
 *   instead of the 8 B/px of two float planes).
 * Gray conversion is OpenCV's 8-bit rule Y = (1868 B + 9617 G + 4899 R + 8192) >> 14.
 */
int dvo_amd_pyramid_create_raw(int device, const unsigned char *image, int channels, int image_stride_bytes,
                               const unsigned short *depth, int depth_stride, float depth_scale, int on_device, int width,
                               int height, float fx, float fy, float ox, float oy, int levels, double timestamp,
                               dvo_amd_pyramid **out);
void dvo_amd_pyramid_retain(dvo_amd_pyramid *p);
void dvo_amd_pyramid_release(dvo_amd_pyramid *p);
int dvo_amd_pyramid_levels(const dvo_amd_pyramid *p);
double dvo_amd_pyramid_timestamp(const dvo_amd_pyramid *p);
/* RgbdImagePyramid::level(l): size and intrinsics {fx,fy,ox,oy} of a level */
int dvo_amd_pyramid_level_info(const dvo_amd_pyramid *p, int level, int *width, int *height, float k[4]);
/* download one plane of a level: 0 I, 1 Z, 2 Ix, 3 Iy, 4 Zx, 5 Zy (RgbdImage::intensity, depth, *_dx, *_dy) */
int dvo_amd_pyramid_download_plane(const dvo_amd_pyramid *p, int level, int plane, float *dst);
/* PointSelection::select(level) (point_selection.cpp:89-117): number of selected pixels for the thresholds, and optionally
 * the per-pixel mask (1 = selected, row-major) */
int dvo_amd_pyramid_select(dvo_amd_pyramid *p, int level, float intensity_threshold, float depth_threshold, int *count,
                           unsigned char *mask);

/* DenseTracker::match(RgbdImagePyramid& reference, RgbdImagePyramid& current, Result&), dense_tracking.cpp:123-376.
 * T_init (may be NULL) is read only if use_initial_estimate (dense_tracking.cpp:137-144). */
int dvo_amd_match(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, const double *T_init,
                  dvo_amd_result *result);

/* DenseTracker::match(PointSelection& reference, RgbdImagePyramid& current, Result&), dense_tracking.cpp:131-376: the
 * reference pixels are the ones the PointSelection's own predicate keeps (point_selection.h:49-67: z, zdx, zdy valid and a
 * gradient above the thresholds; thresholds of -1 reproduce ValidPointPredicate), whatever thresholds the tracker's
 * configuration holds.  Selections are cached per (pyramid, threshold pair), like a PointSelection caches per level. */
int dvo_amd_match_selection(dvo_amd_context *ctx, dvo_amd_pyramid *reference, float intensity_threshold, float depth_threshold,
                            dvo_amd_pyramid *current, const double *T_init, dvo_amd_result *result);

/* n independent match() calls advanced in lock step on one GPU (the shape of LocalTracker::update's tbb::parallel_invoke,
 * local_tracker.cpp:184, and of the loop-closure validator's parallel_reduce, keyframe_graph.cpp:576-593).
 * T_inits: n x 16 doubles or NULL.  Results are identical to n dvo_amd_match() calls BIT FOR BIT: a pair's result is a function
 * of its inputs and the configuration alone -- the reference's guarantee for match() calls that run side by side under TBB
 * (keyframe_graph.cpp:587-590, local_tracker.cpp:184).  The same holds for dvo_amd_match_many at any max_in_flight, for the
 * submit / wait / poll queue whatever else is queued, for dvo_amd_validate_proposals at any number of workers, and for
 * dvo_amd_match_banded / dvo_amd_match_sharded at 1, 2, 4, 8 and 16 bands (tests/test_determinism.py): the geometry of a residual
 * pass is the pyramid level's own and every sum across blocks follows one tree per level (csrc/dvo_types.h). */
int dvo_amd_match_batch(dvo_amd_context *ctx, int n, dvo_amd_pyramid *const *references, dvo_amd_pyramid *const *currents,
                        const double *T_inits, dvo_amd_result *results);

/* The same n alignments with at most max_in_flight of them resident at a time: a pair that finishes hands its slot to the
 * next pending pair, so every launch stays full although pairs need different numbers of iterations (the shape of the
 * loop-closure validator working through a proposal list, keyframe_graph.cpp:576-593).  max_in_flight <= 0: all n at once. */
int dvo_amd_match_many(dvo_amd_context *ctx, int n, dvo_amd_pyramid *const *references, dvo_amd_pyramid *const *currents,
                       const double *T_inits, dvo_amd_result *results, int max_in_flight);

/*
 * The same queue without draining between calls.  A tracker that works through proposals as they come (the loop-closure
 * validator's tbb::parallel_reduce with grain 1 over a proposal list, keyframe_graph.cpp:587-590, called again for every new
 * keyframe, :434-498) keeps `max_in_flight` pairs resident ACROSS calls: dvo_amd_match_submit appends n pairs to the
 * context's queue and returns at once (pairs start as soon as a slot is free), dvo_amd_match_wait drives the queue until every
 * pair of that submission is finished (ticket 0: everything submitted so far), dvo_amd_match_poll advances whatever has
 * landed without waiting for the GPU and reports whether the submission is complete.  dvo_amd_match_many(...) is
 * submit + wait.  Contract: one host thread per context, as for every entry point; `results` (and their iteration arrays)
 * stay valid and untouched until the submission is complete; the queue retains the pyramids itself; the tracker's
 * configuration must not change while pairs are queued (dvo_amd_configure refuses); a different max_in_flight or larger
 * frames than the queue was laid out for let the queued pairs run to completion first.  Results are those of n dvo_amd_match()
 * calls, bit for bit.  While pairs are queued every entry point that works outside the queue in the context's scratch
 * (dvo_amd_match_banded / _sharded, dvo_amd_match_selection, dvo_amd_residuals, dvo_amd_error_image, the debug and bench
 * entries) returns DVO_AMD_ERR_INVALID_ARGUMENT, like dvo_amd_configure.  If a tick fails, every queued pair is dropped,
 * nothing of the context is running any more when the error is returned, and the wait / poll of each submission that was still
 * open then returns that status (a submission that had completed before keeps its OK; ticket 0 returns a failure that no
 * wait / poll has reported yet, once).
 */
int dvo_amd_match_submit(dvo_amd_context *ctx, int n, dvo_amd_pyramid *const *references, dvo_amd_pyramid *const *currents,
                         const double *T_inits, dvo_amd_result *results, int max_in_flight, unsigned long long *ticket);
int dvo_amd_match_wait(dvo_amd_context *ctx, unsigned long long ticket);
int dvo_amd_match_poll(dvo_amd_context *ctx, unsigned long long ticket, int *done);

/*
 * One pair tile-sharded over several GPUs (BASELINE config 4).  Every rank holds both pyramids and processes one band of
 * scan-order blocks of every level (whole chunks of the level's summation tree); per Gauss-Newton tick the ranks all-gather
 * one 784-byte record per band over RCCL and fold them along that tree, so every rank runs the identical state machine and
 * returns the identical result -- for 1, 2, 4, 8 and 16 ranks the result of dvo_amd_match() on one GPU, bit for bit (other
 * rank counts: to the rounding of the fp64 sums).
 *   id = dvo_amd_comm_unique_id() on rank 0, broadcast by the caller (torch.distributed, MPI, a file ...);
 *   dvo_amd_comm_create(ctx, id, nranks, rank) on every rank;  dvo_amd_match_sharded(...) on every rank, same arguments.
 * dvo_amd_match_banded runs the same band pipeline with all n_bands bands on ONE GPU (no communicator): it is how the band
 * logic is verified against the unsharded path on a single-GPU box.
 */
int dvo_amd_comm_unique_id(unsigned char *id128);
int dvo_amd_comm_create(dvo_amd_context *ctx, const unsigned char *id128, int nranks, int rank);
void dvo_amd_comm_destroy(dvo_amd_context *ctx);
/* The same exchange without a collective (SURVEY.md 8e "implementation note"): every rank maps every peer's exchange buffer
 * (hipIpc, fine-grained device memory) and its finalize record of a tick is written straight into all of them over xGMI,
 * payload first, a sequence word last; one small kernel per tick pushes, waits (bounded) for the peers' records and forwards
 * them to pinned host memory the host polls: one hop, no D2H copy, no stream synchronisation, deterministic fold in rank
 * order.  dvo_amd_exchange_create on every rank returns the 64-byte handle of its buffer; the caller all-gathers the handles
 * (torch.distributed, MPI, a file ...) and passes all nranks x 64 bytes, in rank order, to dvo_amd_exchange_attach.  Once
 * attached, dvo_amd_match_sharded uses this path; the RCCL communicator above stays available as the fallback. */
int dvo_amd_exchange_create(dvo_amd_context *ctx, int nranks, int rank, unsigned char *handle64);
int dvo_amd_exchange_attach(dvo_amd_context *ctx, const unsigned char *handles);
void dvo_amd_exchange_destroy(dvo_amd_context *ctx);
int dvo_amd_match_sharded(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, const double *T_init,
                          dvo_amd_result *result);
int dvo_amd_match_banded(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, const double *T_init,
                         dvo_amd_result *result, int n_bands);

/*
 * Batched 2-stage loop-closure validation: dvo_slam::constraints::ConstraintProposalValidator::validate
 * (constraint_proposal_validator.cpp:69-166) with the voters of constraint_proposal_voter.cpp:34-186, fed by ONE call: every
 * stage aligns all of its proposals (and the cross-validation inverses) as one batch on the GPU (dvo_amd_match_many), in
 * place of the tbb::parallel_reduce over single proposals in keyframe_graph.cpp:525-593.
 */
/* dvo_slam::TrackingResultEvaluation subclasses (tracking_result_evaluation.cpp:54-67): value(r) */
typedef enum {
  DVO_AMD_EVAL_LOGLIKELIHOOD = 0,            /* -r.LogLikelihood (the one KeyframeTracker creates, keyframe_tracker.cpp:95) */
  DVO_AMD_EVAL_NORMALIZED_LOGLIKELIHOOD = 1, /* -r.LogLikelihood / Levels.back().Iterations.back().ValidConstraints */
  DVO_AMD_EVAL_ENTROPY = 2                   /* log(det(r.Information)) */
} dvo_amd_evaluation_kind;

/* dvo_slam::Keyframe as the validator reads it (id(), image(), pose(), evaluation()) */
typedef struct {
  int id;
  dvo_amd_pyramid *image;
  double pose[16];               /* column-major 4x4 */
  int evaluation_kind;           /* dvo_amd_evaluation_kind */
  double evaluation_average;     /* TrackingResultEvaluation::average_ (sum of the values added so far) */
  double evaluation_n;           /* TrackingResultEvaluation::n_ */
} dvo_amd_keyframe;

typedef enum {
  DVO_AMD_VOTER_ODOMETRY_CONSTRAINT = 0,        /* reject |ref.id - cur.id| <= 1                    (voter.cpp:167-184) */
  DVO_AMD_VOTER_NAN_RESULT = 1,                 /* reject TrackingResult.isNaN()                     (voter.cpp:147-162) */
  DVO_AMD_VOTER_CONSTRAINT_RATIO = 2,           /* ValidConstraints / ValidPixels >= threshold       (voter.cpp:124-142) */
  DVO_AMD_VOTER_TRACKING_RESULT_EVALUATION = 3, /* ratioWithAverage >= threshold, Score = ratio      (voter.cpp:103-119) */
  DVO_AMD_VOTER_CROSS_VALIDATION = 4            /* |t(T_inverse * T)| <= threshold; adds the inverse proposals (voter.cpp:34-97) */
} dvo_amd_voter_kind;

#define DVO_AMD_MAX_VOTERS 8
typedef struct {
  int kind;          /* dvo_amd_voter_kind */
  double threshold;
} dvo_amd_voter;

/* ConstraintProposalValidator::Stage (constraint_proposal_validator.h) */
typedef struct {
  int id;
  int only_keep_best;               /* keepBest() / keepAll() */
  dvo_amd_config tracking_config;
  int n_voters;
  dvo_amd_voter voters[DVO_AMD_MAX_VOTERS];
} dvo_amd_validator_stage;

/* ConstraintProposal::Vote; `value` is the quantity the voter tested (what the reference prints into Vote::Reason) */
typedef struct {
  int voter_kind;
  int reject;       /* Vote::Decision: 0 Accept, 1 Reject */
  double score;
  double value;
} dvo_amd_vote;

/* dvo_slam::constraints::ConstraintProposal: reference / current are indices into the keyframe array */
typedef struct {
  int reference, current;
  double initial_transformation[16];
  dvo_amd_result tracking_result;  /* of the last stage the proposal went through; iterations is ignored (set to NULL) */
  int n_votes;
  dvo_amd_vote votes[DVO_AMD_MAX_VOTERS];
  /* instrumentation (not in the reference), out only: which input proposal a survivor descends from -- its index i in the
   * array passed to dvo_amd_validate_proposals, or -(i + 1) if it is that proposal's cross-validation inverse.  Lets a checker
   * compare a survivor with the alignment of the SAME (reference, current, initial transformation) on its own side. */
  int origin;
  int reserved;
  /* instrumentation, out only: the initial transformation the LAST stage aligned this proposal from (initial_transformation
   * above is updated behind every stage to the inverse of the stage's estimate, validator.cpp:95-100, and the inverse proposals
   * of the cross-validation are formed inside the call): what a checker needs to repeat exactly this alignment */
  double stage_initial_transformation[16];
} dvo_amd_constraint_proposal;

/* the two stages KeyframeGraph builds (keyframe_graph.cpp:500-523) with the tracker configs of configureValidationTracking
 * (:819-838): stage 1 = level 3 only, keepAll, {Odometry, NaN, ConstraintRatio(min_constraint_ratio), Evaluation(ratio_coarse),
 * CrossValidation(1.0)}; stage 2 = levels 3..1, keepBest, {NaN, ConstraintRatio, Evaluation(ratio_fine)} */
void dvo_amd_default_validator_stages(const dvo_amd_config *frontend_cfg, double min_constraint_ratio, double ratio_coarse,
                                      double ratio_fine, dvo_amd_validator_stage stages[2]);
/* the initial proposal list of validateKeyframeConstraintsParallel (keyframe_graph.cpp:577-585): for every candidate one
 * proposal with identity and one with the relative pose current.pose^-1 * reference.pose.  proposals: 2 * n_candidates */
int dvo_amd_proposals_for_candidates(const dvo_amd_keyframe *keyframes, int keyframe, int n_candidates, const int *candidates,
                                     dvo_amd_constraint_proposal *proposals);
/* validate(): proposals[0..n_proposals) in, survivors compacted to the front in the reference's order, *n_out of them.
 * The context's tracker configuration is restored before returning.  max_in_flight as in dvo_amd_match_many.  A stage of many
 * alignments is dealt over a few worker contexts of the library's own (same device, created on first use, own host threads for
 * the duration of the call: keyframe_graph.cpp:576-593 deals the proposals over TBB workers the same way). */
int dvo_amd_validate_proposals(dvo_amd_context *ctx, int n_keyframes, const dvo_amd_keyframe *keyframes, int n_stages,
                               const dvo_amd_validator_stage *stages, int n_proposals,
                               dvo_amd_constraint_proposal *proposals, int *n_out, int max_in_flight);

/*
 * Dual-match front-end step: the two alignments LocalTracker::update runs per frame with
 * tbb::parallel_invoke (local_tracker.cpp:170-186) -- keyframe -> frame starting from last_keyframe_pose^-1 and
 * last frame -> frame starting from identity -- as ONE two-pair batch sharing the frame's pyramid, plus the quantities the
 * accept callbacks of KeyframeTracker test on the two results (keyframe_tracker.cpp:105-190).
 */
typedef struct {
  int odometry_is_nan, keyframe_is_nan;   /* force a new keyframe (local_tracker.cpp:191) */
  double odometry_translation_norm;       /* onAcceptCriterionEstimateDivergence: rejects > 0.1 */
  double keyframe_translation_norm;       /* ... and > 1.5 MaxTranslationalDistance; onAcceptCriterionDistance */
  double keyframe_constraint_ratio;       /* onAcceptCriterionConstraintRatio: Levels.back().Iterations.back().ValidConstraints / ValidPixels */
  double odometry_neg_loglik;             /* onAcceptCriterionTrackingResultEvaluation: -LogLikelihood (value() of the evaluation) */
  double keyframe_neg_loglik;
  double odometry_condition_number;       /* onAcceptCriterionConditionNumber: |lambda_max / lambda_min| of Information */
  double keyframe_condition_number;
} dvo_amd_frame_criteria;
/* last_keyframe_pose: column-major 4x4 (LocalTrackerImpl::last_keyframe_pose_), NULL = identity.  r_keyframe / r_odometry
 * follow dvo_amd_match's conventions; criteria may be NULL. */
int dvo_amd_track_frame(dvo_amd_context *ctx, dvo_amd_pyramid *keyframe, dvo_amd_pyramid *last_frame, dvo_amd_pyramid *frame,
                        const double *last_keyframe_pose, dvo_amd_result *r_keyframe, dvo_amd_result *r_odometry,
                        dvo_amd_frame_criteria *criteria);

/*
 * TUM RGB-D benchmark on-disk formats, host code only.
 * The reference reads frames with cv::imread (benchmark_slam.cpp:50-51): the colour image as 3-channel 8-bit BGR
 * (flag 1), the depth image unchanged (flag -1, 16-bit gray in TUM sequences).  OpenCV is not available here, so a PNG
 * decoder (zlib inflate + the five PNG filters; 8/16-bit gray, gray+alpha, RGB, RGBA, and 1/2/4/8-bit palette or gray;
 * non-interlaced only) stands in for those two calls.  16-bit colour samples keep their high byte, alpha is dropped,
 * gray is replicated to B=G=R: what imread(.., 1) returns.
 */
int dvo_amd_png_info(const char *path, int *width, int *height, int *channels, int *bit_depth);
int dvo_amd_png_read_bgr8(const char *path, unsigned char *dst, int width, int height);      /* width*height*3 bytes */
int dvo_amd_png_read_gray16(const char *path, unsigned short *dst, int width, int height);   /* gray PNGs only; 8-bit values are widened */
/* One line of the estimated trajectory exactly as benchmark_slam.cpp:490-504 / map_serializer.cpp:61-66 print it:
 * "<sec>.<nsec, 9 digits> tx ty tz qx qy qz qw \n" with the default ostream formatting of doubles (%g, 6 significant
 * digits), the stamp split like ros::Time::fromSec, the quaternion as Eigen::Quaterniond(rotation) builds it.
 * T: column-major 4x4.  Returns the number of characters written (excluding the terminator), or -1 if capacity is too small. */
int dvo_amd_format_trajectory_line(double timestamp, const double *T, char *buf, int capacity);

/* dvo::core::computeResidualsAndValidFlagsSse (dense_tracking_impl.cpp:400-403) for one level and one float transform
 * (column-major 4x4, reference -> current).  residuals: width*height x 2 floats in pixel order, NaN where the pixel is not
 * selected or its warp is invalid.  Used by the parity tests and by dvo_amd_error_image. */
int dvo_amd_residuals(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, int level,
                      const float *T, float *residuals, int *n_valid);
/* DenseTracker::computeIntensityErrorImage, dense_tracking.cpp:378-444: |intensity residual| per reference pixel, 0 elsewhere */
int dvo_amd_error_image(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, const double *T,
                        int level, float *image);


/* Host-side helpers (no GPU needed): the SE(3) exponential / logarithm with Sophus' tangent order (upsilon, omega) and the
 * pivoted LDL^T 6x6 solve the driver uses in place of Sophus::SE3d::exp/log and Eigen::LDLT (dense_tracking.cpp:238,259,347).
 * Exported so that bindings do not need Sophus to build a T_init or to compare poses. */
void dvo_amd_se3_exp(const double *xi, double *T);
void dvo_amd_se3_log(const double *T, double *xi);
void dvo_amd_solve6(const double *A, const double *b, double *x);


/* Test probes, diagnostics and micro-benchmarks (dvo_amd_debug_*, dvo_amd_bench_*, dvo_amd_kernel_timing) are exported by the
 * same library but declared in dvo_amd_debug.h: they are scaffolding of this repository's tests and bench, not part of the
 * boundary a caller of the reference would bind. */

#ifdef __cplusplus
}
#endif
#endif
