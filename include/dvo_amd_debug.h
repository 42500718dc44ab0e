/*
 * dvo_amd_debug.h -- test probes, diagnostics and micro-benchmarks of libdvo_amd.so.
 *
 * NOT part of the drop-in boundary (that is dvo_amd.h: what a caller of dvo::DenseTracker / RgbdImagePyramid binds).  These
 * entry points exist for this repository's parity tests (stage probes, the receiving side of the record hand-off, band folds),
 * its bench (kernel timing, the isolated residual pass) and its profiling scripts (block trace, reducer stamps).  They are
 * exported by the same library, follow the same conventions (int status codes, nothing thrown) and may change without an ABI
 * version bump.
 */
#ifndef DVO_AMD_DEBUG_H_
#define DVO_AMD_DEBUG_H_

#include "dvo_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* (diagnostic; libraries built with -DDVO_TRACE_BLOCKS only, else -1) the per-block trace of every k_tick block since the last
 * call: 4 x 64-bit words per block {start, after the first step, end (100 MHz clock), info | hw_id << 32}; returns the number of
 * blocks recorded and resets the trace.  scripts/block_trace.py */
long long dvo_amd_debug_block_trace(dvo_amd_context *ctx, unsigned long long *out, long long capacity_blocks);
/* (test entry) out[i] = the table reciprocal of in[i] as the kernels compute it; needs DVO_AMD_RCP_HOST_SSE */
int dvo_amd_debug_rcp(dvo_amd_context *ctx, int n, const float *in, float *out);

/* host-only: the ordered combine of band records {valid, first_w, last_r0, last_r1, S[3], S_odd[3]} (10 doubles per band)
 * -> {valid, S[3], S_odd[3]}; exported for the CPU tests of the multi-GPU path */
int dvo_amd_debug_combine_bands(int n_bands, const double *bands, double *out);
/* host-only: the receiving side of the record hand-off.  A tick's 784-byte record reaches the host (and, for a sharded pair,
 * the peers) as `*n_pieces` pieces of 16 bytes -- two 8-byte halves {payload word, tick number} -- each written by one store
 * of the device; a piece counts when BOTH its tags are the tick waited for, in whatever order the pieces arrive (so a piece
 * that should ever arrive as two 8-byte halves is simply not accepted until both are there).
 * dvo_amd_debug_wire_layout reports the piece and payload-word counts; dvo_amd_debug_take_wire copies the payload of the
 * pieces from `from_piece` on that carry `tick` out of `wire` (16-byte aligned, 4 words per piece) into `record_words`
 * and returns the index of the first piece that does not (n_pieces when the record is complete), or minus an error code.
 * Exported for the CPU tests. */
/* host-only: the successor of a tick number (32 bits; never 0, the value of a fresh buffer; the wrap keeps the parity alternating) */
unsigned dvo_amd_debug_next_seq(unsigned seq);
int dvo_amd_debug_wire_layout(int *n_pieces, int *n_record_words);
int dvo_amd_debug_take_wire(const unsigned *wire, unsigned tick, int from_piece, unsigned *record_words);

/* Stage-wise probe of ONE Gauss-Newton iteration body at a fixed pose (dense_tracking.cpp:271-347 without the accept test and
 * the solve): computeResidualsSse, computeWeightsSse (unit weights when precision_in is NULL = first iteration of a level,
 * else the t-distribution weights of the column-major 2x2 precision_in), computeScaleSse + inverse, the normal equations
 * (Mu = 0) and computeCompleteDataLogLikelihood, through exactly the kernels and host arithmetic dvo_amd_match uses (two
 * ticks).  Exists so that the parity tests can compare the weighted stages (iterations k >= 1) with their CPU checker directly
 * instead of only through the final pose. */
typedef struct {
  int valid_constraints;
  int reserved;
  double scale_sums[3];   /* unscaled pair sums (xx, xy, yy) of computeScaleSse incl. the Q5 pairing */
  float scale[4];         /* column-major 2x2: sums / (V - 3) */
  float precision[4];     /* its inverse (Eigen Matrix2f::inverse) */
  double moments[87];     /* the P-free sums (layout: dvo_types.h kAcc*) */
  double information[36]; /* A = sum w J^T P J, column-major 6x6 */
  double rhs[6];          /* b = -sum w J^T P r */
  double loglik_sum;      /* sum of log(1 + 0.2 r^T P r) over the first 50 floor(V/50) valid residuals (Q6) */
  float loglik;           /* what computeCompleteDataLogLikelihood returns */
  float reserved_f;
} dvo_amd_iteration_probe;
/* precision_eval (column-major 2x2, may be NULL): evaluate information / rhs / loglik with this precision instead of the one the
 * probe computed itself (out->precision is the computed one either way) */
int dvo_amd_debug_iteration(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, int level,
                            const float *T, const float *precision_in, const float *precision_eval,
                            dvo_amd_iteration_probe *out);

/* Host-rcpps mode only (dvo_amd_set_reciprocal_mode): the t-distribution weights of ONE residual pass at the float transform T
 * under the column-major 2x2 precision_in, as the product kernels form them -- the residual pass stores every pixel's weight
 * (weights[n pixels of the level], NaN where the pixel is no constraint) -- together with what k_q7_tail did about the pass's
 * last V mod 4 pixels (Q7, dense_tracking_impl.cpp:702-706: computeWeightsSse divides exactly there): which pixels, the weight
 * the pass gave them, computeWeight's, and what the difference adds to the pair sums and the 87 moments (k_finalize has added it
 * to the record by the time this returns).  The parity tests compare all of it with computeWeightsSse as this host runs it. */
typedef struct {
  int n_tail;             /* V mod 4 */
  int valid_constraints;  /* V of the record */
  int valid_counted;      /* V as k_q7_tail counted it from the block records */
  int recomputed_equal;   /* the tail pixels' recomputed residuals are the spilled ones, bit for bit */
  int pixel[3];           /* index in the level's scan order, ascending; -1 beyond n_tail */
  float weight_table[3];  /* 7 rcpps(5 + d) */
  float weight_exact[3];  /* (float)((2.0 + 5.0f) / (5.0f + d)) */
  double scale_sums_delta[3];
  double moments_delta[87];
} dvo_amd_q7_probe;
int dvo_amd_debug_weights(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, int level, const float *T,
                          const float *precision_in, float *weights, dvo_amd_q7_probe *tail);

/* Diagnostic: the hardware queue the context's main stream runs on, asked of the GPU by a one-wave kernel on that stream (HW_ID:
 * pipe << 3 | queue).  The runtime maps streams onto its four hardware queues and a hardware queue runs one kernel at a time, so
 * how the trackers of a GPU are spread over them decides up to a third of a batch's throughput
 * (profiles/r05_stream_queue_assignment_ab.txt); the library takes the stream the runtime deals it (create the
 * trackers of a GPU back to back) and this entry lets an integrator look at the outcome.  Refused while pairs are queued. */
int dvo_amd_debug_hw_queue(dvo_amd_context *ctx, int *pipe_queue);

/* (test entry) the residual pass's geometry on one pyramid level under the context's configuration: `steps` 64-point steps per
 * wave segment -- what dvo_amd_config::segment_geometry documents (dvo_amd.h), a function of the level's size and the configuration
 * alone --, `points` the points the pass walks (the pixels the configuration's thresholds select, without an odd trailing one) and
 * `blocks` the blocks of four segments that cover them (tests/test_determinism.py pins the tables). */
int dvo_amd_debug_level_geometry(dvo_amd_context *ctx, dvo_amd_pyramid *reference, int level, int *steps, int *blocks, int *points);

/* Micro-benchmark of the dominant kernel alone (used by bench.py for the roofline figure and by the tuning scripts):
 * `reps` timed repetitions of the fused residual pass over `n_items` copies of one (reference level, current level) pair
 * at the float transform T (column-major 4x4), `rounds` 256-pixel rounds (four 64-pixel steps each) per wave segment (1, 2, 4, 8 or 16; 0 = the driver's choice).
 * avg_ms: HIP-event time per repetition on the context's stream; alg_bytes: 56 B x selected points x n_items;
 * n_launches: kernel launches one repetition needs (the argument block holds 62 items). */
int dvo_amd_bench_residual_pass(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, int level,
                                const float *T, int n_items, int rounds, int reps, double *avg_ms, double *alg_bytes,
                                int *n_launches);

/* the same over n_items DIFFERENT (reference, current) pairs (same size): no two items of a launch read the same planes, so
 * nothing is deduplicated by the caches */
int dvo_amd_bench_residual_pass_pairs(dvo_amd_context *ctx, int n_items, dvo_amd_pyramid *const *references,
                                      dvo_amd_pyramid *const *currents, int level, const float *T, int rounds, int reps,
                                      double *avg_ms, double *alg_bytes, int *n_launches);

/* Diagnostic: with DVO_AMD_FIN_STAMPS=1 in the environment the finalize kernel records 8 shader-clock stamps of its phases
 * (block 0 of the most recent launch); this reads them back. */
int dvo_amd_debug_finalize_stamps(dvo_amd_context *ctx, unsigned long long *stamps8);

/* Diagnostic: while dvo_amd_kernel_timing is enabled every k_tick launch is logged as 8 doubles {ms, items, residual-pass
 * blocks, likelihood blocks, grid.x, selected reference pixels of the residual items, 64-pixel wave steps of the residual
 * items, 64-pixel wave steps of the likelihood items}; this reads and clears the log (out may be NULL to query the count). */
int dvo_amd_debug_tick_log(dvo_amd_context *ctx, double *out, int capacity_records, int *n_records);

/* timing helper for bench.py: HIP-event milliseconds the context's stream spent in its residual-pass kernel since the last
 * reset, and the number of launches */
int dvo_amd_kernel_timing(dvo_amd_context *ctx, int enable, double *ms_residual_pass, long long *n_launches, int reset);

/* (diagnostic) which form of the host-rcpps mode the context's kernels run: 0 the mode is off, 1 a table in global memory (a gather
 * in the dependent chain of every step), 2 the device's own reciprocal of the cell midpoint plus 4-bit corrections from LDS (round
 * 5; bit-identical to form 1 by construction).  note: why form 2 is not in use ("" if it is or the mode is off). */
int dvo_amd_debug_rcp_form(const dvo_amd_context *ctx, int *form, char *note, int note_capacity);

/* (profiling aid) one no-op dispatch named `k_marker` on the context's main stream, waited for: bench.py brackets its
 * single-stream timing pass with two of them so that the summaries of a rocprofv3 run (kernel trace or counter pass) can select
 * exactly the k_tick dispatches in between (dvo_slam_amd/pmc.py) */
int dvo_amd_debug_marker(dvo_amd_context *ctx, unsigned tag);

/* (test entry) k_ll_overflow -- the exact answer to "did one of the reference's 50-term likelihood products overflow?"
 * (dense_tracking_impl.cpp:413-419) -- over a residual buffer the CALLER supplies, for one band of it: `residuals` = n_blocks x
 * 4 wave segments x (64 x steps) pixels x 2 floats in scan order, NaN = invalid pixel; the band = wave segments
 * [seg_first, seg_first + n_segs), rank_offset = valid pixels in earlier bands (the prefix table is built here from the NaN
 * pattern, relative to the band start, as k_finalize builds it); rank_end < 0: an open band (the walk may run on behind it);
 * rank_end >= 0: a CLOSED band of a pair sharded over several GPUs -- the residuals behind it are another rank's, whatever the
 * buffer holds there is stale and must not be looked at.  *overflowed = the kernel's verdict. */
int dvo_amd_debug_ll_overflow(dvo_amd_context *ctx, const float *residuals, int n_blocks, int steps, int seg_first, int n_segs,
                              int rank_offset, int rank_end, int cut_rank, const float *precision, int *overflowed);

#ifdef __cplusplus
}
#endif
#endif
