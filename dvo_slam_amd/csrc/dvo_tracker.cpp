// Host side of the MI355X dense tracking core: device pyramids, the Gauss-Newton driver and the C ABI (include/dvo_amd.h).
//
// The driver restates DenseTracker::match (dvo_core/src/dense_tracking.cpp:131-376) as a per-pair state machine that is
// advanced in "ticks".  One tick = one k_tick launch (+ one k_finalize) + one stream synchronisation, for all pairs of a
// batch at once.  Within a level the log-likelihood of iteration k (which needs the precision matrix of iteration k, which
// needs a global reduction over iteration k's residuals) is evaluated in the same launch as the residual pass of iteration
// k+1: the increment x_k is applied speculatively and rolled back if the likelihood test of iteration k fails
// (dense_tracking.cpp:312-322), which ends the level anyway.  So a level costs (iterations + 1) round trips.
//
// The 6x6 solve, SE(3) exp/log and the 2x2 inverse stay on the host (se3.h), as in the reference.
#include <dlfcn.h>
#include <emmintrin.h>  // the host side of the record hand-off takes 16 bytes at a time (x86-64 hosts)

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "dvo_internal.h"

namespace dvo_amd {
namespace host {

thread_local std::string g_last_error;

int fail_hip(const char *what, hipError_t e) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return e == hipErrorOutOfMemory ? DVO_AMD_ERR_OUT_OF_MEMORY : DVO_AMD_ERR_HIP;
}

dvo_amd_iteration_stats *stats_push(Job &j) {
  dvo_amd_result *r = j.result;
  dvo_amd_level_stats &ls = r->levels[r->n_levels - 1];
  ls.n_iterations++;
  if (j.recent_count == 2) j.recent[0] = j.recent[1], j.recent_count = 1;
  dvo_amd_iteration_stats *e = &j.recent[j.recent_count++];
  std::memset(e, 0, sizeof(*e));
  r->n_iterations++;
  return e;
}

// mirror the newest entry of `recent` into the caller's array (if it has room)
void stats_publish(Job &j) {
  dvo_amd_result *r = j.result;
  const int idx = r->n_iterations - 1;
  if (r->iterations && idx < r->iterations_capacity) r->iterations[idx] = j.recent[j.recent_count - 1];
}

void begin_iteration(Job &j, IterCtx &it, int k) {
  // dense_tracking.cpp:259-261
  it.k = k;
  std::memcpy(it.x_before, j.x, sizeof(j.x));
  it.inc = se3_exp(j.x);
  it.initial_before = j.initial;
  it.estimate_before = j.estimate;
  j.inc = it.inc;
  j.initial = se3_compose(se3_inverse(it.inc), j.initial);
  j.estimate = se3_compose(it.inc, j.estimate);
  it.initial_after = j.initial;
  it.estimate_after = j.estimate;
  it.buf = (k & 1) ^ j.buf_flip;
}

void finish_job(Job &j) {
  // dense_tracking.cpp:368-373
  dvo_amd_result *r = j.result;
  const dvo_amd_level_stats &last = r->levels[r->n_levels - 1];
  const int want = last.termination != DVO_AMD_TERM_LOGLIKELIHOOD_DECREASED ? last.n_iterations - 1 : last.n_iterations - 2;
  const dvo_amd_iteration_stats *e = nullptr;
  if (want >= 0) {
    const int back = (last.n_iterations - 1) - want;  // 0 or 1 entries before the newest
    if (back < j.recent_count) e = &j.recent[j.recent_count - 1 - back];
  }
  se3_matrix(se3_inverse(j.estimate), r->transformation);
  if (e && e->has_increment) {
    for (int i = 0; i < 36; ++i) r->information[i] = e->information[i] * 0.008 * 0.008;
    r->loglik = e->tdist_loglik + e->prior_loglik;
  } else {
    // the reference reads an IterationStats that was never filled (uninitialised Eigen storage) or indexes before the
    // start of the vector here; report NaN so that Result::isNaN() fires
    for (int i = 0; i < 36; ++i) r->information[i] = NAN;
    r->loglik = NAN;
  }
  double s = 0.0, si = 0.0;
  for (int i = 0; i < 16; ++i) s += r->transformation[i];
  for (int i = 0; i < 36; ++i) si += r->information[i];
  r->is_nan = !(std::isfinite(s) && std::isfinite(si));
  r->alg_bytes = 56.0 * j.alg_px;
  r->alg_bytes_discarded = 56.0 * j.discarded_px;
  j.done = true;
}

void start_level(Job &j);

void end_level(Job &j) {
  dvo_amd_result *r = j.result;
  dvo_amd_level_stats &ls = r->levels[r->n_levels - 1];
  // dense_tracking.cpp:359-363, evaluated after a break as well
  if (inf_norm6(j.x) <= j.cfg->precision) ls.termination = DVO_AMD_TERM_INCREMENT_TOO_SMALL;
  if (j.iteration >= j.cfg->max_iterations_per_level) ls.termination = DVO_AMD_TERM_ITERATIONS_EXCEEDED;
  j.have_a = j.have_b = false;
  j.level--;
  if (j.level < j.cfg->last_level)
    finish_job(j);
  else
    start_level(j);
}

void start_level(Job &j) {
  dvo_amd_result *r = j.result;
  dvo_amd_level_stats &ls = r->levels[r->n_levels++];
  const Selection &sel = *j.sel;
  const LevelData &L0 = j.ref->lv[0];
  ls.id = j.level;
  // PointSelection::getMaximumNumberOfPoints, point_selection.cpp:68-71
  ls.max_valid_pixels = (int)(size_t)((double)((size_t)L0.w * L0.h) * std::pow(0.25, (double)j.level));
  ls.valid_pixels = sel.count[j.level];
  ls.termination = DVO_AMD_TERM_UNSET;
  ls.n_iterations = 0;
  ls.first_iteration = r->n_iterations;
  j.level_first_iteration = r->n_iterations;
  j.recent_count = 0;
  j.iteration = 0;
  j.error = DBL_MAX;  // dense_tracking.cpp:209-210
  j.last_error = DBL_MAX;
  std::memset(j.precision, 0, sizeof(j.precision));
  se3_log(j.inc, j.x);  // :238 (Q1: re-applies the last applied or rejected increment)
  j.buf_flip = j.next_flip;
  begin_iteration(j, j.b, 0);
  j.have_a = false;
  j.have_b = true;
}

// Iteration 0 of the next level as start_level() will set it up if iteration a's likelihood is accepted.  Works on a copy:
// nothing of the pair's state or result is touched.  (An accepted likelihood leaves initial / estimate / inc as they are,
// dense_tracking.cpp:312-357; the increment itself only enters the termination statistics of the finished level.)
void speculate_next_level(const Job &j, IterCtx &b_out) {
  Job c = j;
  c.level = j.level - 1;
  se3_log(c.inc, c.x);
  c.buf_flip = j.a.buf ^ 1;
  begin_iteration(c, b_out, 0);
}

// K * T[0:3,0:4] in float, evaluated like Eigen's coefficient-based 3x3 * 3x4 product (dense_tracking_impl.cpp:142-152)
void make_kt(const LevelData &C, const SE3 &estimate, float kt[12]) {
  double Td[16];
  se3_matrix(estimate, Td);
  float T[16];
  for (int i = 0; i < 16; ++i) T[i] = (float)Td[i];  // estimate().matrix().cast<float>(), dense_tracking.cpp:263
  const float K[9] = {C.fx, 0.0f, C.ox, 0.0f, C.fy, C.oy, 0.0f, 0.0f, 1.0f};
  for (int i = 0; i < 3; ++i)
    for (int c = 0; c < 4; ++c)
      kt[i * 4 + c] = (K[i * 3 + 0] * T[c * 4 + 0] + K[i * 3 + 1] * T[c * 4 + 1]) + K[i * 3 + 2] * T[c * 4 + 2];
}

int blocks_for(int n, int steps) {
  const int px_per_block = kStepPx * kWavesPerBlock * steps;
  return (n + px_per_block - 1) / px_per_block;
}

// blocks of the residual pass over the compacted selection of a level (at least one: an empty selection still runs a block of
// padding, whose pixels are all invalid -- the driver's TooFewConstraints path needs its record)
int level_blocks(const Selection *sel, int level, int steps) { return std::max(1, blocks_for(sel->n_pts[level], steps)); }

// computeScaleSse's 1/(n-2-1) and the 2x2 inverse (dense_tracking.cpp:295); S holds the unscaled pair sums
void scale_and_precision(const FinOut &o, int n, float cov[4], float P[4]) {
  const float scale = 1.0f / (float)(size_t)(n - 2 - 1);
  cov[0] = (float)(o.S[0] * (double)scale);
  cov[1] = cov[2] = (float)(o.S[1] * (double)scale);
  cov[3] = (float)(o.S[2] * (double)scale);
  inverse2x2f(cov, P);
}

// A = sum w J^T P J, b = -sum w J^T P r from the P-free moments; + Mu terms (dense_tracking.cpp:341-346)
void system_from_moments(const FinOut &o, const float P[4], double mu, const double xi_initial[6], double A[36], double b[6]) {
  const double p00 = P[0], p10 = P[1], p01 = P[2], p11 = P[3];
  const double pab = 0.5 * (p01 + p10);
  int t = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = r; c < 6; ++c, ++t) {
      const double v = p00 * o.acc[kAccAA + t] + pab * o.acc[kAccAB + t] + p11 * o.acc[kAccBB + t];
      A[c * 6 + r] = v;
      A[r * 6 + c] = v;
    }
  for (int i = 0; i < 6; ++i) {
    A[i * 6 + i] += mu;
    const double bi = -(p00 * o.acc[kAccAR0 + i] + p10 * o.acc[kAccBR0 + i] + p01 * o.acc[kAccAR1 + i] +
                        p11 * o.acc[kAccBR1 + i]);
    b[i] = bi + mu * xi_initial[i];
  }
}

// computeCompleteDataLogLikelihood's last line, dense_tracking_impl.cpp:424
// `overflowed`: one of the reference's 50-term products ran past the double range: its error_sum is +inf (:416-419)
float loglik_from_sum(int n, const float P[4], double ll_sum, bool overflowed) {
  const float det = P[0] * P[3] - P[1] * P[2];
  if (overflowed) ll_sum = HUGE_VAL;
  return (float)(0.5 * (double)(size_t)n * (double)std::log(det) - 0.5 * (5.0 + 2.0) * ll_sum);
}

// the residual pass of iteration `it` came back: dense_tracking.cpp:273-347 minus the likelihood test
void process_residual(Job &j, IterCtx &it, const FinOut &o) {
  dvo_amd_iteration_stats *e = stats_push(j);
  e->id = it.k;
  se3_matrix(it.estimate_after, e->estimate);
  se3_matrix(it.initial_after, e->initial);
  it.n = o.valid;
  it.cut_rank = 50 * (it.n / 50);
  e->valid_constraints = it.n;
  it.stats_index = j.result->n_iterations - 1;
  if (it.n < 6) {  // :276-284
    j.initial = it.initial_before;
    j.estimate = it.estimate_before;
    j.result->levels[j.result->n_levels - 1].termination = DVO_AMD_TERM_TOO_FEW_CONSTRAINTS;
    stats_publish(j);
    end_level(j);
    return;
  }
  scale_and_precision(o, it.n, it.cov, it.P);
  std::memcpy(j.precision, it.P, sizeof(it.P));

  double xi_initial[6];
  se3_log(it.initial_after, xi_initial);
  double sq = 0.0;
  for (int i = 0; i < 6; ++i) sq += xi_initial[i] * xi_initial[i];
  it.prior = j.cfg->mu * sq;  // :302

  system_from_moments(o, it.P, j.cfg->mu, xi_initial, it.A, it.b);
  solve_ldlt6(it.A, it.b, it.x_new);  // :347
  it.cont = inf_norm6(it.x_new) > j.cfg->precision && !(it.k + 1 >= j.cfg->max_iterations_per_level);

  // this iteration now waits for its likelihood; if the loop would go on, run the next residual pass alongside
  j.a = it;
  j.have_a = true;
  j.have_b = false;
  if (j.a.cont) {
    double keep_x[6];
    std::memcpy(keep_x, j.x, sizeof(keep_x));
    std::memcpy(j.x, j.a.x_new, sizeof(j.x));
    begin_iteration(j, j.b, j.a.k + 1);
    std::memcpy(j.x, keep_x, sizeof(keep_x));  // x is only committed once iteration a is accepted
    j.have_b = true;
  }
}

// the likelihood of iteration a came back: dense_tracking.cpp:297-322 and the tail of the loop (:351-357)
void process_loglik(Job &j, const FinOut *outs, bool ll_overflowed) {
  IterCtx &a = j.a;
  const FinOut &o = outs[0];
  const float ll = loglik_from_sum(a.n, a.P, o.ll_sum, ll_overflowed);
  dvo_amd_iteration_stats *e = &j.recent[j.recent_count - 1];
  e->tdist_loglik = -(double)ll;
  e->tdist_mean[0] = e->tdist_mean[1] = 0.0;
  for (int i = 0; i < 4; ++i) e->tdist_precision[i] = a.P[i];
  e->prior_loglik = a.prior;
  j.last_error = j.error;
  j.error = -(double)ll;
  const bool accept = j.error < j.last_error;  // :312
  if (!accept) {
    if (j.sub_res) j.discarded_px += j.sub_px;  // iteration k+1 (or the next level's first pass) ran alongside: thrown away
    j.have_spec = false;  // a speculative start of the next level assumed acceptance: discarded
    j.next_flip = a.buf ^ 1;
    // :314-322: roll back iteration a (and the speculative iteration b, if any)
    j.initial = a.initial_before;
    j.estimate = a.estimate_before;
    j.inc = a.inc;
    j.result->levels[j.result->n_levels - 1].termination = DVO_AMD_TERM_LOGLIKELIHOOD_DECREASED;
    stats_publish(j);
    end_level(j);
    return;
  }
  std::memcpy(e->increment, a.x_new, sizeof(a.x_new));
  std::memcpy(e->information, a.A, sizeof(a.A));
  e->has_increment = 1;
  stats_publish(j);
  std::memcpy(j.x, a.x_new, sizeof(j.x));
  j.iteration = a.k + 1;
  if (!a.cont) {
    const bool spec = j.have_spec;
    const IterCtx spec_b = j.spec_b;
    j.have_spec = false;
    j.next_flip = a.buf ^ 1;
    end_level(j);
    if (spec && !j.done && j.have_b) {
      // the next level's first residual pass ran in this tick with exactly the state start_level() has just set up
      j.b.steps = spec_b.steps, j.b.n_blocks = spec_b.n_blocks;
      IterCtx b = j.b;
      process_residual(j, b, o);
    }
    return;
  }
  // iteration b's residual pass ran in the same tick
  j.have_a = false;
  IterCtx b = j.b;
  process_residual(j, b, o);
}

// frees every slot and forgets the capacity, so that the next ensure_slots() rebuilds from scratch
void release_slots(dvo_amd_context *ctx) {
  for (JobSlot &s : ctx->slots)
    if (s.dev_block) (void)hipFree(s.dev_block);
  ctx->slots.clear();
  if (ctx->slot_desc) (void)hipFree(ctx->slot_desc);
  ctx->slot_desc = nullptr;
  ctx->slot_n_pad = 0;
}

int ensure_slots_impl(dvo_amd_context *ctx, int n_jobs, int n_pad) {
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (hipStream_t st : ctx->extra_streams) HIP_TRY(hipStreamSynchronize(st));
  const int new_pad = std::max(n_pad, ctx->slot_n_pad);
  release_slots(ctx);
  const int n_slots = std::max(n_jobs, 1);
  if (ctx->out_capacity < n_slots) {
    if (ctx->out_wire) (void)hipHostFree(ctx->out_wire);
    ctx->out_wire = nullptr;
    ctx->out_capacity = 0;
    HIP_TRY(hipHostMalloc((void **)&ctx->out_wire, sizeof(FinWire) * n_slots, hipHostMallocMapped | hipHostMallocCoherent));
    ctx->out_capacity = n_slots;
    ctx->out_store.assign((size_t)n_slots, FinOut());
    ctx->out_host = ctx->out_store.data();
  }
  std::memset(ctx->out_wire, 0, sizeof(FinWire) * (size_t)n_slots);  // tick numbers restart below: no piece may carry an old one
  HIP_TRY(hipMalloc((void **)&ctx->slot_desc, sizeof(SlotDesc) * n_slots));
  std::vector<SlotDesc> slot_host((size_t)n_slots);
  const int max_blocks = new_pad / (kStepPx * kWavesPerBlock);  // one-step segments: the most blocks a level can have
  const size_t b_res = align_up(sizeof(float2) * new_pad, 256);
  const size_t b_rec = align_up(sizeof(float) * kRecStride * max_blocks, 256);
  const size_t b_ll = align_up(sizeof(double) * max_blocks, 256), b_lq = align_up(sizeof(float) * max_blocks, 256);
  const size_t b_sp = align_up(sizeof(int) * kWavesPerBlock * max_blocks, 256);
  const size_t b_out = align_up(sizeof(FinOut), 256);
  const size_t b_q7 = align_up(sizeof(Q7Rec), 256);  // behind ll_qmax: k_finalize finds it from ll_partials (FinItem::q7_off256)
  const size_t total = 2 * b_res + b_rec + b_ll + b_lq + b_q7 + 2 * b_sp + b_out;
  if ((b_ll + b_lq) / 256 > 0xFFFFu) return fail_hip("level too large for the slot layout", hipErrorInvalidValue);
  ctx->q7_off256 = (int)((b_ll + b_lq) / 256);
  ctx->slots.resize(n_slots);
  for (int i = 0; i < n_slots; ++i) {
    JobSlot &s = ctx->slots[i];
    if (i == ctx->fault_slot_alloc) {  // DVO_AMD_FAULT_SLOT_ALLOC=i (tests): this allocation fails once
      ctx->fault_slot_alloc = -1;
      return fail_hip("slot allocation (injected fault)", hipErrorOutOfMemory);
    }
    HIP_TRY(hipMalloc(&s.dev_block, total));
    char *p = (char *)s.dev_block;
    s.res[0] = (float2 *)p, p += b_res;
    s.res[1] = (float2 *)p, p += b_res;
    s.records = (float *)p, p += b_rec;
    s.ll_partials = (double *)p, p += b_ll;
    s.ll_qmax = (float *)p, p += b_lq;
    s.ll_qmax_off = (unsigned)(b_ll / sizeof(double));
    s.q7 = (Q7Rec *)p, p += b_q7;
    s.seg_prefix[0] = (int *)p, p += b_sp;
    s.seg_prefix[1] = (int *)p, p += b_sp;
    s.out_dev = (FinOut *)p, p += b_out;
    FinWire *dev_out = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void **)&dev_out, ctx->out_wire + i, 0));
    s.out = dev_out;
    SlotDesc &sd = slot_host[(size_t)i];
    sd.res[0] = s.res[0], sd.res[1] = s.res[1];
    sd.records = s.records, sd.ll_partials = s.ll_partials, sd.ll_qmax = s.ll_qmax;
    sd.seg_prefix[0] = s.seg_prefix[0], sd.seg_prefix[1] = s.seg_prefix[1];
    sd.dbg_w = nullptr;
  }
  HIP_TRY(hipMemcpy(ctx->slot_desc, slot_host.data(), sizeof(SlotDesc) * n_slots, hipMemcpyHostToDevice));
  ctx->tick_seq = 0;
  ctx->slot_n_pad = new_pad;  // set last: only a completely built set of slots counts as capacity
  return DVO_AMD_OK;
}

// Scratch for n_jobs resident pairs of up to n_pad padded pixels.  Failure atomic: if any allocation fails, everything built
// so far is released and the recorded capacity is zero, so a retry (e.g. with fewer resident pairs after
// DVO_AMD_ERR_OUT_OF_MEMORY) rebuilds instead of launching on half-initialised slots.
int ensure_slots(dvo_amd_context *ctx, int n_jobs, int n_pad) {
  if ((int)ctx->slots.size() >= n_jobs && ctx->slot_n_pad >= n_pad && ctx->slot_n_pad > 0) return DVO_AMD_OK;
  const int rc = ensure_slots_impl(ctx, n_jobs, n_pad);
  if (rc) release_slots(ctx);
  return rc;
}

// Steps per wave segment of a residual pass over a level of n_px pixels.  A pair's result must depend on its inputs only
// (the reference runs independent match() calls under tbb::parallel_reduce / parallel_invoke, keyframe_graph.cpp:587-590,
// local_tracker.cpp:184: whatever runs beside a pair cannot change it), and the segment length decides where the fp32 sums of
// a pass are cut.  So it is a function of the level's size alone: the same for a single match(), a pair in a batch of any
// residency, the submit queue, every validator worker and every band count.  Until round 3 it was picked per tick from the
// pixels of everything resident (short segments for small ticks, long ones for saturating launches), and a batched result
// moved by up to 5.8e-5 from the same pair's single match().
// The trade-off the table settles: a step is a dependent chain (reference scalars -> projection -> gathers -> arithmetic ->
// staging, ~1.5 us when nothing else hides it), so short segments make a single pair's tick shorter; a block's prologue and
// epilogue (descriptors, seven wave reductions, the Gram tile, the block record) are amortised over long ones.
int level_steps(const dvo_amd_context *ctx, const LevelData &lv) {
  const int n_px = lv.n;
  const long long waves = n_px / kStepPx;
  const long long *t = ctx->cfg.segment_geometry == DVO_AMD_GEOMETRY_LATENCY ? ctx->level_steps_at_latency : ctx->level_steps_at;
  int steps = waves >= t[4] ? 32 : waves >= t[3] ? 16 : waves >= t[2] ? 8 : waves >= t[1] ? 4 : waves >= t[0] ? 2 : 1;
  // Row-aligned segments (end of round 5): on the levels the table gives its longest segments, a wave segment is a whole number
  // of image rows when the row is a whole number of steps -- ten steps for a 640-pixel row, ten (two rows) for 320, twenty for
  // 1280.  The four waves of a block then walk the SAME columns one row (or two) apart: the lower bilinear row of one wave is
  // the upper row of the next at the same step, and the gathers of a block share their lines while they are in L1 -- L2-miss
  // traffic 0.83 x the algorithmic bytes against 1.11 x with 16-step and 0.98 x with 8-step segments, a launch alone 0.40 of the
  // HBM roofline against 0.36 / 0.385, the batch +1 % on 16 steps (profiles/r05_row_aligned_segments_ab.txt).  Only where the last
  // block of the level still lies inside the planes' padding; DVO_AMD_FINE_STEPS=n forces a length, =16 turns the rule off.
  if (steps == 16 && ctx->cfg.segment_geometry != DVO_AMD_GEOMETRY_LATENCY) {
    int want = ctx->fine_steps;
    if (want == 0 && lv.w % kStepPx == 0) {
      want = lv.w / kStepPx;
      if (want % 2) want *= 2;
      while (want < 8) want *= 2;
    }
    TickItem probe_item;
    std::memset(&probe_item, 0, sizeof(probe_item));
    if (want >= 2 && want <= 32 && want % 2 == 0) {
      item_set_steps(probe_item, want, 1);
      const long long block_px = (long long)kStepPx * kWavesPerBlock * want;
      if (item_res_steps(probe_item) == want && (n_px + block_px - 1) / block_px * block_px <= lv.n_pad && (n_px + block_px - 1) / block_px <= 2048)
        return want;
    }
  }
  while (steps < kMaxSteps && (n_px + kStepPx * kWavesPerBlock * steps - 1) / (kStepPx * kWavesPerBlock * steps) > 2048) steps *= 2;
  return steps;
}
// residual wave segments one likelihood wave walks (a likelihood step is a tenth of a residual step's work): a function of the
// pass's segment length alone, like level_steps -- a lane's running product, and so its logs, are cut at the ends of ITS segment
int level_ll_merge(const dvo_amd_context *ctx, int res_steps) {
  int merge = 1;
  while (merge < ctx->ll_merge && res_steps * merge * 2 <= kMaxSteps) merge *= 2;
  return merge;
}

// a pair of events for the next timed launch; the launch itself stamps them (begin / end of that dispatch)
int timing_begin(dvo_amd_context *ctx, size_t *slot) {
  if (ctx->events_used == ctx->events.size()) {
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    ctx->events.emplace_back(a, b);
  }
  *slot = ctx->events_used++;
  return DVO_AMD_OK;
}


int tick_stream(dvo_amd_context *ctx, size_t index, hipStream_t *out) {
  index %= kMaxTickStreams;
  if (index == 0) {
    *out = ctx->stream;
    return DVO_AMD_OK;
  }
  while (ctx->extra_streams.size() < index) {
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    ctx->extra_streams.push_back(s);
  }
  *out = ctx->extra_streams[index - 1];
  return DVO_AMD_OK;
}

// ---- which hardware queue does a context's stream run on? --------------------------------------------------------------------
// The runtime maps streams onto GPU_MAX_HW_QUEUES (4) hardware queues, and a hardware queue runs one kernel at a time: how the
// trackers of a GPU are spread over them decides up to a third of a batch's throughput (six trackers: 55-56 k pairs/s as the
// runtime places them when they are created back to back -- two per queue on three queues, neighbours together --, 51 k with the
// same two per queue but every other tracker together, 47 k on two queues, 39 k on one:
// profiles/r05_stream_queue_assignment_ab.txt).  The library takes the stream the runtime deals it: a context that picked among
// candidate streams by probing them was built and measured at the end of round 5 and made the common case worse (the probe is a
// stream's first use and changes what the runtime does next: 53.2 k), so what is left of it is the probe as a diagnostic
// (dvo_amd_debug_hw_queue) and the advice in INTEGRATION.md: create the trackers of a GPU back to back.
int probe_hw_queue(dvo_amd_context *ctx, int *pipe_queue) {
  unsigned *word = nullptr, *word_dev = nullptr;
  HIP_TRY(hipHostMalloc((void **)&word, 64, hipHostMallocMapped | hipHostMallocCoherent));
  __atomic_store_n(word, 0u, __ATOMIC_RELEASE);
  hipError_t e = hipHostGetDevicePointer((void **)&word_dev, word, 0);
  if (e == hipSuccess) e = launch_queue_probe(word_dev, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  const unsigned v = __atomic_load_n(word, __ATOMIC_ACQUIRE);
  (void)hipHostFree(word);
  if (e != hipSuccess) return fail_hip("dvo_amd_debug_hw_queue", e);
  *pipe_queue = (v & 0x80000000u) ? (int)(v & 0xFFu) : -1;
  return DVO_AMD_OK;
}

int timing_collect(dvo_amd_context *ctx) {
  for (size_t i = 0; i < ctx->events_used; ++i) {
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->events[i].first, ctx->events[i].second));
    ctx->timing_ms += ms;
    ctx->timing_launches++;
    if (kTickLogFields * (i + 1) <= ctx->tick_log_pending.size() && ctx->tick_log.size() < (size_t)kTickLogFields * 65536) {
      ctx->tick_log.push_back((double)ms);
      for (size_t k = 1; k < kTickLogFields; ++k) ctx->tick_log.push_back(ctx->tick_log_pending[kTickLogFields * i + k]);
    }
  }
  ctx->tick_log_pending.clear();
  ctx->events_used = 0;
  return DVO_AMD_OK;
}

// Take the pieces of a record that carry tick `seq` out of a pinned buffer into *dst; returns the index of the first piece
// that is not there yet (kFinWirePieces when the record is complete).  A piece is one aligned 16-byte load: payload and tag
// come from the same store of the device, and each 8-byte half of the piece carries the tag.
int take_wire(const FinWire *w, FinOut *dst_record, unsigned seq, int from_piece) {
  unsigned *dst = reinterpret_cast<unsigned *>(dst_record);
  for (int i = from_piece; i < kFinWirePieces; ++i) {
    __asm__ __volatile__("" ::: "memory");
    alignas(16) unsigned u[4];
    _mm_store_si128(reinterpret_cast<__m128i *>(u), _mm_load_si128(reinterpret_cast<const __m128i *>(w->piece[i])));
    if (u[1] != seq || u[3] != seq) return i;  // both 8-byte halves carry the tick (FinWire)
    dst[2 * i] = u[0];
    if (2 * i + 1 < kFinWords) dst[2 * i + 1] = u[2];
  }
  dst_record->seq = seq;
  return kFinWirePieces;
}
int take_record(dvo_amd_context *ctx, size_t slot, unsigned seq, int from_piece) {
  return take_wire(ctx->out_wire + slot, ctx->out_host + slot, seq, from_piece);
}

// after a stream synchronisation every piece must be there
int take_record_synced(dvo_amd_context *ctx, size_t slot, unsigned seq) {
  if (take_record(ctx, slot, seq, 0) != kFinWirePieces)
    return fail_hip("tick finished without publishing its record", hipErrorUnknown);
  return DVO_AMD_OK;
}

// wait until the finalize kernel has published this tick's record of every submitted job
int wait_tick(dvo_amd_context *ctx, const std::vector<Job> &jobs, size_t lo, size_t hi, unsigned seq) {
  const bool synced = !ctx->poll || ctx->timing;
  if (synced) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (hipStream_t s : ctx->extra_streams) HIP_TRY(hipStreamSynchronize(s));
  }
  for (size_t ji = lo; ji < hi; ++ji) {
    const Job &j = jobs[ji];
    if (j.done || !(j.sub_ll || j.sub_res)) continue;
    const size_t slot = (size_t)(j.slot - ctx->slots.data());
    if (synced) {
      int rc = take_record_synced(ctx, slot, seq);
      if (rc) return rc;
      continue;
    }
    unsigned long long spins = 0;
    int have = 0;
    while ((have = take_record(ctx, slot, seq, have)) != kFinWirePieces) {
      __builtin_ia32_pause();  // be polite to the sibling hardware thread while spinning
      if ((++spins & 0xFFFFF) == 0) {  // every ~1M polls make sure the streams are still alive
        bool all_idle = true;
        for (size_t si = 0; si <= ctx->extra_streams.size(); ++si) {
          hipError_t e = hipStreamQuery(si == 0 ? ctx->stream : ctx->extra_streams[si - 1]);
          if (e == hipErrorNotReady) {
            all_idle = false;
          } else if (e != hipSuccess) {
            return fail_hip("stream died while waiting for a tick", e);
          }
        }
        if (all_idle && (have = take_record(ctx, slot, seq, have)) != kFinWirePieces)
          return fail_hip("tick finished without publishing its record", hipErrorUnknown);
      }
    }
  }
  return DVO_AMD_OK;
}

// One tick of a group of resident pairs (slots [lo, hi) of a context): submit_tick enqueues what every unfinished pair of
// the group needs, complete_tick waits for the records and advances the pairs.  Groups of one context tick independently
// on their own streams, so the host work of one group overlaps the kernels of the others.
inline double now_ns() {
  return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int submit_tick(dvo_amd_context *ctx, std::vector<Job> &jobs, GroupTick &grp) {
  const double t_begin = ctx->host_prof ? now_ns() : 0.0;
  grp.in_flight = false;
  int active_pairs = 0;
  for (size_t ji = grp.lo; ji < grp.hi; ++ji) active_pairs += jobs[ji].done ? 0 : 1;
  const bool speculate_levels = ctx->spec_levels == 1 || (ctx->spec_levels < 0 && active_pairs <= kSpecLevelsMaxPairs);
  auto wants_spec = [&](const Job &j) {
    return speculate_levels && j.have_a && !j.have_b && !j.a.cont && j.level > j.cfg->last_level;
  };
  const unsigned seq = ctx->tick_seq = next_seq(ctx->tick_seq);
  grp.seq = seq;

  std::vector<TickItem> items;
  std::vector<FinItem> fin_items;
  items.reserve(grp.hi - grp.lo);
  fin_items.reserve(grp.hi - grp.lo);
  for (size_t ji = grp.lo; ji < grp.hi; ++ji) {
    Job &j = jobs[ji];
    if (j.done) continue;
    j.sub_ll = j.sub_res = false;
    if (!j.have_a && !j.have_b) continue;
    const size_t slot_index = (size_t)(j.slot - ctx->slots.data());
    TickItem w;
    std::memset(&w, 0, sizeof(w));
    int res_steps = 1, ll_steps = 1;
    w.ref = j.sel->ref_desc + j.level;
    w.cur = j.cur->cur_desc + j.level;
    w.slot = ctx->slot_desc + slot_index;
    FinItem f;
    std::memset(&f, 0, sizeof(f));
    f.ll_partials = j.slot->ll_partials;
    f.seg_prefix_out = j.slot->seg_prefix[0], f.out = j.slot->out, f.out_dev = nullptr, f.seq = seq;
    f.ll_qmax_off = j.slot->ll_qmax_off;
    if (j.have_b) {
      j.b.steps = level_steps(ctx, j.ref->lv[j.level]);  // the level's own geometry, whatever else this tick carries
      j.b.n_blocks = level_blocks(j.sel, j.level, j.b.steps);
    }
    if (j.have_a) {
      if (j.a.buf) w.flags |= kItemLlBuf;
      w.ll_cut_rank = j.a.cut_rank;
      // the likelihood blocks walk the residuals in merged blocks: up to ll_merge residual blocks per likelihood block, a
      // function of the pass's geometry alone (level_ll_merge).  Measured: 45.8k -> 48.1k pairs/s at 4; folding the likelihood
      // into the residual-pass waves of the next iteration instead (no likelihood blocks at all) gave nothing on top of it and
      // cost 22 registers and 5 % single-pair latency
      item_set_ll_merge(w, level_ll_merge(ctx, j.a.steps));
      w.ll_level_blocks = (uint16_t)j.a.n_blocks;
      w.ll_first = 0;
      w.ll_blocks = (uint16_t)ll_blocks_total(j.a.n_blocks, item_ll_merge_log2(w));
      ll_steps = j.a.steps;
      f.n_ll_blocks = w.ll_blocks, f.ll_first = 0, f.ll_level_blocks = w.ll_level_blocks, f.ll_merge_log2 = (uint16_t)item_ll_merge_log2(w);
      j.sub_ll = true;
    }
    if (j.have_b) {
      w.res_blocks = (uint16_t)j.b.n_blocks;
      res_steps = j.b.steps;
      if (j.b.buf) w.flags |= kItemResBuf;
      if (j.b.k == 0) w.flags |= kItemUnitWeights;  // dense_tracking.cpp:286-293
      make_kt(j.cur->lv[j.level], j.b.estimate_after, w.kt);
      f.records = j.slot->records;
      f.n_blocks = w.res_blocks, f.level_blocks = w.res_blocks;
      f.seg_prefix_out = j.slot->seg_prefix[j.b.buf];
      // host-rcpps mode: the pass's last V mod 4 weights are exact divisions (Q7): k_q7_tail leaves the difference, k_finalize adds it
      if (ctx->rcp.table && j.b.k != 0) f.q7_off256 = (uint16_t)ctx->q7_off256;
      j.sub_res = true;
      j.sub_px = (double)j.sel->count[j.level];
      j.result->n_residual_passes++;
      j.alg_px += j.sub_px;
    } else if (wants_spec(j)) {
      // iteration a ends its level whatever its likelihood says: start the next level in this tick, assuming acceptance
      const int nl = j.level - 1;
      speculate_next_level(j, j.spec_b);
      j.spec_b.steps = level_steps(ctx, j.ref->lv[nl]);
      j.spec_b.n_blocks = level_blocks(j.sel, nl, j.spec_b.steps);
      w.ref = j.sel->ref_desc + nl;  // the likelihood pass only uses the slot's buffers
      w.cur = j.cur->cur_desc + nl;
      w.res_blocks = (uint16_t)j.spec_b.n_blocks;
      res_steps = j.spec_b.steps;
      if (j.spec_b.buf) w.flags |= kItemResBuf;
      w.flags |= kItemUnitWeights;
      make_kt(j.cur->lv[nl], j.spec_b.estimate_after, w.kt);
      f.records = j.slot->records;
      f.n_blocks = w.res_blocks, f.level_blocks = w.res_blocks;
      f.seg_prefix_out = j.slot->seg_prefix[j.spec_b.buf];
      j.have_spec = true;
      j.sub_res = true;
      j.sub_px = (double)j.sel->count[nl];
      j.result->n_residual_passes++;
      j.alg_px += j.sub_px;
    }
    // the likelihood of iteration k and the weights of iteration k+1 both use the precision of iteration k (a's); without a
    // pending likelihood the weights use the job's current precision (unused at the first iteration of a level)
    std::memcpy(w.P, j.have_a ? j.a.P : j.precision, sizeof(w.P));
    item_set_steps(w, res_steps, ll_steps);
    items.push_back(w);
    fin_items.push_back(f);
    j.result->n_ticks++;
  }
  if (items.empty()) return DVO_AMD_OK;

  // split evenly over as few launches as the argument block allows; launch i (and the finalize of its jobs) goes to stream
  // i so that the launches of one tick overlap instead of queueing behind each other's latency floor
  const size_t cap = (size_t)std::min(std::max(ctx->items_per_launch, 1), kMaxItemsPerLaunch);
  const size_t n_launch = (items.size() + cap - 1) / cap;
  const size_t per = (items.size() + n_launch - 1) / n_launch;
  size_t launch_index = 0;
  for (size_t first = 0; first < items.size(); first += per, ++launch_index) {
    hipStream_t st;
    {
      int rc = tick_stream(ctx, ctx->timing ? 0 : grp.stream_first + launch_index, &st);  // timed launches run alone, on stream 0
      if (rc) return rc;
    }
    TickArgs ta;
    const int n_here = (int)std::min(per, items.size() - first);
    ta.n_items = n_here;
    ta.compact = 0;
    ta.rcp = ctx->rcp;
    int max_blocks = 0;
    // Blocks are dispatched in grid order and a launch ends with its last block: items whose blocks live longest (the most steps
    // per wave segment) go first, so that the launch's tail is made of short blocks.  The order of the items inside a launch
    // changes no result (every item's blocks, records and reducer are its own).  DVO_AMD_SORT_ITEMS=0: slot order.
    int order[kMaxItemsPerLaunch];
    for (int i = 0; i < n_here; ++i) order[i] = i;
    if (ctx->sort_items)
      std::stable_sort(order, order + n_here, [&](int a, int b) {
        const TickItem &x = items[first + (size_t)a], &y = items[first + (size_t)b];
        const int kx = x.res_blocks ? item_res_steps(x) : 0, ky = y.res_blocks ? item_res_steps(y) : 0;
        return kx > ky;
      });
    for (int i = 0; i < n_here; ++i) {
      ta.items[i] = items[first + (size_t)order[i]];
      max_blocks = std::max(max_blocks, (int)ta.items[i].res_blocks + (int)ta.items[i].ll_blocks);
    }
    size_t ev = 0;
    if (ctx->timing) {
      int rc = timing_begin(ctx, &ev);
      if (rc) return rc;
      double rb = 0, lb = 0, px = 0, res_steps = 0, ll_steps = 0;
      for (int i = 0; i < n_here; ++i) {
        rb += ta.items[i].res_blocks, lb += ta.items[i].ll_blocks;
        res_steps += (double)ta.items[i].res_blocks * kWavesPerBlock * item_res_steps(ta.items[i]);
        if (ta.items[i].ll_blocks) ll_steps += (double)ta.items[i].ll_level_blocks * kWavesPerBlock * item_ll_steps(ta.items[i]);
      }
      for (size_t ji = grp.lo, k = 0; ji < grp.hi; ++ji) {
        const Job &j = jobs[ji];
        if (j.done || !(j.sub_ll || j.sub_res)) continue;
        if (k >= first && k < first + (size_t)n_here && j.sub_res) px += j.sub_px;
        ++k;
      }
      const double rec[kTickLogFields] = {0.0, (double)n_here, rb, lb, (double)max_blocks, px, res_steps, ll_steps};
      ctx->tick_log_pending.insert(ctx->tick_log_pending.end(), rec, rec + kTickLogFields);
    }
    hipEvent_t t0 = ctx->timing ? ctx->events[ev].first : nullptr, t1 = ctx->timing ? ctx->events[ev].second : nullptr;
    if (ctx->small_args && n_launch == 1 && n_here <= kMaxSmallItems) {
      // a single match() or the two-pair front-end step: the same two kernels behind argument blocks a tenth the size
      TickArgsSmall ts;
      ts.n_items = n_here, ts.compact = 0, ts.rcp = ctx->rcp;
      for (int i = 0; i < kMaxSmallItems; ++i) ts.items[i] = ta.items[i < n_here ? i : 0];
      (void)tick_args_layout(ts, max_blocks);
      const hipError_t es = launch_tick_small(ts, max_blocks, st, t0, t1);
      if (es == hipSuccess) {
        if (ctx->rcp.table) {
          Q7ArgsSmall qs;
          qs.n_items = 0, qs.q7_off256 = ctx->q7_off256, qs.rcp = ctx->rcp;
          for (int i = 0; i < n_here; ++i)
            if (fin_items[first + (size_t)order[i]].q7_off256) qs.items[qs.n_items++] = ta.items[i];
          for (int i = qs.n_items; i < kMaxSmallItems; ++i) qs.items[i] = ta.items[0];
          const hipError_t eq = launch_q7_tail_small(qs, st);
          if (eq != hipSuccess) return fail_hip("launch_q7_tail", eq);
        }
        FinArgsSmall fs;
        fs.n_items = n_here, fs.pad = ctx->fin_stamps ? 0x57A3 : 0;
        for (int i = 0; i < kMaxSmallItems; ++i) fs.items[i] = fin_items[first + (size_t)order[i < n_here ? i : 0]];
        const hipError_t ef = launch_finalize_small(fs, st);
        if (ef != hipSuccess) return fail_hip("launch_finalize", ef);
        continue;
      }
      if (es != hipErrorNotSupported) return fail_hip("launch_tick", es);
      (void)hipGetLastError();  // DVO_AMD_ACCUM=valu selected the register form: the full-size launch below
    }
    (void)tick_args_layout(ta, max_blocks);
    hipError_t e = launch_tick(ta, max_blocks, st, t0, t1);
    if (e != hipSuccess) return fail_hip("launch_tick", e);
    static_assert(kMaxFinItems >= kMaxItemsPerLaunch, "one reduce launch per tick launch");
    if (ctx->rcp.table) {
      Q7Args qa;
      qa.n_items = 0, qa.q7_off256 = ctx->q7_off256, qa.rcp = ctx->rcp;
      for (int i = 0; i < n_here; ++i)
        if (fin_items[first + (size_t)order[i]].q7_off256) qa.items[qa.n_items++] = ta.items[i];
      for (int i = qa.n_items; i < kMaxItemsPerLaunch; ++i) qa.items[i] = ta.items[0];
      e = launch_q7_tail(qa, st);
      if (e != hipSuccess) return fail_hip("launch_q7_tail", e);
    }
    FinArgs fa;
    fa.n_items = n_here;
    fa.pad = ctx->fin_stamps ? 0x57A3 : 0;
    fa.exchange = nullptr, fa.xseq = 0, fa.pad2 = ctx->fin_priority ? kFinFlagPriority : 0u;
    for (int i = 0; i < n_here; ++i) fa.items[i] = fin_items[first + (size_t)order[i]];
    for (int i = n_here; i < kMaxFinItems; ++i) fa.items[i] = fa.items[0];  // the whole block is copied by the launch: no stale stack bytes
    e = launch_finalize(fa, st);
    if (e != hipSuccess) return fail_hip("launch_finalize", e);
  }
  grp.in_flight = true;
  if (ctx->host_prof) ctx->prof_submit_ns += now_ns() - t_begin, ctx->prof_ticks++, ctx->prof_job_ticks += (long long)items.size();
  return DVO_AMD_OK;
}

// The reference's log-likelihood multiplies 50 consecutive terms 1 + 0.2 r^T P r in a double before it takes a log
// (dense_tracking_impl.cpp:413-419); with precisions of 1e9 and more (noise-free synthetic depth) and a run of 50 large residuals
// that product overflows, the likelihood is -inf and the iteration is rejected (:312).  The likelihood pass reports the largest
// r^T P r it saw; only when a group of fifty COULD have overflowed (kLlOverflowScreen) this asks k_ll_overflow, which redoes
// the reference's own multiplications group by group over the iteration's residual buffer (still intact: the next residual pass
// wrote the other one).  Blocking and slow (0.2 ms for a 640x480 level), and rare: never on sensor data.
int ll_overflowed(dvo_amd_context *ctx, const float2 *res, const int *seg_prefix, int n_blocks, int steps, int cut_rank,
                  const float P[4], const OvfBand *bands, int n_bands, bool *overflowed) {
  *overflowed = false;
  if (cut_rank < 50) return DVO_AMD_OK;
  if (!ctx->ovf_host) {
    HIP_TRY(hipHostMalloc((void **)&ctx->ovf_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
    HIP_TRY(hipHostGetDevicePointer((void **)&ctx->ovf_dev, ctx->ovf_host, 0));
  }
  __atomic_store_n(ctx->ovf_host, 0u, __ATOMIC_RELEASE);
  const int seg_px = kStepPx * steps, n_px = n_blocks * kWavesPerBlock * seg_px;
  OvfBand whole;
  whole.seg_first = 0, whole.n_segs = n_blocks * kWavesPerBlock, whole.rank_offset = 0;
  for (int b = 0; b < (bands ? n_bands : 1); ++b) {
    const OvfBand &B = bands ? bands[b] : whole;
    hipError_t e = launch_ll_overflow(res, seg_prefix, B.seg_first, B.n_segs, seg_px, B.rank_offset, n_px, cut_rank, B.rank_end, P,
                                      ctx->ovf_dev, ctx->stream);
    if (e != hipSuccess) return fail_hip("launch_ll_overflow", e);
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  *overflowed = __atomic_load_n(ctx->ovf_host, __ATOMIC_ACQUIRE) != 0u;
  ctx->ovf_checks++, ctx->ovf_hits += *overflowed ? 1 : 0;
  return DVO_AMD_OK;
}

int complete_tick(dvo_amd_context *ctx, std::vector<Job> &jobs, GroupTick &grp) {
  if (!grp.in_flight) return DVO_AMD_OK;
  grp.in_flight = false;
  const double t_begin = ctx->host_prof ? now_ns() : 0.0;
  {
    int rc = wait_tick(ctx, jobs, grp.lo, grp.hi, grp.seq);
    if (rc) return rc;
  }
  const double t_waited = ctx->host_prof ? now_ns() : 0.0;
  if (ctx->timing) {
    int rc = timing_collect(ctx);
    if (rc) return rc;
  }

  for (size_t ji = grp.lo; ji < grp.hi; ++ji) {
    Job &j = jobs[ji];
    if (j.done || !(j.sub_ll || j.sub_res)) continue;
    const FinOut *o = ctx->out_host + (j.slot - ctx->slots.data());
    if (j.sub_ll) {
      bool overflowed = false;
      if (o->ll_qmax >= kLlOverflowScreen) {
        int rc = ll_overflowed(ctx, j.slot->res[j.a.buf], j.slot->seg_prefix[j.a.buf], j.a.n_blocks, j.a.steps, j.a.cut_rank, j.a.P,
                               nullptr, 0, &overflowed);
        if (rc) return rc;
      }
      process_loglik(j, o, overflowed);
    } else {
      IterCtx b = j.b;
      process_residual(j, b, *o);
    }
  }
  if (ctx->host_prof) ctx->prof_wait_ns += t_waited - t_begin, ctx->prof_process_ns += now_ns() - t_waited;
  return DVO_AMD_OK;
}

// true when every record of the group's tick in flight has arrived completely (never waits)
bool tick_landed(dvo_amd_context *ctx, const std::vector<Job> &jobs, const GroupTick &grp) {
  if (!ctx->poll || ctx->timing) {
    for (size_t si = 0; si <= ctx->extra_streams.size(); ++si)
      if (hipStreamQuery(si == 0 ? ctx->stream : ctx->extra_streams[si - 1]) == hipErrorNotReady) return false;
    return true;
  }
  for (size_t ji = grp.lo; ji < grp.hi; ++ji) {
    const Job &j = jobs[ji];
    if (j.done || !(j.sub_ll || j.sub_res)) continue;
    const FinWire *w = ctx->out_wire + (size_t)(j.slot - ctx->slots.data());
    for (int i = kFinWirePieces - 1; i >= 0; --i) {  // (the last pieces are written by the highest lanes: most likely missing)
      __asm__ __volatile__("" ::: "memory");
      alignas(16) unsigned u[4];
      _mm_store_si128(reinterpret_cast<__m128i *>(u), _mm_load_si128(reinterpret_cast<const __m128i *>(w->piece[i])));
      if (u[1] != grp.seq || u[3] != grp.seq) return false;
    }
  }
  return true;
}


// ---- the queue behind a context: resident pairs + pending pairs ------------------------------------------------------------
// dvo_amd_match_submit appends pairs, every tick of a group hands the slots of finished pairs to pending ones, so the launches
// stay full across calls: a tracker that is fed before it runs dry never drains (the shape of tbb::parallel_reduce with
// grain 1 over a proposal list that keeps growing, keyframe_graph.cpp:587-590).

void runner_finish_slot(Runner &R, size_t sidx) {
  Job &j = R.jobs[sidx];
  for (Batch &b : R.batches)
    if (b.id == R.batch_of_slot[sidx]) {
      b.remaining--;
      break;
    }
  R.batch_of_slot[sidx] = 0;
  R.resident--;
  dvo_amd_pyramid_release(j.ref);
  dvo_amd_pyramid_release(j.cur);
  j.ref = j.cur = nullptr;
}

// a tick failed: nothing of this context may still run when the caller gets the error (it is free to release its pyramids and
// result arrays); every resident and pending pair is dropped and its batch closed with the status
int runner_fail(dvo_amd_context *ctx, int code) {
  Runner &R = *ctx->runner;
  (void)hipStreamSynchronize(ctx->stream);
  for (hipStream_t st : ctx->extra_streams) (void)hipStreamSynchronize(st);
  for (size_t sidx = 0; sidx < R.jobs.size(); ++sidx)
    if (R.batch_of_slot[sidx] != 0) {
      R.jobs[sidx].done = true;
      runner_finish_slot(R, sidx);
    }
  for (Pending &q : R.pending) {
    dvo_amd_pyramid_release(q.ref);
    dvo_amd_pyramid_release(q.cur);
  }
  R.pending.clear();
  for (Batch &b : R.batches) {
    if (b.remaining > 0) R.failures.push_back({b.id, code});  // open when the tick failed: dropped with this status
    b.remaining = 0;
  }
  if (R.failures.size() > 4096) R.failures.erase(R.failures.begin(), R.failures.end() - 2048);
  for (GroupTick &g : R.groups) g.in_flight = false;
  R.unreported_failure = code;
  R.batches.clear();
  return code;
}

// the same for a failure that goes straight back to the caller of the failing submit / wait / poll: that caller has been told,
// so wait / poll of ticket 0 must not report it a second time after a clean run; the submissions that were
// open keep their status and report it through their own tickets
int runner_fail_told(dvo_amd_context *ctx, int code) {
  (void)runner_fail(ctx, code);
  ctx->runner->unreported_failure = DVO_AMD_OK;
  return code;
}

// what wait / poll of `ticket` returns once the ticket is no longer open: the status its submission was dropped with, if it
// was; ticket 0 ("everything submitted so far") reports a failure that no wait / poll has returned yet
int runner_reported_status(Runner &R, unsigned long long ticket) {
  if (ticket == 0) {
    const int st = R.unreported_failure;
    R.unreported_failure = DVO_AMD_OK;
    return st;
  }
  for (const Runner::Failure &f : R.failures)
    if (f.batch == ticket) {
      R.unreported_failure = DVO_AMD_OK;
      return f.status;
    }
  return DVO_AMD_OK;
}

// Entry points that work in slot 0 outside the queue (the band pipeline, the residual / error-image / stage probes, the
// kernel bench) or change what the queue reads (match_selection) must find the queue empty: they would overwrite resident pair
// 0's buffers and record tags, and re-allocating the slots for larger frames would leave the queue's jobs pointing at freed
// memory.  (dvo_amd_configure refuses the same way.)
int queue_must_be_idle(dvo_amd_context *ctx, const char *what) {
  if (ctx->runner && (ctx->runner->resident > 0 || !ctx->runner->pending.empty())) {
    g_last_error = std::string(what) + " while submitted pairs are still in flight (dvo_amd_match_wait first)";
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  }
  return DVO_AMD_OK;
}

// complete the group's tick in flight (waits for it), hand free slots to pending pairs, submit the next tick
int runner_step(dvo_amd_context *ctx, size_t g) {
  Runner &R = *ctx->runner;
  GroupTick &grp = R.groups[g];
  const dvo_amd_config &cfg = ctx->cfg;
  int rc = complete_tick(ctx, R.jobs, grp);
  if (rc) return rc;
  for (size_t sidx = grp.lo; sidx < grp.hi; ++sidx)
    if (R.jobs[sidx].done && R.batch_of_slot[sidx] != 0) runner_finish_slot(R, sidx);
  for (size_t sidx = grp.lo; sidx < grp.hi && !R.pending.empty(); ++sidx) {
    Job &j = R.jobs[sidx];
    if (!j.done) continue;
    const Pending q = R.pending.front();
    R.pending.pop_front();
    j = Job();
    j.ref = q.ref, j.cur = q.cur;
    j.result = q.result;
    j.slot = &ctx->slots[sidx];
    j.cfg = &ctx->cfg;
    R.batch_of_slot[sidx] = q.batch;
    R.resident++;
    rc = pyramid_selection(j.ref, q.ti, q.td, &j.sel);
    if (rc) return rc;
    dvo_amd_result *r = j.result;
    r->n_levels = 0, r->n_iterations = 0, r->n_ticks = 0, r->n_residual_passes = 0, r->alg_bytes = 0.0, r->alg_bytes_discarded = 0.0, r->is_nan = 0;
    if (!r->iterations) r->iterations_capacity = 0;
    // dense_tracking.cpp:137-150
    j.inc = q.has_init ? se3_from_matrix(q.T_init) : SE3::identity();
    j.initial = j.inc;
    j.estimate = SE3::identity();
    j.level = cfg.first_level;
    j.done = false;
    start_level(j);
  }
  return submit_tick(ctx, R.jobs, grp);
}

int runner_drain(dvo_amd_context *ctx) {
  Runner &R = *ctx->runner;
  while (R.resident > 0 || !R.pending.empty()) {
    const size_t g = R.next_group;
    R.next_group = (R.next_group + 1) % R.groups.size();
    int rc = runner_step(ctx, g);
    if (rc) return runner_fail_told(ctx, rc);
  }
  return DVO_AMD_OK;
}

// Lay the context out for `in_flight` resident pairs of up to n_pad padded pixels.  Pairs enter a free slot as soon as one
// opens up: the launch of every tick stays full although pairs need different numbers of iterations.  A pair's state machine
// never looks at another pair, so results do not depend on the schedule (except through the wave-segment length a tick picks,
// which only changes the order partial sums are taken in).  More resident pairs than one launch takes are split into groups
// that tick independently, each on its own stream, in round-robin: while the host advances the pairs of one group the kernels
// of the other groups keep the GPU busy.  (Kernel timing wants every launch alone on the GPU: one group then.)
int runner_configure(dvo_amd_context *ctx, int in_flight, int n_pad) {
  Runner &R = *ctx->runner;
  in_flight = std::max(in_flight, 1);
  const bool same = R.in_flight == in_flight && R.timing == ctx->timing && (int)ctx->slots.size() >= in_flight &&
                    ctx->slot_n_pad >= n_pad && ctx->slot_n_pad > 0;
  if (same) return DVO_AMD_OK;
  if (R.in_flight > 0) {  // a different residency or larger frames: what is queued runs to completion in the old layout first
    int rc = runner_drain(ctx);
    if (rc) return rc;
  }
  int rc = ensure_slots(ctx, in_flight, n_pad);
  if (rc) return rc;
  R.jobs.assign((size_t)in_flight, Job());
  for (Job &j : R.jobs) j.done = true;
  R.batch_of_slot.assign((size_t)in_flight, 0ull);
  const int cap = std::min(std::max(ctx->items_per_launch, 1), kMaxItemsPerLaunch);
  const int n_groups = ctx->timing ? 1 : std::min(kMaxTickStreams, (in_flight + cap - 1) / cap);
  const int per_group = (in_flight + n_groups - 1) / n_groups;
  R.groups.assign((size_t)n_groups, GroupTick());
  for (int g = 0; g < n_groups; ++g) {
    R.groups[(size_t)g].lo = (size_t)std::min(g * per_group, in_flight);
    R.groups[(size_t)g].hi = (size_t)std::min((g + 1) * per_group, in_flight);
    R.groups[(size_t)g].stream_first = (size_t)g;
    R.groups[(size_t)g].id = g;
  }
  R.next_group = 0;
  R.in_flight = in_flight;
  R.timing = ctx->timing;
  R.resident = 0;
  return DVO_AMD_OK;
}

int check_config(const dvo_amd_config *c) {
  if (!c) return DVO_AMD_ERR_INVALID_ARGUMENT;
  if (c->first_level < c->last_level) return DVO_AMD_ERR_INSANE_CONFIG;  // Config::IsSane
  if (c->last_level < 0 || c->first_level >= DVO_AMD_MAX_LEVELS || c->max_iterations_per_level < 1)
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  if (c->segment_geometry != DVO_AMD_GEOMETRY_THROUGHPUT && c->segment_geometry != DVO_AMD_GEOMETRY_LATENCY)
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  return DVO_AMD_OK;
}



}  // namespace host
}  // namespace dvo_amd

// ---- the host's _mm_rcp_ps as a table (opt-in reciprocal mode) --------------------------------------------------------------
// The reference forms 1 / z of the projection and the t-distribution weights with rcpps (dense_tracking_impl.cpp:192,700), a
// ~12-bit approximation whose bits differ between CPU vendors.  What the instruction is ON THIS HOST is probed once: all 2^23
// mantissas of [1, 2) give the smallest k such that rcpps(1.m) depends on the top k mantissa bits only (11 on the Xeons and
// EPYCs probed so far: 2^11 entries; in the worst case k = 23 and the table is the function itself, 32 MB); a sample of
// exponents confirms rcpps(x 2^e) = rcpps(x) 2^-e and the special cases the device code models (rcp_host_table in
// dvo_kernels.hip).  If the host's instruction does not have that structure the mode is refused, with the reason.
namespace {
struct HostRcp {
  int k = 0;
  std::vector<uint32_t> table;
  std::string problem;
};
inline uint32_t f32_bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
inline float f32_from(uint32_t u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
inline uint32_t host_rcpps_bits(uint32_t x) { return f32_bits(_mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(f32_from(x))))); }
// what rcp_host_table (device) computes, on the host: the model the probe checks the instruction against
uint32_t rcp_model(const HostRcp &h, uint32_t u) {
  const uint32_t au = u & 0x7fffffffu, e = au >> 23, m = au & 0x7fffffu;
  const uint32_t t = h.table[m >> (23 - h.k)];
  const int re = (int)(t >> 23) - ((int)e - 127);
  uint32_t r = re >= 1 ? (((uint32_t)re << 23) | (t & 0x7fffffu)) : 0u;
  r = e == 0u ? 0x7f800000u : r;
  r = e == 255u ? (m ? (au | 0x00400000u) : 0u) : r;
  return r | (u & 0x80000000u);
}
const HostRcp &host_rcp_table() {
  static HostRcp h;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<uint32_t> all((size_t)1 << 23);
    for (uint32_t m = 0; m < (1u << 23); ++m) all[m] = host_rcpps_bits(0x3f800000u | m);
    int k = 0;
    for (k = 0; k <= 23; ++k) {  // smallest k such that the result is constant over every run of 2^(23-k) mantissas
      bool ok = true;
      const uint32_t span = 1u << (23 - k);
      for (uint32_t base = 0; base < (1u << 23) && ok; base += span)
        for (uint32_t lo = 1; lo < span; ++lo)
          if (all[base + lo] != all[base]) {
            ok = false;
            break;
          }
      if (ok) break;
    }
    h.k = k;
    h.table.resize((size_t)1 << k);
    for (uint32_t i = 0; i < (1u << k); ++i) h.table[i] = all[(size_t)i << (23 - k)];
    for (uint32_t t : h.table)
      if ((t >> 23) != 126u && (t >> 23) != 127u) h.problem = "rcpps(1.m) left (0.5, 1]";
    // the model against the instruction: every exponent x a stride of mantissas, both signs, and the special inputs
    for (uint32_t e = 0; e <= 255 && h.problem.empty(); ++e)
      for (uint32_t m = 0; m < (1u << 23); m += 9973u) {
        const uint32_t x = (e << 23) | m;
        for (int neg = 0; neg < 2; ++neg) {
          const uint32_t sgn = neg ? 0x80000000u : 0u;
          const uint32_t want = host_rcpps_bits(x | sgn), got = rcp_model(h, x | sgn);
          const bool both_nan = (want & 0x7fffffffu) > 0x7f800000u && (got & 0x7fffffffu) > 0x7f800000u;
          if (want != got && !both_nan) {
            char buf[160];
            std::snprintf(buf, sizeof(buf), "_mm_rcp_ps(0x%08x) = 0x%08x on this host, the table model gives 0x%08x", x | sgn, want, got);
            h.problem = buf;
          }
        }
      }
  });
  return h;
}
}  // namespace

using namespace dvo_amd;
using namespace dvo_amd::host;


// ------------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------------
extern "C" {

int dvo_amd_abi_version(void) { return DVO_AMD_ABI_VERSION; }

#ifndef DVO_AMD_BUILD_ID
#define DVO_AMD_BUILD_ID "unknown"
#endif
// (behind a marker, so that the id can also be read from the file without loading it: dvo_slam_amd/_build.py library_id)
static const char kBuildIdString[] = "DVO_AMD_BUILD_ID=" DVO_AMD_BUILD_ID ";";
const char *dvo_amd_build_id(void) {
  static const std::string id(kBuildIdString + sizeof("DVO_AMD_BUILD_ID=") - 1, sizeof(kBuildIdString) - sizeof("DVO_AMD_BUILD_ID=") - 1);
  return id.c_str();
}

const char *dvo_amd_status_string(int s) {
  switch (s) {
    case DVO_AMD_OK: return "ok";
    case DVO_AMD_ERR_INVALID_ARGUMENT: return "invalid argument";
    case DVO_AMD_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case DVO_AMD_ERR_HIP: return "HIP runtime error";
    case DVO_AMD_ERR_OUT_OF_MEMORY: return "out of device memory";
    case DVO_AMD_ERR_INSANE_CONFIG: return "configuration is not sane (FirstLevel < LastLevel)";
    case DVO_AMD_ERR_TOO_FEW_LEVELS: return "pyramid has fewer levels than FirstLevel + 1";
    case DVO_AMD_ERR_CAPACITY: return "caller-provided array too small";
    case DVO_AMD_ERR_DEVICE_MISMATCH: return "pyramid and context live on different devices";
    case DVO_AMD_ERR_NAN_INIT: return "initial estimate is NaN";
    case DVO_AMD_ERR_COMM: return "communicator error";
    case DVO_AMD_ERR_IO: return "file cannot be opened or read";
    case DVO_AMD_ERR_FORMAT: return "unsupported or corrupt file format";
    default: return "unknown status";
  }
}

const char *dvo_amd_last_error(void) { return g_last_error.c_str(); }

int dvo_amd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void dvo_amd_default_config(dvo_amd_config *c) {
  if (!c) return;
  c->first_level = 3;
  c->last_level = 1;
  c->max_iterations_per_level = 100;
  c->precision = 5e-7;
  c->mu = 0.0;
  c->use_initial_estimate = 0;
  c->intensity_derivative_threshold = 0.0f;
  c->depth_derivative_threshold = 0.0f;
  c->segment_geometry = DVO_AMD_GEOMETRY_THROUGHPUT;
  c->reserved = 0;
}

int dvo_amd_context_create(int device, const dvo_amd_config *cfg, dvo_amd_context **out) {
  if (!out) return DVO_AMD_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DVO_AMD_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev || device >= kMaxDevices) return DVO_AMD_ERR_INVALID_ARGUMENT;
  dvo_amd_config c;
  dvo_amd_default_config(&c);
  if (cfg) c = *cfg;
  int rc = check_config(&c);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(device));
  dvo_amd_context *ctx = new dvo_amd_context();
  ctx->device = device;
  ctx->cfg = c;
  hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete ctx;
    return fail_hip("hipStreamCreate", e);
  }
  const char *pe = getenv("DVO_AMD_POLL");
  ctx->poll = !(pe && pe[0] == '0');
  const char *fs = getenv("DVO_AMD_FIN_STAMPS");
  ctx->fin_stamps = fs && fs[0] == '1';
  const char *fp = getenv("DVO_AMD_FIN_PRIORITY");
  ctx->fin_priority = !(fp && fp[0] == '0');
  if (const char *so = getenv("DVO_AMD_SORT_ITEMS")) ctx->sort_items = so[0] != '0';
  const char *sa = getenv("DVO_AMD_SMALL_ARGS");
  ctx->small_args = !(sa && sa[0] == '0');
  const char *hp = getenv("DVO_AMD_HOST_PROF");
  ctx->host_prof = hp && hp[0] == '1';
  if (const char *sl = getenv("DVO_AMD_SPEC_LEVELS")) ctx->spec_levels = sl[0] == '1' ? 1 : 0;
  if (const char *fs2 = getenv("DVO_AMD_FAULT_SLOT_ALLOC")) ctx->fault_slot_alloc = atoi(fs2);
  if (const char *sa = getenv("DVO_AMD_LEVEL_STEPS_AT"))
    (void)sscanf(sa, "%lld,%lld,%lld,%lld,%lld", &ctx->level_steps_at[0], &ctx->level_steps_at[1], &ctx->level_steps_at[2],
                 &ctx->level_steps_at[3], &ctx->level_steps_at[4]);
  if (const char *fsx = getenv("DVO_AMD_FINE_STEPS")) ctx->fine_steps = atoi(fsx);  // (tuning) see level_steps
  if (const char *lm = getenv("DVO_AMD_LL_MERGE")) {
    const int v = atoi(lm);
    if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) ctx->ll_merge = v;
  }
  if (const char *rm = getenv("DVO_AMD_RCP"))  // DVO_AMD_RCP=host: every new tracker starts in the host-rcpps mode
    if (rm[0] == 'h' || rm[0] == 'H' || rm[0] == 's' || rm[0] == 'S') {
      const int rrc = dvo_amd_set_reciprocal_mode(ctx, DVO_AMD_RCP_HOST_SSE);
      if (rrc) {
        dvo_amd_context_destroy(ctx);
        return rrc;
      }
    }
  if (const char *ipl = getenv("DVO_AMD_ITEMS_PER_LAUNCH")) {
    const int v = atoi(ipl);
    if (v >= 1 && v <= kMaxItemsPerLaunch) ctx->items_per_launch = v;
  }
  *out = ctx;
  return DVO_AMD_OK;
}

void dvo_amd_context_destroy(dvo_amd_context *ctx) {
  if (!ctx) return;
  if (ctx->host_prof && ctx->prof_ticks > 0)
    std::fprintf(stderr, "[dvo_amd host profile] ticks %lld, pair-ticks %lld: submit %.2f us/tick, wait %.2f us/tick, process %.2f us/tick "
                         "(%.2f us per pair-tick of host work)\n",
                 ctx->prof_ticks, ctx->prof_job_ticks, ctx->prof_submit_ns / ctx->prof_ticks * 1e-3,
                 ctx->prof_wait_ns / ctx->prof_ticks * 1e-3, ctx->prof_process_ns / ctx->prof_ticks * 1e-3,
                 (ctx->prof_submit_ns + ctx->prof_process_ns) / std::max(1LL, ctx->prof_job_ticks) * 1e-3);
  if (ctx->host_prof && ctx->ovf_checks > 0)
    std::fprintf(stderr, "[dvo_amd host profile] exact likelihood-overflow checks: %lld (%lld said the reference's product overflowed)\n",
                 ctx->ovf_checks, ctx->ovf_hits);
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->runner) {
    // pairs still queued are dropped (their results stay unfinished): stop the kernels, give the pyramids back
    if (ctx->runner->resident > 0 || !ctx->runner->pending.empty()) (void)runner_fail(ctx, DVO_AMD_ERR_INVALID_ARGUMENT);
    delete ctx->runner;
    ctx->runner = nullptr;
  }
  dvo_amd_comm_destroy(ctx);
  dvo_amd_exchange_destroy(ctx);
  for (hipStream_t st : ctx->extra_streams) {
    (void)hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
  }
  if (ctx->desc_ready) (void)hipEventDestroy(ctx->desc_ready);
  release_slots(ctx);
  if (ctx->out_wire) (void)hipHostFree(ctx->out_wire);
  if (ctx->ovf_host) (void)hipHostFree(ctx->ovf_host);
  if (ctx->rcp_table_dev) (void)hipFree(ctx->rcp_table_dev);
  if (ctx->rcp_nibbles_dev) (void)hipFree(ctx->rcp_nibbles_dev);
  if (ctx->dbg_w_dev) (void)hipFree(ctx->dbg_w_dev);
  for (auto &ev : ctx->events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int dvo_amd_configure(dvo_amd_context *ctx, const dvo_amd_config *cfg) {
  if (!ctx) return DVO_AMD_ERR_INVALID_ARGUMENT;
  int rc = check_config(cfg);
  if (rc) return rc;
  if (ctx->runner && (ctx->runner->resident > 0 || !ctx->runner->pending.empty())) {
    g_last_error = "dvo_amd_configure while submitted pairs are still in flight (dvo_amd_match_wait first)";
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  }
  ctx->cfg = *cfg;
  return DVO_AMD_OK;
}

int dvo_amd_set_reciprocal_mode(dvo_amd_context *ctx, int mode) {
  if (!ctx || (mode != DVO_AMD_RCP_EXACT && mode != DVO_AMD_RCP_HOST_SSE)) return DVO_AMD_ERR_INVALID_ARGUMENT;
  int rc = queue_must_be_idle(ctx, "dvo_amd_set_reciprocal_mode");
  if (rc) return rc;
  if (mode == DVO_AMD_RCP_EXACT) {
    ctx->rcp = RcpTable{nullptr, 0, 0, nullptr};
    return DVO_AMD_OK;
  }
  if (acc_mode() == 0) {  // (the mode used to run the matrix-pipe accumulator silently under the switch)
    g_last_error = "the host-rcpps mode is built for the default accumulator only: unset DVO_AMD_ACCUM=valu";
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  }
  const HostRcp &h = host_rcp_table();
  if (!h.problem.empty()) {
    g_last_error = "the host's _mm_rcp_ps cannot be reproduced from a table: " + h.problem;
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  if (!ctx->rcp_table_dev) {
    HIP_TRY(hipMalloc((void **)&ctx->rcp_table_dev, sizeof(uint32_t) * h.table.size()));
    HIP_TRY(hipMemcpy(ctx->rcp_table_dev, h.table.data(), sizeof(uint32_t) * h.table.size(), hipMemcpyHostToDevice));
  }
  // The nibble form (round 5): rcpps(1.m) = (the device's v_rcp_f32 of the cell's midpoint, bits below `unit` cleared) +
  // correction << unit, the corrections in LDS instead of a gather from global memory in the dependent chain of every step.  The
  // corrections are formed here against the device's OWN reciprocal, evaluated by a probe kernel in both rounding modes the residual
  // pass uses it in; if the two differ, a correction leaves [-8, 7], or the table has more than 2^12 cells, the table form stays.
  if (!ctx->rcp_nibbles_dev && ctx->rcp_form_note.empty()) {
    const char *form = getenv("DVO_AMD_RCP_FORM");  // "table": keep the global-memory table (A/B)
    const int n = 1 << h.k;
    if (form && (form[0] == 't' || form[0] == 'T')) {
      ctx->rcp_form_note = "DVO_AMD_RCP_FORM=table";
    } else if (h.k > 12 || h.k < 4) {
      ctx->rcp_form_note = "rcpps depends on more than 12 mantissa bits on this host";
    } else {
      unsigned *probe = nullptr;
      HIP_TRY(hipMalloc((void **)&probe, sizeof(unsigned) * 2 * (size_t)n));
      std::vector<unsigned> mid(2 * (size_t)n);
      hipError_t e = launch_rcp_midpoint_probe(h.k, probe, probe + n, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e == hipSuccess) e = hipMemcpy(mid.data(), probe, sizeof(unsigned) * mid.size(), hipMemcpyDeviceToHost);
      (void)hipFree(probe);
      if (e != hipSuccess) return fail_hip("rcp midpoint probe", e);
      uint32_t low = 0;
      for (uint32_t t : h.table) low |= t & 0x7fffffu;
      int unit = 0;
      while (unit < 22 && !((low >> unit) & 1u)) ++unit;  // (every entry a power of two: unit stops at 22)
      std::vector<unsigned> words((size_t)std::max(n / 8, 1), 0u);
      for (int i = 0; i < n && ctx->rcp_form_note.empty(); ++i) {
        if (mid[(size_t)i] != mid[(size_t)n + i]) {
          ctx->rcp_form_note = "v_rcp_f32 depends on the rounding mode";
          break;
        }
        const int64_t base = (int64_t)(mid[(size_t)i] & ~((1u << unit) - 1u));
        const int64_t diff = (int64_t)h.table[(size_t)i] - base;
        if (diff % (1ll << unit) != 0 || diff / (1ll << unit) < -8 || diff / (1ll << unit) > 7) {
          ctx->rcp_form_note = "a correction does not fit four bits";
          break;
        }
        const unsigned nib = (unsigned)((diff / (1ll << unit)) & 15);
        words[(size_t)(i >> 3)] |= nib << ((i & 7) * 4);
      }
      if (ctx->rcp_form_note.empty()) {
        HIP_TRY(hipMalloc((void **)&ctx->rcp_nibbles_dev, sizeof(unsigned) * words.size()));
        HIP_TRY(hipMemcpy(ctx->rcp_nibbles_dev, words.data(), sizeof(unsigned) * words.size(), hipMemcpyHostToDevice));
        ctx->rcp_unit = unit;
      }
    }
  }
  ctx->rcp = RcpTable{ctx->rcp_table_dev, 23 - h.k, ctx->rcp_unit, ctx->rcp_nibbles_dev};
  return DVO_AMD_OK;
}

int dvo_amd_get_reciprocal_mode(const dvo_amd_context *ctx, int *mode, int *table_mantissa_bits) {
  if (!ctx) return DVO_AMD_ERR_INVALID_ARGUMENT;
  if (mode) *mode = ctx->rcp.table ? DVO_AMD_RCP_HOST_SSE : DVO_AMD_RCP_EXACT;
  if (table_mantissa_bits) *table_mantissa_bits = ctx->rcp.table ? 23 - ctx->rcp.shift : 0;
  return DVO_AMD_OK;
}

int dvo_amd_context_device(const dvo_amd_context *ctx, int *device) {
  if (!ctx || !device) return DVO_AMD_ERR_INVALID_ARGUMENT;
  *device = ctx->device;
  return DVO_AMD_OK;
}

int dvo_amd_get_config(const dvo_amd_context *ctx, dvo_amd_config *cfg) {
  if (!ctx || !cfg) return DVO_AMD_ERR_INVALID_ARGUMENT;
  *cfg = ctx->cfg;
  return DVO_AMD_OK;
}

int dvo_amd_match_submit(dvo_amd_context *ctx, int n, dvo_amd_pyramid *const *references, dvo_amd_pyramid *const *currents,
                         const double *T_inits, dvo_amd_result *results, int max_in_flight, unsigned long long *ticket) {
  if (!ctx || !ticket || n < 0 || (n > 0 && (!references || !currents || !results))) return DVO_AMD_ERR_INVALID_ARGUMENT;
  *ticket = 0;
  const dvo_amd_config &cfg = ctx->cfg;
  int rc = check_config(&cfg);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  const int need_levels = cfg.first_level + 1;  // Config::getNumLevels
  int n_pad = 0;
  if (cfg.use_initial_estimate && n > 0 && !T_inits) return DVO_AMD_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < n; ++i) {
    if (!references[i] || !currents[i]) return DVO_AMD_ERR_INVALID_ARGUMENT;
    if (references[i]->device != ctx->device || currents[i]->device != ctx->device) return DVO_AMD_ERR_DEVICE_MISMATCH;
    if (references[i]->n_levels < need_levels || currents[i]->n_levels < need_levels) return DVO_AMD_ERR_TOO_FEW_LEVELS;
    for (int l = cfg.last_level; l <= cfg.first_level; ++l)
      if (references[i]->lv[l].w != currents[i]->lv[l].w || references[i]->lv[l].h != currents[i]->lv[l].h)
        return DVO_AMD_ERR_INVALID_ARGUMENT;
    n_pad = std::max(n_pad, references[i]->lv[cfg.last_level].n_pad);
    const int its_needed = (cfg.first_level - cfg.last_level + 1) * (cfg.max_iterations_per_level + 1);
    if (results[i].iterations && results[i].iterations_capacity > 0 && results[i].iterations_capacity < its_needed)
      return DVO_AMD_ERR_CAPACITY;
    if (cfg.use_initial_estimate) {  // dense_tracking.cpp:139
      double s = 0.0;
      for (int k = 0; k < 16; ++k) s += T_inits[16 * (size_t)i + k];
      if (!std::isfinite(s)) return DVO_AMD_ERR_NAN_INIT;
    }
  }
  if (!ctx->runner) ctx->runner = new Runner();
  Runner &R = *ctx->runner;
  if (n == 0) {  // an empty batch is complete at once
    *ticket = R.next_batch++;
    return DVO_AMD_OK;
  }
  rc = runner_configure(ctx, max_in_flight <= 0 ? n : max_in_flight, n_pad);
  if (rc) return rc;
  Batch b;
  b.id = R.next_batch++;
  b.remaining = n;
  R.batches.push_back(b);
  for (int i = 0; i < n; ++i) {
    Pending q;
    q.ref = references[i], q.cur = currents[i], q.result = &results[i], q.batch = b.id;
    q.has_init = cfg.use_initial_estimate != 0;
    q.ti = cfg.intensity_derivative_threshold, q.td = cfg.depth_derivative_threshold;
    if (q.has_init) std::memcpy(q.T_init, T_inits + 16 * (size_t)i, sizeof(q.T_init));
    // the queue holds its own references: the caller may release a pyramid right after submitting
    dvo_amd_pyramid_retain(q.ref);
    dvo_amd_pyramid_retain(q.cur);
    R.pending.push_back(q);
  }
  *ticket = b.id;
  // start what can start without waiting: groups that have no tick in flight take pending pairs and launch
  for (size_t g = 0; g < R.groups.size(); ++g)
    if (!R.groups[g].in_flight) {
      rc = runner_step(ctx, g);
      if (rc) return runner_fail_told(ctx, rc);
    }
  return DVO_AMD_OK;
}

int dvo_amd_match_wait(dvo_amd_context *ctx, unsigned long long ticket) {
  if (!ctx) return DVO_AMD_ERR_INVALID_ARGUMENT;
  if (!ctx->runner) return ticket == 0 ? DVO_AMD_OK : DVO_AMD_ERR_INVALID_ARGUMENT;
  Runner &R = *ctx->runner;
  if (ticket >= R.next_batch) return DVO_AMD_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(ctx->device));
  // ticket 0: everything submitted so far
  auto open_batch = [&]() -> bool {
    for (const Batch &b : R.batches)
      if ((ticket == 0 || b.id == ticket) && b.remaining > 0) return true;
    return false;
  };
  while (open_batch()) {
    const size_t g = R.next_group;
    R.next_group = (R.next_group + 1) % R.groups.size();
    int rc = runner_step(ctx, g);
    if (rc) return runner_fail_told(ctx, rc);
  }
  while (!R.batches.empty() && R.batches.front().remaining == 0) R.batches.pop_front();
  return runner_reported_status(R, ticket);
}

int dvo_amd_match_poll(dvo_amd_context *ctx, unsigned long long ticket, int *done) {
  if (!ctx || !done) return DVO_AMD_ERR_INVALID_ARGUMENT;
  *done = 1;
  if (!ctx->runner) return ticket == 0 ? DVO_AMD_OK : DVO_AMD_ERR_INVALID_ARGUMENT;
  Runner &R = *ctx->runner;
  if (ticket >= R.next_batch) return DVO_AMD_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(ctx->device));
  // advance every group whose tick has landed (or that has nothing in flight); never waits for the GPU
  for (size_t g = 0; g < R.groups.size(); ++g) {
    if (R.groups[g].in_flight && !tick_landed(ctx, R.jobs, R.groups[g])) continue;
    int rc = runner_step(ctx, g);
    if (rc) return runner_fail_told(ctx, rc);
  }
  for (const Batch &b : R.batches)
    if ((ticket == 0 || b.id == ticket) && b.remaining > 0) *done = 0;
  if (*done)
    while (!R.batches.empty() && R.batches.front().remaining == 0) R.batches.pop_front();
  return runner_reported_status(R, ticket);
}

int dvo_amd_match_many(dvo_amd_context *ctx, int n, dvo_amd_pyramid *const *references, dvo_amd_pyramid *const *currents,
                       const double *T_inits, dvo_amd_result *results, int max_in_flight) {
  if (!ctx || n < 0 || (n > 0 && (!references || !currents || !results))) return DVO_AMD_ERR_INVALID_ARGUMENT;
  if (n == 0) return DVO_AMD_OK;
  unsigned long long ticket = 0;
  int rc = dvo_amd_match_submit(ctx, n, references, currents, T_inits, results, max_in_flight, &ticket);
  if (rc) return rc;
  return dvo_amd_match_wait(ctx, ticket);
}

int dvo_amd_match_batch(dvo_amd_context *ctx, int n, dvo_amd_pyramid *const *references, dvo_amd_pyramid *const *currents,
                        const double *T_inits, dvo_amd_result *results) {
  return dvo_amd_match_many(ctx, n, references, currents, T_inits, results, 0);
}

int dvo_amd_match(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, const double *T_init,
                  dvo_amd_result *result) {
  dvo_amd_pyramid *r[1] = {reference}, *c[1] = {current};
  return dvo_amd_match_batch(ctx, 1, r, c, T_init, result);
}

int dvo_amd_match_selection(dvo_amd_context *ctx, dvo_amd_pyramid *reference, float intensity_threshold, float depth_threshold,
                            dvo_amd_pyramid *current, const double *T_init, dvo_amd_result *result) {
  if (!ctx) return DVO_AMD_ERR_INVALID_ARGUMENT;
  {
    int rc = queue_must_be_idle(ctx, "dvo_amd_match_selection");
    if (rc) return rc;
  }
  // the PointSelection's predicate decides which reference pixels take part, not the tracker's configuration
  // (dense_tracking.cpp:131,226: reference.select(level)); a context is single-threaded by contract
  const float keep_i = ctx->cfg.intensity_derivative_threshold, keep_d = ctx->cfg.depth_derivative_threshold;
  ctx->cfg.intensity_derivative_threshold = intensity_threshold;
  ctx->cfg.depth_derivative_threshold = depth_threshold;
  const int rc = dvo_amd_match(ctx, reference, current, T_init, result);
  ctx->cfg.intensity_derivative_threshold = keep_i;
  ctx->cfg.depth_derivative_threshold = keep_d;
  return rc;
}

void dvo_amd_se3_exp(const double *xi, double *T) { se3_matrix(se3_exp(xi), T); }
void dvo_amd_se3_log(const double *T, double *xi) { se3_log(se3_from_matrix(T), xi); }
void dvo_amd_solve6(const double *A, const double *b, double *x) { solve_ldlt6(A, b, x); }

}  // extern "C"
