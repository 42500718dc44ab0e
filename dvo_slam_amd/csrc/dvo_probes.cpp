// Entries that work in slot 0 of a context outside the match driver: dvo_amd_residuals / dvo_amd_error_image
// (computeResidualsAndValidFlagsSse, computeIntensityErrorImage: dense_tracking_impl.cpp:400-403, dense_tracking.cpp:378-444) and the
// test probes, micro-benchmarks and diagnostics declared in include/dvo_amd_debug.h.
#include <dlfcn.h>
#include <emmintrin.h>  // the host side of the record hand-off takes 16 bytes at a time (x86-64 hosts)

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "dvo_internal.h"

using namespace dvo_amd;
using namespace dvo_amd::host;

extern "C" {

int dvo_amd_debug_rcp_form(const dvo_amd_context *ctx, int *form, char *note, int note_capacity) {
  if (!ctx || !form) return DVO_AMD_ERR_INVALID_ARGUMENT;
  *form = ctx->rcp.nibbles ? 2 : ctx->rcp.table ? 1 : 0;
  if (note && note_capacity > 0) std::snprintf(note, (size_t)note_capacity, "%s", ctx->rcp_form_note.c_str());
  return DVO_AMD_OK;
}

int dvo_amd_debug_rcp(dvo_amd_context *ctx, int n, const float *in, float *out) {
  if (!ctx || n < 0 || (n > 0 && (!in || !out))) return DVO_AMD_ERR_INVALID_ARGUMENT;
  if (!ctx->rcp.table) {
    g_last_error = "dvo_amd_debug_rcp: the host-rcpps mode is not on (dvo_amd_set_reciprocal_mode)";
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  }
  if (n == 0) return DVO_AMD_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  float *d_in = nullptr, *d_out = nullptr;
  HIP_TRY(hipMalloc((void **)&d_in, sizeof(float) * (size_t)n));
  hipError_t e = hipMalloc((void **)&d_out, sizeof(float) * (size_t)n);
  if (e == hipSuccess) e = hipMemcpyAsync(d_in, in, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = launch_rcp_table_probe(ctx->rcp, d_in, d_out, n, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  if (e != hipSuccess) return fail_hip("dvo_amd_debug_rcp", e);
  return DVO_AMD_OK;
}

unsigned dvo_amd_debug_next_seq(unsigned seq) { return next_seq(seq); }

int dvo_amd_debug_wire_layout(int *n_pieces, int *n_record_words) {
  if (!n_pieces || !n_record_words) return DVO_AMD_ERR_INVALID_ARGUMENT;
  *n_pieces = kFinWirePieces, *n_record_words = kFinWords;
  return DVO_AMD_OK;
}

int dvo_amd_debug_take_wire(const unsigned *wire, unsigned tick, int from_piece, unsigned *record_words) {
  if (!wire || !record_words || from_piece < 0 || from_piece > kFinWirePieces || (reinterpret_cast<uintptr_t>(wire) & 15u))
    return -DVO_AMD_ERR_INVALID_ARGUMENT;
  FinOut rec;
  std::memcpy(&rec, record_words, sizeof(rec));
  const int next = take_wire(reinterpret_cast<const FinWire *>(wire), &rec, tick, from_piece);
  std::memcpy(record_words, &rec, sizeof(rec));
  return next;
}

}  // extern "C"

namespace dvo_amd {
namespace host {

// One k_tick + k_finalize over slot 0 outside the match driver (the stage-wise parity entries): optionally the residual pass
// at the float transform T (into residual buffer 0) and / or the log-likelihood pass over residual buffer 0.
int single_tick(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, int level, const Selection *sel, const float *T,
                const float P[4], bool unit_weights, bool residual_pass, bool loglik_pass, int ll_cut_rank) {
  const LevelData &R = reference->lv[level];
  const LevelData &C = current->lv[level];
  JobSlot &s = ctx->slots[0];
  TickArgs ta;
  std::memset(&ta, 0, sizeof(ta));
  ta.n_items = 1;
  ta.rcp = ctx->rcp;
  TickItem &w = ta.items[0];
  w.ref = sel->ref_desc + level;
  w.cur = current->cur_desc + level;
  w.slot = ctx->slot_desc;
  const int steps = level_steps(ctx, R);
  item_set_steps(w, steps, steps);
  const int nb = level_blocks(sel, level, steps);
  if (unit_weights) w.flags |= kItemUnitWeights;
  if (P) std::memcpy(w.P, P, sizeof(w.P));
  FinArgs fa;
  std::memset(&fa, 0, sizeof(fa));
  fa.n_items = 1;
  FinItem &f = fa.items[0];
  f.ll_partials = s.ll_partials;
  f.ll_qmax_off = s.ll_qmax_off;
  f.seg_prefix_out = s.seg_prefix[0];
  f.out = s.out;
  f.out_dev = nullptr;
  f.seq = ctx->tick_seq = next_seq(ctx->tick_seq);
  if (residual_pass) {
    w.res_blocks = (uint16_t)nb;
    const float K[9] = {C.fx, 0.0f, C.ox, 0.0f, C.fy, C.oy, 0.0f, 0.0f, 1.0f};
    for (int i = 0; i < 3; ++i)
      for (int c = 0; c < 4; ++c)
        w.kt[i * 4 + c] = (K[i * 3 + 0] * T[c * 4 + 0] + K[i * 3 + 1] * T[c * 4 + 1]) + K[i * 3 + 2] * T[c * 4 + 2];
    f.records = s.records;
    f.n_blocks = (uint16_t)nb, f.level_blocks = (uint16_t)nb;
  }
  if (loglik_pass) {  // the merged blocks match() runs for this level (level_ll_merge)
    item_set_ll_merge(w, level_ll_merge(ctx, steps));
    w.ll_level_blocks = (uint16_t)nb;
    w.ll_blocks = (uint16_t)ll_blocks_total(nb, item_ll_merge_log2(w));
    w.ll_cut_rank = ll_cut_rank;
    f.n_ll_blocks = w.ll_blocks, f.ll_level_blocks = (uint16_t)nb, f.ll_merge_log2 = (uint16_t)item_ll_merge_log2(w);
  }
  hipError_t e = launch_tick(ta, (int)w.res_blocks + (int)w.ll_blocks, ctx->stream);
  if (e != hipSuccess) return fail_hip("launch_tick", e);
  if (ctx->rcp.table && residual_pass && !unit_weights) {  // host-rcpps mode: the Q7 tail, as submit_tick runs it
    Q7ArgsSmall qs;
    qs.n_items = 1, qs.q7_off256 = ctx->q7_off256, qs.rcp = ctx->rcp;
    for (int i = 0; i < kMaxSmallItems; ++i) qs.items[i] = w;
    e = launch_q7_tail_small(qs, ctx->stream);
    if (e != hipSuccess) return fail_hip("launch_q7_tail", e);
    f.q7_off256 = (uint16_t)ctx->q7_off256;
  }
  e = launch_finalize(fa, ctx->stream);
  if (e != hipSuccess) return fail_hip("launch_finalize", e);
  return DVO_AMD_OK;
}

// Per-point results of a pass (the residual pass walks the compacted selection, k_compact) back in the image: `floats_per` floats
// per point from device memory to their pixels, NaN everywhere else -- the layout the entries below have always returned.
int scatter_points_to_image(dvo_amd_context *ctx, const Selection *sel, int level, int n_pixels, const float *points_dev, int floats_per,
                            float *image, std::vector<int> *pix_out = nullptr) {
  const int n_pts = sel->n_pts[level];
  std::vector<int> pix((size_t)std::max(n_pts, 1));
  std::vector<float> pts((size_t)std::max(n_pts, 1) * floats_per);
  if (n_pts > 0) {
    HIP_TRY(hipMemcpyAsync(pix.data(), sel->pts[level].pix, sizeof(int) * (size_t)n_pts, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(pts.data(), points_dev, sizeof(float) * (size_t)n_pts * floats_per, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (image) {
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    std::fill(image, image + (size_t)n_pixels * floats_per, qnan);
    for (int p = 0; p < n_pts; ++p) {
      if (pix[p] < 0 || pix[p] >= n_pixels) {
        g_last_error = "compacted selection: a point without a pixel";
        return DVO_AMD_ERR_HIP;
      }
      for (int k = 0; k < floats_per; ++k) image[(size_t)pix[p] * floats_per + k] = pts[(size_t)p * floats_per + k];
    }
  }
  if (pix_out) pix.resize((size_t)n_pts), pix_out->swap(pix);
  return DVO_AMD_OK;
}

int check_level_pair(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, int level) {
  if (level >= reference->n_levels || level >= current->n_levels) return DVO_AMD_ERR_TOO_FEW_LEVELS;
  if (reference->device != ctx->device || current->device != ctx->device) return DVO_AMD_ERR_DEVICE_MISMATCH;
  if (reference->lv[level].w != current->lv[level].w || reference->lv[level].h != current->lv[level].h)
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  return DVO_AMD_OK;
}

}  // namespace host
}  // namespace dvo_amd

extern "C" {

int dvo_amd_residuals(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, int level, const float *T,
                      float *residuals, int *n_valid) {
  if (!ctx || !reference || !current || !T || level < 0) return DVO_AMD_ERR_INVALID_ARGUMENT;
  int rc = check_level_pair(ctx, reference, current, level);
  if (rc) return rc;
  rc = queue_must_be_idle(ctx, "dvo_amd_residuals / dvo_amd_error_image");
  if (rc) return rc;
  const LevelData &R = reference->lv[level];
  HIP_TRY(hipSetDevice(ctx->device));
  const Selection *sel = nullptr;
  rc = pyramid_selection(reference, ctx->cfg.intensity_derivative_threshold, ctx->cfg.depth_derivative_threshold, &sel);
  if (rc) return rc;
  rc = ensure_slots(ctx, 1, R.n_pad);
  if (rc) return rc;
  rc = single_tick(ctx, reference, current, level, sel, T, nullptr, true, true, false, 0);
  if (rc) return rc;
  if (residuals) {  // (the spill is per point of the compacted selection; an unselected pixel reads NaN as it always has)
    rc = scatter_points_to_image(ctx, sel, level, R.n, (const float *)ctx->slots[0].res[0], 2, residuals);
    if (rc) return rc;
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  rc = take_record_synced(ctx, 0, ctx->tick_seq);
  if (rc) return rc;
  if (n_valid) *n_valid = ctx->out_host[0].valid;
  return DVO_AMD_OK;
}

int dvo_amd_debug_iteration(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, int level,
                            const float *T, const float *precision_in, const float *precision_eval,
                            dvo_amd_iteration_probe *out) {
  if (!ctx || !reference || !current || !T || !out || level < 0) return DVO_AMD_ERR_INVALID_ARGUMENT;
  int rc = check_level_pair(ctx, reference, current, level);
  if (rc) return rc;
  rc = queue_must_be_idle(ctx, "dvo_amd_debug_iteration");
  if (rc) return rc;
  const LevelData &R = reference->lv[level];
  HIP_TRY(hipSetDevice(ctx->device));
  const Selection *sel = nullptr;
  rc = pyramid_selection(reference, ctx->cfg.intensity_derivative_threshold, ctx->cfg.depth_derivative_threshold, &sel);
  if (rc) return rc;
  rc = ensure_slots(ctx, 1, R.n_pad);
  if (rc) return rc;
  std::memset(out, 0, sizeof(*out));
  // tick 1: residuals, weights (unit, or from precision_in), pair-quirk scale sums, the 87 moments
  rc = single_tick(ctx, reference, current, level, sel, T, precision_in, precision_in == nullptr, true, false, 0);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  rc = take_record_synced(ctx, 0, ctx->tick_seq);
  if (rc) return rc;
  const FinOut o = ctx->out_host[0];
  out->valid_constraints = o.valid;
  for (int i = 0; i < 3; ++i) out->scale_sums[i] = o.S[i];
  for (int i = 0; i < kNumAcc; ++i) out->moments[i] = o.acc[i];
  if (o.valid < 6) return DVO_AMD_OK;  // dense_tracking.cpp:276-284
  float cov[4], P[4];
  scale_and_precision(o, o.valid, cov, P);
  std::memcpy(out->scale, cov, sizeof(cov));
  std::memcpy(out->precision, P, sizeof(P));
  // the normal equations and the likelihood are evaluated with this iteration's precision -- the one just computed, or the
  // caller's (a checker that wants to separate "is P right" from "are the sums right" passes its own)
  if (precision_eval) std::memcpy(P, precision_eval, sizeof(P));
  const double zero6[6] = {0, 0, 0, 0, 0, 0};
  system_from_moments(o, P, 0.0, zero6, out->information, out->rhs);
  // tick 2: the log-likelihood of the same residuals under the new precision, cut at 50 * floor(V / 50) (Q6)
  rc = single_tick(ctx, reference, current, level, sel, T, P, false, false, true, 50 * (o.valid / 50));
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  rc = take_record_synced(ctx, 0, ctx->tick_seq);
  if (rc) return rc;
  out->loglik_sum = ctx->out_host[0].ll_sum;
  bool overflowed = false;
  if (ctx->out_host[0].ll_qmax >= kLlOverflowScreen) {
    const int st = level_steps(ctx, R);  // (the geometry single_tick used)
    rc = ll_overflowed(ctx, ctx->slots[0].res[0], ctx->slots[0].seg_prefix[0], level_blocks(sel, level, st), st, 50 * (o.valid / 50), P, nullptr, 0,
                       &overflowed);
    if (rc) return rc;
  }
  out->loglik = loglik_from_sum(o.valid, P, out->loglik_sum, overflowed);
  return DVO_AMD_OK;
}

int dvo_amd_debug_weights(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, int level, const float *T,
                          const float *precision_in, float *weights, dvo_amd_q7_probe *tail) {
  if (!ctx || !reference || !current || !T || !precision_in || !weights || !tail || level < 0) return DVO_AMD_ERR_INVALID_ARGUMENT;
  if (!ctx->rcp.table) {
    g_last_error = "dvo_amd_debug_weights: only the host-rcpps kernels can store their weights (dvo_amd_set_reciprocal_mode)";
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  }
  int rc = check_level_pair(ctx, reference, current, level);
  if (rc) return rc;
  rc = queue_must_be_idle(ctx, "dvo_amd_debug_weights");
  if (rc) return rc;
  const LevelData &R = reference->lv[level];
  HIP_TRY(hipSetDevice(ctx->device));
  const Selection *sel = nullptr;
  rc = pyramid_selection(reference, ctx->cfg.intensity_derivative_threshold, ctx->cfg.depth_derivative_threshold, &sel);
  if (rc) return rc;
  rc = ensure_slots(ctx, 1, R.n_pad);
  if (rc) return rc;
  if (ctx->dbg_w_capacity < (size_t)R.n_pad) {
    if (ctx->dbg_w_dev) (void)hipFree(ctx->dbg_w_dev);
    ctx->dbg_w_dev = nullptr, ctx->dbg_w_capacity = 0;
    HIP_TRY(hipMalloc((void **)&ctx->dbg_w_dev, sizeof(float) * (size_t)R.n_pad));
    ctx->dbg_w_capacity = (size_t)R.n_pad;
  }
  // slot 0's descriptor points the residual pass at the buffer for this one tick
  char *field = reinterpret_cast<char *>(ctx->slot_desc) + offsetof(SlotDesc, dbg_w);
  float *on = ctx->dbg_w_dev, *off = nullptr;
  HIP_TRY(hipMemcpy(field, &on, sizeof(on), hipMemcpyHostToDevice));
  rc = single_tick(ctx, reference, current, level, sel, T, precision_in, false, true, false, 0);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  const hipError_t e_off = hipMemcpy(field, &off, sizeof(off), hipMemcpyHostToDevice);
  if (rc) return rc;
  if (e != hipSuccess) return fail_hip("dvo_amd_debug_weights", e);
  if (e_off != hipSuccess) return fail_hip("dvo_amd_debug_weights", e_off);
  rc = take_record_synced(ctx, 0, ctx->tick_seq);
  if (rc) return rc;
  std::vector<int> pix;
  rc = scatter_points_to_image(ctx, sel, level, R.n, ctx->dbg_w_dev, 1, weights, &pix);
  if (rc) return rc;
  Q7Rec q;
  HIP_TRY(hipMemcpy(&q, ctx->slots[0].q7, sizeof(q), hipMemcpyDeviceToHost));
  std::memset(tail, 0, sizeof(*tail));
  tail->n_tail = q.n_tail, tail->valid_constraints = ctx->out_host[0].valid, tail->recomputed_equal = q.recomputed_equal;
  for (int i = 0; i < 3; ++i)
    tail->pixel[i] = (q.idx[i] >= 0 && q.idx[i] < (int)pix.size()) ? pix[(size_t)q.idx[i]] : -1, tail->weight_table[i] = q.w_table[i], tail->weight_exact[i] = q.w_exact[i], tail->scale_sums_delta[i] = q.S[i];
  for (int i = 0; i < kNumAcc; ++i) tail->moments_delta[i] = q.acc[i];
  tail->valid_counted = q.valid;
  return DVO_AMD_OK;
}

int dvo_amd_debug_level_geometry(dvo_amd_context *ctx, dvo_amd_pyramid *reference, int level, int *steps, int *blocks, int *points) {
  if (!ctx || !reference || !steps || !blocks || !points || level < 0) return DVO_AMD_ERR_INVALID_ARGUMENT;
  if (level >= reference->n_levels) return DVO_AMD_ERR_TOO_FEW_LEVELS;
  if (reference->device != ctx->device) return DVO_AMD_ERR_DEVICE_MISMATCH;
  HIP_TRY(hipSetDevice(ctx->device));
  const Selection *sel = nullptr;
  int rc = pyramid_selection(reference, ctx->cfg.intensity_derivative_threshold, ctx->cfg.depth_derivative_threshold, &sel);
  if (rc) return rc;
  *steps = level_steps(ctx, reference->lv[level]);
  *blocks = level_blocks(sel, level, *steps);
  *points = sel->n_pts[level];
  return DVO_AMD_OK;
}

int dvo_amd_debug_hw_queue(dvo_amd_context *ctx, int *pipe_queue) {
  if (!ctx || !pipe_queue) return DVO_AMD_ERR_INVALID_ARGUMENT;
  int rc = queue_must_be_idle(ctx, "dvo_amd_debug_hw_queue");
  if (rc) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  return probe_hw_queue(ctx, pipe_queue);
}

int dvo_amd_error_image(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, const double *T,
                        int level, float *image) {
  if (!ctx || !reference || !current || !T || !image || level < 0) return DVO_AMD_ERR_INVALID_ARGUMENT;
  if (level >= reference->n_levels) return DVO_AMD_ERR_TOO_FEW_LEVELS;
  const int n = reference->lv[level].n;
  std::vector<float> res((size_t)n * 2);
  float Tf[16];
  for (int i = 0; i < 16; ++i) Tf[i] = (float)T[i];  // transformation.cast<float>(), dense_tracking.cpp:413
  int rc = dvo_amd_residuals(ctx, reference, current, level, Tf, res.data(), nullptr);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) {
    const float r0 = res[(size_t)2 * i];
    image[i] = (r0 == r0) ? std::fabs(r0) : 0.0f;  // :426-438
  }
  return DVO_AMD_OK;
}

int dvo_amd_bench_residual_pass_pairs(dvo_amd_context *ctx, int n_items, dvo_amd_pyramid *const *references,
                                      dvo_amd_pyramid *const *currents, int level, const float *T, int rounds, int reps,
                                      double *avg_ms, double *alg_bytes, int *n_launches) {
  if (!ctx || !references || !currents || !T || level < 0 || n_items < 1 || n_items > 1024 || reps < 1)
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  {
    int rc = queue_must_be_idle(ctx, "dvo_amd_bench_residual_pass");
    if (rc) return rc;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  std::vector<const Selection *> sels((size_t)n_items);
  double px = 0.0;
  for (int i = 0; i < n_items; ++i) {
    if (!references[i] || !currents[i]) return DVO_AMD_ERR_INVALID_ARGUMENT;
    int rc = check_level_pair(ctx, references[i], currents[i], level);
    if (rc) return rc;
    if (references[i]->lv[level].n != references[0]->lv[level].n) return DVO_AMD_ERR_INVALID_ARGUMENT;
    rc = pyramid_selection(references[i], ctx->cfg.intensity_derivative_threshold, ctx->cfg.depth_derivative_threshold, &sels[(size_t)i]);
    if (rc) return rc;
    px += (double)sels[(size_t)i]->count[level];
  }
  const LevelData &R = references[0]->lv[level];
  int rc = ensure_slots(ctx, n_items, R.n_pad);
  if (rc) return rc;
  // `rounds` of the public interface = 256-pixel rounds per wave segment (four steps each); 0 = the driver's choice
  if (rounds != 0 && rounds != 1 && rounds != 2 && rounds != 4 && rounds != 8 && rounds != 16) return DVO_AMD_ERR_INVALID_ARGUMENT;
  int steps = rounds <= 0 ? level_steps(ctx, R) : rounds * 4;
  while (steps < kMaxSteps && blocks_for(R.n, steps) > 2048) steps *= 2;
  TickItem proto;
  std::memset(&proto, 0, sizeof(proto));
  item_set_steps(proto, steps, 1);
  int max_blocks = 1;  // (an item's blocks cover its compacted selection: level_blocks)
  for (int i = 0; i < n_items; ++i) max_blocks = std::max(max_blocks, level_blocks(sels[(size_t)i], level, steps));
  proto.flags = 0;
  proto.P[0] = 1500.0f, proto.P[3] = 7000.0f;  // a typical precision: the weights take the non-trivial branch
  const int launches = (n_items + kMaxItemsPerLaunch - 1) / kMaxItemsPerLaunch;
  const int per = (n_items + launches - 1) / launches;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  double total_ms = 0.0;
  for (int rep = -1; rep < reps; ++rep) {  // rep -1 warms up
    for (int first = 0; first < n_items; first += per) {
      TickArgs ta;
      ta.n_items = std::min(per, n_items - first);
      ta.compact = 0;
      ta.rcp = ctx->rcp;
      for (int i = 0; i < ta.n_items; ++i) {
        TickItem &w = ta.items[i];
        w = proto;
        w.res_blocks = (uint16_t)level_blocks(sels[(size_t)(first + i)], level, steps);
        const LevelData &C = currents[first + i]->lv[level];
        w.ref = sels[(size_t)(first + i)]->ref_desc + level;
        w.cur = currents[first + i]->cur_desc + level;
        w.slot = ctx->slot_desc + (first + i);
        const float K[9] = {C.fx, 0.0f, C.ox, 0.0f, C.fy, C.oy, 0.0f, 0.0f, 1.0f};
        for (int r = 0; r < 3; ++r)
          for (int cc = 0; cc < 4; ++cc)
            w.kt[r * 4 + cc] = (K[r * 3 + 0] * T[cc * 4 + 0] + K[r * 3 + 1] * T[cc * 4 + 1]) + K[r * 3 + 2] * T[cc * 4 + 2];
      }
      hipError_t e = launch_tick(ta, max_blocks, ctx->stream, e0, e1);  // stamped by the dispatch itself
      if (e != hipSuccess) return fail_hip("launch_tick", e);
      HIP_TRY(hipEventSynchronize(e1));
      float ms = 0.0f;
      HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 0) total_ms += ms;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (avg_ms) *avg_ms = total_ms / reps;
  if (alg_bytes) *alg_bytes = 56.0 * px;
  if (n_launches) *n_launches = launches;
  return DVO_AMD_OK;
}

int dvo_amd_bench_residual_pass(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, int level,
                                const float *T, int n_items, int rounds, int reps, double *avg_ms, double *alg_bytes,
                                int *n_launches) {
  if (n_items < 1 || n_items > 1024) return DVO_AMD_ERR_INVALID_ARGUMENT;
  std::vector<dvo_amd_pyramid *> r((size_t)n_items, reference), c((size_t)n_items, current);
  return dvo_amd_bench_residual_pass_pairs(ctx, n_items, r.data(), c.data(), level, T, rounds, reps, avg_ms, alg_bytes, n_launches);
}

int dvo_amd_debug_ll_overflow(dvo_amd_context *ctx, const float *residuals, int n_blocks, int steps, int seg_first, int n_segs,
                              int rank_offset, int rank_end, int cut_rank, const float *precision, int *overflowed) {
  if (!ctx || !residuals || !precision || !overflowed || n_blocks < 1 || steps < 1 || seg_first < 0 || n_segs < 1 ||
      seg_first + n_segs > n_blocks * kWavesPerBlock)
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  int rc = queue_must_be_idle(ctx, "dvo_amd_debug_ll_overflow");
  if (rc) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  const int seg_px = kStepPx * steps, n_px = n_blocks * kWavesPerBlock * seg_px;
  // the prefix table as k_finalize leaves it: valid pixels of the band before each of its wave segments
  std::vector<int> prefix((size_t)n_blocks * kWavesPerBlock, 0);
  int run = 0;
  for (int sgi = seg_first; sgi < seg_first + n_segs; ++sgi) {
    prefix[(size_t)sgi] = run;
    for (int i = 0; i < seg_px; ++i) {
      const float x = residuals[2 * ((size_t)sgi * seg_px + i)];
      run += x == x ? 1 : 0;
    }
  }
  float2 *res_dev = nullptr;
  int *prefix_dev = nullptr;
  HIP_TRY(hipMalloc((void **)&res_dev, sizeof(float2) * (size_t)n_px));
  hipError_t e = hipMalloc((void **)&prefix_dev, sizeof(int) * prefix.size());
  if (e == hipSuccess) e = hipMemcpy(res_dev, residuals, sizeof(float2) * (size_t)n_px, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(prefix_dev, prefix.data(), sizeof(int) * prefix.size(), hipMemcpyHostToDevice);
  bool ovf = false;
  if (e == hipSuccess) {
    OvfBand ob;
    ob.seg_first = seg_first, ob.n_segs = n_segs, ob.rank_offset = rank_offset, ob.rank_end = rank_end;
    rc = ll_overflowed(ctx, res_dev, prefix_dev, n_blocks, steps, cut_rank, precision, &ob, 1, &ovf);
  }
  (void)hipFree(res_dev);
  if (prefix_dev) (void)hipFree(prefix_dev);
  if (e != hipSuccess) return fail_hip("dvo_amd_debug_ll_overflow", e);
  *overflowed = ovf ? 1 : 0;
  return rc;
}

long long dvo_amd_debug_block_trace(dvo_amd_context *ctx, unsigned long long *out, long long capacity_blocks) {
  if (!ctx) return -(long long)DVO_AMD_ERR_INVALID_ARGUMENT;
  if (hipSetDevice(ctx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -(long long)DVO_AMD_ERR_HIP;
  return read_block_trace(out, capacity_blocks);
}

int dvo_amd_debug_finalize_stamps(dvo_amd_context *ctx, unsigned long long *stamps8) {
  if (!ctx || !stamps8) return DVO_AMD_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(read_finalize_stamps(stamps8));
  return DVO_AMD_OK;
}

int dvo_amd_debug_tick_log(dvo_amd_context *ctx, double *out, int capacity_records, int *n_records) {
  if (!ctx || !n_records) return DVO_AMD_ERR_INVALID_ARGUMENT;
  const int n = (int)(ctx->tick_log.size() / kTickLogFields);
  *n_records = n;
  if (out) {
    for (int i = 0; i < std::min(n, capacity_records) * (int)kTickLogFields; ++i) out[i] = ctx->tick_log[(size_t)i];
    ctx->tick_log.clear();
  }
  return DVO_AMD_OK;
}

int dvo_amd_debug_marker(dvo_amd_context *ctx, unsigned tag) {
  if (!ctx) return DVO_AMD_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(ctx->device));
  const hipError_t e = launch_marker(tag, ctx->stream);
  if (e != hipSuccess) return fail_hip("launch_marker", e);
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DVO_AMD_OK;
}

int dvo_amd_kernel_timing(dvo_amd_context *ctx, int enable, double *ms_residual_pass, long long *n_launches, int reset) {
  if (!ctx) return DVO_AMD_ERR_INVALID_ARGUMENT;
  if (ms_residual_pass) *ms_residual_pass = ctx->timing_ms;
  if (n_launches) *n_launches = ctx->timing_launches;
  if (reset) ctx->timing_ms = 0.0, ctx->timing_launches = 0;
  ctx->timing = enable != 0;
  return DVO_AMD_OK;
}

}  // extern "C"
