// One pair tile-sharded into bands of scan-order blocks (BASELINE config 4): the band pipeline (all bands on one GPU, or one band
// per rank), the fold of the band records along the level's summation tree, the one-hop peer exchange and its RCCL fallback, the
// reference's overflowing likelihood across band edges, and the entries of the C ABI that go with them.
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

// ---- one pair tile-sharded into bands of scan-order blocks (BASELINE config 4) ---------------------------------------
// Band b of n covers whole chunks of the level's summation tree (band_blocks, dvo_types.h).  Per-pixel work is independent
// given the pose; what couples the bands is exactly what couples blocks inside one GPU: the ordered fold of (count, pair-quirk
// scale sums under both start parities, boundary weight / residual) and the plain sums of the 87 moments and of the
// likelihood.  So the exchange per tick is one record (FinOut, 784 B) per band, combined on every rank along the same tree.

// the ordered part of a record (count, pair-quirk scale sums under both start parities, boundary weight / residual) and the
// rule that joins two neighbouring runs of pixels: the host's copy of seg_combine in dvo_kernels.hip, operation for operation
struct HostSeg {
  int c;
  float first_w, l0, l1;
  double s0[3], s1[3];
};
HostSeg host_seg_combine(const HostSeg &a, const HostSeg &b) {
  if (b.c == 0) return a;
  if (a.c == 0) return b;
  HostSeg o;
  const bool flip = (a.c & 1) != 0;  // b starts on the opposite parity of everything before it
  const double rxx = (double)a.l0 * a.l0, rxy = (double)a.l0 * a.l1, ryy = (double)a.l1 * a.l1;
  for (int i = 0; i < 3; ++i) {
    o.s0[i] = a.s0[i] + (flip ? b.s1[i] : b.s0[i]);
    o.s1[i] = a.s1[i] + (flip ? b.s0[i] : b.s1[i]);
  }
  // b's first pixel is a pair-second under exactly one hypothesis: there it weights a's last residual
  double *tgt = flip ? o.s0 : o.s1;
  tgt[0] += (double)b.first_w * rxx, tgt[1] += (double)b.first_w * rxy, tgt[2] += (double)b.first_w * ryy;
  o.c = a.c + b.c;
  o.first_w = a.first_w;
  o.l0 = b.l0, o.l1 = b.l1;
  return o;
}

// Band records -> the record of the level.  When the band count divides 16 every band is a subtree of the level's summation
// tree (dvo_types.h, level_chunks_log2) and its reducer has produced that subtree's value: folding the bands with the rest of the
// SAME tree -- a perfect binary tree over the bands -- gives, bit for bit, the record one reducer would have produced from the
// whole level.  Other band counts (3, 5, ...) are folded left to right: deterministic, and equal to the unsharded record up to
// the rounding of the fp64 sums.
void combine_bands(const FinOut *const *recs, int n, FinOut &out) {
  std::memset(&out, 0, sizeof(out));
  HostSeg seg_small[kMaxBands];
  double acc_small[kMaxBands][kNumAcc], ll_small[kMaxBands];
  std::vector<HostSeg> seg_big;
  std::vector<double> acc_big, ll_big;
  HostSeg *seg = seg_small;
  double(*acc)[kNumAcc] = acc_small, *ll = ll_small;
  if (n > kMaxBands) {  // (only the debug entry folds more bands than a node has GPUs)
    seg_big.resize((size_t)n), acc_big.resize((size_t)n * kNumAcc), ll_big.resize((size_t)n);
    seg = seg_big.data(), acc = reinterpret_cast<double(*)[kNumAcc]>(acc_big.data()), ll = ll_big.data();
  }
  for (int b = 0; b < n; ++b) {
    const FinOut &r = *recs[b];
    out.has_res |= r.has_res, out.has_ll |= r.has_ll;
    out.ll_qmax = r.ll_qmax > out.ll_qmax ? r.ll_qmax : out.ll_qmax;
    HostSeg &g = seg[b];
    g.c = r.has_res ? r.valid : 0;
    g.first_w = r.first_w, g.l0 = r.last_r0, g.l1 = r.last_r1;
    for (int i = 0; i < 3; ++i) g.s0[i] = r.S[i], g.s1[i] = r.S_odd[i];
    for (int i = 0; i < kNumAcc; ++i) acc[b][i] = r.acc[i];
    ll[b] = r.ll_sum;
  }
  if (kLevelChunksMax % n == 0) {
    for (int m = n; m > 1; m >>= 1)  // one level of the tree per round
      for (int b = 0; b < m / 2; ++b) {
        seg[b] = host_seg_combine(seg[2 * b], seg[2 * b + 1]);
        for (int i = 0; i < kNumAcc; ++i) acc[b][i] = acc[2 * b][i] + acc[2 * b + 1][i];
        ll[b] = ll[2 * b] + ll[2 * b + 1];
      }
  } else {
    for (int b = 1; b < n; ++b) {
      seg[0] = host_seg_combine(seg[0], seg[b]);
      for (int i = 0; i < kNumAcc; ++i) acc[0][i] += acc[b][i];
      ll[0] += ll[b];
    }
  }
  out.valid = seg[0].c;
  for (int i = 0; i < 3; ++i) out.S[i] = seg[0].s0[i], out.S_odd[i] = seg[0].s1[i];
  out.first_w = seg[0].first_w, out.last_r0 = seg[0].l0, out.last_r1 = seg[0].l1;
  for (int i = 0; i < kNumAcc; ++i) out.acc[i] = acc[0][i];
  out.ll_sum = ll[0];
}

// The records of all bands of the tick (or of the overflow exchange) that was just launched on a tile-sharded pair, in band
// order.  Peer exchange attached: the kernel that carried x_seq pushed this rank's record into every peer's mapped buffer and
// forwards theirs to pinned host memory, which is polled here (no collective, no copy, no stream synchronisation).  Otherwise
// the RCCL all-gather of slot 0's device record + one D2H copy.
int collect_exchange(dvo_amd_context *ctx, int n_bands, const FinOut **recs) {
  if (ctx->x_ranks > 0) {
    const unsigned xseq = ctx->x_seq;  // the kernel of this exchange carried it (set before the launch)
    for (int b = 0; b < n_bands; ++b) {
      unsigned long long spins = 0;
      int have = 0;
      while ((have = take_wire(ctx->x_host + b, ctx->x_store + b, xseq, have)) != kFinWirePieces) {
        __builtin_ia32_pause();
        if (__atomic_load_n(ctx->x_host_seq, __ATOMIC_ACQUIRE) == (xseq | 0x80000000u)) {
          // After a timeout the ranks no longer agree on the tick number (a peer may have taken this rank's record and
          // moved on): the exchange is dead for good.  Later calls fail at once; all ranks must destroy and re-create it.
          ctx->x_broken = true;
          g_last_error = "peer exchange timed out: a rank did not publish its band record (the exchange is now unusable: "
                         "destroy and re-create it on every rank)";
          return DVO_AMD_ERR_COMM;
        }
        if ((++spins & 0xFFFFF) == 0) {
          const hipError_t q = hipStreamQuery(ctx->stream);
          if (q != hipErrorNotReady && q != hipSuccess) return fail_hip("stream died while waiting for the exchange", q);
          if (q == hipSuccess && (have = take_wire(ctx->x_host + b, ctx->x_store + b, xseq, have)) != kFinWirePieces)
            return fail_hip("exchange finished without publishing", hipErrorUnknown);
        }
      }
      recs[b] = ctx->x_store + b;
    }
    return DVO_AMD_OK;
  }
  // per-iteration RCCL all-gather of the band records over xGMI, then one D2H copy of all of them
  if (ctx->p_allgather(ctx->slots[0].out_dev, ctx->gather_dev, sizeof(FinOut), ncclChar, ctx->comm, ctx->stream) != ncclSuccess) {
    g_last_error = "ncclAllGather failed";
    return DVO_AMD_ERR_COMM;
  }
  HIP_TRY(hipMemcpyAsync(ctx->gather_host, ctx->gather_dev, sizeof(FinOut) * (size_t)n_bands, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (int b = 0; b < n_bands; ++b) recs[b] = ctx->gather_host + b;
  return DVO_AMD_OK;
}

// The reference's overflowing 50-term likelihood product (dense_tracking_impl.cpp:413-419) for a pair tile-sharded over several
// GPUs.  A rank holds only its own band's residuals, and a group of fifty can straddle a band edge.  All ranks see the same
// combined record, so all of them come here together when its largest Mahalanobis distance makes an overflow possible:
//   1. every rank judges the groups that lie inside its band (k_ll_overflow on a CLOSED band);
//   2. it extracts its edge terms on the host from small copies of the band's ends: the `head` terms 1 + 0.2 q that complete the
//      group begun in earlier bands, and the running product of the `tail` terms that begin a group the next band completes
//      (multiplied from 1.0 in scan order: the reference's own loop up to that point);
//   3. one more exchange of a record per rank (through whichever exchange the tick records use);
//   4. every rank replays the straddling groups in band order -- acc *= term, fifty at a time, exactly the reference's loop.
// The verdict is the reference's, bit for bit, and the same on every rank.  Rare (never on sensor data) and slow (a few copies
// and a second exchange).
int edge_terms(dvo_amd_context *ctx, const float2 *res, long long px_lo, long long px_hi, bool forward, int want, const float P[4],
               std::vector<double> &terms) {
  terms.clear();
  std::vector<float2> buf;
  const long long chunk = 8192;
  long long at = forward ? px_lo : px_hi;
  while ((int)terms.size() < want && (forward ? at < px_hi : at > px_lo)) {
    const long long lo = forward ? at : std::max(px_lo, at - chunk), hi = forward ? std::min(px_hi, at + chunk) : at;
    buf.resize((size_t)(hi - lo));
    HIP_TRY(hipMemcpyAsync(buf.data(), res + lo, sizeof(float2) * (size_t)(hi - lo), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (long long k = 0; k < hi - lo && (int)terms.size() < want; ++k) {
      const float2 r = buf[(size_t)(forward ? k : hi - lo - 1 - k)];
      if (!(r.x == r.x)) continue;  // NaN marks an invalid pixel
      const float t0 = r.x * P[0] + r.y * P[1];  // k_ll_overflow's expression, operation for operation (no contraction)
      const float t1 = r.x * P[2] + r.y * P[3];
      const float q = t0 * r.x + t1 * r.y;
      terms.push_back(1.0 + 0.2 * (double)q);
    }
    at = forward ? hi : lo;
  }
  if ((int)terms.size() != want) return fail_hip("band edge holds fewer valid residuals than its valid count says", hipErrorUnknown);
  if (!forward) std::reverse(terms.begin(), terms.end());  // back to scan order
  return DVO_AMD_OK;
}

int sharded_overflow(dvo_amd_context *ctx, Job &j, int n_bands, int band, bool *overflowed) {
  *overflowed = false;
  const IterCtx &a = j.a;
  if (a.cut_rank < 50) return DVO_AMD_OK;
  int off[kMaxBands + 1], ll_count[kMaxBands], head_len[kMaxBands], tail_cnt[kMaxBands];
  off[0] = 0;
  for (int b = 0; b < n_bands; ++b) {
    off[b + 1] = off[b] + a.band_valid[b];
    ll_count[b] = std::max(0, std::min(a.cut_rank - off[b], a.band_valid[b]));  // this band's residuals that enter the likelihood
    head_len[b] = std::min((50 - off[b] % 50) % 50, ll_count[b]);
    // the trailing partial group of a band is completed by the next band only when the band's whole tail enters the likelihood;
    // when the cut falls inside the band (ll_count < band_valid) nothing behind it counts (cut_rank is a multiple of 50: the
    // residuals up to it end on a group boundary) -- and edge_terms(forward = false) would read the band's LAST residuals, which
    // are not the ones ranked below the cut
    tail_cnt[b] = ll_count[b] < a.band_valid[b] ? 0 : (ll_count[b] - head_len[b]) % 50;
  }
  // 1. the groups inside this rank's band
  int first = 0, count = 0;
  band_blocks(a.n_blocks, n_bands, band, &first, &count);
  bool inside = false;
  if (count > 0 && ll_count[band] - head_len[band] >= 50) {
    OvfBand ob;
    ob.seg_first = first * kWavesPerBlock, ob.n_segs = count * kWavesPerBlock, ob.rank_offset = off[band], ob.rank_end = off[band + 1];
    int rc = ll_overflowed(ctx, ctx->slots[0].res[a.buf], ctx->slots[0].seg_prefix[a.buf], a.n_blocks, a.steps, a.cut_rank, a.P, &ob, 1,
                           &inside);
    if (rc) return rc;
  }
  // 2. this rank's edge terms
  FinOut rec;
  std::memset(&rec, 0, sizeof(rec));
  const long long block_px = (long long)kStepPx * kWavesPerBlock * a.steps;
  const long long px_lo = first * block_px, px_hi = (long long)(first + count) * block_px;
  std::vector<double> head, tail;
  if (head_len[band] > 0) {
    int rc = edge_terms(ctx, ctx->slots[0].res[a.buf], px_lo, px_hi, true, head_len[band], a.P, head);
    if (rc) return rc;
  }
  double tail_acc = 1.0;
  if (tail_cnt[band] > 0) {
    int rc = edge_terms(ctx, ctx->slots[0].res[a.buf], px_lo, px_hi, false, tail_cnt[band], a.P, tail);
    if (rc) return rc;
    for (double t : tail) tail_acc *= t;
  }
  rec.acc[0] = inside ? 1.0 : 0.0, rec.acc[1] = (double)head_len[band], rec.acc[2] = (double)tail_cnt[band], rec.acc[3] = tail_acc;
  for (size_t i = 0; i < head.size(); ++i) rec.acc[4 + i] = head[i];
  static_assert(kNumAcc >= 4 + 49, "the edge record rides in the moment slots of a FinOut");
  // 3. exchange
  HIP_TRY(hipMemcpyAsync(ctx->slots[0].out_dev, &rec, sizeof(rec), hipMemcpyHostToDevice, ctx->stream));
  if (ctx->x_ranks > 0) {
    ctx->x_seq = next_seq(ctx->x_seq);
    hipError_t e = launch_exchange_record(ctx->slots[0].out_dev, ctx->x_args_dev, ctx->x_seq, ctx->stream);
    if (e != hipSuccess) return fail_hip("launch_exchange_record", e);
  }
  const FinOut *recs[kMaxBands];
  int rc = collect_exchange(ctx, n_bands, recs);
  if (rc) return rc;
  // 4. the straddling groups, in band order
  bool any = false;
  double acc = 1.0;
  int cnt = 0;
  for (int b = 0; b < n_bands; ++b) {
    const FinOut &r = *recs[b];
    if ((int)r.acc[1] != head_len[b] || (int)r.acc[2] != tail_cnt[b]) {
      g_last_error = "ranks disagree on the band edges of the likelihood's groups of fifty";
      return DVO_AMD_ERR_COMM;
    }
    any = any || r.acc[0] != 0.0;
    for (int i = 0; i < head_len[b]; ++i) {
      acc *= r.acc[4 + i];
      if (++cnt == 50) {
        any = any || !(acc <= 1.7976931348623157e308);
        acc = 1.0, cnt = 0;
      }
    }
    if (tail_cnt[b] > 0) acc = r.acc[3], cnt = tail_cnt[b];  // (cnt is 0 here: the head closed the group before it)
  }
  *overflowed = any;
  return DVO_AMD_OK;
}

int run_tick_banded(dvo_amd_context *ctx, Job &j, int n_bands, int band_first, int n_local, bool exchange) {
  const unsigned seq = ctx->tick_seq = next_seq(ctx->tick_seq);
  if (!j.have_a && !j.have_b) return DVO_AMD_OK;
  // the level's own segment length (level_steps): the same for every band count and rank, and the unsharded driver's
  const int steps_level = level_steps(ctx, j.ref->lv[j.level]);
  const int nb_level = level_blocks(j.sel, j.level, steps_level);
  TickArgs ta;
  FinArgs fa;
  std::memset(&ta, 0, sizeof(ta));
  std::memset(&fa, 0, sizeof(fa));
  ta.rcp = ctx->rcp;
  j.sub_ll = j.have_a, j.sub_res = j.have_b;
  if (j.have_b) {
    j.b.steps = steps_level, j.b.n_blocks = nb_level;
    j.result->n_residual_passes++;
    j.alg_px += (double)j.sel->count[j.level];
  }
  j.result->n_ticks++;
  int max_blocks = 0;
  for (int li = 0; li < n_local; ++li) {
    const int band = band_first + li;
    TickItem &w = ta.items[li];
    w.ref = j.sel->ref_desc + j.level;
    w.cur = j.cur->cur_desc + j.level;
    w.slot = ctx->slot_desc;  // every band works in slot 0's buffers (logical block indexing), disjoint ranges
    FinItem &f = fa.items[li];
    f.ll_partials = ctx->slots[0].ll_partials;
    f.ll_qmax_off = ctx->slots[0].ll_qmax_off;
    f.seg_prefix_out = ctx->slots[0].seg_prefix[0];
    f.out = ctx->slots[(size_t)li].out;
    f.out_dev = exchange ? ctx->slots[(size_t)li].out_dev : nullptr;  // device copy: source of the all-gather / peer exchange
    f.seq = seq;
    if (j.have_a) {
      // the merged likelihood blocks of the band's chunks (a band is a run of whole chunks: none straddles its edge) -- the
      // very blocks the unsharded pass runs
      item_set_ll_merge(w, level_ll_merge(ctx, j.a.steps));
      const int C = 1 << level_chunks_log2(j.a.n_blocks);
      const int first = ll_blocks_before(j.a.n_blocks, item_ll_merge_log2(w), C * band / n_bands);
      const int count = ll_blocks_before(j.a.n_blocks, item_ll_merge_log2(w), C * (band + 1) / n_bands) - first;
      w.ll_first = (uint16_t)first, w.ll_blocks = (uint16_t)count, w.ll_level_blocks = (uint16_t)j.a.n_blocks;
      if (j.a.buf) w.flags |= kItemLlBuf;
      int before = 0;
      for (int b = 0; b < band; ++b) before += j.a.band_valid[b];
      w.ll_cut_rank = j.a.cut_rank - before;  // rank inside the band below which residuals enter the likelihood
      f.n_ll_blocks = w.ll_blocks, f.ll_first = w.ll_first, f.ll_level_blocks = w.ll_level_blocks;
      f.ll_merge_log2 = (uint16_t)item_ll_merge_log2(w);
    }
    if (j.have_b) {
      int first = 0, count = 0;
      band_blocks(nb_level, n_bands, band, &first, &count);
      w.res_first = (uint16_t)first, w.res_blocks = (uint16_t)count;
      if (j.b.buf) w.flags |= kItemResBuf;
      if (j.b.k == 0) w.flags |= kItemUnitWeights;
      make_kt(j.cur->lv[j.level], j.b.estimate_after, w.kt);
      f.records = ctx->slots[0].records;
      f.n_blocks = w.res_blocks, f.block_first = w.res_first, f.level_blocks = (uint16_t)nb_level;
      f.seg_prefix_out = ctx->slots[0].seg_prefix[j.b.buf];
    }
    std::memcpy(w.P, j.have_a ? j.a.P : j.precision, sizeof(w.P));
    item_set_steps(w, steps_level, j.have_a ? j.a.steps : steps_level);
    max_blocks = std::max(max_blocks, w.res_blocks + w.ll_blocks);
  }
  ta.n_items = n_local, fa.n_items = n_local;
  if (exchange && ctx->x_ranks > 0) fa.exchange = ctx->x_args_dev, fa.xseq = ctx->x_seq = next_seq(ctx->x_seq);  // the tail of k_finalize exchanges
  hipError_t e = launch_tick(ta, std::max(max_blocks, 1), ctx->stream);
  if (e != hipSuccess) return fail_hip("launch_tick", e);
  // Host-rcpps mode, Q7 (k_q7_tail): which pixels are a pass's last V mod 4 is a property of the whole level.  With every band on
  // this GPU the level's block records all stand in slot 0: one whole-level item, and the host adds what it leaves to the combined
  // record -- the addition k_finalize makes at the root of the same tree for an unsharded pair, so banded == unsharded holds in
  // this mode too.  A pair sharded over several GPUs has only its own band's records: its tail weights stay the table's (stated in
  // include/dvo_amd.h).
  const bool q7 = ctx->rcp.table && j.have_b && j.b.k != 0 && !exchange && band_first == 0 && n_local == n_bands;
  if (q7) {
    Q7ArgsSmall qs;
    qs.n_items = 1, qs.q7_off256 = ctx->q7_off256, qs.rcp = ctx->rcp;
    TickItem whole = ta.items[0];
    whole.res_first = 0, whole.res_blocks = (uint16_t)nb_level;
    for (int i = 0; i < kMaxSmallItems; ++i) qs.items[i] = whole;
    e = launch_q7_tail_small(qs, ctx->stream);
    if (e != hipSuccess) return fail_hip("launch_q7_tail", e);
  }
  e = launch_finalize(fa, ctx->stream);
  if (e != hipSuccess) return fail_hip("launch_finalize", e);

  const FinOut *recs[kMaxBands];
  if (exchange) {
    int rc = collect_exchange(ctx, n_bands, recs);
    if (rc) return rc;
  } else {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < n_bands; ++b) {
      int rc = take_record_synced(ctx, (size_t)b, seq);
      if (rc) return rc;
      recs[b] = ctx->out_host + b;
    }
  }
  FinOut comb;
  combine_bands(recs, n_bands, comb);
  if (q7) {
    Q7Rec q;
    HIP_TRY(hipMemcpy(&q, ctx->slots[0].q7, sizeof(q), hipMemcpyDeviceToHost));
    for (int i = 0; i < 3; ++i) comb.S[i] = comb.S[i] + q.S[i];
    for (int i = 0; i < kNumAcc; ++i) comb.acc[i] = comb.acc[i] + q.acc[i];
  }
  if (j.sub_res)
    for (int b = 0; b < n_bands; ++b) j.b.band_valid[b] = recs[b]->valid;
  if (j.sub_ll) {
    bool overflowed = false;
    if (comb.ll_qmax >= kLlOverflowScreen && exchange) {
      // a pair sharded over several GPUs holds only its own band here: the ranks settle the groups of fifty that straddle band
      // edges together (every rank sees the same combined record, so all of them take this branch in the same tick)
      int rc = sharded_overflow(ctx, j, n_bands, band_first, &overflowed);
      if (rc) return rc;
    } else if (comb.ll_qmax >= kLlOverflowScreen) {
      // (all bands of the level were computed on this GPU, in slot 0's buffers: the exact check sees the whole level)
      OvfBand ob[kMaxBands];
      int before = 0;
      for (int b = 0; b < n_bands; ++b) {  // the prefix table is relative to each band (band_blocks of the pass's blocks)
        int first = 0, count = 0;
        band_blocks(j.a.n_blocks, n_bands, b, &first, &count);
        ob[b].seg_first = first * kWavesPerBlock, ob[b].n_segs = count * kWavesPerBlock, ob[b].rank_offset = before;
        before += j.a.band_valid[b];
      }
      int rc = ll_overflowed(ctx, ctx->slots[0].res[j.a.buf], ctx->slots[0].seg_prefix[j.a.buf], j.a.n_blocks, j.a.steps, j.a.cut_rank,
                             j.a.P, ob, n_bands, &overflowed);
      if (rc) return rc;
    }
    process_loglik(j, &comb, overflowed);
  } else {
    IterCtx bcopy = j.b;
    process_residual(j, bcopy, comb);
  }
  return DVO_AMD_OK;
}

int match_one_banded(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, const double *T_init,
                     dvo_amd_result *result, int n_bands, int band_first, int n_local, bool exchange) {
  if (!ctx || !reference || !current || !result || n_bands < 1 || n_bands > kMaxBands || n_local < 1 ||
      band_first < 0 || band_first + n_local > n_bands || n_local > kMaxItemsPerLaunch)
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  const dvo_amd_config &cfg = ctx->cfg;
  int rc = check_config(&cfg);
  if (rc) return rc;
  rc = queue_must_be_idle(ctx, "dvo_amd_match_banded / _sharded");
  if (rc) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  if (reference->device != ctx->device || current->device != ctx->device) return DVO_AMD_ERR_DEVICE_MISMATCH;
  if (reference->n_levels < cfg.first_level + 1 || current->n_levels < cfg.first_level + 1) return DVO_AMD_ERR_TOO_FEW_LEVELS;
  for (int l = cfg.last_level; l <= cfg.first_level; ++l)
    if (reference->lv[l].w != current->lv[l].w || reference->lv[l].h != current->lv[l].h) return DVO_AMD_ERR_INVALID_ARGUMENT;
  if (cfg.use_initial_estimate) {
    if (!T_init) return DVO_AMD_ERR_INVALID_ARGUMENT;
    double s = 0.0;
    for (int k = 0; k < 16; ++k) s += T_init[k];
    if (!std::isfinite(s)) return DVO_AMD_ERR_NAN_INIT;
  }
  const int its_needed = (cfg.first_level - cfg.last_level + 1) * (cfg.max_iterations_per_level + 1);
  if (result->iterations && result->iterations_capacity > 0 && result->iterations_capacity < its_needed) return DVO_AMD_ERR_CAPACITY;
  rc = ensure_slots(ctx, std::max(n_local, n_bands), reference->lv[cfg.last_level].n_pad);
  if (rc) return rc;
  Job j;
  j.ref = reference, j.cur = current, j.result = result, j.slot = &ctx->slots[0], j.cfg = &ctx->cfg;
  rc = pyramid_selection(j.ref, cfg.intensity_derivative_threshold, cfg.depth_derivative_threshold, &j.sel);
  if (rc) return rc;
  result->n_levels = 0, result->n_iterations = 0, result->n_ticks = 0, result->n_residual_passes = 0;
  result->alg_bytes = 0.0, result->alg_bytes_discarded = 0.0, result->is_nan = 0;
  if (!result->iterations) result->iterations_capacity = 0;
  j.inc = cfg.use_initial_estimate ? se3_from_matrix(T_init) : SE3::identity();
  j.initial = j.inc;
  j.estimate = SE3::identity();
  j.level = cfg.first_level;
  j.done = false;
  for (int b = 0; b < kMaxBands; ++b) j.a.band_valid[b] = j.b.band_valid[b] = 0;
  start_level(j);
  while (!j.done) {
    rc = run_tick_banded(ctx, j, n_bands, band_first, n_local, exchange);
    if (rc) return rc;
  }
  return DVO_AMD_OK;
}


}  // namespace host
}  // namespace dvo_amd

using namespace dvo_amd;
using namespace dvo_amd::host;

extern "C" {

int dvo_amd_match_banded(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, const double *T_init,
                         dvo_amd_result *result, int n_bands) {
  return match_one_banded(ctx, reference, current, T_init, result, n_bands, 0, n_bands, false);
}

int dvo_amd_comm_unique_id(unsigned char *id128) {
  if (!id128) return DVO_AMD_ERR_INVALID_ARGUMENT;
  void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) {
    g_last_error = std::string("dlopen librccl: ") + dlerror();
    return DVO_AMD_ERR_COMM;
  }
  auto get_id = (ncclResult_t(*)(ncclUniqueId *))dlsym(lib, "ncclGetUniqueId");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  if (!get_id || get_id(&id) != ncclSuccess) {
    g_last_error = "ncclGetUniqueId failed";
    return DVO_AMD_ERR_COMM;
  }
  std::memcpy(id128, &id, 128);
  return DVO_AMD_OK;
}

int dvo_amd_comm_create(dvo_amd_context *ctx, const unsigned char *id128, int nranks, int rank) {
  if (!ctx || !id128 || nranks < 1 || nranks > kMaxBands || rank < 0 || rank >= nranks) return DVO_AMD_ERR_INVALID_ARGUMENT;
  if (ctx->comm) return DVO_AMD_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(ctx->device));
  ctx->rccl_lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!ctx->rccl_lib) ctx->rccl_lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!ctx->rccl_lib) {
    g_last_error = std::string("dlopen librccl: ") + dlerror();
    return DVO_AMD_ERR_COMM;
  }
  auto init_rank = (ncclResult_t(*)(ncclComm_t *, int, ncclUniqueId, int))dlsym(ctx->rccl_lib, "ncclCommInitRank");
  ctx->p_allgather = (decltype(ctx->p_allgather))dlsym(ctx->rccl_lib, "ncclAllGather");
  ctx->p_comm_destroy = (decltype(ctx->p_comm_destroy))dlsym(ctx->rccl_lib, "ncclCommDestroy");
  if (!init_rank || !ctx->p_allgather || !ctx->p_comm_destroy) {
    g_last_error = "librccl lacks ncclCommInitRank / ncclAllGather / ncclCommDestroy";
    return DVO_AMD_ERR_COMM;
  }
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  if (init_rank(&ctx->comm, nranks, id, rank) != ncclSuccess) {
    ctx->comm = nullptr;
    g_last_error = "ncclCommInitRank failed";
    return DVO_AMD_ERR_COMM;
  }
  ctx->comm_ranks = nranks, ctx->comm_rank = rank;
  hipError_t e = hipMalloc((void **)&ctx->gather_dev, sizeof(FinOut) * kMaxBands);
  if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->gather_host, sizeof(FinOut) * kMaxBands, hipHostMallocDefault);
  if (e != hipSuccess) {  // never leave a communicator behind whose exchange buffers do not exist
    dvo_amd_comm_destroy(ctx);
    return fail_hip("communicator buffers", e);
  }
  return DVO_AMD_OK;
}

void dvo_amd_comm_destroy(dvo_amd_context *ctx) {
  if (!ctx || !ctx->comm) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->p_comm_destroy(ctx->comm);
  ctx->comm = nullptr;
  if (ctx->gather_dev) (void)hipFree(ctx->gather_dev);
  if (ctx->gather_host) (void)hipHostFree(ctx->gather_host);
  ctx->gather_dev = nullptr, ctx->gather_host = nullptr;
}

int dvo_amd_exchange_create(dvo_amd_context *ctx, int nranks, int rank, unsigned char *handle64) {
  if (!ctx || !handle64 || nranks < 1 || nranks > kMaxExchangeRanks || rank < 0 || rank >= nranks) return DVO_AMD_ERR_INVALID_ARGUMENT;
  if (ctx->xbuf) return DVO_AMD_ERR_INVALID_ARGUMENT;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t bytes = sizeof(FinWire) * 2 * (size_t)nranks;
  // fine-grained device memory: writes of other agents become visible to a running kernel (coarse-grained memory is only
  // coherent at kernel boundaries)
  // (no fallback to hipMalloc: a running k_finalize would never see a peer's record there and every tick would end in the
  // timeout -- the caller gets DVO_AMD_ERR_COMM here and uses the RCCL path, dvo_amd_comm_create, instead)
  hipError_t e = hipExtMallocWithFlags((void **)&ctx->xbuf, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    ctx->xbuf = nullptr;
    (void)hipGetLastError();
    g_last_error = std::string("fine-grained device memory for the peer exchange is not available (") + hipGetErrorString(e) +
                   "): use the RCCL exchange (dvo_amd_comm_create)";
    return DVO_AMD_ERR_COMM;
  }
  e = hipMemset(ctx->xbuf, 0, bytes);
  if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->x_host, sizeof(FinWire) * kMaxExchangeRanks, hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) std::memset(ctx->x_host, 0, sizeof(FinWire) * kMaxExchangeRanks);
  if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->x_host_seq, 64, hipHostMallocMapped | hipHostMallocCoherent);
  hipIpcMemHandle_t h;
  std::memset(&h, 0, sizeof(h));
  if (e == hipSuccess && nranks > 1) e = hipIpcGetMemHandle(&h, ctx->xbuf);
  if (e != hipSuccess) {
    dvo_amd_exchange_destroy(ctx);
    return fail_hip("exchange buffer", e);
  }
  *ctx->x_host_seq = 0;
  std::memcpy(handle64, &h, 64);
  ctx->x_rank = rank;
  ctx->x_ranks = -nranks;  // created, not attached yet
  return DVO_AMD_OK;
}

int dvo_amd_exchange_attach(dvo_amd_context *ctx, const unsigned char *handles) {
  if (!ctx || !ctx->xbuf || ctx->x_ranks >= 0) return DVO_AMD_ERR_INVALID_ARGUMENT;
  const int n = -ctx->x_ranks;
  if (n > 1 && !handles) return DVO_AMD_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(ctx->device));
  for (int r = 0; r < n; ++r) {
    if (r == ctx->x_rank) {
      ctx->xpeers[r] = ctx->xbuf;
      continue;
    }
    hipIpcMemHandle_t h;
    std::memcpy(&h, handles + 64 * (size_t)r, 64);
    void *p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      dvo_amd_exchange_destroy(ctx);
      return fail_hip("hipIpcOpenMemHandle", e);
    }
    ctx->xpeers[r] = (FinWire *)p;
    ctx->xpeer_opened[r] = true;
  }
  ExchangeArgs xa;
  std::memset(&xa, 0, sizeof(xa));
  for (int r = 0; r < n; ++r) xa.peers[r] = ctx->xpeers[r];
  xa.local = ctx->xbuf;
  hipError_t e = hipHostGetDevicePointer((void **)&xa.host_records, ctx->x_host, 0);
  if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&xa.host_seq, ctx->x_host_seq, 0);
  xa.n_ranks = n, xa.rank = ctx->x_rank;
  xa.timeout_ticks = 500000000u;  // 5 s
  if (e == hipSuccess) e = hipMalloc((void **)&ctx->x_args_dev, sizeof(xa));
  if (e == hipSuccess) e = hipMemcpy(ctx->x_args_dev, &xa, sizeof(xa), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    dvo_amd_exchange_destroy(ctx);
    return fail_hip("exchange description", e);
  }
  ctx->x_ranks = n;
  ctx->comm_ranks = n, ctx->comm_rank = ctx->x_rank;
  return DVO_AMD_OK;
}

void dvo_amd_exchange_destroy(dvo_amd_context *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (int r = 0; r < kMaxExchangeRanks; ++r) {
    if (ctx->xpeer_opened[r] && ctx->xpeers[r]) (void)hipIpcCloseMemHandle(ctx->xpeers[r]);
    ctx->xpeers[r] = nullptr, ctx->xpeer_opened[r] = false;
  }
  if (ctx->xbuf) (void)hipFree(ctx->xbuf);
  if (ctx->x_args_dev) (void)hipFree(ctx->x_args_dev);
  ctx->x_args_dev = nullptr;
  if (ctx->x_host) (void)hipHostFree(ctx->x_host);
  if (ctx->x_host_seq) (void)hipHostFree(ctx->x_host_seq);
  ctx->xbuf = nullptr, ctx->x_host = nullptr, ctx->x_host_seq = nullptr;
  ctx->x_ranks = 0, ctx->x_seq = 0, ctx->x_broken = false;
}

int dvo_amd_match_sharded(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, const double *T_init,
                          dvo_amd_result *result) {
  if (!ctx || (!ctx->comm && ctx->x_ranks <= 0)) return DVO_AMD_ERR_COMM;
  if (ctx->x_ranks > 0 && ctx->x_broken) {
    g_last_error = "the peer exchange timed out earlier: destroy and re-create it on every rank";
    return DVO_AMD_ERR_COMM;
  }
  return match_one_banded(ctx, reference, current, T_init, result, ctx->comm_ranks, ctx->comm_rank, 1, true);
}

int dvo_amd_debug_combine_bands(int n_bands, const double *bands, double *out) {
  // bands: n x {valid, first_w, last_r0, last_r1, S[3], S_odd[3]} = 10 doubles each; out: {valid, S[3], S_odd[3]}
  if (n_bands < 1 || n_bands > 4096 || !bands || !out) return DVO_AMD_ERR_INVALID_ARGUMENT;
  std::vector<FinOut> recs((size_t)n_bands);
  std::vector<const FinOut *> ptrs((size_t)n_bands);
  for (int b = 0; b < n_bands; ++b) {
    FinOut &r = recs[(size_t)b];
    std::memset(&r, 0, sizeof(r));
    const double *s = bands + 10 * (size_t)b;
    r.has_res = 1, r.valid = (int)s[0], r.first_w = (float)s[1], r.last_r0 = (float)s[2], r.last_r1 = (float)s[3];
    for (int i = 0; i < 3; ++i) r.S[i] = s[4 + i], r.S_odd[i] = s[7 + i];
    ptrs[(size_t)b] = &r;
  }
  FinOut comb;
  combine_bands(ptrs.data(), n_bands, comb);
  out[0] = comb.valid;
  for (int i = 0; i < 3; ++i) out[1 + i] = comb.S[i], out[4 + i] = comb.S_odd[i];
  return DVO_AMD_OK;
}

}  // extern "C"
