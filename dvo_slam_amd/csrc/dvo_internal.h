// Internal declarations shared by the host-side translation units of libdvo_amd.so (NOT part of any interface):
//   dvo_pyramid.cpp   device pools, pyramid construction, point selection, the dvo_amd_pyramid_* entries
//   dvo_tracker.cpp   contexts, the Gauss-Newton driver (ticks), the queue behind a context, the match entries
//   dvo_sharded.cpp   one pair tile-sharded into bands / over several GPUs: band pipeline, record exchange, RCCL fallback
//   dvo_probes.cpp    entries that work in slot 0 outside the match driver: residuals / error image, the stage probe, the kernel
//                     micro-benchmarks and the diagnostics of include/dvo_amd_debug.h
// Until round 4 all of it was one 3 000-line translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dvo_amd.h"
#include "../../include/dvo_amd_debug.h"
#include "dvo_types.h"
#include "se3.h"

namespace dvo_amd {
namespace host {

extern thread_local std::string g_last_error;  // text behind dvo_amd_last_error() on the calling thread
int fail_hip(const char *what, hipError_t e);

#define HIP_TRY(expr)                                                     \
  do {                                                                    \
    hipError_t e_ = (expr);                                               \
    if (e_ != hipSuccess) return ::dvo_amd::host::fail_hip(#expr, e_);    \
  } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }


constexpr int kSpecLevelsMaxPairs = 8;  // speculative level starts (context::spec_levels) up to this many resident pairs
constexpr size_t kTickLogFields = 8;  // doubles per logged launch (dvo_amd_debug_tick_log)
constexpr int kMaxTickStreams = 8;  // streams a context spreads the launches / pair groups of its ticks over

constexpr int kMaxDevices = 16;
constexpr int kMaxBands = 16;

}  // namespace host
}  // namespace dvo_amd

using namespace dvo_amd;  // (an internal header of four translation units that all say the same)

// ------------------------------------------------------------------------------------------------------------------
// pyramid
// ------------------------------------------------------------------------------------------------------------------

// the compacted selection of one level (k_compact): seven arrays of n_pad entries each, point p = the p-th selected pixel
struct CompactLevel {
  float *z, *i, *ix, *iy, *tx, *ty;
  int *pix;  // pixel index of the point (-1 in the padding): how per-point results find their way back to the image
};
constexpr size_t kCompactBytesPerPoint = 7 * sizeof(float);
struct Selection {
  float ti, td;
  float *zsel[DVO_AMD_MAX_LEVELS];        // pixel space: depth where selected, NaN elsewhere (the mask; the source of the compaction)
  CompactLevel pts[DVO_AMD_MAX_LEVELS];   // what the residual pass reads
  int count[DVO_AMD_MAX_LEVELS];  // PointSelection size (includes an odd trailing point)
  int n_pts[DVO_AMD_MAX_LEVELS];  // points the passes walk: count without an odd trailing point (Q3)
  int last[DVO_AMD_MAX_LEVELS];   // index of the last selected pixel
  RefLevelDesc *ref_desc;         // device, [levels]
  void *desc_entry = nullptr;     // arena entry holding ref_desc (null: shares the pyramid's entry)
  void *extra_slab;               // owned allocation (null for the selection carved from the pyramid slab)
  size_t extra_bytes;
};

struct LevelData {
  int w, h, n, n_pad;
  float fx, fy, ox, oy;
  float *i_plane, *z_plane;
  float4 *c_a;
  float2 *c_b;
  float *r_i, *r_ix, *r_iy;
  float *tx, *ty;
  float *zsel0;  // room for the first selection
  char *pts0;    // ... and for its compacted arrays (kCompactBytesPerPoint * n_pad)
};

struct dvo_amd_pyramid {
  std::atomic<int> refs{1};
  int device = 0;
  int n_levels = 0;
  double timestamp = 0.0;
  LevelData lv[DVO_AMD_MAX_LEVELS];
  void *slab = nullptr;
  size_t slab_bytes = 0;
  int *counters = nullptr;  // device, [levels][2], inside the slab
  int2 *sel_partials = nullptr;  // device scratch of the selection kernels (level 0's block count), inside the slab
  int *sel_prefix = nullptr;     // ... and the exclusive prefix of the partials' counts (k_select_prefix)
  void *desc_entry = nullptr;         // this pyramid's entry of the device's descriptor arena
  CurLevelDesc *cur_desc = nullptr;   // device, [levels], in desc_entry
  RefLevelDesc *ref_desc0 = nullptr;  // device, [levels], in desc_entry: room for the first selection's descriptors
  std::mutex mu;
  // entries are never moved or removed while the pyramid lives: a pointer handed out by pyramid_selection() stays valid and may
  // be read without the lock (only the vector itself needs `mu`)
  std::vector<std::unique_ptr<Selection>> selections;
};

// ------------------------------------------------------------------------------------------------------------------
// context + Gauss-Newton driver
// ------------------------------------------------------------------------------------------------------------------

struct Runner;  // the resident pairs of a context and the queue behind them (defined with the driver below)

struct JobSlot {
  float2 *res[2] = {nullptr, nullptr};
  float *records = nullptr;
  double *ll_partials = nullptr;
  float *ll_qmax = nullptr;   // per likelihood block: the largest Mahalanobis distance it took (behind ll_partials)
  unsigned ll_qmax_off = 0;   // ... its distance from ll_partials in doubles
  Q7Rec *q7 = nullptr;        // (host-rcpps mode) what k_q7_tail leaves for k_finalize, behind ll_qmax
  int *seg_prefix[2] = {nullptr, nullptr};
  FinWire *out = nullptr;     // pinned host memory as the device sees it: the record arrives here as tagged pieces
  FinOut *out_dev = nullptr;  // device staging of the record
  void *dev_block = nullptr;
};

struct dvo_amd_context {
  int device = 0;
  dvo_amd_config cfg;
  hipStream_t stream = nullptr;             // stream 0
  std::vector<hipStream_t> extra_streams;   // further streams for the launches of one tick (batches > one launch)
  hipEvent_t desc_ready = nullptr;
  std::vector<JobSlot> slots;
  int slot_n_pad = 0;  // capacity every slot was sized for
  FinWire *out_wire = nullptr;           // pinned, device-visible: one per slot, written by k_finalize
  std::vector<FinOut> out_store;         // the records decoded from out_wire (plain host memory)
  FinOut *out_host = nullptr;            // = out_store.data()
  int out_capacity = 0;
  SlotDesc *slot_desc = nullptr;       // device, [slot]
  Runner *runner = nullptr;            // resident pairs + pending queue (dvo_amd_match_submit / _wait, dvo_amd_match_many)
  int items_per_launch = kMaxItemsPerLaunch;           // DVO_AMD_ITEMS_PER_LAUNCH (<= kMaxItemsPerLaunch: tuning)
  int ll_merge = 4;                                    // residual wave segments per likelihood wave segment (DVO_AMD_LL_MERGE=1|2|4|8)
  int spec_levels = -1;                                // start the next level speculatively in the tick of a level's last
                                                       // likelihood: -2..3 ticks per pair, but a converged level's last likelihood is
                                                       // rejected about half the time (+3 % residual work).  -1 (default): only
                                                       // while at most kSpecLevelsMaxPairs pairs are resident in the tick (latency
                                                       // matters, the GPU has room); DVO_AMD_SPEC_LEVELS=0 never, =1 always
  // DVO_AMD_HOST_PROF=1: where the host thread spends its time (printed when the context is destroyed)
  bool host_prof = false;
  double prof_submit_ns = 0.0, prof_wait_ns = 0.0, prof_process_ns = 0.0;
  long long prof_ticks = 0, prof_job_ticks = 0;
  // tile-shard exchange (RCCL, loaded with dlopen so that single-GPU users do not depend on it)
  void *rccl_lib = nullptr;
  ncclComm_t comm = nullptr;
  int comm_ranks = 0, comm_rank = 0;
  FinOut *gather_dev = nullptr, *gather_host = nullptr;
  // one-hop peer exchange (replaces the all-gather + D2H copy + stream sync of a tick when attached)
  FinWire *xbuf = nullptr;                      // own exchange buffer: 2 generations x n ranks, fine-grained device memory
  FinWire *xpeers[kMaxExchangeRanks] = {};      // every rank's buffer as mapped into this process (own one included)
  bool xpeer_opened[kMaxExchangeRanks] = {};    // mapped with hipIpcOpenMemHandle (to be closed)
  int x_ranks = 0, x_rank = 0;
  FinWire *x_host = nullptr;                    // pinned: the records of a tick in rank order, as tagged pieces
  FinOut x_store[kMaxExchangeRanks];            // ... decoded
  unsigned *x_host_seq = nullptr;               // pinned: tick | 0x80000000 when the exchange kernel gave up waiting for a peer
  unsigned x_seq = 0;
  bool x_broken = false;                        // a tick of the exchange timed out: every later dvo_amd_match_sharded fails fast
  ExchangeArgs *x_args_dev = nullptr;           // device copy of the exchange description k_finalize reads
  ncclResult_t (*p_allgather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*p_comm_destroy)(ncclComm_t) = nullptr;
  unsigned tick_seq = 0;
  RcpTable rcp = {nullptr, 0, 0, nullptr};  // opt-in: the host's _mm_rcp_ps from a table (dvo_amd_set_reciprocal_mode); null = exact
  unsigned *rcp_table_dev = nullptr;    // the device copy of the table (kept once built)
  unsigned *rcp_nibbles_dev = nullptr;  // ... and of the packed corrections of the nibble form (null: they do not fit four bits)
  int rcp_unit = 0;
  std::string rcp_form_note;            // why the nibble form is not in use (diagnostic)
  int q7_off256 = 0;                    // a slot's Q7Rec lives 256 * q7_off256 bytes behind its ll_partials (ensure_slots)
  float *dbg_w_dev = nullptr;           // dvo_amd_debug_weights: per-pixel weights of slot 0's residual pass
  size_t dbg_w_capacity = 0;
  unsigned *ovf_host = nullptr, *ovf_dev = nullptr;  // pinned word for the verdict of k_ll_overflow (rare path)
  long long ovf_checks = 0, ovf_hits = 0;            // how often the exact overflow check ran / said yes (diagnostic)
  // Wave-step counts OF A LEVEL (its pixels / 64) from which its wave segments take 2 / 4 / 8 / 16 steps: the geometry of a
  // residual pass -- and with it the order every fp32 sum of the pass is taken in -- is a function of the level alone, never
  // of what else is resident in the tick (level_steps below).  DVO_AMD_LEVEL_STEPS_AT="a,b,c,d", read when the context is
  // created (a tuning knob: it changes results in the last bits like any other summation order would; a fifth value e: 32
  // steps from e).
  // Default (DVO_AMD_GEOMETRY_THROUGHPUT) since the end of round 5: 640x480 levels 3..0 (75 / 300 / 1 200 / 4 800 wave steps) take
  // 4 / 4 / 16 / 16 steps per wave -- every level of 1 000 wave steps or more takes 16.  Until round 4 the table was 2 / 4 / 8 / 8
  // (thresholds 70, 250, 1000, 9600): one table had to serve the batch and the single match(); now the latency-first table is a
  // configuration of its own (segment_geometry) and this one is the fastest for batches.  Interleaved runs of the streaming bench
  // in round 5 (pairs/s): profiles/r05_geometry_and_persistent_ab.txt 2/4/8/8 52.1 / 51.7 k, 4/4/8/8 52.8 / 53.4 k, 4/8/8/8 53.0 /
  // 53.0 k; profiles/r05_geometry_ab_16_steps.txt 4/4/8/8 52.3 / 52.2 / 52.4 / 52.5 k, 4/4/8/16 52.5 / 52.8 k, 4/4/16/16 53.9 /
  // 54.1 / 54.1 / 54.6 k, 4/8/16/16 54.3 / 54.3 k, 8/8/16/16 54.3 / 54.5 k, 4/16/16/16 54.6 / 54.4 / 54.2 / 54.0 k, 4/16/16/32
  // 52.6 / 52.9 k, 4/16/32/32 52.0 / 51.6 k: sixteen steps on the two fine levels is worth +3.5 %, thirty-two lose it again (a
  // nine-tile block's prologue and epilogue are ~400 vector instructions per wave: an eighth of a 16-step segment's work, a
  // quarter of an 8-step one's; longer blocks make a launch's tail longer -- the per-launch figure of a launch ALONE on the GPU
  // drops from 0.385 to 0.352 of the HBM roofline while the timed region rises from 0.485 to 0.502) -- and where the table says
  // 16, level_steps makes the segment a whole number of image rows when it can (ten steps for 640x480's two fine levels): see there.
  // Round 4's runs
  // (gpurun_out/r4b, r4c; pairs/s | single-pair latency): 4/8/8/8 46.6 k | 0.83 ms; 8/8/8/8 45.9 k | 0.84; 2/4/8/8 46.2 k | 0.76;
  // 2/4/4/8 45.2 k | 0.72; 1/2/4/8 44.5 k | 0.70; 8/8/8/16 46.1 k | 0.84; 1/1/1/4 (a single pair until round 3) 35.8 k | 0.71.
  long long level_steps_at[5] = {18, 70, 1000, 1000, 1LL << 40};
  // dvo_amd_config::segment_geometry = DVO_AMD_GEOMETRY_LATENCY: 640x480 levels 3..0 take 1 / 2 / 2 / 4 steps per wave (1280x960
  // levels 4..0: 1 / 2 / 2 / 4 / 4): short segments spread a level over more waves -- the shortest single match() of the tables
  // measured (profiles/r05_latency_geometries.txt: 0.60 ms against 0.63 for 1/2/4/8 and 0.66 for the batch's 4/4/8/8) --, a
  // configuration of the tracker honoured by match(), the batched forms, the queue, the validator's stages and the band pipeline
  // alike (round 5)
  long long level_steps_at_latency[5] = {250, 4000, 38400, 999999, 1LL << 40};
  int fine_steps = 0;                  // DVO_AMD_FINE_STEPS: 0 = row-aligned segments on the levels the table gives 16 steps (level_steps), n = n steps
  int fault_slot_alloc = -1;           // DVO_AMD_FAULT_SLOT_ALLOC: fail the allocation of this slot once (tests of the error path)
  bool fin_stamps = false;             // DVO_AMD_FIN_STAMPS=1: k_finalize records phase stamps (diagnostic)
  bool sort_items = true;              // longest-lived blocks first inside a launch (DVO_AMD_SORT_ITEMS=0: slot order)
  bool fin_priority = true;            // the batch reducer's waves run at raised issue priority (DVO_AMD_FIN_PRIORITY=0: off)
  bool small_args = true;              // ticks of at most kMaxSmallItems pairs use the small argument blocks (DVO_AMD_SMALL_ARGS=0: never)
  bool poll = true;                    // wait for a tick by polling the records' sequence words instead of hipStreamSynchronize
  // optional kernel timing (bench.py roofline section)
  bool timing = false;
  double timing_ms = 0.0;
  long long timing_launches = 0;
  std::vector<double> tick_log;  // timing mode: per launch kTickLogFields doubles, see dvo_amd_debug_tick_log
  std::vector<double> tick_log_pending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
};

namespace dvo_amd {
namespace host {

// one Gauss-Newton iteration whose residual pass has been submitted
struct IterCtx {
  int k = 0;
  int buf = 0;
  int steps = 4;  // 64-pixel steps per wave segment of this iteration's residual pass
  int n_blocks = 0;
  SE3 inc;
  SE3 initial_before, estimate_before;
  SE3 initial_after, estimate_after;
  double x_before[6];
  // after the residual pass
  int n = 0;
  float cov[4], P[4];
  double A[36], b[6], x_new[6], prior = 0.0;
  int cut_rank = 0;  // 50 * floor(n / 50): the likelihood keeps the valid residuals ranked below it (Q6)
  int band_valid[kMaxBands];  // valid constraints per band of this iteration's residual pass (sharded pairs only)
  bool cont = false;
  int stats_index = -1;
};

struct Job {
  dvo_amd_pyramid *ref = nullptr, *cur = nullptr;
  const Selection *sel = nullptr;  // stable for the life of `ref` (pyramid_selection)
  dvo_amd_result *result = nullptr;
  JobSlot *slot = nullptr;
  const dvo_amd_config *cfg = nullptr;
  // reference-visible state (names follow dense_tracking.cpp:131-376)
  int level = 0, iteration = 0;
  SE3 inc, initial, estimate;
  double x[6];
  double error = DBL_MAX, last_error = DBL_MAX;
  float precision[4] = {0, 0, 0, 0};
  bool done = false;
  int status = DVO_AMD_OK;
  // in flight
  bool have_a = false, have_b = false;  // a: iteration awaiting its likelihood; b: iteration whose residual pass is in flight
  bool sub_ll = false, sub_res = false;
  IterCtx a, b;
  // Level transitions: when iteration a is the last of its level whatever its likelihood says (a.cont == false), the first
  // residual pass of the next level is submitted in the same tick, assuming a is accepted (it almost always is).
  IterCtx spec_b;
  bool have_spec = false;
  double sub_px = 0.0;  // selected pixels of the residual pass submitted in the current tick
  int buf_flip = 0;   // residual-buffer parity of the current level's iteration 0 (the other one than the previous level's
  int next_flip = 0;  // last likelihood pass reads, so that both can share a launch)
  // the last two iteration entries of the current level (the final result reads one of them, dense_tracking.cpp:368-373)
  dvo_amd_iteration_stats recent[2];
  int recent_count = 0;
  int level_first_iteration = 0;
  double alg_px = 0.0;
  double discarded_px = 0.0;  // selected pixels of speculative residual passes that were thrown away
};

struct GroupTick {
  size_t lo = 0, hi = 0;
  size_t stream_first = 0;  // tick stream of the group's first launch
  int id = 0;
  unsigned seq = 0;
  bool in_flight = false;
};

struct OvfBand {  // a band of the residual pass: wave segments [seg_first, seg_first + n_segs), valid pixels in earlier bands;
  int seg_first, n_segs, rank_offset;  // rank_end >= 0: a closed band (its successor lives on another GPU), see launch_ll_overflow
  int rank_end = -1;
};
}  // namespace host
}  // namespace dvo_amd

using namespace dvo_amd::host;

struct Pending {
  dvo_amd_pyramid *ref = nullptr, *cur = nullptr;
  dvo_amd_result *result = nullptr;
  unsigned long long batch = 0;
  bool has_init = false;
  double T_init[16];
  float ti = 0.0f, td = 0.0f;  // the point-selection thresholds of the configuration the pair was submitted under
};
struct Batch {
  unsigned long long id = 0;
  int remaining = 0;
};
struct Runner {
  std::vector<Job> jobs;                    // one per slot; done = free
  std::vector<unsigned long long> batch_of_slot;  // 0 = free
  std::vector<GroupTick> groups;
  std::deque<Pending> pending;
  std::deque<Batch> batches;                // in submission order; the front is popped once complete
  unsigned long long next_batch = 1;
  size_t next_group = 0;
  int in_flight = 0;
  bool timing = false;                      // the layout was made for kernel timing (one group)
  int resident = 0;
  // a tick failed: the submissions that were still open then ended with its status (a submission that had completed before
  // keeps its OK); wait / poll of ticket 0 ("everything") reports a failure nobody has been told about yet
  struct Failure {
    unsigned long long batch;
    int status;
  };
  std::vector<Failure> failures;
  int unreported_failure = DVO_AMD_OK;
};

namespace dvo_amd {
namespace host {

// ---- dvo_pyramid.cpp
int device_prep_stream(int device, hipStream_t *s);
int slab_alloc(int device, size_t bytes, void **out);
int desc_alloc(int device, void **out);
void desc_free(int device, void *p);
void slab_free(int device, size_t bytes, void *p);
int pyramid_selection(dvo_amd_pyramid *p, float ti, float td, const Selection **out);

// ---- dvo_tracker.cpp
dvo_amd_iteration_stats *stats_push(Job &j);
void stats_publish(Job &j);
void begin_iteration(Job &j, IterCtx &it, int k);
void finish_job(Job &j);
void end_level(Job &j);
void start_level(Job &j);
void speculate_next_level(const Job &j, IterCtx &b_out);
void make_kt(const LevelData &C, const SE3 &estimate, float kt[12]);
int blocks_for(int n, int steps);
int level_blocks(const Selection *sel, int level, int steps);
void scale_and_precision(const FinOut &o, int n, float cov[4], float P[4]);
void system_from_moments(const FinOut &o, const float P[4], double mu, const double xi_initial[6], double A[36], double b[6]);
float loglik_from_sum(int n, const float P[4], double ll_sum, bool overflowed = false);
void process_residual(Job &j, IterCtx &it, const FinOut &o);
void process_loglik(Job &j, const FinOut *outs, bool ll_overflowed = false);
void release_slots(dvo_amd_context *ctx);
int ensure_slots_impl(dvo_amd_context *ctx, int n_jobs, int n_pad);
int ensure_slots(dvo_amd_context *ctx, int n_jobs, int n_pad);
int level_steps(const dvo_amd_context *ctx, const LevelData &lv);
int level_ll_merge(const dvo_amd_context *ctx, int res_steps);
int timing_begin(dvo_amd_context *ctx, size_t *slot);
int tick_stream(dvo_amd_context *ctx, size_t index, hipStream_t *out);
int probe_hw_queue(dvo_amd_context *ctx, int *pipe_queue);  // the hardware queue the main stream runs on, asked of the GPU
int timing_collect(dvo_amd_context *ctx);
int take_wire(const FinWire *w, FinOut *dst_record, unsigned seq, int from_piece);
int take_record(dvo_amd_context *ctx, size_t slot, unsigned seq, int from_piece);
int take_record_synced(dvo_amd_context *ctx, size_t slot, unsigned seq);
int wait_tick(dvo_amd_context *ctx, const std::vector<Job> &jobs, size_t lo, size_t hi, unsigned seq);
int submit_tick(dvo_amd_context *ctx, std::vector<Job> &jobs, GroupTick &grp);
int ll_overflowed(dvo_amd_context *ctx, const float2 *res, const int *seg_prefix, int n_blocks, int steps, int cut_rank,
                  const float P[4], const OvfBand *bands, int n_bands, bool *overflowed);
int complete_tick(dvo_amd_context *ctx, std::vector<Job> &jobs, GroupTick &grp);
bool tick_landed(dvo_amd_context *ctx, const std::vector<Job> &jobs, const GroupTick &grp);
void runner_finish_slot(Runner &R, size_t sidx);
int runner_fail(dvo_amd_context *ctx, int code);
int runner_fail_told(dvo_amd_context *ctx, int code);
int runner_reported_status(Runner &R, unsigned long long ticket);
int queue_must_be_idle(dvo_amd_context *ctx, const char *what);
int runner_step(dvo_amd_context *ctx, size_t g);
int runner_drain(dvo_amd_context *ctx);
int runner_configure(dvo_amd_context *ctx, int in_flight, int n_pad);
int check_config(const dvo_amd_config *c);

// ---- dvo_sharded.cpp
void combine_bands(const FinOut *const *recs, int n, FinOut &out);
int match_one_banded(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, const double *T_init,
                     dvo_amd_result *result, int n_bands, int band_first, int n_local, bool exchange);

}  // namespace host
}  // namespace dvo_amd
