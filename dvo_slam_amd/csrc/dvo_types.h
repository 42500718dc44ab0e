// Shared host/device declarations of the dense-tracking kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dvo_amd {

// ---- geometry of one residual pass -------------------------------------------------------------------------------
// A level's pixels are processed in row-major scan order (the order PointSelection::selectPointsFromImage walks them,
// point_selection.cpp:128-149).  A wave owns one contiguous "segment" of kStepPx * steps pixels and walks it in steps of 64
// consecutive pixels, one per lane; a 256-thread block owns four consecutive segments.  steps is a power of two chosen per
// tick: 1 for small launches (every step of a wave is a dependent chain of memory round trips, so short segments spread a
// coarse level over many waves), 8 or 16 when the launch saturates the GPU anyway (fewer per-block records to reduce).
constexpr int kWave = 64;
constexpr int kBlockThreads = 256;
constexpr int kWavesPerBlock = kBlockThreads / kWave;
constexpr int kStepPx = kWave;
constexpr int kMaxSteps = 64;
constexpr int kPlanePad = kStepPx * kMaxSteps * kWavesPerBlock;  // 16384: planes are padded to a multiple of this

// ---- per-block record written by the residual pass ----------------------------------------------------------------
// [0] count (int bits)  [1] first_w  [2] last_r0  [3] last_r1
// [4..6] S under start parity 0 (xx, xy, yy)   [7..9] S under start parity 1
// [10..13] per-wave valid counts (int bits)    [14..15] unused
// [16..102] the 87 plain sums (see kAcc* below)
constexpr int kRecCount = 0, kRecFirstW = 1, kRecLastR = 2, kRecS0 = 4, kRecS1 = 7, kRecWaveCnt = 10, kRecAcc = 16;
constexpr int kNumAcc = 87;
constexpr int kRecStride = 104;
// layout of the 87 sums: for the 21 pairs (i<=j) in row-major upper-triangular order:
//   [0..20]  sum w Ja_i Ja_j      [21..41] sum w (Ja_i Jb_j + Jb_i Ja_j)      [42..62] sum w Jb_i Jb_j
//   [63..68] sum w Ja_i r0   [69..74] sum w Ja_i r1   [75..80] sum w Jb_i r0   [81..86] sum w Jb_i r1
constexpr int kAccAA = 0, kAccAB = 21, kAccBB = 42, kAccAR0 = 63, kAccAR1 = 69, kAccBR0 = 75, kAccBR1 = 81;

// ---- the summation tree of a level ----------------------------------------------------------------------------------------
// A pair's result is a function of its inputs alone: not of what else is resident in the tick, not of which form of the
// reduce kernel ran, and not of how many bands (GPUs) a level was cut into.  Everything that is summed across blocks is
// therefore summed along ONE tree that depends on the level's geometry only:
//   - the nb block records of a level's residual pass are cut into C = level_chunks(nb) contiguous chunks (a power of two up to
//     16; fewer for levels of few blocks), chunk c = blocks [nb c / C, nb (c + 1) / C);
//   - a chunk is cut into 64 / C (ordered part) or 16 / C (moments) leaf ranges the same way; inside a leaf range records are
//     folded in ascending order (fp64);
//   - the leaves are folded by a perfect binary tree ((l0 + l1) + (l2 + l3)) + ...
// A band of a tile-sharded level is a run of whole chunks -- band b of n owns chunks [C b / n, C (b + 1) / n) --, its reducer
// produces the value of its subtree (the leaves it does not own contribute exact zeros / empty records), and the host folds
// the band records with the rest of the same tree: 1, 2, 4, 8 and 16 bands, on one GPU or on as many, give bit-identical sums.
// The likelihood pass walks merged blocks of about m residual blocks that never straddle a chunk.
constexpr int kLevelChunksMax = 16;
__host__ __device__ inline int level_chunks_log2(int nb) { return nb >= 128 ? 4 : nb >= 64 ? 3 : nb >= 32 ? 2 : nb >= 16 ? 1 : 0; }
__host__ __device__ inline int chunk_lo(int nb, int c) { return (nb * c) >> level_chunks_log2(nb); }  // nb <= 2048: no overflow
// leaf range i of 1 << g_log2 (g_log2 = 6: the lanes of the ordered fold, 4: the slices of the moment sums)
__host__ __device__ inline void leaf_range(int nb, int g_log2, int i, int *lo, int *hi) {
  const int per = g_log2 - level_chunks_log2(nb);  // leaves per chunk, log2
  const int c = i >> per, q = i & ((1 << per) - 1);
  const int c_lo = chunk_lo(nb, c), len = chunk_lo(nb, c + 1) - c_lo;
  *lo = c_lo + ((len * q) >> per), *hi = c_lo + ((len * (q + 1)) >> per);
}
// the blocks of band b of n: whole chunks
__host__ __device__ inline void band_blocks(int nb, int n_bands, int b, int *first, int *count) {
  const int C = 1 << level_chunks_log2(nb);
  const int lo = chunk_lo(nb, C * b / n_bands), hi = chunk_lo(nb, C * (b + 1) / n_bands);
  *first = lo, *count = hi - lo;
}
// merged likelihood blocks of a chunk of `len` residual blocks: about 1 << m_log2 residual blocks each, dealt evenly
__host__ __device__ inline int ll_chunk_blocks(int len, int m_log2) {
  const int k = (len + ((1 << m_log2) >> 1)) >> m_log2;
  return len <= 0 ? 0 : (k < 1 ? 1 : k);
}
// merged likelihood blocks in chunks [0, c)
__host__ __device__ inline int ll_blocks_before(int nb, int m_log2, int c) {
  int n = 0;
  for (int i = 0; i < c; ++i) n += ll_chunk_blocks(chunk_lo(nb, i + 1) - chunk_lo(nb, i), m_log2);
  return n;
}
__host__ __device__ inline int ll_blocks_total(int nb, int m_log2) { return ll_blocks_before(nb, m_log2, 1 << level_chunks_log2(nb)); }
// merged likelihood block `lb` of a pass with nb residual blocks -> its residual blocks [first, first + count)
__host__ __device__ inline void ll_block_range(int nb, int m_log2, int lb, int *first, int *count) {
  int before = 0;
  *first = 0, *count = 0;
  const int C = 1 << level_chunks_log2(nb);
  for (int c = 0; c < C; ++c) {
    const int lo = chunk_lo(nb, c), len = chunk_lo(nb, c + 1) - lo;
    const int here = ll_chunk_blocks(len, m_log2);
    if (lb < before + here) {
      const int j = lb - before;
      *first = lo + len * j / here;
      *count = lo + len * (j + 1) / here - *first;
      return;
    }
    before += here;
  }
}

// Static descriptors, resident in device memory next to what they describe and read by every block through scalar loads.
// They are written once (pyramid build / point selection / scratch allocation); nothing is uploaded per match or per tick.
// The reference side of a pair: the SELECTED pixels of the level in scan order, compacted (k_compact; round 5, end -- until then
// these were the level's planes with NaN depth at unselected pixels, and the pass walked every pixel).  Point p of the arrays is
// the p-th selected pixel; entries from the (even) point count up to n_pad are padding with NaN depth.
struct RefLevelDesc {  // one per (pyramid, selection thresholds, level)
  const float *r_zsel;              // depth of the point (NaN in the padding)
  const float *r_i, *r_ix, *r_iy;   // intensity and its derivatives at the point
  const float *tx, *ty;             // the point's ray: ((float)x - ox)/fx, ((float)y - oy)/fy of its pixel (rgbd_image.cpp:198-199)
};
struct CurLevelDesc {  // one per (pyramid, level): the current side of a pair
  const float4 *c_a;  // {I, Z, Ix, Iy} per pixel (gather layout)
  const float2 *c_b;  // {Zx, Zy} per pixel
  int w, h;
  float wc[6];       // wcur: {1/255, 1, .5fx/255, .5fy/255, fx, fy}  (dense_tracking.cpp:219)
  float wr[4];       // wref: {-1/255, -1, .5fx/255, .5fy/255}        (dense_tracking.cpp:220)
  float ub_x, ub_y;  // (float)(w-2), (float)(h-2)
};
struct SlotDesc {  // one per job slot of a context: scratch of the pair being aligned in that slot
  float2 *res[2];       // residual buffers (double buffered by iteration parity), NaN = invalid
  float *records;       // residual pass: n_blocks x kRecStride floats
  double *ll_partials;  // log-likelihood pass: one double per block
  float *ll_qmax;       // ... and the largest Mahalanobis distance the block took (see loglik_pass)
  int *seg_prefix[2];   // per wave segment: valid pixels before it within its band, for the pass that filled res[i]
  float *dbg_w;         // test instrumentation (dvo_amd_debug_weights): when set, the residual pass of the host-rcpps kernels also
                        // stores every pixel's t-distribution weight here; null in every product path
};

// One job's device work of a tick, passed by value in the kernel arguments (no H2D copy per iteration):
// blocks [0, res_blocks) run the residual pass of iteration k+1, blocks [res_blocks, res_blocks + ll_blocks) the
// log-likelihood pass of iteration k.
struct TickItem {
  const RefLevelDesc *ref;
  const CurLevelDesc *cur;
  const SlotDesc *slot;
  uint16_t res_blocks, ll_blocks;  // blocks this item runs (a band of the level when the pair is sharded), <= 2048 each
  uint16_t res_first, ll_first;    // first logical block of the band (0 for the whole level): a residual block / a merged
                                   // likelihood block (ll_block_range)
  uint8_t steps_log2;              // 64-pixel steps per wave segment, log2: low nibble = this item's residual pass, high nibble
                                   // = the residual pass that filled the buffer the likelihood pass reads
  uint8_t flags;                   // kItem* bits; bits 4..6: log2 of the residual blocks per merged likelihood block
  uint16_t ll_level_blocks;        // residual blocks of the WHOLE level in the pass that filled the likelihood pass's buffer
  int ll_cut_rank;                 // log-likelihood pass: keep valid pixels whose rank within the band is below this (Q6)
  float kt[12];                    // K * T[0:3,0:4], row-major, float (dense_tracking_impl.cpp:142-152)
  float P[4];                      // column-major 2x2 precision: of the iteration whose likelihood is evaluated, which is
                                   // also the one the residual pass of the next iteration takes its weights from
};
// step codes: 0..6 = 1 << code; 7.. = segment lengths that are no power of two (an image row of 640 pixels is ten steps)
__host__ __device__ inline int steps_of_code(int c) {
  return c < 7 ? 1 << c : c == 7 ? 10 : c == 8 ? 20 : c == 9 ? 12 : c == 10 ? 14 : c == 11 ? 6 : c == 12 ? 18 : c == 13 ? 24 : c == 14 ? 40 : 30;
}
__host__ __device__ inline int item_res_steps(const TickItem &it) { return steps_of_code(it.steps_log2 & 15); }
__host__ __device__ inline int item_ll_steps(const TickItem &it) { return steps_of_code(it.steps_log2 >> 4); }
__host__ __device__ inline int item_ll_merge_log2(const TickItem &it) { return (it.flags >> 4) & 7; }
inline void item_set_steps(TickItem &it, int res_steps, int ll_steps) {
  auto code = [](int steps) {
    for (int c = 0; c < 16; ++c)
      if (steps_of_code(c) == steps) return c;
    int c = 0;
    while ((1 << c) < steps) ++c;
    return c;
  };
  it.steps_log2 = (uint8_t)(code(res_steps) | (code(ll_steps) << 4));
}
inline void item_set_ll_merge(TickItem &it, int merge) {
  int m = 0;
  while ((1 << m) < merge) ++m;
  it.flags = (uint8_t)((it.flags & 0x8F) | (m << 4));
}
constexpr unsigned kItemResBuf = 1;       // which residual buffer the residual pass writes
constexpr unsigned kItemLlBuf = 2;        // which residual buffer the log-likelihood pass reads
constexpr unsigned kItemUnitWeights = 4;  // first iteration on a level: weights = 1 (dense_tracking.cpp:286-289)
static_assert(sizeof(TickItem) == 104, "TickItem is packed to fit many items into one kernel-argument block");

constexpr int kMaxItemsPerLaunch = 62;  // (tick_locate: lanes 0 .. n_items of one wave look the owner of a block up)
// Opt-in reciprocal mode (dvo_amd_set_reciprocal_mode): the HOST's _mm_rcp_ps, reproduced bit for bit from a table of
// rcpps(1.m) indexed by the top mantissa bits that instruction looks at on this machine (probed when the mode is switched on,
// csrc/dvo_tracker.cpp host_rcp_table).  table == nullptr: the default, exactly truncated quotient / v_rcp_f32.
struct RcpTable {
  const unsigned *table;  // device memory: bits of rcpps(1.m) for m = index << shift
  int shift;              // 23 - (mantissa bits rcpps depends on)
  int unit;               // (nibble form) the lowest mantissa bit any table entry sets: the corrections are in units of 1 << unit
  // The same function without a global-memory gather in the dependent chain of every step (round 5): rcpps(1.m) =
  // (v_rcp_f32(midpoint of m's cell) with the bits below `unit` cleared) + correction << unit, the corrections -- signed 4-bit,
  // one per cell, 0 or 1 on the hosts seen so far -- packed eight to a word: 2 KiB for 2^12 cells, copied into LDS by every
  // block.  Built on the DEVICE from the table above and the device's own v_rcp_f32 when the mode is switched on; null when a
  // correction does not fit four bits (then the table form above runs).
  const unsigned *nibbles;
};
// Q7 (dense_tracking_impl.cpp:702-706) in the host-rcpps mode: computeWeightsSse forms the first 4 floor(V / 4) weights of a pass
// with rcpps and the last V mod 4 by an exact division in double.  Which pixels those are is only known once the pass has counted
// its valid pixels, so the residual pass weights every pixel with the table and k_q7_tail -- one wave per residual pass, between
// k_tick and k_finalize on the same stream -- finds the <= 3 tail pixels, recomputes them and leaves what their exact weights add
// to the pair sums and the 87 moments; k_finalize adds it at the root of its tree.
struct Q7Rec {
  double S[3];          // added to FinOut::S
  double acc[87];       // added to FinOut::acc (kNumAcc)
  int n_tail, valid;    // V mod 4 (0 for a pass with unit weights), V
  int idx[3];           // the tail pixels (index in the level's scan order), ascending
  float w_table[3];     // the weight the residual pass gave them: 7 rcpps(5 + d)
  float w_exact[3];     // computeWeight's: (float)((2.0 + 5.0f) / (5.0f + d))
  int recomputed_equal; // the recomputed residuals of the tail pixels are the spilled ones, bit for bit (always; checked by the tests)
};
static_assert(sizeof(Q7Rec) == 768, "Q7Rec: three 256-byte lines of the slot's block");
constexpr int kRcpNibbleWordsMax = 512;  // 2^12 cells: what fits beside four blocks' staging areas in a CU's LDS
struct TickArgs {
  int n_items;
  int compact;  // 0: grid (blocks of the largest item, n_items); 1: one-dimensional grid without the blocks no item owns
  RcpTable rcp;
  // compact grid: item i owns block groups [group_first[i], group_first[i + 1]) of 8 blocks each (its blocks start on a
  // multiple of 8, so that "blocks b and b + 8 share an XCD" holds inside every item)
  uint16_t group_first[kMaxItemsPerLaunch + 4];
  // Workgroup i of a launch runs on XCD i mod 8, and an item's blocks start on a multiple of 8: without more, the blocks that
  // exist only on some XCDs (the item's count is no multiple of 8; its likelihood blocks are lighter than its residual blocks)
  // always favour the same XCDs.  xcd_rot[i] rotates item i's blocks within their groups of eight -- block b of the grid does
  // the work of block (b & ~7) | ((b + rot) & 7) -- chosen by tick_args_layout so that the XCDs' shares of the launch even out.
  uint8_t xcd_rot[kMaxItemsPerLaunch + 2] = {};
  TickItem items[kMaxItemsPerLaunch];
};
static_assert(sizeof(TickArgs) <= 8192, "kernel argument block too large");

// what the finalize kernel hands to the host for one job (lives in pinned host memory)
struct FinOut {
  int valid;       // V: number of valid constraints of the residual pass
  int has_res;     // a residual pass was reduced
  int has_ll;
  unsigned seq;    // written last (system scope): the tick number this record belongs to
  double S[3];     // sum over pairs (w_2j + w_2j+1) r_2j r_2j^T  (xx, xy, yy), unscaled (Q5 pairing), band starts even
  double S_odd[3]; // the same if the band's first valid pixel had an odd global rank (used when bands are combined)
  float first_w, last_r0, last_r1;  // boundary data of the band for the ordered combine
  float ll_qmax;   // the largest Mahalanobis distance r^T P r among the residuals of the likelihood pass (overflow screen)
  double acc[kNumAcc];
  double ll_sum;   // sum of log(1 + 0.2 r^T P r) over the first 50*floor(V/50) valid residuals (Q6)
};
static_assert(sizeof(FinOut) % 16 == 0, "FinOut is copied in 16-byte pieces");

// How a record travels to the host: 16-byte pieces, each written by one store instruction of one lane, made of two 8-byte
// halves {payload word, tag} that each carry the tick's sequence number.  The host takes a piece with one aligned 16-byte
// load and accepts it when BOTH tags are the tick it waits for, so the kernel needs neither a fence behind the payload nor a
// separate "ready" word behind the fence (that dependent chain was a quarter of k_finalize).  What this rests on is the
// single-copy atomicity of a naturally aligned 8-byte access, which x86-64, PCIe and xGMI guarantee; a 16-byte store that
// arrives as two 8-byte halves at different times (never observed: scripts/probes/piece_atomicity.hip) only delays the
// acceptance of the piece, it cannot pair a new tag with old payload.  The tag 0 is never used (buffers start zeroed).
constexpr int kFinWords = (int)(sizeof(FinOut) / 4);
constexpr int kFinWirePieces = (kFinWords + 1) / 2;
struct alignas(16) FinWire {
  unsigned piece[kFinWirePieces][4];  // {word 2i, tag, word 2i+1, tag}
};
// sequence numbers skip 0, the value of a fresh buffer; the wrap lands on 2 so that the parity -- the generation of an exchange
// slot -- keeps alternating (... 0xFFFFFFFE, 0xFFFFFFFF, 2, 3 ...)
inline unsigned next_seq(unsigned s) { return ++s == 0u ? 2u : s; }

// ---- one-hop exchange of band records between the GPUs of a node (tile-sharded pairs) ------------------------------------
// Every rank owns an exchange buffer of 2 x n_ranks record slots in fine-grained device memory that its peers have mapped
// (hipIpc).  Per tick (sequence number q) rank r writes its band record into slot (q & 1) * n_ranks + r of EVERY rank's
// buffer as 16-byte pieces that carry q as their tag (FinWire: no fence, no ready word); every rank waits until all pieces of
// the n_ranks slots of generation q & 1 carry q and forwards them, still tagged, to its own host.  All of it happens in the tail of k_finalize (no extra launch).  A peer can be at most one tick ahead (it needs this rank's record of tick q to
// finish tick q, and only then publishes q + 1), so two generations never collide.
constexpr int kMaxExchangeRanks = 16;
struct ExchangeArgs {                 // constant per context once the peers are attached; lives in device memory
  FinWire *peers[kMaxExchangeRanks];  // exchange buffers of all ranks as mapped here (own one included)
  const FinWire *local;               // this rank's own exchange buffer
  FinWire *host_records;              // pinned host memory, n_ranks records in rank order (tagged pieces, like every record)
  unsigned *host_seq;                 // pinned host word: q | 0x80000000 when the wait for a peer's record of tick q timed out
  int n_ranks, rank;
  unsigned timeout_ticks;             // bound on the wait in units of 10 ns (s_memrealtime)
  unsigned pad;
};

struct FinItem {
  const float *records;   // residual-pass block records (or null), indexed by logical block of the level
  uint16_t n_blocks, block_first;      // the band this item reduces (the whole level: 0, level_blocks)
  uint16_t n_ll_blocks, ll_first;      // ... and its merged likelihood blocks
  uint16_t level_blocks;               // residual blocks of the whole level in this tick's residual pass (the chunk structure)
  uint16_t ll_level_blocks;            // ... in the pass whose likelihood is summed
  uint16_t ll_merge_log2;              // log2 of the residual blocks per merged likelihood block
  uint16_t q7_off256;                  // host-rcpps mode: the pass's Q7Rec lives 256 * q7_off256 bytes behind ll_partials and its
                                       // S / acc are added to the record's (0: nothing to add)
  const double *ll_partials;
  int *seg_prefix_out;    // per wave segment of the band: valid pixels before it (exclusive scan from the band start)
  FinWire *out;           // host (pinned, device-visible): where the record is published, as tagged pieces
  FinOut *out_dev;        // optional device copy of the record (multi-GPU exchange), or null
  unsigned seq;
  unsigned ll_qmax_off;   // the per-block maxima of the likelihood pass live ll_qmax_off doubles behind ll_partials
};
static_assert(sizeof(FinItem) == 64, "FinItem is packed: one per resident pair in the reducer's argument block");
constexpr unsigned kFinFlagPriority = 1;  // the reducer's waves raise their issue priority (s_setprio)
constexpr int kMaxFinItems = kMaxItemsPerLaunch;  // 64 B each: one reduce launch per tick launch
struct FinArgs {
  int n_items;
  int pad;
  const struct ExchangeArgs *exchange;  // tile-sharded pair: push item 0's record to the peers, gather theirs (else null)
  unsigned xseq;          // sequence number of this tick's exchange
  unsigned pad2;          // kFinFlag* bits
  FinItem items[kMaxFinItems];
};
static_assert(sizeof(FinArgs) <= 4096, "kernel argument block too large");

// k_q7_tail's arguments: the residual-pass items of a tick launch (a TickItem each: level descriptors, slot, geometry, K T, P)
template <int N>
struct Q7ArgsT {
  int n_items;
  int q7_off256;  // where a slot's Q7Rec lives: 256 * q7_off256 bytes behind its ll_partials (the same for every slot of a context)
  RcpTable rcp;
  TickItem items[N];
};
typedef Q7ArgsT<kMaxItemsPerLaunch> Q7Args;

// A tick of at most eight pairs (a single match(), the two-pair front-end step, small batches) goes out with argument blocks a tenth the size:
// the runtime copies the kernel arguments into device-visible memory at every launch, and both launches sit on the critical
// path of a single-pair tick.
constexpr int kMaxSmallItems = 8;
struct TickArgsSmall {
  int n_items;
  int compact;
  RcpTable rcp;
  uint16_t group_first[kMaxSmallItems + 4];
  uint8_t xcd_rot[kMaxSmallItems] = {};
  TickItem items[kMaxSmallItems];
};
struct FinArgsSmall {
  int n_items;
  int pad;  // 0x57A3: record phase stamps (diagnostic)
  FinItem items[kMaxSmallItems];
};

// ---- launch wrappers (dvo_kernels.hip) ----------------------------------------------------------------------------
// Every wrapper takes the process-wide launch lock when DVO_AMD_LAUNCH_LOCK=1 (profiled multi-thread runs only: see
// profiles/r03_rocprofv3_sigsegv_root_cause.md).
// t_start / t_stop (both or neither): events that receive the begin / end time stamps of this dispatch itself
// (hipExtLaunchKernelGGL), i.e. the kernel's own duration without the launch latency an event pair around it would add
// grid: args.compact ? (args.group_first[n_items] * 8) blocks : (max_blocks rounded up to 8, n_items)
hipError_t launch_tick(const TickArgs &args, int max_blocks, hipStream_t stream, hipEvent_t t_start = nullptr,
                       hipEvent_t t_stop = nullptr);
// fills group_first / compact from the items' block counts; returns the number of blocks of the compact grid
int tick_args_layout(TickArgs &args, int max_blocks);
int tick_args_layout(TickArgsSmall &args, int max_blocks);
// the same kernels with the small argument blocks (only the default k_tick form: returns hipErrorNotSupported when
// DVO_AMD_ACCUM=valu selected the register form, and the caller falls back to the full-size launch)
hipError_t launch_tick_small(const TickArgsSmall &args, int max_blocks, hipStream_t stream, hipEvent_t t_start = nullptr,
                             hipEvent_t t_stop = nullptr);
hipError_t launch_finalize_small(const FinArgsSmall &args, hipStream_t stream);
// (host-rcpps mode) the Q7 tail of every residual pass of a tick launch: between launch_tick and launch_finalize, same stream
typedef Q7ArgsT<kMaxSmallItems> Q7ArgsSmall;
hipError_t launch_q7_tail(const Q7Args &args, hipStream_t stream);
hipError_t launch_q7_tail_small(const Q7ArgsSmall &args, hipStream_t stream);
hipError_t launch_finalize(const FinArgs &args, hipStream_t stream);
// exact emulation of the reference's overflowing 50-term likelihood product over (a band of) one residual buffer (rare; see
// k_ll_overflow): wave segments [seg_first, seg_first + n_segs) of seg_px pixels each, seg_prefix = the pass's prefix table
// (valid pixels before each segment, relative to the band), rank_offset = valid pixels in earlier bands, n_px = pixels the
// pass wrote.  *result_host (pinned, zeroed by the caller) is set to 1 when a group of fifty overflowed.
// rank_end < 0: an open band (the residuals behind it are in the same buffer: the last chunk walks on until its last group is
// complete); rank_end >= 0: a CLOSED band of a pair sharded over several GPUs, rank_end = rank of the first valid pixel behind
// it -- only the groups that end inside the band are judged
hipError_t launch_ll_overflow(const float2 *res, const int *seg_prefix, int seg_first, int n_segs, int seg_px, int rank_offset,
                              int n_px, int cut_rank, int rank_end, const float P[4], unsigned *result_host, hipStream_t stream);
// push a ready-made record (device memory) through the one-hop exchange as tick `xseq`
hipError_t launch_exchange_record(const FinOut *rec_dev, const struct ExchangeArgs *exchange_dev, unsigned xseq, hipStream_t stream);
// a Mahalanobis distance below this cannot make a group of fifty terms 1 + 0.2 q overflow a double (50 log2(1 + 0.2 q) < 1024)
constexpr float kLlOverflowScreen = 7.0e6f;
hipError_t read_finalize_stamps(unsigned long long out[8]);
long long read_block_trace(unsigned long long *out, long long capacity);  // -1: not a trace build (see dvo_kernels.hip)
// out[i] = the table reciprocal of in[i] (device pointers): the unit test of the opt-in host-rcpps mode (the nibble form when
// rcp.nibbles is set, else the table form)
hipError_t launch_rcp_table_probe(const RcpTable &rcp, const float *in, float *out, int n, hipStream_t stream);
// out_rn[i] / out_rtz[i] = bits of v_rcp_f32(midpoint of cell i of [1, 2)) under round-to-nearest / round-toward-zero, i < 1 << k
hipError_t launch_rcp_midpoint_probe(int k, unsigned *out_rn, unsigned *out_rtz, hipStream_t stream);

hipError_t launch_marker(unsigned tag, hipStream_t stream);
// the hardware queue a stream runs on: *out_pinned (pinned host word, zeroed by the caller) = 0x80000000 | pipe << 3 | queue
hipError_t launch_queue_probe(unsigned *out_pinned, hipStream_t stream);  // a no-op dispatch named k_marker (profile bracketing)
int acc_mode();  // 1: Gram matrix on the matrix pipe (default); 0: DVO_AMD_ACCUM=valu, the 87-register cross-check form

// prep (pyramid construction) kernels
hipError_t launch_pyr_down(const float *i_prev, const float *z_prev, int w_prev, float *i_out, float *z_out, int w, int h,
                           hipStream_t stream);
hipError_t launch_level_planes(const float *i_plane, const float *z_plane, int w, int h, int n_pad, float fx, float fy,
                               float ox, float oy, float4 *c_a, float2 *c_b, float *r_i, float *r_ix, float *r_iy,
                               float *tx, float *ty, int ty_len, hipStream_t stream);
// selection: writes zsel (n_pad floats), counters[0] = count, counters[1] = index of last selected pixel (or -1);
// then un-selects the last selected pixel if count is odd (Q3).
// block_partials: scratch of n_pad / 256 int2 (per-block count and last index; no atomics)
hipError_t launch_select(const float *z_plane, const float4 *c_a, const float2 *c_b, int n, int n_pad, float ti, float td,
                         float *zsel, int *counters, int2 *block_partials, hipStream_t stream);
// raw frame -> float base planes of level 0 (uint8 gray or BGR, uint16 depth with 0 = invalid)
hipError_t launch_ingest(const unsigned char *img, int channels, int img_stride_bytes, const unsigned short *raw_z,
                         int z_stride, float z_scale, float *i_plane, float *z_plane, int w, int h, hipStream_t stream);
hipError_t launch_copy_strided(const float *src, int stride, float *dst, int w, int h, hipStream_t stream);
hipError_t launch_compact(const float *zsel, const float *r_i, const float *r_ix, const float *r_iy, const float *tx, const float *ty, int w,
                          int n, int n_pad, const int2 *block_partials, int *prefix, const int *counters, float *cz, float *ci, float *cix,
                          float *ciy, float *ctx, float *cty, int *cpix, hipStream_t stream);
hipError_t launch_mask_from_zsel(const float *zsel, int n, int last_dropped, unsigned char *mask, hipStream_t stream);
hipError_t launch_unpack_plane(const float4 *c_a, const float2 *c_b, int plane, int n, float *dst, hipStream_t stream);

}  // namespace dvo_amd
