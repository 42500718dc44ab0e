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
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <emmintrin.h>  // the host side of the record hand-off takes 16 bytes at a time (x86-64 hosts)
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dvo_amd.h"
#include "../../include/dvo_amd_debug.h"
#include "dvo_types.h"
#include "se3.h"

using namespace dvo_amd;

namespace {

thread_local std::string g_last_error;

int fail_hip(const char *what, hipError_t e) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return e == hipErrorOutOfMemory ? DVO_AMD_ERR_OUT_OF_MEMORY : DVO_AMD_ERR_HIP;
}

#define HIP_TRY(expr)                                    \
  do {                                                   \
    hipError_t e_ = (expr);                              \
    if (e_ != hipSuccess) return fail_hip(#expr, e_);    \
  } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr int kSpecLevelsMaxPairs = 8;  // speculative level starts (context::spec_levels) up to this many resident pairs
constexpr size_t kTickLogFields = 8;  // doubles per logged launch (dvo_amd_debug_tick_log)
constexpr int kMaxTickStreams = 8;  // streams a context spreads the launches / pair groups of its ticks over

// ---- per-device shared state: a prep stream and a pool of pyramid slabs ------------------------------------------
// Level descriptors (pointers + intrinsics of a pyramid level, ~1 KB per pyramid) are read by every block of every launch
// before it can touch a pixel.  Inside a pyramid's own 20 MB slab they would be a cold line in HBM each time a pair comes
// back to it; kept together in a small arena per device they stay in L2 / Infinity Cache.
constexpr size_t kDescEntryBytes = 1024;
constexpr size_t kDescChunkEntries = 256;
struct DeviceState {
  std::mutex mu;
  hipStream_t prep_stream = nullptr;
  std::vector<std::pair<size_t, void *>> free_slabs;
  std::vector<void *> desc_chunks, desc_free;
};
constexpr int kMaxDevices = 16;
DeviceState g_dev[kMaxDevices];

int device_prep_stream(int device, hipStream_t *s) {
  DeviceState &d = g_dev[device];
  std::lock_guard<std::mutex> lk(d.mu);
  if (!d.prep_stream) HIP_TRY(hipStreamCreateWithFlags(&d.prep_stream, hipStreamNonBlocking));
  *s = d.prep_stream;
  return DVO_AMD_OK;
}

int slab_alloc(int device, size_t bytes, void **out) {
  DeviceState &d = g_dev[device];
  {
    std::lock_guard<std::mutex> lk(d.mu);
    for (size_t i = 0; i < d.free_slabs.size(); ++i)
      if (d.free_slabs[i].first == bytes) {
        *out = d.free_slabs[i].second;
        d.free_slabs.erase(d.free_slabs.begin() + (long)i);
        return DVO_AMD_OK;
      }
  }
  HIP_TRY(hipMalloc(out, bytes));
  return DVO_AMD_OK;
}

int desc_alloc(int device, void **out) {
  DeviceState &d = g_dev[device];
  std::lock_guard<std::mutex> lk(d.mu);
  if (d.desc_free.empty()) {
    void *chunk = nullptr;
    HIP_TRY(hipMalloc(&chunk, kDescEntryBytes * kDescChunkEntries));
    d.desc_chunks.push_back(chunk);
    for (size_t i = kDescChunkEntries; i-- > 0;) d.desc_free.push_back((char *)chunk + i * kDescEntryBytes);
  }
  *out = d.desc_free.back();
  d.desc_free.pop_back();
  return DVO_AMD_OK;
}

void desc_free(int device, void *p) {
  if (!p) return;
  DeviceState &d = g_dev[device];
  std::lock_guard<std::mutex> lk(d.mu);
  d.desc_free.push_back(p);
}

void slab_free(int device, size_t bytes, void *p) {
  DeviceState &d = g_dev[device];
  std::lock_guard<std::mutex> lk(d.mu);
  if (d.free_slabs.size() < 64) {
    d.free_slabs.emplace_back(bytes, p);
  } else {
    (void)hipFree(p);
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// pyramid
// ------------------------------------------------------------------------------------------------------------------

struct Selection {
  float ti, td;
  float *zsel[DVO_AMD_MAX_LEVELS];
  int count[DVO_AMD_MAX_LEVELS];  // PointSelection size (includes an odd trailing point)
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
  void *desc_entry = nullptr;         // this pyramid's entry of the device's descriptor arena
  CurLevelDesc *cur_desc = nullptr;   // device, [levels], in desc_entry
  RefLevelDesc *ref_desc0 = nullptr;  // device, [levels], in desc_entry: room for the first selection's descriptors
  std::mutex mu;
  // entries are never moved or removed while the pyramid lives: a pointer handed out by pyramid_selection() stays valid and may
  // be read without the lock (only the vector itself needs `mu`)
  std::vector<std::unique_ptr<Selection>> selections;
};

namespace {

size_t pyramid_layout(dvo_amd_pyramid *p, char *base) {
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    char *ptr = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return ptr;
  };
  for (int l = 0; l < p->n_levels; ++l) {
    LevelData &L = p->lv[l];
    L.i_plane = (float *)carve(sizeof(float) * L.n);
    L.z_plane = (float *)carve(sizeof(float) * L.n);
    L.c_a = (float4 *)carve(sizeof(float4) * L.n);
    L.c_b = (float2 *)carve(sizeof(float2) * L.n);
    L.r_i = (float *)carve(sizeof(float) * L.n_pad);
    L.r_ix = (float *)carve(sizeof(float) * L.n_pad);
    L.r_iy = (float *)carve(sizeof(float) * L.n_pad);
    L.zsel0 = (float *)carve(sizeof(float) * L.n_pad);
    L.tx = (float *)carve(sizeof(float) * L.w);
    L.ty = (float *)carve(sizeof(float) * L.h);
  }
  p->counters = (int *)carve(sizeof(int) * 2 * DVO_AMD_MAX_LEVELS);
  p->sel_partials = (int2 *)carve(sizeof(int2) * (size_t)(p->lv[0].n_pad / 256 + 1));
  static_assert(sizeof(CurLevelDesc) * DVO_AMD_MAX_LEVELS <= 640 && 640 + sizeof(RefLevelDesc) * DVO_AMD_MAX_LEVELS <= kDescEntryBytes,
                "a pyramid's level descriptors fit one arena entry");
  return off;
}

// a raw sensor frame (frame ingest on the device)
struct RawFrame {
  const unsigned char *image;  // uint8, `channels` interleaved channels (1 = gray, 3 = BGR)
  int channels, image_stride_bytes;
  const unsigned short *depth;  // uint16, 0 = invalid
  int depth_stride;             // in elements
  float depth_scale;
};

int pyramid_build(int device, const float *src_i, const float *src_z, const RawFrame *raw, bool src_on_device, int width,
                  int height, int stride, float fx, float fy, float ox, float oy, int levels, double timestamp,
                  dvo_amd_pyramid **out) {
  if (!out) return DVO_AMD_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (width < 4 || height < 2 || levels < 1 || levels > DVO_AMD_MAX_LEVELS) return DVO_AMD_ERR_INVALID_ARGUMENT;
  if (raw) {
    if (!raw->image || !raw->depth || (raw->channels != 1 && raw->channels != 3) ||
        raw->image_stride_bytes < width * raw->channels || raw->depth_stride < width || !(raw->depth_scale > 0.0f))
      return DVO_AMD_ERR_INVALID_ARGUMENT;
  } else if (!src_i || !src_z || stride < width) {
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DVO_AMD_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev || device >= kMaxDevices) return DVO_AMD_ERR_INVALID_ARGUMENT;
  {
    int w = width, h = height;
    for (int l = 0; l < levels; ++l, w /= 2, h /= 2)
      if (w < 4 || h < 2 || (w % 4) != 0) return DVO_AMD_ERR_INVALID_ARGUMENT;  // see header: width % 4 on every level
  }
  HIP_TRY(hipSetDevice(device));
  hipStream_t st;
  int rc = device_prep_stream(device, &st);
  if (rc) return rc;

  dvo_amd_pyramid *p = new dvo_amd_pyramid();
  p->device = device;
  p->n_levels = levels;
  p->timestamp = timestamp;
  for (int l = 0; l < levels; ++l) {
    LevelData &L = p->lv[l];
    if (l == 0) {
      L.w = width, L.h = height, L.fx = fx, L.fy = fy, L.ox = ox, L.oy = oy;
    } else {
      // RgbdCameraPyramid::build (rgbd_image.cpp:283-296) with IntrinsicMatrix::scale(0.5f) (intrinsic_matrix.cpp:90-93)
      const LevelData &P = p->lv[l - 1];
      L.w = P.w / 2, L.h = P.h / 2;
      L.fx = P.fx * 0.5f, L.fy = P.fy * 0.5f, L.ox = P.ox * 0.5f, L.oy = P.oy * 0.5f;
    }
    L.n = L.w * L.h;
    L.n_pad = (int)align_up((size_t)L.n, kPlanePad);
  }
  p->slab_bytes = pyramid_layout(p, nullptr);
  rc = slab_alloc(device, p->slab_bytes, &p->slab);
  if (rc) {
    delete p;
    return rc;
  }
  pyramid_layout(p, (char *)p->slab);
  rc = desc_alloc(device, &p->desc_entry);
  if (rc) {
    slab_free(device, p->slab_bytes, p->slab);
    delete p;
    return rc;
  }
  p->cur_desc = (CurLevelDesc *)p->desc_entry;
  p->ref_desc0 = (RefLevelDesc *)((char *)p->desc_entry + 640);

  // everything below is enqueued on the device's prep stream; the mutex serialises users of that stream's ordering needs
  auto bail = [&](int code) {
    slab_free(device, p->slab_bytes, p->slab);
    desc_free(device, p->desc_entry);
    delete p;
    return code;
  };
  LevelData &L0 = p->lv[0];
  hipError_t e;
  if (raw) {
    const unsigned char *d_img = raw->image;
    const unsigned short *d_z = raw->depth;
    int img_stride = raw->image_stride_bytes, z_stride = raw->depth_stride;
    if (!src_on_device) {
      // stage the raw bytes (5 B/px instead of 8 B/px of float planes over PCIe) in level 0's gather plane, which is only
      // written by launch_level_planes further down the same stream
      unsigned char *stage_img = (unsigned char *)L0.c_a;
      unsigned short *stage_z = (unsigned short *)(stage_img + align_up((size_t)L0.n * raw->channels, 256));
      const size_t row_img = (size_t)width * raw->channels, row_z = sizeof(unsigned short) * (size_t)width;
      e = hipMemcpy2DAsync(stage_img, row_img, raw->image, (size_t)raw->image_stride_bytes, row_img, height,
                           hipMemcpyHostToDevice, st);
      if (e == hipSuccess)
        e = hipMemcpy2DAsync(stage_z, row_z, raw->depth, sizeof(unsigned short) * (size_t)raw->depth_stride, row_z, height,
                             hipMemcpyHostToDevice, st);
      if (e != hipSuccess) return bail(fail_hip("raw frame upload", e));
      d_img = stage_img, d_z = stage_z, img_stride = (int)row_img, z_stride = width;
    }
    e = launch_ingest(d_img, raw->channels, img_stride, d_z, z_stride, raw->depth_scale, L0.i_plane, L0.z_plane, width,
                      height, st);
  } else if (src_on_device) {
    if (stride == width) {
      e = hipMemcpyAsync(L0.i_plane, src_i, sizeof(float) * L0.n, hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess) e = hipMemcpyAsync(L0.z_plane, src_z, sizeof(float) * L0.n, hipMemcpyDeviceToDevice, st);
    } else {
      e = launch_copy_strided(src_i, stride, L0.i_plane, width, height, st);
      if (e == hipSuccess) e = launch_copy_strided(src_z, stride, L0.z_plane, width, height, st);
    }
  } else {
    e = hipMemcpy2DAsync(L0.i_plane, sizeof(float) * width, src_i, sizeof(float) * stride, sizeof(float) * width, height,
                         hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
      e = hipMemcpy2DAsync(L0.z_plane, sizeof(float) * width, src_z, sizeof(float) * stride, sizeof(float) * width, height,
                           hipMemcpyHostToDevice, st);
  }
  if (e != hipSuccess) return bail(fail_hip("pyramid upload", e));
  for (int l = 0; l < levels; ++l) {
    LevelData &L = p->lv[l];
    if (l > 0) {
      const LevelData &P = p->lv[l - 1];
      e = launch_pyr_down(P.i_plane, P.z_plane, P.w, L.i_plane, L.z_plane, L.w, L.h, st);
      if (e != hipSuccess) return bail(fail_hip("pyr_down", e));
    }
    e = launch_level_planes(L.i_plane, L.z_plane, L.w, L.h, L.n_pad, L.fx, L.fy, L.ox, L.oy, L.c_a, L.c_b, L.r_i, L.r_ix,
                            L.r_iy, L.tx, L.ty, L.h, st);
    if (e != hipSuccess) return bail(fail_hip("level_planes", e));
  }
  CurLevelDesc cur_host[DVO_AMD_MAX_LEVELS];
  std::memset(cur_host, 0, sizeof(cur_host));
  for (int l = 0; l < levels; ++l) {
    const LevelData &C = p->lv[l];
    CurLevelDesc &d = cur_host[l];
    d.c_a = C.c_a, d.c_b = C.c_b, d.w = C.w, d.h = C.h;
    // wcur / wref, dense_tracking.cpp:215-220
    const float wcur_id = 0.5f, wref_id = 0.5f, wcur_zd = 1.0f;
    d.wc[0] = 1.0f / 255.0f, d.wc[1] = 1.0f;
    d.wc[2] = wcur_id * C.fx / 255.0f, d.wc[3] = wcur_id * C.fy / 255.0f;
    d.wc[4] = wcur_zd * C.fx, d.wc[5] = wcur_zd * C.fy;
    d.wr[0] = -1.0f / 255.0f, d.wr[1] = -1.0f;
    d.wr[2] = wref_id * C.fx / 255.0f, d.wr[3] = wref_id * C.fy / 255.0f;
    d.ub_x = (float)(size_t)(C.w - 2), d.ub_y = (float)(size_t)(C.h - 2);
  }
  e = hipMemcpyAsync(p->cur_desc, cur_host, sizeof(CurLevelDesc) * levels, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return bail(fail_hip("pyramid descriptors", e));
  e = hipStreamSynchronize(st);
  if (e != hipSuccess) return bail(fail_hip("pyramid build", e));
  *out = p;
  return DVO_AMD_OK;
}

// PointSelection::select for every level, cached per threshold pair (the reference caches per PointSelection object until
// setRgbdImagePyramid, point_selection.cpp:51-59,100; pyramids are immutable here, so the cache never goes stale)
int pyramid_selection(dvo_amd_pyramid *p, float ti, float td, const Selection **out) {
  std::lock_guard<std::mutex> lk(p->mu);
  for (size_t i = 0; i < p->selections.size(); ++i)
    if (p->selections[i]->ti == ti && p->selections[i]->td == td) {
      *out = p->selections[i].get();
      return DVO_AMD_OK;
    }
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t st;
  int rc = device_prep_stream(p->device, &st);
  if (rc) return rc;
  std::unique_ptr<Selection> sp(new Selection());
  Selection &s = *sp;
  s.ti = ti, s.td = td, s.extra_slab = nullptr, s.extra_bytes = 0;
  if (p->selections.empty()) {
    for (int l = 0; l < p->n_levels; ++l) s.zsel[l] = p->lv[l].zsel0;
    s.ref_desc = p->ref_desc0;
  } else {
    size_t bytes = 0;
    for (int l = 0; l < p->n_levels; ++l) bytes += align_up(sizeof(float) * p->lv[l].n_pad, 256);
    rc = desc_alloc(p->device, &s.desc_entry);
    if (rc) return rc;
    const hipError_t em = hipMalloc(&s.extra_slab, bytes);
    if (em != hipSuccess) {
      desc_free(p->device, s.desc_entry);
      return fail_hip("selection planes", em);
    }
    s.extra_bytes = bytes;
    s.ref_desc = (RefLevelDesc *)s.desc_entry;
    size_t off = 0;
    for (int l = 0; l < p->n_levels; ++l) {
      s.zsel[l] = (float *)((char *)s.extra_slab + off);
      off += align_up(sizeof(float) * p->lv[l].n_pad, 256);
    }
  }
  // any failure below must not leak the selection's own allocation
  auto fail = [&](const char *what, hipError_t e) {
    (void)hipStreamSynchronize(st);
    if (s.extra_slab) (void)hipFree(s.extra_slab);
    desc_free(p->device, s.desc_entry);
    return fail_hip(what, e);
  };
  RefLevelDesc ref_host[DVO_AMD_MAX_LEVELS];
  std::memset(ref_host, 0, sizeof(ref_host));
  for (int l = 0; l < p->n_levels; ++l) {
    const LevelData &R = p->lv[l];
    ref_host[l].r_zsel = s.zsel[l];
    ref_host[l].r_i = R.r_i, ref_host[l].r_ix = R.r_ix, ref_host[l].r_iy = R.r_iy;
    ref_host[l].tx = R.tx, ref_host[l].ty = R.ty;
  }
  hipError_t e = hipMemcpyAsync(s.ref_desc, ref_host, sizeof(RefLevelDesc) * p->n_levels, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return fail("selection descriptors", e);
  for (int l = 0; l < p->n_levels; ++l) {
    const LevelData &L = p->lv[l];
    e = launch_select(L.z_plane, L.c_a, L.c_b, L.n, L.n_pad, ti, td, s.zsel[l], p->counters + 2 * l, p->sel_partials, st);
    if (e != hipSuccess) return fail("select", e);
  }
  int host_counters[2 * DVO_AMD_MAX_LEVELS];
  e = hipMemcpyAsync(host_counters, p->counters, sizeof(int) * 2 * p->n_levels, hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return fail("selection counters", e);
  e = hipStreamSynchronize(st);  // (also keeps ref_host alive until the copy has read it)
  if (e != hipSuccess) return fail("selection", e);
  for (int l = 0; l < p->n_levels; ++l) s.count[l] = host_counters[2 * l], s.last[l] = host_counters[2 * l + 1];
  p->selections.push_back(std::move(sp));
  *out = p->selections.back().get();
  return DVO_AMD_OK;
}

}  // namespace

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
  unsigned *ovf_host = nullptr, *ovf_dev = nullptr;  // pinned word for the verdict of k_ll_overflow (rare path)
  long long ovf_checks = 0, ovf_hits = 0;            // how often the exact overflow check ran / said yes (diagnostic)
  // Wave-step counts OF A LEVEL (its pixels / 64) from which its wave segments take 2 / 4 / 8 / 16 steps: the geometry of a
  // residual pass -- and with it the order every fp32 sum of the pass is taken in -- is a function of the level alone, never
  // of what else is resident in the tick (level_steps below).  DVO_AMD_LEVEL_STEPS_AT="a,b,c,d", read when the context is
  // created (a tuning knob: it changes results in the last bits like any other summation order would).
  // Default (DVO_AMD_GEOMETRY_THROUGHPUT) since round 5: 640x480 levels 3..0 (75 / 300 / 1 200 / 4 800 wave steps) take 4 / 4 / 8 / 8
  // steps per wave, a 1280x960 level 0 (19 200) takes 16.  Until round 4 the table was 2 / 4 / 8 / 8 (thresholds 70, 250, 1000,
  // 9600): one table had to serve the batch and the single match(); now the latency-first table is a configuration of its own
  // (segment_geometry) and this one is the fastest for batches: interleaved runs of the streaming bench in round 5
  // (profiles/r05_geometry_ab.txt; pairs/s): 2/4/8/8 52.1 / 51.7 k, 4/4/8/8 52.8 / 53.4 k, 4/8/8/8 53.0 / 53.0 k.  Round 4's runs
  // (gpurun_out/r4b, r4c; pairs/s | single-pair latency): 4/8/8/8 46.6 k | 0.83 ms; 8/8/8/8 45.9 k | 0.84; 2/4/8/8 46.2 k | 0.76;
  // 2/4/4/8 45.2 k | 0.72; 1/2/4/8 44.5 k | 0.70; 8/8/8/16 46.1 k | 0.84; 1/1/1/4 (a single pair until round 3) 35.8 k | 0.71.
  long long level_steps_at[4] = {18, 70, 1000, 9600};
  // dvo_amd_config::segment_geometry = DVO_AMD_GEOMETRY_LATENCY: 640x480 levels 3..0 take 1 / 2 / 4 / 8 steps per wave (1280x960
  // levels 4..0: 1 / 2 / 4 / 8 / 8): what a single match() got until round 3, as a configuration of the tracker -- honoured by
  // match(), the batched forms, the queue, the validator's stages and the band pipeline alike (round 5)
  long long level_steps_at_latency[4] = {250, 1000, 4000, 38400};
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

namespace {

constexpr int kMaxBands = 16;

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
float loglik_from_sum(int n, const float P[4], double ll_sum, bool overflowed = false) {
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
void process_loglik(Job &j, const FinOut *outs, bool ll_overflowed = false) {
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
  const size_t total = 2 * b_res + b_rec + b_ll + b_lq + 2 * b_sp + b_out;
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
int level_steps(const dvo_amd_context *ctx, int n_px) {
  const long long waves = n_px / kStepPx;
  const long long *t = ctx->cfg.segment_geometry == DVO_AMD_GEOMETRY_LATENCY ? ctx->level_steps_at_latency : ctx->level_steps_at;
  int steps = waves >= t[3] ? 16 : waves >= t[2] ? 8 : waves >= t[1] ? 4 : waves >= t[0] ? 2 : 1;
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
struct GroupTick {
  size_t lo = 0, hi = 0;
  size_t stream_first = 0;  // tick stream of the group's first launch
  int id = 0;
  unsigned seq = 0;
  bool in_flight = false;
};

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
      j.b.steps = level_steps(ctx, j.ref->lv[j.level].n);  // the level's own geometry, whatever else this tick carries
      j.b.n_blocks = blocks_for(j.ref->lv[j.level].n, j.b.steps);
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
      j.sub_res = true;
      j.sub_px = (double)j.sel->count[j.level];
      j.result->n_residual_passes++;
      j.alg_px += j.sub_px;
    } else if (wants_spec(j)) {
      // iteration a ends its level whatever its likelihood says: start the next level in this tick, assuming acceptance
      const int nl = j.level - 1;
      speculate_next_level(j, j.spec_b);
      j.spec_b.steps = level_steps(ctx, j.ref->lv[nl].n);
      j.spec_b.n_blocks = blocks_for(j.ref->lv[nl].n, j.spec_b.steps);
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
struct OvfBand {  // a band of the residual pass: wave segments [seg_first, seg_first + n_segs), valid pixels in earlier bands;
  int seg_first, n_segs, rank_offset;  // rank_end >= 0: a closed band (its successor lives on another GPU), see launch_ll_overflow
  int rank_end = -1;
};
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

}  // namespace

// ---- the queue behind a context: resident pairs + pending pairs ------------------------------------------------------------
// dvo_amd_match_submit appends pairs, every tick of a group hands the slots of finished pairs to pending ones, so the launches
// stay full across calls: a tracker that is fed before it runs dry never drains (the shape of tbb::parallel_reduce with
// grain 1 over a proposal list that keeps growing, keyframe_graph.cpp:587-590).
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

namespace {

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
  const int steps_level = level_steps(ctx, j.ref->lv[j.level].n);
  const int nb_level = blocks_for(j.ref->lv[j.level].n, steps_level);
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

}  // namespace

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
    (void)sscanf(sa, "%lld,%lld,%lld,%lld", &ctx->level_steps_at[0], &ctx->level_steps_at[1], &ctx->level_steps_at[2],
                 &ctx->level_steps_at[3]);
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

int dvo_amd_pyramid_create(int device, const float *intensity, const float *depth, int width, int height, int stride,
                           float fx, float fy, float ox, float oy, int levels, double timestamp, dvo_amd_pyramid **out) {
  return pyramid_build(device, intensity, depth, nullptr, false, width, height, stride, fx, fy, ox, oy, levels, timestamp, out);
}

int dvo_amd_pyramid_create_from_device(int device, const float *d_intensity, const float *d_depth, int width, int height,
                                       int stride, float fx, float fy, float ox, float oy, int levels, double timestamp,
                                       dvo_amd_pyramid **out) {
  return pyramid_build(device, d_intensity, d_depth, nullptr, true, width, height, stride, fx, fy, ox, oy, levels, timestamp,
                       out);
}

int dvo_amd_pyramid_create_raw(int device, const unsigned char *image, int channels, int image_stride_bytes,
                               const unsigned short *depth, int depth_stride, float depth_scale, int on_device, int width,
                               int height, float fx, float fy, float ox, float oy, int levels, double timestamp,
                               dvo_amd_pyramid **out) {
  RawFrame raw{image, channels, image_stride_bytes, depth, depth_stride, depth_scale};
  return pyramid_build(device, nullptr, nullptr, &raw, on_device != 0, width, height, width, fx, fy, ox, oy, levels,
                       timestamp, out);
}

void dvo_amd_pyramid_retain(dvo_amd_pyramid *p) {
  if (p) p->refs.fetch_add(1);
}

void dvo_amd_pyramid_release(dvo_amd_pyramid *p) {
  if (!p) return;
  if (p->refs.fetch_sub(1) != 1) return;
  (void)hipSetDevice(p->device);
  for (auto &s : p->selections) {
    if (s->extra_slab) (void)hipFree(s->extra_slab);
    desc_free(p->device, s->desc_entry);
  }
  desc_free(p->device, p->desc_entry);
  slab_free(p->device, p->slab_bytes, p->slab);
  delete p;
}

int dvo_amd_pyramid_levels(const dvo_amd_pyramid *p) { return p ? p->n_levels : 0; }
double dvo_amd_pyramid_timestamp(const dvo_amd_pyramid *p) { return p ? p->timestamp : 0.0; }

int dvo_amd_pyramid_level_info(const dvo_amd_pyramid *p, int level, int *width, int *height, float k[4]) {
  if (!p || level < 0 || level >= p->n_levels) return DVO_AMD_ERR_INVALID_ARGUMENT;
  const LevelData &L = p->lv[level];
  if (width) *width = L.w;
  if (height) *height = L.h;
  if (k) k[0] = L.fx, k[1] = L.fy, k[2] = L.ox, k[3] = L.oy;
  return DVO_AMD_OK;
}

int dvo_amd_pyramid_download_plane(const dvo_amd_pyramid *p, int level, int plane, float *dst) {
  if (!p || !dst || level < 0 || level >= p->n_levels || plane < 0 || plane > 5) return DVO_AMD_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t st;
  int rc = device_prep_stream(p->device, &st);
  if (rc) return rc;
  const LevelData &L = p->lv[level];
  float *tmp = nullptr;
  HIP_TRY(hipMalloc((void **)&tmp, sizeof(float) * L.n));
  hipError_t e = launch_unpack_plane(L.c_a, L.c_b, plane, L.n, tmp, st);
  if (e == hipSuccess) e = hipMemcpyAsync(dst, tmp, sizeof(float) * L.n, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(tmp);
  if (e != hipSuccess) return fail_hip("download_plane", e);
  return DVO_AMD_OK;
}

int dvo_amd_pyramid_select(dvo_amd_pyramid *p, int level, float ti, float td, int *count, unsigned char *mask) {
  if (!p || level < 0 || level >= p->n_levels) return DVO_AMD_ERR_INVALID_ARGUMENT;
  const Selection *sp = nullptr;
  int rc = pyramid_selection(p, ti, td, &sp);
  if (rc) return rc;
  const Selection &s = *sp;
  if (count) *count = s.count[level];
  if (mask) {
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t st;
    rc = device_prep_stream(p->device, &st);
    if (rc) return rc;
    const LevelData &L = p->lv[level];
    unsigned char *tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&tmp, L.n));
    const int dropped = (s.count[level] & 1) ? s.last[level] : -1;
    hipError_t e = launch_mask_from_zsel(s.zsel[level], L.n, dropped, tmp, st);
    if (e == hipSuccess) e = hipMemcpyAsync(mask, tmp, L.n, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail_hip("select mask", e);
  }
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

namespace {
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
  const int steps = level_steps(ctx, R.n);
  item_set_steps(w, steps, steps);
  const int nb = blocks_for(R.n, steps);
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
  e = launch_finalize(fa, ctx->stream);
  if (e != hipSuccess) return fail_hip("launch_finalize", e);
  return DVO_AMD_OK;
}

int check_level_pair(dvo_amd_context *ctx, dvo_amd_pyramid *reference, dvo_amd_pyramid *current, int level) {
  if (level >= reference->n_levels || level >= current->n_levels) return DVO_AMD_ERR_TOO_FEW_LEVELS;
  if (reference->device != ctx->device || current->device != ctx->device) return DVO_AMD_ERR_DEVICE_MISMATCH;
  if (reference->lv[level].w != current->lv[level].w || reference->lv[level].h != current->lv[level].h)
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  return DVO_AMD_OK;
}
}  // namespace

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
  if (residuals)
    HIP_TRY(hipMemcpyAsync(residuals, ctx->slots[0].res[0], sizeof(float2) * R.n, hipMemcpyDeviceToHost, ctx->stream));
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
    const int st = level_steps(ctx, R.n);  // (the geometry single_tick used)
    rc = ll_overflowed(ctx, ctx->slots[0].res[0], ctx->slots[0].seg_prefix[0], blocks_for(R.n, st), st, 50 * (o.valid / 50), P, nullptr, 0,
                       &overflowed);
    if (rc) return rc;
  }
  out->loglik = loglik_from_sum(o.valid, P, out->loglik_sum, overflowed);
  return DVO_AMD_OK;
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
  int steps = rounds <= 0 ? level_steps(ctx, R.n) : rounds * 4;
  while (steps < kMaxSteps && blocks_for(R.n, steps) > 2048) steps *= 2;
  TickItem proto;
  std::memset(&proto, 0, sizeof(proto));
  item_set_steps(proto, steps, 1);
  proto.res_blocks = (uint16_t)blocks_for(R.n, steps);
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
        const LevelData &C = currents[first + i]->lv[level];
        w.ref = sels[(size_t)(first + i)]->ref_desc + level;
        w.cur = currents[first + i]->cur_desc + level;
        w.slot = ctx->slot_desc + (first + i);
        const float K[9] = {C.fx, 0.0f, C.ox, 0.0f, C.fy, C.oy, 0.0f, 0.0f, 1.0f};
        for (int r = 0; r < 3; ++r)
          for (int cc = 0; cc < 4; ++cc)
            w.kt[r * 4 + cc] = (K[r * 3 + 0] * T[cc * 4 + 0] + K[r * 3 + 1] * T[cc * 4 + 1]) + K[r * 3 + 2] * T[cc * 4 + 2];
      }
      hipError_t e = launch_tick(ta, proto.res_blocks, ctx->stream, e0, e1);  // stamped by the dispatch itself
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

void dvo_amd_se3_exp(const double *xi, double *T) { se3_matrix(se3_exp(xi), T); }
void dvo_amd_se3_log(const double *T, double *xi) { se3_log(se3_from_matrix(T), xi); }
void dvo_amd_solve6(const double *A, const double *b, double *x) { solve_ldlt6(A, b, x); }

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
