// Device pools, pyramid construction and point selection: RgbdCameraPyramid / RgbdImagePyramid / PointSelection of the reference
// (rgbd_image.cpp:38-55,127-172,186-296,419-543, rgbd_image_sse.cpp:241-284, point_selection.cpp:68-152) as device-resident,
// immutable handles, and the dvo_amd_pyramid_* entries of the C ABI.
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
    L.pts0 = carve(kCompactBytesPerPoint * L.n_pad);
    L.tx = (float *)carve(sizeof(float) * L.w);
    L.ty = (float *)carve(sizeof(float) * L.h);
  }
  p->counters = (int *)carve(sizeof(int) * 2 * DVO_AMD_MAX_LEVELS);
  p->sel_partials = (int2 *)carve(sizeof(int2) * (size_t)(p->lv[0].n_pad / 256 + 1));
  p->sel_prefix = (int *)carve(sizeof(int) * (size_t)(p->lv[0].n_pad / 256 + 1));
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
  auto compact_at = [](char *base, int n_pad) {
    CompactLevel c;
    float *f = (float *)base;
    c.z = f, c.i = f + n_pad, c.ix = f + 2 * (size_t)n_pad, c.iy = f + 3 * (size_t)n_pad, c.tx = f + 4 * (size_t)n_pad,
    c.ty = f + 5 * (size_t)n_pad, c.pix = (int *)(f + 6 * (size_t)n_pad);
    return c;
  };
  if (p->selections.empty()) {
    for (int l = 0; l < p->n_levels; ++l) s.zsel[l] = p->lv[l].zsel0, s.pts[l] = compact_at(p->lv[l].pts0, p->lv[l].n_pad);
    s.ref_desc = p->ref_desc0;
  } else {
    size_t bytes = 0;
    for (int l = 0; l < p->n_levels; ++l) bytes += align_up((sizeof(float) + kCompactBytesPerPoint) * p->lv[l].n_pad, 256);
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
      s.pts[l] = compact_at((char *)s.extra_slab + off + sizeof(float) * p->lv[l].n_pad, p->lv[l].n_pad);
      off += align_up((sizeof(float) + kCompactBytesPerPoint) * p->lv[l].n_pad, 256);
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
    ref_host[l].r_zsel = s.pts[l].z;
    ref_host[l].r_i = s.pts[l].i, ref_host[l].r_ix = s.pts[l].ix, ref_host[l].r_iy = s.pts[l].iy;
    ref_host[l].tx = s.pts[l].tx, ref_host[l].ty = s.pts[l].ty;
  }
  hipError_t e = hipMemcpyAsync(s.ref_desc, ref_host, sizeof(RefLevelDesc) * p->n_levels, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return fail("selection descriptors", e);
  for (int l = 0; l < p->n_levels; ++l) {
    const LevelData &L = p->lv[l];
    e = launch_select(L.z_plane, L.c_a, L.c_b, L.n, L.n_pad, ti, td, s.zsel[l], p->counters + 2 * l, p->sel_partials, st);
    if (e != hipSuccess) return fail("select", e);
    // (the partials and the prefix are scratch of the pyramid shared by its levels: the prep stream runs them in order)
    e = launch_compact(s.zsel[l], L.r_i, L.r_ix, L.r_iy, L.tx, L.ty, L.w, L.n, L.n_pad, p->sel_partials, p->sel_prefix, p->counters + 2 * l,
                       s.pts[l].z, s.pts[l].i, s.pts[l].ix, s.pts[l].iy, s.pts[l].tx, s.pts[l].ty, s.pts[l].pix, st);
    if (e != hipSuccess) return fail("compact", e);
  }
  int host_counters[2 * DVO_AMD_MAX_LEVELS];
  e = hipMemcpyAsync(host_counters, p->counters, sizeof(int) * 2 * p->n_levels, hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return fail("selection counters", e);
  e = hipStreamSynchronize(st);  // (also keeps ref_host alive until the copy has read it)
  if (e != hipSuccess) return fail("selection", e);
  for (int l = 0; l < p->n_levels; ++l) s.count[l] = host_counters[2 * l], s.n_pts[l] = s.count[l] & ~1, s.last[l] = host_counters[2 * l + 1];
  p->selections.push_back(std::move(sp));
  *out = p->selections.back().get();
  return DVO_AMD_OK;
}


}  // namespace host
}  // namespace dvo_amd

using namespace dvo_amd;
using namespace dvo_amd::host;

extern "C" {

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

}  // extern "C"
