// TUM RGB-D benchmark file formats: a PNG reader standing in for cv::imread
// (benchmark_slam.cpp:50-51) and the trajectory line of benchmark_slam.cpp:490-504.  Host code only, zlib for inflate.
#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <new>
#include <vector>

#include "../../include/dvo_amd.h"

namespace {

struct Png {
  int width = 0, height = 0, bit_depth = 0, color_type = 0, channels = 0;
  std::vector<unsigned char> palette;  // RGB triples
  std::vector<unsigned char> pixels;   // unfiltered scanlines, `stride` bytes each, samples as stored (16-bit big endian)
  size_t stride = 0;
};

uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int channels_of(int color_type) {
  switch (color_type) {
    case 0: return 1;  // gray
    case 2: return 3;  // RGB
    case 3: return 1;  // palette index
    case 4: return 2;  // gray + alpha
    case 6: return 4;  // RGBA
    default: return 0;
  }
}

int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

int png_load_impl(const char *path, bool header_only, Png *out) {
  if (!path) return DVO_AMD_ERR_INVALID_ARGUMENT;
  FILE *f = std::fopen(path, "rb");
  if (!f) return DVO_AMD_ERR_IO;
  std::vector<unsigned char> file;
  unsigned char buf[65536];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
  std::fclose(f);
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) return DVO_AMD_ERR_FORMAT;
  std::vector<unsigned char> idat;
  bool have_header = false, done = false;
  int interlace = 0;
  size_t pos = 8;
  while (!done && pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    if (pos + 12 + (size_t)len > file.size()) return DVO_AMD_ERR_FORMAT;
    const unsigned char *type = &file[pos + 4], *data = &file[pos + 8];
    if (be32(data + len) != (uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4)) return DVO_AMD_ERR_FORMAT;
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) return DVO_AMD_ERR_FORMAT;
      out->width = (int)be32(data), out->height = (int)be32(data + 4);
      out->bit_depth = data[8], out->color_type = data[9];
      interlace = data[12];
      out->channels = channels_of(out->color_type);
      // sensor frames are a few thousand pixels a side: anything beyond 16384 is a corrupt or hostile header, and the cap keeps
      // (stride + 1) * height far inside size_t
      if (out->width <= 0 || out->height <= 0 || out->width > 16384 || out->height > 16384 || out->channels == 0 ||
          data[10] != 0 || data[11] != 0)
        return DVO_AMD_ERR_FORMAT;
      const int bd = out->bit_depth;
      const bool ok = (out->color_type == 0 && (bd == 1 || bd == 2 || bd == 4 || bd == 8 || bd == 16)) ||
                      (out->color_type == 3 && (bd == 1 || bd == 2 || bd == 4 || bd == 8)) ||
                      ((out->color_type == 2 || out->color_type == 4 || out->color_type == 6) && (bd == 8 || bd == 16));
      if (!ok) return DVO_AMD_ERR_FORMAT;
      have_header = true;
      if (header_only) return DVO_AMD_OK;
    } else if (!std::memcmp(type, "PLTE", 4)) {
      out->palette.assign(data, data + len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      done = true;
    }
    pos += 12 + (size_t)len;
  }
  if (!have_header || !done || idat.empty()) return DVO_AMD_ERR_FORMAT;
  if (interlace != 0) return DVO_AMD_ERR_FORMAT;  // Adam7 is not supported (TUM sequences are not interlaced)
  if (out->color_type == 3 && out->palette.size() < 3) return DVO_AMD_ERR_FORMAT;
  const size_t bits_per_px = (size_t)out->channels * (size_t)out->bit_depth;
  out->stride = ((size_t)out->width * bits_per_px + 7) / 8;
  const size_t bpp = bits_per_px >= 8 ? bits_per_px / 8 : 1;  // filter distance in bytes
  std::vector<unsigned char> raw((out->stride + 1) * (size_t)out->height);
  uLongf raw_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return DVO_AMD_ERR_FORMAT;
  out->pixels.assign(out->stride * (size_t)out->height, 0);
  std::vector<unsigned char> zero(out->stride, 0);
  for (int y = 0; y < out->height; ++y) {
    const unsigned char *src = &raw[(out->stride + 1) * (size_t)y];
    unsigned char *cur = &out->pixels[out->stride * (size_t)y];
    const unsigned char *up = y > 0 ? cur - out->stride : zero.data();
    const int filter = src[0];
    ++src;
    if (filter > 4) return DVO_AMD_ERR_FORMAT;
    for (size_t i = 0; i < out->stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
      int pred = 0;
      switch (filter) {
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: pred = paeth(a, b, c); break;
        default: break;
      }
      cur[i] = (unsigned char)(src[i] + pred);
    }
  }
  return DVO_AMD_OK;
}

// nothing may be thrown across the C ABI: allocation failures of the decoder come back as a status
int png_load(const char *path, bool header_only, Png *out) {
  try {
    return png_load_impl(path, header_only, out);
  } catch (const std::bad_alloc &) {
    return DVO_AMD_ERR_OUT_OF_MEMORY;
  } catch (...) {
    return DVO_AMD_ERR_FORMAT;
  }
}

// sample s of a scanline with bit depth < 8 (packed most significant bits first)
inline int packed_sample(const unsigned char *row, int bit_depth, int s) {
  const int per_byte = 8 / bit_depth, shift = (per_byte - 1 - s % per_byte) * bit_depth;
  return (row[s / per_byte] >> shift) & ((1 << bit_depth) - 1);
}

// Eigen::Quaterniond(rotation matrix): Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Other,3,3>
void quaternion_from_rotation(const double *T, double q[4] /* x y z w */) {
  auto m = [&](int r, int c) { return T[c * 4 + r]; };
  double t = m(0, 0) + m(1, 1) + m(2, 2);
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m(2, 1) - m(1, 2)) * t, q[1] = (m(0, 2) - m(2, 0)) * t, q[2] = (m(1, 0) - m(0, 1)) * t;
  } else {
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m(k, j) - m(j, k)) * t;
    q[j] = (m(j, i) + m(i, j)) * t;
    q[k] = (m(k, i) + m(i, k)) * t;
  }
}

}  // namespace

extern "C" {

int dvo_amd_png_info(const char *path, int *width, int *height, int *channels, int *bit_depth) {
  Png p;
  const int rc = png_load(path, true, &p);
  if (rc) return rc;
  if (width) *width = p.width;
  if (height) *height = p.height;
  if (channels) *channels = p.color_type == 3 ? 3 : p.channels;
  if (bit_depth) *bit_depth = p.bit_depth;
  return DVO_AMD_OK;
}

int dvo_amd_png_read_bgr8(const char *path, unsigned char *dst, int width, int height) {
  if (!dst) return DVO_AMD_ERR_INVALID_ARGUMENT;
  Png p;
  const int rc = png_load(path, false, &p);
  if (rc) return rc;
  if (p.width != width || p.height != height) return DVO_AMD_ERR_INVALID_ARGUMENT;
  const int bd = p.bit_depth, bytes = bd == 16 ? 2 : 1;
  for (int y = 0; y < height; ++y) {
    const unsigned char *row = &p.pixels[p.stride * (size_t)y];
    unsigned char *o = dst + (size_t)y * width * 3;
    for (int x = 0; x < width; ++x, o += 3) {
      int r, g, b;
      if (p.color_type == 3) {
        const int idx = bd == 8 ? row[x] : packed_sample(row, bd, x);
        if ((size_t)idx * 3 + 2 >= p.palette.size()) return DVO_AMD_ERR_FORMAT;
        r = p.palette[3 * idx], g = p.palette[3 * idx + 1], b = p.palette[3 * idx + 2];
      } else if (p.color_type == 0 || p.color_type == 4) {
        int v;
        if (bd < 8) {
          v = packed_sample(row, bd, x) * 255 / ((1 << bd) - 1);  // expanded to the 8-bit range
        } else {
          v = row[(size_t)x * p.channels * bytes];  // 16-bit: the high byte
        }
        r = g = b = v;
      } else {
        const unsigned char *px = row + (size_t)x * p.channels * bytes;
        r = px[0], g = px[bytes], b = px[2 * bytes];
      }
      o[0] = (unsigned char)b, o[1] = (unsigned char)g, o[2] = (unsigned char)r;
    }
  }
  return DVO_AMD_OK;
}

int dvo_amd_png_read_gray16(const char *path, unsigned short *dst, int width, int height) {
  if (!dst) return DVO_AMD_ERR_INVALID_ARGUMENT;
  Png p;
  const int rc = png_load(path, false, &p);
  if (rc) return rc;
  if (p.width != width || p.height != height) return DVO_AMD_ERR_INVALID_ARGUMENT;
  if (p.color_type != 0 && p.color_type != 4) return DVO_AMD_ERR_FORMAT;
  const int bd = p.bit_depth, bytes = bd == 16 ? 2 : 1;
  for (int y = 0; y < height; ++y) {
    const unsigned char *row = &p.pixels[p.stride * (size_t)y];
    for (int x = 0; x < width; ++x) {
      unsigned v;
      if (bd < 8) {
        v = (unsigned)packed_sample(row, bd, x);
      } else {
        const unsigned char *px = row + (size_t)x * p.channels * bytes;
        v = bd == 16 ? ((unsigned)px[0] << 8) | px[1] : px[0];
      }
      dst[(size_t)y * width + x] = (unsigned short)v;
    }
  }
  return DVO_AMD_OK;
}

int dvo_amd_format_trajectory_line(double timestamp, const double *T, char *buf, int capacity) {
  if (!T || !buf || capacity <= 0) return -1;
  // ros::Time::fromSec: sec = floor(t), nsec = round((t - sec) * 1e9), carried into sec when it reaches 1e9;
  // operator<<(ostream&, ros::Time): sec << "." << setw(9) << setfill('0') << nsec
  double sec_d = std::floor(timestamp);
  long long sec = (long long)sec_d;
  long long nsec = (long long)std::floor((timestamp - sec_d) * 1e9 + 0.5);
  if (nsec >= 1000000000LL) sec += 1, nsec -= 1000000000LL;
  double q[4];
  quaternion_from_rotation(T, q);
  const int n = std::snprintf(buf, (size_t)capacity, "%lld.%09lld %g %g %g %g %g %g %g \n", sec, nsec, T[12], T[13], T[14],
                              q[0], q[1], q[2], q[3]);
  return (n < 0 || n >= capacity) ? -1 : n;
}

}  // extern "C"
