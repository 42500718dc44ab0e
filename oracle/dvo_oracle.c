/*
 * dvo_oracle.c -- TEST INFRASTRUCTURE ONLY (see dvo_oracle.h).  PARITY UNPINNED.
 *
 * Plain-C restatement of dvo::DenseTracker::match() and the parts of dvo::core it reaches.
 * Build with:  gcc -O3 -msse3 -mno-fma -ffp-contract=off -frounding-math   (oracle/Makefile)
 * -ffp-contract=off / -mno-fma keep every float product and sum a separately rounded SSE
 * operation, which is what the reference's 2013-era SSE3 build executed; the residual stage
 * additionally runs with MXCSR set to round-toward-zero, as the reference does.
 *
 * Third-party arithmetic that is NOT under /root/reference (fetched by its Makefiles at build
 * time, unpinned): Sophus SE3d exp/log/inverse/compose (strasdat/Sophus master, templated
 * se3.hpp / so3.hpp), Eigen 3 Matrix2f::inverse / determinant / LDLT.  Their published closed
 * forms are restated below (se3_* and ldlt6_solve, inverse2f).
 */
#include "dvo_oracle.h"

#include <float.h>
#include <math.h>
#include <pmmintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <xmmintrin.h>

/* ------------------------------------------------------------------------------------------ */
/* data                                                                                        */
/* ------------------------------------------------------------------------------------------ */

/* PointWithIntensityAndDepth, rgbd_image.h:39-89: {x,y,z,w} + {i,z,idx,idy,zdx,zdy,0,0} */
typedef struct {
  float p[4];
  float e[8];
} orc_record;

typedef struct {
  int w, h;
  float fx, fy, ox, oy;
  float *plane[6]; /* I Z Ix Iy Zx Zy */
  float *accel;    /* 8 floats per pixel, rgbd_image.cpp:534-543 */
  float *cloud;    /* 4 floats per pixel, rgbd_image.cpp:245-262 */
  orc_record *sel;
  int *sel_idx;
  int n_sel;
  int sel_ok;
  float sel_ti, sel_td;
} orc_level;

struct orc_pyramid {
  int n_levels;
  orc_level lv[ORC_MAX_LEVELS];
};

static void *xalloc(size_t bytes) {
  void *p = NULL;
  if (bytes == 0) bytes = 16;
  if (posix_memalign(&p, 64, bytes) != 0) return NULL;
  return p;
}

void orc_default_config(orc_config *c) {
  /* dense_tracking_config.cpp:27-41 */
  c->first_level = 3;
  c->last_level = 1;
  c->max_iterations_per_level = 100;
  c->precision = 5e-7;
  c->mu = 0.0;
  c->use_initial_estimate = 0;
  c->intensity_derivative_threshold = 0.0f;
  c->depth_derivative_threshold = 0.0f;
  c->rcp_mode = ORC_RCP_SSE;
  c->sum_mode = ORC_SUM_REFERENCE;
  c->ll_guard = 0;
}

/* ------------------------------------------------------------------------------------------ */
/* pyramid: rgbd_image.cpp                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* pyrDownMeanSmooth<float>, rgbd_image.cpp:38-55 */
static void down_mean(const float *in, int iw, float *out, int ow, int oh) {
  for (int y = 0; y < oh; ++y)
    for (int x = 0; x < ow; ++x) {
      const float *r0 = in + (size_t)(2 * y) * iw + 2 * x;
      const float *r1 = r0 + iw;
      out[(size_t)y * ow + x] = (r0[0] + r0[1] + r1[0] + r1[1]) / 4.0f;
    }
}

/* pyrDownSubsample<float>, rgbd_image.cpp:127-139 */
static void down_subsample(const float *in, int iw, float *out, int ow, int oh) {
  for (int y = 0; y < oh; ++y)
    for (int x = 0; x < ow; ++x) out[(size_t)y * ow + x] = in[(size_t)(2 * y) * iw + 2 * x];
}

/* calculateDerivativeX / Y, rgbd_image.cpp:419-472 and rgbd_image_sse.cpp:241-284 (same values) */
static void deriv_x(const float *img, int w, int h, float *out) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int prev = x - 1 < 0 ? 0 : x - 1;
      int next = x + 1 > w - 1 ? w - 1 : x + 1;
      out[(size_t)y * w + x] = (img[(size_t)y * w + next] - img[(size_t)y * w + prev]) * 0.5f;
    }
}

static void deriv_y(const float *img, int w, int h, float *out) {
  for (int y = 0; y < h; ++y) {
    int prev = y - 1 < 0 ? 0 : y - 1;
    int next = y + 1 > h - 1 ? h - 1 : y + 1;
    for (int x = 0; x < w; ++x)
      out[(size_t)y * w + x] = (img[(size_t)next * w + x] - img[(size_t)prev * w + x]) * 0.5f;
  }
}

static void level_finish(orc_level *L) {
  const size_t n = (size_t)L->w * L->h;
  for (int k = 2; k < 6; ++k) L->plane[k] = (float *)xalloc(n * sizeof(float));
  deriv_x(L->plane[0], L->w, L->h, L->plane[2]);
  deriv_y(L->plane[0], L->w, L->h, L->plane[3]);
  deriv_x(L->plane[1], L->w, L->h, L->plane[4]);
  deriv_y(L->plane[1], L->w, L->h, L->plane[5]);

  /* buildAccelerationStructure, rgbd_image.cpp:534-543: {I,Z,Ix,Iy,Zx,Zy,0,0} */
  L->accel = (float *)xalloc(n * 8 * sizeof(float));
  for (size_t i = 0; i < n; ++i) {
    float *a = L->accel + i * 8;
    for (int k = 0; k < 6; ++k) a[k] = L->plane[k][i];
    a[6] = 0.0f;
    a[7] = 0.0f;
  }

  /* RgbdCamera::RgbdCamera + buildPointCloud, rgbd_image.cpp:186-204,245-262 */
  L->cloud = (float *)xalloc(n * 4 * sizeof(float));
  size_t idx = 0;
  for (int y = 0; y < L->h; ++y)
    for (int x = 0; x < L->w; ++x, ++idx) {
      float tx = ((float)(size_t)x - L->ox) / L->fx;
      float ty = ((float)(size_t)y - L->oy) / L->fy;
      float d = L->plane[1][idx];
      float *c = L->cloud + idx * 4;
      c[0] = tx * d;
      c[1] = ty * d;
      c[2] = 1.0f * d;
      c[3] = 1.0f;
    }
  L->sel = NULL;
  L->sel_idx = NULL;
  L->sel_ok = 0;
  L->n_sel = 0;
}

orc_pyramid *orc_pyramid_create(const float *intensity, const float *depth, int width, int height, float fx,
                                float fy, float ox, float oy, int levels) {
  if (levels < 1 || levels > ORC_MAX_LEVELS || width < 2 || height < 2) return NULL;
  orc_pyramid *p = (orc_pyramid *)calloc(1, sizeof(orc_pyramid));
  p->n_levels = levels;
  for (int l = 0; l < levels; ++l) {
    orc_level *L = &p->lv[l];
    if (l == 0) {
      L->w = width;
      L->h = height;
      L->fx = fx;
      L->fy = fy;
      L->ox = ox;
      L->oy = oy;
    } else {
      /* RgbdCameraPyramid::build, rgbd_image.cpp:283-296; IntrinsicMatrix::scale, intrinsic_matrix.cpp:90-93 */
      const orc_level *P = &p->lv[l - 1];
      L->w = P->w / 2;
      L->h = P->h / 2;
      L->fx = P->fx * 0.5f;
      L->fy = P->fy * 0.5f;
      L->ox = P->ox * 0.5f;
      L->oy = P->oy * 0.5f;
    }
    const size_t n = (size_t)L->w * L->h;
    L->plane[0] = (float *)xalloc(n * sizeof(float));
    L->plane[1] = (float *)xalloc(n * sizeof(float));
    if (l == 0) {
      memcpy(L->plane[0], intensity, n * sizeof(float));
      memcpy(L->plane[1], depth, n * sizeof(float));
    } else {
      /* RgbdImagePyramid::build, rgbd_image.cpp:156-172 */
      const orc_level *P = &p->lv[l - 1];
      down_mean(P->plane[0], P->w, L->plane[0], L->w, L->h);
      down_subsample(P->plane[1], P->w, L->plane[1], L->w, L->h);
    }
    level_finish(L);
  }
  return p;
}

void orc_pyramid_destroy(orc_pyramid *p) {
  if (!p) return;
  for (int l = 0; l < p->n_levels; ++l) {
    orc_level *L = &p->lv[l];
    for (int k = 0; k < 6; ++k) free(L->plane[k]);
    free(L->accel);
    free(L->cloud);
    free(L->sel);
    free(L->sel_idx);
  }
  free(p);
}

int orc_pyramid_levels(const orc_pyramid *p) { return p->n_levels; }
void orc_level_size(const orc_pyramid *p, int level, int *w, int *h) {
  *w = p->lv[level].w;
  *h = p->lv[level].h;
}
void orc_level_intrinsics(const orc_pyramid *p, int level, float k[4]) {
  k[0] = p->lv[level].fx;
  k[1] = p->lv[level].fy;
  k[2] = p->lv[level].ox;
  k[3] = p->lv[level].oy;
}
const float *orc_level_plane(orc_pyramid *p, int level, int plane) { return p->lv[level].plane[plane]; }

/* PointSelection::select / selectPointsFromImage, point_selection.cpp:89-152;
 * ValidPointAndGradientThresholdPredicate::isPointOk, point_selection.h:63-66 */
static void level_select(orc_level *L, float ti, float td) {
  if (L->sel_ok && L->sel_ti == ti && L->sel_td == td) return;
  const size_t n = (size_t)L->w * L->h;
  if (!L->sel) {
    L->sel = (orc_record *)xalloc(n * sizeof(orc_record));
    L->sel_idx = (int *)xalloc(n * sizeof(int));
  }
  int cnt = 0;
  for (size_t i = 0; i < n; ++i) {
    const float *c = L->cloud + i * 4;
    const float *a = L->accel + i * 8;
    const float z = c[2], idx = a[2], idy = a[3], zdx = a[4], zdy = a[5];
    int ok = z == z && zdx == zdx && zdy == zdy &&
             (fabsf(idx) > ti || fabsf(idy) > ti || fabsf(zdx) > td || fabsf(zdy) > td);
    if (ok) {
      memcpy(L->sel[cnt].p, c, 4 * sizeof(float));
      memcpy(L->sel[cnt].e, a, 8 * sizeof(float));
      L->sel_idx[cnt] = (int)i;
      ++cnt;
    }
  }
  L->n_sel = cnt;
  L->sel_ok = 1;
  L->sel_ti = ti;
  L->sel_td = td;
}

int orc_select(orc_pyramid *p, int level, float ti, float td, const float **records) {
  level_select(&p->lv[level], ti, td);
  if (records) *records = (const float *)p->lv[level].sel;
  return p->lv[level].n_sel;
}

const int *orc_select_index(orc_pyramid *p, int level, float ti, float td) {
  level_select(&p->lv[level], ti, td);
  return p->lv[level].sel_idx;
}

/* ------------------------------------------------------------------------------------------ */
/* residual stage: computeResidualsSse<Debug>, dense_tracking_impl.cpp:133-393                 */
/* ------------------------------------------------------------------------------------------ */

float orc_host_rcp(float x) { return _mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(x))); }
/* the same over an array: what _mm_rcp_ps (dense_tracking_impl.cpp:192,700) is on THIS host, for the tests of the GPU's table mode */
void orc_host_rcp_many(const float *in, float *out, int n) {
  for (int i = 0; i < n; ++i) out[i] = _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(in[i])));
}

/* depthStdDevZ, dense_tracking_impl.cpp:122-128 */
static inline float depth_sigma(float depth) {
  float s = depth - 0.4f;
  s = 0.0012f + 0.0019f * s * s;
  return s;
}

typedef struct {
  const orc_record *first;
  int n_sel;
  const float *accel;
  int w, h;
  float kt[12]; /* row-major 3x4 */
  float wref[8], wcur[8];
  int rcp_mode;
  orc_record *out_pe;
  float *out_r;
  unsigned char *out_valid;
} warp_args;

/*
 * The body below must run with MXCSR.RC = toward zero (dense_tracking_impl.cpp:165-167); the caller sets the mode
 * around this noinline function so that the compiler cannot move float operations across the mode switch.
 * The reference handles two points per loop trip with packed arithmetic; lanes never mix between the two points,
 * so handling them one after the other with the same operation order gives the same bits.
 */
static __attribute__((noinline)) int warp_residuals_rtz(const warp_args *a) {
  const float *kt = a->kt;
  const __m128 wcur_a = _mm_loadu_ps(a->wcur), wcur_b = _mm_loadu_ps(a->wcur + 4);
  const __m128 wref_a = _mm_loadu_ps(a->wref), wref_b = _mm_loadu_ps(a->wref + 4);
  const float ub_x = (float)(size_t)(a->w - 2), ub_y = (float)(size_t)(a->h - 2);
  /* Q3: an odd trailing point is never looked at (dense_tracking_impl.cpp:169-171) */
  const int n_proc = a->n_sel - (a->n_sel % 2);
  int n_out = 0;

  for (int i = 0; i < n_proc; ++i) {
    const orc_record *pt = a->first + i;
    const float x = pt->p[0], y = pt->p[1], z = pt->p[2], w = pt->p[3];
    /* hadd(hadd(.)) pairs lanes (0,1) and (2,3) first: dense_tracking_impl.cpp:178-188 */
    const float sx = (kt[0] * x + kt[1] * y) + (kt[2] * z + kt[3] * w);
    const float sy = (kt[4] * x + kt[5] * y) + (kt[6] * z + kt[7] * w);
    const float sz = (kt[8] * x + kt[9] * y) + (kt[10] * z + kt[11] * w);
    float rz;
    if (a->rcp_mode == ORC_RCP_SSE)
      rz = _mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(sz))); /* :192 */
    else
      rz = _mm_cvtss_f32(_mm_div_ps(_mm_set1_ps(1.0f), _mm_set1_ps(sz)));
    const float u = sx * rz, v = sy * rz;
    /* _mm_cvtps_epi32 under RTZ == truncation (:195) */
    const int iu = _mm_cvtt_ss2si(_mm_set_ss(u)), iv = _mm_cvtt_ss2si(_mm_set_ss(v));
    const float fu = (float)iu, fv = (float)iv;
    const float w1u = u - fu, w1v = v - fv;
    const float w0u = 1.0f - w1u, w0v = 1.0f - w1v;
    const int inb = (u >= 0.0f) && (u <= ub_x) && (v >= 0.0f) && (v <= ub_y); /* :203 */
    unsigned char ok = 0;
    if (inb) {
      const float *r0 = a->accel + ((size_t)iv * a->w + iu) * 8;
      const float *r1 = r0 + (size_t)a->w * 8;
      const __m128 W0u = _mm_set1_ps(w0u), W1u = _mm_set1_ps(w1u);
      const __m128 W0v = _mm_set1_ps(w0v), W1v = _mm_set1_ps(w1v);
      /* :227-258 */
      __m128 a1 = _mm_mul_ps(W0v, _mm_add_ps(_mm_mul_ps(W0u, _mm_load_ps(r0)), _mm_mul_ps(W1u, _mm_load_ps(r0 + 8))));
      __m128 b1 = _mm_mul_ps(W0v, _mm_add_ps(_mm_mul_ps(W0u, _mm_load_ps(r0 + 4)), _mm_mul_ps(W1u, _mm_load_ps(r0 + 12))));
      __m128 a2 = _mm_mul_ps(W1v, _mm_add_ps(_mm_mul_ps(W0u, _mm_load_ps(r1)), _mm_mul_ps(W1u, _mm_load_ps(r1 + 8))));
      __m128 b2 = _mm_mul_ps(W1v, _mm_add_ps(_mm_mul_ps(W0u, _mm_load_ps(r1 + 4)), _mm_mul_ps(W1u, _mm_load_ps(r1 + 12))));
      __m128 ia = _mm_add_ps(a1, a2), ib = _mm_add_ps(b1, b2);
      if (_mm_movemask_ps(_mm_cmpunord_ps(ia, ib)) == 0) { /* :261 */
        float refa[4] = {pt->e[0], sz, pt->e[2], pt->e[3]}; /* :269, lane 1 <- transformed depth */
        __m128 ra = _mm_add_ps(_mm_mul_ps(wcur_a, ia), _mm_mul_ps(wref_a, _mm_loadu_ps(refa)));
        float ra_s[4];
        _mm_storeu_ps(ra_s, ra);
        /* occlusion test :275 */
        if (ra_s[1] > -20.0f * depth_sigma(pt->e[1])) {
          __m128 rb = _mm_add_ps(_mm_mul_ps(wcur_b, ib), _mm_mul_ps(wref_b, _mm_loadu_ps(pt->e + 4)));
          if (a->out_pe) {
            orc_record *o = a->out_pe + n_out;
            memcpy(o->p, pt->p, 4 * sizeof(float));
            _mm_storeu_ps(o->e, ra);
            _mm_storeu_ps(o->e + 4, rb);
          }
          if (a->out_r) {
            a->out_r[2 * n_out] = ra_s[0];
            a->out_r[2 * n_out + 1] = ra_s[1];
          }
          ++n_out;
          ok = 1;
        }
      }
    }
    if (a->out_valid) a->out_valid[i] = ok;
  }
  return n_out;
}

static int warp_residuals(warp_args *a) {
  const unsigned int old_mode = _MM_GET_ROUNDING_MODE();
  _MM_SET_ROUNDING_MODE(_MM_ROUND_TOWARD_ZERO);
  const int n = warp_residuals_rtz(a);
  _MM_SET_ROUNDING_MODE(old_mode);
  return n;
}

/* K * T(0:3, 0:4) in float as Eigen's coefficient-based small product evaluates it, dense_tracking_impl.cpp:142-148 */
static void make_kt(const orc_level *cur, const float *T /* col-major 4x4 */, float kt[12]) {
  const float K[9] = {cur->fx, 0.0f, cur->ox, 0.0f, cur->fy, cur->oy, 0.0f, 0.0f, 1.0f};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j)
      kt[i * 4 + j] = (K[i * 3 + 0] * T[j * 4 + 0] + K[i * 3 + 1] * T[j * 4 + 1]) + K[i * 3 + 2] * T[j * 4 + 2];
}

/* wcur / wref, dense_tracking.cpp:215-220 */
static void make_weights8(const orc_level *cur, float wref[8], float wcur[8]) {
  const float wcur_id = 0.5f, wref_id = 0.5f, wcur_zd = 1.0f, wref_zd = 0.0f;
  const float c[8] = {1.0f / 255.0f, 1.0f, wcur_id * cur->fx / 255.0f, wcur_id * cur->fy / 255.0f,
                      wcur_zd * cur->fx, wcur_zd * cur->fy, 0.0f, 0.0f};
  const float r[8] = {-1.0f / 255.0f, -1.0f, wref_id * cur->fx / 255.0f, wref_id * cur->fy / 255.0f,
                      wref_zd * cur->fx, wref_zd * cur->fy, 0.0f, 0.0f};
  memcpy(wcur, c, sizeof(c));
  memcpy(wref, r, sizeof(r));
}

int orc_compute_residuals(orc_pyramid *ref, orc_pyramid *cur, int level, float ti, float td, const float *T,
                          int rcp_mode, float *out_points_error, float *out_residuals, unsigned char *out_valid) {
  orc_level *R = &ref->lv[level];
  orc_level *C = &cur->lv[level];
  level_select(R, ti, td);
  warp_args a;
  a.first = R->sel;
  a.n_sel = R->n_sel;
  a.accel = C->accel;
  a.w = C->w;
  a.h = C->h;
  make_kt(C, T, a.kt);
  make_weights8(C, a.wref, a.wcur);
  a.rcp_mode = rcp_mode;
  a.out_pe = (orc_record *)out_points_error;
  a.out_r = out_residuals;
  a.out_valid = out_valid;
  return warp_residuals(&a);
}

/* ------------------------------------------------------------------------------------------ */
/* t-distribution weights / scale / log-likelihood                                             */
/* ------------------------------------------------------------------------------------------ */

/* r^T P r as Eigen evaluates (r^T * P) * r for column-major P = {p00,p10,p01,p11} */
static inline float mahalanobis(const float *r, const float *P) {
  const float t0 = r[0] * P[0] + r[1] * P[1];
  const float t1 = r[0] * P[2] + r[1] * P[3];
  return t0 * r[0] + t1 * r[1];
}

/* computeWeightsSse, dense_tracking_impl.cpp:657-707 (mean is always zero, Q4) */
static void tdist_weights(const float *res, int n, const float *P, int rcp_mode, float *w) {
  const int n4 = n - (n % 4);
  for (int i = 0; i < n4; ++i) {
    const float d = mahalanobis(res + 2 * i, P);
    if (rcp_mode == ORC_RCP_SSE)
      w[i] = 7.0f * orc_host_rcp(5.0f + d); /* :700 */
    else
      w[i] = 7.0f / (5.0f + d);
  }
  for (int i = n4; i < n; ++i) /* computeWeight :640-644 */
    w[i] = (float)((2.0 + 5.0f) / (5.0f + mahalanobis(res + 2 * i, P)));
}

/* test instrumentation: the summation order of orc_match on this thread (orc_config.sum_mode); every other entry point
 * sums like the reference */
static __thread int g_sum_mode = ORC_SUM_REFERENCE;
static __thread int g_ll_guard = 0; /* orc_config.ll_guard of the orc_match running on this thread */
/* partial sums over this many consecutive points (ORC_SUM_BLOCKED: 256, ORC_SUM_BLOCKED_32: 32, ORC_SUM_BLOCKED_2048: 2048) */
static int sum_block(void) { return g_sum_mode == ORC_SUM_BLOCKED_32 ? 32 : g_sum_mode == ORC_SUM_BLOCKED_2048 ? 2048 : 256; }
static int sum_blocked(void) { return g_sum_mode >= ORC_SUM_BLOCKED; }

/* computeScaleSse, dense_tracking_impl.cpp:590-638, including Q5 (first residual of a pair used twice) */
static void tdist_scale(const float *res, int n, const float *w, float cov[4], int clean) {
  const int n2 = n - (n % 2);
  const float scale = 1.0f / (float)(size_t)(n - 2 - 1);
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  double acc64[4] = {0.0, 0.0, 0.0, 0.0};
  float blk[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int i = 0; i < n2; i += 2) {
    const float x = res[2 * i], y = res[2 * i + 1];
    const float f[4] = {x * x, y * x, x * y, y * y}; /* fac1*fac2, :608-612 */
    /* ORC_CLEAN: the second residual of the pair contributes its own outer product (what the code was meant to do) */
    const float x2 = clean ? res[2 * i + 2] : x, y2 = clean ? res[2 * i + 3] : y;
    const float g[4] = {x2 * x2, y2 * x2, x2 * y2, y2 * y2};
    for (int k = 0; k < 4; ++k) {
      const float p1 = scale * (w[i] * f[k]);
      const float p2 = scale * (w[i + 1] * g[k]);
      if (g_sum_mode == ORC_SUM_FP64)
        acc64[k] += (double)(p1 + p2);
      else if (sum_blocked())
        blk[k] = blk[k] + (p1 + p2);
      else
        acc[k] = acc[k] + (p1 + p2);
    }
    if (sum_blocked() && ((i + 2) % sum_block() == 0 || i + 2 >= n2))
      for (int k = 0; k < 4; ++k) acc[k] += blk[k], blk[k] = 0.0f;
  }
  if (g_sum_mode == ORC_SUM_FP64)
    for (int k = 0; k < 4; ++k) acc[k] = (float)acc64[k];
  cov[0] = acc[0];
  cov[1] = acc[1]; /* (1,0) */
  cov[2] = acc[1]; /* (0,1) */
  cov[3] = acc[3];
  for (int i = n2; i < n; ++i) { /* computeScalePart :566-572 */
    const float x = res[2 * i], y = res[2 * i + 1];
    const float wx = w[i] * x, wy = w[i] * y;
    cov[0] += scale * (wx * x);
    cov[1] += scale * (wy * x);
    cov[2] += scale * (wx * y);
    cov[3] += scale * (wy * y);
  }
}

/* Eigen 3 Matrix2f::inverse (compute_inverse_size2_helper): column-major in/out */
static void inverse2f(const float m[4], float r[4]) {
  const float det = m[0] * m[3] - m[1] * m[2];
  const float invdet = 1.0f / det;
  r[0] = m[3] * invdet;
  r[1] = -m[1] * invdet;
  r[2] = -m[2] * invdet;
  r[3] = m[0] * invdet;
}

/* computeCompleteDataLogLikelihood, dense_tracking_impl.cpp:406-425, including Q6 (tail of n % 50 dropped) */
static float tdist_loglik(const float *res, int n, const float *P, int clean) {
  size_t c = 1;
  double error_sum = 0.0, error_acc = 1.0;
  for (int i = 0; i < n; ++i, ++c) {
    error_acc *= (1.0 + 0.2 * mahalanobis(res + 2 * i, P));
    if (g_ll_guard && error_acc > 1e200) { /* instrumentation only: same sum, no overflow (orc_config.ll_guard) */
      error_sum += log(error_acc);
      error_acc = 1.0;
    }
    if ((c % 50) == 0) {
      error_sum += log(error_acc);
      error_acc = 1.0;
    }
  }
  if (clean) error_sum += log(error_acc); /* ORC_CLEAN: the last n % 50 residuals count too */
  const float det = P[0] * P[3] - P[1] * P[2];
  return (float)(0.5 * (size_t)n * logf(det) - 0.5 * (5.0 + 2.0) * error_sum);
}

float orc_weights_scale_loglik(const float *residuals, int n, const float *prec_in, int unit_weights, int rcp_mode,
                               float *weights_out, float *scale_out, float *prec_out) {
  float *w = weights_out ? weights_out : (float *)xalloc((size_t)n * sizeof(float));
  if (unit_weights)
    for (int i = 0; i < n; ++i) w[i] = 1.0f;
  else
    tdist_weights(residuals, n, prec_in, rcp_mode, w);
  float cov[4], P[4];
  const int clean = rcp_mode == ORC_RCP_CLEAN;
  tdist_scale(residuals, n, w, cov, clean);
  inverse2f(cov, P);
  if (scale_out) memcpy(scale_out, cov, sizeof(cov));
  if (prec_out) memcpy(prec_out, P, sizeof(P));
  const float ll = tdist_loglik(residuals, n, P, clean);
  if (!weights_out) free(w);
  return ll;
}

/* ------------------------------------------------------------------------------------------ */
/* Jacobians and the packed normal-equation accumulator                                        */
/* ------------------------------------------------------------------------------------------ */

/* computeJacobianOfProjectionAndTransformation / compute3rdRowOfJacobianOfTransformation, dense_tracking.cpp:448-476 */
void orc_jacobian(const float p[3], float Jw[12], float Jz[6]) {
  const float z = 1.0f / p[2];
  const float z_sqr = 1.0f / (p[2] * p[2]);
  Jw[0] = z;
  Jw[1] = 0.0f;
  Jw[2] = -p[0] * z_sqr;
  Jw[3] = Jw[2] * p[1];
  Jw[4] = 1.0f - Jw[2] * p[0];
  Jw[5] = -p[1] * z;
  Jw[6] = 0.0f;
  Jw[7] = z;
  Jw[8] = -p[1] * z_sqr;
  Jw[9] = -1.0f + Jw[8] * p[1];
  Jw[10] = -Jw[3];
  Jw[11] = p[0] * z;
  Jz[0] = 0.0f;
  Jz[1] = 0.0f;
  Jz[2] = 1.0f;
  Jz[3] = p[1];
  Jz[4] = -p[0];
  Jz[5] = 0.0f;
}

/* J (column-major 2x6: {1a,1b,2a,2b,...}) from one points_error record, dense_tracking.cpp:333-339 */
static inline void point_jacobian(const orc_record *pe, float J[12]) {
  float Jw[12], Jz[6];
  orc_jacobian(pe->p, Jw, Jz);
  const float gi0 = pe->e[2], gi1 = pe->e[3], gz0 = pe->e[4], gz1 = pe->e[5];
  for (int k = 0; k < 6; ++k) {
    J[2 * k] = gi0 * Jw[k] + gi1 * Jw[6 + k];
    J[2 * k + 1] = (gz0 * Jw[k] + gz1 * Jw[6 + k]) - Jz[k];
  }
}

/*
 * OptimizedSelfAdjointMatrix6x6f::rankUpdate(2x6, 2x2), math_sse.cpp:82-178.
 * acc holds the six upper 2x2 blocks (0,0)(0,1)(0,2)(1,1)(1,2)(2,2), each row-major, 24 floats.
 * Per block entry: acc += (ua_i*va_j + ub_i*vb_j) with u = alpha^T-weighted rows; one add into the accumulator.
 */
static inline void rank_update_2x6(float acc[24], const float J[12], const float al[4]) {
  const __m128 a1313 = _mm_setr_ps(al[0], al[1], al[0], al[1]);
  const __m128 a2424 = _mm_setr_ps(al[2], al[3], al[2], al[3]);
  __m128 v[3], ua[3], ub[3];
  for (int c = 0; c < 3; ++c) {
    v[c] = _mm_loadu_ps(J + 4 * c); /* {ka, kb, (k+1)a, (k+1)b} */
    const __m128 u = _mm_hadd_ps(_mm_mul_ps(v[c], a1313), _mm_mul_ps(v[c], a2424)); /* {ka',(k+1)a',kb',(k+1)b'} */
    ua[c] = _mm_shuffle_ps(u, u, _MM_SHUFFLE(2, 0, 2, 0));                          /* {ka',kb',ka',kb'} */
    ub[c] = _mm_shuffle_ps(u, u, _MM_SHUFFLE(3, 1, 3, 1));                          /* {(k+1)a',(k+1)b',...} */
  }
  int blk = 0;
  for (int r = 0; r < 3; ++r)
    for (int c = r; c < 3; ++c, ++blk) {
      const __m128 b = _mm_hadd_ps(_mm_mul_ps(ua[r], v[c]), _mm_mul_ps(ub[r], v[c]));
      _mm_storeu_ps(acc + 4 * blk, _mm_add_ps(_mm_loadu_ps(acc + 4 * blk), b));
    }
}

/* OptimizedSelfAdjointMatrix6x6f::toEigen, math_sse.cpp:190-207 (upper triangle mirrored), column-major out */
static void packed_to_dense(const float acc[24], float A[36]) {
  float tmp[36];
  memset(tmp, 0, sizeof(tmp));
  int idx = 0;
  for (int i = 0; i < 6; i += 2)
    for (int j = i; j < 6; j += 2) {
      tmp[j * 6 + i] = acc[idx++];
      tmp[(j + 1) * 6 + i] = acc[idx++];
      tmp[j * 6 + i + 1] = acc[idx++];
      tmp[(j + 1) * 6 + i + 1] = acc[idx++];
    }
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      const int r = i < j ? i : j, c = i < j ? j : i; /* take the upper element */
      A[j * 6 + i] = tmp[c * 6 + r];
    }
}

void orc_rank_update(const float *J, const float *alpha, int n, float *A36) {
  float acc[24];
  memset(acc, 0, sizeof(acc));
  for (int i = 0; i < n; ++i) rank_update_2x6(acc, J + 12 * i, alpha + 4 * i);
  packed_to_dense(acc, A36);
}

/* the point loop of dense_tracking.cpp:327-342: ls.update(J, residual, weight * precision) for every valid point, i.e.
 * A += J^T (w P) J through the packed accumulator (math_sse.cpp:82-178) and b -= (J^T (w P)) r (least_squares.cpp:58-64),
 * sequential fp32 accumulation in scan order */
static void normal_equations(const orc_record *points_error, int n, const float *weights, const float precision[4],
                             float acc[24], float bvec[6]) {
  memset(acc, 0, 24 * sizeof(float));
  memset(bvec, 0, 6 * sizeof(float));
  if (g_sum_mode != ORC_SUM_REFERENCE) { /* the same per-point terms, added up in another order (ORC_SUM_*) */
    double acc64[24] = {0}, b64[6] = {0};
    float blk[24] = {0}, bblk[6] = {0};
    for (int i = 0; i < n; ++i) {
      float J[12], W[4], term[24] = {0}, bt[6];
      point_jacobian(points_error + i, J);
      for (int k = 0; k < 4; ++k) W[k] = weights[i] * precision[k];
      rank_update_2x6(term, J, W);
      const float r0 = points_error[i].e[0], r1 = points_error[i].e[1];
      for (int k = 0; k < 6; ++k) {
        const float t0 = J[2 * k] * W[0] + J[2 * k + 1] * W[1];
        const float t1 = J[2 * k] * W[2] + J[2 * k + 1] * W[3];
        bt[k] = t0 * r0 + t1 * r1;
      }
      if (g_sum_mode == ORC_SUM_FP64) {
        for (int k = 0; k < 24; ++k) acc64[k] += (double)term[k];
        for (int k = 0; k < 6; ++k) b64[k] -= (double)bt[k];
      } else {
        for (int k = 0; k < 24; ++k) blk[k] += term[k];
        for (int k = 0; k < 6; ++k) bblk[k] -= bt[k];
        if ((i + 1) % sum_block() == 0 || i + 1 == n) {
          for (int k = 0; k < 24; ++k) acc[k] += blk[k], blk[k] = 0.0f;
          for (int k = 0; k < 6; ++k) bvec[k] += bblk[k], bblk[k] = 0.0f;
        }
      }
    }
    if (g_sum_mode == ORC_SUM_FP64) {
      for (int k = 0; k < 24; ++k) acc[k] = (float)acc64[k];
      for (int k = 0; k < 6; ++k) bvec[k] = (float)b64[k];
    }
    return;
  }
  for (int i = 0; i < n; ++i) {
    float J[12], W[4];
    point_jacobian(points_error + i, J);
    for (int k = 0; k < 4; ++k) W[k] = weights[i] * precision[k];
    rank_update_2x6(acc, J, W);
    /* b -= (J^T * W) * r, least_squares.cpp:58-64 */
    const float r0 = points_error[i].e[0], r1 = points_error[i].e[1];
    for (int k = 0; k < 6; ++k) {
      const float t0 = J[2 * k] * W[0] + J[2 * k + 1] * W[1];
      const float t1 = J[2 * k] * W[2] + J[2 * k + 1] * W[3];
      bvec[k] -= t0 * r0 + t1 * r1;
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* SE(3) in double: Sophus SE3d (unit quaternion + translation), tangent order (upsilon, omega) */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
  double q[4]; /* w x y z */
  double t[3];
} se3;

static const double SOPHUS_EPS = 1e-10;

static void quat_mul(const double a[4], const double b[4], double r[4]) {
  const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  const double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  const double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  const double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w;
  r[1] = x;
  r[2] = y;
  r[3] = z;
}

static void quat_normalize(double q[4]) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] /= n;
}

static void quat_to_rot(const double q[4], double R[9] /* row-major */) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1 - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1 - (txx + tyy);
}

static void rot_to_quat(const double R[9], double q[4]) {
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double t = sqrt(tr + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (R[7] - R[5]) * t;
    q[2] = (R[2] - R[6]) * t;
    q[3] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    q[1 + i] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
  quat_normalize(q);
}

static void quat_rotate(const double q[4], const double v[3], double r[3]) {
  double R[9];
  quat_to_rot(q, R);
  for (int i = 0; i < 3; ++i) r[i] = R[i * 3] * v[0] + R[i * 3 + 1] * v[1] + R[i * 3 + 2] * v[2];
}

static void se3_identity(se3 *a) {
  a->q[0] = 1;
  a->q[1] = a->q[2] = a->q[3] = 0;
  a->t[0] = a->t[1] = a->t[2] = 0;
}

static void se3_from_matrix(const double T[16], se3 *a) {
  double R[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = T[j * 4 + i];
  rot_to_quat(R, a->q);
  for (int i = 0; i < 3; ++i) a->t[i] = T[12 + i];
}

static void se3_to_matrix(const se3 *a, double T[16]) {
  double R[9];
  quat_to_rot(a->q, R);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[j * 4 + i] = R[i * 3 + j];
    T[12 + i] = a->t[i];
    T[i * 4 + 3] = 0.0;
  }
  T[15] = 1.0;
}

static void se3_mul(const se3 *a, const se3 *b, se3 *r) {
  se3 o;
  quat_mul(a->q, b->q, o.q);
  quat_normalize(o.q);
  double rt[3];
  quat_rotate(a->q, b->t, rt);
  for (int i = 0; i < 3; ++i) o.t[i] = a->t[i] + rt[i];
  *r = o;
}

static void se3_inverse(const se3 *a, se3 *r) {
  se3 o;
  o.q[0] = a->q[0];
  o.q[1] = -a->q[1];
  o.q[2] = -a->q[2];
  o.q[3] = -a->q[3];
  const double nt[3] = {-a->t[0], -a->t[1], -a->t[2]};
  quat_rotate(o.q, nt, o.t);
  *r = o;
}

static void hat_sq(const double w[3], double O[9], double O2[9]) {
  O[0] = 0;
  O[1] = -w[2];
  O[2] = w[1];
  O[3] = w[2];
  O[4] = 0;
  O[5] = -w[0];
  O[6] = -w[1];
  O[7] = w[0];
  O[8] = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += O[i * 3 + k] * O[k * 3 + j];
      O2[i * 3 + j] = s;
    }
}

/* Sophus SE3Group::exp with SO3Group::expAndTheta */
static void se3_exp(const double xi[6], se3 *r) {
  const double *ups = xi, *om = xi + 3;
  const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  const double half = 0.5 * theta;
  double imag, real;
  if (theta < SOPHUS_EPS) {
    const double t2 = theta * theta, t4 = t2 * t2;
    imag = 0.5 - t2 / 48.0 + t4 / 3840.0;
    real = 1.0 - t2 / 8.0 + t4 / 384.0;
  } else {
    imag = sin(half) / theta;
    real = cos(half);
  }
  r->q[0] = real;
  r->q[1] = imag * om[0];
  r->q[2] = imag * om[1];
  r->q[3] = imag * om[2];
  double O[9], O2[9], V[9];
  hat_sq(om, O, O2);
  if (theta < SOPHUS_EPS) {
    quat_to_rot(r->q, V);
  } else {
    const double t2 = theta * theta;
    const double a = (1.0 - cos(theta)) / t2, b = (theta - sin(theta)) / (t2 * theta);
    for (int i = 0; i < 9; ++i) V[i] = a * O[i] + b * O2[i];
    V[0] += 1.0;
    V[4] += 1.0;
    V[8] += 1.0;
  }
  for (int i = 0; i < 3; ++i) r->t[i] = V[i * 3] * ups[0] + V[i * 3 + 1] * ups[1] + V[i * 3 + 2] * ups[2];
}

/* Sophus SE3Group::log with SO3Group::logAndTheta */
static void se3_log(const se3 *a, double xi[6]) {
  const double *q = a->q;
  const double n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const double n = sqrt(n2), w = q[0];
  double two_atan;
  if (n < SOPHUS_EPS) {
    two_atan = 2.0 / w - 2.0 * n2 / (w * w * w);
  } else if (fabs(w) < SOPHUS_EPS) {
    two_atan = (w > 0 ? M_PI : -M_PI) / n;
  } else {
    two_atan = 2.0 * atan(n / w) / n;
  }
  const double theta = two_atan * n;
  double om[3] = {two_atan * q[1], two_atan * q[2], two_atan * q[3]};
  double O[9], O2[9], Vi[9];
  hat_sq(om, O, O2);
  double c;
  if (fabs(theta) < SOPHUS_EPS) {
    c = 1.0 / 12.0;
  } else {
    const double half = 0.5 * theta;
    c = (1.0 - theta * cos(half) / (2.0 * sin(half))) / (theta * theta);
  }
  for (int i = 0; i < 9; ++i) Vi[i] = -0.5 * O[i] + c * O2[i];
  Vi[0] += 1.0;
  Vi[4] += 1.0;
  Vi[8] += 1.0;
  for (int i = 0; i < 3; ++i) {
    xi[i] = Vi[i * 3] * a->t[0] + Vi[i * 3 + 1] * a->t[1] + Vi[i * 3 + 2] * a->t[2];
    xi[3 + i] = om[i];
  }
}

void orc_se3_exp(const double xi[6], double T[16]) {
  se3 a;
  se3_exp(xi, &a);
  se3_to_matrix(&a, T);
}

void orc_se3_log(const double T[16], double xi[6]) {
  se3 a;
  se3_from_matrix(T, &a);
  se3_log(&a, xi);
}

/* Eigen LDLT<Matrix6d>::solve: pivoted (largest remaining diagonal) L D L^T, column-major A */
static void ldlt6_solve(const double Ain[36], const double bin[6], double x[6]) {
  double A[36];
  int perm[6];
  memcpy(A, Ain, sizeof(A));
  for (int i = 0; i < 6; ++i) perm[i] = i;
#define AT(i, j) A[(j) * 6 + (i)]
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    double best = fabs(AT(k, k));
    for (int i = k + 1; i < 6; ++i)
      if (fabs(AT(i, i)) > best) {
        best = fabs(AT(i, i));
        piv = i;
      }
    if (piv != k) { /* symmetric row/column swap */
      for (int j = 0; j < 6; ++j) {
        double t = AT(k, j);
        AT(k, j) = AT(piv, j);
        AT(piv, j) = t;
      }
      for (int i = 0; i < 6; ++i) {
        double t = AT(i, k);
        AT(i, k) = AT(i, piv);
        AT(i, piv) = t;
      }
      int t = perm[k];
      perm[k] = perm[piv];
      perm[piv] = t;
    }
    const double d = AT(k, k);
    if (d == 0.0) continue;
    for (int i = k + 1; i < 6; ++i) AT(i, k) /= d;
    for (int j = k + 1; j < 6; ++j)
      for (int i = j; i < 6; ++i) {
        AT(i, j) -= AT(i, k) * d * AT(j, k);
        AT(j, i) = AT(i, j);
      }
  }
  double y[6];
  for (int i = 0; i < 6; ++i) y[i] = bin[perm[i]];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < i; ++j) y[i] -= AT(i, j) * y[j];
  for (int i = 0; i < 6; ++i) {
    const double d = AT(i, i);
    y[i] = (fabs(d) > DBL_MIN) ? y[i] / d : 0.0;
  }
  for (int i = 5; i >= 0; --i)
    for (int j = i + 1; j < 6; ++j) y[i] -= AT(j, i) * y[j];
  for (int i = 0; i < 6; ++i) x[perm[i]] = y[i];
#undef AT
}

/* ------------------------------------------------------------------------------------------ */
/* the Gauss-Newton driver: DenseTracker::match, dense_tracking.cpp:131-376                    */
/* ------------------------------------------------------------------------------------------ */

static double inf_norm6(const double x[6]) {
  /* Eigen lpNorm<Infinity> = cwiseAbs().maxCoeff(): a NaN never wins the max */
  double m = fabs(x[0]);
  for (int i = 1; i < 6; ++i)
    if (fabs(x[i]) > m) m = fabs(x[i]);
  return m;
}

/* The driver proper.  `from` == NULL: DenseTracker::match from its beginning.  Otherwise the loop is entered at the top of the
 * iteration body (:259) of level from->level with the state a run of the reference would hold there -- see orc_match_state in
 * dvo_oracle.h.  Nothing else differs: a continuation from a state this function itself passed through reproduces the rest of
 * that run bit for bit (tests/test_oracle.py::test_continuation_reproduces_the_rest_of_a_match). */
static int match_run(const orc_config *cfg, orc_pyramid *ref, orc_pyramid *cur, const double *T_init, const orc_match_state *from,
                     orc_result *res) {
  const int n_levels_needed = cfg->first_level + 1;
  if (cfg->first_level < cfg->last_level || cfg->last_level < 0) return -1;
  if (ref->n_levels < n_levels_needed || cur->n_levels < n_levels_needed) return -2;
  if (from && (from->level > cfg->first_level || from->level < cfg->last_level || from->iteration < 0 ||
               from->iteration >= cfg->max_iterations_per_level))
    return -4;

  double nan = NAN;
  res->n_levels = 0;
  res->n_iterations = 0;
  res->is_nan = 0;

  /* :137-150 */
  se3 inc, initial, initial_old, estimate, estimate_old;
  if (from) {
    /* `inc` is overwritten by exp(x) before it is read (:259); initial / estimate are the Revertables' current values */
    se3_identity(&inc);
    se3_from_matrix(from->initial, &initial);
    se3_from_matrix(from->estimate, &estimate);
    initial_old = initial;
    estimate_old = estimate;
  } else {
    if (cfg->use_initial_estimate && T_init)
      se3_from_matrix(T_init, &inc);
    else
      se3_identity(&inc);
    initial = inc;
    initial_old = inc;
    se3_identity(&estimate);
    se3_identity(&estimate_old);
  }

  const size_t max_pts = (size_t)ref->lv[cfg->last_level].w * ref->lv[cfg->last_level].h;
  orc_record *points_error = (orc_record *)xalloc(max_pts * sizeof(orc_record));
  float *residuals = (float *)xalloc(max_pts * 2 * sizeof(float));
  float *weights = (float *)xalloc(max_pts * sizeof(float));

  float precision[4] = {0, 0, 0, 0};
  double x[6];
  /* iteration log: the caller's buffer, or a private one (the final result needs the last two entries) */
  const int its_needed = (cfg->first_level - cfg->last_level + 1) * (cfg->max_iterations_per_level + 1);
  orc_iteration_stats *its = res->iterations;
  int its_cap = res->iterations_capacity, its_own = 0;
  if (!its) {
    its = (orc_iteration_stats *)calloc((size_t)its_needed, sizeof(orc_iteration_stats));
    its_cap = its_needed;
    its_own = 1;
  } else if (its_cap < its_needed) {
    free(points_error);
    free(residuals);
    free(weights);
    return -3;
  }
  int last_level_first_iter = 0;
  g_sum_mode = cfg->sum_mode; /* (reset before the single return below) */
  g_ll_guard = cfg->ll_guard;

  for (int level = from ? from->level : cfg->first_level; level >= cfg->last_level; --level) {
    orc_level_stats *ls = &res->levels[res->n_levels++];
    memset(precision, 0, sizeof(precision));
    int iteration = 0;
    double error = DBL_MAX, last_error = DBL_MAX;
    const int resumed = from && level == from->level;
    if (resumed) { /* the level's own state: Iteration (:354), Error (:304), the precision the next weights use (:291) */
      iteration = from->iteration;
      error = from->last_error;
      memcpy(precision, from->precision, sizeof(precision));
    }

    orc_level *C = &cur->lv[level];
    orc_level *R = &ref->lv[level];
    warp_args wa;
    make_weights8(C, wa.wref, wa.wcur);
    level_select(R, cfg->intensity_derivative_threshold, cfg->depth_derivative_threshold);

    ls->id = level;
    /* getMaximumNumberOfPoints, point_selection.cpp:68-71 */
    ls->max_valid_pixels = (int)(size_t)((double)((size_t)ref->lv[0].w * ref->lv[0].h) * pow(0.25, (double)level));
    ls->valid_pixels = R->n_sel;
    ls->termination = ORC_TERM_UNSET;
    ls->n_iterations = 0;
    ls->first_iteration = res->n_iterations;
    last_level_first_iter = res->n_iterations;

    if (resumed)
      memcpy(x, from->x, sizeof(x)); /* the increment the resumed iteration applies (at a level start: the caller's log(inc)) */
    else
      se3_log(&inc, x); /* :238, Q1 */
    int accept = 1;

    do {
      orc_iteration_stats *it = &its[res->n_iterations];
      res->n_iterations++;
      ls->n_iterations++;
      memset(it, 0, sizeof(*it));
      it->id = iteration;

      /* :259-263 */
      se3_exp(x, &inc);
      se3 inc_inv, tmp;
      se3_inverse(&inc, &inc_inv);
      initial_old = initial;
      se3_mul(&inc_inv, &initial, &tmp);
      initial = tmp;
      estimate_old = estimate;
      se3_mul(&inc, &estimate, &tmp);
      estimate = tmp;

      double Td[16];
      float Tf[16];
      se3_to_matrix(&estimate, Td);
      for (int i = 0; i < 16; ++i) Tf[i] = (float)Td[i];
      memcpy(it->estimate, Td, sizeof(Td));
      se3_to_matrix(&initial, it->initial);

      wa.first = R->sel;
      wa.n_sel = R->n_sel;
      wa.accel = C->accel;
      wa.w = C->w;
      wa.h = C->h;
      make_kt(C, Tf, wa.kt);
      wa.rcp_mode = cfg->rcp_mode;
      wa.out_pe = points_error;
      wa.out_r = residuals;
      wa.out_valid = NULL;
      const int n = warp_residuals(&wa);
      it->valid_constraints = n;

      if (n < 6) { /* :276-284 */
        initial = initial_old;
        estimate = estimate_old;
        ls->termination = ORC_TERM_TOO_FEW_CONSTRAINTS;
        break;
      }

      if (iteration == 0) /* :286-293, Q2 */
        for (int i = 0; i < n; ++i) weights[i] = 1.0f;
      else
        tdist_weights(residuals, n, precision, cfg->rcp_mode, weights);

      float cov[4];
      tdist_scale(residuals, n, weights, cov, cfg->rcp_mode == ORC_RCP_CLEAN);
      inverse2f(cov, precision); /* :295 */
      const float ll = tdist_loglik(residuals, n, precision, cfg->rcp_mode == ORC_RCP_CLEAN);

      double xi_initial[6];
      se3_log(&initial, xi_initial);
      double sq = 0;
      for (int i = 0; i < 6; ++i) sq += xi_initial[i] * xi_initial[i];
      it->tdist_loglik = -ll;
      it->tdist_mean[0] = it->tdist_mean[1] = 0.0;
      for (int i = 0; i < 4; ++i) it->tdist_precision[i] = precision[i];
      memcpy(it->scale, cov, sizeof(cov));
      it->prior_loglik = cfg->mu * sq;

      last_error = error;
      error = -ll;
      accept = error < last_error; /* :312 */
      if (!accept) {
        initial = initial_old;
        estimate = estimate_old;
        ls->termination = ORC_TERM_LOGLIKELIHOOD_DECREASED;
        break;
      }

      /* :327-347 normal equations */
      float acc[24], bvec[6];
      normal_equations(points_error, n, weights, precision, acc, bvec);
      float Af[36];
      packed_to_dense(acc, Af);
      double A[36], b[6];
      for (int i = 0; i < 36; ++i) A[i] = (double)Af[i];
      for (int i = 0; i < 6; ++i) {
        A[i * 6 + i] += cfg->mu;
        b[i] = (double)bvec[i] + cfg->mu * xi_initial[i];
      }
      ldlt6_solve(A, b, x);

      memcpy(it->increment, x, sizeof(double) * 6);
      memcpy(it->information, A, sizeof(A));
      memcpy(it->rhs, b, sizeof(b));
      it->has_increment = 1;
      iteration++;
    } while (accept && inf_norm6(x) > cfg->precision && !(iteration >= cfg->max_iterations_per_level));

    /* :359-363: evaluated after a break as well, with whatever x holds */
    if (inf_norm6(x) <= cfg->precision) ls->termination = ORC_TERM_INCREMENT_TOO_SMALL;
    if (iteration >= cfg->max_iterations_per_level) ls->termination = ORC_TERM_ITERATIONS_EXCEEDED;
  }

  /* :368-373 */
  orc_level_stats *last = &res->levels[res->n_levels - 1];
  int li = last->termination != ORC_TERM_LOGLIKELIHOOD_DECREASED ? last->n_iterations - 1 : last->n_iterations - 2;
  se3 est_inv;
  se3_inverse(&estimate, &est_inv);
  se3_to_matrix(&est_inv, res->T);
  const orc_iteration_stats *lit = li >= 0 ? &its[last_level_first_iter + li] : NULL;
  if (lit && lit->has_increment) {
    for (int i = 0; i < 36; ++i) res->information[i] = lit->information[i] * 0.008 * 0.008;
    res->loglik = lit->tdist_loglik + lit->prior_loglik;
  } else if (li == -1 && from && from->level == cfg->last_level && from->iteration > 0 && from->has_previous) {
    /* resumed inside the last level and rejected at once: the statistics entry :369-372 reads is the one before the resume */
    for (int i = 0; i < 36; ++i) res->information[i] = from->previous_information[i] * 0.008 * 0.008;
    res->loglik = from->previous_loglik;
  } else {
    /* the reference reads an uninitialised / out-of-range IterationStats here (Q9): report NaN */
    for (int i = 0; i < 36; ++i) res->information[i] = nan;
    res->loglik = nan;
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += res->T[i];
  double si = 0;
  for (int i = 0; i < 36; ++i) si += res->information[i];
  res->is_nan = !(isfinite(s) && isfinite(si));

  g_sum_mode = ORC_SUM_REFERENCE;
  g_ll_guard = 0;
  if (its_own) free(its);
  (void)its_cap;
  free(points_error);
  free(residuals);
  free(weights);
  return 0;
}

int orc_match(const orc_config *cfg, orc_pyramid *ref, orc_pyramid *cur, const double *T_init, orc_result *res) {
  return match_run(cfg, ref, cur, T_init, NULL, res);
}

int orc_match_from(const orc_config *cfg, orc_pyramid *ref, orc_pyramid *cur, const orc_match_state *from, orc_result *res) {
  if (!from) return -4;
  return match_run(cfg, ref, cur, NULL, from, res);
}

/* One Gauss-Newton iteration body at a FIXED pose and a FIXED previous precision (dense_tracking.cpp:271-347 without the
 * accept test and the solve): residuals, weights (unit, or t-distribution from prec_in), scale, precision, log-likelihood,
 * normal equations.  For the stage-wise parity tests of the weighted iterations (k >= 1). */
int orc_iteration(orc_pyramid *ref, orc_pyramid *cur, int level, float ti, float td, const float *T, const float prec_in[4],
                  int unit_weights, int rcp_mode, float scale_out[4], float prec_out[4], float *ll_out, float A36[36],
                  float b6[6]) {
  orc_level *R = &ref->lv[level];
  orc_level *C = &cur->lv[level];
  level_select(R, ti, td);
  const size_t max_pts = (size_t)R->w * R->h;
  orc_record *points_error = (orc_record *)xalloc((max_pts + 1) * sizeof(orc_record));
  float *residuals = (float *)xalloc((max_pts + 1) * 2 * sizeof(float));
  float *weights = (float *)xalloc((max_pts + 1) * sizeof(float));
  warp_args wa;
  wa.first = R->sel;
  wa.n_sel = R->n_sel;
  wa.accel = C->accel;
  wa.w = C->w;
  wa.h = C->h;
  make_kt(C, T, wa.kt);
  make_weights8(C, wa.wref, wa.wcur);
  wa.rcp_mode = rcp_mode;
  wa.out_pe = points_error;
  wa.out_r = residuals;
  wa.out_valid = NULL;
  const int n = warp_residuals(&wa);
  if (n >= 6) {
    if (unit_weights)
      for (int i = 0; i < n; ++i) weights[i] = 1.0f;
    else
      tdist_weights(residuals, n, prec_in, rcp_mode, weights);
    float cov[4], P[4], acc[24];
    tdist_scale(residuals, n, weights, cov, rcp_mode == ORC_RCP_CLEAN);
    inverse2f(cov, P);
    if (scale_out) memcpy(scale_out, cov, sizeof(cov));
    if (prec_out) memcpy(prec_out, P, sizeof(P));
    if (ll_out) *ll_out = tdist_loglik(residuals, n, P, rcp_mode == ORC_RCP_CLEAN);
    if (A36 && b6) {
      normal_equations(points_error, n, weights, P, acc, b6);
      packed_to_dense(acc, A36);
    }
  }
  free(points_error);
  free(residuals);
  free(weights);
  return n;
}

/* ------------------------------------------------------------------------------------------------------------------
 * throughput of the restated path on many host threads (bench.py's cpu_baseline_many_threads leg): one tracker per thread over
 * independent pairs -- the shape of tbb::parallel_reduce over proposals (keyframe_graph.cpp:587-590) -- driven from C so that no
 * interpreter lock sits between two match() calls (round 4's Python-threaded leg scaled 10.6x on 64 threads: most of a call was
 * the binding's own result marshalling under the GIL).  The pyramids are shared read-only (the reference selection of `ref` must
 * have been built: run one orc_match first).
 * ------------------------------------------------------------------------------------------------------------------ */
#include <pthread.h>
#include <time.h>
typedef struct {
  const orc_config *cfg;
  orc_pyramid *ref;
  orc_pyramid *const *curs;
  int n_curs, index;
  double seconds;
  long long done;
} bench_arg;
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void *bench_worker(void *p) {
  bench_arg *a = (bench_arg *)p;
  const double t0 = now_s();
  int k = a->index;
  while (now_s() - t0 < a->seconds) {
    orc_result res;
    res.iterations = NULL, res.iterations_capacity = 0;
    if (orc_match(a->cfg, a->ref, a->curs[k % a->n_curs], NULL, &res) != 0) break;
    ++k, ++a->done;
  }
  return NULL;
}
long long orc_bench_threads(const orc_config *cfg, orc_pyramid *ref, orc_pyramid *const *curs, int n_curs, int n_threads,
                            double seconds, double *elapsed_s) {
  if (n_threads < 1 || n_threads > 1024 || n_curs < 1) return -1;
  pthread_t th[1024];
  static bench_arg args[1024];
  const double t0 = now_s();
  int started = 0;
  for (int t = 0; t < n_threads; ++t) {
    args[t].cfg = cfg, args[t].ref = ref, args[t].curs = curs, args[t].n_curs = n_curs, args[t].index = t, args[t].seconds = seconds;
    args[t].done = 0;
    if (pthread_create(&th[t], NULL, bench_worker, &args[t]) != 0) break;
    ++started;
  }
  long long total = 0;
  for (int t = 0; t < started; ++t) {
    pthread_join(th[t], NULL);
    total += args[t].done;
  }
  if (elapsed_s) *elapsed_s = now_s() - t0;
  return started == n_threads ? total : -2;
}

/* ------------------------------------------------------------------------------------------------------------------
 * frame ingest
 * ------------------------------------------------------------------------------------------------------------------ */

/* surface_pyramid.cpp:44-63 (scalar) and :65-105 (SSE: cvtepi32_ps, cmpeq 0 -> or NaN, mulps): same values */
void orc_ingest_depth_u16(const unsigned short *raw, int width, int height, int stride, float scale, float *out) {
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      unsigned short v = raw[(size_t)y * stride + x];
      out[(size_t)y * width + x] = v == 0 ? NAN : (float)v * scale;
    }
}

/* cv::cvtColor(CV_BGR2GRAY) for CV_8UC3 (OpenCV 2.4 fixed-point rule, see the header) + convertTo(CV_32F):
 * benchmark_slam.cpp:60-68, camera_dense_tracking.cpp:219-224 */
void orc_ingest_gray_from_bgr8(const unsigned char *bgr, int width, int height, int stride_bytes, float *out) {
  for (int y = 0; y < height; ++y) {
    const unsigned char *row = bgr + (size_t)y * stride_bytes;
    for (int x = 0; x < width; ++x) {
      int b = row[3 * x], g = row[3 * x + 1], r = row[3 * x + 2];
      out[(size_t)y * width + x] = (float)((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14);
    }
  }
}

/* single-channel input: convertTo(CV_32F) only (benchmark_slam.cpp:64-68, camera_dense_tracking.cpp:226-229) */
void orc_ingest_gray_from_gray8(const unsigned char *gray, int width, int height, int stride_bytes, float *out) {
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) out[(size_t)y * width + x] = (float)gray[(size_t)y * stride_bytes + x];
}
