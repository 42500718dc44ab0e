// HIP kernels (gfx950 / CDNA4) of the dense RGB-D alignment hot path.
//
//  k_tick      one launch per Gauss-Newton tick.  blockIdx.y selects a WorkItem:
//              - residual pass  = computeResidualsSse + computeWeightsSse + computeScaleSse + Jacobians +
//                NormalEquationsLeastSquares::update fused (dense_tracking_impl.cpp:133-393,590-707,
//                dense_tracking.cpp:333-342,448-476, math_sse.cpp:82-178): one read of the reference planes, one
//                bilinear gather of the current planes, residuals spilled once (8 B/px) for the log-likelihood;
//              - log-likelihood pass = computeCompleteDataLogLikelihood (dense_tracking_impl.cpp:406-425).
//  k_finalize  second-pass block reduce: ordered combine of the per-block records (fp64), one block per job.
//  prep kernels: pyramid down-sampling, derivatives, gather layout, point selection (rgbd_image.cpp, point_selection.cpp).
//
// Numerics: the reference evaluates the warp/residual stage in round-toward-zero (MXCSR, dense_tracking_impl.cpp:165-167)
// and everything else in round-to-nearest.  The residual section below switches MODE.FP_ROUND the same way, is compiled
// without fp contraction, and forms 1/z as the exactly truncated quotient, so every residual and every validity decision
// is bit-identical to the CPU restatement in oracle/ (rcp_mode = EXACT).  Sums are accumulated per thread in fp32 (fma),
// reduced across a wave with DPP, across waves through LDS and across blocks in fp64 by k_finalize, in a fixed order
// (deterministic run to run).
//
// This file is compiled with -ffp-contract=off: every a*b+c that may fuse is written as __builtin_fmaf explicitly.
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "dvo_types.h"

namespace dvo_amd {

// ------------------------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------------------------

// hwreg(HW_REG_MODE, offset 0, width 2) = single-precision rounding mode: 0 nearest-even, 3 toward zero
#define DVO_HWREG_MODE_FP32_ROUND (1 | (0 << 6) | ((2 - 1) << 11))
// Make a value opaque to the optimiser.  Every float that is live across a rounding-mode switch goes through this once
// before and once after the s_setreg, so no operation producing or consuming it can be scheduled on the wrong side.
#define DVO_OPAQUE(x) asm volatile("" : "+v"(x))
#ifndef DVO_KT_SGPR
#define DVO_KT_SGPR 1
#endif

__device__ __forceinline__ void round_toward_zero() { __builtin_amdgcn_s_setreg(DVO_HWREG_MODE_FP32_ROUND, 3); }
__device__ __forceinline__ void round_to_nearest() { __builtin_amdgcn_s_setreg(DVO_HWREG_MODE_FP32_ROUND, 0); }

// v_mul_legacy_f32: 0 * x = 0 for every x (NaN and infinity included); otherwise an ordinary IEEE multiply
__device__ __forceinline__ float mul_legacy(float a, float b) {
  float r;
  asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ float u2f(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned f2u(float f) { return __builtin_bit_cast(unsigned, f); }

// DPP move: returns 0 in lanes the row mask disables or whose source is invalid
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_read(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}

// Sum over the 64 lanes of a wave with DPP row operations; the total is valid in lane 63.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp_read<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_read<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_read<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_read<0x140, 0xF>(v);  // row_mirror: every lane holds its row's sum
  v += dpp_read<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  v += dpp_read<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
  return v;
}

__device__ __forceinline__ double wave_sum_double(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// Blocks b and b+8 share an XCD (round-robin dispatch); give each XCD one contiguous run of logical blocks so that
// neighbouring scan-order ranges (which gather overlapping rows of the current image) meet in the same 4 MiB L2.
// Bijective for any n_blocks; affects speed only.
__device__ __forceinline__ int xcd_contiguous_block(int b, int n_blocks) {
  const int q = n_blocks >> 3, r = n_blocks & 7;
  const int xcd = b & 7, pos = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
}

// trunc(1/z): the quotient rounded toward zero.  MUST run with MODE.FP_ROUND = toward zero (project_pixel_rtz does).
// r = v_rcp_f32(|z|) is within 1 ulp of rho = 1/|z|: r = rho (1 + d), |d| <= 2^-22.  One Newton step r' = r + r (1 - |z| r),
// its residual e = 1 - |z| r exact in one fma (|z| r is within 2^-22 of 1) and the step itself one fma ROUNDED TOWARD ZERO,
// evaluates V = rho (1 - d^2): below rho by at most rho 2^-44, far less than the spacing of floats there.  So r' = floor(V) is
// the truncated quotient T or its lower neighbour, and the sign of the exact residual 1 - |z| nextup(r') (again one fma: a
// non-zero value never rounds across zero) tells which.  NaN goes through as NaN (every compare is false, every candidate a
// NaN): an unselected reference pixel -- depth NaN by construction, i.e. most lanes of most steps -- costs nothing extra.
// Zero, denormal, huge and infinite |z| (never a pixel that ends up in bounds with a sane pose, but the result must still be
// the division's) take the IEEE division; until round 3 NaN lanes took it too, on three of four wave steps.
__device__ __forceinline__ float rcp_toward_zero(float z) {
  const float az = __builtin_fabsf(z);
  if (__builtin_expect(az < 1.1754944e-38f || az > 8.5070592e+37f, 0)) return 1.0f / z;  // (ordered compares: false for NaN)
  const float r = __builtin_amdgcn_rcpf(az);
  const float e = __builtin_fmaf(-az, r, 1.0f);
  const unsigned lo = f2u(__builtin_fmaf(e, r, r));  // T or T - 1 ulp (toward-zero rounding)
  const unsigned up = lo + 1u;
  const bool up_fits = __builtin_fmaf(-az, u2f(up), 1.0f) >= 0.0f;  // nextup(r') <= 1/|z|
  return __builtin_copysignf(u2f(up_fits ? up : lo), z);
}

// The host's _mm_rcp_ps (opt-in, dvo_amd_set_reciprocal_mode): rcpps(x) = rcpps(1.m) 2^-e exactly, rcpps(1.m) a function of the
// top mantissa bits only -- both probed on the host when the mode is switched on, together with the special cases: zero and
// denormal input give infinity, a result below the normal range is flushed to zero, infinity gives zero, NaN stays NaN (quiet).
__device__ __forceinline__ float rcp_host_table(float z, const RcpTable &rcp) {
  const unsigned u = f2u(z), au = u & 0x7fffffffu, e = au >> 23, m = au & 0x7fffffu;
  const unsigned t = ((const __attribute__((address_space(1))) unsigned *)rcp.table)[m >> rcp.shift];
  const int re = (int)(t >> 23) - ((int)e - 127);  // exponent field of the result
  unsigned r = re >= 1 ? (((unsigned)re << 23) | (t & 0x7fffffu)) : 0u;
  r = e == 0u ? 0x7f800000u : r;
  r = e == 255u ? (m ? (au | 0x00400000u) : 0u) : r;
  return u2f(r | (u & 0x80000000u));
}

// The nibble form (RcpTable::nibbles, round 5): no global-memory gather.  rcpps(1.m) is rebuilt from the device's own reciprocal of
// the midpoint of m's cell -- its top bits, a pure function of the cell index -- plus a signed 4-bit correction from a table
// in LDS (lds_nib: eight cells to a word); exponent and special cases as above.  Bit-identical to rcp_host_table by construction:
// the corrections are (table entry - that very expression) evaluated on this device when the mode is switched on, and the
// switch refuses the form if the expression differs between the two rounding modes the kernel uses it in.
__device__ __forceinline__ unsigned rcp_midpoint_bits(unsigned cell, int shift) {
  return f2u(__builtin_amdgcn_rcpf(u2f(0x3f800000u | (cell << shift) | (1u << (shift - 1)))));
}
// NAN_ANY: a NaN input may give ANY result (the residual pass: a NaN depth makes sx, sy NaN as well, so u, v are NaN and the pixel
// invalid whatever 1 / z is; most wave steps hold such lanes -- unselected pixels -- and must not send the wave down the rare path).
template <bool NAN_ANY>
__device__ __forceinline__ float rcp_host_nibbles(float z, const unsigned *lds_nib, int shift, int unit) {
  const unsigned u = f2u(z), au = u & 0x7fffffffu, e = au >> 23, m = au & 0x7fffffu;
  const unsigned cell = m >> shift;
  const unsigned word = lds_nib[cell >> 3];
  const int corr = (int)(((word >> ((cell & 7u) * 4u)) & 15u) ^ 8u) - 8;  // signed nibble
  const unsigned t = (rcp_midpoint_bits(cell, shift) & ~((1u << unit) - 1u)) + (unsigned)(corr << unit);
  // Common case, branch free: t's exponent field is 126 or 127, so for 1 <= e <= 252 the result's exponent stays a normal one and
  // scaling by 2^-(e - 127) is an integer subtraction on the bit pattern.  Zero / denormal inputs, results that would leave the
  // normal range and infinities / NaNs (e = 0 or e >= 253) take the full rule; the wave branches only if one of its lanes has one.
  unsigned r = t - ((e - 127u) << 23);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(e - 1u >= 252u && !(NAN_ANY && au > 0x7f800000u)) != 0ull, 0)) {
    const int re = (int)(t >> 23) - ((int)e - 127);
    r = re >= 1 ? (((unsigned)re << 23) | (t & 0x7fffffu)) : 0u;
    r = e == 0u ? 0x7f800000u : r;
    r = e == 255u ? (m ? (au | 0x00400000u) : 0u) : r;
  }
  return u2f(r | (u & 0x80000000u));
}

// Pointers read from a descriptor in memory are generic ("flat") to the compiler; flat loads are slower and cannot be
// counted separately from LDS traffic.  Everything the kernels touch lives in device global memory: say so.
#define DVO_GLOBAL __attribute__((address_space(1)))
typedef float v4f __attribute__((ext_vector_type(4)));               // plain vector types: loadable from any address space
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f_a8 __attribute__((ext_vector_type(4), aligned(8)));  // 16 bytes that are only 8-byte aligned
typedef const DVO_GLOBAL float *gcf;
// load at (uniform base) + (32-bit unsigned byte offset): selects the scalar-base + 32-bit vector-offset addressing
// mode of global_load, so no 64-bit address arithmetic is spent per access
template <class T, int IMM = 0>
__device__ __forceinline__ T ld_off(const DVO_GLOBAL void *base, unsigned byte_off) {
  return *reinterpret_cast<const DVO_GLOBAL T *>(reinterpret_cast<const DVO_GLOBAL char *>(base) + byte_off + IMM);
}
typedef const DVO_GLOBAL v4f *gcf4;


// what a block needs of the three descriptors, in registers
struct LevelPairDesc {
  const float *r_zsel, *r_i, *r_ix, *r_iy, *tx, *ty;
  const float4 *c_a;
  const float2 *c_b;
  float2 *res[2];
  float *records;
  double *ll_partials;
  float *ll_qmax;
  int *seg_prefix[2];
  int w, h;
  float wc[6], wr[4], ub_x, ub_y;
  RcpTable rcp;  // (only read by the RCP >= 1 kernels)
  const unsigned *rcp_lds;  // (RCP = 2) the block's LDS copy of rcp.nibbles
  float *dbg_w;  // (RCP >= 1 kernels, test instrumentation) SlotDesc::dbg_w
};

// Block-level trace (builds with -DDVO_TRACE_BLOCKS only: scripts/variant.sh trace -DDVO_TRACE_BLOCKS; never in the shipped
// library): every block of k_tick appends {start, after its first step, end} on the 100 MHz constant clock, what it ran and where
// (HW_ID, XCC_ID).  scripts/block_trace.py turns a run's trace into slot occupancy and the phases of a block's life.
#ifdef DVO_TRACE_BLOCKS
struct BlockTrace {
  unsigned long long t0, t_first, t_end;
  unsigned info;   // bits 0..3 steps (log2) | bit 4 likelihood block | bits 8..15 level width / 8 | bits 16..31 blocks of the item
  unsigned hw;     // HW_ID (bits 0..27) | XCC_ID << 28
};
constexpr unsigned kTraceCapacity = 1u << 22;
__device__ BlockTrace g_trace[kTraceCapacity];
__device__ unsigned g_trace_n;
__device__ __forceinline__ unsigned long long trace_clock() { return __builtin_amdgcn_s_memrealtime(); }
__shared__ unsigned long long trace_first;  // set by thread 0 of a residual block after its first step
__shared__ unsigned long long trace_tail;   // -DDVO_TRACE_TAIL: when thread 128 (the ordered combine of the four wave segments) was done
#endif

// ------------------------------------------------------------------------------------------------------------------
// residual pass
// ------------------------------------------------------------------------------------------------------------------

// computeResidualsSse for one reference pixel, dense_tracking_impl.cpp:171-294, split in three so that the 24 gather
// loads of a round (4 pixels x 6) are issued back to back instead of one pixel's loads waiting behind the previous
// pixel's arithmetic.  All three parts run in round-toward-zero.
struct Proj {
  float u, v, sz;
  int base;   // top-left gather index, 0 when the point falls outside
  bool inb;
};
struct Gathered {
  v4f a00, a10, a01, a11;  // {I, Z, Ix, Iy} of the four neighbours
  v4f_a8 b0, b1;           // {Zx, Zy} pairs of the upper and lower row
};

template <int RCP>
__device__ __forceinline__ Proj project_pixel_rtz(const float *kt, const LevelPairDesc &d, float x, float y, float z) {
  // hadd(hadd()) adds lanes (0,1) and (2,3) first (:178-188); the point's w is 1
  const float sx = (kt[0] * x + kt[1] * y) + (kt[2] * z + kt[3]);
  const float sy = (kt[4] * x + kt[5] * y) + (kt[6] * z + kt[7]);
  Proj p;
  p.sz = (kt[8] * x + kt[9] * y) + (kt[10] * z + kt[11]);
  const float rz = RCP == 2   ? rcp_host_nibbles<true>(p.sz, d.rcp_lds, d.rcp.shift, d.rcp.unit)  // :192 (_mm_rcp_ps) ...
                   : RCP == 1 ? rcp_host_table(p.sz, d.rcp)
                              : rcp_toward_zero(p.sz);                                        // ... / the exact quotient
  p.u = sx * rz, p.v = sy * rz;
  // 0 <= u <= w-2 and 0 <= v <= h-2 (:160-161,203); NaN compares false.  (Bitwise and: four compares and three scalar ands;
  // the short-circuit form compiles to an exec-masked region.)
  p.inb = (bool)((int)(p.u >= 0.0f) & (int)(p.u <= d.ub_x) & (int)(p.v >= 0.0f) & (int)(p.v <= d.ub_y));
  const int iu = (int)p.u, iv = (int)p.v;  // truncation == _mm_cvtps_epi32 under RTZ (:195)
  p.base = p.inb ? __mul24(iv, d.w) + iu : 0;  // both factors are below 2^24 for a pixel in bounds: one full-rate v_mad_i32_i24
  return p;
}

__device__ __forceinline__ Gathered gather_pixel(const LevelPairDesc &d, int base) {
  const DVO_GLOBAL void *ca = (const DVO_GLOBAL void *)d.c_a, *cb = (const DVO_GLOBAL void *)d.c_b;
  const unsigned oa0 = (unsigned)base * 16u, oa1 = oa0 + (unsigned)d.w * 16u;  // {I, Z, Ix, Iy}: 16 bytes per pixel
  const unsigned ob0 = (unsigned)base * 8u, ob1 = ob0 + (unsigned)d.w * 8u;    // {Zx, Zy}: 8 bytes per pixel
  Gathered g;
  g.a00 = ld_off<v4f>(ca, oa0), g.a10 = ld_off<v4f, 16>(ca, oa0);
  g.a01 = ld_off<v4f>(ca, oa1), g.a11 = ld_off<v4f, 16>(ca, oa1);
  g.b0 = ld_off<v4f_a8>(cb, ob0);
  g.b1 = ld_off<v4f_a8>(cb, ob1);
  return g;
}

// ZERO_E: also zero the gradient terms of an invalid pixel (the register-accumulator form needs that; the staged forms scale
// them by a zero weight with a multiply whose 0 * anything is 0, so the four selects are saved)
template <bool ZERO_E>
__device__ __forceinline__ void finish_pixel_rtz(const LevelPairDesc &d, const Proj &p, const Gathered &g, float z, float ri,
                                                 float rix, float riy, float &r0, float &r1, float &e2, float &e3, float &e4,
                                                 float &e5, bool &valid) {
  // u - (float)(int)u (:211-216): for a point in bounds (u >= 0) that is u - floor(u), an exact difference: v_fract_f32
  const float w1u = __builtin_amdgcn_fractf(p.u), w1v = __builtin_amdgcn_fractf(p.v);
  const float w0u = 1.0f - w1u, w0v = 1.0f - w1v;
  // bilinear blend, per channel: w0v*(w0u*c00 + w1u*c10) + w1v*(w0u*c01 + w1u*c11)  (:227-258)
#define DVO_BLEND(c00, c10, c01, c11) ((w0v * (w0u * (c00) + w1u * (c10))) + (w1v * (w0u * (c01) + w1u * (c11))))
  const float ci = DVO_BLEND(g.a00.x, g.a10.x, g.a01.x, g.a11.x);
  const float cz = DVO_BLEND(g.a00.y, g.a10.y, g.a01.y, g.a11.y);
  const float cix = DVO_BLEND(g.a00.z, g.a10.z, g.a01.z, g.a11.z);
  const float ciy = DVO_BLEND(g.a00.w, g.a10.w, g.a01.w, g.a11.w);
  const float czx = DVO_BLEND(g.b0.x, g.b0.z, g.b1.x, g.b1.z);
  const float czy = DVO_BLEND(g.b0.y, g.b0.w, g.b1.y, g.b1.w);
#undef DVO_BLEND
  // any NaN among the blended channels rejects the point (:261); channels 6,7 are always 0
  // (one unordered compare tests two channels)
  const bool has_nan = __builtin_isunordered(ci, cz) || __builtin_isunordered(cix, ciy) || __builtin_isunordered(czx, czy);
  // e = wcur * cur + wref * ref', ref' = {I, transformed depth, Ix, Iy} (:269-271)
  const float t0 = d.wc[0] * ci + d.wr[0] * ri;
  const float t1 = d.wc[1] * cz + d.wr[1] * p.sz;
  // occlusion test (:275) with depthStdDevZ (:122-128) of the reference depth
  float s = z - 0.4f;
  s = 0.0012f + (0.0019f * s) * s;
  valid = p.inb && !has_nan && (t1 > -20.0f * s);
  r0 = valid ? t0 : 0.0f;
  r1 = valid ? t1 : 0.0f;
  e2 = d.wc[2] * cix + d.wr[2] * rix;
  e3 = d.wc[3] * ciy + d.wr[3] * riy;
  e4 = d.wc[4] * czx;  // wref is 0 for the depth derivatives (dense_tracking.cpp:217-220)
  e5 = d.wc[5] * czy;
  if (ZERO_E) e2 = valid ? e2 : 0.0f, e3 = valid ? e3 : 0.0f, e4 = valid ? e4 : 0.0f, e5 = valid ? e5 : 0.0f;
}

typedef float v4acc __attribute__((ext_vector_type(4)));
// The Gram matrix of the staged 16-vector v = sqrt(w) (Ja[0..5], Jb[0..5], r0, r1, 0, 0), as 4 x 4 tiles of 4-component chunks:
// of the 16 chunk pairs only the 9 with ci <= cj < 4, ci < 3 hold moments (the tile is symmetric, r r^T is not needed).
// v_mfma_f32_4x4x1_16b_f32 forms sixteen independent 4 x 4 outer products per instruction in 8 issue cycles (measured 8.9;
// the 16x16x4 form: 33 for 4 points x 256 products, and matrix and vector instructions of a SIMD do NOT overlap on gfx950 --
// scripts/probes/mfma_valu_coexec.hip, mfma_4x4_blocks.hip): block b of an instruction is point 16 g + b of the step, so 4 x 9 = 36
// instructions (320 cycles) do the work of sixteen 16x16x4 ones (530).  DVO_GRAM_BLOCKS=0 builds the 16x16x4 form.
#ifndef DVO_GRAM_BLOCKS
#define DVO_GRAM_BLOCKS 1
#endif
constexpr int kGramPairs = 9;
// tile index of the chunk pair (ci <= cj): (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3)
__host__ __device__ constexpr int gram_pair(int ci, int cj) { return ci == 0 ? cj : ci == 1 ? 3 + cj : 5 + cj; }

// wave-local LDS hand-off: the LDS queue of a wave is in order, the fences only pin the compiler
#define DVO_WAVE_LDS_SYNC()                                 \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                        \
  } while (0)

// Timing-only ablation builds (scripts/ablate.sh: -DDVO_ABLATE=<mask>); never defined in the shipped library.
// 1: no LDS staging / MFMA   2: no gather loads   4: no rank / pair-sum logic   8: no residual spill store
// 16: no pixel steps at all (prologue + epilogue only)   32: no epilogue (reductions, block record)
// 64: epilogue without the seven wave reductions   128: without the moment sums of the four waves   256: without the ordered
// combine of the four wave segments
#ifndef DVO_ABLATE
#define DVO_ABLATE 0
#endif
#define DVO_KEEP(x) asm volatile("" ::"v"(x))

// The fused residual pass.  It walks the SELECTED pixels of the reference level in scan order -- the compacted arrays k_compact
// builds, as the reference walks PointSelection's dense array (dense_tracking_impl.cpp:169-171) -- so its cost follows the
// selection, not the image (round 5, end; until then every pixel of the level took a lane and unselected ones carried a NaN depth).
// A wave walks its segment in steps of 64 consecutive points, one point per lane: neighbouring selected pixels, so that every
// gather instruction of a step touches ~64 neighbouring pixels of the current image (whole cache lines, reused by the
// other three neighbour loads of the same step while they are still in L1).
//
// ACC 0: the 87 moments live in 87 fp32 registers per lane (fma), reduced across the wave with DPP at the end.
// ACC 1: the moments are the Gram matrix G = sum_p w_p v_p v_p^T of v = [Ja(6), Jb(6), r0, r1, 0, 0]; sqrt(w) v is staged
//        through LDS (lane-per-point -> lane-per-component) and accumulated by v_mfma_f32_16x16x4_f32 into 2 x 4
//        registers.  The MFMA sums over the points itself, so no 87-value wave reduction is needed and the kernel fits
//        4 waves per SIMD (the 87-register form is capped at 2).
//        (round 4: as nine 4 x 4 tiles on v_mfma_f32_4x4x1_16b_f32 into 9 x 4 registers, DVO_GRAM_BLOCKS above -- the default;
//        round 2's first 4x4 build had spills at four waves per SIMD and was dropped then.)
// (Measured and removed in round 3, DESIGN.md section 10: a five-waves-per-SIMD build, physical blocks walking several logical
// ones, item tables in device memory.)
// RCP 0 (default): 1 / z of the projection is the exactly truncated quotient, the t-distribution weight 7 v_rcp_f32(5 + d).
// RCP 1 (dvo_amd_set_reciprocal_mode): both reciprocals are the host's _mm_rcp_ps, bit for bit (dense_tracking_impl.cpp:192,700),
//       from a table in global memory; RCP 2: the same function from a 2 KiB correction table in LDS (rcp_host_nibbles).
template <int ACC, int RCP>
__device__ void residual_pass(const TickItem &it, const LevelPairDesc &d_in, const int lb) {
  LevelPairDesc d = d_in;
  __shared__ unsigned rcp_lds[RCP == 2 ? kRcpNibbleWordsMax : 1];
  if (RCP == 2) {
    // every block copies the corrections (L2-resident, 2 KiB) next to its staging area; the first projection waits for them
    const int words = (1 << (23 - d.rcp.shift)) >> 3;
    for (int i = threadIdx.x; i < words; i += kBlockThreads) rcp_lds[i] = ((const DVO_GLOBAL unsigned *)d.rcp.nibbles)[i];
    __syncthreads();
    d.rcp_lds = rcp_lds;
  }
  constexpr int kBufs = 2;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int seg = lb * kWavesPerBlock + wave;
  const int steps = item_res_steps(it);  // a segment is steps * 64 points
  unsigned idx = (unsigned)(seg * (kStepPx * steps) + lane);  // a POINT of the compacted selection (k_compact), not a pixel

  float acc[ACC == 0 ? kNumAcc : 1];
#pragma unroll
  for (int i = 0; i < (ACC == 0 ? kNumAcc : 1); ++i) acc[i] = 0.0f;
  v4acc gram_a = {0.0f, 0.0f, 0.0f, 0.0f}, gram_b = {0.0f, 0.0f, 0.0f, 0.0f};
  v4acc gram_p[kGramPairs];  // (DVO_GRAM_BLOCKS) one 4 x 4 tile per needed pair of 4-component chunks, 16 point classes each
#pragma unroll
  for (int k = 0; k < kGramPairs; ++k) gram_p[k] = v4acc{0.0f, 0.0f, 0.0f, 0.0f};
  // staging for the MFMA operands: [wave][point = lane][16 components], 16-byte chunks XOR-swizzled by point
  __shared__ __attribute__((aligned(16))) float stage[ACC >= 1 ? kWavesPerBlock * kBufs * kWave * 16 : 4];
  float S0[3] = {0.0f, 0.0f, 0.0f}, S1[3] = {0.0f, 0.0f, 0.0f};
  float first_w = 0.0f;
  int run_count = 0;                        // wave uniform
  // The pair quirk needs, for every valid pixel, the residual of the valid pixel before it in scan order.  Each wave keeps
  // the valid residuals of the current step compacted in LDS: slot 0 = the last valid pixel of earlier steps (zeros at
  // the start of a segment), slot 1 + k = the k-th valid pixel of this step; a pixel with k valid pixels below it in the
  // wave finds its predecessor in slot k.  (Ballot / mbcnt give k; no cross-lane shuffles, no carry registers.)
  __shared__ __attribute__((aligned(8))) v2f pair_slots[kWavesPerBlock][kWave + 2];
  v2f *const slots = pair_slots[wave];
  if (lane == 0) slots[0] = v2f{0.0f, 0.0f};

  // K*T lives in vector registers: as scalars the 12 values do not fit next to the descriptors, and the compiler re-reads
  // them from the kernel arguments inside every step, with a full scalar-memory wait in front of the projection
  float kt[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    kt[i] = it.kt[i];
    if (DVO_KT_SGPR)
      asm volatile("" : "+s"(kt[i]));
    else
      DVO_OPAQUE(kt[i]);
  }
  const bool unit_w = (it.flags & kItemUnitWeights) != 0;
  const float P0 = it.P[0], P1 = it.P[1], P2 = it.P[2], P3 = it.P[3];
  const DVO_GLOBAL void *const p_z = (const DVO_GLOBAL void *)d.r_zsel, *const p_i = (const DVO_GLOBAL void *)d.r_i,
                         *const p_ix = (const DVO_GLOBAL void *)d.r_ix, *const p_iy = (const DVO_GLOBAL void *)d.r_iy,
                         *const p_tx = (const DVO_GLOBAL void *)d.tx, *const p_ty = (const DVO_GLOBAL void *)d.ty;
  DVO_GLOBAL char *const p_res = (DVO_GLOBAL char *)((it.flags & kItemResBuf) ? d.res[1] : d.res[0]);  // no dynamic index: keeps d in registers

  // the points of the step about to be processed (depth, intensity, derivatives, ray: six coalesced dword loads, one step ahead);
  // all accesses are uniform base + 32-bit offset
  // (a two-step lead with two alternating register sets was measured: +10 VGPRs, +9 VALU per step, no gain in or out of cache)
  float n_z = ld_off<float>(p_z, 4u * idx), n_i = ld_off<float>(p_i, 4u * idx), n_ix = ld_off<float>(p_ix, 4u * idx),
        n_iy = ld_off<float>(p_iy, 4u * idx);
  float n_tx = ld_off<float>(p_tx, 4u * idx), n_ty = ld_off<float>(p_ty, 4u * idx);  // the point's ray (padding: zeros under a NaN depth)

  // Operand fetch of the Gram-matrix accumulation: lane l supplies component l & 15 of points 4 m + (l >> 4), m = 0..15, of
  // the 64 points a step staged.  With the write swizzle below, chunk (comp >> 2) of point p sits at chunk position
  // (comp >> 2) ^ ((p >> 1) & 3) = cb for even m and cb ^ 2 for odd m (cb = (comp >> 2) ^ (l >> 5)): two lane-constant
  // offsets plus compile-time immediates, no per-step address arithmetic.
  const int g_comp = lane & 15, g_sub = lane >> 4, g_cb = (g_comp >> 2) ^ (g_sub >> 1);
  const float *const g_even = stage + (wave * kBufs * kWave + g_sub) * 16 + ((g_cb << 2) | (g_comp & 3));
  const float *const g_odd = stage + (wave * kBufs * kWave + g_sub) * 16 + (((g_cb ^ 2) << 2) | (g_comp & 3));
  // The 4 x 4 form: lane l supplies component l & 3 of chunk c of point 16 g + (l >> 2) as the A (row) and B (column) operand of
  // the tiles chunk c takes part in.  The point's swizzle ((p >> 1) & 3 = (l >> 3) & 3, the same for every g) makes the four
  // chunk positions lane constants; g and the buffer are immediates.  Eight consecutive points x one chunk = 32 distinct banks.
  const int b_swz = (lane >> 3) & 3;
  const float *const b_base = stage + (wave * kBufs * kWave + (lane >> 2)) * 16 + (lane & 3);
  const float *const b_c0 = b_base + 4 * (0 ^ b_swz), *const b_c1 = b_base + 4 * (1 ^ b_swz), *const b_c2 = b_base + 4 * (2 ^ b_swz),
                     *const b_c3 = b_base + 4 * (3 ^ b_swz);
  auto gram_from_stage = [&](const int q) __attribute__((always_inline)) {
    if (DVO_GRAM_BLOCKS) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int off = (q & (kBufs - 1)) * kWave * 16 + g * 16 * 16;
        const float x0 = b_c0[off], x1 = b_c1[off], x2 = b_c2[off], x3 = b_c3[off];
        gram_p[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x0, gram_p[0], 0, 0, 0);
        gram_p[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, gram_p[1], 0, 0, 0);
        gram_p[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x2, gram_p[2], 0, 0, 0);
        gram_p[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x3, gram_p[3], 0, 0, 0);
        gram_p[4] = __builtin_amdgcn_mfma_f32_4x4x1f32(x1, x1, gram_p[4], 0, 0, 0);
        gram_p[5] = __builtin_amdgcn_mfma_f32_4x4x1f32(x1, x2, gram_p[5], 0, 0, 0);
        gram_p[6] = __builtin_amdgcn_mfma_f32_4x4x1f32(x1, x3, gram_p[6], 0, 0, 0);
        gram_p[7] = __builtin_amdgcn_mfma_f32_4x4x1f32(x2, x2, gram_p[7], 0, 0, 0);
        gram_p[8] = __builtin_amdgcn_mfma_f32_4x4x1f32(x2, x3, gram_p[8], 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int m = 0; m < 16; m += 2) {
      const float va = g_even[(q & (kBufs - 1)) * kWave * 16 + 64 * m];
      const float vb = g_odd[(q & (kBufs - 1)) * kWave * 16 + 64 * (m + 1)];
      gram_a = __builtin_amdgcn_mfma_f32_16x16x4f32(va, va, gram_a, 0, 0, 0);
      gram_b = __builtin_amdgcn_mfma_f32_16x16x4f32(vb, vb, gram_b, 0, 0, 0);
    }
  };

  // One step = 64 consecutive points of the compacted selection, one per lane.  The loop is unrolled by two (a segment always has a multiple of four
  // steps) so that the staging-buffer parity q is a compile-time constant and the one-step-ahead prefetch registers need
  // no rotation moves.
  auto do_step = [&](const int step, const int q, const bool prefetch) __attribute__((always_inline)) {
    const unsigned cur_idx = idx;
    // ---- round-to-nearest: reference point = pixel ray * depth (RgbdCamera::buildPointCloud, rgbd_image.cpp:245-262)
    float z = n_z, ri = n_i, rix = n_ix, riy = n_iy;
    float x = n_tx * z, y = n_ty * z;
    // prefetch the next step's reference scalars; they are consumed a whole step later
    idx += kWave;
    auto prefetch_next = [&]() __attribute__((always_inline)) {
      n_z = ld_off<float>(p_z, 4u * idx), n_i = ld_off<float>(p_i, 4u * idx), n_ix = ld_off<float>(p_ix, 4u * idx),
      n_iy = ld_off<float>(p_iy, 4u * idx);
      n_tx = ld_off<float>(p_tx, 4u * idx), n_ty = ld_off<float>(p_ty, 4u * idx);
    };
    if (prefetch) prefetch_next();

    // ---- switch to round-toward-zero: every float that crosses is made opaque on both sides of the s_setreg
    DVO_OPAQUE(x); DVO_OPAQUE(y); DVO_OPAQUE(z); DVO_OPAQUE(ri); DVO_OPAQUE(rix); DVO_OPAQUE(riy);
    round_toward_zero();
    DVO_OPAQUE(x); DVO_OPAQUE(y); DVO_OPAQUE(z); DVO_OPAQUE(ri); DVO_OPAQUE(rix); DVO_OPAQUE(riy);
    float r0, r1, e2, e3, e4, e5;
    bool ok;
    {
      const Proj p = project_pixel_rtz<RCP>(kt, d, x, y, z);
      Gathered g;
      if (DVO_ABLATE & 2) {
        const v4f c = {p.u, 1.5f, p.v, 0.25f};
        g.a00 = c, g.a10 = c, g.a01 = c, g.a11 = c, g.b0 = c, g.b1 = c;
        DVO_KEEP(p.base);
      } else {
        g = gather_pixel(d, p.base);
      }

      // While this step's gathers are in flight, feed the matrix pipe with the vectors the PREVIOUS step staged
      // (v_mfma_f32 ignores MODE.FP_ROUND -- probed on gfx950, scripts/probes/mfma_round.hip -- so it may sit in
      // the toward-zero window).  This overlaps ~512 matrix-pipe cycles with the gather latency and with the next
      // wave's arithmetic instead of serialising them behind this wave's own VALU work.
      if (ACC >= 1 && !(DVO_ABLATE & 1) && step > 0) {
        gram_from_stage(q ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      finish_pixel_rtz<ACC == 0>(d, p, g, z, ri, rix, riy, r0, r1, e2, e3, e4, e5, ok);
    }
    // ---- back to round-to-nearest
    DVO_OPAQUE(r0); DVO_OPAQUE(r1); DVO_OPAQUE(e2); DVO_OPAQUE(e3); DVO_OPAQUE(e4); DVO_OPAQUE(e5);
    DVO_OPAQUE(x); DVO_OPAQUE(y); DVO_OPAQUE(z);
    round_to_nearest();
    DVO_OPAQUE(r0); DVO_OPAQUE(r1); DVO_OPAQUE(e2); DVO_OPAQUE(e3); DVO_OPAQUE(e4); DVO_OPAQUE(e5);
    DVO_OPAQUE(x); DVO_OPAQUE(y); DVO_OPAQUE(z);

    // spill the residual of this iteration for the log-likelihood pass (NaN marks an invalid pixel)
    {
      const float qnan = u2f(0x7fc00000u);
      v2f sv;
      sv.x = ok ? r0 : qnan, sv.y = ok ? r1 : qnan;
      if (DVO_ABLATE & 8) {
        DVO_KEEP(sv.x);
        DVO_KEEP(sv.y);
      } else {
        *reinterpret_cast<DVO_GLOBAL v2f *>(p_res + 8u * cur_idx) = sv;
      }
    }

    // Branch free from here: an invalid pixel carries weight 0, zero residuals and a harmless point (exact zeros).
    // computeWeightsSse / computeWeight: w = (2+5)/(5 + r^T P r), mean 0 (dense_tracking_impl.cpp:640-707)
    float wgt = 1.0f;
    if (!unit_w) {
      // (round-to-nearest section: in the default mode fused multiply-adds are fine here, only the toward-zero stage is
      //  bit-matched; the host-rcpps kernels form the distance as computeWeightsSse does, product by product (:669-697), so that
      //  every weight is the reference's bit for bit)
      const float t0 = RCP ? r0 * P0 + r1 * P1 : __builtin_fmaf(r0, P0, r1 * P1);
      const float t1 = RCP ? r0 * P2 + r1 * P3 : __builtin_fmaf(r0, P2, r1 * P3);
      const float dd = RCP ? t0 * r0 + t1 * r1 : __builtin_fmaf(t0, r0, t1 * r1);
      wgt = 7.0f * (RCP == 2   ? rcp_host_nibbles<true>(5.0f + dd, d.rcp_lds, d.rcp.shift, d.rcp.unit)
                    : RCP == 1 ? rcp_host_table(5.0f + dd, d.rcp)
                               : __builtin_amdgcn_rcpf(5.0f + dd));
    }
    wgt = ok ? wgt : 0.0f;
    if (RCP && d.dbg_w) ((DVO_GLOBAL float *)d.dbg_w)[cur_idx] = ok ? wgt : u2f(0x7fc00000u);  // (dvo_amd_debug_weights only)

    // Jacobians at the untransformed reference point (dense_tracking.cpp:333-339,448-476)
    // (Formed from the pixel's ray instead -- x / z is tx, y / z is ty: five instructions fewer, same Jacobian to an ulp -- the
    //  batch ran at the same 47.0 k pairs/s, three interleaved runs, gpurun_out/r4f: not kept, the arithmetic stays as it was.)
    x = ok ? x : 0.0f, y = ok ? y : 0.0f, z = ok ? z : 1.0f;
    const float iz = __builtin_amdgcn_rcpf(z);
    const float iz2 = iz * iz;
    const float j02 = -x * iz2, j12 = -y * iz2;
    const float j03 = j02 * y, j13 = __builtin_fmaf(j12, y, -1.0f);
    const float j04 = __builtin_fmaf(-j02, x, 1.0f), j14 = -j03;
    const float j05 = -y * iz, j15 = x * iz;
    // The staged forms accumulate sqrt(w) v: the factor goes onto the four gradient terms and the three constants of Jz
    // before the 2 x 6 rows are formed (8 multiplies instead of 14 on the finished rows).  v_mul_legacy_f32: 0 * x = 0 for
    // every x, so the zero weight of an invalid pixel also wipes whatever its gradient terms hold (NaN included).
    const float sw = ACC == 0 ? 1.0f : __builtin_amdgcn_sqrtf(wgt);
    const float g2 = ACC == 0 ? e2 : mul_legacy(sw, e2), g3 = ACC == 0 ? e3 : mul_legacy(sw, e3);
    const float g4 = ACC == 0 ? e4 : mul_legacy(sw, e4), g5 = ACC == 0 ? e5 : mul_legacy(sw, e5);
    const float swx = sw * x, swy = sw * y;
    float Ja[6], Jb[6];
    Ja[0] = g2 * iz;
    Ja[1] = g3 * iz;
    Ja[2] = __builtin_fmaf(g2, j02, g3 * j12);
    Ja[3] = __builtin_fmaf(g2, j03, g3 * j13);
    Ja[4] = __builtin_fmaf(g2, j04, g3 * j14);
    Ja[5] = __builtin_fmaf(g2, j05, g3 * j15);
    Jb[0] = g4 * iz;
    Jb[1] = g5 * iz;
    Jb[2] = __builtin_fmaf(g4, j02, __builtin_fmaf(g5, j12, -sw));
    Jb[3] = __builtin_fmaf(g4, j03, __builtin_fmaf(g5, j13, -swy));
    Jb[4] = __builtin_fmaf(g4, j04, __builtin_fmaf(g5, j14, swx));
    Jb[5] = __builtin_fmaf(g4, j05, g5 * j15);
    // A += J^T (w P) J and b -= J^T (w P) r are linear in P: accumulate the P-free moments (87 sums)
    if (ACC == 0) {
      float wa[6], wb[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) wa[i] = wgt * Ja[i], wb[i] = wgt * Jb[i];
      int t = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = i; j < 6; ++j, ++t) {
          acc[kAccAA + t] = __builtin_fmaf(wa[i], Ja[j], acc[kAccAA + t]);
          acc[kAccAB + t] = __builtin_fmaf(wa[i], Jb[j], __builtin_fmaf(wb[i], Ja[j], acc[kAccAB + t]));
          acc[kAccBB + t] = __builtin_fmaf(wb[i], Jb[j], acc[kAccBB + t]);
        }
        acc[kAccAR0 + i] = __builtin_fmaf(wa[i], r0, acc[kAccAR0 + i]);
        acc[kAccAR1 + i] = __builtin_fmaf(wa[i], r1, acc[kAccAR1 + i]);
        acc[kAccBR0 + i] = __builtin_fmaf(wb[i], r0, acc[kAccBR0 + i]);
        acc[kAccBR1 + i] = __builtin_fmaf(wb[i], r1, acc[kAccBR1 + i]);
      }
    } else if (DVO_ABLATE & 1) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        DVO_KEEP(Ja[i]);
        DVO_KEEP(Jb[i]);
      }
      DVO_KEEP(wgt);
    } else {
      // stage sqrt(w) * v for this lane's pixel; chunk c of point p sits at chunk position c ^ ((p >> 1) & 3).
      // Two buffers alternate: the matrix pipe consumes this one during the NEXT step (see above).
      float *buf = stage + ((wave * kBufs + (q & (kBufs - 1))) * kWave) * 16;
      v4f *row = reinterpret_cast<v4f *>(buf + lane * 16);
      const int swz = (lane >> 1) & 3;
      v4f c0 = {Ja[0], Ja[1], Ja[2], Ja[3]};  // the rows carry sqrt(w) already
      v4f c1 = {Ja[4], Ja[5], Jb[0], Jb[1]};
      v4f c2 = {Jb[2], Jb[3], Jb[4], Jb[5]};
      v4f c3 = {sw * r0, sw * r1, 0.0f, 0.0f};
      // the LDS queue of a wave is in order: these writes land after the reads of two steps ago and before the reads of
      // the next step; the fences only stop the compiler from reordering across them
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      row[0 ^ swz] = c0;
      row[1 ^ swz] = c1;
      row[2 ^ swz] = c2;
      row[3 ^ swz] = c3;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }

    if (DVO_ABLATE & 4) {
      S0[0] += wgt * r0;
      run_count += __popcll(__ballot(ok));
      return;
    }
    // ---- rank of every valid pixel in scan order within this wave's segment (needed by the pair quirk Q5)
    const unsigned long long bk = __builtin_amdgcn_ballot_w64(ok);  // (__ballot() takes the flag through a vector register and back)
    const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bk, 0u));
    const int rank = run_count + pos;
    const int n_here = __popcll(bk);
    DVO_WAVE_LDS_SYNC();
    if (ok) slots[pos + 1] = v2f{r0, r1};
    DVO_WAVE_LDS_SYNC();
    const v2f prev = slots[pos];  // the valid pixel before this one (invalid lanes read a valid slot too: weight 0)
    DVO_WAVE_LDS_SYNC();
    if (n_here) slots[0] = slots[n_here];  // every lane moves the same value: the carry for the next step
    {
      // computeScaleSse with Q5: a pair (2j, 2j+1) contributes (w_2j + w_2j+1) r_2j r_2j^T (:603-621).
      // S0 assumes this segment starts on an even global rank, S1 on an odd one.  Invalid pixels have weight 0.
      // the segment's first valid pixel reads the zeros slot 0 starts with, so its "predecessor" products vanish.
      // (the residual is picked before it is squared: four selects instead of six)
      const bool odd = (rank & 1) != 0;
      const float a0 = odd ? prev.x : r0, a1 = odd ? prev.y : r1;  // what the even-start hypothesis weights at this pixel
      const float b0 = odd ? r0 : prev.x, b1 = odd ? r1 : prev.y;  // ... the odd-start hypothesis
      S0[0] = __builtin_fmaf(wgt, a0 * a0, S0[0]);
      S0[1] = __builtin_fmaf(wgt, a0 * a1, S0[1]);
      S0[2] = __builtin_fmaf(wgt, a1 * a1, S0[2]);
      S1[0] = __builtin_fmaf(wgt, b0 * b0, S1[0]);
      S1[1] = __builtin_fmaf(wgt, b0 * b1, S1[1]);
      S1[2] = __builtin_fmaf(wgt, b1 * b1, S1[2]);
      // first valid pixel of the segment: its partner (if any) lives in an earlier segment
      first_w = (ok && rank == 0) ? wgt : first_w;
    }
    run_count += n_here;
  };
  // the last pair of steps is peeled so that "is there a next step to prefetch" is a compile-time fact in every copy
  if (!(DVO_ABLATE & 16)) {  // (ablation 16: prologue + epilogue only)
    if (steps == 1) {
      do_step(0, 1, false);  // a one-step segment stages into buffer 1, which the epilogue consumes
#if defined(DVO_TRACE_BLOCKS) && !defined(DVO_TRACE_DESC)
      if (threadIdx.x == 0) trace_first = trace_clock();
#endif
    } else {
#ifdef DVO_TRACE_BLOCKS
      bool marked = false;
#endif
      for (int step = 0; step + 2 < steps; step += 2) {
        do_step(step, 0, true);
#if defined(DVO_TRACE_BLOCKS) && !defined(DVO_TRACE_DESC)
        if (!marked && threadIdx.x == 0) trace_first = trace_clock();
        marked = true;
#endif
        do_step(step + 1, 1, true);
      }
      do_step(steps - 2, 0, true);
#if defined(DVO_TRACE_BLOCKS) && !defined(DVO_TRACE_DESC)
      if (!marked && threadIdx.x == 0) trace_first = trace_clock();
#endif
      do_step(steps - 1, 1, false);
    }
  }

  if (DVO_ABLATE & 32) {  // (ablation 32: no epilogue -- keep the accumulators alive, write nothing)
    float keep = S0[0] + S1[0] + first_w + gram_a[0] + gram_b[0] + (float)run_count;
    for (int k = 0; k < kGramPairs; ++k) keep += gram_p[k][0];
    DVO_KEEP(keep);
    return;
  }
  if (ACC >= 1 && !(DVO_ABLATE & 1) && steps > 0) gram_from_stage(1);  // the last step (odd index) staged into buffer 1

  // ---- wave reduction, then the four waves of the block through LDS
  __shared__ float sm[kWavesPerBlock][kRecStride];
  if (ACC == 0) {
#pragma unroll
    for (int i = 0; i < kNumAcc; ++i) {
      const float s = wave_sum_to_lane63(acc[i]);
      if (lane == 63) sm[wave][kRecAcc + i] = s;
    }
  } else {
    // C/D layout of the 16x16 MFMA: register r of lane l is G[row = (l>>4)*4 + r][col = l&15]
    // into the start of the wave's own staging area (its reads of it are behind it in the in-order LDS queue)
    float *gsm = stage + wave * kBufs * kWave * 16;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (DVO_GRAM_BLOCKS) {
      // D layout of the 4 x 4 form: register r of lane l = tile[block l >> 2][row r][column l & 3].  The four blocks of a
      // 16-lane row are added in registers, (b0 + b1) + (b2 + b3); the four rows and the four waves by the threads that
      // gather the moments below: gsm[tile][row of lanes][column][row] (one 16-byte store per tile)
#pragma unroll
      for (int k = 0; k < kGramPairs; ++k) {
        v4acc t = gram_p[k];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = t[r];
          v += dpp_read<0x124, 0xF>(v);  // row_ror:4
          v += dpp_read<0x128, 0xF>(v);  // row_ror:8
          t[r] = v;
        }
        if ((lane & 12) == 0) *reinterpret_cast<v4f *>(gsm + ((k * 4 + (lane >> 4)) * 4 + (lane & 3)) * 4) = v4f{t[0], t[1], t[2], t[3]};
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) gsm[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = gram_a[r] + gram_b[r];
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float s0 = (DVO_ABLATE & 64) ? S0[i] : wave_sum_to_lane63(S0[i]);
    const float s1 = (DVO_ABLATE & 64) ? S1[i] : wave_sum_to_lane63(S1[i]);
    if (lane == 63) sm[wave][kRecS0 + i] = s0, sm[wave][kRecS1 + i] = s1;
  }
  {
    const float fw = (DVO_ABLATE & 64) ? first_w : wave_sum_to_lane63(first_w);
    if (lane == 63) {
      const v2f last = slots[0];  // the last valid residual of the segment (zeros if there is none)
      sm[wave][kRecFirstW] = fw;
      sm[wave][kRecCount] = u2f((unsigned)run_count);
      sm[wave][kRecLastR] = last.x;
      sm[wave][kRecLastR + 1] = last.y;
    }
  }
  // which Gram-matrix entries make up moment `tid` (rows/cols 0-5 = Ja, 6-11 = Jb, 12 = r0, 13 = r1): worked out while the
  // other waves are still on their way to the barrier
  const int tid = threadIdx.x;
  int e0 = 0, e1 = -1;
  if (ACC >= 1 && tid < kNumAcc) {
    if (tid < kAccAR0) {
      const int kind = tid / 21;
      int t = tid - kind * 21, i = 0;
      while (t >= 6 - i) t -= 6 - i, ++i;
      const int j = i + t;
      if (kind == 0) {
        e0 = i * 16 + j;
      } else if (kind == 1) {
        e0 = i * 16 + 6 + j, e1 = j * 16 + 6 + i;
      } else {
        e0 = (6 + i) * 16 + 6 + j;
      }
    } else {
      const int q = tid - kAccAR0, grp = q / 6, i = q - grp * 6;  // AR0, AR1, BR0, BR1
      e0 = ((grp >> 1) * 6 + i) * 16 + 12 + (grp & 1);
    }
    if (DVO_GRAM_BLOCKS) {  // entry (row, column), row <= column chunk-wise -> offset of gsm[tile][0][column & 3][row & 3]
      auto at = [](const int e) { return ((gram_pair((e >> 4) >> 2, (e & 15) >> 2) * 4) * 4 + (e & 3)) * 4 + ((e >> 4) & 3); };
      e0 = at(e0);
      if (e1 >= 0) e1 = at(e1);
    }
  }
  __syncthreads();

  DVO_GLOBAL float *rec = (DVO_GLOBAL float *)d.records + (size_t)lb * kRecStride;
  if ((DVO_ABLATE & 128) && tid < kNumAcc) {
    rec[kRecAcc + tid] = stage[tid];
  } else if ((DVO_ABLATE & 256) && tid == 128) {
    rec[kRecCount] = sm[0][kRecCount];
  } else if (tid < kNumAcc) {
    if (ACC == 0) {
      rec[kRecAcc + tid] = (sm[0][kRecAcc + tid] + sm[1][kRecAcc + tid]) + (sm[2][kRecAcc + tid] + sm[3][kRecAcc + tid]);
    } else {
      constexpr int kW = kBufs * kWave * 16;  // floats between the staging areas of two waves
      float v;
      if (DVO_GRAM_BLOCKS) {
        auto rows = [&](const float *p) { return (p[0] + p[16]) + (p[32] + p[48]); };  // the four 16-lane rows of a wave
        auto waves = [&](const int e) { return (rows(stage + e) + rows(stage + kW + e)) + (rows(stage + 2 * kW + e) + rows(stage + 3 * kW + e)); };
        v = waves(e0);
        if (e1 >= 0) v += waves(e1);
      } else {
        v = (stage[e0] + stage[kW + e0]) + (stage[2 * kW + e0] + stage[3 * kW + e0]);
        if (e1 >= 0) v += (stage[e1] + stage[kW + e1]) + (stage[2 * kW + e1] + stage[3 * kW + e1]);
      }
      rec[kRecAcc + tid] = v;
    }
  } else if (tid == 128) {
    // ordered combine of the four wave segments (see combine rule in k_finalize).  Everything is read first and the fold is
    // written without branches (an empty segment changes nothing, through selects): one round of LDS latency instead of
    // eight in the tail of every block
    unsigned CB[kWavesPerBlock];
    float FW[kWavesPerBlock], L0[kWavesPerBlock], L1[kWavesPerBlock], A0[kWavesPerBlock][3], A1[kWavesPerBlock][3];
#pragma unroll
    for (int wv = 0; wv < kWavesPerBlock; ++wv) {
      CB[wv] = f2u(sm[wv][kRecCount]), FW[wv] = sm[wv][kRecFirstW], L0[wv] = sm[wv][kRecLastR], L1[wv] = sm[wv][kRecLastR + 1];
#pragma unroll
      for (int i = 0; i < 3; ++i) A0[wv][i] = sm[wv][kRecS0 + i], A1[wv][i] = sm[wv][kRecS1 + i];
    }
    unsigned c = 0;
    float fw = 0.0f, l0 = 0.0f, l1 = 0.0f;
    float s0[3] = {0.0f, 0.0f, 0.0f}, s1[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int wv = 0; wv < kWavesPerBlock; ++wv) {
      rec[kRecWaveCnt + wv] = u2f(CB[wv]);
      const bool ne = CB[wv] != 0;           // an empty segment is skipped
      const bool first = c == 0;             // nothing before it: its first weight is the combined segment's
      const bool flip = (c & 1u) != 0;       // b starts on the opposite parity of a
      const float bfw = FW[wv];
      // b's first pixel is a pair-second when b starts on an odd rank: it weights a's last residual
      const float r[3] = {l0 * l0, l0 * l1, l1 * l1};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float x0 = flip ? A1[wv][i] : A0[wv][i];  // contribution if the combined segment starts even
        float x1 = flip ? A0[wv][i] : A1[wv][i];  // ... starts odd
        const float add = bfw * r[i];
        x0 = (!first && flip) ? x0 + add : x0;
        x1 = (!first && !flip) ? x1 + add : x1;
        s0[i] = ne ? s0[i] + x0 : s0[i];
        s1[i] = ne ? s1[i] + x1 : s1[i];
      }
      fw = (ne && first) ? bfw : fw;
      l0 = ne ? L0[wv] : l0, l1 = ne ? L1[wv] : l1;
      c += CB[wv];
    }
    rec[kRecCount] = u2f(c);
    rec[kRecFirstW] = fw;
    rec[kRecLastR] = l0;
    rec[kRecLastR + 1] = l1;
    for (int i = 0; i < 3; ++i) rec[kRecS0 + i] = s0[i], rec[kRecS1 + i] = s1[i];
    rec[14] = 0.0f, rec[15] = 0.0f;
#if defined(DVO_TRACE_BLOCKS) && defined(DVO_TRACE_TAIL)
    __hip_atomic_store(&trace_tail, trace_clock(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
  }
}

// ------------------------------------------------------------------------------------------------------------------
// log-likelihood pass: sum over the first 50*floor(V/50) valid residuals of log(1 + 0.2 r^T P r)
// (computeCompleteDataLogLikelihood, dense_tracking_impl.cpp:406-425, incl. Q6).  Same segment geometry as the residual
// pass that wrote the residuals; {cut_seg, cut_local} locate global rank 50*floor(V/50).
// ------------------------------------------------------------------------------------------------------------------
// 1 + 0.2 q in double, one fused operation: the reference's expression (dense_tracking_impl.cpp:415) is contracted by its own
// build (-O3 -march=native, GNU contraction default), the oracle's portable build keeps the two roundings apart; the two differ by
// at most half an ulp of a double per term, far below the float the likelihood is returned as.  One fp64 operation less per pixel
// in a pass that is issue-bound on them.
__device__ __forceinline__ double ll_term(float q) { return __builtin_fma(0.2, (double)q, 1.0); }
// v_max_f32 without the canonicalising self-max the compiler puts in front of fmaxf's accumulator (q is never a signalling NaN)
__device__ __forceinline__ float max_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ void loglik_pass(const TickItem &it, const LevelPairDesc &d, const int lb) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  // Merged blocks: likelihood block `lb` covers up to m residual blocks of the pass that filled the buffer, never across a chunk
  // boundary of the level's summation tree (ll_block_range); its wave w walks the `count` consecutive residual wave segments
  // from segment 4 first + w count on.  (A likelihood step is a tenth of a residual step's work: blocks as short as the
  // residual pass's are all prologue and hold a block slot for it.)  The geometry is a function of the level alone, so a lane's
  // running product -- and with it every log -- is cut at the same pixels whatever else the tick carries.
  int blk_first, blk_count;
  ll_block_range((int)it.ll_level_blocks, item_ll_merge_log2(it), lb, &blk_first, &blk_count);
  const int sub = item_ll_steps(it);                          // steps of one residual wave segment
  const int seg = blk_first * kWavesPerBlock + wave * blk_count;  // first residual wave segment of this wave
  const int steps = blk_count * sub;
  // The first chunk of residuals is requested BEFORE the prefix table is looked at (the table decides whether the segment counts
  // at all, but the addresses do not depend on it): one memory round trip less in the dependent chain every block starts with.
  constexpr int kLlChunk = 16;
  v2f cur[kLlChunk], nxt[kLlChunk];
  const DVO_GLOBAL v2f *src = (const DVO_GLOBAL v2f *)((it.flags & kItemLlBuf) ? d.res[1] : d.res[0]) + seg * (kStepPx * sub) + lane;
  auto load_chunk = [&](v2f(&dst)[kLlChunk], const int first) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < kLlChunk; ++k)
      if (first + k < steps) dst[k] = src[(first + k) * kWave];
  };
  load_chunk(cur, 0);
  // valid pixels of this band that precede the segment (written by k_finalize of the residual pass that filled the buffer)
  const int seg_before =
      steps > 0 ? ((const DVO_GLOBAL int *)((it.flags & kItemLlBuf) ? d.seg_prefix[1] : d.seg_prefix[0]))[seg] : 0;
  const int cut_rank = it.ll_cut_rank;
  const float P0 = it.P[0], P1 = it.P[1], P2 = it.P[2], P3 = it.P[3];
  const unsigned long long below = (1ull << lane) - 1ull;
  double total = 0.0;
  // The largest Mahalanobis distance among the residuals that count: the reference multiplies 50 consecutive terms
  // 1 + 0.2 q in a double before it takes a log (dense_tracking_impl.cpp:413-419), and that product overflows -- likelihood
  // -inf, iteration rejected -- when 50 consecutive q are all above ~7e6 (noise-free synthetic depth; never sensor data).
  // Here a log is taken before the running product can leave the double range, so nothing overflows; the host is told the largest q of the pass and, only if one
  // group of 50 COULD have overflowed, asks k_ll_overflow for the exact answer (dvo_tracker.cpp: ll_overflowed).
  float qmax = 0.0f;
  if (steps > 0 && seg_before < cut_rank) {
    int run_count = seg_before;
    // four steps (256 pixels) per trip; the terms 1 + 0.2 r^T P r >= 1 of a lane are multiplied up in a double and a log
    // is taken of the product, like the reference takes one log per 50 residuals (dense_tracking_impl.cpp:415-419) -- here
    // whenever the product has grown past 1e150 (checked once per trip) and at the end of the segment: the fp64 log is ninety
    // instructions, and one per sixteen terms was two fifths of this pass's issue time
    double prod = 1.0;
    const bool all_below_cut = seg_before + steps * kWave <= cut_rank;  // wave uniform: every pixel of the segment counts
    // The residuals are fetched kLlChunk steps at a time and one chunk ahead of the arithmetic: sixteen 8-byte loads per lane in
    // flight while the previous sixteen steps are worked through.  (Until round 4 a trip loaded its four steps and then used them:
    // a block of the likelihood pass spent its life waiting for memory -- 24 us for 32 steps, more than a residual block's 17 us
    // for a tenth of the instructions -- and the pass held a quarter of the GPU's block slots: scripts/block_trace.py.)  The
    // arithmetic, and the order it is done in, are unchanged: trips of four steps, the product checked once per trip.
    for (int c0 = 0; c0 < steps; c0 += kLlChunk) {
      if (c0 + kLlChunk < steps) load_chunk(nxt, c0 + kLlChunk);
#pragma unroll
      for (int t = 0; t < kLlChunk; t += 4) {
        if (c0 + t >= steps) break;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (c0 + t + k >= steps) break;
          const v2f r = cur[t + k];
          const bool valid = r.x == r.x;
          if (all_below_cut) {  // only the segment(s) around rank 50 * floor(V / 50) need ranks: Q6 drops at most 49 residuals
            if (valid) {
              const float t0 = r.x * P0 + r.y * P1;
              const float t1 = r.x * P2 + r.y * P3;
              const float q = t0 * r.x + t1 * r.y;
              prod *= ll_term(q);
              qmax = max_raw(qmax, q);
            }
            continue;
          }
          const unsigned long long b = __ballot(valid);
          const int rank = run_count + __popcll(b & below);
          if (valid && rank < cut_rank) {
            const float t0 = r.x * P0 + r.y * P1;
            const float t1 = r.x * P2 + r.y * P3;
            const float q = t0 * r.x + t1 * r.y;
            prod *= ll_term(q);
            qmax = max_raw(qmax, q);
          }
          run_count += __popcll(b);
        }
        // a log only when the running product gets large: a term is at most 1 + 0.2 * FLT_MAX < 7e37, so four more of them on top
        // of 1e150 stay below the double range; with ordinary residuals (terms of 1 .. 20) a lane takes one log per wave segment
        if (prod > 1e150) {
          total += log(prod);
          prod = 1.0;
        }
      }
#pragma unroll
      for (int k = 0; k < kLlChunk; ++k) cur[k] = nxt[k];
    }
    if (prod != 1.0) total += log(prod);
  }
  total = wave_sum_double(total);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) qmax = __builtin_fmaxf(qmax, __shfl_xor(qmax, m, 64));
  __shared__ double smd[kWavesPerBlock];
  __shared__ float smq[kWavesPerBlock];
  if (lane == 0) smd[wave] = total, smq[wave] = qmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    ((DVO_GLOBAL double *)d.ll_partials)[lb] = (smd[0] + smd[1]) + (smd[2] + smd[3]);
    ((DVO_GLOBAL float *)d.ll_qmax)[lb] = __builtin_fmaxf(__builtin_fmaxf(smq[0], smq[1]), __builtin_fmaxf(smq[2], smq[3]));
  }
}

// The exact answer to "did the reference's 50-term likelihood product overflow?" for one residual buffer, asked only when the
// pass above saw a Mahalanobis distance large enough to make it possible.  A block is ONE wave and owns the groups of fifty
// consecutive valid residuals (in rank order) that START inside its chunk of `segs_per_block` wave segments of the residual
// pass: the prefix table gives the rank of the chunk's first valid pixel, the wave skips what is left of the previous chunk's
// last group, walks on in scan order -- past the end of its chunk until its last group is complete -- compacts the terms
// 1 + 0.2 q into LDS in rank order, and lane g multiplies group g's fifty terms one after the other in a double: the
// reference's own loop (error_acc *= ..., dense_tracking_impl.cpp:415-417), so the decision is the reference's bit for bit.
constexpr int kOvfGroups = 64, kOvfWindow = kOvfGroups * 50;
__global__ __launch_bounds__(kWave) void k_ll_overflow(const float2 *__restrict__ res, const int *__restrict__ seg_prefix, int seg_first,
                                                       int n_segs, int segs_per_block, int seg_px, int rank_offset, int n_px,
                                                       int cut_rank, int rank_end, float P0, float P1, float P2, float P3,
                                                       unsigned *__restrict__ result_host) {
  __shared__ double terms[kOvfWindow + kWave];
  const int lane = threadIdx.x;
  const unsigned long long below = (1ull << lane) - 1ull;
  const int s0 = seg_first + (int)blockIdx.x * segs_per_block;      // first wave segment of the chunk
  const int s1 = s0 + segs_per_block < seg_first + n_segs ? s0 + segs_per_block : seg_first + n_segs;
  const int a = rank_offset + seg_prefix[s0];                        // global rank of the chunk's first valid pixel
  // rank of the first valid pixel BEHIND the chunk (only the chunk's own segments are known to hold this pass's residuals)
  // rank_end >= 0: the band is CLOSED -- the residuals behind it live on another GPU -- and rank_end is the rank of the first
  // valid pixel behind it: the band's last chunk then stops at the last group that ends inside the band (the group that
  // straddles the edge is settled by the host from the ranks' edge records, dvo_sharded.cpp: sharded_overflow)
  const bool last_chunk = s1 >= seg_first + n_segs;
  const int e = !last_chunk ? rank_offset + seg_prefix[s1] : (rank_end >= 0 ? rank_end : 0x7fffffff);
  const int first = a + (50 - a % 50) % 50;                          // first group that starts in the chunk
  int stop = e >= cut_rank ? cut_rank                                // (cut_rank is a multiple of 50)
             : (last_chunk && rank_end >= 0) ? e - e % 50            // closed band: complete groups only
                                             : e + (50 - e % 50) % 50;  // ... else the end of the group that straddles the chunk's end
  // A closed band: NO chunk may follow a group past the band's last valid pixel -- a chunk that is not the band's last one can
  // still hold the start of the group that straddles the band edge when fewer than fifty valid pixels lie behind it (a depth
  // hole at the band edge); the residuals behind the edge are another GPU's, this rank's copy of them is stale
  int px_end = n_px;
  if (rank_end >= 0) {
    const int closed = rank_end - rank_end % 50;
    stop = stop < closed ? stop : closed;
    px_end = (seg_first + n_segs) * seg_px;
  }
  if (first >= stop) return;
  int run = a, base = first;  // rank of the next valid pixel; rank of terms[0]
  bool overflow = false;
  auto flush = [&](int n_groups) {  // lanes 0 .. n_groups - 1 multiply one group each
    __syncthreads();
    if (lane < n_groups) {
      double acc = 1.0;
      for (int i = 0; i < 50; ++i) acc *= terms[lane * 50 + i];
      overflow = overflow || !(acc <= 1.7976931348623157e308);
    }
    __syncthreads();
  };
  for (int i0 = s0 * seg_px; i0 < px_end && run < stop; i0 += kWave) {
    const float2 r = res[i0 + lane];  // (the buffer is padded to whole wave segments)
    const bool valid = r.x == r.x;
    const unsigned long long b = __ballot(valid);
    const int rank = run + __popcll(b & below);
    if (valid && rank >= first && rank < stop) {
      const float t0 = r.x * P0 + r.y * P1;
      const float t1 = r.x * P2 + r.y * P3;
      const float q = t0 * r.x + t1 * r.y;
      terms[rank - base] = 1.0 + 0.2 * (double)q;
    }
    run += __popcll(b);
    const int have = (run < stop ? run : stop) - base;
    if (have >= kOvfWindow) {
      flush(kOvfGroups);
      const double t = terms[kOvfWindow + lane];  // at most 63 terms of the next window: move them to the front
      __syncthreads();
      if (lane < have - kOvfWindow) terms[lane] = t;
      __syncthreads();
      base += kOvfWindow;
    }
  }
  const int left = (run < stop ? run : stop) - base;
  if (left >= 50) flush(left / 50);  // (a last group cut short by the end of the buffer cannot exist: cut_rank <= V)
  const unsigned long long any = __ballot(overflow);
  if (lane == 0 && any) __hip_atomic_store(result_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Copy the descriptors into registers once, through the constant address space (scalar loads).  Read through plain global
// references their fields would be re-loaded with vector loads + s_waitcnt vmcnt(0) at every use inside the pixel loop,
// because the residual stores might alias them: ten dependent L2 round trips per 64 pixels.
#define DVO_CONST __attribute__((address_space(4)))
__device__ __forceinline__ LevelPairDesc load_desc(const TickItem &it) {
  const DVO_CONST RefLevelDesc *r = (const DVO_CONST RefLevelDesc *)it.ref;
  const DVO_CONST CurLevelDesc *c = (const DVO_CONST CurLevelDesc *)it.cur;
  const DVO_CONST SlotDesc *s = (const DVO_CONST SlotDesc *)it.slot;
  LevelPairDesc d;
  d.r_zsel = r->r_zsel, d.r_i = r->r_i, d.r_ix = r->r_ix, d.r_iy = r->r_iy;
  d.tx = r->tx, d.ty = r->ty;
  d.c_a = c->c_a, d.c_b = c->c_b;
  d.w = c->w, d.h = c->h;
#pragma unroll
  for (int i = 0; i < 6; ++i) d.wc[i] = c->wc[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) d.wr[i] = c->wr[i];
  d.ub_x = c->ub_x, d.ub_y = c->ub_y;
  d.res[0] = s->res[0], d.res[1] = s->res[1];
  d.records = s->records, d.ll_partials = s->ll_partials, d.ll_qmax = s->ll_qmax;
  d.seg_prefix[0] = s->seg_prefix[0], d.seg_prefix[1] = s->seg_prefix[1];
  return d;
}

template <int ACC, int RCP>
__device__ __forceinline__ void tick_body(const TickItem &it, const RcpTable &rcp, const int bx) {
  const int rb = it.res_blocks;
  if (bx >= rb + it.ll_blocks) return;
#ifdef DVO_TRACE_BLOCKS
  const unsigned long long trace_t0 = trace_clock();
  // (the record's slot is taken NOW: the atomic's round trip, ~0.6 us under load, must not sit in front of the end stamp -- it did
  //  until the end of round 4, and every block looked that much longer)
  unsigned trace_slot = 0;
  if (threadIdx.x == 0) trace_first = 0, trace_tail = 0, trace_slot = atomicAdd(&g_trace_n, 1u);
#endif
  LevelPairDesc d = load_desc(it);
#ifdef DVO_TRACE_DESC  // (a variant of the trace: the "first step" stamp is taken when the descriptors have arrived instead)
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DVO_KEEP(d.w);
    trace_first = trace_clock();
  }
#endif
  d.rcp = rcp;
  d.rcp_lds = nullptr;
  d.dbg_w = RCP ? ((const DVO_CONST SlotDesc *)it.slot)->dbg_w : nullptr;
  if (bx < rb)
    residual_pass<ACC, RCP>(it, d, it.res_first + xcd_contiguous_block(bx, rb));
  else
    loglik_pass(it, d, (int)it.ll_first + (bx - rb));
#ifdef DVO_TRACE_BLOCKS
  if (threadIdx.x == 0) {
    const unsigned long long trace_t1 = trace_clock();
    const unsigned slot = trace_slot;
    if (slot < kTraceCapacity) {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      BlockTrace t;
      t.t0 = trace_t0, t.t_first = trace_first, t.t_end = trace_t1;
#ifdef DVO_TRACE_TAIL  // (the block's other tail: spin until thread 128 has stamped, at most ~20 us; t_first carries its stamp)
      if (bx < rb) {
        const unsigned long long t_spin = trace_clock();
        unsigned long long tail = 0;
        while ((tail = __hip_atomic_load(&trace_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0 && trace_clock() - t_spin < 2000) {}
        t.t_first = tail;
      }
#endif
      t.info = (unsigned)(bx < rb ? (it.steps_log2 & 15) : (it.steps_log2 >> 4)) | (bx < rb ? 0u : 16u) | ((unsigned)(d.w / 8) << 8) |
               ((unsigned)(bx < rb ? rb : it.ll_blocks) << 16);
      t.hw = (hw & 0x0fffffffu) | (xcc << 28);
      g_trace[slot] = t;
    }
  }
#endif
}

// Which item owns this block, and which of the item's blocks is it?  Two-dimensional grid: (block, item).  One-dimensional
// ("compact") grid: lanes 0 .. n_items look at the items' first block groups; the owner is the last item that starts at or
// before this block's group (every wave finds the same one).
template <class Args>
__device__ __forceinline__ int tick_locate(const Args &args, int &bx, const unsigned block = blockIdx.x) {
  constexpr int kSlots = (int)(sizeof(args.group_first) / sizeof(args.group_first[0]));
  if (!args.compact) {
    const int rot = args.xcd_rot[blockIdx.y];
    bx = (int)((block & ~7u) | ((block + rot) & 7u));
    return (int)blockIdx.y;
  }
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned g = block >> 3;
  const unsigned first = lane <= args.n_items ? (unsigned)args.group_first[lane < kSlots ? lane : 0] : 0xFFFFFFFFu;
  const int idx = __builtin_amdgcn_readfirstlane(__popcll(__ballot(first <= g)) - 1);
  bx = (int)block - ((int)args.group_first[idx] << 3);
  bx = (bx & ~7) | ((bx + (int)args.xcd_rot[idx]) & 7);
  return idx;
}

// ACC 1 (default): Gram matrix on the matrix pipe, 4 waves per SIMD.  ACC 0 (DVO_AMD_ACCUM=valu): 87 fp32 registers per lane,
// 2 waves per SIMD -- kept as the cross-check of the summation (tests/test_gpu_parity.py runs the parity suite's criterion under it).
template <int ACC, int RCP>
__global__ __launch_bounds__(kBlockThreads, ACC == 0 ? 2 : 4) void k_tick(const TickArgs args) {
  int bx;
  const int idx = tick_locate(args, bx);
  tick_body<ACC, RCP>(args.items[idx], args.rcp, bx);
}

// the same kernel behind the small argument block of a tick of at most kMaxSmallItems pairs
template <int RCP>
__global__ __launch_bounds__(kBlockThreads, 4) void k_tick_small(const TickArgsSmall args) {
  int bx;
  const int idx = tick_locate(args, bx);
  tick_body<1, RCP>(args.items[idx], args.rcp, bx);
}

// ---- Q7 in the host-rcpps mode (dense_tracking_impl.cpp:702-706; Q7Rec in dvo_types.h) -------------------------------------------
// One wave per residual pass of a tick launch, after k_tick and before k_finalize on the same stream: V from the pass's block
// records, the last V mod 4 valid pixels from the per-wave counts and the spilled residuals (walked backwards), then those <= 3
// pixels once more through the residual pass's own functions -- lane j takes tail pixel j -- for their Jacobians, and what
// (w_exact - w_table) adds to the pair sums (Q5: an odd rank weights its partner's residual; the tail starts on a multiple of
// four, so the partner is in the tail) and to the 87 moments, in double.  Costs one small dispatch per tick, in this mode only.
template <class Args>
__device__ __forceinline__ void q7_tail_wave(const Args &args) {
  const TickItem &it = args.items[blockIdx.x];
  const int lane = threadIdx.x;
  LevelPairDesc d = load_desc(it);
  d.rcp = args.rcp;
  d.rcp_lds = nullptr, d.dbg_w = nullptr;
  DVO_GLOBAL Q7Rec *const rec = (DVO_GLOBAL Q7Rec *)((DVO_GLOBAL char *)d.ll_partials + 256u * (unsigned)args.q7_off256);
  const gcf recs = (gcf)d.records;
  const int nb = it.res_blocks, steps = item_res_steps(it), seg_px = kStepPx * steps;
  int v = 0;
  for (int b = lane; b < nb; b += kWave) v += (int)f2u(recs[(size_t)b * kRecStride + kRecCount]);
  for (int o = kWave / 2; o; o >>= 1) v += __shfl_xor(v, o);
  const int n_tail = (it.flags & kItemUnitWeights) ? 0 : (v & 3);
  __shared__ double sh_delta[3][3 + kNumAcc];
  __shared__ int sh_idx[3];
  __shared__ float sh_r0[3], sh_r1[3];
  if (lane < 3) {
    sh_idx[lane] = 0, sh_r0[lane] = sh_r1[lane] = 0.0f;
    for (int i = 0; i < 3 + kNumAcc; ++i) sh_delta[lane][i] = 0.0;
  }
  __syncthreads();
  int found = 0;  // wave uniform; tail pixels are found last first
  if (n_tail) {
    const DVO_GLOBAL v2f *const res = (const DVO_GLOBAL v2f *)((it.flags & kItemResBuf) ? d.res[1] : d.res[0]);
    for (int sg = nb * kWavesPerBlock - 1; sg >= 0 && found < n_tail; --sg) {
      const int c = __builtin_amdgcn_readfirstlane((int)f2u(recs[(size_t)(sg >> 2) * kRecStride + kRecWaveCnt + (sg & 3)]));
      if (c == 0) continue;
      for (int st = steps - 1; st >= 0 && found < n_tail; --st) {
        const int first = sg * seg_px + st * kStepPx;
        const v2f r = res[first + lane];
        unsigned long long m = __ballot(r.x == r.x);  // NaN marks an invalid pixel
        while (m && found < n_tail) {
          const int hi = 63 - __builtin_clzll(m);
          const float r0 = __shfl(r.x, hi), r1 = __shfl(r.y, hi);
          if (lane == 0) sh_idx[found] = first + hi, sh_r0[found] = r0, sh_r1[found] = r1;
          m &= ~(1ull << hi);
          ++found;
        }
      }
    }
    __syncthreads();
  }
  const bool mine = lane < n_tail && found == n_tail;
  float w_table = 0.0f, w_exact = 0.0f;
  bool equal = true;
  int my_idx = 0;
  if (n_tail && found == n_tail) {  // (wave uniform; every lane walks a real pixel, lanes past the tail walk tail pixel 0 for nothing)
    const int j = lane < n_tail ? lane : 0;  // ascending rank: tail pixel j was found (n_tail - 1 - j)-th
    const int k = n_tail - 1 - j;
    const unsigned idx = (unsigned)sh_idx[k];
    my_idx = (int)idx;
    float z = ((gcf)d.r_zsel)[idx], ri = ((gcf)d.r_i)[idx], rix = ((gcf)d.r_ix)[idx], riy = ((gcf)d.r_iy)[idx];
    float x = ((gcf)d.tx)[idx] * z, y = ((gcf)d.ty)[idx] * z;
    float kt[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) kt[i] = it.kt[i];
    DVO_OPAQUE(x); DVO_OPAQUE(y); DVO_OPAQUE(z); DVO_OPAQUE(ri); DVO_OPAQUE(rix); DVO_OPAQUE(riy);
    round_toward_zero();
    DVO_OPAQUE(x); DVO_OPAQUE(y); DVO_OPAQUE(z); DVO_OPAQUE(ri); DVO_OPAQUE(rix); DVO_OPAQUE(riy);
    float r0, r1, e2, e3, e4, e5;
    bool ok;
    {
      const Proj p = project_pixel_rtz<1>(kt, d, x, y, z);
      const Gathered g = gather_pixel(d, p.base);
      finish_pixel_rtz<true>(d, p, g, z, ri, rix, riy, r0, r1, e2, e3, e4, e5, ok);
    }
    DVO_OPAQUE(r0); DVO_OPAQUE(r1); DVO_OPAQUE(e2); DVO_OPAQUE(e3); DVO_OPAQUE(e4); DVO_OPAQUE(e5);
    DVO_OPAQUE(x); DVO_OPAQUE(y); DVO_OPAQUE(z);
    round_to_nearest();
    DVO_OPAQUE(r0); DVO_OPAQUE(r1); DVO_OPAQUE(e2); DVO_OPAQUE(e3); DVO_OPAQUE(e4); DVO_OPAQUE(e5);
    DVO_OPAQUE(x); DVO_OPAQUE(y); DVO_OPAQUE(z);
    equal = ok && f2u(r0) == f2u(sh_r0[k]) && f2u(r1) == f2u(sh_r1[k]);
    // the two weights (the distance as computeWeightsSse / computeWeight form it: product by product)
    const float t0 = r0 * it.P[0] + r1 * it.P[1];
    const float t1 = r0 * it.P[2] + r1 * it.P[3];
    const float dd = t0 * r0 + t1 * r1;
    w_table = 7.0f * rcp_host_table(5.0f + dd, d.rcp);
    w_exact = (float)((2.0 + 5.0f) / (5.0f + dd));  // :643, the division in double
    const double dw = mine ? (double)w_exact - (double)w_table : 0.0;
    // Jacobian rows as the residual pass forms them (without the sqrt(w) of the staged form)
    const float iz = __builtin_amdgcn_rcpf(z);
    const float iz2 = iz * iz;
    const float j02 = -x * iz2, j12 = -y * iz2;
    const float j03 = j02 * y, j13 = __builtin_fmaf(j12, y, -1.0f);
    const float j04 = __builtin_fmaf(-j02, x, 1.0f), j14 = -j03;
    const float j05 = -y * iz, j15 = x * iz;
    float Ja[6], Jb[6];
    Ja[0] = e2 * iz;
    Ja[1] = e3 * iz;
    Ja[2] = __builtin_fmaf(e2, j02, e3 * j12);
    Ja[3] = __builtin_fmaf(e2, j03, e3 * j13);
    Ja[4] = __builtin_fmaf(e2, j04, e3 * j14);
    Ja[5] = __builtin_fmaf(e2, j05, e3 * j15);
    Jb[0] = e4 * iz;
    Jb[1] = e5 * iz;
    Jb[2] = __builtin_fmaf(e4, j02, __builtin_fmaf(e5, j12, -1.0f));
    Jb[3] = __builtin_fmaf(e4, j03, __builtin_fmaf(e5, j13, -y));
    Jb[4] = __builtin_fmaf(e4, j04, __builtin_fmaf(e5, j14, x));
    Jb[5] = __builtin_fmaf(e4, j05, e5 * j15);
    if (lane < 3) {
      double *o = sh_delta[lane];
      // pair sums: an even rank weights its own residual, an odd one its partner's (tail pixel 0: the tail starts on a multiple of 4)
      const float a0 = (j & 1) ? sh_r0[n_tail - 1] : r0, a1 = (j & 1) ? sh_r1[n_tail - 1] : r1;
      o[0] = dw * (double)(a0 * a0), o[1] = dw * (double)(a0 * a1), o[2] = dw * (double)(a1 * a1);
      double *acc = o + 3;
      int t = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int c = i; c < 6; ++c, ++t) {
          acc[kAccAA + t] = dw * ((double)Ja[i] * (double)Ja[c]);
          acc[kAccAB + t] = dw * ((double)Ja[i] * (double)Jb[c] + (double)Jb[i] * (double)Ja[c]);
          acc[kAccBB + t] = dw * ((double)Jb[i] * (double)Jb[c]);
        }
        acc[kAccAR0 + i] = dw * ((double)Ja[i] * (double)r0);
        acc[kAccAR1 + i] = dw * ((double)Ja[i] * (double)r1);
        acc[kAccBR0 + i] = dw * ((double)Jb[i] * (double)r0);
        acc[kAccBR1 + i] = dw * ((double)Jb[i] * (double)r1);
      }
    }
  }
  __syncthreads();
  for (int i = lane; i < 3 + kNumAcc; i += kWave) {
    const double sum = (sh_delta[0][i] + sh_delta[1][i]) + sh_delta[2][i];
    if (i < 3) rec->S[i] = sum; else rec->acc[i - 3] = sum;
  }
  const unsigned long long all_equal = __ballot(!mine || equal);
  if (lane < 3) {
    rec->idx[lane] = mine ? my_idx : -1;
    rec->w_table[lane] = mine ? w_table : 0.0f;
    rec->w_exact[lane] = mine ? w_exact : 0.0f;
  }
  if (lane == 0) {
    rec->n_tail = found == n_tail ? n_tail : -1;  // (-1: the counts and the residual buffer disagree -- cannot happen)
    rec->valid = v;
    rec->recomputed_equal = all_equal == ~0ull ? 1 : 0;
  }
}
__global__ __launch_bounds__(kWave) void k_q7_tail(const Q7Args args) { q7_tail_wave(args); }
__global__ __launch_bounds__(kWave) void k_q7_tail_small(const Q7ArgsSmall args) { q7_tail_wave(args); }

__global__ void k_rcp_table_probe(const RcpTable rcp, const float *__restrict__ in, float *__restrict__ out, int n) {
  __shared__ unsigned nib[kRcpNibbleWordsMax];
  if (rcp.nibbles) {
    for (int i = threadIdx.x; i < ((1 << (23 - rcp.shift)) >> 3); i += blockDim.x) nib[i] = rcp.nibbles[i];
    __syncthreads();
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = rcp.nibbles ? rcp_host_nibbles<false>(in[i], nib, rcp.shift, rcp.unit) : rcp_host_table(in[i], rcp);
}
// v_rcp_f32 of every cell midpoint under both rounding modes the residual pass uses it in (the host builds the corrections of the
// nibble form from the round-to-nearest column and refuses the form if the columns differ)
__global__ void k_rcp_midpoint_probe(int k, unsigned *__restrict__ out_rn, unsigned *__restrict__ out_rtz) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << k)) return;
  unsigned cell = i;
  asm volatile("" : "+v"(cell));
  const unsigned a = rcp_midpoint_bits(cell, 23 - k);
  unsigned keep = a;
  asm volatile("" : "+v"(keep));
  round_toward_zero();
  asm volatile("" : "+v"(cell));
  unsigned b = rcp_midpoint_bits(cell, 23 - k);
  asm volatile("" : "+v"(b));
  round_to_nearest();
  out_rn[i] = keep, out_rtz[i] = b;
}

// DVO_AMD_LAUNCH_LOCK=1: a process-wide mutex around every kernel launch.  Only for profiled multi-thread runs: rocprofv3's
// queue interceptor reads past the end of an AQL ring when two host threads publish packets to one hardware queue across
// the ring's wrap (profiles/r03_rocprofv3_sigsegv_root_cause.md); under the lock every doorbell finds exactly one packet.
static std::mutex g_launch_mu;
static bool launch_lock_enabled() {  // (function-local statics: initialised once, thread-safe)
  static const bool on = [] {
    const char *e = getenv("DVO_AMD_LAUNCH_LOCK");
    return e && e[0] == '1';
  }();
  return on;
}
struct LaunchGuard {
  bool held;
  LaunchGuard() {
    held = launch_lock_enabled();
    if (held) g_launch_mu.lock();
  }
  ~LaunchGuard() {
    if (held) g_launch_mu.unlock();
  }
};

// DVO_AMD_ACCUM=valu: 87 register accumulators; default: the Gram matrix on the matrix pipe
int acc_mode() {
  static const int m = [] {
    const char *a = getenv("DVO_AMD_ACCUM");
    return (a && (a[0] == 'v' || a[0] == 'V' || a[0] == '0')) ? 0 : 1;
  }();
  return m;
}

typedef void (*TickKernel)(const TickArgs);
// (the host-rcpps mode exists for the default accumulator only)
static TickKernel pick_tick_kernel(const RcpTable &rcp) {
  return rcp.nibbles ? k_tick<1, 2> : rcp.table ? k_tick<1, 1> : acc_mode() == 0 ? k_tick<0, 0> : k_tick<1, 0>;
}

// A tick's items are at different pyramid levels: the two-dimensional grid (blocks of the largest item x items) launches
// mostly blocks that return at once, and the dispatcher starts only ~4 of them per nanosecond.  When more than half of the
// grid would be such blocks the launch goes out one-dimensional with every item's blocks back to back.
template <class Args>
static int tick_args_layout_impl(Args &args, int max_blocks) {
  constexpr int kSlots = (int)(sizeof(args.group_first) / sizeof(args.group_first[0]));
  unsigned groups = 0;
  for (int i = 0; i < args.n_items; ++i) {
    args.group_first[i] = (uint16_t)groups;
    groups += ((unsigned)args.items[i].res_blocks + args.items[i].ll_blocks + 7u) >> 3;
  }
  for (int i = args.n_items; i < kSlots; ++i) args.group_first[i] = (uint16_t)groups;
  // even out the XCDs' shares (see TickArgs::xcd_rot): greedy over the items in launch order, loads in units of ~0.1 us of block
  // life as the block trace measures it (a residual block: 3.3 us + 2.1 us per step; a likelihood block about half of that)
  static const bool rotate = [] {
    const char *e = getenv("DVO_AMD_XCD_ROTATE");
    return !(e && e[0] == '0');
  }();
  long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < args.n_items; ++i) {
    const TickItem &it = args.items[i];
    const int rb = it.res_blocks, lb = it.ll_blocks;
    const long long w_res = 33 + 21 * item_res_steps(it), w_ll = 40 + 7 * item_ll_steps(it);
    long long phase[8];
    for (int f = 0; f < 8; ++f) {
      const int n_res = (rb >> 3) + (f < (rb & 7) ? 1 : 0);
      const int all = ((rb + lb) >> 3) + (f < ((rb + lb) & 7) ? 1 : 0);
      phase[f] = w_res * n_res + w_ll * (all - n_res);
    }
    int best = 0;
    long long best_max = -1;
    for (int rot = 0; rotate && rot < 8; ++rot) {  // XCD x does the blocks of phase (x + rot) & 7
      long long mx = 0;
      for (int x = 0; x < 8; ++x) mx = std::max(mx, load[x] + phase[(x + rot) & 7]);
      if (best_max < 0 || mx < best_max) best_max = mx, best = rot;
    }
    args.xcd_rot[i] = (uint8_t)best;
    for (int x = 0; x < 8; ++x) load[x] += phase[(x + best) & 7];
  }
  for (int i = args.n_items; i < (int)(sizeof(args.xcd_rot) / sizeof(args.xcd_rot[0])); ++i) args.xcd_rot[i] = 0;
  const long long grid2d = (long long)((max_blocks + 7) & ~7) * args.n_items;
  static const int mode = [] {  // DVO_AMD_COMPACT_GRID=0 / 1 forces a layout (tuning)
    const char *e = getenv("DVO_AMD_COMPACT_GRID");
    return e ? (e[0] == '0' ? 0 : 1) : 2;
  }();
  args.compact = groups > 0 && groups < 65536 && (mode == 1 || (mode == 2 && 2ll * 8 * groups < grid2d)) ? 1 : 0;
  return (int)(groups * 8);
}
int tick_args_layout(TickArgs &args, int max_blocks) { return tick_args_layout_impl(args, max_blocks); }
int tick_args_layout(TickArgsSmall &args, int max_blocks) { return tick_args_layout_impl(args, max_blocks); }

hipError_t launch_tick_small(const TickArgsSmall &args, int max_blocks, hipStream_t stream, hipEvent_t t_start, hipEvent_t t_stop) {
  if (acc_mode() != 1) return hipErrorNotSupported;  // (with or without the reciprocal table: launch_tick decides)
  if (args.n_items <= 0 || max_blocks <= 0) return hipSuccess;
  LaunchGuard guard;
  dim3 grid((unsigned)((max_blocks + 7) & ~7), (unsigned)args.n_items, 1);
  if (args.compact) grid = dim3((unsigned)args.group_first[args.n_items] * 8u, 1, 1);
  if (t_start && t_stop) {
    void *kargs[] = {const_cast<TickArgsSmall *>(&args)};
    const void *kernel = args.rcp.nibbles ? reinterpret_cast<const void *>(&k_tick_small<2>)
                         : args.rcp.table ? reinterpret_cast<const void *>(&k_tick_small<1>)
                                          : reinterpret_cast<const void *>(&k_tick_small<0>);
    const hipError_t e = hipExtLaunchKernel(kernel, grid, dim3(kBlockThreads), kargs, 0, stream, t_start, t_stop, 0);
    if (e != hipSuccess) return e;
  } else if (args.rcp.nibbles) {
    hipLaunchKernelGGL(k_tick_small<2>, grid, dim3(kBlockThreads), 0, stream, args);
  } else if (args.rcp.table) {
    hipLaunchKernelGGL(k_tick_small<1>, grid, dim3(kBlockThreads), 0, stream, args);
  } else {
    hipLaunchKernelGGL(k_tick_small<0>, grid, dim3(kBlockThreads), 0, stream, args);
  }
  return hipGetLastError();
}

hipError_t launch_tick(const TickArgs &args, int max_blocks, hipStream_t stream, hipEvent_t t_start, hipEvent_t t_stop) {
  // the host-rcpps mode is built for the default accumulator only: under DVO_AMD_ACCUM=valu (the cross-check of the summation) it
  // is refused rather than silently run on the matrix pipe (dvo_amd_set_reciprocal_mode refuses first, with the reason)
  if (args.rcp.table && acc_mode() == 0) return hipErrorNotSupported;
  TickKernel kernel = pick_tick_kernel(args.rcp);
  if (args.n_items <= 0 || max_blocks <= 0) return hipSuccess;
  LaunchGuard guard;
  dim3 grid((unsigned)((max_blocks + 7) & ~7), (unsigned)args.n_items, 1);
  if (args.compact) grid = dim3((unsigned)args.group_first[args.n_items] * 8u, 1, 1);
  if (t_start && t_stop) {
    void *kargs[] = {const_cast<TickArgs *>(&args)};
    const hipError_t e = hipExtLaunchKernel(reinterpret_cast<const void *>(kernel), grid, dim3(kBlockThreads), kargs, 0, stream,
                                            t_start, t_stop, 0);
    if (e != hipSuccess) return e;
  } else {
    hipLaunchKernelGGL(kernel, grid, dim3(kBlockThreads), 0, stream, args);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// finalize: one 512-thread block per job.  Sums the 87 moments over the blocks in fp64, combines the ordered part of the
// records (count, S under both start parities, boundary residual/weight) left to right, locates the log-likelihood
// cut, sums the log-likelihood partials.
// ------------------------------------------------------------------------------------------------------------------
#ifndef DVO_FIN_THREADS
#define DVO_FIN_THREADS 512  // measured 384 / 448 / 512 / 640 / 1024: 512 schedules best next to other streams' k_tick blocks
#endif
constexpr int kFinThreads = DVO_FIN_THREADS;
constexpr int kFinSegThreads = 64;    // wave 0: the ordered part of the records (no block barriers inside)
constexpr int kFinAccFirst = 64;      // threads [64, ...): row-chunks x 24 groups of 4 columns (waves 1.. sum the moments)
constexpr int kFinCols = 96;
static_assert(kLevelChunksMax == 16, "finalize_block spells the tree over the 16 moment slices out");
constexpr int kFinCol4 = kFinCols / 4;
constexpr int kFinThreadsBatch = 256;  // the launch of a tick of many pairs: half the block runs 2 % more pairs/s next to other
                                       // streams' k_tick blocks (a small tick keeps 512 threads: its level-0 reduce is bandwidth)
// The reducer follows the level's summation tree (dvo_types.h, kLevelChunks): per chunk a fold in ascending block order, then a
// perfect binary tree over the 16 slices -- the same additions in the same order whether the 512-thread form (a single match())
// or the 256-thread form (a tick of many pairs) runs, and whether the item is a whole level or a band of it.  The 512-thread
// form gives every slice its own threads; in the 256-thread form a thread owns slices 2k and 2k + 1 and adds the two itself
// (the first level of the tree).
template <int NT>
struct FinGeometry {
  static constexpr int kChunks = NT == kFinThreadsBatch ? kLevelChunksMax / 2 : kLevelChunksMax;  // leaf slices the block's threads cover
  static constexpr int kPerThread = kLevelChunksMax / kChunks;                                 // slices per thread
  static_assert(kFinAccFirst + kChunks * kFinCol4 <= NT && kFinAccFirst + kNumAcc + 1 <= NT, "the block holds the summing and the output threads");
};

struct SegRec {
  int c;
  float first_w, l0, l1;
  double s0[3], s1[3];
};

// record of (a followed by b)
__device__ __forceinline__ SegRec seg_combine(const SegRec &a, const SegRec &b) {
  if (b.c == 0) return a;
  if (a.c == 0) return b;
  SegRec o;
  const bool flip = (a.c & 1) != 0;
  const double rxx = (double)a.l0 * a.l0, rxy = (double)a.l0 * a.l1, ryy = (double)a.l1 * a.l1;
  for (int i = 0; i < 3; ++i) {
    o.s0[i] = a.s0[i] + (flip ? b.s1[i] : b.s0[i]);
    o.s1[i] = a.s1[i] + (flip ? b.s0[i] : b.s1[i]);
  }
  // b starts on an odd rank under exactly one of the two hypotheses: there its first pixel pairs with a's last
  double *tgt = flip ? o.s0 : o.s1;
  tgt[0] += (double)b.first_w * rxx, tgt[1] += (double)b.first_w * rxy, tgt[2] += (double)b.first_w * ryy;
  o.c = a.c + b.c;
  o.first_w = a.first_w;
  o.l0 = b.l0, o.l1 = b.l1;
  return o;
}

// Diagnostic only (DVO_AMD_FIN_STAMPS=1): shader-clock stamps of the phases of block 0, read back by
// dvo_amd_debug_finalize_stamps().  Never read by any kernel; no output depends on them.
__device__ unsigned long long g_fin_stamps[8];
#define DVO_FIN_STAMP(i)                                                                  \
  do {                                                                                    \
    if (stamps && threadIdx.x == 0) g_fin_stamps[i] = __builtin_amdgcn_s_memtime(); \
  } while (0)


typedef unsigned v4u __attribute__((ext_vector_type(4)));
// one lane, one instruction, 16 bytes, system scope: the unit every record travels in (FinWire)
// (s_nop: a store of more than 8 bytes reads its data registers over several cycles, and the compiler's hazard recogniser
// does not see through inline assembly -- without it the next vector instruction may overwrite the tail of the data)
__device__ __forceinline__ void st16_system(DVO_GLOBAL void *p, v4u v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ v4u ld16_system(const DVO_GLOBAL void *p) {
  v4u v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// piece i of a record: two 8-byte halves {payload word, tag} (FinWire)
__device__ __forceinline__ v4u wire_piece(const unsigned *record_words, int i, unsigned tag) {
  v4u piece;
  piece.x = record_words[2 * i];
  piece.y = tag;
  piece.z = 2 * i + 1 < kFinWords ? record_words[2 * i + 1] : 0u;
  piece.w = tag;
  return piece;
}

// The one-hop exchange of a tile-sharded pair, run by the waves of k_finalize once the rank's record stands in LDS: wave
// (p mod n_waves) pushes it into rank p's exchange buffer as tagged 16-byte pieces, then every lane waits -- bounded -- for
// "its" pieces of rank p's record of this tick to land in the local buffer and forwards them, tag included, to the host.
// No fence and no ready word anywhere: a piece is valid when its tag is the tick.  Returns false on a timeout.
// The exchange description stays in memory and is read through the constant address space (scalar loads, the peer table
// indexed dynamically): a by-value copy with a dynamically indexed array would live in scratch.
__device__ bool exchange_records(const ExchangeArgs *xa, const unsigned *own_record, unsigned seq, int wave, int n_waves, int lane) {
  const DVO_CONST ExchangeArgs *a = (const DVO_CONST ExchangeArgs *)xa;
  const int n_ranks = a->n_ranks, rank = a->rank;
  const unsigned long long timeout = (unsigned long long)a->timeout_ticks;
  const int slot = (int)(seq & 1u) * n_ranks;
  bool ok = true;
  for (int p = wave; p < n_ranks; p += n_waves) {
    FinWire *dst = a->peers[p] + slot + rank;
    for (int i = lane; i < kFinWirePieces; i += kWave) st16_system((DVO_GLOBAL void *)dst->piece[i], wire_piece(own_record, i, seq));
  }
  const FinWire *local = a->local;
  FinWire *host_records = a->host_records;
  for (int p = wave; p < n_ranks; p += n_waves) {
    const FinWire *in = local + slot + p;
    FinWire *out = host_records + p;
    for (int i = lane; i < kFinWirePieces; i += kWave) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      v4u piece = ld16_system((const DVO_GLOBAL void *)in->piece[i]);
      while (piece.y != seq || piece.w != seq) {  // both halves of the piece carry the tick (FinWire)
        __builtin_amdgcn_s_sleep(8);
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout) {
          ok = false;
          break;
        }
        piece = ld16_system((const DVO_GLOBAL void *)in->piece[i]);
      }
      if (piece.y == seq && piece.w == seq) st16_system((DVO_GLOBAL void *)out->piece[i], piece);
    }
  }
  return ok;
}

// One block per job.  Wave 0 folds the ordered part of the block records (count, S under both start parities, boundary
// residual / weight) left to right and locates the log-likelihood cut; waves 1.. sum the 87 moments and the
// log-likelihood partials in fp64 with 16-byte loads, eight in flight per thread.  Two block barriers in all; the record is
// assembled in LDS and pushed to the pinned host buffer as self-validating 16-byte pieces (FinWire): no fence, no ready word.
// One block reduces one item (see above); `exchange`: the tile-sharded pair's one-hop exchange instead of the hand-off to the host.
// EXCHANGE is a compile-time property of the launch: the batch and single-pair kernels carry no trace of the exchange tail.
template <int NT, bool EXCHANGE>
__device__ __forceinline__ void finalize_block(const FinItem &it, const bool stamps, const ExchangeArgs *exchange, const unsigned xseq) {
  constexpr int kFinChunks = FinGeometry<NT>::kChunks;
  const int t = threadIdx.x;
  __shared__ double sh_acc[kFinChunks][kFinCols];
  __shared__ SegRec sh_seg[kFinSegThreads];
  __shared__ int sh_cnt[kFinSegThreads];
  __shared__ __attribute__((aligned(16))) FinOut sh_out;

  DVO_FIN_STAMP(0);
  const gcf recs = (gcf)it.records;  // indexed by logical block of the level
  const int nb = it.records ? (int)it.n_blocks : 0;
  const int band_lo = (int)it.block_first, band_hi = band_lo + nb;  // the blocks this item reduces
  const int level_nb = (int)it.level_blocks;                        // the blocks the level's chunks partition
  const DVO_GLOBAL Q7Rec *const q7 =
      it.q7_off256 ? (const DVO_GLOBAL Q7Rec *)((const DVO_GLOBAL char *)it.ll_partials + 256u * (unsigned)it.q7_off256) : nullptr;
  if (t < kFinSegThreads) {
    // Wave 0: the ordered part.  Lane 4 c + q owns quarter q of chunk c (what of it lies in the band), folds its records in
    // ascending order, and the 64 lane records are folded by a binary tree -- chunk subtrees first, then the tree over chunks.
    SegRec r;
    r.c = 0, r.first_w = 0.0f, r.l0 = r.l1 = 0.0f;
    for (int i = 0; i < 3; ++i) r.s0[i] = r.s1[i] = 0.0;
    int b_lo, b_end;
    leaf_range(level_nb, 6, t, &b_lo, &b_end);
    b_lo = b_lo > band_lo ? b_lo : band_lo, b_end = b_end < band_hi ? b_end : band_hi;
    if (b_end < b_lo) b_end = b_lo;
    // four records' headers (64 bytes each) in flight at a time; the per-wave counts of the first four are kept in registers
    // for the prefix pass below, so a lane with at most four records (every level up to 256 blocks) reads its headers once
    // (the 256-thread batch form keeps two in flight: it is built for 96 registers, see k_finalize)
    constexpr int kHdr = NT == kFinThreadsBatch ? 2 : 4;
    unsigned cw_first[4][4] = {};
    for (int b0 = b_lo; b0 < b_end; b0 += kHdr) {
      v4f h0[kHdr], h1[kHdr], h2[kHdr], h3[kHdr];
#pragma unroll
      for (int k = 0; k < kHdr; ++k)
        if (b0 + k < b_end) {
          const gcf4 hp = reinterpret_cast<gcf4>(recs + (size_t)(b0 + k) * kRecStride);
          h0[k] = hp[0], h1[k] = hp[1], h2[k] = hp[2], h3[k] = hp[3];
        }
#pragma unroll
      for (int k = 0; k < kHdr; ++k)
        if (b0 + k < b_end) {
          SegRec q;
          q.c = (int)f2u(h0[k].x);
          q.first_w = h0[k].y, q.l0 = h0[k].z, q.l1 = h0[k].w;
          q.s0[0] = h1[k].x, q.s0[1] = h1[k].y, q.s0[2] = h1[k].z;
          q.s1[0] = h1[k].w, q.s1[1] = h2[k].x, q.s1[2] = h2[k].y;
          r = seg_combine(r, q);
          const int j = b0 - b_lo + k;  // the lane's j-th record
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            if (j == jj) cw_first[jj][0] = f2u(h2[k].z), cw_first[jj][1] = f2u(h2[k].w), cw_first[jj][2] = f2u(h3[k].x), cw_first[jj][3] = f2u(h3[k].y);
        }
    }
    const int own_total = r.c;
    sh_seg[t] = r;
    sh_cnt[t] = r.c;
    DVO_WAVE_LDS_SYNC();
    // ordered combine (binary tree over the 64 lanes; an empty record is the identity of seg_combine, bit for bit) and inclusive
    // scan of the per-lane valid counts: wave-local steps
    for (int stride = 1; stride < kFinSegThreads; stride <<= 1) {
      SegRec merged;
      int add = 0;
      const bool do_merge = (t % (2 * stride)) == 0;
      if (do_merge) merged = seg_combine(sh_seg[t], sh_seg[t + stride]);
      if (t >= stride) add = sh_cnt[t - stride];
      DVO_WAVE_LDS_SYNC();
      if (do_merge) sh_seg[t] = merged;
      sh_cnt[t] += add;
      DVO_WAVE_LDS_SYNC();
    }
    if (nb > 0) {
      if (t == 0) {
        const SegRec &a = sh_seg[0];
        sh_out.valid = a.c;
        sh_out.has_res = 1;
        // (host-rcpps mode, whole levels only: what the exact weights of the pass's last V mod 4 pixels add -- k_q7_tail)
        for (int i = 0; i < 3; ++i) sh_out.S[i] = a.s0[i] + (q7 ? q7->S[i] : 0.0), sh_out.S_odd[i] = a.s1[i];
        sh_out.first_w = a.first_w, sh_out.last_r0 = a.l0, sh_out.last_r1 = a.l1;
      }
      // exclusive scan of the valid counts over the band's wave segments: the log-likelihood pass needs each pixel's
      // rank to honour the V % 50 cut (Q6)
      int prefix = sh_cnt[t] - own_total;  // valid pixels of the band before this lane's blocks
      DVO_GLOBAL int *sp = (DVO_GLOBAL int *)it.seg_prefix_out;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (b_lo + k < b_end) {
#pragma unroll
          for (int wv = 0; wv < 4; ++wv) {
            sp[(b_lo + k) * kWavesPerBlock + wv] = prefix;
            prefix += (int)cw_first[k][wv];
          }
        }
      for (int b0 = b_lo + 4; b0 < b_end; b0 += 4) {  // levels with more than 256 blocks: the rest is read again
        v4f h2[4], h3[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (b0 + k < b_end) {
            const gcf4 hp = reinterpret_cast<gcf4>(recs + (size_t)(b0 + k) * kRecStride);
            h2[k] = hp[2], h3[k] = hp[3];
          }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (b0 + k < b_end) {
            const int cw[4] = {(int)f2u(h2[k].z), (int)f2u(h2[k].w), (int)f2u(h3[k].x), (int)f2u(h3[k].y)};
            for (int wv = 0; wv < 4; ++wv) {
              sp[(b0 + k) * kWavesPerBlock + wv] = prefix;
              prefix += cw[wv];
            }
          }
      }
    } else if (t == 0) {
      sh_out.valid = 0;
      sh_out.has_res = 0;
      for (int i = 0; i < 3; ++i) sh_out.S[i] = 0.0, sh_out.S_odd[i] = 0.0;
      sh_out.first_w = sh_out.last_r0 = sh_out.last_r1 = 0.0f;
    }
  } else if (t >= kFinAccFirst && (t - kFinAccFirst) / kFinCol4 < kFinChunks) {
    // moments: 16-byte loads (4 columns).  A chunk's rows are added up in ascending order (the loads of several rows are in
    // flight at a time, the additions stay sequential)
    constexpr int kPer = FinGeometry<NT>::kPerThread;
    constexpr int kInFlight = 4;
    const int c4 = (t - kFinAccFirst) % kFinCol4, tc = (t - kFinAccFirst) / kFinCol4;
    double s[kPer][4];
#pragma unroll
    for (int v = 0; v < kPer; ++v) {
      s[v][0] = s[v][1] = s[v][2] = s[v][3] = 0.0;
      const int c = tc * kPer + v;  // the slice
      if (c4 * 4 < kNumAcc) {
        int lo, hi;
        leaf_range(level_nb, 4, c, &lo, &hi);
        lo = lo > band_lo ? lo : band_lo, hi = hi < band_hi ? hi : band_hi;
        const gcf base = recs + kRecAcc + c4 * 4;
        int b = lo;
        for (; b + kInFlight <= hi; b += kInFlight) {
          v4f r[kInFlight];
#pragma unroll
          for (int k = 0; k < kInFlight; ++k) r[k] = *reinterpret_cast<gcf4>(base + (size_t)(b + k) * kRecStride);
#pragma unroll
          for (int k = 0; k < kInFlight; ++k)
            s[v][0] += (double)r[k].x, s[v][1] += (double)r[k].y, s[v][2] += (double)r[k].z, s[v][3] += (double)r[k].w;
        }
        for (; b < hi; ++b) {
          const v4f r0 = *reinterpret_cast<gcf4>(base + (size_t)b * kRecStride);
          s[v][0] += (double)r0.x, s[v][1] += (double)r0.y, s[v][2] += (double)r0.z, s[v][3] += (double)r0.w;
        }
      } else if (c4 == (kNumAcc + 3) / 4) {
        // the first spare column group sums the log-likelihood partials: the first slice of a chunk folds the chunk's merged
        // blocks in ascending order, the chunk's other slices add exact zeros
        const int m_log2 = (int)it.ll_merge_log2, ll_nb = (int)it.ll_level_blocks;
        const int per = 4 - level_chunks_log2(ll_nb);
        int lo = 0, hi = 0;
        if ((c & ((1 << per) - 1)) == 0) lo = ll_blocks_before(ll_nb, m_log2, c >> per), hi = ll_blocks_before(ll_nb, m_log2, (c >> per) + 1);
        const int ll_lo = (int)it.ll_first, ll_hi = ll_lo + (int)it.n_ll_blocks;
        lo = lo > ll_lo ? lo : ll_lo, hi = hi < ll_hi ? hi : ll_hi;
        const DVO_GLOBAL double *llp = (const DVO_GLOBAL double *)it.ll_partials;
        const DVO_GLOBAL float *qp = (const DVO_GLOBAL float *)((const DVO_GLOBAL double *)it.ll_partials + it.ll_qmax_off);
        float qm = 0.0f;
        for (int b = lo; b < hi; ++b) s[v][0] += llp[b], qm = __builtin_fmaxf(qm, qp[b]);
        s[v][1] = (double)qm;  // (rides in the second column of the spare group)
      }
    }
    if (kPer == 2) {  // chunks 2k and 2k + 1 meet here: the first level of the tree (the 512-thread form adds them after the barrier)
      const bool is_ll = c4 * 4 >= kNumAcc;
      s[0][0] += s[kPer - 1][0], s[0][2] += s[kPer - 1][2], s[0][3] += s[kPer - 1][3];
      s[0][1] = is_ll ? (s[kPer - 1][1] > s[0][1] ? s[kPer - 1][1] : s[0][1]) : s[0][1] + s[kPer - 1][1];
    }
    sh_acc[tc][c4 * 4 + 0] = s[0][0], sh_acc[tc][c4 * 4 + 1] = s[0][1];
    sh_acc[tc][c4 * 4 + 2] = s[0][2], sh_acc[tc][c4 * 4 + 3] = s[0][3];
  }
  __syncthreads();
  DVO_FIN_STAMP(1);

  // chunk sums -> totals: the perfect binary tree over the 16 slices
  constexpr bool kPaired = FinGeometry<NT>::kPerThread == 2;  // the threads added chunks 2k and 2k + 1 already
  auto tree = [&](const int col) __attribute__((always_inline)) {
    double p[kLevelChunksMax / 2];
#pragma unroll
    for (int k = 0; k < kLevelChunksMax / 2; ++k) p[k] = kPaired ? sh_acc[k][col] : sh_acc[kPaired ? k : 2 * k][col] + sh_acc[kPaired ? k : 2 * k + 1][col];
    return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  };
  if (t >= kFinAccFirst && t < kFinAccFirst + kNumAcc)
    sh_out.acc[t - kFinAccFirst] = tree(t - kFinAccFirst) + (q7 ? q7->acc[t - kFinAccFirst] : 0.0);
  if (t == kFinAccFirst + kNumAcc) {
    double qm = 0.0;
    for (int c = 0; c < kFinChunks; ++c) qm = sh_acc[c][89] > qm ? sh_acc[c][89] : qm;
    sh_out.ll_sum = tree(88);
    sh_out.ll_qmax = (float)qm;
    sh_out.has_ll = it.n_ll_blocks > 0 ? 1 : 0;
  }
  __syncthreads();
  DVO_FIN_STAMP(2);

  // ---- tile-sharded pair: the record goes to the peers instead, theirs come back to the host (see exchange_records)
  if (EXCHANGE && exchange && blockIdx.x == 0) {
    __shared__ int sh_bad;
    if (t == 0) sh_bad = 0;
    __syncthreads();
    if (!exchange_records(exchange, reinterpret_cast<const unsigned *>(&sh_out), xseq, t >> 6, NT / kWave, t & (kWave - 1)))
      atomicOr(&sh_bad, 1);
    __syncthreads();
    if (t == 0 && sh_bad)
      __hip_atomic_store(((const DVO_CONST ExchangeArgs *)exchange)->host_seq, xseq | 0x80000000u, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  // ---- publish: every piece of the record goes to the pinned host buffer with the tick's sequence number inside it, in one
  // 16-byte system-scope store per lane.  The host validates piece by piece (FinWire), so nothing is fenced or ordered here.
  static_assert(kFinWirePieces <= 256, "one lane per piece in the smallest finalize block");
  if (t < kFinWirePieces)
    st16_system((DVO_GLOBAL void *)it.out->piece[t], wire_piece(reinterpret_cast<const unsigned *>(&sh_out), t, it.seq));
  if (it.out_dev && t < (int)(sizeof(FinOut) / 16))  // device copy for the collective fallback of a tile-sharded pair
    ((DVO_GLOBAL v4f *)it.out_dev)[t] = reinterpret_cast<const v4f *>(&sh_out)[t];
  DVO_FIN_STAMP(3);
}

// The batch form (256 threads) is built for under 96 registers and 11 KB of LDS (beside 4 x 36.5 KB of k_tick blocks).  Until the
// Gram matrix moved to nine tiles a SIMD full of k_tick waves (4 x 104 registers) had 96 left and a reducer block could start on
// a full CU; with 4 x 128 it takes the place of the next k_tick block that retires there (one retires somewhere every ~20 ns).
template <int NT, bool EXCHANGE>
__global__ __launch_bounds__(NT, NT == kFinThreadsBatch ? 5 : 1) void k_finalize(const FinArgs args) {
  // A reducer block is a few thousand cycles of work on the critical path of its group's tick, started on a CU whose SIMDs are
  // busy with four k_tick waves each: at the default priority it gets a fifth of the issue slots and takes twice as long as alone
  // on the GPU.  (DVO_AMD_FIN_PRIORITY=0 turns this off for an A/B.)
  if (args.pad2 & kFinFlagPriority) __builtin_amdgcn_s_setprio(3);
  finalize_block<NT, EXCHANGE>(args.items[blockIdx.x], args.pad == 0x57A3 && blockIdx.x == 0,
                               EXCHANGE && blockIdx.x == 0 ? args.exchange : nullptr, args.xseq);
}

// the same behind the small argument block of a tick of at most kMaxSmallItems pairs
__global__ __launch_bounds__(kFinThreads) void k_finalize_small(const FinArgsSmall args) {
  finalize_block<kFinThreads, false>(args.items[blockIdx.x], args.pad == 0x57A3 && blockIdx.x == 0, nullptr, 0u);
}

// The one-hop exchange of a ready-made record (device memory): the rare second exchange of a tick of a tile-sharded pair -- the
// band-edge terms of the reference's 50-term likelihood products (dvo_sharded.cpp: sharded_overflow) travel as a FinOut-shaped
// record through the same mapped buffers, tags and generations as the tick records do.
__global__ __launch_bounds__(kFinThreadsBatch) void k_exchange_record(const FinOut *__restrict__ rec, const ExchangeArgs *exchange, unsigned xseq) {
  __shared__ __attribute__((aligned(16))) FinOut sh_out;
  __shared__ int sh_bad;
  const int t = threadIdx.x;
  if (t < (int)(sizeof(FinOut) / 16)) reinterpret_cast<v4f *>(&sh_out)[t] = reinterpret_cast<const v4f *>(rec)[t];
  if (t == 0) sh_bad = 0;
  __syncthreads();
  if (!exchange_records(exchange, reinterpret_cast<const unsigned *>(&sh_out), xseq, t >> 6, kFinThreadsBatch / kWave, t & (kWave - 1)))
    atomicOr(&sh_bad, 1);
  __syncthreads();
  if (t == 0 && sh_bad)
    __hip_atomic_store(((const DVO_CONST ExchangeArgs *)exchange)->host_seq, xseq | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_exchange_record(const FinOut *rec_dev, const ExchangeArgs *exchange_dev, unsigned xseq, hipStream_t stream) {
  LaunchGuard guard;
  hipLaunchKernelGGL(k_exchange_record, dim3(1), dim3(kFinThreadsBatch), 0, stream, rec_dev, exchange_dev, xseq);
  return hipGetLastError();
}

// A dispatch whose only purpose is its name in a profiler's dispatch table: bench.py brackets its single-stream timing pass with
// two of them, and the summaries of the counter passes (dvo_slam_amd/pmc.py) select exactly the k_tick dispatches in between.
__global__ void k_marker(unsigned *sink, unsigned tag) {
  if (sink && tag == 0xFFFFFFFFu) *sink = tag;  // (never true: the kernel has no effect)
}
hipError_t launch_marker(unsigned tag, hipStream_t stream) {
  LaunchGuard guard;
  hipLaunchKernelGGL(k_marker, dim3(1), dim3(64), 0, stream, (unsigned *)nullptr, tag & 0x7FFFFFFFu);
  return hipGetLastError();
}

// Which hardware queue does the runtime run this stream on?  One thread reports HW_ID's PIPE_ID (bits 7:6) and QUEUE_ID (bits
// 26:24) of the wave it runs in (context creation: host::pick_balanced_stream).
__global__ void k_queue_probe(unsigned *out) {
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) __hip_atomic_store(out, 0x80000000u | (((hw >> 6) & 3u) << 3) | ((hw >> 24) & 7u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_queue_probe(unsigned *out_pinned, hipStream_t stream) {
  LaunchGuard guard;
  hipLaunchKernelGGL(k_queue_probe, dim3(1), dim3(64), 0, stream, out_pinned);
  return hipGetLastError();
}

hipError_t launch_rcp_midpoint_probe(int k, unsigned *out_rn, unsigned *out_rtz, hipStream_t stream) {
  LaunchGuard guard;
  hipLaunchKernelGGL(k_rcp_midpoint_probe, dim3((unsigned)(((1 << k) + 255) / 256)), dim3(256), 0, stream, k, out_rn, out_rtz);
  return hipGetLastError();
}

hipError_t launch_rcp_table_probe(const RcpTable &rcp, const float *in, float *out, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  LaunchGuard guard;
  hipLaunchKernelGGL(k_rcp_table_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rcp, in, out, n);
  return hipGetLastError();
}

// (trace builds only) copies up to `capacity` block records (5 x 64-bit words each: t0, t_first, t_end, info | hw << 32, 0) out and
// resets the trace; returns the number of blocks recorded, or -1 in a build without the trace
long long read_block_trace(unsigned long long *out, long long capacity) {
#ifdef DVO_TRACE_BLOCKS
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_trace_n), sizeof(n)) != hipSuccess) return -2;
  const unsigned have = n < kTraceCapacity ? n : kTraceCapacity;
  const unsigned take = (long long)have < capacity ? have : (unsigned)capacity;
  if (out && take) {
    std::vector<BlockTrace> tmp(take);
    if (hipMemcpyFromSymbol(tmp.data(), HIP_SYMBOL(g_trace), sizeof(BlockTrace) * take) != hipSuccess) return -2;
    for (unsigned i = 0; i < take; ++i) {
      out[4 * i] = tmp[i].t0, out[4 * i + 1] = tmp[i].t_first, out[4 * i + 2] = tmp[i].t_end;
      out[4 * i + 3] = (unsigned long long)tmp[i].info | ((unsigned long long)tmp[i].hw << 32);
    }
  }
  const unsigned zero = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_n), &zero, sizeof(zero));
  return (long long)n;
#else
  (void)out, (void)capacity;
  return -1;
#endif
}

hipError_t read_finalize_stamps(unsigned long long out[8]) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fin_stamps), sizeof(unsigned long long) * 8);
}

hipError_t launch_q7_tail(const Q7Args &args, hipStream_t stream) {
  if (args.n_items <= 0) return hipSuccess;
  LaunchGuard guard;
  hipLaunchKernelGGL(k_q7_tail, dim3((unsigned)args.n_items), dim3(kWave), 0, stream, args);
  return hipGetLastError();
}
hipError_t launch_q7_tail_small(const Q7ArgsSmall &args, hipStream_t stream) {
  if (args.n_items <= 0) return hipSuccess;
  LaunchGuard guard;
  hipLaunchKernelGGL(k_q7_tail_small, dim3((unsigned)args.n_items), dim3(kWave), 0, stream, args);
  return hipGetLastError();
}

hipError_t launch_finalize_small(const FinArgsSmall &args, hipStream_t stream) {
  if (args.n_items <= 0) return hipSuccess;
  LaunchGuard guard;
  hipLaunchKernelGGL(k_finalize_small, dim3((unsigned)args.n_items), dim3(kFinThreads), 0, stream, args);
  return hipGetLastError();
}

hipError_t launch_finalize(const FinArgs &args, hipStream_t stream) {
  if (args.n_items <= 0) return hipSuccess;
  LaunchGuard guard;
  if (args.exchange)
    hipLaunchKernelGGL((k_finalize<kFinThreads, true>), dim3((unsigned)args.n_items), dim3(kFinThreads), 0, stream, args);
  else if (args.n_items > kMaxSmallItems)
    hipLaunchKernelGGL((k_finalize<kFinThreadsBatch, false>), dim3((unsigned)args.n_items), dim3(kFinThreadsBatch), 0, stream, args);
  else
    hipLaunchKernelGGL((k_finalize<kFinThreads, false>), dim3((unsigned)args.n_items), dim3(kFinThreads), 0, stream, args);
  return hipGetLastError();
}

hipError_t launch_ll_overflow(const float2 *res, const int *seg_prefix, int seg_first, int n_segs, int seg_px, int rank_offset,
                              int n_px, int cut_rank, int rank_end, const float P[4], unsigned *result_host, hipStream_t stream) {
  if (n_segs <= 0) return hipSuccess;
  LaunchGuard guard;
  const int segs_per_block = 8;
  hipLaunchKernelGGL(k_ll_overflow, dim3((unsigned)((n_segs + segs_per_block - 1) / segs_per_block)), dim3(kWave), 0, stream, res,
                     seg_prefix, seg_first, n_segs, segs_per_block, seg_px, rank_offset, n_px, cut_rank, rank_end, P[0], P[1], P[2], P[3],
                     result_host);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// pyramid construction (all round-to-nearest, no contraction: bit-identical to the CPU restatement)
// ------------------------------------------------------------------------------------------------------------------

// pyrDownMeanSmooth<float> for intensity and pyrDownSubsample<float> for depth, rgbd_image.cpp:38-55,127-139
__global__ void k_pyr_down(const float *__restrict__ ip, const float *__restrict__ zp, int wp, float *__restrict__ io,
                           float *__restrict__ zo, int w, int h) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w || y >= h) return;
  const float *r0 = ip + (size_t)(2 * y) * wp + 2 * x;
  const float *r1 = r0 + wp;
  io[(size_t)y * w + x] = (((r0[0] + r0[1]) + r1[0]) + r1[1]) / 4.0f;
  zo[(size_t)y * w + x] = zp[(size_t)(2 * y) * wp + 2 * x];
}

hipError_t launch_pyr_down(const float *i_prev, const float *z_prev, int w_prev, float *i_out, float *z_out, int w, int h,
                           hipStream_t stream) {
  LaunchGuard guard;
  dim3 grid((unsigned)((w + 255) / 256), (unsigned)h);
  hipLaunchKernelGGL(k_pyr_down, grid, dim3(256), 0, stream, i_prev, z_prev, w_prev, i_out, z_out, w, h);
  return hipGetLastError();
}

// calculateDerivativeX/Y (rgbd_image.cpp:419-472, rgbd_image_sse.cpp:241-284), buildAccelerationStructure (:534-543) in
// the gather layout, the planar copies the reference side reads, and the pixel-ray tables of RgbdCamera (:186-204)
__global__ void k_level_planes(const float *__restrict__ ip, const float *__restrict__ zp, int w, int h, int n_pad, float fx,
                               float fy, float ox, float oy, float4 *__restrict__ c_a, float2 *__restrict__ c_b,
                               float *__restrict__ r_i, float *__restrict__ r_ix, float *__restrict__ r_iy,
                               float *__restrict__ tx, float *__restrict__ ty, int ty_len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = w * h;
  if (i < w) tx[i] = ((float)i - ox) / fx;
  if (i < ty_len) ty[i] = ((float)i - oy) / fy;
  if (i >= n_pad) return;
  if (i >= n) {
    r_i[i] = 0.0f, r_ix[i] = 0.0f, r_iy[i] = 0.0f;
    return;
  }
  const int y = i / w, x = i - y * w;
  const int xp = x > 0 ? x - 1 : 0, xn = x < w - 1 ? x + 1 : w - 1;
  const int yp = y > 0 ? y - 1 : 0, yn = y < h - 1 ? y + 1 : h - 1;
  const float I = ip[i], Z = zp[i];
  const float Ix = (ip[y * w + xn] - ip[y * w + xp]) * 0.5f;
  const float Iy = (ip[yn * w + x] - ip[yp * w + x]) * 0.5f;
  const float Zx = (zp[y * w + xn] - zp[y * w + xp]) * 0.5f;
  const float Zy = (zp[yn * w + x] - zp[yp * w + x]) * 0.5f;
  c_a[i] = make_float4(I, Z, Ix, Iy);
  c_b[i] = make_float2(Zx, Zy);
  r_i[i] = I, r_ix[i] = Ix, r_iy[i] = Iy;
}

hipError_t launch_level_planes(const float *i_plane, const float *z_plane, int w, int h, int n_pad, float fx, float fy,
                               float ox, float oy, float4 *c_a, float2 *c_b, float *r_i, float *r_ix, float *r_iy,
                               float *tx, float *ty, int ty_len, hipStream_t stream) {
  LaunchGuard guard;
  int span = n_pad > ty_len ? n_pad : ty_len;
  if (w > span) span = w;
  hipLaunchKernelGGL(k_level_planes, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, stream, i_plane, z_plane, w, h,
                     n_pad, fx, fy, ox, oy, c_a, c_b, r_i, r_ix, r_iy, tx, ty, ty_len);
  return hipGetLastError();
}

// ValidPointAndGradientThresholdPredicate::isPointOk (point_selection.h:63-66) applied in place: zsel = z where the pixel
// is selected, NaN elsewhere (and in the padding).  Count and last selected index (selectPointsFromImage's running output
// pointer, point_selection.cpp:119-152) leave every block as one {count, last index} pair -- no atomics: 10 000 same-address
// atomics of a 640x480 level serialise to 100 us, the kernel streams its 9 MB in a tenth of that.
__global__ __launch_bounds__(256) void k_select(const float *__restrict__ zp, const float4 *__restrict__ c_a,
                                                const float2 *__restrict__ c_b, int n, int n_pad, float ti, float td,
                                                float *__restrict__ zsel, int2 *__restrict__ block_partials) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float out = u2f(0x7fc00000u);
  bool ok = false;
  if (i < n) {
    const float z = zp[i];
    const float4 a = c_a[i];
    const float2 b = c_b[i];
    ok = (z == z) && (b.x == b.x) && (b.y == b.y) &&
         (__builtin_fabsf(a.z) > ti || __builtin_fabsf(a.w) > ti || __builtin_fabsf(b.x) > td || __builtin_fabsf(b.y) > td);
    if (ok) out = z;
  }
  if (i < n_pad) zsel[i] = out;
  __shared__ int sh_cnt[4], sh_last[4];
  const unsigned long long m = __ballot(ok);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh_cnt[wave] = __popcll(m);
    sh_last[wave] = m ? (int)(blockIdx.x * blockDim.x) + wave * kWave + (63 - __builtin_clzll(m)) : -1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0, last = -1;
    for (int w = 0; w < 4; ++w) c += sh_cnt[w], last = sh_last[w] > last ? sh_last[w] : last;
    block_partials[blockIdx.x] = make_int2(c, last);
  }
}

// one block: the level's count and last selected index from the block partials, then Q3 -- computeResidualsSse walks the
// selection two points at a time and never looks at an odd trailing point (dense_tracking_impl.cpp:169-171)
__global__ __launch_bounds__(256) void k_select_finish(const int2 *__restrict__ block_partials, int n_blocks, float *zsel,
                                                       int *__restrict__ counters) {
  int c = 0, last = -1;
  for (int b = threadIdx.x; b < n_blocks; b += 256) {
    const int2 p = block_partials[b];
    c += p.x, last = p.y > last ? p.y : last;
  }
  __shared__ int sh_c[256], sh_l[256];
  sh_c[threadIdx.x] = c, sh_l[threadIdx.x] = last;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sh_c[threadIdx.x] += sh_c[threadIdx.x + s];
      sh_l[threadIdx.x] = sh_l[threadIdx.x + s] > sh_l[threadIdx.x] ? sh_l[threadIdx.x + s] : sh_l[threadIdx.x];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    counters[0] = sh_c[0], counters[1] = sh_l[0];
    if ((sh_c[0] & 1) && sh_l[0] >= 0) zsel[sh_l[0]] = u2f(0x7fc00000u);
  }
}

hipError_t launch_select(const float *z_plane, const float4 *c_a, const float2 *c_b, int n, int n_pad, float ti, float td,
                         float *zsel, int *counters, int2 *block_partials, hipStream_t stream) {
  LaunchGuard guard;
  const int n_blocks = (n_pad + 255) / 256;
  hipLaunchKernelGGL(k_select, dim3((unsigned)n_blocks), dim3(256), 0, stream, z_plane, c_a, c_b, n, n_pad, ti, td, zsel,
                     block_partials);
  hipLaunchKernelGGL(k_select_finish, dim3(1), dim3(256), 0, stream, (const int2 *)block_partials, n_blocks, zsel, counters);
  return hipGetLastError();
}

// The selected pixels of a level in scan order, compacted (round 5, end): what the residual pass walks.  The reference compacts
// too -- PointSelection::select fills a dense array of the selected points (point_selection.cpp:119-152) and computeResidualsSse
// walks that array -- so the pass's cost follows the selection, not the image.  One exclusive prefix over k_select's per-block
// counts (one block), then every 256-pixel block writes its selected pixels behind the blocks before it: depth, intensity and its
// derivatives, the pixel's ray (tx[column], ty[row]: the residual pass no longer tracks rows and columns) and the pixel index (the
// debug entries scatter per-point results back to the image with it).  Entries from the even point count (Q3, see
// k_select_finish) up to n_pad are padding: depth NaN, everything else zero.
__global__ __launch_bounds__(256) void k_select_prefix(const int2 *__restrict__ block_partials, int n_blocks, int *__restrict__ prefix) {
  const int chunk = (n_blocks + 255) / 256;
  const int b0 = (int)threadIdx.x * chunk, b1 = b0 + chunk < n_blocks ? b0 + chunk : n_blocks;
  int c = 0;
  for (int b = b0; b < b1; ++b) c += block_partials[b].x;
  __shared__ int sh[256];
  sh[threadIdx.x] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int t = 0; t < 256; ++t) {
      const int v = sh[t];
      sh[t] = run, run += v;
    }
  }
  __syncthreads();
  int run = sh[threadIdx.x];
  for (int b = b0; b < b1; ++b) prefix[b] = run, run += block_partials[b].x;
}

__global__ __launch_bounds__(256) void k_compact(const float *__restrict__ zsel, const float *__restrict__ r_i, const float *__restrict__ r_ix,
                                                 const float *__restrict__ r_iy, const float *__restrict__ tx, const float *__restrict__ ty,
                                                 int w, int n, int n_pad, const int *__restrict__ prefix, const int *__restrict__ counters,
                                                 float *__restrict__ cz, float *__restrict__ ci, float *__restrict__ cix,
                                                 float *__restrict__ ciy, float *__restrict__ ctx, float *__restrict__ cty,
                                                 int *__restrict__ cpix) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_pts = counters[0] & ~1;  // Q3: an odd trailing point is never looked at (its zsel entry is NaN already)
  const float z = i < n ? zsel[i] : u2f(0x7fc00000u);
  const bool ok = z == z;
  const unsigned long long m = __ballot(ok);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  __shared__ int sh_cnt[4];
  if (lane == 0) sh_cnt[wave] = __popcll(m);
  __syncthreads();
  int before = prefix[blockIdx.x];
  for (int k = 0; k < wave; ++k) before += sh_cnt[k];
  if (ok) {
    const int p = before + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    const int row = i / w, col = i - row * w;
    cz[p] = z, ci[p] = r_i[i], cix[p] = r_ix[i], ciy[p] = r_iy[i], ctx[p] = tx[col], cty[p] = ty[row], cpix[p] = i;
  }
  if (i >= n_pts && i < n_pad) cz[i] = u2f(0x7fc00000u), ci[i] = 0.0f, cix[i] = 0.0f, ciy[i] = 0.0f, ctx[i] = 0.0f, cty[i] = 0.0f, cpix[i] = -1;
}

hipError_t launch_compact(const float *zsel, const float *r_i, const float *r_ix, const float *r_iy, const float *tx, const float *ty, int w,
                          int n, int n_pad, const int2 *block_partials, int *prefix, const int *counters, float *cz, float *ci, float *cix,
                          float *ciy, float *ctx, float *cty, int *cpix, hipStream_t stream) {
  LaunchGuard guard;
  const int n_blocks = (n_pad + 255) / 256;
  hipLaunchKernelGGL(k_select_prefix, dim3(1), dim3(256), 0, stream, block_partials, n_blocks, prefix);
  hipLaunchKernelGGL(k_compact, dim3((unsigned)n_blocks), dim3(256), 0, stream, zsel, r_i, r_ix, r_iy, tx, ty, w, n, n_pad,
                     (const int *)prefix, counters, cz, ci, cix, ciy, ctx, cty, cpix);
  return hipGetLastError();
}

__global__ void k_copy_strided(const float *__restrict__ src, int stride, float *__restrict__ dst, int w, int h) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x < w && y < h) dst[(size_t)y * w + x] = src[(size_t)y * stride + x];
}

hipError_t launch_copy_strided(const float *src, int stride, float *dst, int w, int h, hipStream_t stream) {
  LaunchGuard guard;
  hipLaunchKernelGGL(k_copy_strided, dim3((unsigned)((w + 255) / 256), (unsigned)h), dim3(256), 0, stream, src, stride, dst, w,
                     h);
  return hipGetLastError();
}

// Frame ingest: raw sensor frame -> the two float base planes of level 0.
//   depth: 0 -> NaN, else (float)raw * scale           (SurfacePyramid::convertRawDepthImageSse, surface_pyramid.cpp:65-105)
//   gray : (B*1868 + G*9617 + R*4899 + 2^13) >> 14      (cv::cvtColor(CV_BGR2GRAY) 8-bit rule + convertTo(CV_32F),
//          or the byte itself for 1-channel input        benchmark_slam.cpp:60-68, camera_dense_tracking.cpp:219-229)
// One thread converts 4 consecutive pixels of a row (level widths are multiples of 4): 12 + 8 bytes in, 2 x 16 bytes out.
__global__ void k_ingest(const unsigned char *__restrict__ img, int channels, int img_stride_bytes,
                         const unsigned short *__restrict__ raw_z, int z_stride, float z_scale, float *__restrict__ i_plane,
                         float *__restrict__ z_plane, int w, int h) {
  const int x4 = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x4 * 4 >= w || y >= h) return;
  const unsigned char *ip = img + (size_t)y * img_stride_bytes + (size_t)x4 * 4 * channels;
  unsigned char px[12];
  const int nb = 4 * channels;
  if ((((size_t)ip) & 3) == 0) {
    const unsigned *ip4 = (const unsigned *)ip;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k < channels) {
        const unsigned v = ip4[k];
        px[4 * k] = (unsigned char)v, px[4 * k + 1] = (unsigned char)(v >> 8), px[4 * k + 2] = (unsigned char)(v >> 16),
               px[4 * k + 3] = (unsigned char)(v >> 24);
      }
  } else {
#pragma unroll
    for (int k = 0; k < 12; ++k)
      if (k < nb) px[k] = ip[k];
  }
  float g[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (channels == 3) {
      const int b = px[3 * k], gg = px[3 * k + 1], r = px[3 * k + 2];
      g[k] = (float)((b * 1868 + gg * 9617 + r * 4899 + (1 << 13)) >> 14);
    } else {
      g[k] = (float)px[k];
    }
  }
  const unsigned short *zp = raw_z + (size_t)y * z_stride + (size_t)x4 * 4;
  unsigned short zr[4];
  if ((((size_t)zp) & 7) == 0) {
    const uint2 v = *(const uint2 *)zp;
    zr[0] = (unsigned short)v.x, zr[1] = (unsigned short)(v.x >> 16), zr[2] = (unsigned short)v.y,
    zr[3] = (unsigned short)(v.y >> 16);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) zr[k] = zp[k];
  }
  float z[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) z[k] = zr[k] == 0 ? __builtin_nanf("") : (float)zr[k] * z_scale;
  const size_t o = (size_t)y * w + (size_t)x4 * 4;
  *(float4 *)(i_plane + o) = make_float4(g[0], g[1], g[2], g[3]);
  *(float4 *)(z_plane + o) = make_float4(z[0], z[1], z[2], z[3]);
}

hipError_t launch_ingest(const unsigned char *img, int channels, int img_stride_bytes, const unsigned short *raw_z,
                         int z_stride, float z_scale, float *i_plane, float *z_plane, int w, int h, hipStream_t stream) {
  LaunchGuard guard;
  const int wq = w / 4;
  hipLaunchKernelGGL(k_ingest, dim3((unsigned)((wq + 63) / 64), (unsigned)h), dim3(64), 0, stream, img, channels,
                     img_stride_bytes, raw_z, z_stride, z_scale, i_plane, z_plane, w, h);
  return hipGetLastError();
}

__global__ void k_mask_from_zsel(const float *__restrict__ zsel, int n, int last_dropped, unsigned char *__restrict__ mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) mask[i] = (zsel[i] == zsel[i] || i == last_dropped) ? 1 : 0;
}

hipError_t launch_mask_from_zsel(const float *zsel, int n, int last_dropped, unsigned char *mask, hipStream_t stream) {
  LaunchGuard guard;
  hipLaunchKernelGGL(k_mask_from_zsel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, zsel, n, last_dropped, mask);
  return hipGetLastError();
}

__global__ void k_unpack_plane(const float4 *__restrict__ c_a, const float2 *__restrict__ c_b, int plane, int n,
                               float *__restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v;
  switch (plane) {
    case 0: v = c_a[i].x; break;
    case 1: v = c_a[i].y; break;
    case 2: v = c_a[i].z; break;
    case 3: v = c_a[i].w; break;
    case 4: v = c_b[i].x; break;
    default: v = c_b[i].y; break;
  }
  dst[i] = v;
}

hipError_t launch_unpack_plane(const float4 *c_a, const float2 *c_b, int plane, int n, float *dst, hipStream_t stream) {
  LaunchGuard guard;
  hipLaunchKernelGGL(k_unpack_plane, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, c_a, c_b, plane, n, dst);
  return hipGetLastError();
}

}  // namespace dvo_amd
