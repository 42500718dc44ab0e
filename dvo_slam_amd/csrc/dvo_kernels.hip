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

__device__ __forceinline__ void round_toward_zero() { __builtin_amdgcn_s_setreg(DVO_HWREG_MODE_FP32_ROUND, 3); }
__device__ __forceinline__ void round_to_nearest() { __builtin_amdgcn_s_setreg(DVO_HWREG_MODE_FP32_ROUND, 0); }

__device__ __forceinline__ float u2f(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned f2u(float f) { return __builtin_bit_cast(unsigned, f); }

// DPP move: returns 0 in lanes the row mask disables or whose source is invalid
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_read(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}

// Sum over the 64 lanes of a wave; the total is valid in lane 63.
// RMODE 0: ds_bpermute butterfly (reference implementation), RMODE 1: DPP row operations.
template <int RMODE>
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  if (RMODE == 0) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
  } else {
    v += dpp_read<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_read<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_read<0x141, 0xF>(v);  // row_half_mirror
    v += dpp_read<0x140, 0xF>(v);  // row_mirror: every lane holds its row's sum
    v += dpp_read<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_read<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
    return v;
  }
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

// trunc(1/z): the quotient rounded toward zero, independent of the current rounding mode.
// v_rcp_f32 is accurate to 1 ulp; the exact sign of 1 - |z| c (one fma, never rounds across zero) picks the largest
// candidate c with c <= 1/|z| among r-2ulp .. r+2ulp.
__device__ __forceinline__ float rcp_toward_zero(float z) {
  const float az = __builtin_fabsf(z);
  if (!(az >= 1.1754944e-38f && az <= 8.5070592e+37f)) return 1.0f / z;  // 0, denormal, huge, inf, NaN: never in bounds
  const unsigned r = f2u(__builtin_amdgcn_rcpf(az));
  const float c_m1 = u2f(r - 1), c_0 = u2f(r), c_p1 = u2f(r + 1), c_p2 = u2f(r + 2);
  const bool ok_m1 = __builtin_fmaf(-az, c_m1, 1.0f) >= 0.0f;
  const bool ok_0 = __builtin_fmaf(-az, c_0, 1.0f) >= 0.0f;
  const bool ok_p1 = __builtin_fmaf(-az, c_p1, 1.0f) >= 0.0f;
  const bool ok_p2 = __builtin_fmaf(-az, c_p2, 1.0f) >= 0.0f;
  const unsigned pick = ok_p2 ? r + 2 : ok_p1 ? r + 1 : ok_0 ? r : ok_m1 ? r - 1 : r - 2;
  return __builtin_copysignf(u2f(pick), z);
}

// 16 bytes = two adjacent {Zx,Zy} pairs; only 8-byte aligned
struct __attribute__((packed, aligned(8))) f4_align8 {
  float a, b, c, d;
};

// ------------------------------------------------------------------------------------------------------------------
// residual pass
// ------------------------------------------------------------------------------------------------------------------

struct PixelIn {   // live across the RN -> RTZ switch
  float x[kPxPerLane], y[kPxPerLane], z[kPxPerLane];
  float ri[kPxPerLane], rix[kPxPerLane], riy[kPxPerLane];
};
struct PixelOut {  // live across the RTZ -> RN switch
  float r0[kPxPerLane], r1[kPxPerLane], e2[kPxPerLane], e3[kPxPerLane], e4[kPxPerLane], e5[kPxPerLane];
  bool valid[kPxPerLane];
};

// computeResidualsSse for one reference pixel, dense_tracking_impl.cpp:171-294.  Runs in round-toward-zero.
__device__ __forceinline__ void warp_pixel_rtz(const TickItem &it, const LevelPairDesc &d, float x, float y, float z, float ri, float rix, float riy,
                                               float &r0, float &r1, float &e2, float &e3, float &e4, float &e5,
                                               bool &valid) {
  const float *kt = it.kt;
  // hadd(hadd()) adds lanes (0,1) and (2,3) first (:178-188); the point's w is 1
  const float sx = (kt[0] * x + kt[1] * y) + (kt[2] * z + kt[3]);
  const float sy = (kt[4] * x + kt[5] * y) + (kt[6] * z + kt[7]);
  const float sz = (kt[8] * x + kt[9] * y) + (kt[10] * z + kt[11]);
  const float rz = rcp_toward_zero(sz);
  const float u = sx * rz, v = sy * rz;
  valid = false;
  r0 = r1 = e2 = e3 = e4 = e5 = 0.0f;
  // 0 <= u <= w-2 and 0 <= v <= h-2 (:160-161,203); NaN compares false
  if ((u >= 0.0f) && (u <= d.ub_x) && (v >= 0.0f) && (v <= d.ub_y)) {
    const int iu = (int)u, iv = (int)v;  // truncation == _mm_cvtps_epi32 under RTZ (:195)
    const float w1u = u - (float)iu, w1v = v - (float)iv;
    const float w0u = 1.0f - w1u, w0v = 1.0f - w1v;
    const int base = iv * d.w + iu;
    const float4 a00 = d.c_a[base], a10 = d.c_a[base + 1];
    const float4 a01 = d.c_a[base + d.w], a11 = d.c_a[base + d.w + 1];
    const f4_align8 b0 = *reinterpret_cast<const f4_align8 *>(d.c_b + base);
    const f4_align8 b1 = *reinterpret_cast<const f4_align8 *>(d.c_b + base + d.w);
    // bilinear blend, per channel: w0v*(w0u*c00 + w1u*c10) + w1v*(w0u*c01 + w1u*c11)  (:227-258)
#define DVO_BLEND(c00, c10, c01, c11) ((w0v * (w0u * (c00) + w1u * (c10))) + (w1v * (w0u * (c01) + w1u * (c11))))
    const float ci = DVO_BLEND(a00.x, a10.x, a01.x, a11.x);
    const float cz = DVO_BLEND(a00.y, a10.y, a01.y, a11.y);
    const float cix = DVO_BLEND(a00.z, a10.z, a01.z, a11.z);
    const float ciy = DVO_BLEND(a00.w, a10.w, a01.w, a11.w);
    const float czx = DVO_BLEND(b0.a, b0.c, b1.a, b1.c);
    const float czy = DVO_BLEND(b0.b, b0.d, b1.b, b1.d);
#undef DVO_BLEND
    // any NaN among the blended channels rejects the point (:261); channels 6,7 are always 0
    const bool has_nan = (ci != ci) || (cz != cz) || (cix != cix) || (ciy != ciy) || (czx != czx) || (czy != czy);
    if (!has_nan) {
      // e = wcur * cur + wref * ref', ref' = {I, transformed depth, Ix, Iy} (:269-271)
      const float t0 = d.wc[0] * ci + d.wr[0] * ri;
      const float t1 = d.wc[1] * cz + d.wr[1] * sz;
      // occlusion test (:275) with depthStdDevZ (:122-128) of the reference depth
      float s = z - 0.4f;
      s = 0.0012f + (0.0019f * s) * s;
      if (t1 > -20.0f * s) {
        r0 = t0;
        r1 = t1;
        e2 = d.wc[2] * cix + d.wr[2] * rix;
        e3 = d.wc[3] * ciy + d.wr[3] * riy;
        e4 = d.wc[4] * czx;  // wref is 0 for the depth derivatives (dense_tracking.cpp:217-220)
        e5 = d.wc[5] * czy;
        valid = true;
      }
    }
  }
}

template <int RMODE>
__device__ void residual_pass(const TickItem &it, const LevelPairDesc &d, const int lb) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int seg = lb * kWavesPerBlock + wave;
  const int w = d.w;
  int idx = seg * (kSegPxPerRound * it.res_rounds) + lane * kPxPerLane;
  int prow = idx / w;
  int pcol = idx - prow * w;

  float acc[kNumAcc];
#pragma unroll
  for (int i = 0; i < kNumAcc; ++i) acc[i] = 0.0f;
  float S0[3] = {0.0f, 0.0f, 0.0f}, S1[3] = {0.0f, 0.0f, 0.0f};
  float first_w = 0.0f;
  int run_count = 0;                        // wave uniform
  float carry_r0 = 0.0f, carry_r1 = 0.0f;   // wave uniform: residual of the last valid pixel of earlier rounds
  bool carry_has = false;

  const unsigned long long below = (1ull << lane) - 1ull;
  const bool unit_w = it.unit_weights != 0;
  const float P0 = it.P_res[0], P1 = it.P_res[1], P2 = it.P_res[2], P3 = it.P_res[3];

  for (int round = 0; round < it.res_rounds; ++round) {
    // ---- round-to-nearest: reference point = pixel ray * depth (RgbdCamera::buildPointCloud, rgbd_image.cpp:245-262)
    const float4 zs = *reinterpret_cast<const float4 *>(d.r_zsel + idx);
    const float4 rI = *reinterpret_cast<const float4 *>(d.r_i + idx);
    const float4 rIx = *reinterpret_cast<const float4 *>(d.r_ix + idx);
    const float4 rIy = *reinterpret_cast<const float4 *>(d.r_iy + idx);
    const float4 txv = *reinterpret_cast<const float4 *>(d.tx + pcol);
    const float tyv = d.ty[prow < d.h ? prow : d.h - 1];
    PixelIn pin;
    pin.z[0] = zs.x, pin.z[1] = zs.y, pin.z[2] = zs.z, pin.z[3] = zs.w;
    pin.x[0] = txv.x * zs.x, pin.x[1] = txv.y * zs.y, pin.x[2] = txv.z * zs.z, pin.x[3] = txv.w * zs.w;
    pin.y[0] = tyv * zs.x, pin.y[1] = tyv * zs.y, pin.y[2] = tyv * zs.z, pin.y[3] = tyv * zs.w;
    pin.ri[0] = rI.x, pin.ri[1] = rI.y, pin.ri[2] = rI.z, pin.ri[3] = rI.w;
    pin.rix[0] = rIx.x, pin.rix[1] = rIx.y, pin.rix[2] = rIx.z, pin.rix[3] = rIx.w;
    pin.riy[0] = rIy.x, pin.riy[1] = rIy.y, pin.riy[2] = rIy.z, pin.riy[3] = rIy.w;

    // ---- switch to round-toward-zero
#pragma unroll
    for (int k = 0; k < kPxPerLane; ++k) {
      DVO_OPAQUE(pin.x[k]); DVO_OPAQUE(pin.y[k]); DVO_OPAQUE(pin.z[k]);
      DVO_OPAQUE(pin.ri[k]); DVO_OPAQUE(pin.rix[k]); DVO_OPAQUE(pin.riy[k]);
    }
    round_toward_zero();
#pragma unroll
    for (int k = 0; k < kPxPerLane; ++k) {
      DVO_OPAQUE(pin.x[k]); DVO_OPAQUE(pin.y[k]); DVO_OPAQUE(pin.z[k]);
      DVO_OPAQUE(pin.ri[k]); DVO_OPAQUE(pin.rix[k]); DVO_OPAQUE(pin.riy[k]);
    }
    PixelOut po;
#pragma unroll
    for (int k = 0; k < kPxPerLane; ++k)
      warp_pixel_rtz(it, d, pin.x[k], pin.y[k], pin.z[k], pin.ri[k], pin.rix[k], pin.riy[k], po.r0[k], po.r1[k], po.e2[k],
                     po.e3[k], po.e4[k], po.e5[k], po.valid[k]);
    // ---- back to round-to-nearest
#pragma unroll
    for (int k = 0; k < kPxPerLane; ++k) {
      DVO_OPAQUE(po.r0[k]); DVO_OPAQUE(po.r1[k]); DVO_OPAQUE(po.e2[k]); DVO_OPAQUE(po.e3[k]);
      DVO_OPAQUE(po.e4[k]); DVO_OPAQUE(po.e5[k]);
      DVO_OPAQUE(pin.x[k]); DVO_OPAQUE(pin.y[k]); DVO_OPAQUE(pin.z[k]);
    }
    round_to_nearest();
#pragma unroll
    for (int k = 0; k < kPxPerLane; ++k) {
      DVO_OPAQUE(po.r0[k]); DVO_OPAQUE(po.r1[k]); DVO_OPAQUE(po.e2[k]); DVO_OPAQUE(po.e3[k]);
      DVO_OPAQUE(po.e4[k]); DVO_OPAQUE(po.e5[k]);
      DVO_OPAQUE(pin.x[k]); DVO_OPAQUE(pin.y[k]); DVO_OPAQUE(pin.z[k]);
    }

    // spill the residuals of this iteration for the log-likelihood pass (NaN marks an invalid pixel)
    {
      const float qnan = u2f(0x7fc00000u);
      float4 s0, s1;
      s0.x = po.valid[0] ? po.r0[0] : qnan, s0.y = po.valid[0] ? po.r1[0] : qnan;
      s0.z = po.valid[1] ? po.r0[1] : qnan, s0.w = po.valid[1] ? po.r1[1] : qnan;
      s1.x = po.valid[2] ? po.r0[2] : qnan, s1.y = po.valid[2] ? po.r1[2] : qnan;
      s1.z = po.valid[3] ? po.r0[3] : qnan, s1.w = po.valid[3] ? po.r1[3] : qnan;
      float4 *dst = reinterpret_cast<float4 *>(d.res[it.res_buf] + idx);
      dst[0] = s0;
      dst[1] = s1;
    }

    // ---- rank of every valid pixel in scan order within this wave's segment (needed by the pair quirk Q5)
    unsigned long long B[kPxPerLane];
    int before = 0, round_total = 0;
    unsigned long long any_mask = 0;
#pragma unroll
    for (int k = 0; k < kPxPerLane; ++k) {
      B[k] = __ballot(po.valid[k]);
      before += __popcll(B[k] & below);
      round_total += __popcll(B[k]);
      any_mask |= B[k];
    }
    float lane_last_r0 = 0.0f, lane_last_r1 = 0.0f;
#pragma unroll
    for (int k = 0; k < kPxPerLane; ++k)
      if (po.valid[k]) lane_last_r0 = po.r0[k], lane_last_r1 = po.r1[k];
    // residual of the valid pixel that precedes this lane's first one
    const unsigned long long prev_lanes = any_mask & below;
    const int src_lane = prev_lanes ? 63 - __clzll((long long)prev_lanes) : 0;
    const float sh_r0 = __shfl(lane_last_r0, src_lane, 64), sh_r1 = __shfl(lane_last_r1, src_lane, 64);
    float prev_r0 = prev_lanes ? sh_r0 : carry_r0;
    float prev_r1 = prev_lanes ? sh_r1 : carry_r1;
    bool prev_has = prev_lanes ? true : carry_has;
    int rank = run_count + before;

#pragma unroll
    for (int k = 0; k < kPxPerLane; ++k) {
      if (po.valid[k]) {
        const float r0 = po.r0[k], r1 = po.r1[k];
        // computeWeightsSse / computeWeight: w = (2+5)/(5 + r^T P r), mean 0 (dense_tracking_impl.cpp:640-707)
        float wgt = 1.0f;
        if (!unit_w) {
          const float t0 = r0 * P0 + r1 * P1;
          const float t1 = r0 * P2 + r1 * P3;
          const float dd = t0 * r0 + t1 * r1;
          wgt = 7.0f * __builtin_amdgcn_rcpf(5.0f + dd);
        }
        // computeScaleSse with Q5: a pair (2j, 2j+1) contributes (w_2j + w_2j+1) r_2j r_2j^T (:603-621).
        // S0 assumes this segment starts on an even global rank, S1 on an odd one.
        const float sxx = r0 * r0, sxy = r0 * r1, syy = r1 * r1;
        const float pxx = prev_has ? prev_r0 * prev_r0 : 0.0f, pxy = prev_has ? prev_r0 * prev_r1 : 0.0f,
                    pyy = prev_has ? prev_r1 * prev_r1 : 0.0f;
        const bool odd = (rank & 1) != 0;
        S0[0] = __builtin_fmaf(wgt, odd ? pxx : sxx, S0[0]);
        S0[1] = __builtin_fmaf(wgt, odd ? pxy : sxy, S0[1]);
        S0[2] = __builtin_fmaf(wgt, odd ? pyy : syy, S0[2]);
        S1[0] = __builtin_fmaf(wgt, odd ? sxx : pxx, S1[0]);
        S1[1] = __builtin_fmaf(wgt, odd ? sxy : pxy, S1[1]);
        S1[2] = __builtin_fmaf(wgt, odd ? syy : pyy, S1[2]);
        if (!prev_has) first_w = wgt;  // first valid pixel of the segment: its partner (if any) lives in an earlier segment

        // Jacobians at the untransformed reference point (dense_tracking.cpp:333-339,448-476)
        const float x = pin.x[k], y = pin.y[k], z = pin.z[k];
        const float iz = __builtin_amdgcn_rcpf(z);
        const float iz2 = iz * iz;
        const float j02 = -x * iz2, j12 = -y * iz2;
        const float j03 = j02 * y, j13 = -1.0f + j12 * y;
        const float j04 = 1.0f - j02 * x, j14 = -j03;
        const float j05 = -y * iz, j15 = x * iz;
        const float e2 = po.e2[k], e3 = po.e3[k], e4 = po.e4[k], e5 = po.e5[k];
        float Ja[6], Jb[6];
        Ja[0] = e2 * iz;
        Ja[1] = e3 * iz;
        Ja[2] = e2 * j02 + e3 * j12;
        Ja[3] = e2 * j03 + e3 * j13;
        Ja[4] = e2 * j04 + e3 * j14;
        Ja[5] = e2 * j05 + e3 * j15;
        Jb[0] = e4 * iz;
        Jb[1] = e5 * iz;
        Jb[2] = (e4 * j02 + e5 * j12) - 1.0f;
        Jb[3] = (e4 * j03 + e5 * j13) - y;
        Jb[4] = (e4 * j04 + e5 * j14) + x;
        Jb[5] = e4 * j05 + e5 * j15;
        // A += J^T (w P) J and b -= J^T (w P) r are linear in P: accumulate the P-free moments (87 sums)
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
        prev_r0 = r0, prev_r1 = r1, prev_has = true;
        ++rank;
      }
    }

    if (any_mask) {
      const int top = 63 - __clzll((long long)any_mask);
      carry_r0 = u2f(__builtin_amdgcn_readlane(f2u(lane_last_r0), top));
      carry_r1 = u2f(__builtin_amdgcn_readlane(f2u(lane_last_r1), top));
      carry_has = true;
    }
    run_count += round_total;

    idx += kSegPxPerRound;
    pcol += kSegPxPerRound;
    while (pcol >= w) pcol -= w, ++prow;
  }

  // ---- wave reduction, then the four waves of the block through LDS
  __shared__ float sm[kWavesPerBlock][kRecStride];
#pragma unroll
  for (int i = 0; i < kNumAcc; ++i) {
    const float s = wave_sum_to_lane63<RMODE>(acc[i]);
    if (lane == 63) sm[wave][kRecAcc + i] = s;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float s0 = wave_sum_to_lane63<RMODE>(S0[i]);
    const float s1 = wave_sum_to_lane63<RMODE>(S1[i]);
    if (lane == 63) sm[wave][kRecS0 + i] = s0, sm[wave][kRecS1 + i] = s1;
  }
  {
    const float fw = wave_sum_to_lane63<RMODE>(first_w);
    if (lane == 63) {
      sm[wave][kRecFirstW] = fw;
      sm[wave][kRecCount] = u2f((unsigned)run_count);
      sm[wave][kRecLastR] = carry_r0;
      sm[wave][kRecLastR + 1] = carry_r1;
    }
  }
  __syncthreads();

  float *rec = d.records + (size_t)lb * kRecStride;
  const int tid = threadIdx.x;
  if (tid < kNumAcc) {
    rec[kRecAcc + tid] = (sm[0][kRecAcc + tid] + sm[1][kRecAcc + tid]) + (sm[2][kRecAcc + tid] + sm[3][kRecAcc + tid]);
  } else if (tid == 128) {
    // ordered combine of the four wave segments (see combine rule in k_finalize)
    int c = 0;
    float fw = 0.0f, l0 = 0.0f, l1 = 0.0f;
    float s0[3] = {0.0f, 0.0f, 0.0f}, s1[3] = {0.0f, 0.0f, 0.0f};
    for (int wv = 0; wv < kWavesPerBlock; ++wv) {
      const int cb = (int)f2u(sm[wv][kRecCount]);
      rec[kRecWaveCnt + wv] = u2f((unsigned)cb);
      if (cb == 0) continue;
      const float bfw = sm[wv][kRecFirstW];
      const bool flip = (c & 1) != 0;  // b starts on the opposite parity of a
      float x0[3], x1[3];
      for (int i = 0; i < 3; ++i) {
        x0[i] = sm[wv][(flip ? kRecS1 : kRecS0) + i];  // contribution if the combined segment starts even
        x1[i] = sm[wv][(flip ? kRecS0 : kRecS1) + i];  // ... starts odd
      }
      if (c > 0) {
        const float rxx = l0 * l0, rxy = l0 * l1, ryy = l1 * l1;
        // b's first pixel is a pair-second when b starts on an odd rank: it weights a's last residual
        float *tgt = flip ? x0 : x1;
        tgt[0] += bfw * rxx, tgt[1] += bfw * rxy, tgt[2] += bfw * ryy;
      } else {
        fw = bfw;
      }
      for (int i = 0; i < 3; ++i) s0[i] += x0[i], s1[i] += x1[i];
      l0 = sm[wv][kRecLastR], l1 = sm[wv][kRecLastR + 1];
      c += cb;
    }
    rec[kRecCount] = u2f((unsigned)c);
    rec[kRecFirstW] = fw;
    rec[kRecLastR] = l0;
    rec[kRecLastR + 1] = l1;
    for (int i = 0; i < 3; ++i) rec[kRecS0 + i] = s0[i], rec[kRecS1 + i] = s1[i];
    rec[14] = 0.0f, rec[15] = 0.0f;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// log-likelihood pass: sum over the first 50*floor(V/50) valid residuals of log(1 + 0.2 r^T P r)
// (computeCompleteDataLogLikelihood, dense_tracking_impl.cpp:406-425, incl. Q6).  Same segment geometry as the residual
// pass that wrote the residuals; {cut_seg, cut_local} locate global rank 50*floor(V/50).
// ------------------------------------------------------------------------------------------------------------------
__device__ void loglik_pass(const TickItem &it, const LevelPairDesc &d, const int lb) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int seg = lb * kWavesPerBlock + wave;
  const int cut_seg = d.cut[it.ll_buf][0], cut_local = d.cut[it.ll_buf][1];
  const float P0 = it.P_ll[0], P1 = it.P_ll[1], P2 = it.P_ll[2], P3 = it.P_ll[3];
  const unsigned long long below = (1ull << lane) - 1ull;
  double total = 0.0;
  if (seg <= cut_seg) {
    const bool partial = seg == cut_seg;
    int idx = seg * (kSegPxPerRound * it.ll_rounds) + lane * kPxPerLane;
    int run_count = 0;
    for (int round = 0; round < it.ll_rounds; ++round) {
      const float4 *src = reinterpret_cast<const float4 *>(d.res[it.ll_buf] + idx);
      const float4 s0 = src[0], s1 = src[1];
      const float r0[kPxPerLane] = {s0.x, s0.z, s1.x, s1.z};
      const float r1[kPxPerLane] = {s0.y, s0.w, s1.y, s1.w};
      bool valid[kPxPerLane];
      int before = 0, round_total = 0;
#pragma unroll
      for (int k = 0; k < kPxPerLane; ++k) {
        valid[k] = r0[k] == r0[k];
        const unsigned long long b = __ballot(valid[k]);
        before += __popcll(b & below);
        round_total += __popcll(b);
      }
      int rank = run_count + before;
      double prod = 1.0;
#pragma unroll
      for (int k = 0; k < kPxPerLane; ++k) {
        if (valid[k]) {
          if (!partial || rank < cut_local) {
            const float t0 = r0[k] * P0 + r1[k] * P1;
            const float t1 = r0[k] * P2 + r1[k] * P3;
            const float q = t0 * r0[k] + t1 * r1[k];
            prod *= (1.0 + 0.2 * (double)q);
          }
          ++rank;
        }
      }
      if (prod != 1.0) total += log(prod);
      run_count += round_total;
      idx += kSegPxPerRound;
    }
  }
  total = wave_sum_double(total);
  __shared__ double smd[kWavesPerBlock];
  if (lane == 0) smd[wave] = total;
  __syncthreads();
  if (threadIdx.x == 0) d.ll_partials[lb] = (smd[0] + smd[1]) + (smd[2] + smd[3]);
}

template <int RMODE>
__global__ __launch_bounds__(kBlockThreads) void k_tick(const TickArgs args) {
  const TickItem &it = args.items[blockIdx.y];
  const int bx = (int)blockIdx.x;
  if (bx >= it.res_blocks + it.ll_blocks) return;
  const LevelPairDesc &d = *it.desc;
  if (bx < it.res_blocks)
    residual_pass<RMODE>(it, d, xcd_contiguous_block(bx, it.res_blocks));
  else
    loglik_pass(it, d, bx - it.res_blocks);
}

static int g_reduce_mode = -1;

hipError_t launch_tick(const TickArgs &args, int max_blocks, hipStream_t stream) {
  if (g_reduce_mode < 0) {
    const char *e = getenv("DVO_AMD_REDUCE");
    g_reduce_mode = (e && e[0] == '0') ? 0 : 1;
  }
  if (args.n_items <= 0 || max_blocks <= 0) return hipSuccess;
  dim3 grid((unsigned)((max_blocks + 7) & ~7), (unsigned)args.n_items, 1);
  if (g_reduce_mode == 0)
    hipLaunchKernelGGL(k_tick<0>, grid, dim3(kBlockThreads), 0, stream, args);
  else
    hipLaunchKernelGGL(k_tick<1>, grid, dim3(kBlockThreads), 0, stream, args);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// finalize: one 1024-thread block per job.  Sums the 87 moments over the blocks in fp64, combines the ordered part of the
// records (count, S under both start parities, boundary residual/weight) left to right, locates the log-likelihood
// cut, sums the log-likelihood partials.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kFinThreads = 1024;
constexpr int kFinChunks = 10;
constexpr int kFinCols = 96;

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

__global__ __launch_bounds__(kFinThreads) void k_finalize(const FinArgs args) {
  const FinItem &it = args.items[blockIdx.x];
  const int t = threadIdx.x;
  FinOut *out = it.out;
  __shared__ double sh_acc[kFinChunks][kFinCols];
  __shared__ SegRec sh_seg[256];
  __shared__ int sh_scan[kFinThreads];
  __shared__ double sh_ll[kFinThreads];

  const int nb = it.records ? it.n_blocks : 0;
  if (nb > 0) {
    // (a) the 87 plain sums
    {
      const int col = t % kFinCols, chunk = t / kFinCols;
      if (chunk < kFinChunks) {
        double s = 0.0;
        if (col < kNumAcc)
          for (int b = chunk; b < nb; b += kFinChunks) s += (double)it.records[(size_t)b * kRecStride + kRecAcc + col];
        sh_acc[chunk][col] = s;
      }
    }
    // (b) ordered part: 256 threads each fold a contiguous run of blocks, then a log tree
    if (t < 256) {
      const int per = (nb + 255) / 256;
      SegRec r;
      r.c = 0, r.first_w = 0.0f, r.l0 = r.l1 = 0.0f;
      for (int i = 0; i < 3; ++i) r.s0[i] = r.s1[i] = 0.0;
      for (int b = t * per; b < (t + 1) * per && b < nb; ++b) {
        const float *rec = it.records + (size_t)b * kRecStride;
        SegRec q;
        q.c = (int)f2u(rec[kRecCount]);
        q.first_w = rec[kRecFirstW];
        q.l0 = rec[kRecLastR], q.l1 = rec[kRecLastR + 1];
        for (int i = 0; i < 3; ++i) q.s0[i] = rec[kRecS0 + i], q.s1[i] = rec[kRecS1 + i];
        r = seg_combine(r, q);
      }
      sh_seg[t] = r;
    }
    __syncthreads();
    for (int stride = 1; stride < 256; stride <<= 1) {
      if (t < 256 && (t % (2 * stride)) == 0) sh_seg[t] = seg_combine(sh_seg[t], sh_seg[t + stride]);
      __syncthreads();
    }
    if (t < kNumAcc) {
      double s = 0.0;
      for (int c = 0; c < kFinChunks; ++c) s += sh_acc[c][t];
      out->acc[t] = s;
    }
    const int V = sh_seg[0].c;
    if (t == 0) {
      out->valid = V;
      out->has_res = 1;
      for (int i = 0; i < 3; ++i) out->S[i] = sh_seg[0].s0[i];
    }
    // (c) segment that holds global rank 50*floor(V/50)
    {
      const int nseg = nb * kWavesPerBlock;
      const int per = (nseg + kFinThreads - 1) / kFinThreads;
      const int cutoff = 50 * (V / 50);
      int local = 0;
      for (int s = t * per; s < (t + 1) * per && s < nseg; ++s)
        local += (int)f2u(it.records[(size_t)(s >> 2) * kRecStride + kRecWaveCnt + (s & 3)]);
      sh_scan[t] = local;
      __syncthreads();
      for (int off = 1; off < kFinThreads; off <<= 1) {
        const int v = t >= off ? sh_scan[t - off] : 0;
        __syncthreads();
        sh_scan[t] += v;
        __syncthreads();
      }
      int prefix = sh_scan[t] - local;  // exclusive
      if (t == 0 && cutoff >= V) it.cut_out[0] = 0x7fffffff, it.cut_out[1] = 0;
      if (cutoff < V) {
        for (int s = t * per; s < (t + 1) * per && s < nseg; ++s) {
          const int c = (int)f2u(it.records[(size_t)(s >> 2) * kRecStride + kRecWaveCnt + (s & 3)]);
          if (prefix <= cutoff && cutoff < prefix + c) it.cut_out[0] = s, it.cut_out[1] = cutoff - prefix;
          prefix += c;
        }
      }
    }
  } else if (t == 0) {
    out->has_res = 0;
    out->valid = 0;
  }

  // (d) log-likelihood partials
  {
    double s = 0.0;
    for (int b = t; b < it.n_ll_blocks; b += kFinThreads) s += it.ll_partials[b];
    sh_ll[t] = s;
    __syncthreads();
    for (int stride = kFinThreads / 2; stride > 0; stride >>= 1) {
      if (t < stride) sh_ll[t] += sh_ll[t + stride];
      __syncthreads();
    }
    if (t == 0) {
      out->ll_sum = sh_ll[0];
      out->has_ll = it.n_ll_blocks > 0 ? 1 : 0;
    }
  }
  // publish: every thread's stores to the (host) record are ordered before the sequence word
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(&out->seq, it.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_finalize(const FinArgs &args, hipStream_t stream) {
  if (args.n_items <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_finalize, dim3((unsigned)args.n_items), dim3(kFinThreads), 0, stream, args);
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
  int span = n_pad > ty_len ? n_pad : ty_len;
  if (w > span) span = w;
  hipLaunchKernelGGL(k_level_planes, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, stream, i_plane, z_plane, w, h,
                     n_pad, fx, fy, ox, oy, c_a, c_b, r_i, r_ix, r_iy, tx, ty, ty_len);
  return hipGetLastError();
}

// ValidPointAndGradientThresholdPredicate::isPointOk (point_selection.h:63-66) applied in place: zsel = z where the pixel
// is selected, NaN elsewhere (and in the padding)
__global__ void k_select(const float *__restrict__ zp, const float4 *__restrict__ c_a, const float2 *__restrict__ c_b, int n,
                         int n_pad, float ti, float td, float *__restrict__ zsel, int *__restrict__ counters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  float out = u2f(0x7fc00000u);
  if (i < n) {
    const float z = zp[i];
    const float4 a = c_a[i];
    const float2 b = c_b[i];
    const bool ok = (z == z) && (b.x == b.x) && (b.y == b.y) &&
                    (__builtin_fabsf(a.z) > ti || __builtin_fabsf(a.w) > ti || __builtin_fabsf(b.x) > td ||
                     __builtin_fabsf(b.y) > td);
    if (ok) {
      out = z;
      atomicAdd(&counters[0], 1);
      atomicMax(&counters[1], i);
    }
  }
  zsel[i] = out;
}

// Q3: computeResidualsSse walks the selection two points at a time and never looks at an odd trailing point
// (dense_tracking_impl.cpp:169-171)
__global__ void k_select_drop_odd(float *zsel, const int *counters) {
  if ((counters[0] & 1) && counters[1] >= 0) zsel[counters[1]] = u2f(0x7fc00000u);
}

hipError_t launch_select(const float *z_plane, const float4 *c_a, const float2 *c_b, int n, int n_pad, float ti, float td,
                         float *zsel, int *counters, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(counters, 0, sizeof(int), stream);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(counters + 1, 0xFF, sizeof(int), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_select, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, stream, z_plane, c_a, c_b, n, n_pad, ti,
                     td, zsel, counters);
  hipLaunchKernelGGL(k_select_drop_odd, dim3(1), dim3(1), 0, stream, zsel, (const int *)counters);
  return hipGetLastError();
}

__global__ void k_copy_strided(const float *__restrict__ src, int stride, float *__restrict__ dst, int w, int h) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x < w && y < h) dst[(size_t)y * w + x] = src[(size_t)y * stride + x];
}

hipError_t launch_copy_strided(const float *src, int stride, float *dst, int w, int h, hipStream_t stream) {
  hipLaunchKernelGGL(k_copy_strided, dim3((unsigned)((w + 255) / 256), (unsigned)h), dim3(256), 0, stream, src, stride, dst, w,
                     h);
  return hipGetLastError();
}

__global__ void k_mask_from_zsel(const float *__restrict__ zsel, int n, int last_dropped, unsigned char *__restrict__ mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) mask[i] = (zsel[i] == zsel[i] || i == last_dropped) ? 1 : 0;
}

hipError_t launch_mask_from_zsel(const float *zsel, int n, int last_dropped, unsigned char *mask, hipStream_t stream) {
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
  hipLaunchKernelGGL(k_unpack_plane, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, c_a, c_b, plane, n, dst);
  return hipGetLastError();
}

}  // namespace dvo_amd
