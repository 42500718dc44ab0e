// Dual-match front-end step on top of the tracker's C ABI: the per-frame pair of alignments of
// dvo_slam::LocalTracker::update (local_tracker.cpp:157-213) as one two-pair GPU batch, and the inputs of the accept
// criteria KeyframeTracker evaluates on the two results (keyframe_tracker.cpp:105-190).  Host code only.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/dvo_amd.h"

namespace {

// eigenvalues of a symmetric 6x6 matrix by cyclic Jacobi rotations (Eigen::SelfAdjointEigenSolver in the reference,
// keyframe_tracker.cpp:172-181; only the extreme eigenvalues are used)
void sym_eigenvalues6(const double *A_colmajor, double ev[6]) {
  double a[6][6];
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) a[r][c] = 0.5 * (A_colmajor[c * 6 + r] + A_colmajor[r * 6 + c]);
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int r = 0; r < 6; ++r) {
      diag += a[r][r] * a[r][r];
      for (int c = r + 1; c < 6; ++c) off += a[r][c] * a[r][c];
    }
    if (!(off > 1e-30 * diag)) break;
    for (int p = 0; p < 5; ++p)
      for (int q = p + 1; q < 6; ++q) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 6; ++k) {  // columns p, q
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq, a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 6; ++k) {  // rows p, q
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk, a[q][k] = s * apk + c * aqk;
        }
      }
  }
  for (int i = 0; i < 6; ++i) ev[i] = a[i][i];
  std::sort(ev, ev + 6);
}

double condition_number(const double *information) {
  double ev[6];
  sym_eigenvalues6(information, ev);
  return std::fabs(ev[5] / ev[0]);
}

double translation_norm(const double *T) { return std::sqrt(T[12] * T[12] + T[13] * T[13] + T[14] * T[14]); }

}  // namespace

extern "C" int dvo_amd_track_frame(dvo_amd_context *ctx, dvo_amd_pyramid *keyframe, dvo_amd_pyramid *last_frame,
                                   dvo_amd_pyramid *frame, const double *last_keyframe_pose, dvo_amd_result *r_keyframe,
                                   dvo_amd_result *r_odometry, dvo_amd_frame_criteria *criteria) {
  if (!ctx || !keyframe || !last_frame || !frame || !r_keyframe || !r_odometry) return DVO_AMD_ERR_INVALID_ARGUMENT;
  dvo_amd_config cfg;
  int rc = dvo_amd_get_config(ctx, &cfg);
  if (rc) return rc;
  // r_keyframe.Transformation = last_keyframe_pose_.inverse(Eigen::Isometry): R^T, -R^T t (local_tracker.cpp:173);
  // r_odometry.Transformation.setIdentity() (:172)
  double inits[32];
  for (int i = 0; i < 32; ++i) inits[i] = (i % 16) % 5 == 0 ? 1.0 : 0.0;
  if (last_keyframe_pose) {
    const double *P = last_keyframe_pose;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) inits[c * 4 + r] = P[r * 4 + c];
      inits[12 + r] = -(P[r * 4 + 0] * P[12] + P[r * 4 + 1] * P[13] + P[r * 4 + 2] * P[14]);
    }
  }
  // the constraint-ratio criterion reads the last iteration of the last level: make sure iteration statistics exist
  const int its_needed = (cfg.first_level - cfg.last_level + 1) * (cfg.max_iterations_per_level + 1);
  std::vector<dvo_amd_iteration_stats> scratch;
  dvo_amd_result results[2] = {*r_keyframe, *r_odometry};
  const bool own_kf_its = !(results[0].iterations && results[0].iterations_capacity >= its_needed);
  if (own_kf_its) {
    scratch.resize((size_t)std::max(its_needed, 1));
    results[0].iterations = scratch.data(), results[0].iterations_capacity = its_needed;
  }
  dvo_amd_pyramid *refs[2] = {keyframe, last_frame}, *curs[2] = {frame, frame};
  rc = dvo_amd_match_batch(ctx, 2, refs, curs, inits, results);
  if (rc) return rc;
  double ratio = 0.0;
  if (results[0].n_levels > 0) {
    const dvo_amd_level_stats &L = results[0].levels[results[0].n_levels - 1];
    if (L.n_iterations > 0)
      ratio = (double)results[0].iterations[L.first_iteration + L.n_iterations - 1].valid_constraints / (double)L.valid_pixels;
  }
  if (own_kf_its) {
    results[0].iterations = r_keyframe->iterations;
    results[0].iterations_capacity = r_keyframe->iterations ? r_keyframe->iterations_capacity : 0;
    results[0].n_iterations = 0;
  }
  *r_keyframe = results[0], *r_odometry = results[1];
  if (criteria) {
    std::memset(criteria, 0, sizeof(*criteria));
    criteria->keyframe_is_nan = r_keyframe->is_nan, criteria->odometry_is_nan = r_odometry->is_nan;
    criteria->keyframe_translation_norm = translation_norm(r_keyframe->transformation);
    criteria->odometry_translation_norm = translation_norm(r_odometry->transformation);
    criteria->keyframe_constraint_ratio = ratio;
    criteria->keyframe_neg_loglik = -r_keyframe->loglik, criteria->odometry_neg_loglik = -r_odometry->loglik;
    criteria->keyframe_condition_number = condition_number(r_keyframe->information);
    criteria->odometry_condition_number = condition_number(r_odometry->information);
  }
  return DVO_AMD_OK;
}
