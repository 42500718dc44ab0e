// Batched two-stage loop-closure validation on top of the C ABI of the tracker.
//
// Follows dvo_slam::constraints::ConstraintProposalValidator::validate (constraint_proposal_validator.cpp:69-166), the
// voters of constraint_proposal_voter.cpp:34-186 and ConstraintProposal (constraint_proposal.cpp:31-110).  The reference
// hands one proposal at a time to a per-thread validator (keyframe_graph.cpp:525-593); here a stage aligns ALL of its
// proposals, including the cross-validation inverses, in one dvo_amd_match_many() call, so the GPU sees full launches.
// Only the public C ABI of the tracker is used: no kernels, no device memory in this file.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/dvo_amd.h"

namespace {

struct Mat4 {
  double m[16];  // column-major
  double &at(int r, int c) { return m[c * 4 + r]; }
  double at(int r, int c) const { return m[c * 4 + r]; }
};

Mat4 mat4_load(const double *p) {
  Mat4 a;
  std::memcpy(a.m, p, sizeof(a.m));
  return a;
}

Mat4 mat4_identity() {
  Mat4 a;
  for (int i = 0; i < 16; ++i) a.m[i] = (i % 5 == 0) ? 1.0 : 0.0;
  return a;
}

Mat4 mat4_mul(const Mat4 &a, const Mat4 &b) {
  Mat4 c;
  for (int col = 0; col < 4; ++col)
    for (int row = 0; row < 4; ++row) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += a.at(row, k) * b.at(k, col);
      c.at(row, col) = s;
    }
  return c;
}

// Eigen::Affine3d::inverse(): general inverse of the linear part, then -L^-1 t (the transforms here are rigid, so this is
// R^T up to rounding; the general form is what the reference evaluates)
Mat4 affine_inverse(const Mat4 &a) {
  const double a00 = a.at(0, 0), a01 = a.at(0, 1), a02 = a.at(0, 2);
  const double a10 = a.at(1, 0), a11 = a.at(1, 1), a12 = a.at(1, 2);
  const double a20 = a.at(2, 0), a21 = a.at(2, 1), a22 = a.at(2, 2);
  const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
  const double det = a00 * c00 + a01 * c01 + a02 * c02;
  const double inv = 1.0 / det;
  Mat4 r = mat4_identity();
  r.at(0, 0) = c00 * inv, r.at(0, 1) = (a02 * a21 - a01 * a22) * inv, r.at(0, 2) = (a01 * a12 - a02 * a11) * inv;
  r.at(1, 0) = c01 * inv, r.at(1, 1) = (a00 * a22 - a02 * a20) * inv, r.at(1, 2) = (a02 * a10 - a00 * a12) * inv;
  r.at(2, 0) = c02 * inv, r.at(2, 1) = (a01 * a20 - a00 * a21) * inv, r.at(2, 2) = (a00 * a11 - a01 * a10) * inv;
  for (int row = 0; row < 3; ++row)
    r.at(row, 3) = -(r.at(row, 0) * a.at(0, 3) + r.at(row, 1) * a.at(1, 3) + r.at(row, 2) * a.at(2, 3));
  return r;
}

// determinant of the column-major 6x6 Information matrix by LU with partial pivoting
double det6(const double *A) {
  double m[6][6];
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) m[r][c] = A[c * 6 + r];
  double det = 1.0;
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    for (int r = k + 1; r < 6; ++r)
      if (std::fabs(m[r][k]) > std::fabs(m[piv][k])) piv = r;
    if (m[piv][k] == 0.0) return 0.0;
    if (piv != k) {
      for (int c = 0; c < 6; ++c) std::swap(m[piv][c], m[k][c]);
      det = -det;
    }
    det *= m[k][k];
    for (int r = k + 1; r < 6; ++r) {
      const double f = m[r][k] / m[k][k];
      for (int c = k; c < 6; ++c) m[r][c] -= f * m[k][c];
    }
  }
  return det;
}

struct Prop {
  dvo_amd_constraint_proposal p;
  int uid;
};

bool rejected(const Prop &q) {  // ConstraintProposal::Reject, constraint_proposal.cpp:75-81
  for (int i = 0; i < q.p.n_votes; ++i)
    if (q.p.votes[i].reject) return true;
  return false;
}

double total_score(const Prop &q) {  // ConstraintProposal::TotalScore, :55-65
  double s = 0.0;
  for (int i = 0; i < q.p.n_votes; ++i) s += q.p.votes[i].score;
  return s;
}

bool same_frames(const Prop &a, const Prop &b) {  // isConstraintBetweenSameFrames, :98-101 (either direction)
  return (a.p.reference == b.p.reference && a.p.current == b.p.current) ||
         (a.p.reference == b.p.current && a.p.current == b.p.reference);
}

int find_uid(const std::vector<Prop> &v, int uid) {
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i].uid == uid) return (int)i;
  return -1;
}

// the last level's statistics a voter may need, captured before the iteration buffers are released
struct LastLevel {
  int has_iteration_with_increment;
  int valid_constraints_with_increment;  // LastIterationWithIncrement().ValidConstraints
  int valid_constraints_last;            // Iterations.back().ValidConstraints
  int valid_pixels;
};

LastLevel last_level_of(const dvo_amd_result &r) {
  LastLevel out{0, 0, 0, 0};
  if (r.n_levels <= 0) return out;
  const dvo_amd_level_stats &L = r.levels[r.n_levels - 1];
  out.valid_pixels = L.valid_pixels;
  // LevelStats::HasIterationWithIncrement / LastIterationWithIncrement, dense_tracking_config.cpp:138-171
  const int min_its = (L.termination == DVO_AMD_TERM_LOGLIKELIHOOD_DECREASED || L.termination == DVO_AMD_TERM_TOO_FEW_CONSTRAINTS)
                          ? 2 : 1;
  out.has_iteration_with_increment = L.n_iterations >= min_its;
  if (r.iterations && L.n_iterations > 0) {
    const dvo_amd_iteration_stats *its = r.iterations + L.first_iteration;
    out.valid_constraints_last = its[L.n_iterations - 1].valid_constraints;
    if (out.has_iteration_with_increment) {
      const int k = L.termination == DVO_AMD_TERM_LOGLIKELIHOOD_DECREASED ? L.n_iterations - 2 : L.n_iterations - 1;
      out.valid_constraints_with_increment = its[k].valid_constraints;
    }
  }
  return out;
}

double evaluation_value(int kind, const dvo_amd_result &r, const LastLevel &ll) {  // tracking_result_evaluation.cpp:54-67
  switch (kind) {
    case DVO_AMD_EVAL_NORMALIZED_LOGLIKELIHOOD: return -r.loglik / (double)ll.valid_constraints_last;
    case DVO_AMD_EVAL_ENTROPY: return std::log(det6(r.information));
    default: return -r.loglik;
  }
}

}  // namespace

extern "C" {

void dvo_amd_default_validator_stages(const dvo_amd_config *frontend_cfg, double min_constraint_ratio, double ratio_coarse,
                                      double ratio_fine, dvo_amd_validator_stage stages[2]) {
  if (!stages) return;
  dvo_amd_config base;
  dvo_amd_default_config(&base);
  std::memset(stages, 0, sizeof(dvo_amd_validator_stage) * 2);
  for (int s = 0; s < 2; ++s) {
    // configureValidationTracking, keyframe_graph.cpp:819-838: defaults + {Precision, Mu, thresholds} of the front end
    dvo_amd_config c = base;
    c.first_level = 3;
    c.last_level = s == 0 ? 3 : 1;
    c.use_initial_estimate = 1;
    if (frontend_cfg) {
      c.precision = frontend_cfg->precision, c.mu = frontend_cfg->mu;
      c.intensity_derivative_threshold = frontend_cfg->intensity_derivative_threshold;
      c.depth_derivative_threshold = frontend_cfg->depth_derivative_threshold;
      c.segment_geometry = frontend_cfg->segment_geometry;  // (not a reference field: the tracker's wave-segment geometry)
    }
    stages[s].tracking_config = c;
  }
  // createConstraintProposalValidator, keyframe_graph.cpp:500-523
  dvo_amd_validator_stage &a = stages[0];
  a.id = 1, a.only_keep_best = 0, a.n_voters = 5;
  a.voters[0] = {DVO_AMD_VOTER_ODOMETRY_CONSTRAINT, 0.0};
  a.voters[1] = {DVO_AMD_VOTER_NAN_RESULT, 0.0};
  a.voters[2] = {DVO_AMD_VOTER_CONSTRAINT_RATIO, min_constraint_ratio};
  a.voters[3] = {DVO_AMD_VOTER_TRACKING_RESULT_EVALUATION, ratio_coarse};
  a.voters[4] = {DVO_AMD_VOTER_CROSS_VALIDATION, 1.0};
  dvo_amd_validator_stage &b = stages[1];
  b.id = 2, b.only_keep_best = 1, b.n_voters = 3;
  b.voters[0] = {DVO_AMD_VOTER_NAN_RESULT, 0.0};
  b.voters[1] = {DVO_AMD_VOTER_CONSTRAINT_RATIO, min_constraint_ratio};
  b.voters[2] = {DVO_AMD_VOTER_TRACKING_RESULT_EVALUATION, ratio_fine};
}

int dvo_amd_proposals_for_candidates(const dvo_amd_keyframe *keyframes, int keyframe, int n_candidates, const int *candidates,
                                     dvo_amd_constraint_proposal *proposals) {
  if (!keyframes || keyframe < 0 || n_candidates < 0 || (n_candidates > 0 && (!candidates || !proposals)))
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  const Mat4 ref_pose = mat4_load(keyframes[keyframe].pose);
  for (int i = 0; i < n_candidates; ++i) {
    if (candidates[i] < 0) return DVO_AMD_ERR_INVALID_ARGUMENT;
    for (int k = 0; k < 2; ++k) {
      dvo_amd_constraint_proposal &p = proposals[2 * i + k];
      std::memset(&p, 0, sizeof(p));
      p.reference = keyframe, p.current = candidates[i];
      // createWithIdentity / createWithRelative, constraint_proposal.cpp:31-49
      const Mat4 init = k == 0 ? mat4_identity() : mat4_mul(affine_inverse(mat4_load(keyframes[candidates[i]].pose)), ref_pose);
      std::memcpy(p.initial_transformation, init.m, sizeof(init.m));
    }
  }
  return DVO_AMD_OK;
}

extern "C++" {  // (helpers with C++ types inside the extern "C" block of the entry points)
namespace {
// worker contexts of the validator, per device, kept for the life of the process
struct WorkerPool {
  std::mutex mu;
  std::vector<dvo_amd_context *> idle[16];
};
WorkerPool &worker_pool() {
  static WorkerPool *p = new WorkerPool;  // never destroyed: contexts may outlive static destruction order
  return *p;
}
int worker_acquire(int device, const dvo_amd_config *cfg, dvo_amd_context **out) {
  *out = nullptr;
  if (device < 0 || device >= 16) return DVO_AMD_ERR_INVALID_ARGUMENT;
  {
    std::lock_guard<std::mutex> lock(worker_pool().mu);
    std::vector<dvo_amd_context *> &idle = worker_pool().idle[device];
    if (!idle.empty()) {
      *out = idle.back();
      idle.pop_back();
    }
  }
  if (*out) return dvo_amd_configure(*out, cfg);
  return dvo_amd_context_create(device, cfg, out);
}
// A worker goes back to the pool only when its share succeeded (a context whose batch failed is destroyed: nothing of it
// may be reused), and the pool keeps at most kMaxIdleWorkers contexts per device (each holds the scratch of its resident pairs).
constexpr size_t kMaxIdleWorkers = 8;
void worker_release(int device, dvo_amd_context *c, bool healthy) {
  if (healthy) {
    std::lock_guard<std::mutex> lock(worker_pool().mu);
    std::vector<dvo_amd_context *> &idle = worker_pool().idle[device];
    if (idle.size() < kMaxIdleWorkers) {
      idle.push_back(c);
      return;
    }
  }
  dvo_amd_context_destroy(c);
}
// read at every call (a getenv is nothing next to a validation): a test or a host application may change it between calls.
// The number of workers changes which context aligns a proposal and what shares its ticks -- never its result
// (tests/test_validator.py::test_validator_output_does_not_depend_on_the_worker_count)
int validator_threads() {
  const char *e = getenv("DVO_AMD_VALIDATOR_THREADS");
  int n = e ? atoi(e) : 3;
  return n < 1 ? 1 : (n > 8 ? 8 : n);
}
}  // namespace
}  // extern "C++"

int dvo_amd_validate_proposals(dvo_amd_context *ctx, int n_keyframes, const dvo_amd_keyframe *keyframes, int n_stages,
                               const dvo_amd_validator_stage *stages, int n_proposals,
                               dvo_amd_constraint_proposal *proposals, int *n_out, int max_in_flight) {
  if (n_out) *n_out = 0;
  if (!ctx || !n_out || n_keyframes < 0 || n_stages < 0 || n_proposals < 0 || (n_keyframes > 0 && !keyframes) ||
      (n_stages > 0 && !stages) || (n_proposals > 0 && !proposals))
    return DVO_AMD_ERR_INVALID_ARGUMENT;
  for (int s = 0; s < n_stages; ++s) {
    if (stages[s].n_voters < 0 || stages[s].n_voters > DVO_AMD_MAX_VOTERS) return DVO_AMD_ERR_INVALID_ARGUMENT;
    for (int v = 0; v < stages[s].n_voters; ++v)
      if (stages[s].voters[v].kind < DVO_AMD_VOTER_ODOMETRY_CONSTRAINT || stages[s].voters[v].kind > DVO_AMD_VOTER_CROSS_VALIDATION)
        return DVO_AMD_ERR_INVALID_ARGUMENT;
  }
  for (int i = 0; i < n_proposals; ++i) {
    const dvo_amd_constraint_proposal &p = proposals[i];
    if (p.reference < 0 || p.reference >= n_keyframes || p.current < 0 || p.current >= n_keyframes) return DVO_AMD_ERR_INVALID_ARGUMENT;
    if (!keyframes[p.reference].image || !keyframes[p.current].image) return DVO_AMD_ERR_INVALID_ARGUMENT;
  }
  dvo_amd_config saved;
  int rc = dvo_amd_get_config(ctx, &saved);
  if (rc) return rc;

  int next_uid = 0;
  std::vector<Prop> live((size_t)n_proposals);
  for (int i = 0; i < n_proposals; ++i) {
    live[(size_t)i].p = proposals[i];
    live[(size_t)i].p.tracking_result.iterations = nullptr;
    live[(size_t)i].p.tracking_result.iterations_capacity = 0;
    live[(size_t)i].p.origin = i, live[(size_t)i].p.reserved = 0;
    live[(size_t)i].uid = next_uid++;
  }

  for (int s = 0; s < n_stages && rc == DVO_AMD_OK; ++s) {
    const dvo_amd_validator_stage &stage = stages[s];
    // reset votes and tracking statistics (validator.cpp:73-80)
    for (Prop &q : live) {
      q.p.n_votes = 0;
      q.p.tracking_result.n_levels = 0, q.p.tracking_result.n_iterations = 0;
    }
    // additional proposals: every cross-validation voter appends the inverse of each proposal present at that moment and
    // remembers the pairs (voter.cpp:38-51)
    std::vector<std::vector<std::pair<int, int>>> pairs((size_t)stage.n_voters);
    for (int v = 0; v < stage.n_voters; ++v) {
      if (stage.voters[v].kind != DVO_AMD_VOTER_CROSS_VALIDATION) continue;
      const size_t old_size = live.size();
      for (size_t idx = 0; idx < old_size; ++idx) {
        Prop inv;
        std::memset(&inv.p, 0, sizeof(inv.p));
        inv.p.reference = live[idx].p.current, inv.p.current = live[idx].p.reference;  // createInverseProposal, :88-96
        const Mat4 init = affine_inverse(mat4_load(live[idx].p.initial_transformation));
        std::memcpy(inv.p.initial_transformation, init.m, sizeof(init.m));
        inv.p.origin = -(live[idx].p.origin + 1);  // the inverse of input i is -(i + 1), the inverse of that i again
        inv.uid = next_uid++;
        pairs[(size_t)v].emplace_back(live[idx].uid, inv.uid);
        live.push_back(inv);
      }
    }
    // one batch for the whole stage (validator.cpp:137-143 aligns them one by one)
    const size_t n = live.size();
    std::vector<LastLevel> last((size_t)n);
    if (n > 0) {
      rc = dvo_amd_configure(ctx, &stage.tracking_config);
      if (rc) break;
      const dvo_amd_config &c = stage.tracking_config;
      const int its_per_pair = (c.first_level - c.last_level + 1) * (c.max_iterations_per_level + 1);
      // per-iteration statistics of the stage's alignments: 5-8 MB for 64 proposals.  Kept between calls (per calling thread)
      // instead of allocated and zeroed per stage -- first-touch page faults on 13 MB were a third of a validate() call;
      // only the entries a result counts (n_iterations) are ever read
      static thread_local std::vector<dvo_amd_iteration_stats> its;
      if (its.size() < n * (size_t)its_per_pair) its.resize(n * (size_t)its_per_pair);
      std::vector<dvo_amd_pyramid *> refs(n), curs(n);
      std::vector<double> inits(n * 16);
      std::vector<dvo_amd_result> results(n);
      for (size_t i = 0; i < n; ++i) {
        refs[i] = keyframes[live[i].p.reference].image, curs[i] = keyframes[live[i].p.current].image;
        std::memcpy(&inits[16 * i], live[i].p.initial_transformation, sizeof(double) * 16);
        std::memcpy(live[i].p.stage_initial_transformation, live[i].p.initial_transformation, sizeof(double) * 16);  // (instrumentation)
        std::memset(&results[i], 0, sizeof(dvo_amd_result));
        results[i].iterations = &its[i * (size_t)its_per_pair];
        results[i].iterations_capacity = its_per_pair;
      }
      // The reference deals the proposals over TBB workers (keyframe_graph.cpp:576-593); one host thread advances ~36 pairs
      // per 20 us, which for a stage of many short alignments is slower than the GPU: a large stage is split over a few
      // worker contexts (own stream and scratch each, kept between calls), the caller's thread taking the first share.
      const int n_workers = (int)std::min<size_t>((size_t)validator_threads(), n / 24);
      if (n_workers <= 1) {
        rc = dvo_amd_match_many(ctx, (int)n, refs.data(), curs.data(), inits.data(), results.data(), max_in_flight);
      } else {
        int device = 0;
        rc = dvo_amd_context_device(ctx, &device);
        if (rc) break;
        std::vector<dvo_amd_context *> wctx((size_t)n_workers, nullptr);
        std::vector<int> wrc((size_t)n_workers, DVO_AMD_OK);
        wctx[0] = ctx;
        for (int w = 1; w < n_workers && rc == DVO_AMD_OK; ++w) rc = worker_acquire(device, &stage.tracking_config, &wctx[(size_t)w]);
        if (rc == DVO_AMD_OK) {
          auto share = [&](int w) {
            const size_t lo = n * (size_t)w / (size_t)n_workers, hi = n * (size_t)(w + 1) / (size_t)n_workers;
            wrc[(size_t)w] = dvo_amd_match_many(wctx[(size_t)w], (int)(hi - lo), refs.data() + lo, curs.data() + lo, inits.data() + 16 * lo,
                                                results.data() + lo, max_in_flight);
          };
          std::vector<std::thread> threads;
          for (int w = 1; w < n_workers; ++w) threads.emplace_back(share, w);
          share(0);
          for (std::thread &t : threads) t.join();
          for (int w = 0; w < n_workers; ++w)
            if (wrc[(size_t)w] != DVO_AMD_OK && rc == DVO_AMD_OK) rc = wrc[(size_t)w];
        }
        for (int w = 1; w < n_workers; ++w)
          if (wctx[(size_t)w]) worker_release(device, wctx[(size_t)w], wrc[(size_t)w] == DVO_AMD_OK);
      }
      if (rc) break;
      for (size_t i = 0; i < n; ++i) {
        last[i] = last_level_of(results[i]);
        live[i].p.tracking_result = results[i];
        live[i].p.tracking_result.iterations = nullptr;
        live[i].p.tracking_result.iterations_capacity = 0;
      }
    }
    // votes, in voter order, stopping at a proposal's first rejection (validator.cpp:146-157)
    for (size_t i = 0; i < n; ++i) {
      Prop &q = live[i];
      const dvo_amd_result &r = q.p.tracking_result;
      for (int v = 0; v < stage.n_voters; ++v) {
        dvo_amd_vote vote;
        vote.voter_kind = stage.voters[v].kind, vote.reject = 1, vote.score = 0.0, vote.value = 0.0;
        const double thr = stage.voters[v].threshold;
        switch (stage.voters[v].kind) {
          case DVO_AMD_VOTER_ODOMETRY_CONSTRAINT: {
            const int d = keyframes[q.p.reference].id - keyframes[q.p.current].id;
            const bool odometry = (d < 0 ? -d : d) <= 1;
            vote.value = odometry ? 1.0 : 0.0, vote.reject = odometry ? 1 : 0;
          } break;
          case DVO_AMD_VOTER_NAN_RESULT:
            vote.value = r.is_nan ? 1.0 : 0.0, vote.reject = r.is_nan ? 1 : 0;
            break;
          case DVO_AMD_VOTER_CONSTRAINT_RATIO: {
            const double ratio = last[i].has_iteration_with_increment
                                     ? (double)last[i].valid_constraints_with_increment / (double)last[i].valid_pixels : 0.0;
            vote.value = ratio, vote.reject = ratio >= thr ? 0 : 1;
          } break;
          case DVO_AMD_VOTER_TRACKING_RESULT_EVALUATION: {
            const dvo_amd_keyframe &kf = keyframes[q.p.reference];
            // ratioWithAverage, tracking_result_evaluation.cpp:39-42
            const double ratio = evaluation_value(kf.evaluation_kind, r, last[i]) / kf.evaluation_average * kf.evaluation_n;
            vote.value = ratio, vote.score = ratio, vote.reject = ratio >= thr ? 0 : 1;
          } break;
          case DVO_AMD_VOTER_CROSS_VALIDATION: {
            int partner = -1;  // findInverse, voter.cpp:90-99
            for (const auto &pr : pairs[(size_t)v]) {
              if (pr.first == q.uid) { partner = pr.second; break; }
              if (pr.second == q.uid) { partner = pr.first; break; }
            }
            const int pi = partner >= 0 ? find_uid(live, partner) : -1;
            if (pi >= 0) {
              const Mat4 diff = mat4_mul(mat4_load(live[(size_t)pi].p.tracking_result.transformation), mat4_load(r.transformation));
              const double tn = std::sqrt(diff.at(0, 3) * diff.at(0, 3) + diff.at(1, 3) * diff.at(1, 3) + diff.at(2, 3) * diff.at(2, 3));
              vote.value = tn, vote.reject = tn <= thr ? 0 : 1;
            }
          } break;
        }
        q.p.votes[q.p.n_votes++] = vote;
        if (vote.reject) break;
      }
    }
    // drop the worse half of every cross-validation pair, voters in reverse order (validator.cpp:160-162, voter.cpp:53-71)
    for (int v = stage.n_voters - 1; v >= 0; --v)
      for (const auto &pr : pairs[(size_t)v]) {
        const int a = find_uid(live, pr.first), b = find_uid(live, pr.second);
        if (a < 0 || b < 0) {  // its partner is already gone: nothing to compare against, the reference erases by pointer
          const int worse_uid = a < 0 ? pr.first : pr.second;
          const int w = find_uid(live, worse_uid);
          if (w >= 0) live.erase(live.begin() + w);
          continue;
        }
        const bool keep_first = total_score(live[(size_t)a]) >= total_score(live[(size_t)b]) && !rejected(live[(size_t)a]);
        live.erase(live.begin() + (keep_first ? b : a));
      }
    // remove rejected proposals (validator.cpp:88)
    {
      size_t w = 0;
      for (size_t i = 0; i < live.size(); ++i)
        if (!rejected(live[i])) {
          if (w != i) live[w] = live[i];
          ++w;
        }
      live.resize(w);
    }
    // keepBest (validator.cpp:104-130): among proposals between the same two frames the best total score survives, at the
    // position of the first of them
    if (stage.only_keep_best)
      for (size_t i = 0; i < live.size(); ++i)
        for (size_t k = i + 1; k < live.size();) {
          if (same_frames(live[i], live[k])) {
            if (total_score(live[k]) > total_score(live[i])) std::swap(live[i], live[k]);
            live.erase(live.begin() + (long)k);
          } else {
            ++k;
          }
        }
    // the next stage starts from this stage's estimate (validator.cpp:95-100)
    for (Prop &q : live) {
      const Mat4 init = affine_inverse(mat4_load(q.p.tracking_result.transformation));
      std::memcpy(q.p.initial_transformation, init.m, sizeof(init.m));
    }
  }
  const int rc_restore = dvo_amd_configure(ctx, &saved);
  if (rc) return rc;
  if (rc_restore) return rc_restore;
  for (size_t i = 0; i < live.size(); ++i) proposals[i] = live[i].p;
  *n_out = (int)live.size();
  return DVO_AMD_OK;
}

}  // extern "C"
