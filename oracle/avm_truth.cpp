// oracle/avm_truth.cpp - TEST INFRASTRUCTURE ONLY.  The extended-precision ARBITER of the marginalization parity tests
// (VERDICT round 2, item 2): the oracle's own restatement of
//   MarginalizationInfo::addResidualBlockInfo / preMarginalize / marginalize   marginalization_factor.cpp:89-297
//   as driven by Estimator::optimization()                                       estimator.cpp:817-990
// (oracle/solver.hpp: marginalize(), with the factors of oracle/factors.hpp and the Eigen restatements of linalg.hpp)
// compiled with __float128 as the scalar type.  It follows the reference's algorithm literally - the joint
// SelfAdjointEigenSolver pseudo-inverse of Amm with the 1e-8 clamp (:267-272), the Schur complement (:275-281), the eigen
// square root (:283-291) - on the SAME FP64 inputs (states, observations, raw IMU samples, the old prior), from the
// pre-integration on.  With a unit roundoff of 1e-34 against entries that span 1e12 its rounding noise (1e-22) is far below
// the 1e-8 clamp and below anything FP64 can resolve: its result, rounded once to FP64 at the very end, is the value the
// reference's algorithm DEFINES for these inputs.  The FP64 oracle and the GPU are both measured against it.
//
// Nothing under anticipated-vins-mono_amd/ may include, link or call this file.
#include "quad_prelude.hpp"

#include "window_io.hpp"  // (-> solver.hpp -> factors.hpp -> linalg.hpp, all in the wide scalar type from here on)
#include "fsel_io.hpp"    // (-> fsel.hpp)

#undef double

using namespace avmo;

extern "C" {

// The marginalization of window w of the batch at the state the batch holds (what avmo_window_solve_batch does after a solve
// with max_num_iterations = 0, minus the gauge-fix round trip, which is the identity on these states up to rounding).
// Outputs (all FP64, rounded from binary128 at the end; any pointer may be NULL):
//   n_out, nblk_out, blk_kind / blk_frame [max_pblk], x0 [max_pblk][9]
//   J [n][n] row-major linearized_jacobians, r [n] linearized_residuals          (eigenvector signs are arbitrary)
//   H [n][n] = J^T J, g [n] = J^T r, cost = 1/2 |r|^2                             (what a consumer of the prior sees)
//   A [n][n], b [n]: the Schur complement before the square root
//   ev_mm [cap_mm], n_mm: eigenvalues of Amm (ascending); ev_rr [n]: eigenvalues of the Schur complement
// returns 0, or -1 when MARGIN_SECOND_NEW has nothing to drop (the old prior stays, estimator.cpp:926-927).
int avmt_marginalize(const avm_options* opt, const avm_window_batch* batch, int w, int max_pblk, int32_t* n_out, int32_t* nblk_out,
                     int32_t* blk_kind, int32_t* blk_frame, double* x0, double* J, double* r, double* H, double* g, double* cost,
                     double* A, double* b, double* ev_mm, int cap_mm, int32_t* n_mm, double* ev_rr) {
  Window W;
  load_window(*opt, *batch, w, W);
  Prior np;
  MargDiag D;
  marginalize(W, W.x, *opt, np, &D);
  if (n_out) *n_out = np.n;
  if (np.n < 0) return -1;
  const int n = np.n, nb = (int)np.blk_kind.size();
  if (nblk_out) *nblk_out = nb;
  for (int k = 0; k < nb && k < max_pblk; k++) {
    if (blk_kind) blk_kind[k] = np.blk_kind[k];
    if (blk_frame) blk_frame[k] = np.blk_frame[k];
    if (x0)
      for (size_t q = 0; q < np.x0[k].size(); q++) x0[(size_t)k * 9 + q] = (double)np.x0[k][q];
  }
  for (int i = 0; i < n; i++) {
    if (r) r[i] = (double)np.r[i];
    for (int j = 0; j < n; j++) {
      if (J) J[(size_t)i * n + j] = (double)np.J(i, j);
      if (A) A[(size_t)i * n + j] = (double)D.A_rr(i, j);
    }
    if (b) b[i] = (double)D.b_rr[i];
    if (ev_rr) ev_rr[i] = (double)D.ev_rr[i];
  }
  if (H || g || cost) {
    avmo_real c = 0;
    for (int i = 0; i < n; i++) {
      c += np.r[i] * np.r[i];
      avmo_real gi = 0;
      for (int k = 0; k < n; k++) gi += np.J(k, i) * np.r[k];
      if (g) g[i] = (double)gi;
      if (H)
        for (int j = 0; j < n; j++) {
          avmo_real h = 0;
          for (int k = 0; k < n; k++) h += np.J(k, i) * np.J(k, j);
          H[(size_t)i * n + j] = (double)h;
        }
    }
    if (cost) *cost = (double)(c / 2);
  }
  if (n_mm) *n_mm = D.m;
  if (ev_mm)
    for (int i = 0; i < D.m && i < cap_mm; i++) ev_mm[i] = (double)D.ev_mm[i];
  return 0;
}

// The solve of window w in the wide scalar type: Problem::build + trust_region_solve (the restated Ceres 1.14 dogleg minimizer,
// oracle/solver.hpp) + the gauge-fix round trip of double2vector (estimator.cpp:206-284), from the pre-integration on, on the
// same FP64 inputs.  What comes back, rounded once to FP64, is the state the reference's ALGORITHM defines for these inputs when
// no arithmetic error is made on the way; the FP64 oracle and the GPU are both measured against it (tests/test_solve_truth.py).
// Outputs (any pointer may be NULL): pose [11][7], speedbias [11][9], ex_pose [7], td [1], relo_pose [7], inv_depth [n_feat];
// summary: the oracle's record (costs, iterations, termination), its doubles rounded from binary128.
int avmt_solve(const avm_options* opt, const avm_window_batch* batch, int w, double* pose, double* speedbias, double* ex_pose, double* td,
               double* relo_pose, double* inv_depth, avm_solve_summary* summary) {
  Window W;
  load_window(*opt, *batch, w, W);
  Problem P;
  P.build(W, *opt);
  SolveResult R = trust_region_solve(P, W.x);
  State out;
  out.lam = R.x.lam;
  gauge_fix_roundtrip(W.x, R.x, out, W.failure_occur ? W.last_pose0 : nullptr, W.has_relo);
  for (int f = 0; f < AVM_NFRAMES; f++) {
    if (pose)
      for (int k = 0; k < 7; k++) pose[f * 7 + k] = (double)out.pose[f][k];
    if (speedbias)
      for (int k = 0; k < 9; k++) speedbias[f * 9 + k] = (double)out.sb[f][k];
  }
  for (int k = 0; k < 7; k++) {
    if (ex_pose) ex_pose[k] = (double)out.ex[k];
    if (relo_pose && W.has_relo) relo_pose[k] = (double)out.relo[k];
  }
  if (td) *td = (double)out.td;
  if (inv_depth)
    for (size_t e = 0; e < out.lam.size(); e++) inv_depth[e] = (double)out.lam[e];
  if (summary) *summary = R.sum;
  return 0;
}

// The same solve with its result handed out BEYOND FP64: Ceres' own solution (before double2vector's gauge fix) as pose [11][7] |
// speed-bias [11][9] | inverse depths [n_feat] | ex_pose [7] | td [1] | relo_Pose [7], and the cost at the start point and at the solution, each as a double-double pair
// hi + lo (the binary128 value to 32 digits), followed by the cost after every iteration.  For tests/test_solve_trace_mp.py: the 50-digit run of the independent numpy minimizer
// (tests/golden/gen_solve_trace_mp.py) against this restatement.  cost_hi / cost_lo: [2 + AVM_MAX_ITER_TRACE]; returns the number of iterations.
int avmt_solve_dd(const avm_options* opt, const avm_window_batch* batch, int w, double* x_hi, double* x_lo, double* cost_hi, double* cost_lo,
                  int32_t* accept_mask) {
  Window W;
  load_window(*opt, *batch, w, W);
  Problem P;
  P.build(W, *opt);
  SolveResult R = trust_region_solve(P, W.x);
  auto put = [](avmo_real v, double* hi, double* lo) {
    *hi = (double)v;
    *lo = (double)(v - (avmo_real)*hi);
  };
  size_t q = 0;
  for (int f = 0; f < AVM_NFRAMES; f++)
    for (int k = 0; k < 7; k++, q++) put(R.x.pose[f][k], x_hi + q, x_lo + q);
  for (int f = 0; f < AVM_NFRAMES; f++)
    for (int k = 0; k < 9; k++, q++) put(R.x.sb[f][k], x_hi + q, x_lo + q);
  for (size_t e = 0; e < R.x.lam.size(); e++, q++) put(R.x.lam[e], x_hi + q, x_lo + q);
  for (int k = 0; k < 7; k++, q++) put(R.x.ex[k], x_hi + q, x_lo + q);   // (the optional members: constants of the base problem)
  put(R.x.td, x_hi + q, x_lo + q), q++;
  for (int k = 0; k < 7; k++, q++) put(R.x.relo[k], x_hi + q, x_lo + q);
  put(P.evaluate(W.x, false), cost_hi, cost_lo);
  put(P.evaluate(R.x, false), cost_hi + 1, cost_lo + 1);
  for (size_t k = 0; k < R.cost_after.size(); k++) put(R.cost_after[k], cost_hi + 2 + k, cost_lo + 2 + k);
  if (accept_mask) *accept_mask = R.sum.accept_mask;
  return R.sum.num_iterations;
}

// FeatureSelector::select of frame p of the batch in the wide scalar type (oracle/fsel.hpp: calcInfoFromRobotMotion,
// calcInfoFromFeatures, the lazy greedy of selectInformativeFeatures with sortedlogDetUB's std::map, Utility::logdet by Cholesky -
// feature_selector.cpp:239-728): the ids the reference's algorithm selects when no logdet comparison is decided by rounding.
// selected_ids [max_features], fvalues [max_features] (rounded to FP64; may be NULL); returns the number selected.
int avmt_fsel_select(const avm_fsel_batch* batch, int p, int32_t* selected_ids, double* fvalues) {
  FselProblem P;
  load_fsel(*batch, p, P);
  FselResult R = fsel_select(P);
  for (size_t i = 0; i < R.selected.size(); i++) {
    selected_ids[i] = R.selected[i];
    if (fvalues) fvalues[i] = (double)R.fvalues[i];
  }
  return (int)R.selected.size();
}

// Omega (calcInfoFromRobotMotion incl. addOmegaPrior) [N][N] and every candidate's Delta [max_cand][N][N] (zero when the
// candidate is not triangulable) of frame p, formed in binary128 and rounded to FP64 (either pointer may be NULL).
int avmt_fsel_information(const avm_fsel_batch* batch, int p, double* omega, double* delta_full) {
  FselProblem P;
  load_fsel(*batch, p, P);
  const int N = 9 * (P.H + 1);
  Mat Om = calcInfoFromRobotMotion(P);
  if (omega)
    for (int i = 0; i < N * N; i++) omega[i] = (double)Om.a[i];
  if (delta_full) {
    std::map<int, Mat> D = calcInfoFromFeatures(P, P.cand_id, P.cand_x, P.cand_y);
    for (size_t c = 0; c < P.cand_id.size(); c++) {
      auto it = D.find(P.cand_id[c]);
      for (int i = 0; i < N * N; i++) delta_full[c * N * N + i] = it == D.end() ? 0.0 : (double)it->second.a[i];
    }
  }
  return 0;
}

int avmt_digits(void) { return FLT128_DIG; }

}  // extern "C"
