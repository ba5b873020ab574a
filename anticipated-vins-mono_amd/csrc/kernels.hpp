// kernels.hpp — argument blocks shared by the host C-ABI layer (avm_api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/avm.h"

namespace avm {

// compile-time problem limits of the window-solve kernel (WINDOW_SIZE = 10 => 11 frames)
constexpr int NFR = AVM_NFRAMES;       // 11
// The solve kernel is compiled twice from the same source (csrc/Makefile):
//   window_solve.o    the reference's default configuration (ESTIMATE_EXTRINSIC = 0, ESTIMATE_TD = 0, no relocalization frame):
//                     f-block = pose 66 | speed-bias 99 = 165 columns
//   window_solve_x.o  (-DAVM_X) every optional member of the problem: the dense "pose-like" part grows to
//                     pose 66 | relo_Pose 6 | ex_pose 6 | td 1 = 79 columns (relo_Pose is treated as a 12th frame), then
//                     speed-bias 99 = 178 columns; members that are switched off keep a unit diagonal
#ifdef AVM_X
constexpr int NFRP = NFR + 1;          // pose-like frames of the projection factors: 11 window frames + relo_Pose
constexpr int NPOSE = NFRP * 6 + 7;    // 79 dense columns in front of the speed-biases
constexpr int XC_EX = NFRP * 6;        // 72: first ex_pose column
constexpr int XC_TD = XC_EX + 6;       // 78: the td column
#else
constexpr int NFRP = NFR;
constexpr int NPOSE = NFR * 6;         // 66 pose columns (local)
#endif
constexpr int NF = NPOSE + NFR * 9;    // 165 (178) f-block columns: poses (+ relo, ex, td) first, then speed-bias
constexpr int SB0 = NPOSE;             // column of speedbias[0]
constexpr int MAXE = 150;              // inverse depths (e-blocks)
constexpr int NCOL = NF + MAXE;        // 315 (328)
constexpr int MAXOBS = MAXE * NFR;     // 1650 observation slots
constexpr int MAXPRIOR = 96;           // prior residual dimension limit
constexpr int MAXKEEP = 76;            // rows a marginalization may keep (prior_eig.hip holds A' and V in half a CU's LDS); this problem keeps <= 75
constexpr int MAXPBLK = 16;

struct PreintArgs {
  int n_windows, max_samp;
  const int32_t* imu_n;
  const double *imu_dt, *imu_acc, *imu_gyr, *imu_lin_ba, *imu_lin_bg;
  double acc_n, gyr_n, acc_w, gyr_w;
  double *out_delta, *out_jacobian, *out_covariance, *out_sum_dt, *out_sqrt_info;
};

// per-slot global scratch layout (doubles), one slot per resident workgroup (stays L2 resident)
// (one layout for both builds of the solve kernel and the marginalization kernel: the slots are shared)
struct Scratch {
  static constexpr size_t PF = 0;                               // per-factor products, feature-major: [14..15 quantities][frames][152] (solve), [8 + 7][11][152] (marginalization)
  static constexpr size_t PART = PF + 17 * (size_t)MAXOBS;       // per (frame b, start a) partial blocks: [12][11][69] + [12][35]
  static constexpr size_t IJRAW = PART + 9600;                  // [10][15][31] IMU residual + Jacobian before sqrt_info
  static constexpr size_t W = IJRAW + 4656;                     // E^T F, feature-major: Wt[80][152] (solve), Wt[73][152] (marginalization)
  static constexpr size_t HP = W + (size_t)80 * 152;            // [MAXPRIOR][MAXPRIOR] J0^T J0
  static constexpr size_t TOTAL = HP + (size_t)MAXPRIOR * MAXPRIOR;
};
constexpr int ISCRATCH = MAXOBS + (NFR + 1) * MAXE;  // ints per slot: observation slot -> feature, then cov[12][150] (row 11: the relocalization frame)

constexpr int PROF_SLOTS = 64;
struct SolveArgs {
  avm_window_batch b;  // device pointers
  avm_options opt;
  const double *pre_delta, *pre_jac, *pre_sqrt, *pre_sum_dt;  // [B][10][...]
  double* scratch;                                            // [n_slots][Scratch::TOTAL]
  int32_t* iscratch;                                          // [n_slots][ISCRATCH]
  avm_solve_summary* summary;                                 // [B] or null
  int n_slots;
  long long* prof;  // optional [n_slots][PROF_SLOTS] per-phase shader-clock accumulators (debug)
  int speculate;    // 1: evaluate the Jacobian at the candidate directly while steps keep being accepted (window_solve.hip)
  long long time_cap_ticks;  // avm_options::max_solver_time_s in ticks of the device wall clock (wall_clock64); 0 = no cap
};

struct EvalArgs {
  avm_window_batch b;
  avm_options opt;
  const double *pre_delta, *pre_jac, *pre_sqrt, *pre_sum_dt;
  int apply_loss;
  double *proj_r, *proj_J, *imu_r, *imu_J, *prior_res, *cost;
};

// device work buffers of the feature selector (owned by the ctx)
constexpr int FS_SYNC_INTS = 64 + 16 * 32 + 16 * 2 * 2 * 512 * 4;  // header, 16 team headers, per team (fValue, ub) x two parities x 512 16-byte records
constexpr int FS_MAX_CLOUD = 4096;  // depth-cloud points per frame the kd-tree builder has LDS for (28 bytes each; the reference's cloud is the window's <= 150 landmarks)
struct FselBuffers {
  double *C, *dpp, *consts, *delta, *delta_pk, *ddiag, *delta_u, *fval, *ub;
  double* kd;  // [P][8 + 11 max_cloud] the frames' kd-trees over their depth clouds (csrc/fsel.hip, fsel_kdtree_kernel)
  int32_t *valid, *valid_u, *black, *nsel, *done, *live, *pos, *nlive;
  int32_t* sync;  // [FS_SYNC_INTS] slot counter / failure flag / cycle trace / per-slot records of the single-frame kernel (csrc/fsel.hip)
};

// ---- table validation (every entry point that takes tables runs it before any kernel indexes with them) -----------
// Which groups of an avm_window_batch an entry point reads.
enum { CHK_TRACKS = 1, CHK_IMU = 2, CHK_PRIOR = 4 };
// 0 = fine; otherwise the first violated rule (messages in avm_api.hip, table_rule_text)
enum { BAD_NFEAT = 1, BAD_TRACK = 2, BAD_ORDER = 3, BAD_OBS = 4, BAD_IMU = 5, BAD_PRIOR = 6, BAD_FSEL = 7 };

// The lowest-numbered violated rule of window w (0 = fine), looking at features e = first, first + stride, ... only, so
// that the host (first 0, stride 1) and a wavefront (first = lane, stride 64, then a min over the lanes) agree.
__host__ __device__ inline int check_window_tables(const avm_window_batch& B, int w, int what, int first = 0, int stride = 1) {
  int bad = 1 << 30;
  auto rule = [&](int r) { bad = r < bad ? r : bad; };
  if (what & CHK_TRACKS) {
    const int nf = B.n_feat[w];
    if (nf < 0 || nf > B.max_feat) return BAD_NFEAT;
    for (int e = first; e < nf; e += stride) {
      const size_t k = (size_t)w * B.max_feat + e;
      const int a = B.feat_start[k], no = B.feat_nobs[k], ob = B.feat_obs_begin[k];
      if (a < 0 || no < 1 || a + no > AVM_NFRAMES) rule(BAD_TRACK);
      if (e > 0 && a < B.feat_start[k - 1]) rule(BAD_ORDER);
      if (ob < 0 || ob + no > B.max_obs) rule(BAD_OBS);
    }
  }
  if (first == 0) {
    if (what & CHK_IMU)
      for (int j = 0; j < AVM_WINDOW_SIZE; j++) {
        const int n = B.imu_n[(size_t)w * AVM_WINDOW_SIZE + j];
        if (n < 0 || n > B.max_samp) rule(BAD_IMU);
      }
    if ((what & CHK_PRIOR) && B.prior_n) {
      const int pn = B.prior_n[w];
      if (pn < 0 || pn > B.max_prior) rule(BAD_PRIOR);
      else if (pn > 0) {
        const int nb = B.prior_nblk[w];
        if (nb < 1 || nb > B.max_pblk) rule(BAD_PRIOR);
        else {
          int off = 0;
          for (int k = 0; k < nb; k++) {
            const int kind = B.prior_blk_kind[(size_t)w * B.max_pblk + k], fr = B.prior_blk_frame[(size_t)w * B.max_pblk + k];
            if (kind < AVM_BLK_POSE || kind > AVM_BLK_TD || fr < 0 || fr >= AVM_NFRAMES) rule(BAD_PRIOR);
            off += kind == AVM_BLK_SPEEDBIAS ? 9 : (kind == AVM_BLK_TD ? 1 : 6);
          }
          if (off != pn) rule(BAD_PRIOR);
        }
      }
    }
  }
  return bad == (1 << 30) ? 0 : bad;
}

__host__ __device__ inline int check_fsel_tables(const avm_fsel_batch& b, int p) {
  if (b.n_cand[p] < 0 || b.n_cand[p] > b.max_cand) return BAD_FSEL;
  if (b.n_used && (b.n_used[p] < 0 || b.n_used[p] > b.max_used)) return BAD_FSEL;
  if (b.n_cloud && (b.n_cloud[p] < 0 || b.n_cloud[p] > b.max_cloud)) return BAD_FSEL;
  if (b.nr_imu[p] < 0) return BAD_FSEL;
  return 0;
}

// The throughput form of the solve (window_solve_tp.o) keeps the speed-bias rows of the system in their structural form plus ONE strip
// "the prior's speed-bias block x every pose" (window_solve.hip, s_off): a prior with more than one speed-bias block does not fit it.
// The reference never builds one (estimator.cpp:904-916 keeps para_SpeedBias[1] only); such a batch simply takes the other kernel.
// The throughput form of the marginalization (marginalize_tp_kernel, round 5) holds the joint system over poses | speed-biases 0, 1 |
// ex_pose | td only: a prior with a speed-bias block of a later frame does not fit it (the reference keeps frame 1's, as frame 0).
// Returns a bit mask of what window w's prior does NOT fit: bit 0 the throughput solve, bit 1 the throughput marginalization.
__host__ __device__ inline int window_prior_tp_misfit(const avm_window_batch& B, int w) {
  if (!B.prior_n || B.prior_n[w] <= 0) return 0;
  int nsb = 0, late = 0, later = 0;
  const int nb = B.prior_nblk[w];
  for (int k = 0; k < nb && k < B.max_pblk; k++)
    if (B.prior_blk_kind[(size_t)w * B.max_pblk + k] == AVM_BLK_SPEEDBIAS) {
      const int fr = B.prior_blk_frame[(size_t)w * B.max_pblk + k];
      nsb++, late |= fr > 1, later |= fr > 0;
    }
  // (the throughput solve eliminates the speed-bias blocks last frame first, so that the one block the prior couples to every pose comes
  //  last of them: window_solve.hip, chol_regs - a prior whose block is not frame 0's would fill the factor in)
  return (nsb > 1 || later ? 1 : 0) | (late ? 2 : 0);
}

// first_bad: TWO ints.  [0]: INT_MAX when every window / problem passes, else (index * 8 + rule) of the lowest failing index;
// [1] (windows, with CHK_PRIOR): the OR of window_prior_tp_misfit() over the windows (left alone when every prior fits)
hipError_t launch_validate_windows(const avm_window_batch& b, int what, int* first_bad, hipStream_t stream);
hipError_t launch_validate_fsel(const avm_fsel_batch& b, int* first_bad, hipStream_t stream);

void launch_preint(const PreintArgs& a, hipStream_t stream);
hipError_t launch_fsel(const avm_fsel_batch& b, const FselBuffers& w, const avm_fsel_out& out, double* omega_out, bool run_rounds,
                       int frame_mode, const int* vflag, hipStream_t stream);
bool fsel_horizon_supported(int H);
hipError_t launch_fsel_build_cloud(const avm_window_batch& b, const double* k1_pos, const double* k1_quat, int max_cloud, int32_t* n_cloud,
                                   double* cloud_xy, double* cloud_depth, hipStream_t stream);
hipError_t launch_fsel_nn_depth(const avm_fsel_batch& b, double* kd, double* depth_out, hipStream_t stream);
size_t fsel_kd_doubles(const avm_fsel_batch& b);  // doubles of FselBuffers::kd / the kd argument above for this batch
hipError_t launch_fsel_horizon_imu(const avm_fsel_horizon_in& in, double* hor_pos, double* hor_quat, hipStream_t stream);
hipError_t launch_triangulate(const avm_window_batch& b, double init_depth, hipStream_t stream);
hipError_t launch_imu_propagate(const avm_window_batch& b, const double* g, hipStream_t stream);
hipError_t launch_slide_window(const avm_window_batch& b, int flag, int shift_depth, double init_depth, int* err, hipStream_t stream);
hipError_t launch_projection_td_eval(const avm_td_factor_batch& f, double* residual, double* jac, hipStream_t stream);
hipError_t launch_window_solve(const SolveArgs& a, hipStream_t stream);
hipError_t launch_window_solve_x(const SolveArgs& a, hipStream_t stream);  // window_solve_x.o: estimate_extrinsic / estimate_td / relocalization
int window_solve_x_lds_bytes();
hipError_t launch_window_solve_tp(const SolveArgs& a, hipStream_t stream);  // window_solve_tp.o: two 256-thread workgroups per CU (large batches)
int window_solve_tp_lds_bytes();
int window_solve_tp_occupancy();
int window_solve_tp_pattern(int* out);  // the throughput factorization's tile pattern / ownership / elimination order (tests)
int window_solve_pattern(int* out);     // ... the latency build's (the same pattern from the packed triangle, same owners)
int window_solve_x_pattern(int* out);   // ... the extended build's (twelve tile columns on eight wavefronts)
hipError_t launch_eval_factors(const EvalArgs& a, hipStream_t stream);
// scale: [n_windows][po.max_prior] device array: the magnitude every diagonal entry of A' was formed at (for launch_prior_eig's noise test)
hipError_t launch_marginalize(const SolveArgs& a, const avm_prior_out& po, int* err, double* scale, hipStream_t stream);
// window_solve_tp.o: the same marginalization as two 256-thread workgroups per CU (a.n_slots = the throughput solve's 2 x CUs slots)
hipError_t launch_marginalize_tp(const SolveArgs& a, const avm_prior_out& po, int* err, double* scale, hipStream_t stream);
// second half of the marginalization: eigen-decomposition of A' (left in po.J / po.r by launch_marginalize) -> sqrt prior
// noise_rel: avm_options::marg_noise_rel (0 = the reference-literal clamp S > eps and nothing else)
hipError_t launch_prior_eig(const avm_prior_out& po, int n_windows, double eps, double noise_rel, const double* scale, long long* prof,
                            int* done /* [n_windows] device scratch, may be null */, hipStream_t stream);
int window_solve_lds_bytes();

}  // namespace avm
