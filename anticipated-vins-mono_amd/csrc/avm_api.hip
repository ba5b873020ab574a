// avm_api.hip — the C ABI of include/avm.h on top of the gfx950 kernels.
// Host language is C++ (the reference's host code is C++: vins_estimator/src/estimator.cpp,
// feature_selector.cpp).  No torch types, no exceptions across the boundary.  There is no CPU
// fallback: without a HIP device avm_create() fails with AVM_ERR_NO_DEVICE.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is dlopen'ed (avm_comm_*), a single-GPU host never needs it

#include "kernels.hpp"

using namespace avm;

namespace {
// raw rccl.h entry points, resolved on first use
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;  // every symbol resolved; decided once for the process
  std::once_flag once;
  bool load() {
    std::call_once(once, [this] {
      for (const char* name : {"librccl.so.1", "librccl.so"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);  // (a process that already maps an RCCL under this soname, e.g. PyTorch's, gets that one)
        if (lib) break;
      }
      if (!lib) return;
      GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
      CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
      AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
      CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
      GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
      ok = GetUniqueId && CommInitRank && AllGather && CommDestroy && GetErrorString;
      if (!ok) {  // a library without the five entry points is no RCCL for us: never call through a null pointer later
        dlclose(lib);
        lib = nullptr;
      }
    });
    return ok;
  }
};
Rccl& rccl() {
  static Rccl r;
  return r;
}
}  // namespace

struct avm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  int n_slots = 0;
  // device buffers owned by the ctx
  double* scratch = nullptr;
  int32_t* iscratch = nullptr;
  int64_t n_allocs = 0;       // device / pinned (re)allocations since avm_create (avm_debug_counters: a timed region should see none)
  int last_marg_windows = 0;  // batch size of the last marginalization (its per-window "the one-wavefront kernel finished it" flags: pool "pe_done")
  bool last_solve_tp = false;  // which form of the solve kernel the last avm_window_solve_batch took (avm_debug_last_solve_form)
  bool last_marg_tp = false;   // ... and which form of the marginalization kernel (avm_debug_last_marg_form)
  int scratch_slots = 0;  // slots allocated (n_slots, or 2 n_slots once a batch has taken the throughput form of the solve)
  double *pre_delta = nullptr, *pre_jac = nullptr, *pre_cov = nullptr, *pre_sqrt = nullptr, *pre_sum = nullptr;
  size_t pre_cap = 0;  // windows
  avm_solve_summary* d_summary = nullptr;
  long long* prof = nullptr;  // [n_slots][32], enabled by AVM_PROFILE=1
  size_t summary_cap = 0;
  // staging pool for AVM_MEM_HOST calls: name -> (ptr, bytes)
  std::map<std::string, std::pair<void*, size_t>> pool;
  // pinned host staging (small AVM_MEM_HOST batches travel as one packed copy each way): name -> (ptr, bytes)
  std::map<std::string, std::pair<void*, size_t>> pinned;
  bool packed_in = false;  // the last stage_window_batch took the packed path (states are contiguous on the device)
  hipEvent_t ev[8];
  hipEvent_t ev_flag = nullptr;  // recorded behind the copy of a table check's verdict (validate_windows_begin)
  std::map<std::string, float> last_ms;
  int last_fsel_mode = -1;  // the form the last avm_fsel_select_batch took (3: fsel_solo_kernel)
  int64_t last_fsel_evals = -1;  // candidate evaluations the last select executed on the device (solo form: counted by the kernel; else -1)
  int fsel_frame_mode = 2;  // how a single-frame select runs (avm_fsel_select_batch); AVM_FSEL_FRAME=0/1/2 caps it
  ncclComm_t comm = nullptr;  // avm_comm_init
  int comm_ranks = 0, comm_rank = 0;
  double wall_clock_hz = 1.0e8;  // rate of wall_clock64() (hipDeviceAttributeWallClockRate; 100 MHz on gfx950)
  // fallbacks of the selector's all-rounds-in-one-launch kernel (avm_fsel_fallback_stats)
  int64_t fsel_calls = 0, fsel_reruns = 0, fsel_failed_launches = 0;
  int fsel_cooldown = 0;  // calls left before a degraded ctx probes the fast mode again
  int fsel_backoff = 16;  // the next cool-down (doubles with every failed probe up to 4096 calls, back to 16 after a fast-mode call that went through)
};

namespace {

#define HIPCHK(ctx, call)                                                                  \
  do {                                                                                     \
    hipError_t e__ = (call);                                                               \
    if (e__ != hipSuccess) {                                                               \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                     \
      return AVM_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)

int fail(avm_ctx* c, int code, const char* msg) {
  c->err = msg;
  return code;
}

void* pool_get(avm_ctx* c, const std::string& name, size_t bytes) {
  auto& e = c->pool[name];
  if (e.second < bytes || e.first == nullptr) {
    c->n_allocs++;
    if (e.first) (void)hipFree(e.first);
    e.first = nullptr;
    if (hipMalloc(&e.first, bytes ? bytes : 8) != hipSuccess) return nullptr;
    e.second = bytes;
  }
  return e.first;
}

void* pinned_get(avm_ctx* c, const std::string& name, size_t bytes) {
  auto& e = c->pinned[name];
  if (e.second < bytes || e.first == nullptr) {
    c->n_allocs++;
    if (e.first) (void)hipHostFree(e.first);
    e.first = nullptr;
    if (hipHostMalloc(&e.first, bytes ? bytes : 8, hipHostMallocDefault) != hipSuccess) return nullptr;
    e.second = bytes;
  }
  return e.first;
}

const char* table_rule_text(int rule) {
  switch (rule) {
    case BAD_NFEAT: return "n_feat outside [0, max_feat]";
    case BAD_TRACK: return "a feature track leaves the window (need start >= 0, nobs >= 1, start + nobs <= 11)";
    case BAD_ORDER: return "feat_start must be non-decreasing in the feature index (std::list order)";
    case BAD_OBS: return "feat_obs_begin + feat_nobs runs past max_obs";
    case BAD_IMU: return "imu_n outside [0, max_samp]";
    case BAD_PRIOR: return "prior tables inconsistent (prior_n / prior_nblk ranges, block kinds / frames, sizes must add up to prior_n)";
    case BAD_FSEL: return "n_cand / n_used / n_cloud outside their strides, or nr_imu < 0";
  }
  return "bad table";
}

int report_bad(avm_ctx* c, int first_bad, const char* unit) {
  c->err = std::string(unit) + " " + std::to_string(first_bad / 8) + ": " + table_rule_text(first_bad % 8);
  return AVM_ERR_INVALID;
}

// Runs before any kernel indexes with the caller's tables: host tables are checked on the host, device-resident ones by a
// one-thread-per-window kernel whose 4-byte verdict is read back (the only extra synchronization of a device-mode call).
// tp_misfit (optional, with CHK_PRIOR): bit 0 set when some window's prior does not fit the throughput form of the solve, bit 1 when one
// does not fit the throughput form of the marginalization (kernels.hpp, window_prior_tp_misfit)
int validate_windows(avm_ctx* c, avm_mem mem, const avm_window_batch* b, int what, int* tp_misfit = nullptr) {
  if (tp_misfit) *tp_misfit = 0;
  if ((what & CHK_TRACKS) && (!b->n_feat || !b->feat_start || !b->feat_nobs || !b->feat_obs_begin)) return fail(c, AVM_ERR_INVALID, "null feature table");
  if ((what & CHK_IMU) && !b->imu_n) return fail(c, AVM_ERR_INVALID, "null imu_n");
  if ((what & CHK_PRIOR) && b->prior_n && (!b->prior_nblk || !b->prior_blk_kind || !b->prior_blk_frame)) return fail(c, AVM_ERR_INVALID, "null prior table");
  if (mem == AVM_MEM_HOST) {
    for (int w = 0; w < b->n_windows; w++) {
      const int rule = check_window_tables(*b, w, what);
      if (rule) return report_bad(c, w * 8 + rule, "window");
      if (tp_misfit && (what & CHK_PRIOR)) *tp_misfit |= window_prior_tp_misfit(*b, w);
    }
    return AVM_OK;
  }
  int* flag = static_cast<int*>(pool_get(c, "v_flag", 2 * sizeof(int)));
  if (!flag) return fail(c, AVM_ERR_HIP, "hipMalloc failed (validation flag)");
  HIPCHK(c, hipMemsetAsync(flag, 0x7f, sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(flag + 1, 0, sizeof(int), c->stream));
  HIPCHK(c, launch_validate_windows(*b, what, flag, c->stream));
  int h[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(h, flag, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (tp_misfit) *tp_misfit = h[1];
  return h[0] == 0x7f7f7f7f ? AVM_OK : report_bad(c, h[0], "window");
}

// The same check for a device-resident batch of avm_window_solve_batch, in two halves: _begin enqueues the check, the copy of its
// verdict into pinned memory and an event; _end waits for that event only.  What the caller enqueues in between (the pre-integration,
// which clamps the one table entry it indexes with) runs while the host reads the verdict and prepares the next launches, instead of
// the device idling through a blocking round trip at the top of every call.  On a failed check _end drains the stream before it
// reports, so the caller's buffers are no longer being read when the error returns.
int validate_windows_begin(avm_ctx* c, const avm_window_batch* b, int what, int** host_flag) {
  if ((what & CHK_TRACKS) && (!b->n_feat || !b->feat_start || !b->feat_nobs || !b->feat_obs_begin)) return fail(c, AVM_ERR_INVALID, "null feature table");
  if ((what & CHK_IMU) && !b->imu_n) return fail(c, AVM_ERR_INVALID, "null imu_n");
  if ((what & CHK_PRIOR) && b->prior_n && (!b->prior_nblk || !b->prior_blk_kind || !b->prior_blk_frame)) return fail(c, AVM_ERR_INVALID, "null prior table");
  int* flag = static_cast<int*>(pool_get(c, "v_flag", 2 * sizeof(int)));
  int* h = static_cast<int*>(pinned_get(c, "v_flag_h", 2 * sizeof(int)));
  if (!flag || !h) return fail(c, AVM_ERR_HIP, "allocation failed (validation flag)");
  HIPCHK(c, hipMemsetAsync(flag, 0x7f, sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(flag + 1, 0, sizeof(int), c->stream));
  HIPCHK(c, launch_validate_windows(*b, what, flag, c->stream));
  HIPCHK(c, hipMemcpyAsync(h, flag, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipEventRecord(c->ev_flag, c->stream));
  *host_flag = h;
  return AVM_OK;
}
int validate_windows_end(avm_ctx* c, const int* host_flag, int* tp_misfit) {
  HIPCHK(c, hipEventSynchronize(c->ev_flag));
  if (tp_misfit) *tp_misfit = host_flag[1];
  if (host_flag[0] == 0x7f7f7f7f) return AVM_OK;
  (void)hipStreamSynchronize(c->stream);
  return report_bad(c, host_flag[0], "window");
}

int validate_fsel(avm_ctx* c, avm_mem mem, const avm_fsel_batch* b) {
  if (!b->n_cand || !b->nr_imu) return fail(c, AVM_ERR_INVALID, "null n_cand / nr_imu");
  if (mem == AVM_MEM_HOST) {
    for (int p = 0; p < b->n_problems; p++) {
      const int rule = check_fsel_tables(*b, p);
      if (rule) return report_bad(c, p * 8 + rule, "frame");
    }
    return AVM_OK;
  }
  int* flag = static_cast<int*>(pool_get(c, "v_flag", 2 * sizeof(int)));
  if (!flag) return fail(c, AVM_ERR_HIP, "hipMalloc failed (validation flag)");
  HIPCHK(c, hipMemsetAsync(flag, 0x7f, sizeof(int), c->stream));
  HIPCHK(c, launch_validate_fsel(*b, flag, c->stream));
  int h = 0;
  HIPCHK(c, hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return h == 0x7f7f7f7f ? AVM_OK : report_bad(c, h, "frame");
}

template <class T>
int stage_in(avm_ctx* c, const char* name, const T* host, size_t count, const T** dev) {
  if (!host || count == 0) {
    *dev = nullptr;
    return AVM_OK;
  }
  void* d = pool_get(c, name, count * sizeof(T));
  if (!d) return fail(c, AVM_ERR_HIP, "hipMalloc failed (staging)");
  HIPCHK(c, hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
  *dev = static_cast<const T*>(d);
  return AVM_OK;
}

int ensure_window_buffers(avm_ctx* c, int n_windows, bool tp = false) {
  // (the throughput form of the solve runs two workgroups per CU: twice the slots; allocated when a batch first takes it)
  const int want = tp ? 2 * c->n_slots : c->n_slots;
  if (!c->scratch || c->scratch_slots < want || (size_t)n_windows > c->pre_cap) c->n_allocs++;
  if (!c->scratch || c->scratch_slots < want) {
    if (c->scratch) (void)hipFree(c->scratch), c->scratch = nullptr;
    if (c->iscratch) (void)hipFree(c->iscratch), c->iscratch = nullptr;
    c->scratch_slots = 0;
    HIPCHK(c, hipMalloc(&c->scratch, sizeof(double) * Scratch::TOTAL * want));
    HIPCHK(c, hipMalloc(&c->iscratch, sizeof(int32_t) * ISCRATCH * want));
    HIPCHK(c, hipMemsetAsync(c->scratch, 0, sizeof(double) * Scratch::TOTAL * want, c->stream));
    HIPCHK(c, hipMemsetAsync(c->iscratch, 0, sizeof(int32_t) * ISCRATCH * want, c->stream));
    c->scratch_slots = want;
  }
  if ((size_t)n_windows > c->pre_cap) {
    c->pre_cap = 0;  // a failed hipMalloc below must not leave the old capacity next to freed / partial buffers
    for (double** p : {&c->pre_delta, &c->pre_jac, &c->pre_cov, &c->pre_sqrt, &c->pre_sum})
      if (*p) (void)hipFree(*p), *p = nullptr;
    const size_t iv = (size_t)n_windows * 10;
    HIPCHK(c, hipMalloc(&c->pre_delta, sizeof(double) * iv * 10));
    HIPCHK(c, hipMalloc(&c->pre_jac, sizeof(double) * iv * 225));
    HIPCHK(c, hipMalloc(&c->pre_cov, sizeof(double) * iv * 225));
    HIPCHK(c, hipMalloc(&c->pre_sqrt, sizeof(double) * iv * 225));
    HIPCHK(c, hipMalloc(&c->pre_sum, sizeof(double) * iv));
    c->pre_cap = n_windows;
  }
  return AVM_OK;
}

int check_window_batch(avm_ctx* c, const avm_options* opt, const avm_window_batch* b) {
  if (!opt || !b || b->n_windows < 0) return fail(c, AVM_ERR_INVALID, "null/negative argument");
  if (opt->estimate_td && (!b->obs_vel_td || !b->td))
    return fail(c, AVM_ERR_INVALID, "estimate_td != 0 needs obs_vel_td (velocity, cur_td, uv.y per observation) and td");
  if (opt->estimate_td && !(opt->row > 0.0)) return fail(c, AVM_ERR_INVALID, "estimate_td != 0 needs opt->row (image height) > 0");
  if (b->relo_n && (!b->relo_feat || !b->relo_xy || !b->relo_pose))
    return fail(c, AVM_ERR_INVALID, "relo_n is set but relo_feat / relo_xy / relo_pose is NULL");
  if (b->failure_occur && !b->last_pose0) return fail(c, AVM_ERR_INVALID, "failure_occur is set but last_pose0 is NULL");
  if (b->max_feat > MAXE) return fail(c, AVM_ERR_CAPACITY, "max_feat > 150");
  if (b->max_obs > MAXOBS) return fail(c, AVM_ERR_CAPACITY, "max_obs > 1650");
  if (b->max_prior > MAXPRIOR || b->max_pblk > MAXPBLK) return fail(c, AVM_ERR_CAPACITY, "prior larger than 96 / 16 blocks");
  if (opt->max_num_iterations > AVM_MAX_ITER_TRACE) return fail(c, AVM_ERR_CAPACITY, "max_num_iterations > 16");
  return AVM_OK;
}

// copy a host batch to the device; out = batch with device pointers
// (field, element type, elements) of every table of an avm_window_batch; the four state arrays come first so that
// the packed path can bring them back with one copy
#define AVM_WINDOW_FIELDS(X, B, h)                                                                                        \
  X(pose, double, (B) * 77) X(speedbias, double, (B) * 99) X(ex_pose, double, (B) * 7) X(inv_depth, double, (B) * (h)->max_feat) \
  X(n_feat, int32_t, (B)) X(feat_start, int32_t, (B) * (h)->max_feat) X(feat_nobs, int32_t, (B) * (h)->max_feat)           \
  X(feat_obs_begin, int32_t, (B) * (h)->max_feat) X(obs_xy, double, (B) * (h)->max_obs * 2) X(imu_n, int32_t, (B) * 10)    \
  X(imu_dt, double, (B) * 10 * (h)->max_samp) X(imu_acc, double, (B) * 10 * ((h)->max_samp + 1) * 3)                       \
  X(imu_gyr, double, (B) * 10 * ((h)->max_samp + 1) * 3) X(imu_lin_ba, double, (B) * 30) X(imu_lin_bg, double, (B) * 30)   \
  X(prior_n, int32_t, (B)) X(prior_nblk, int32_t, (B)) X(prior_blk_kind, int32_t, (B) * (h)->max_pblk)                     \
  X(prior_blk_frame, int32_t, (B) * (h)->max_pblk) X(prior_J, double, (B) * (h)->max_prior * (h)->max_prior)               \
  X(prior_r, double, (B) * (h)->max_prior) X(prior_x0, double, (B) * (h)->max_pblk * 9)                                        \
  X(obs_vel_td, double, (B) * (h)->max_obs * 4) X(td, double, (B)) X(relo_n, int32_t, (B)) X(relo_frame, int32_t, (B))            \
  X(relo_feat, int32_t, (B) * (h)->max_feat) X(relo_xy, double, (B) * (h)->max_feat * 2) X(relo_pose, double, (B) * 7)           \
  X(failure_occur, int32_t, (B)) X(last_pose0, double, (B) * 7)

constexpr size_t PACK_LIMIT = 4u << 20;  // batches below 4 MiB (a few windows: the real-time use) travel packed
inline size_t pack_up(size_t n) { return (n + 63) & ~size_t(63); }

// copy a host batch to the device; out = batch with device pointers.  Small batches go through one pinned buffer and
// one copy (22 pageable copies of a few hundred bytes each cost more than the solve's pre-integration kernel).
int stage_window_batch(avm_ctx* c, const avm_window_batch* h, avm_window_batch* d) {
  *d = *h;
  const size_t B = h->n_windows;
  int rc;
  size_t total = 0;
#define SZ(field, type, count) total += h->field ? pack_up(sizeof(type) * (count)) : 0;
  AVM_WINDOW_FIELDS(SZ, B, h)
#undef SZ
  c->packed_in = total <= PACK_LIMIT && h->pose && h->speedbias && h->ex_pose && h->inv_depth;
  if (c->packed_in) {
    char* hp = static_cast<char*>(pinned_get(c, "w_pack", total));
    char* dp = static_cast<char*>(pool_get(c, "w_pack", total));
    if (!hp || !dp) return fail(c, AVM_ERR_HIP, "allocation failed (packed staging)");
    size_t off = 0;
#define PK(field, type, count)                                             \
  if (h->field) {                                                          \
    std::memcpy(hp + off, h->field, sizeof(type) * (count));               \
    *(const type**)&d->field = reinterpret_cast<const type*>(dp + off);    \
    off += pack_up(sizeof(type) * (count));                                \
  }
    AVM_WINDOW_FIELDS(PK, B, h)
#undef PK
    HIPCHK(c, hipMemcpyAsync(dp, hp, total, hipMemcpyHostToDevice, c->stream));
    return AVM_OK;
  }
#define ST(field, type, count) \
  if ((rc = stage_in<type>(c, "w_" #field, h->field, (count), (const type**)&d->field)) != AVM_OK) return rc;
  AVM_WINDOW_FIELDS(ST, B, h)
#undef ST
  return AVM_OK;
}

// the four state arrays back to the caller (after the kernels, before the final synchronize)
int unstage_window_states(avm_ctx* c, const avm_window_batch* h, const avm_window_batch* d, char** pinned_states) {
  const size_t B = h->n_windows;
  *pinned_states = nullptr;
  if (c->packed_in) {
    // pose | speedbias | ex_pose | inv_depth are the first four blocks of the packed buffer
    const size_t bytes = pack_up(sizeof(double) * B * 77) + pack_up(sizeof(double) * B * 99) + pack_up(sizeof(double) * B * 7) +
                         pack_up(sizeof(double) * B * h->max_feat);
    char* hp = static_cast<char*>(pinned_get(c, "w_states", bytes));
    if (!hp) return fail(c, AVM_ERR_HIP, "allocation failed (packed states)");
    HIPCHK(c, hipMemcpyAsync(hp, d->pose, bytes, hipMemcpyDeviceToHost, c->stream));
    *pinned_states = hp;
    return AVM_OK;
  }
  HIPCHK(c, hipMemcpyAsync(h->pose, d->pose, sizeof(double) * B * 77, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h->speedbias, d->speedbias, sizeof(double) * B * 99, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h->ex_pose, d->ex_pose, sizeof(double) * B * 7, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h->inv_depth, d->inv_depth, sizeof(double) * B * h->max_feat, hipMemcpyDeviceToHost, c->stream));
  return AVM_OK;
}

// the optional in/out members (para_Td, relo_Pose) back to the caller
int unstage_window_extras(avm_ctx* c, const avm_window_batch* h, const avm_window_batch* d) {
  const size_t B = h->n_windows;
  if (h->td && d->td) HIPCHK(c, hipMemcpyAsync(h->td, d->td, sizeof(double) * B, hipMemcpyDeviceToHost, c->stream));
  if (h->relo_pose && d->relo_pose) HIPCHK(c, hipMemcpyAsync(h->relo_pose, d->relo_pose, sizeof(double) * B * 7, hipMemcpyDeviceToHost, c->stream));
  return AVM_OK;
}

// after the synchronize: scatter the packed states into the caller's arrays
void finish_window_states(const avm_window_batch* h, const char* pinned_states) {
  if (!pinned_states) return;
  const size_t B = h->n_windows;
  size_t off = 0;
  std::memcpy(h->pose, pinned_states + off, sizeof(double) * B * 77), off += pack_up(sizeof(double) * B * 77);
  std::memcpy(h->speedbias, pinned_states + off, sizeof(double) * B * 99), off += pack_up(sizeof(double) * B * 99);
  std::memcpy(h->ex_pose, pinned_states + off, sizeof(double) * B * 7), off += pack_up(sizeof(double) * B * 7);
  std::memcpy(h->inv_depth, pinned_states + off, sizeof(double) * B * h->max_feat);
}

int run_preint(avm_ctx* c, const avm_options* opt, const avm_window_batch* d) {
  PreintArgs pa;
  pa.n_windows = d->n_windows, pa.max_samp = d->max_samp;
  pa.imu_n = d->imu_n, pa.imu_dt = d->imu_dt, pa.imu_acc = d->imu_acc, pa.imu_gyr = d->imu_gyr;
  pa.imu_lin_ba = d->imu_lin_ba, pa.imu_lin_bg = d->imu_lin_bg;
  pa.acc_n = opt->acc_n, pa.gyr_n = opt->gyr_n, pa.acc_w = opt->acc_w, pa.gyr_w = opt->gyr_w;
  pa.out_delta = c->pre_delta, pa.out_jacobian = c->pre_jac, pa.out_covariance = c->pre_cov, pa.out_sum_dt = c->pre_sum,
  pa.out_sqrt_info = c->pre_sqrt;
  launch_preint(pa, c->stream);
  HIPCHK(c, hipGetLastError());
  return AVM_OK;
}

}  // namespace

extern "C" {

const char* avm_version(void) { return "avm-mi355x 0.1 (gfx950, fp64)"; }
int avm_abi_version(void) { return AVM_ABI_VERSION; }

int avm_default_options(avm_options* o) {
  if (!o) return AVM_ERR_INVALID;
  std::memset(o, 0, sizeof *o);
  o->max_num_iterations = 8;  // config/euroc/euroc_config.yaml:55
  o->estimate_extrinsic = 0;
  o->estimate_td = 0;
  o->marginalization_flag = AVM_MARGIN_OLD;
  o->focal_length = 460.0;  // parameters.h:13
  o->g[0] = 0, o->g[1] = 0, o->g[2] = 9.81007;
  o->acc_n = 0.08, o->gyr_n = 0.004, o->acc_w = 0.00004, o->gyr_w = 2.0e-6;
  o->cauchy_a = 1.0;
  o->max_sum_dt = 10.0;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
  o->marg_eps = 1e-8;
  o->tr = 0.0, o->row = 480.0;  // global shutter (config/euroc/euroc_config.yaml:66), image_height
  o->max_solver_time_s = 0.0;   // no wall-clock cap (the host sets SOLVER_TIME, estimator.cpp:803-806; avm_host.hpp does)
  o->marg_noise_rel = 1e-18;    // the eigenvalue clamp also tests against the rounding noise of the eigenvector's variables (0: literal).  Round 5: 1e-18
                                // (was 1e-16, which dropped GENUINE weak directions on 11 of 160 stream frames: profiles/r05_noise_rel.md)
  return AVM_OK;
}

int avm_create(const avm_config* cfg, avm_ctx** out) {
  if (!out) return AVM_ERR_INVALID;
  *out = nullptr;
  if (cfg && cfg->abi_version != AVM_ABI_VERSION) return AVM_ERR_INVALID;  // a caller compiled against another avm.h: before anything reads its structs
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return AVM_ERR_NO_DEVICE;
  const int dev = cfg ? cfg->device : 0;
  if (dev < 0 || dev >= ndev) return AVM_ERR_INVALID;
  if (hipSetDevice(dev) != hipSuccess) return AVM_ERR_HIP;
  avm_ctx* c = new avm_ctx();
  c->device = dev;
  // A BLOCKING stream (not hipStreamNonBlocking): device-resident buffers are usually produced on the legacy default
  // stream (PyTorch's current stream, plain hipMemcpy), and a blocking stream is ordered after that work and before
  // whatever the default stream does next - AVM_MEM_DEVICE calls need no extra synchronization from such callers.
  // Producers on other streams order themselves against avm_ctx_stream() (see avm.h).
  if (hipStreamCreateWithFlags(&c->stream, hipStreamDefault) != hipSuccess) {
    delete c;
    return AVM_ERR_HIP;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    delete c;
    return AVM_ERR_HIP;
  }
  // one resident 512-thread workgroup per CU (the solve kernel takes ~158 KiB of the 160 KiB LDS)
  c->n_slots = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) c->wall_clock_hz = 1.0e3 * khz;
  }
  for (auto& e : c->ev) (void)hipEventCreate(&e);
  (void)hipEventCreateWithFlags(&c->ev_flag, hipEventDisableTiming);
  if (const char* pe = getenv("AVM_PROFILE"))
    if (pe[0] == '1') (void)hipMalloc(&c->prof, sizeof(long long) * PROF_SLOTS * 2 * c->n_slots);  // (two workgroups per CU in the throughput form)
  *out = c;
  return AVM_OK;
}

void avm_destroy(avm_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)avm_comm_destroy(c);
  for (auto& kv : c->pool)
    if (kv.second.first) (void)hipFree(kv.second.first);
  for (auto& kv : c->pinned)
    if (kv.second.first) (void)hipHostFree(kv.second.first);
  for (void* p : {(void*)c->scratch, (void*)c->iscratch, (void*)c->pre_delta, (void*)c->pre_jac, (void*)c->pre_cov, (void*)c->pre_sqrt,
                  (void*)c->pre_sum, (void*)c->d_summary})
    if (p) (void)hipFree(p);
  for (auto& e : c->ev) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* avm_last_error(const avm_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

// ---- multi-GPU: raw RCCL (rccl.h) all-gather of the final states, one communicator per ctx ---------------------------------
#define RCCLCHK(ctx, call)                                                                                  \
  do {                                                                                                      \
    ncclResult_t r__ = (call);                                                                              \
    if (r__ != ncclSuccess) {                                                                               \
      (ctx)->err = std::string(#call) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r__) : "rccl error");  \
      return AVM_ERR_HIP;                                                                                   \
    }                                                                                                       \
  } while (0)

int avm_comm_unique_id(avm_ctx* c, void* id) {
  if (!c || !id) return AVM_ERR_INVALID;
  if (!rccl().load()) return fail(c, AVM_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded (dlopen)");
  static_assert(sizeof(ncclUniqueId) == AVM_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclUniqueId u;
  RCCLCHK(c, rccl().GetUniqueId(&u));
  std::memcpy(id, &u, sizeof u);
  return AVM_OK;
}

int avm_comm_init(avm_ctx* c, int32_t n_ranks, int32_t rank, const void* id) {
  if (!c || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return c ? fail(c, AVM_ERR_INVALID, "bad rank / n_ranks / id") : AVM_ERR_INVALID;
  if (c->comm) return fail(c, AVM_ERR_INVALID, "this ctx already has a communicator (avm_comm_destroy first)");
  if (!rccl().load()) return fail(c, AVM_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded (dlopen)");
  (void)hipSetDevice(c->device);
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  RCCLCHK(c, rccl().CommInitRank(&c->comm, n_ranks, u, rank));
  c->comm_ranks = n_ranks, c->comm_rank = rank;
  return AVM_OK;
}

int avm_gather_states(avm_ctx* c, const double* send, double* recv, size_t count) {
  if (!c || !send || !recv) return c ? fail(c, AVM_ERR_INVALID, "null buffer") : AVM_ERR_INVALID;
  if (!c->comm) return fail(c, AVM_ERR_INVALID, "avm_comm_init has not been called on this ctx");
  (void)hipSetDevice(c->device);
  if (count == 0) return AVM_OK;
  HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
  RCCLCHK(c, rccl().AllGather(send, recv, count, ncclDouble, c->comm, c->stream));
  HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev[3], c->ev[4]) == hipSuccess) c->last_ms["gather_states"] = ms;
  return AVM_OK;
}

int avm_comm_destroy(avm_ctx* c) {
  if (!c) return AVM_ERR_INVALID;
  if (c->comm) {
    (void)hipSetDevice(c->device);
    (void)rccl().CommDestroy(c->comm);
    c->comm = nullptr, c->comm_ranks = 0;
  }
  return AVM_OK;
}

int avm_ctx_stream(const avm_ctx* c, void** stream) {
  if (!c || !stream) return AVM_ERR_INVALID;
  *stream = reinterpret_cast<void*>(c->stream);
  return AVM_OK;
}

int avm_last_kernel_ms(const avm_ctx* c, const char* which, float* ms) {
  if (!c || !which || !ms) return AVM_ERR_INVALID;
  auto it = c->last_ms.find(which);
  if (it == c->last_ms.end()) return AVM_ERR_INVALID;
  *ms = it->second;
  return AVM_OK;
}

int avm_window_solve_batch(avm_ctx* c, const avm_options* opt, avm_mem mem, const avm_window_batch* batch, avm_prior_out* prior_out,
                           avm_solve_summary* summary) {
  if (!c) return AVM_ERR_INVALID;
  (void)hipSetDevice(c->device);
  int rc = check_window_batch(c, opt, batch);
  if (rc != AVM_OK) return rc;
  const bool marg = opt->marginalization_flag != AVM_MARGIN_NONE;
  if (marg) {
    if (!prior_out || !prior_out->n || !prior_out->nblk || !prior_out->blk_kind || !prior_out->blk_frame || !prior_out->J || !prior_out->r ||
        !prior_out->x0)
      return fail(c, AVM_ERR_INVALID, "prior_out (or one of its arrays) is NULL but marginalization_flag != AVM_MARGIN_NONE");
    if (prior_out->max_prior > MAXPRIOR || prior_out->max_prior < 1 || prior_out->max_pblk < 1) return fail(c, AVM_ERR_CAPACITY, "prior_out dims");
  }
  if (batch->n_windows == 0) return AVM_OK;
  // table check: host tables on the host, now; device-resident ones by a kernel whose verdict is read while the pre-integration runs
  int tp_misfit = 0;
  int* vflag_host = nullptr;
  if (mem == AVM_MEM_HOST) {
    if ((rc = validate_windows(c, mem, batch, CHK_TRACKS | CHK_IMU | CHK_PRIOR, &tp_misfit)) != AVM_OK) return rc;
  } else {
    if ((rc = validate_windows_begin(c, batch, CHK_TRACKS | CHK_IMU | CHK_PRIOR, &vflag_host)) != AVM_OK) return rc;
  }
  // Which form of the solve kernel: the throughput form (two 256-thread workgroups per CU, window_solve_tp.o) for batches that give
  // every CU more than one window, the latency form (one 512-thread workgroup per CU) otherwise - and always for the extended
  // problem or a prior the structural form cannot hold.  (Round 5: a wall-clock cap no longer forces the latency form - both kernels
  // check options.max_solver_time_in_seconds against a clock that starts with the window's own solve, and a batch that is larger than
  // the CU count is not a real-time call.  A window shares its CU there and runs 1.6 ms instead of 0.9: a cap between those two
  // durations ends it an iteration earlier than the latency form would - wall-clock semantics.)
  // AVM_SOLVE_TP=0 / 1 forces the choice where both are possible (tests, A/B runs).
  const bool extended = opt->estimate_extrinsic != 0 || opt->estimate_td != 0 || batch->relo_n != nullptr;
  bool use_tp = batch->n_windows > c->n_slots;
  if (const char* e = getenv("AVM_SOLVE_TP")) use_tp = e[0] == '1' ? true : (e[0] == '0' ? false : use_tp);
  use_tp = use_tp && !extended;
  // (the marginalization is the same problem whatever the solve estimated: a large batch of the extended problem takes its throughput form too)
  const bool big_x = extended && marg && batch->n_windows > c->n_slots;
  // (the slots for the throughput forms are sized before the priors' verdict is in: a batch that then takes the latency forms uses half of them)
  if ((rc = ensure_window_buffers(c, batch->n_windows, use_tp || big_x)) != AVM_OK) {
    if (vflag_host) (void)hipStreamSynchronize(c->stream);
    return rc;
  }
  avm_window_batch d;
  avm_solve_summary* d_sum = nullptr;
  if (mem == AVM_MEM_HOST) {
    if ((rc = stage_window_batch(c, batch, &d)) != AVM_OK) return rc;
    if (summary) {
      d_sum = static_cast<avm_solve_summary*>(pool_get(c, "w_summary", sizeof(avm_solve_summary) * batch->n_windows));
      if (!d_sum) return fail(c, AVM_ERR_HIP, "hipMalloc failed (summary)");
    }
  } else {
    d = *batch;
    d_sum = summary;
  }
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  if ((rc = run_preint(c, opt, &d)) != AVM_OK) {
    if (vflag_host) (void)hipStreamSynchronize(c->stream);  // (the table check of a device-resident batch is still in flight: drain it before the caller may free its tables)
    return rc;
  }
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  if (vflag_host && (rc = validate_windows_end(c, vflag_host, &tp_misfit)) != AVM_OK) return rc;
  use_tp = use_tp && (tp_misfit & 1) == 0;
  // ... and the marginalization follows the solve: its throughput form (two 256-thread workgroups per CU on the solve's 2 x CUs slots)
  // for the batches that took the throughput solve - and for a batch of the extended problem that is larger than the CU count: the marginalization
  // does not depend on what the solve estimated -, unless a prior keeps a speed-bias block beyond frame 1 (AVM_MARG_TP=0: never)
  bool use_marg_tp = (use_tp || big_x) && (tp_misfit & 2) == 0;
  if (const char* e = getenv("AVM_MARG_TP")) use_marg_tp = use_marg_tp && e[0] != '0';
  SolveArgs sa;
  sa.b = d, sa.opt = *opt;
  sa.pre_delta = c->pre_delta, sa.pre_jac = c->pre_jac, sa.pre_sqrt = c->pre_sqrt, sa.pre_sum_dt = c->pre_sum;
  sa.scratch = c->scratch, sa.iscratch = c->iscratch, sa.summary = d_sum, sa.n_slots = c->n_slots;
  sa.prof = c->prof;
  // (a cap that is not finite, or beyond 1e9 s, means "no cap": the conversion to device ticks must not overflow)
  sa.time_cap_ticks = (opt->max_solver_time_s > 0.0 && opt->max_solver_time_s <= 1.0e9) ? (long long)(opt->max_solver_time_s * c->wall_clock_hz) + 1 : 0;
  {
    const char* ns = getenv("AVM_NO_SPECULATE");
    sa.speculate = (ns && ns[0] == '1') ? 0 : 1;
  }
  if (c->prof) HIPCHK(c, hipMemsetAsync(c->prof, 0, sizeof(long long) * PROF_SLOTS * 2 * c->n_slots, c->stream));
  // ex_pose / td as variables, relocalization factors: the build of the solve kernel with the wider dense block
  if (use_tp) {
    sa.n_slots = 2 * c->n_slots;
    if (const char* e = getenv("AVM_TP_GRID")) sa.n_slots = std::max(1, std::min(atoi(e), 2 * c->n_slots));  // (experiments: fewer resident workgroups)
    HIPCHK(c, launch_window_solve_tp(sa, c->stream));
    sa.n_slots = use_marg_tp ? 2 * c->n_slots : c->n_slots;  // (the marginalization below: two workgroups per CU in its throughput form, else one)
  } else {
    HIPCHK(c, extended ? launch_window_solve_x(sa, c->stream) : launch_window_solve(sa, c->stream));
    sa.n_slots = use_marg_tp ? 2 * c->n_slots : c->n_slots;
  }
  c->last_solve_tp = use_tp;
  c->last_marg_tp = false;
  HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  avm_prior_out dpo;
  int* marg_err = nullptr;
  int marg_err_host = 0x7f7f7f7f;
  char* po_pinned = nullptr;
  size_t po_offsets[7] = {0, 0, 0, 0, 0, 0, 0};
  if (marg) {
    const size_t B = batch->n_windows, mp = prior_out->max_prior, mb = prior_out->max_pblk;
    dpo = *prior_out;
    char* po_dev = nullptr;
    size_t po_bytes = 0, po_off[7] = {0, 0, 0, 0, 0, 0, 0};
    if (mem == AVM_MEM_HOST) {
      // one device block n | nblk | blk_kind | blk_frame | J | r | x0: one memset, and (small batches) one copy back
      const size_t sz[7] = {sizeof(int32_t) * B, sizeof(int32_t) * B, sizeof(int32_t) * B * mb, sizeof(int32_t) * B * mb,
                            sizeof(double) * B * mp * mp, sizeof(double) * B * mp, sizeof(double) * B * mb * 9};
      for (int k = 0; k < 7; k++) po_off[k] = po_bytes, po_bytes += pack_up(sz[k]);
      po_dev = static_cast<char*>(pool_get(c, "po_pack", po_bytes));
      if (!po_dev) return fail(c, AVM_ERR_HIP, "hipMalloc failed (prior out)");
      dpo.n = reinterpret_cast<int32_t*>(po_dev + po_off[0]), dpo.nblk = reinterpret_cast<int32_t*>(po_dev + po_off[1]);
      dpo.blk_kind = reinterpret_cast<int32_t*>(po_dev + po_off[2]), dpo.blk_frame = reinterpret_cast<int32_t*>(po_dev + po_off[3]);
      dpo.J = reinterpret_cast<double*>(po_dev + po_off[4]), dpo.r = reinterpret_cast<double*>(po_dev + po_off[5]);
      dpo.x0 = reinterpret_cast<double*>(po_dev + po_off[6]);
      HIPCHK(c, hipMemsetAsync(po_dev, 0, po_bytes, c->stream));
    }
    // raised by the kernel when a window's kept set does not fit (prior_out->max_prior / max_pblk, or the 76 rows /
    // 16 blocks the eigen-solver holds): the call then fails with AVM_ERR_CAPACITY instead of returning a truncated prior
    marg_err = static_cast<int*>(pool_get(c, "marg_err", sizeof(int)));
    if (!marg_err) return fail(c, AVM_ERR_HIP, "hipMalloc failed (marginalization flag)");
    HIPCHK(c, hipMemsetAsync(marg_err, 0x7f, sizeof(int), c->stream));
    // the magnitude every diagonal entry of A' was formed at (marginalize_kernel -> the eigenvalue clamp's noise test): the ctx's own array
    double* marg_scale = static_cast<double*>(pool_get(c, "marg_scale", sizeof(double) * B * mp));
    if (!marg_scale) return fail(c, AVM_ERR_HIP, "hipMalloc failed (marginalization scales)");
    HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
    HIPCHK(c, use_marg_tp ? launch_marginalize_tp(sa, dpo, marg_err, marg_scale, c->stream) : launch_marginalize(sa, dpo, marg_err, marg_scale, c->stream));
    c->last_marg_tp = use_marg_tp;
    HIPCHK(c, hipEventRecord(c->ev[7], c->stream));
    int* pe_done = static_cast<int*>(pool_get(c, "pe_done", sizeof(int) * B));
    if (!pe_done) return fail(c, AVM_ERR_HIP, "hipMalloc failed (prior flags)");
    const double noise_rel = (opt->marg_noise_rel > 0.0 && opt->marg_noise_rel < 1.0) ? opt->marg_noise_rel : 0.0;
    HIPCHK(c, launch_prior_eig(dpo, B, opt->marg_eps, noise_rel, marg_scale, c->prof, pe_done, c->stream));
    c->last_marg_windows = (int)B;
    HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
    if (mem == AVM_MEM_HOST) {
      if (po_bytes <= PACK_LIMIT) {
        po_pinned = static_cast<char*>(pinned_get(c, "po_pack", po_bytes));
        if (!po_pinned) return fail(c, AVM_ERR_HIP, "allocation failed (packed prior out)");
        HIPCHK(c, hipMemcpyAsync(po_pinned, po_dev, po_bytes, hipMemcpyDeviceToHost, c->stream));
        for (int k = 0; k < 7; k++) po_offsets[k] = po_off[k];
      } else {
        HIPCHK(c, hipMemcpyAsync(prior_out->n, dpo.n, sizeof(int32_t) * B, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(prior_out->nblk, dpo.nblk, sizeof(int32_t) * B, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(prior_out->blk_kind, dpo.blk_kind, sizeof(int32_t) * B * mb, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(prior_out->blk_frame, dpo.blk_frame, sizeof(int32_t) * B * mb, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(prior_out->J, dpo.J, sizeof(double) * B * mp * mp, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(prior_out->r, dpo.r, sizeof(double) * B * mp, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(prior_out->x0, dpo.x0, sizeof(double) * B * mb * 9, hipMemcpyDeviceToHost, c->stream));
      }
    }
  }
  char* pinned_states = nullptr;
  if (mem == AVM_MEM_HOST) {
    const size_t B = batch->n_windows;
    if ((rc = unstage_window_states(c, batch, &d, &pinned_states)) != AVM_OK) return rc;
    if ((rc = unstage_window_extras(c, batch, &d)) != AVM_OK) return rc;
    if (summary) HIPCHK(c, hipMemcpyAsync(summary, d_sum, sizeof(avm_solve_summary) * B, hipMemcpyDeviceToHost, c->stream));
  }
  if (marg_err) HIPCHK(c, hipMemcpyAsync(&marg_err_host, marg_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (mem == AVM_MEM_HOST) finish_window_states(batch, pinned_states);  // (also before a capacity error: the states WERE solved)
  if (marg_err_host != 0x7f7f7f7f) {
    c->err = "window " + std::to_string(marg_err_host) +
             ": the new prior does not fit (prior_out->max_prior / max_pblk too small, or more than 76 rows / 16 blocks to keep); "
             "the states were solved, prior_out is not valid";
    return AVM_ERR_CAPACITY;
  }
  if (po_pinned) {
    const size_t B = batch->n_windows, mp = prior_out->max_prior, mb = prior_out->max_pblk;
    std::memcpy(prior_out->n, po_pinned + po_offsets[0], sizeof(int32_t) * B);
    std::memcpy(prior_out->nblk, po_pinned + po_offsets[1], sizeof(int32_t) * B);
    std::memcpy(prior_out->blk_kind, po_pinned + po_offsets[2], sizeof(int32_t) * B * mb);
    std::memcpy(prior_out->blk_frame, po_pinned + po_offsets[3], sizeof(int32_t) * B * mb);
    std::memcpy(prior_out->J, po_pinned + po_offsets[4], sizeof(double) * B * mp * mp);
    std::memcpy(prior_out->r, po_pinned + po_offsets[5], sizeof(double) * B * mp);
    std::memcpy(prior_out->x0, po_pinned + po_offsets[6], sizeof(double) * B * mb * 9);
  }
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->last_ms["preint"] = ms;
  if (hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) c->last_ms["window_solve"] = ms;
  c->last_ms["marginalize"] = 0.f, c->last_ms["prior_eig"] = 0.f;
  if (marg && hipEventElapsedTime(&ms, c->ev[6], c->ev[7]) == hipSuccess) c->last_ms["marginalize"] = ms;
  if (marg && hipEventElapsedTime(&ms, c->ev[7], c->ev[5]) == hipSuccess) c->last_ms["prior_eig"] = ms;
  return AVM_OK;
}

int avm_window_solve(avm_ctx* c, const avm_options* opt, avm_mem mem, const avm_window_batch* window, avm_prior_out* prior_out,
                     avm_solve_summary* summary) {
  if (!c) return AVM_ERR_INVALID;
  if (!window || window->n_windows != 1) return fail(c, AVM_ERR_INVALID, "avm_window_solve takes exactly one window (n_windows == 1)");
  return avm_window_solve_batch(c, opt, mem, window, prior_out, summary);
}

int avm_imu_preintegrate_batch(avm_ctx* c, const avm_options* opt, avm_mem mem, const avm_window_batch* batch, double* out_delta,
                               double* out_jacobian, double* out_covariance, double* out_sum_dt) {
  if (!c) return AVM_ERR_INVALID;
  (void)hipSetDevice(c->device);
  int rc = check_window_batch(c, opt, batch);
  if (rc != AVM_OK) return rc;
  if (batch->n_windows == 0) return AVM_OK;
  if ((rc = validate_windows(c, mem, batch, CHK_IMU)) != AVM_OK) return rc;
  if ((rc = ensure_window_buffers(c, batch->n_windows)) != AVM_OK) return rc;
  avm_window_batch d;
  if (mem == AVM_MEM_HOST) {
    if ((rc = stage_window_batch(c, batch, &d)) != AVM_OK) return rc;
  } else {
    d = *batch;
  }
  if ((rc = run_preint(c, opt, &d)) != AVM_OK) return rc;
  const size_t iv = (size_t)batch->n_windows * 10;
  const hipMemcpyKind kind = mem == AVM_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if (out_delta) HIPCHK(c, hipMemcpyAsync(out_delta, c->pre_delta, sizeof(double) * iv * 10, kind, c->stream));
  if (out_jacobian) HIPCHK(c, hipMemcpyAsync(out_jacobian, c->pre_jac, sizeof(double) * iv * 225, kind, c->stream));
  if (out_covariance) HIPCHK(c, hipMemcpyAsync(out_covariance, c->pre_cov, sizeof(double) * iv * 225, kind, c->stream));
  if (out_sum_dt) HIPCHK(c, hipMemcpyAsync(out_sum_dt, c->pre_sum, sizeof(double) * iv, kind, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return AVM_OK;
}

// debug hook (not in avm.h): per-phase shader clocks of the last solve, summed over slots, [PROF_SLOTS = 64]
int avm_debug_copy_profile(avm_ctx* c, long long* host_out) {
  if (!c || !c->prof) return AVM_ERR_INVALID;
  std::vector<long long> h((size_t)PROF_SLOTS * 2 * c->n_slots);
  HIPCHK(c, hipMemcpy(h.data(), c->prof, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
  for (int k = 0; k < PROF_SLOTS; k++) host_out[k] = 0;
  for (int s = 0; s < 2 * c->n_slots; s++)
    for (int k = 0; k < PROF_SLOTS; k++) host_out[k] += h[(size_t)s * PROF_SLOTS + k];
  return AVM_OK;
}

// test / bench hook (not in avm.h): 1 when the last avm_window_solve_batch ran the throughput form of the solve kernel
int avm_debug_last_solve_form(const avm_ctx* c) { return c ? (c->last_solve_tp ? 1 : 0) : -1; }
// ... 1 when its marginalization ran the throughput form (marginalize_tp_kernel: two 256-thread workgroups per CU)
int avm_debug_last_marg_form(const avm_ctx* c) { return c ? (c->last_marg_tp ? 1 : 0) : -1; }
// ... and which form the last avm_fsel_select_batch STARTED in: 3 = one workgroup per frame with lazy evaluation (fsel_solo_kernel), 2 / 1 = the
// frame kernel's teams, 0 = one launch per greedy round
int avm_debug_last_fsel_form(const avm_ctx* c) { return c ? c->last_fsel_mode : -1; }

// test / bench hook (not in avm.h): out[0] = workgroups of the throughput kernel per CU (runtime's occupancy query), out[1] = its LDS bytes
int avm_debug_solve_tp_occupancy(int* out) {
  out[0] = window_solve_tp_occupancy(), out[1] = window_solve_tp_lds_bytes();
  return 2;
}

// test hook (not in avm.h): the compile-time tables of the throughput solve's sparse factorization (window_solve.hip, chol_regs); out: >= 512 ints
int avm_debug_solve_tp_pattern(int* out) { return window_solve_tp_pattern(out); }
// which = 0: throughput build, 1: latency build, 2: extended build
int avm_debug_solve_pattern(int which, int* out) {
  return which == 0 ? window_solve_tp_pattern(out) : (which == 1 ? window_solve_pattern(out) : window_solve_x_pattern(out));
}

// test / bench hook (not in avm.h): out[0] = device / pinned (re)allocations of this ctx so far, out[1] = windows of the last
// marginalization whose square root the one-wavefront kernel (prior_chol_kernel) finished, out[2] = windows of that marginalization
// (out[2] - out[1] went through prior_eig_kernel's pivoted path), out[3] = 1 if the last solve took the throughput form
int avm_debug_counters(avm_ctx* c, int64_t* out) {
  if (!c || !out) return AVM_ERR_INVALID;
  out[0] = c->n_allocs, out[1] = 0, out[2] = c->last_marg_windows, out[3] = c->last_solve_tp ? 1 : 0;
  auto it = c->pool.find("pe_done");
  if (c->last_marg_windows > 0 && it != c->pool.end() && it->second.first) {
    std::vector<int> h((size_t)c->last_marg_windows);
    HIPCHK(c, hipMemcpy(h.data(), it->second.first, sizeof(int) * h.size(), hipMemcpyDeviceToHost));
    for (int v : h) out[1] += v != 0;
  }
  return AVM_OK;
}

// bench hook (not in avm.h): candidate evaluations the last avm_fsel_select_batch executed on the device - counted by fsel_solo_kernel
// (the lazy form scores a fraction of the live candidates per round); -1 for the forms that score every live candidate every round
int avm_debug_fsel_evaluations(avm_ctx* c, int64_t* out) {
  if (!c || !out) return AVM_ERR_INVALID;
  *out = c->last_fsel_evals;
  return AVM_OK;
}

// test hook (not in avm.h): sizeof of every ABI struct, for the ctypes mirror check
int avm_debug_struct_sizes(int* out) {
  out[0] = (int)sizeof(avm_options), out[1] = (int)sizeof(avm_window_batch), out[2] = (int)sizeof(avm_prior_out);
  out[3] = (int)sizeof(avm_solve_summary), out[4] = (int)sizeof(avm_fsel_batch), out[5] = (int)sizeof(avm_fsel_out);
  out[6] = (int)sizeof(avm_config);
  return 7;
}

// test hook (not in avm.h): sqrt_info of the last pre-integration, [B][10][15][15]
int avm_debug_copy_sqrt_info(avm_ctx* c, int n_windows, double* host_out) {
  if (!c || !c->pre_sqrt) return AVM_ERR_INVALID;
  HIPCHK(c, hipMemcpy(host_out, c->pre_sqrt, sizeof(double) * (size_t)n_windows * 2250, hipMemcpyDeviceToHost));
  return AVM_OK;
}

int avm_triangulate_batch(avm_ctx* c, avm_mem mem, avm_window_batch* batch, double init_depth) {
  if (!c) return AVM_ERR_INVALID;
  (void)hipSetDevice(c->device);
  if (!batch || batch->n_windows < 0) return fail(c, AVM_ERR_INVALID, "null/negative argument");
  if (batch->max_feat > MAXE) return fail(c, AVM_ERR_CAPACITY, "max_feat > 150");
  if (batch->max_obs > MAXOBS) return fail(c, AVM_ERR_CAPACITY, "max_obs > 1650");
  if (batch->n_windows == 0) return AVM_OK;
  {
    const int vrc = validate_windows(c, mem, batch, CHK_TRACKS);
    if (vrc != AVM_OK) return vrc;
  }
  const size_t B = batch->n_windows;
  avm_window_batch d = *batch;
  if (mem == AVM_MEM_HOST) {
    // only the fields the triangulation reads travel
    int rc;
#define ST(field, type, count)                                                                                        \
  if ((rc = stage_in<type>(c, "w_" #field, batch->field, (count), (const type**)&d.field)) != AVM_OK) return rc;
    ST(pose, double, B * 77)
    ST(ex_pose, double, B * 7)
    ST(inv_depth, double, B * batch->max_feat)
    ST(n_feat, int32_t, B)
    ST(feat_start, int32_t, B * batch->max_feat)
    ST(feat_nobs, int32_t, B * batch->max_feat)
    ST(feat_obs_begin, int32_t, B * batch->max_feat)
    ST(obs_xy, double, B * batch->max_obs * 2)
#undef ST
  }
  HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
  HIPCHK(c, launch_triangulate(d, init_depth, c->stream));
  HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
  if (mem == AVM_MEM_HOST)
    HIPCHK(c, hipMemcpyAsync(batch->inv_depth, d.inv_depth, sizeof(double) * B * batch->max_feat, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev[3], c->ev[4]) == hipSuccess) c->last_ms["triangulate"] = ms;
  return AVM_OK;
}

int avm_slide_window(avm_ctx* c, avm_mem mem, avm_window_batch* batch, int32_t flag, int32_t shift_depth, double init_depth) {
  if (!c) return AVM_ERR_INVALID;
  (void)hipSetDevice(c->device);
  if (!batch || batch->n_windows < 0) return fail(c, AVM_ERR_INVALID, "null/negative argument");
  if (flag != AVM_MARGIN_OLD && flag != AVM_MARGIN_SECOND_NEW) return fail(c, AVM_ERR_INVALID, "marginalization_flag must be MARGIN_OLD or MARGIN_SECOND_NEW");
  if (batch->n_windows == 0) return AVM_OK;
  {
    const int vrc = validate_windows(c, mem, batch, CHK_TRACKS | CHK_IMU);
    if (vrc != AVM_OK) return vrc;
  }
  const size_t B = batch->n_windows;
  avm_window_batch d = *batch;
  // (field, element type, elements) of everything the roll reads or rewrites
#define AVM_SLIDE_FIELDS(X)                                                                                              \
  X(pose, double, B * 77) X(speedbias, double, B * 99) X(ex_pose, double, B * 7) X(inv_depth, double, B * batch->max_feat)    \
  X(n_feat, int32_t, B) X(feat_start, int32_t, B * batch->max_feat) X(feat_nobs, int32_t, B * batch->max_feat)                \
  X(feat_obs_begin, int32_t, B * batch->max_feat) X(obs_xy, double, B * batch->max_obs * 2) X(imu_n, int32_t, B * 10)         \
  X(imu_dt, double, B * 10 * batch->max_samp) X(imu_acc, double, B * 10 * (batch->max_samp + 1) * 3)                          \
  X(imu_gyr, double, B * 10 * (batch->max_samp + 1) * 3) X(imu_lin_ba, double, B * 30) X(imu_lin_bg, double, B * 30)
  if (mem == AVM_MEM_HOST) {
    int rc;
#define ST(field, type, count) \
  if ((rc = stage_in<type>(c, "w_" #field, batch->field, (count), (const type**)&d.field)) != AVM_OK) return rc;
    AVM_SLIDE_FIELDS(ST)
#undef ST
  }
  int* derr = static_cast<int*>(pool_get(c, "slide_err", sizeof(int)));
  if (!derr) return fail(c, AVM_ERR_HIP, "hipMalloc failed (slide flag)");
  HIPCHK(c, hipMemsetAsync(derr, 0, sizeof(int), c->stream));
  HIPCHK(c, launch_slide_window(d, flag, shift_depth, init_depth, derr, c->stream));
  int herr = 0;
  HIPCHK(c, hipMemcpyAsync(&herr, derr, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  if (mem == AVM_MEM_HOST) {
#define BK(field, type, count) \
  HIPCHK(c, hipMemcpyAsync(const_cast<type*>(batch->field), d.field, sizeof(type) * (count), hipMemcpyDeviceToHost, c->stream));
    AVM_SLIDE_FIELDS(BK)
#undef BK
  }
#undef AVM_SLIDE_FIELDS
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (herr) return fail(c, AVM_ERR_CAPACITY, "MARGIN_SECOND_NEW: interval 8 + interval 9 exceed max_samp samples");
  return AVM_OK;
}

int avm_imu_propagate_batch(avm_ctx* c, avm_mem mem, avm_window_batch* batch, const double g[3]) {
  if (!c) return AVM_ERR_INVALID;
  (void)hipSetDevice(c->device);
  if (!batch || !g || batch->n_windows < 0) return fail(c, AVM_ERR_INVALID, "null/negative argument");
  if (batch->n_windows == 0) return AVM_OK;
  {
    const int vrc = validate_windows(c, mem, batch, CHK_IMU);
    if (vrc != AVM_OK) return vrc;
  }
  const size_t B = batch->n_windows;
  avm_window_batch d = *batch;
  if (mem == AVM_MEM_HOST) {
    int rc;
#define ST(field, type, count)                                                                                        \
  if ((rc = stage_in<type>(c, "w_" #field, batch->field, (count), (const type**)&d.field)) != AVM_OK) return rc;
    ST(pose, double, B * 77)
    ST(speedbias, double, B * 99)
    ST(imu_n, int32_t, B * 10)
    ST(imu_dt, double, B * 10 * batch->max_samp)
    ST(imu_acc, double, B * 10 * (batch->max_samp + 1) * 3)
    ST(imu_gyr, double, B * 10 * (batch->max_samp + 1) * 3)
#undef ST
  }
  HIPCHK(c, launch_imu_propagate(d, g, c->stream));
  if (mem == AVM_MEM_HOST) {
    HIPCHK(c, hipMemcpyAsync(batch->pose, d.pose, sizeof(double) * B * 77, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(batch->speedbias, d.speedbias, sizeof(double) * B * 99, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return AVM_OK;
}

int avm_projection_td_eval(avm_ctx* c, avm_mem mem, const avm_td_factor_batch* f, double* residual, double* jac) {
  if (!c) return AVM_ERR_INVALID;
  (void)hipSetDevice(c->device);
  if (!f || !residual || f->n < 0) return fail(c, AVM_ERR_INVALID, "null/negative argument");
  if (f->n == 0) return AVM_OK;
  const size_t n = f->n;
  avm_td_factor_batch d = *f;
  double *dr = residual, *dj = jac;
  if (mem == AVM_MEM_HOST) {
    int rc;
#define ST(field, count)                                                                                          \
  if ((rc = stage_in<double>(c, "td_" #field, f->field, (count), (const double**)&d.field)) != AVM_OK) return rc;
    ST(pose_i, n * 7) ST(pose_j, n * 7) ST(ex_pose, n * 7) ST(inv_depth, n) ST(td, n)
    ST(pts_i, n * 2) ST(pts_j, n * 2) ST(vel_i, n * 2) ST(vel_j, n * 2)
    ST(td_i, n) ST(td_j, n) ST(row_i, n) ST(row_j, n)
#undef ST
    dr = static_cast<double*>(pool_get(c, "td_res", sizeof(double) * n * 2));
    dj = jac ? static_cast<double*>(pool_get(c, "td_jac", sizeof(double) * n * 40)) : nullptr;
    if (!dr || (jac && !dj)) return fail(c, AVM_ERR_HIP, "hipMalloc failed (td factor out)");
  }
  HIPCHK(c, launch_projection_td_eval(d, dr, dj, c->stream));
  if (mem == AVM_MEM_HOST) {
    HIPCHK(c, hipMemcpyAsync(residual, dr, sizeof(double) * n * 2, hipMemcpyDeviceToHost, c->stream));
    if (jac) HIPCHK(c, hipMemcpyAsync(jac, dj, sizeof(double) * n * 40, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return AVM_OK;
}

int avm_window_eval_factors(avm_ctx* c, const avm_options* opt, avm_mem mem, const avm_window_batch* batch, int apply_loss,
                            double* proj_r, double* proj_J, double* imu_r, double* imu_J, double* prior_res, double* cost) {
  if (!c) return AVM_ERR_INVALID;
  (void)hipSetDevice(c->device);
  int rc = check_window_batch(c, opt, batch);
  if (rc != AVM_OK) return rc;
  if (batch->n_windows == 0) return AVM_OK;
  if ((rc = validate_windows(c, mem, batch, CHK_TRACKS | CHK_IMU | CHK_PRIOR)) != AVM_OK) return rc;
  if ((rc = ensure_window_buffers(c, batch->n_windows)) != AVM_OK) return rc;
  avm_window_batch d;
  const size_t B = batch->n_windows;
  EvalArgs ea;
  struct Out {
    double** dst;
    double* host;
    size_t n;
    const char* name;
  };
  Out outs[6] = {{&ea.proj_r, proj_r, B * batch->max_obs * 2, "e_pr"},   {&ea.proj_J, proj_J, B * batch->max_obs * 26, "e_pJ"},
                 {&ea.imu_r, imu_r, B * 150, "e_ir"},                      {&ea.imu_J, imu_J, B * 4500, "e_iJ"},
                 {&ea.prior_res, prior_res, B * batch->max_prior, "e_pres"}, {&ea.cost, cost, B, "e_cost"}};
  if (mem == AVM_MEM_HOST) {
    if ((rc = stage_window_batch(c, batch, &d)) != AVM_OK) return rc;
    for (auto& o : outs) {
      *o.dst = nullptr;
      if (o.host) {
        *o.dst = static_cast<double*>(pool_get(c, o.name, o.n * sizeof(double)));
        if (!*o.dst) return fail(c, AVM_ERR_HIP, "hipMalloc failed (eval out)");
        HIPCHK(c, hipMemsetAsync(*o.dst, 0, o.n * sizeof(double), c->stream));
      }
    }
  } else {
    d = *batch;
    for (auto& o : outs) *o.dst = o.host;
  }
  if ((rc = run_preint(c, opt, &d)) != AVM_OK) return rc;
  ea.b = d, ea.opt = *opt, ea.apply_loss = apply_loss;
  ea.pre_delta = c->pre_delta, ea.pre_jac = c->pre_jac, ea.pre_sqrt = c->pre_sqrt, ea.pre_sum_dt = c->pre_sum;
  HIPCHK(c, launch_eval_factors(ea, c->stream));
  if (mem == AVM_MEM_HOST)
    for (auto& o : outs)
      if (o.host) HIPCHK(c, hipMemcpyAsync(o.host, *o.dst, o.n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return AVM_OK;
}

namespace {

int check_fsel(avm_ctx* c, const avm_fsel_batch* b) {
  if (!b || b->n_problems < 0) return fail(c, AVM_ERR_INVALID, "null/negative argument");
  if (!fsel_horizon_supported(b->horizon)) return fail(c, AVM_ERR_UNSUPPORTED, "horizon must be one of 2,3,5,10,13");
  if (b->max_cand <= 0 || b->max_features < 0) return fail(c, AVM_ERR_INVALID, "bad max_cand / max_features");
  if (b->n_cloud && b->max_cloud > FS_MAX_CLOUD) return fail(c, AVM_ERR_UNSUPPORTED, "max_cloud above 4096 (the kd-tree of the depth cloud is built in LDS)");
  return AVM_OK;
}

#define AVM_FSEL_FIELDS(X, P, H1, h)                                                                                  \
  X(hor_pos, double, (P) * (H1) * 3) X(hor_quat, double, (P) * (H1) * 4) X(nr_imu, int32_t, (P)) X(delta_imu, double, (P))   \
  X(n_cand, int32_t, (P)) X(cand_id, int32_t, (P) * (h)->max_cand) X(cand_xy, double, (P) * (h)->max_cand * 2)            \
  X(cand_prob, double, (P) * (h)->max_cand) X(n_used, int32_t, (P)) X(used_id, int32_t, (P) * (h)->max_used)              \
  X(used_xy, double, (P) * (h)->max_used * 2) X(n_cloud, int32_t, (P)) X(cloud_xy, double, (P) * (h)->max_cloud * 2)      \
  X(cloud_depth, double, (P) * (h)->max_cloud)

// like stage_window_batch: a frame or a few travel as one pinned copy
int stage_fsel(avm_ctx* c, const avm_fsel_batch* h, avm_fsel_batch* d) {
  *d = *h;
  const size_t P = h->n_problems, H1 = h->horizon + 1;
  int rc;
  size_t total = 0;
#define SZ(field, type, count) total += h->field ? pack_up(sizeof(type) * (count)) : 0;
  AVM_FSEL_FIELDS(SZ, P, H1, h)
#undef SZ
  if (total <= PACK_LIMIT) {
    char* hp = static_cast<char*>(pinned_get(c, "f_pack", total));
    char* dp = static_cast<char*>(pool_get(c, "f_pack", total));
    if (!hp || !dp) return fail(c, AVM_ERR_HIP, "allocation failed (packed staging)");
    size_t off = 0;
#define PK(field, type, count)                                             \
  if (h->field) {                                                          \
    std::memcpy(hp + off, h->field, sizeof(type) * (count));               \
    *(const type**)&d->field = reinterpret_cast<const type*>(dp + off);    \
    off += pack_up(sizeof(type) * (count));                                \
  }
    AVM_FSEL_FIELDS(PK, P, H1, h)
#undef PK
    HIPCHK(c, hipMemcpyAsync(dp, hp, total, hipMemcpyHostToDevice, c->stream));
    return AVM_OK;
  }
#define ST(field, type, count) \
  if ((rc = stage_in<type>(c, "f_" #field, h->field, (count), (const type**)&d->field)) != AVM_OK) return rc;
  AVM_FSEL_FIELDS(ST, P, H1, h)
#undef ST
  return AVM_OK;
}

constexpr size_t AVM_FSEL_SOLO_MIN = 33;  // frames per call from which a batch takes the solo form of the selector.  A solo select takes 4.3 ms (H = 10; 7.2 ms at
                                          // H = 13) however few frames run side by side; the sixteen teams take 2.0 ms (3.7 ms) per sixteen frames: up to 32 frames
                                          // two passes of the teams are ahead, from the third pass on the solo form is (measured: profiles/r05_solo_crossover.txt)
// Does a select of this batch take the solo form (one workgroup per frame, lazy evaluation: fsel_solo_kernel)?  ONE rule for the mode
// choice in avm_fsel_select_batch and for the solo form's packed Delta copy in fsel_buffers (0.8 GB at 256 frames x 512 candidates,
// H = 13: not to be held by a ctx that never runs the form).  AVM_FSEL_SOLO=0/1 overrides the batch-size rule, AVM_FSEL_FRAME vetoes it.
bool fsel_takes_solo(const avm_fsel_batch* b) {
  const bool can = b->max_cand <= 512 && 3 * b->horizon <= 39 && b->n_problems >= 1;
  bool solo = can && (size_t)b->n_problems >= AVM_FSEL_SOLO_MIN && !getenv("AVM_FSEL_FRAME");
  if (const char* e = getenv("AVM_FSEL_SOLO")) solo = can && e[0] == '1';
  return solo;
}

int fsel_buffers(avm_ctx* c, const avm_fsel_batch* b, FselBuffers* w, bool may_solo) {
  const size_t P = b->n_problems, T = 3 * (size_t)b->horizon, mc = b->max_cand, mu = b->max_used > 0 ? b->max_used : 1;
#define GET(field, type, count)                                                             \
  w->field = static_cast<type*>(pool_get(c, "fw_" #field, sizeof(type) * (count)));          \
  if (!w->field) return fail(c, AVM_ERR_HIP, "hipMalloc failed (selector work buffer)");
  GET(C, double, 2 * P * T * T)  // (two buffers: csrc/fsel.hip, FselPar)
  GET(dpp, double, 2 * P * T)  // (two buffers: csrc/fsel.hip, FselPar)
  GET(consts, double, P * 4)
  GET(delta, double, P * mc * T * T)
  // (the solo form's packed copy - csrc/fsel.hip, FselDev::delta_pk - exists whenever avm_fsel_select_batch can choose that form: the rule is there)
  GET(delta_pk, double, may_solo ? P * mc * (T * (T + 1) / 2) : 1)
  GET(ddiag, double, (may_solo && T > 30) ? P * mc * T : 1)
  GET(delta_u, double, P * mu * T * T)
  GET(fval, double, 2 * P * mc)  // (two buffers: csrc/fsel.hip, FselPar)
  GET(ub, double, 2 * P * mc)  // (two buffers: csrc/fsel.hip, FselPar)
  GET(valid, int32_t, P * mc)
  GET(valid_u, int32_t, P * mu)
  GET(black, int32_t, P * mc)
  GET(nsel, int32_t, P)
  GET(done, int32_t, P)
  GET(live, int32_t, 2 * P * mc)
  GET(pos, int32_t, 2 * P * mc)
  GET(nlive, int32_t, 2 * P)
  GET(sync, int32_t, FS_SYNC_INTS)
  GET(kd, double, fsel_kd_doubles(*b))
#undef GET
  return AVM_OK;
}

}  // namespace

int avm_fsel_select_batch(avm_ctx* c, avm_mem mem, const avm_fsel_batch* batch, avm_fsel_out* out) {
  if (!c) return AVM_ERR_INVALID;
  (void)hipSetDevice(c->device);
  int rc = check_fsel(c, batch);
  if (rc != AVM_OK) return rc;
  if (!out || !out->n_selected || !out->selected_ids) return fail(c, AVM_ERR_INVALID, "null output");
  if (batch->n_problems == 0) return AVM_OK;
  // Host tables are checked on the host.  Device-resident ones by a kernel that runs AHEAD of the select on the same stream:
  // every kernel of the select looks at its flag before it indexes with a table, and the host reads the flag with the results
  // (no extra synchronization for the check).
  int* vflag = nullptr;
  int32_t* hflag = nullptr;
  if (mem == AVM_MEM_HOST) {
    if ((rc = validate_fsel(c, mem, batch)) != AVM_OK) return rc;
  } else {
    if (!batch->n_cand || !batch->nr_imu) return fail(c, AVM_ERR_INVALID, "null n_cand / nr_imu");
    vflag = static_cast<int*>(pool_get(c, "v_flag", sizeof(int)));
    hflag = static_cast<int32_t*>(pinned_get(c, "v_flag_h", sizeof(int32_t)));
    if (!vflag || !hflag) return fail(c, AVM_ERR_HIP, "allocation failed (validation flag)");
    HIPCHK(c, hipMemsetAsync(vflag, 0x7f, sizeof(int), c->stream));
    HIPCHK(c, launch_validate_fsel(*batch, vflag, c->stream));
  }
  avm_fsel_batch d;
  avm_fsel_out dout;
  const size_t P = batch->n_problems, mf = batch->max_features;
  if (mem == AVM_MEM_HOST) {
    if ((rc = stage_fsel(c, batch, &d)) != AVM_OK) return rc;
    dout.n_selected = static_cast<int32_t*>(pool_get(c, "fo_n", sizeof(int32_t) * P));
    dout.selected_ids = static_cast<int32_t*>(pool_get(c, "fo_ids", sizeof(int32_t) * P * (mf ? mf : 1)));
    dout.fvalues = out->fvalues ? static_cast<double*>(pool_get(c, "fo_fv", sizeof(double) * P * (mf ? mf : 1))) : nullptr;
    dout.min_gap = out->min_gap ? static_cast<double*>(pool_get(c, "fo_gap", sizeof(double) * P * (mf ? mf : 1))) : nullptr;
    if (!dout.n_selected || !dout.selected_ids) return fail(c, AVM_ERR_HIP, "hipMalloc failed (selector out)");
  } else {
    d = *batch;
    dout = *out;
  }
  FselBuffers w;
  if ((rc = fsel_buffers(c, &d, &w, fsel_takes_solo(&d))) != AVM_OK) return rc;
  // Every frame's greedy rounds in ONE launch (csrc/fsel.hip, fsel_frame_kernel): 2 = a team of workgroups per XCD, the teams
  // take frames from a queue; 1 = one team over all XCDs (a single frame only); 0 = one launch per round.  A kernel that reports
  // a timed-out wait, or that did not finish every frame, is re-run one mode down, and the ctx stays there.
  // The downgrade is NOT sticky: a transient cause (an XCD busy with another ctx's solve, so that a team does not fill within
  // its 2 ms) costs this call one re-run and the next AVM_FSEL_REPROBE_CALLS calls the slower mode; then the fast mode is
  // probed again.  avm_fsel_fallback_stats() counts both.
  constexpr int AVM_FSEL_REPROBE_CALLS = 16, AVM_FSEL_REPROBE_MAX = 4096;
  c->fsel_calls++;
  const bool probing = c->fsel_cooldown > 0 && --c->fsel_cooldown == 0;  // this call tries the fast mode again
  if (probing) c->fsel_frame_mode = 2;
  bool rerun = false;
  int mode = (d.max_cand <= 512 && mf < 4096) ? c->fsel_frame_mode : 0;  // (512: FS_FRAME_MAXC)
  if (mode == 1 && P != 1) mode = 0;  // (the one-team-over-all-XCDs form takes one frame)
  if (const char* e = getenv("AVM_FSEL_FRAME"))
    if (e[0] >= '0' && e[0] <= '2') mode = std::min(mode, e[0] - '0');
  // 3 = one workgroup per frame with lazy evaluation (fsel_solo_kernel): what a batch of many frames takes - it has no waits between
  // workgroups, so it cannot time out and is never re-run.  AVM_FSEL_SOLO=0/1 overrides the batch-size rule (tests, measurements).
  if (fsel_takes_solo(&d)) mode = 3;
  c->last_fsel_mode = mode;
  int32_t* hsync = mode ? static_cast<int32_t*>(pinned_get(c, "f_sync", sizeof(int32_t) * 64)) : nullptr;
  if (mode && !hsync) mode = 0;
  for (;;) {
    HIPCHK(c, hipMemsetAsync(dout.n_selected, 0, sizeof(int32_t) * P, c->stream));
    HIPCHK(c, hipMemsetAsync(dout.selected_ids, 0xff, sizeof(int32_t) * P * mf, c->stream));
    HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    HIPCHK(c, launch_fsel(d, w, dout, nullptr, true, mode, vflag, c->stream));
    HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
    if (mode) HIPCHK(c, hipMemcpyAsync(hsync, w.sync, sizeof(int32_t) * 64, hipMemcpyDeviceToHost, c->stream));
    if (mem == AVM_MEM_HOST) {
      HIPCHK(c, hipMemcpyAsync(out->n_selected, dout.n_selected, sizeof(int32_t) * P, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(out->selected_ids, dout.selected_ids, sizeof(int32_t) * P * mf, hipMemcpyDeviceToHost, c->stream));
      if (out->fvalues) HIPCHK(c, hipMemcpyAsync(out->fvalues, dout.fvalues, sizeof(double) * P * mf, hipMemcpyDeviceToHost, c->stream));
      if (out->min_gap) HIPCHK(c, hipMemcpyAsync(out->min_gap, dout.min_gap, sizeof(double) * P * mf, hipMemcpyDeviceToHost, c->stream));
    }
    if (vflag) HIPCHK(c, hipMemcpyAsync(hflag, vflag, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (the one synchronization of the call)
    if (vflag && *hflag != 0x7f7f7f7f) return report_bad(c, *hflag, "frame");  // (no kernel of the select has touched a table)
    if (!mode) break;
    if (getenv("AVM_FSEL_TRACE")) {  // (cycle counters of a -DFS_TRACE_EVAL build of fsel.hip; zeros otherwise)
      const long long* q = reinterpret_cast<const long long*>(hsync + 32);
      fprintf(stderr, "fsel frame kernel, mode %d (cycles): pick %lld update %lld eval %lld wait %lld | eval: loads %lld bound %lld elimination %lld logdet %lld | setup: before the elimination %lld, elimination %lld\n",
              mode, q[0], q[1], q[2], q[4], q[5], q[6], q[7], q[8], q[9], q[3]);
    }
    if (mode == 3 && getenv("AVM_FSEL_LAZY_STATS")) {  // (development: frame 0's workgroup of fsel_solo_kernel)
      const long long* q = reinterpret_cast<const long long*>(hsync + 32);
      fprintf(stderr, "fsel solo kernel, frame 0 (cycles): bounds %lld list %lld scores %lld pick+check %lld second-pass scores %lld fold %lld | %lld candidates scored in %lld rounds, %lld second passes; %lld passes of the pick, %lld with exact bounds for unscored candidates | pick: maxima %lld flags %lld hits %lld check %lld\n",
              q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[8], q[7], q[10], q[9], q[11], q[12], q[13], q[14]);
    }
    c->last_fsel_evals = mode == 3 ? *reinterpret_cast<const int64_t*>(hsync + 16) : -1;
    if (hsync[2] == 0 && hsync[4] == (int32_t)P) {
      // a fast-mode call that went through: the back-off starts from the beginning next time
      if (mode == 2 || (mode == c->fsel_frame_mode && !rerun)) c->fsel_backoff = AVM_FSEL_REPROBE_CALLS;
      break;
    }
    // (the outputs of the failed attempt are overwritten by the next one)
    c->fsel_failed_launches++;  // one per launch that did not finish; the call counts once, below
    rerun = true;
    mode = mode == 3 ? std::min(2, c->fsel_frame_mode) : mode - 1;  // (3 cannot fail; kept for completeness)
    c->fsel_frame_mode = std::min(c->fsel_frame_mode, std::max(mode, P != 1 ? 1 : 0));  // a failed batch leaves mode 1 to single frames
    // exponential back-off of the re-probe: a host where the fast mode can never become resident pays a failed launch (up to its
    // 20 ms spin time-out) after 16, 32, 64 ... 4096 calls instead of every 16
    c->fsel_cooldown = c->fsel_backoff;
    c->fsel_backoff = std::min(2 * c->fsel_backoff, AVM_FSEL_REPROBE_MAX);
    if (mode == 1 && P != 1) mode = 0;
  }
  if (rerun) c->fsel_reruns++;
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev[3], c->ev[4]) == hipSuccess) c->last_ms["fsel_select"] = ms;
  return AVM_OK;
}

int avm_fsel_select(avm_ctx* c, avm_mem mem, const avm_fsel_batch* frame, int32_t* selected_ids, int32_t* n_selected, double* fvalues_opt) {
  if (!c) return AVM_ERR_INVALID;
  if (!frame || frame->n_problems != 1) return fail(c, AVM_ERR_INVALID, "avm_fsel_select takes exactly one frame (n_problems == 1)");
  avm_fsel_out out{n_selected, selected_ids, fvalues_opt, nullptr};
  return avm_fsel_select_batch(c, mem, frame, &out);
}

int avm_fsel_fallback_stats(const avm_ctx* c, int64_t out[4]) {
  if (!c || !out) return AVM_ERR_INVALID;
  out[0] = c->fsel_reruns, out[1] = c->fsel_failed_launches, out[2] = c->fsel_frame_mode, out[3] = c->fsel_calls;
  return AVM_OK;
}

int avm_fsel_horizon_imu(avm_ctx* c, avm_mem mem, const avm_fsel_horizon_in* in, double* hor_pos, double* hor_quat) {
  if (!c) return AVM_ERR_INVALID;
  (void)hipSetDevice(c->device);
  if (!in || !hor_pos || !hor_quat || in->n_problems < 0 || in->horizon < 1) return fail(c, AVM_ERR_INVALID, "null/negative argument");
  if (in->n_problems == 0) return AVM_OK;
  const size_t P = in->n_problems, H1 = (size_t)in->horizon + 1;
  avm_fsel_horizon_in d = *in;
  double *dp = hor_pos, *dq = hor_quat;
  if (mem == AVM_MEM_HOST) {
    int rc;
#define ST(field, type, count)                                                                                     \
  if ((rc = stage_in<type>(c, "h_" #field, in->field, (count), (const type**)&d.field)) != AVM_OK) return rc;
    ST(k_pos, double, P * 3)
    ST(k_quat, double, P * 4)
    ST(k_ba, double, P * 3)
    ST(k1_pos, double, P * 3)
    ST(k1_vel, double, P * 3)
    ST(k1_quat, double, P * 4)
    ST(acc, double, P * 3)
    ST(gyr, double, P * 3)
    ST(nr_imu, int32_t, P)
    ST(delta_imu, double, P)
#undef ST
    dp = static_cast<double*>(pool_get(c, "h_pos", sizeof(double) * P * H1 * 3));
    dq = static_cast<double*>(pool_get(c, "h_quat", sizeof(double) * P * H1 * 4));
    if (!dp || !dq) return fail(c, AVM_ERR_HIP, "hipMalloc failed (horizon out)");
  }
  HIPCHK(c, launch_fsel_horizon_imu(d, dp, dq, c->stream));
  if (mem == AVM_MEM_HOST) {
    HIPCHK(c, hipMemcpyAsync(hor_pos, dp, sizeof(double) * P * H1 * 3, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hor_quat, dq, sizeof(double) * P * H1 * 4, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return AVM_OK;
}

int avm_fsel_build_cloud(avm_ctx* c, avm_mem mem, const avm_window_batch* windows, const double* k1_pos, const double* k1_quat, int32_t max_cloud,
                         int32_t* n_cloud, double* cloud_xy, double* cloud_depth) {
  if (!c) return AVM_ERR_INVALID;
  (void)hipSetDevice(c->device);
  if (!windows || !k1_pos || !k1_quat || !n_cloud || !cloud_xy || !cloud_depth || windows->n_windows < 0 || max_cloud < 1)
    return fail(c, AVM_ERR_INVALID, "null/negative argument");
  if (windows->n_windows == 0) return AVM_OK;
  {
    const int vrc = validate_windows(c, mem, windows, CHK_TRACKS);
    if (vrc != AVM_OK) return vrc;
  }
  const size_t B = windows->n_windows;
  avm_window_batch d = *windows;
  const double *dp = k1_pos, *dq = k1_quat;
  int32_t* dn = n_cloud;
  double *dxy = cloud_xy, *ddep = cloud_depth;
  if (mem == AVM_MEM_HOST) {
    int rc;
#define ST(field, type, count)                                                                                          \
  if ((rc = stage_in<type>(c, "w_" #field, windows->field, (count), (const type**)&d.field)) != AVM_OK) return rc;
    ST(pose, double, B * 77)
    ST(ex_pose, double, B * 7)
    ST(inv_depth, double, B * windows->max_feat)
    ST(n_feat, int32_t, B)
    ST(feat_start, int32_t, B * windows->max_feat)
    ST(feat_obs_begin, int32_t, B * windows->max_feat)
    ST(obs_xy, double, B * windows->max_obs * 2)
#undef ST
    if ((rc = stage_in<double>(c, "c_k1p", k1_pos, B * 3, &dp)) != AVM_OK) return rc;
    if ((rc = stage_in<double>(c, "c_k1q", k1_quat, B * 4, &dq)) != AVM_OK) return rc;
    dn = static_cast<int32_t*>(pool_get(c, "c_n", sizeof(int32_t) * B));
    dxy = static_cast<double*>(pool_get(c, "c_xy", sizeof(double) * B * max_cloud * 2));
    ddep = static_cast<double*>(pool_get(c, "c_dep", sizeof(double) * B * max_cloud));
    if (!dn || !dxy || !ddep) return fail(c, AVM_ERR_HIP, "hipMalloc failed (cloud out)");
    HIPCHK(c, hipMemsetAsync(dxy, 0, sizeof(double) * B * max_cloud * 2, c->stream));
    HIPCHK(c, hipMemsetAsync(ddep, 0, sizeof(double) * B * max_cloud, c->stream));
  }
  HIPCHK(c, launch_fsel_build_cloud(d, dp, dq, max_cloud, dn, dxy, ddep, c->stream));
  if (mem == AVM_MEM_HOST) {
    HIPCHK(c, hipMemcpyAsync(n_cloud, dn, sizeof(int32_t) * B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cloud_xy, dxy, sizeof(double) * B * max_cloud * 2, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cloud_depth, ddep, sizeof(double) * B * max_cloud, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return AVM_OK;
}

int avm_fsel_nn_depth(avm_ctx* c, avm_mem mem, const avm_fsel_batch* batch, double* depth) {
  if (!c) return AVM_ERR_INVALID;
  (void)hipSetDevice(c->device);
  int rc = check_fsel(c, batch);
  if (rc != AVM_OK) return rc;
  if (!depth) return fail(c, AVM_ERR_INVALID, "null depth");
  if (batch->n_problems == 0) return AVM_OK;
  if ((rc = validate_fsel(c, mem, batch)) != AVM_OK) return rc;
  avm_fsel_batch d;
  if (mem == AVM_MEM_HOST) {
    if ((rc = stage_fsel(c, batch, &d)) != AVM_OK) return rc;
  } else {
    d = *batch;
  }
  const size_t n = (size_t)batch->n_problems * batch->max_cand;
  double* dd = mem == AVM_MEM_HOST ? static_cast<double*>(pool_get(c, "fi_nn", sizeof(double) * n)) : depth;
  if (!dd) return fail(c, AVM_ERR_HIP, "hipMalloc failed (nn depth)");
  HIPCHK(c, hipMemsetAsync(dd, 0, sizeof(double) * n, c->stream));
  double* kd = static_cast<double*>(pool_get(c, "fw_kd", sizeof(double) * fsel_kd_doubles(d)));
  if (!kd) return fail(c, AVM_ERR_HIP, "hipMalloc failed (kd-tree)");
  HIPCHK(c, launch_fsel_nn_depth(d, kd, dd, c->stream));
  if (mem == AVM_MEM_HOST) HIPCHK(c, hipMemcpyAsync(depth, dd, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return AVM_OK;
}

int avm_fsel_information(avm_ctx* c, avm_mem mem, const avm_fsel_batch* batch, double* omega, double* delta_cand, int32_t* cand_valid) {
  if (!c) return AVM_ERR_INVALID;
  (void)hipSetDevice(c->device);
  int rc = check_fsel(c, batch);
  if (rc != AVM_OK) return rc;
  if (batch->n_problems == 0) return AVM_OK;
  if ((rc = validate_fsel(c, mem, batch)) != AVM_OK) return rc;
  avm_fsel_batch d;
  if (mem == AVM_MEM_HOST) {
    if ((rc = stage_fsel(c, batch, &d)) != AVM_OK) return rc;
  } else {
    d = *batch;
  }
  FselBuffers w;
  if ((rc = fsel_buffers(c, &d, &w, false)) != AVM_OK) return rc;
  const size_t P = batch->n_problems, N = 9 * ((size_t)batch->horizon + 1), T = 3 * (size_t)batch->horizon, mc = batch->max_cand;
  double* d_om = nullptr;
  if (omega) {
    d_om = mem == AVM_MEM_HOST ? static_cast<double*>(pool_get(c, "fi_om", sizeof(double) * P * N * N)) : omega;
    if (!d_om) return fail(c, AVM_ERR_HIP, "hipMalloc failed (omega)");
  }
  avm_fsel_out none{nullptr, nullptr, nullptr, nullptr};
  HIPCHK(c, launch_fsel(d, w, none, d_om, false, 0, nullptr, c->stream));
  const hipMemcpyKind kind = mem == AVM_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if (omega && mem == AVM_MEM_HOST) HIPCHK(c, hipMemcpyAsync(omega, d_om, sizeof(double) * P * N * N, kind, c->stream));
  if (delta_cand) HIPCHK(c, hipMemcpyAsync(delta_cand, w.delta, sizeof(double) * P * mc * T * T, kind, c->stream));
  if (cand_valid) HIPCHK(c, hipMemcpyAsync(cand_valid, w.valid, sizeof(int32_t) * P * mc, kind, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return AVM_OK;
}

}  // extern "C"
