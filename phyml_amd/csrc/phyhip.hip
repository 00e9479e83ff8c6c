// phyhip.hip -- host side of libphyhip.so: instance table, device memory, the deferred operation
// queue and the C ABI declared in include/phyhip.h.  gfx950 only; no CPU fallback: every entry point
// fails with PHYHIP_ERROR_NO_RESOURCE when no device is visible.
#include "../../include/phyhip.h"
#include "phyhip_kernels.hpp"
#include "phyhip_aa.hpp"
#include "phyhip_nt2.hpp"
#include "phyhip_big.hpp"

#include <rccl/rccl.h>

#include <cfloat>
#include <functional>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

using namespace phyhip;

// Timing-only kernel variants (PHYHIP_ABLATE, PHYHIP_NOLOADS: results invalid) and the first-generation nucleotide
// kernel as an A/B reference for C <= 4 exist only in builds with -DPHYHIP_DIAG (tools/build_diag.sh); the product
// library reads none of those switches.
#ifdef PHYHIP_DIAG
constexpr bool kDiag = true;
#else
constexpr bool kDiag = false;
#endif
// A/B switches (another kernel or another route to the same numbers: PHYHIP_NT_GROUPS, _NT2_DIST, _DIST, _BLOCK, _GENERIC_NT/AA,
// _FOLD_PMATS, _PM_COPY, _EAGER_PMAT, _ARGS_RECS, _ARG_UPLOADS, _SPLIT_REDUCE, _SPIN, _RESIDENT_DIRECT, _AA_NW) are read by the
// diag build only; the product library reads PHYHIP_DEVICE, _RESIDENT, _RESIDENT_IDLE_US, _RESIDENT_STATS, _HOST_SUM and
// _SHARD_THREADS.  tests/test_gpu_switches.py runs every A/B switch on the diag build and holds it to the default's numbers.
static inline const char *diag_env(const char *name) { return kDiag ? getenv(name) : nullptr; }
// PHYHIP_DIAG + PHYHIP_HOSTPROF=1: where the host's time per scalar-returning call goes (cycle counter, printed at finalize)
struct HostProf { unsigned long long prep = 0, launch = 0, wait = 0, n_launch = 0, n_wait = 0, t_first = 0, t_last = 0; };
static HostProf g_hp;
static inline unsigned long long hp_now() { return kDiag ? __builtin_ia32_rdtsc() : 0ull; }

namespace
{

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
  char    buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(call)                                                                                         \
  do                                                                                                         \
  {                                                                                                          \
    hipError_t e_ = (call);                                                                                  \
    if (e_ != hipSuccess)                                                                                    \
      return fail(e_ == hipErrorOutOfMemory ? PHYHIP_ERROR_OUT_OF_MEMORY : PHYHIP_ERROR_GENERAL,             \
                  "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);                 \
  } while (0)

// Pinned staging ring for small host->device payloads (operation lists, transition matrices, edge
// lengths).  A chunk is recycled only after the copies issued from it have executed.
struct StagingRing
{
  static constexpr int kChunks = 8;
  size_t               chunk_bytes = 0;
  char                *base        = nullptr;
  hipEvent_t           ev[kChunks];
  bool                 pending[kChunks];
  int                  cur  = 0;
  size_t               used = 0;
  std::function<int()> before_rotate;

  int init(size_t bytes)
  {
    chunk_bytes = (bytes + 255) & ~size_t(255);
    HIPCHK(hipHostMalloc((void **)&base, chunk_bytes * kChunks, hipHostMallocDefault));
    for (int i = 0; i < kChunks; ++i)
    {
      HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
      pending[i] = false;
    }
    return 0;
  }
  void destroy()
  {
    if (!base) return;
    for (int i = 0; i < kChunks; ++i) (void)hipEventDestroy(ev[i]);
    (void)hipHostFree(base);
    base = nullptr;
  }
  // reserve `bytes` (<= chunk_bytes) of pinned memory that stays valid until the stream reaches `seal`
  int alloc(size_t bytes, hipStream_t s, void **out)
  {
    bytes = (bytes + 15) & ~size_t(15);
    if (bytes > chunk_bytes) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "staging request of %zu bytes too large", bytes);
    if (used + bytes > chunk_bytes)
    {
      // work that was queued against this chunk but not yet launched (matrix uploads) goes into the stream first, so
      // that the event below really seals everything that reads the chunk
      if (before_rotate)
      {
        int rc = before_rotate();
        if (rc) return rc;
      }
      HIPCHK(hipEventRecord(ev[cur], s));
      pending[cur] = true;
      cur          = (cur + 1) % kChunks;
      used         = 0;
      if (pending[cur])
      {
        HIPCHK(hipEventSynchronize(ev[cur]));
        pending[cur] = false;
      }
    }
    *out = base + (size_t)cur * chunk_bytes + used;
    used += bytes;
    return 0;
  }
};

struct Collective;

constexpr int kResidentDirect = 16;    // up to this many resident workgroups poll the host themselves, above that workgroup 0 relays (measured: 8 / 16 / 32, tools/gpu_direct_ab.sh)
// Host side of one set of resident workgroups (see resident_dlk_kernel / resident_nt2_kernel)
struct Resident
{
  ResidentCmd        *cmd = nullptr;   // host-mapped command record
  unsigned long long *mail = nullptr;  // device mailbox (larger grids: workgroup 0 relays the commands)
  hipStream_t         stream[2] = {nullptr, nullptr};
  unsigned long long  gen = 0, seq = 0; // launch generation; commands issued
  unsigned long long  api_no = 0;       // entry-point call of the last command
  bool                launched = false;
  int                 grid = 0;
  struct timespec     t_launch = {0, 0}, t_cmd = {0, 0};
  unsigned long long  n_cmd = 0, n_launch = 0, n_silent = 0, n_busy = 0; // phyhip_get_resident_stats
  double              ns_wait = 0.0;   // PHYHIP_RESIDENT_STATS: host time from a command's last word to its answer, summed
};

struct Instance
{
  Collective *co         = nullptr; // one-process-per-GPU mode: communicator attached by phyhip_comm_init_rank
  double     *d_red      = nullptr; // ... and this shard's {warning, lnL, dlnL} reduction buffer (owned by co)
  int         dev        = 0;
  hipStream_t stream     = nullptr;
  bool        own_stream = true;
  int         tips = 0, nbuf = 0, S = 0, C = 0, CP = 0, nmat = 0;
  long long   P = 0, Ppad = 0; // Ppad: patterns per buffer as allocated (P, or P rounded up to 16 when perm)
  bool        class_axis = false; // categories are the classes of a mixture (PHYHIP_FLAG_CLASS_AXIS; TreeParams::class_axis)
  bool        generic_loop = false; // PHYHIP_FLAG_GENERIC_LOOP: the reference's generic loop (`--cov`): plain kernel, no all-ones shortcut
  int         NE = 1;          // eigen systems / frequency vectors held: C with the class axis, else 1
  bool        perm = false;    // 20-state buffers in the MFMA fragment-major layout (phyhip_aa.hpp)
  int         aa_nw = 1;               // 20 states: consumer waves (= wave-tiles) per workgroup of traverse_aa_kernel
  bool        soa = false;     // 4-state buffers pattern-minor, lane-per-pattern kernel (phyhip_nt2.hpp)
  int         grid_nt2 = 0;
  int         nt_groups = 1; // lanes per pattern in the lane-per-pattern nucleotide kernel
  double     *d_afrag = nullptr;
  int         grid_aa = 0;

  double   *d_partials = nullptr;
  int      *d_scales   = nullptr;
  uint8_t  *d_tipcodes = nullptr;
  uint32_t *d_tipmasks = nullptr;      // 20-state fragment-major instances: [tip][Ppad] allowed-state masks (traverse_aa_kernel)
  uint32_t *d_masks    = nullptr;
  double   *d_pmats    = nullptr;
  double   *d_wght     = nullptr;
  short    *d_invar    = nullptr;
  double   *d_model    = nullptr; // pi[S] catw[C] catr[C] eval[S] evec[S*S] ivec[S*S]
  double   *d_pi, *d_catw, *d_catr, *d_eval, *d_evec, *d_ivec;
  double   *d_site_lnl = nullptr, *d_site_lk = nullptr, *d_site_cat = nullptr, *d_dot = nullptr;
  int      *d_fact     = nullptr;
  double   *d_block    = nullptr; // [2][grid]
  double   *d_result   = nullptr; // [2]
  double   *h_result   = nullptr; // pinned, device-visible: [0..1] results, [2] sequence number (as u64)
  unsigned long long seq = 0;     // evaluations handed to the host so far
  bool      spin_wait  = true;    // PHYHIP_SPIN=0: always hipStreamSynchronize
  bool      warn_current = false; // *h_warn belongs to the evaluation the host last waited for (none launched since)
  int      *d_warn     = nullptr;
  int      *h_warn     = nullptr;
  int       mix_invar_model = 0;  // +I mixture (phyhip_set_mixture_invariant_sites): carried by the first / class-axis instance
  double    mix_pinvar = 0.0, mix_pi_inv[20] = {0};
  HostBlock *h_blocks  = nullptr; // host-mapped {block sum, tag} records of the host-side final sum
  int       host_sum_n = 0;       // > 0: the evaluation in flight is finished by the host from this many records per sum
  int       host_sum_ns = 1;      // ... and this many sums (1: lnL; 2: lnL and dlnL)
  size_t    h_blocks_cap = 0;
  // resident evaluator (resident_dlk_kernel): dLk / eigen-basis Lk of small alignments without a launch per call
  bool         resident = true;       // PHYHIP_RESIDENT=0: every evaluation is a kernel launch
  int          resident_direct = kResidentDirect; // PHYHIP_RESIDENT_DIRECT: up to this many workgroups poll the host themselves
  double       resident_idle_us = 1000.0; // PHYHIP_RESIDENT_IDLE_US: the workgroups leave after this long without a command
  Resident     rd, rt;                // the dLk evaluator (resident_dlk_kernel) and the short-launch one (resident_nt2_kernel)
  // large-grid resident evaluator (resident_big_kernel, phyhip_big.hpp): nucleotide instances of more than kResidentMaxGrid tiles
  Resident     rb;
  TreeParams   rb_static;             // what its workgroups were launched with
  double      *d_tile_sums = nullptr; // [2][max(grid_nt2, n_vdlk)] tile sums of commands whose final sum runs on the device
  unsigned    *d_big_tickets = nullptr;
  HostBlock   *d_big_recs = nullptr;  // [2][kBigGroupWgs] partial sums per workgroup (BigCtl::wg_recs)
  bool         big_group_sum = true;  // (diag: PHYHIP_BIG_GROUP_SUM=0 keeps the per-tile sums and the tickets)
  unsigned long long *d_big_stamps = nullptr; // PHYHIP_RESIDENT_STATS: stamps of the last command per workgroup (BigCtl::stamps)
  int          n_vdlk = 0;            // virtual blocks (one wave each) of a dLk evaluation: dlk64_kernel's grid
  int          big_wgs = 0, big_nw = 0; // its workgroups and waves per workgroup
  // commands of more tiles than this add their tile sums on the device (one record to the host); below, a record per tile to the
  // host.  Measured (SPR candidate / dLk, us): 79 tiles 11.3 / 8.9 against 14.0 / 10.6, 125 tiles 12.1 / 9.0 against 14.0 / 10.4,
  // 250 tiles 14.6 / 12.4 against 14.6 / 11.0, 625 tiles 16.5 / 13.1 against 15.3 / 11.1 (host / device)
  int          big_device_sum = 200;
  int          big_streak = 0;        // consecutive evaluations the resident workgroups could have taken (they are launched at 2)
  unsigned long long rb_dlk_api = 0;  // entry-point call of the last dLk command they served
  int          cus = 256;
  Resident    *r_inflight = nullptr;  // whose command the evaluation in flight is
  DlkParams    r_static;              // what the resident workgroups were launched with
  TreeParams   rt_static;
  // the evaluation last handed to resident_nt2_kernel, kept until it is answered (unanswered: it is launched instead)
  std::vector<DevOp>  rt_ops;
  std::vector<int>    rt_pm_idx;
  std::vector<double> rt_pm_len;
  bool         rt_skip = false;       // the evaluation being repeated after an unanswered command goes the ordinary way
  unsigned long long clean_epoch = 0, rt_epoch = 0; // times the stream was found finished after having run something; at the last command
  bool         touched_call = false;  // this entry-point call has put something on the stream
  // Is everything queued on the stream known to be finished?  The resident workgroups are not ordered with the stream, so
  // they may only be used when it is.  Conservative bookkeeping: every entry point marks the stream dirty (GET_INST); the
  // ones that queue nothing put the previous state back; Update_Eigen_Lr -- what precedes a chain of dLk calls -- ends with
  // a report to the host (stamp) after which the stream is clean; a stream synchronisation cleans it at once.
  bool               stream_dirty = true, dirty_prev = true;
  unsigned long long stamp_seq = 0, clean_after = 0;
  // Lk(b) with update_eigen_lr (src/lk.c: Update_Eigen_Lr, then Lk_Core on the same edge) opens a chain of dLk calls: the
  // edge evaluation that directly follows an Update_Eigen_Lr makes its workgroups complete their stores before they post
  // their sums (TreeParams::fence_post), so that once the host has the scalar the whole stream is known to be finished.
  unsigned long long api_no = 0, eig_api_no = 0; // entry-point calls so far; the call that queued the last eigen_lr kernel
  bool               fenced_eval = false;        // the evaluation in flight posts behind fences // stamps issued; the stamp whose arrival makes a non-dirty stream clean
  int          wall_khz = 0;          // rate of the device's wall_clock64()
  bool      host_sum   = true;    // PHYHIP_HOST_SUM=0: large grids use final_reduce_kernel instead
  void     *d_pmscratch = nullptr; // [pm_scratch_cap] ints + doubles for phyhip_update_transition_matrices
  int       pm_scratch_cap = 0;
  char     *d_ops      = nullptr; // ring of op lists on the device (slim DevOp or fat IssueRec+ExecRec)
  size_t    ops_slot_bytes = 0;
  int       ops_cap = 0, ops_slots = 4, ops_slot = 0;
  int       grid = 0, grid_nt = 0, block_nt = 64;

  std::vector<DevOp>                     pending;
  std::vector<int>                       pm_idx;    // queued device-side matrix rebuilds (index, edge length)
  std::vector<double>                    pm_len;
  std::vector<int>                       pm_slot;   // matrix index -> position in pm_idx, or -1
  std::vector<std::vector<DevOp>>        slot_ops;  // what each device ring slot currently holds (content cache)
  std::vector<int>                       slot_kind; // 0 slim, 1 fat dist 1, 2 fat dist 2
  std::vector<unsigned char>             mat_in_queue; // matrix index referenced by a queued op
  std::vector<int>                       up_idx;       // host-computed matrices waiting for their upload launch
  std::vector<const double *>            up_src;       // ... their copies in pinned staging memory
  std::vector<int>                       up_slot;      // per matrix: position in up_idx or -1
  std::vector<uint32_t>                  masks;
  std::unordered_map<uint32_t, int>      mask_code;
  bool                                   masks_dirty = false;
  std::vector<double>                    h_rates, h_eval, h_evec, h_ivec;
  std::vector<double>                    h_model;      // host shadow of d_model: a setter called with unchanged values
  std::vector<unsigned char>             h_model_set;  // ... (callers re-push the model before every evaluation) costs nothing
  std::vector<short>                     h_invar;
  bool                                   h_invar_set = false;
  StagingRing                            ring;

  double l_min = 1.e-8, l_max = 100., br_len_mult = 1.0, pinvar = 0.0; // src/init.c:711-714
  int    apply_scaling = 1, invar_model = 0;
  bool   want_site_outputs = true;
  int    nt2_dist = 2;       // PHYHIP_NT2_DIST=1: the lane-per-pattern kernel loads one operation ahead instead of two (a wave more per SIMD)
  int    prefetch_dist = 2;  // PHYHIP_DIST: load-stage distance of the nt pipeline (1 or 2)
  bool   fold_pmats = true;    // PHYHIP_FOLD_PMATS=0: always rebuild transition matrices with a separate pmat_kernel launch
  bool   pm_copy = false;      // PHYHIP_PM_COPY=1: copy the P-matrix work list to the device first (measured: +2 us per step at cfg2, +3..6 at cfg3)
  bool   split_reduce = false, split_reduce_forced = false; // PHYHIP_SPLIT_REDUCE: separate final_reduce_kernel instead of the fused last-workgroup sum
  unsigned *d_tickets = nullptr;
  double   *d_mixexpl = nullptr; // expl pairs of the classes of a mixture evaluation (first instance only)
  int    ablate = 0;         // PHYHIP_ABLATE (-DPHYHIP_DIAG builds only): timing-only kernel variants (results invalid)
  unsigned long long *d_dbg = nullptr; // cycle stamps of PHYHIP_ABLATE=8
  bool   args_recs = true;   // PHYHIP_ARGS_RECS=0: operation records of 1-2-operation launches go through the slot ring too
  bool   fuse_eigen = true;  // PHYHIP_FUSE_EIGEN=0 (diag): Update_Eigen_Lr always as its own eigen_lr_kernel launch
  bool   arg_uploads = true; // PHYHIP_ARG_UPLOADS=0: host-computed matrices always go through upload_matrices_kernel
  bool   eager_pmats = true; // PHYHIP_EAGER_PMAT=0: whole-tree matrix batches wait for the traversal launch too
  bool   no_loads = false;   // PHYHIP_NOLOADS (-DPHYHIP_DIAG builds only): zero-size every child load (timing only)
  bool   generic_nt = false; // PHYHIP_GENERIC_NT=1: run nucleotides through the generic (non-pipelined) kernel

  bool       prof = false;
  hipEvent_t pe0 = nullptr, pe1 = nullptr;
  hipEvent_t ev_sync = nullptr; // orders this instance's stream before another instance's (mixture evaluations)
  hipEvent_t ev_big = nullptr;  // ... and behind the exit of its large-grid resident workgroups (big_release)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pairs;
  struct ProfPair { hipEvent_t a, b; int kind; };
  std::vector<ProfPair> prof_aux;          // eigen-basis kernels while profiling: kind 0 eigen_lr_kernel (K3), 1 dlk_kernel (K4)
  double     prof_aux_ms[2] = {0.0, 0.0};
  int        prof_aux_n[2]  = {0, 0};
  double     prof_ms = 0.0, prof_updates = 0.0;
  double     prof_rd_bytes = 0.0, prof_wr_bytes = 0.0; // traffic model of the profiled launches (phyhip_profile_read_traffic)
  int        prof_n = 0;
};

// The calling thread's current device is ASKED, not remembered: a host application (or another library in its process) may
// call hipSetDevice between two calls of this ABI, and a remembered value would then send the next launch to the wrong device
// without any error.  hipGetDevice reads a thread-local of the runtime (tens of nanoseconds).
thread_local int g_cur_dev = -1; // (what this library last set: only a hint for the cases below that do not ask)
static inline int make_current(int dev)
{
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != dev)
  {
    HIPCHK(hipSetDevice(dev));
  }
  g_cur_dev = dev;
  return 0;
}

static void big_release(Instance *I, bool restart_streak = true);

// ---- the instance table and THE choke point of the resident protocol -------------------------------
// Resident workgroups are not ordered with the instance's stream, so they may only be used while nothing queued on it is still
// running.  That is kept true by construction, not by convention:
//   * the table of instances is PRIVATE to InstanceTable.  The only way from an instance number to an Instance inside an entry
//     point of the C ABI is an Entered<...> object (GET_INST / GET_INST_RES below), whose constructor declares the stream dirty
//     -- "may have enqueued work" is the default -- and, unless the entry point says it keeps them, makes the large-grid
//     resident workgroups leave before anything of this call can reach the stream;
//   * the three ways back to "the stream is as it was found" are member functions that exist only on Entered<true>: an entry
//     point that did not declare itself resident-aware cannot call them (static_assert: it does not compile).
// What remains a reviewed list (tests/test_abi.py::test_resident_choke_point) is WHICH entry points say GET_INST_RES.
// Construction and teardown (create / finalize, the wiring of a sharded group) use the lifecycle accessors, which run no call.
template <bool KeepsResidents> class Entered;
class InstanceTable
{
  static std::mutex              mu_;
  static std::vector<Instance *> tab_;
  static Instance *at(int id)
  {
    std::lock_guard<std::mutex> lk(mu_);
    if (id < 0 || id >= (int)tab_.size()) return nullptr;
    return tab_[id];
  }
  template <bool> friend class Entered;

 public:
  static int add(Instance *I) // phyhip_create_instance
  {
    std::lock_guard<std::mutex> lk(mu_);
    for (size_t i = 0; i < tab_.size(); ++i)
      if (!tab_[i])
      {
        tab_[i] = I;
        return (int)i;
      }
    tab_.push_back(I);
    return (int)tab_.size() - 1;
  }
  static Instance *remove(int id) // phyhip_finalize_instance: out of the table, the caller frees it
  {
    std::lock_guard<std::mutex> lk(mu_);
    if (id < 0 || id >= (int)tab_.size()) return nullptr;
    Instance *I = tab_[id];
    tab_[id]    = nullptr;
    return I;
  }
  static Instance *wiring(int id) { return at(id); } // a sharded group attaching / detaching its sub-instances: no call runs
};
std::mutex              InstanceTable::mu_;
std::vector<Instance *> InstanceTable::tab_;

template <bool KeepsResidents> class Entered
{
  Instance *I_  = nullptr;
  int       rc_ = 0;

 public:
  explicit Entered(int id)
  {
    I_ = InstanceTable::at(id);
    if (!I_)
    {
      rc_ = fail(PHYHIP_ERROR_UNINITIALIZED_INSTANCE, "instance %d does not exist", id);
      return;
    }
    // hipSetDevice costs about a microsecond; the surface is entered hundreds of thousands of times per tree search (SURVEY
    // section 6), so only switch when the calling thread is actually on another device (make_current)
    if ((rc_ = make_current(I_->dev)) != 0) return;
    I_->dirty_prev   = I_->stream_dirty;
    I_->stream_dirty = true;
    I_->touched_call = false;
    ++I_->api_no;
    // Everything but the entry points the large-grid resident workgroups (phyhip_big.hpp) serve, the ones that only queue and
    // the queries that touch no device memory may put copies or kernels on the instance's stream, which must then be ordered
    // behind the resident workgroups' exit (what they wrote sits in their L2s until they leave)
    if (!KeepsResidents) big_release(I_);
  }
  Entered(const Entered &) = delete;
  Entered &operator=(const Entered &) = delete;
  int       rc() const { return rc_; }
  Instance *inst() const { return I_; }
  // the call only queued (operations, matrix rebuilds): nothing went onto the stream unless a flush inside it said so
  void leave_queued_only() const
  {
    static_assert(KeepsResidents, "only an entry point that declared itself resident-aware (GET_INST_RES) may restore the stream's state");
    if (!I_->touched_call) I_->stream_dirty = I_->dirty_prev;
  }
  // the call queues nothing by itself; whatever it runs (flush / eigen_eval) marks the stream itself
  void leave_untouched() const
  {
    static_assert(KeepsResidents, "only an entry point that declared itself resident-aware (GET_INST_RES) may restore the stream's state");
    I_->stream_dirty = I_->dirty_prev;
  }
  // a query: queues nothing and is not a step of the call sequence the resident evaluators watch
  void leave_query() const
  {
    static_assert(KeepsResidents, "only an entry point that declared itself resident-aware (GET_INST_RES) may restore the stream's state");
    I_->stream_dirty = I_->dirty_prev;
    --I_->api_no;
  }
};
#define GET_INST_AS(I, id, keeps)                                                                            \
  const Entered<keeps> I##_call(id);                                                                         \
  if (I##_call.rc()) return I##_call.rc();                                                                   \
  Instance *const I = I##_call.inst();
#define GET_INST_RES(I, id) GET_INST_AS(I, id, true)
#define GET_INST(I, id) GET_INST_AS(I, id, false)

int next_pow2(int x)
{
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}

size_t buf_elems(const Instance *I) { return I->perm ? aa_buf_elems(I->Ppad, I->C) : (size_t)I->Ppad * I->C * I->S; }
// ints per partials buffer in the scale table: one exponent per pattern, or per (class, pattern) with the class axis
size_t scale_elems(const Instance *I) { return (size_t)I->Ppad * (I->class_axis ? I->C : 1); }

// element offset of (pattern, category, state) inside a device partials buffer of a non-host layout
size_t dev_off(const Instance *I, long long p, int c, int s)
{
  if (I->perm) return aa_off(p, I->C, c, s);
  return ((size_t)(c * 2 + (s >> 1)) * I->Ppad + (size_t)p) * 2 + (size_t)(s & 1); // pattern-minor, state pairs of 16 bytes
}

TreeParams base_params(Instance *I)
{
  TreeParams q;
  memset(&q, 0, sizeof q);
  q.partials = I->d_partials; q.scales = I->d_scales;
  q.wght = I->d_wght; q.P = I->P; q.Ppad = I->Ppad; q.perm = I->perm ? 1 : (I->soa ? 2 : 0); q.C = I->C; q.tip_count = I->tips;
  q.apply_scaling = I->apply_scaling; q.pi = I->d_pi; q.cat_w = I->d_catw; q.invar_model = I->invar_model;
  q.pinvar = I->pinvar; q.invar = I->d_invar; q.block_sums = I->d_block; q.warn = I->d_warn; q.fact = I->d_fact;
  q.class_axis = I->class_axis ? 1 : 0;
  q.generic_loop = I->generic_loop ? 1 : 0;
  return q;
}

RO base_ro(Instance *I, const DevOp *ops)
{
  RO r;
  r.ops = ops; r.pmats = I->d_pmats; r.tip_codes = I->d_tipcodes; r.code_masks = I->d_masks;
  return r;
}

template <typename F> int dispatch_shape(const Instance *I, F &&f)
{
  // (S, CP) instantiations: nucleotides / amino acids x category count padded to a power of two
#define CASE(S_, CP_)                                                                                        \
  if (I->S == S_ && I->CP == CP_) return f(std::integral_constant<int, S_>(), std::integral_constant<int, CP_>());
  CASE(4, 1) CASE(4, 2) CASE(4, 4) CASE(4, 8) CASE(20, 1) CASE(20, 2) CASE(20, 4) CASE(20, 8)
#undef CASE
  return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "no kernel for %d states x %d categories", I->S, I->C);
}

int upload_masks(Instance *I)
{
  if (!I->masks_dirty) return 0;
  if (I->masks.size() > 256) return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "more than 256 distinct tip state sets");
  HIPCHK(hipMemcpyAsync(I->d_masks, I->masks.data(), I->masks.size() * sizeof(uint32_t), hipMemcpyHostToDevice, I->stream));
  HIPCHK(hipStreamSynchronize(I->stream));
  I->masks_dirty = false;
  return 0;
}

struct EdgeEval
{
  int     parent, child, pm;
  double *dev_out;  // optional user device pointer
  bool    to_host;
  double *warn_out; // sharded evaluation: device double receiving the numerical-warning flag (or nullptr)
  bool    eigen = false; // not an evaluation: the eigen-basis products of Update_Eigen_Lr for (parent = left, child = right)
};

// Host-computed matrices queued by phyhip_set_transition_matrix: one launch per kUploadBatch of them.
int flush_uploads(Instance *I)
{
  if (!I->up_idx.empty()) I->touched_call = true;
  size_t done = 0;
  while (done < I->up_idx.size())
  {
    const int n = (int)std::min<size_t>(I->up_idx.size() - done, kUploadBatch);
    MatUploadParams q;
    memset(&q, 0, sizeof q);
    q.count = n; q.S = I->S; q.C = I->C; q.pmats = I->d_pmats; q.afrag = I->perm ? I->d_afrag : nullptr;
    for (int k = 0; k < n; ++k) { q.idx[k] = I->up_idx[done + k]; q.src[k] = I->up_src[done + k]; }
    hipLaunchKernelGGL(upload_matrices_kernel, dim3(n), dim3(256), sizeof(double) * (size_t)I->C * I->S * I->S, I->stream, q);
    HIPCHK(hipGetLastError());
    done += n;
  }
  for (int m : I->up_idx) I->up_slot[m] = -1;
  I->up_idx.clear();
  I->up_src.clear();
  return 0;
}

// "Everything queued on this stream before me has finished and is in memory": one thread, one store into host-mapped
// memory.  Launched behind evaluations of large nucleotide instances that the resident workgroups could not take (stream
// not known to be idle, first call of a streak): the host finds the stream idle again without synchronising it -- and without
// the launched kernel's waves writing back their L2 before they post (megabytes of dirty lines at these sizes).
static __global__ void stream_stamp_kernel(unsigned long long *stamp_host, unsigned long long stamp)
{
  __hip_atomic_store(stamp_host, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static int stamp_stream(Instance *I);

// Rebuild every queued transition matrix on the device: one staged copy of (index, length) pairs, one launch.
int flush_pmats(Instance *I)
{
  big_release(I);
  I->touched_call = true;
  int done = 0, rc = 0;
  if (!I->up_idx.empty() && (rc = flush_uploads(I))) return rc;
  const int count = (int)I->pm_idx.size();
  while (done < count)
  {
    const int  n     = std::min(count - done, I->pm_scratch_cap);
    const bool small = n <= kSmallPm; // short lists (SPR: 3 per candidate) ride in the kernel arguments
    PmatParams q;
    memset(&q, 0, sizeof q);
    if (small)
    {
      for (int k = 0; k < n; ++k)
      {
        q.small_idx[k] = I->pm_idx[done + k];
        q.small_len[k] = I->pm_len[done + k];
      }
    }
    else
    {
      void        *st = nullptr;
      const size_t bi = (sizeof(int) * n + 15) & ~size_t(15), bl = sizeof(double) * n;
      rc = I->ring.alloc(bi + bl, I->stream, &st);
      if (rc) return rc;
      memcpy(st, I->pm_idx.data() + done, sizeof(int) * n);
      memcpy((char *)st + bi, I->pm_len.data() + done, bl);
      if (I->pm_copy)
      {
        HIPCHK(hipMemcpyAsync(I->d_pmscratch, st, bi + bl, hipMemcpyHostToDevice, I->stream));
        q.indices = (const int *)I->d_pmscratch;
        q.lengths = (const double *)((char *)I->d_pmscratch + bi);
      }
      else
      { // the kernels read the (index, length) pairs straight from the pinned staging chunk: a few hundred bytes over
        // the host link cost less than a copy command ahead of the launch
        q.indices = (const int *)st;
        q.lengths = (const double *)((char *)st + bi);
      }
    }
    q.count = n;
    q.S = I->S; q.C = I->C; q.U = I->d_evec; q.V = I->d_ivec; q.R = I->d_eval; q.rates = I->d_catr;
    q.br_len_mult = I->br_len_mult; q.l_min = I->l_min; q.l_max = I->l_max; q.pmats = I->d_pmats;
    // (20 states: a short list is latency -- 1024 threads, two entries each, products pre-formed: 7.9 vs 11.5 us for three
    // matrices; a whole tree's 397 matrices: 512 threads without the extra phase 14.7 us; 256 / 1024 threads 16.3-17.4 / 15.5)
    int threads = (I->S == 4) ? 64 : (n <= 16 ? 1024 : 512);
    if (const char *e = diag_env("PHYHIP_PMAT_THREADS"))
    { // (a multiple of 64 within the kernel's launch bounds, or ignored)
      const int v = atoi(e);
      if (v >= 64 && v % 64 == 0 && v <= (I->S == 4 ? 64 : 1024)) threads = v;
    }
    q.afrag = I->perm ? I->d_afrag : nullptr; // 20 states: the MFMA A-operand fragments come out of the same kernel
    q.class_axis = I->class_axis ? 1 : 0;
    const size_t lds = sizeof(double) * ((size_t)2 * I->C * I->S + (size_t)2 * I->C * I->S * I->S + (size_t)2 * I->NE * I->S * I->S);
    if (I->S == 4) hipLaunchKernelGGL((pmat_kernel<4, true>), dim3(n), dim3(threads), lds, I->stream, q);
    else if (n <= 16) hipLaunchKernelGGL((pmat_kernel<20, true>), dim3(n), dim3(threads), lds, I->stream, q);
    else hipLaunchKernelGGL((pmat_kernel<20, false>), dim3(n), dim3(threads), lds, I->stream, q);
    HIPCHK(hipGetLastError());
    done += n;
  }
  for (int m : I->pm_idx) I->pm_slot[m] = -1;
  I->pm_idx.clear();
  I->pm_len.clear();
  return 0;
}

// Final sum inside the producing kernel (last workgroup) or as a separate 1-block kernel?  Measured on MI355X (round 2,
// tools/gpu_step_ab.sh, with PHYHIP_SPLIT_REDUCE actually honoured): fused saves the second launch (~3.4 us + gap)
// whenever the grid is small -- every SPR / Br_Len_Opt call on small and mid-sized alignments.  On large grids it costs
// the traversal kernel ~6-10 % (cfg2 198 vs 186 us, 125 000 patterns 469 vs 414, 1 M 3.54 vs 3.22 ms, cfg3 574 vs 553):
// a workgroup must see its block sum acknowledged by memory before it draws its ticket, i.e. it waits for ALL its
// outstanding result stores instead of retiring behind them, and holds its wave slot meanwhile.  PHYHIP_SPLIT_REDUCE=0/1
// forces either.
static bool fuse_reduce(const Instance *I, int nblocks)
{
  if (I->split_reduce_forced) return !I->split_reduce;
  return nblocks <= 512;
}

// ---- resident evaluators: host side -----------------------------------------------------------------------------
constexpr int kResidentMaxGrid = 64;   // resident evaluator: alignments of up to this many dLk workgroups
constexpr int kResidentSilent = -4242; // wait_host_sum: the resident workgroups did not answer (not an error)
static double ns_since(const struct timespec &t0)
{
  struct timespec t1;
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec);
}

// Tell the resident workgroups (if any) to leave and wait until they have.
static void resident_stop(Resident &R)
{
  if (!R.cmd || !R.launched) return;
  __atomic_store_n(&R.cmd->ctl.w[1], 1ull, __ATOMIC_RELEASE);
  for (hipStream_t st : R.stream)
    if (st) (void)hipStreamSynchronize(st);
  __atomic_store_n(&R.cmd->ctl.w[1], 0ull, __ATOMIC_RELEASE);
  R.launched = false;
}

static void resident_free(Resident &R)
{
  if (R.cmd)
  { // (also a generation that was only told to leave, big_release: nobody may still be polling the record when it is freed)
    __atomic_store_n(&R.cmd->ctl.w[1], 1ull, __ATOMIC_RELEASE);
    for (hipStream_t st : R.stream)
      if (st) (void)hipStreamSynchronize(st);
    R.launched = false;
  }
  resident_stop(R);
  if (R.cmd) (void)hipHostFree(R.cmd);
  if (R.mail) (void)hipFree(R.mail);
  for (hipStream_t st : R.stream)
    if (st) (void)hipStreamDestroy(st);
  R = Resident();
}

// First half of a (re)launch: generation gen + 1 supersedes whatever is left of the previous one (its workgroups see the
// new number at their next poll and leave); commands up to `served` count as done.  The caller launches its kernel with `r`
// on `*st`, then calls resident_launched().
// in_order: every generation on the SAME stream -- the large-grid workgroups leave what they stored to the end of their
// kernel (no write-back per command), so a new generation must not start before the previous one has ended.
static int resident_prepare(Instance *I, Resident &R, int grid, int n_words, unsigned long long served, ResidentCtl &r, hipStream_t *st,
                            bool in_order = false)
{
  if (!R.cmd)
  {
    HIPCHK(hipHostMalloc((void **)&R.cmd, sizeof(ResidentCmd), hipHostMallocMapped | hipHostMallocCoherent)); // (polled from the device while it changes)
    memset(R.cmd, 0, sizeof(ResidentCmd));
    HIPCHK(hipMalloc((void **)&R.mail, sizeof(ResidentCmd)));
    HIPCHK(hipMemset(R.mail, 0, sizeof(ResidentCmd)));
    for (hipStream_t &s2 : R.stream) HIPCHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  }
  ++R.n_launch;
  ++R.gen;
  __atomic_store_n(&R.cmd->ctl.w[0], R.gen, __ATOMIC_RELEASE);
  r.cmd = R.cmd; r.gen = R.gen; r.start_seq = served; r.n_sectors = (n_words + kResidentPay - 1) / kResidentPay;
  r.mail = R.mail; r.relay = grid > I->resident_direct ? 1 : 0;
  if (I->wall_khz <= 0)
  {
    int dev = 0, khz = 0;
    HIPCHK(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    I->wall_khz = khz;
  }
  r.idle_ticks = (unsigned long long)(I->resident_idle_us * 1e-3 * (double)I->wall_khz); // wall_clock64 ticks
  *st = R.stream[in_order ? 0 : (R.gen & 1)];
  return 0;
}
static void resident_launched(Resident &R, int grid)
{
  R.grid = grid; R.launched = true;
  clock_gettime(CLOCK_MONOTONIC, &R.t_launch);
}

// Have the workgroups of the current generation left?  (Workgroup 0 reports it, see ResidentCmd::report.)
static bool resident_gone(const Resident &R)
{
  return R.cmd && __atomic_load_n(&R.cmd->report.w[0], __ATOMIC_ACQUIRE) == R.gen;
}

// The command: payload words into their sectors, each sector's number last (see ResidentCmd)
static void resident_send(Instance *I, Resident &R, const unsigned long long *words, int n_words)
{
  ++R.seq;
  for (int l = 0; l * kResidentPay < n_words; ++l)
  {
    ResidentSector &sc = R.cmd->sec[l];
    for (int k = 0; k < kResidentPay && l * kResidentPay + k < n_words; ++k) sc.w[k] = words[l * kResidentPay + k];
    __atomic_store_n(&sc.seq, R.seq, __ATOMIC_RELEASE);
  }
  R.api_no = I->api_no;
  ++R.n_cmd;
  clock_gettime(CLOCK_MONOTONIC, &R.t_cmd);
  I->r_inflight = &R;
}

static int resident_launch_dlk(Instance *I, const DlkParams &qs, int dgrid, unsigned long long served)
{
  ResidentCtl r;
  hipStream_t st;
  int rc = resident_prepare(I, I->rd, dgrid, 3 + I->C * 2 * I->S, served, r, &st);
  if (rc) return rc;
  rc = dispatch_shape(I, [&](auto s, auto cp) {
    constexpr int S_ = decltype(s)::value, CP_ = decltype(cp)::value;
    hipLaunchKernelGGL((resident_dlk_kernel<S_, CP_>), dim3(dgrid), dim3(256), 0, st, qs, r);
    return 0;
  });
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  I->r_static = qs;
  resident_launched(I->rd, dgrid);
  return 0;
}


// instances whose short launches (SPR candidates, Lk(b), Update_Eigen_Lr) the resident workgroups of resident_nt2_kernel may take
static bool resident_short_eligible(const Instance *I)
{
  return I->resident && I->spin_wait && I->host_sum && I->soa && !I->co && !I->class_axis && I->grid_nt2 <= kResidentMaxGrid &&
         !I->ablate && I->nt_groups <= 2;
}

// ---- the large-grid resident evaluator (phyhip_big.hpp): host side ---------------------------------------------------
// One instance per device at a time: the resident workgroups fill the device (a workgroup per CU at the register budget of
// the traversal kernel), so a second set could not start before the first has left.
static std::atomic<Instance *> g_big_owner[64];

// Instances whose dLk runs in the traversal's tiles (dlk_tile / dlk64_kernel) -- a property of the instance alone, so that
// its evaluations return the same doubles whether the resident workgroups are enabled or not ...
static bool big_shape(const Instance *I)
{
  return I->host_sum && I->soa && !I->co && !I->class_axis && I->grid_nt2 > kResidentMaxGrid && I->nt_groups <= 2 && !I->ablate;
}
// ... and whether those workgroups may serve it
static bool big_eligible(const Instance *I) { return big_shape(I) && I->resident && I->spin_wait && I->dev >= 0 && I->dev < 64; }

// Tell the large-grid resident workgroups (if any) to leave -- without waiting for it: whatever this instance launches next
// needs their wave slots, and gets them as they go.  (A new generation number is all it takes: workgroup 0 sees it at its
// next poll of the host record and passes it on through the mailbox.)
// The instance's stream is ordered behind their exit: the resident workgroups do not write back what they store while they
// stay (a write-back per command of megabytes of dirty lines cost more than the command: measured, round 4) -- the end of
// their kernel does, and kernels and copies of this instance that come later must find it in memory.
static void big_release(Instance *I, bool restart_streak)
{
  if (restart_streak) I->big_streak = 0;
  Resident &R = I->rb;
  if (R.cmd && R.launched)
  {
    hipStream_t st = R.stream[0]; // (where every generation of these workgroups runs, resident_prepare)
    ++R.gen;
    __atomic_store_n(&R.cmd->ctl.w[0], R.gen, __ATOMIC_RELEASE);
    R.launched = false;
    if (I->ev_big && hipEventRecord(I->ev_big, st) == hipSuccess) (void)hipStreamWaitEvent(I->stream, I->ev_big, 0);
    else (void)hipStreamSynchronize(st);
    I->stream_dirty = true; I->touched_call = true; // (the stream now waits for something)
  }
  if (I->dev >= 0 && I->dev < 64)
  {
    Instance *me = I;
    g_big_owner[I->dev].compare_exchange_strong(me, nullptr);
  }
}

// Is everything queued on the instance's stream known to have finished?  (The flags of enter_stream_work and the stamp
// launched behind the last evaluation.  hipStreamQuery was tried instead: it answers "not ready" for a stream whose last
// command is a kernel until a marker it inserts itself has completed -- with a new launch after every query, never.)
static bool big_clean(Instance *I)
{
  if (I->touched_call || I->dirty_prev) return false;
  if (I->clean_after)
  { // the stamp launched behind the last kernel of the stream (stamp_stream): arrived = idle; not yet = launch this one too
    // (a bounded wait: the stamp runs a launch gap behind the kernel whose scalar the host already has -- a host that comes
    // back within microseconds would otherwise find it missing call after call and never get to the resident workgroups)
    volatile unsigned long long *stamp = reinterpret_cast<volatile unsigned long long *>(I->h_result + 3);
    if (*stamp < I->clean_after)
    {
      struct timespec t0;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      for (long it = 1; *stamp < I->clean_after; ++it)
      {
        __builtin_ia32_pause();
        if ((it & 63) == 0 && ns_since(t0) > 40000.0) return false;
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    I->clean_after = 0;
    ++I->clean_epoch; // (kernels ran since the last command: the resident workgroups re-read device memory)
  }
  return true;
}

// Launch the stamp behind what the call has just put on the stream; from now on the stream counts as idle once it arrives.
static int stamp_stream(Instance *I)
{
  const unsigned long long v = ++I->stamp_seq;
  hipLaunchKernelGGL(stream_stamp_kernel, dim3(1), dim3(1), 0, I->stream, reinterpret_cast<unsigned long long *>(I->h_result + 3), v);
  HIPCHK(hipGetLastError());
  I->stream_dirty = false; I->clean_after = v;
  return 0;
}

// Could the resident workgroups take an evaluation of this instance right now?  Counts the calls in a row for which the answer
// was yes: the workgroups are only launched at the second (a launch per call that alternates with other launches would cost
// more than it saves).
static bool big_ready(Instance *I)
{
  if (!big_eligible(I) || I->prof || I->rt_skip) return false;
  Instance *owner = g_big_owner[I->dev].load();
  if (owner && owner != I) return false;
  if (!big_clean(I)) return false;
  return true;
}

// The final sum through one partial sum per workgroup (phyhip_big.hpp, kBigGroupSum): the tiles of a workgroup are one
// accumulator of final_reduce_kernel's order only when there are exactly as many workgroups as accumulators
static bool big_sum_by_group(const Instance *I, int tiles)
{
  return I->big_group_sum && I->big_wgs == kBigGroupWgs && tiles <= kBigGroupWgs * kBigGroupTiles;
}

static int big_launch(Instance *I, const TreeParams &sq)
{
  Resident &R = I->rb;
  if (!I->d_tile_sums)
  {
    const size_t n = (size_t)std::max(I->grid_nt2, I->n_vdlk);
    HIPCHK(hipMalloc((void **)&I->d_tile_sums, 2 * n * sizeof(double)));
    HIPCHK(hipMalloc((void **)&I->d_big_tickets, sizeof(unsigned) * (1 + kTicketGroups)));
    HIPCHK(hipMemsetAsync(I->d_big_tickets, 0, sizeof(unsigned) * (1 + kTicketGroups), I->stream));
    HIPCHK(hipMalloc((void **)&I->d_big_recs, sizeof(HostBlock) * 2 * kBigGroupWgs));
    HIPCHK(hipMemsetAsync(I->d_big_recs, 0, sizeof(HostBlock) * 2 * kBigGroupWgs, I->stream)); // (tag 0: no evaluation's)
    if (getenv("PHYHIP_RESIDENT_STATS"))
    {
      HIPCHK(hipMalloc((void **)&I->d_big_stamps, sizeof(unsigned long long) * 8 * (size_t)I->big_wgs));
      HIPCHK(hipMemsetAsync(I->d_big_stamps, 0, sizeof(unsigned long long) * 8 * (size_t)I->big_wgs, I->stream));
    }
    HIPCHK(hipStreamSynchronize(I->stream)); // (the stream is idle: this evaluation was about to bypass it)
  }
  Instance *none = nullptr;
  if (!g_big_owner[I->dev].compare_exchange_strong(none, I) && none != I) return 1; // (somebody else's: launch the evaluation)
  ResidentCtl r;
  hipStream_t st;
  int rc = resident_prepare(I, R, I->big_wgs, kBigWords, R.seq, r, &st, true);
  if (rc) return rc;
  // (a generation that left in the middle of a command -- workgroups that started late find "leave" in the mailbox before they
  // find the command -- leaves tickets drawn and never reset: every generation starts from zero, in stream order)
  HIPCHK(hipMemsetAsync(I->d_big_tickets, 0, sizeof(unsigned) * (1 + kTicketGroups), st));
  BigArgs a;
  a.t = sq; a.r = r;
  a.b.n_tiles = I->grid_nt2; a.b.n_vdlk = I->n_vdlk; a.b.tile_sums = I->d_tile_sums; a.b.tickets = I->d_big_tickets; a.b.dot_prod = I->d_dot;
  a.b.stamps = I->d_big_stamps; a.b.wg_recs = I->d_big_recs;
  a.pmats = I->d_pmats; a.tip_codes = I->d_tipcodes;
  if (launch_resident_big(I->C, I->nt_groups, I->big_wgs, st, a) != 0)
    return fail(PHYHIP_ERROR_GENERAL, "large-grid resident evaluator: no kernel for %d categories in %d groups", I->C, I->nt_groups);
  HIPCHK(hipGetLastError());
  memcpy(&I->rb_static, &sq, sizeof sq);
  resident_launched(R, I->big_wgs);
  return 0;
}

// The launch arguments of the resident workgroups: everything of a short launch's TreeParams that does not change per call
static TreeParams big_static_params(Instance *I)
{
  TreeParams sq = base_params(I);
  sq.host_blocks = I->h_blocks; sq.warn = I->h_warn; sq.fence_post = 0; sq.recs_in_args = 1; sq.edge_eval = 1;
  sq.br_len_mult = I->br_len_mult; sq.l_min = I->l_min; sq.l_max = I->l_max; sq.pmats_rw = I->d_pmats;
  sq.dot_out = I->d_dot;
  if (I->want_site_outputs) { sq.site_lnl = I->d_site_lnl; sq.site_lk = I->d_site_lk; sq.site_cat = I->d_site_cat; }
  // (4 states, one eigen system, <= 4 categories: big_eligible) the eigen system and the category rates ride in the arguments
  memcpy(sq.m_evec, I->h_evec.data(), 16 * sizeof(double)); memcpy(sq.m_ivec, I->h_ivec.data(), 16 * sizeof(double));
  memcpy(sq.m_eval, I->h_eval.data(), 4 * sizeof(double));
  for (int c = 0; c < 4; ++c) sq.m_rates[c] = c < I->C ? I->h_rates[c] : 0.0;
  return sq;
}

// Make sure the resident workgroups are there (launched with the instance's current parameters).  Returns 0: they are,
// 1: not this time (first call of a streak, or the device belongs to another instance's workgroups), < 0: error.
static int big_ensure(Instance *I)
{
  Resident        &R = I->rb;
  const TreeParams sq = big_static_params(I);
  const bool       same = R.launched && memcmp(&I->rb_static, &sq, sizeof sq) == 0;
  if (same && !resident_gone(R)) return 0;
  if (R.launched && !same) big_release(I), I->big_streak = 2; // (parameters changed: a new generation with the new ones)
  if (++I->big_streak < 2) return 1;
  return big_launch(I, sq);
}

// dLk in the traversal's tiles (dlk_tile): the launched form of what the large-grid resident workgroups serve
template <int CP> static void launch_dlk64(Instance *I, const DlkParams &q, int dgrid)
{
  if (I->nt_groups == 2) hipLaunchKernelGGL((dlk64_kernel<4, CP, (CP >= 2 ? CP / 2 : 1)>), dim3(dgrid), dim3(64), 0, I->stream, q);
  else hipLaunchKernelGGL((dlk64_kernel<4, CP, CP>), dim3(dgrid), dim3(64), 0, I->stream, q);
}

// Launch the queued operations (and optionally the fused edge evaluation) as one traversal kernel.
int flush_impl(Instance *I, const EdgeEval *ee)
{
  const unsigned long long hp0 = hp_now();
  const int n_ops = (int)I->pending.size();
  int rc = 0;
  if (n_ops > 0 || ee || !I->pm_idx.empty() || !I->up_idx.empty()) I->stream_dirty = true;
  if (ee) I->fenced_eval = false;
  // a short list of device-built matrices is folded into the lane-per-pattern nucleotide kernel's prologue when the grid
  // is small (measured: 16.7 vs 17.8 us per scalar-returning call on a 382-pattern search prefix; at 100 000 patterns
  // the redundant per-workgroup rebuild costs more than the launch it saves: 45.1 vs 42.5 us per SPR candidate)
  static const int fold_grid_max = diag_env("PHYHIP_FOLD_GRID") ? atoi(diag_env("PHYHIP_FOLD_GRID")) : 512;
  // large grids: an evaluation the large-grid resident workgroups can take (phyhip_big.hpp) carries its matrices in the command
  const bool big_fit = ee && (ee->eigen || (ee->to_host && !ee->dev_out)) && n_ops <= 2 && I->args_recs && I->fold_pmats &&
                       (int)I->pm_idx.size() <= 4 && I->up_idx.empty() && big_eligible(I) && !I->prof && !I->rt_skip;
  const bool big_try = big_fit && big_ready(I);
  // ... and they are there (or launched now): decided before the records are built -- a resident command of two operations runs
  // them one after the other per tile, without register forwarding between them (phyhip_big.hpp)
  bool big_take = false;
  if (big_try)
  {
    const int brc = big_ensure(I);
    if (brc < 0) return brc;
    big_take = brc == 0;
  }
  if (kDiag && ee && getenv("PHYHIP_RESIDENT_DEBUG") && big_shape(I))
    fprintf(stderr, "big: fit %d try %d | eligible %d ops %d pm %zu up %zu prof %d skip %d | dirty_prev %d touched %d clean_after %llu stamp %llu streak %d launched %d owner %p me %p\n",
            (int)big_fit, (int)big_try, (int)big_eligible(I), n_ops, I->pm_idx.size(), I->up_idx.size(), (int)I->prof, (int)I->rt_skip, (int)I->dirty_prev,
            (int)I->touched_call, I->clean_after, *reinterpret_cast<volatile unsigned long long *>(I->h_result + 3), I->big_streak, (int)I->rb.launched,
            (void *)g_big_owner[I->dev < 64 ? I->dev : 0].load(), (void *)I);
  const bool fold_pm = I->soa && I->fold_pmats && (I->grid_nt2 <= fold_grid_max || big_try) && !I->pm_idx.empty() && (int)I->pm_idx.size() <= 8 &&
                       I->up_idx.empty() && (n_ops > 0 || ee) && I->C <= 4 && !I->class_axis && !(I->ablate & 8);
  // a short list of HOST-computed matrices rides in the arguments of the lane-per-pattern nucleotide kernel at every grid size
  // (TreeParams::n_up): no upload kernel in front of the traversal
  const bool arg_up = I->soa && I->arg_uploads && !I->up_idx.empty() && (int)I->up_idx.size() <= kArgUp && I->pm_idx.empty() &&
                      (n_ops > 0 || ee) && I->C <= 4 && !I->class_axis && !(I->ablate & 8);
  if (!fold_pm && !arg_up && (!I->pm_idx.empty() || !I->up_idx.empty()) && (rc = flush_pmats(I))) return rc;
  if (n_ops == 0 && !ee) return 0;
  rc = upload_masks(I);
  if (rc) return rc;

  TreeParams q = base_params(I);
  RO         ro = base_ro(I, nullptr);
  bool       fused_sum = false;
  if (arg_up)
  {
    q.n_up = (int)I->up_idx.size();
    for (int k = 0; k < q.n_up; ++k)
    {
      q.up_idx[k] = I->up_idx[k];
      memcpy(q.up_val[k], I->up_src[k], sizeof(double) * 16 * I->C); // (pinned staging memory: an ordinary host read)
    }
    q.pmats_rw = I->d_pmats;
    for (int m : I->up_idx) I->up_slot[m] = -1;
    I->up_idx.clear();
    I->up_src.clear();
  }
  if (fold_pm)
  {
    q.n_fresh = (int)I->pm_idx.size();
    for (int k = 0; k < q.n_fresh; ++k) { q.fresh_idx[k] = I->pm_idx[k]; q.fresh_len[k] = I->pm_len[k]; }
    // (4 states, one eigen system, <= 4 categories: see fold_pm) the eigen system rides in the arguments as well
    memcpy(q.m_evec, I->h_evec.data(), 16 * sizeof(double)); memcpy(q.m_ivec, I->h_ivec.data(), 16 * sizeof(double));
    memcpy(q.m_eval, I->h_eval.data(), 4 * sizeof(double));
    for (int c = 0; c < 4; ++c) q.m_rates[c] = c < I->C ? I->h_rates[c] : 0.0;
    q.br_len_mult = I->br_len_mult; q.l_min = I->l_min; q.l_max = I->l_max; q.pmats_rw = I->d_pmats;
    // (the matrix queue is cleared only after the launch that rebuilds it has been issued, see below)
  }
  const bool fat = ((I->S == 4) && !I->generic_nt) || I->perm;
  const IssueRec *d_irec = nullptr;
  const ExecRec  *d_xrec = nullptr;
  q.last_dest = -1;
  const int kind = fat ? I->prefetch_dist : 0;
  int       hit  = -1, new_slot = -1, host_sum_n = 0;
  if (n_ops > 0)
  { // an operation list identical to one still sitting in a device slot (repeated Lk(NULL) on one topology) is
    // neither rebuilt nor re-uploaded
    for (int sl = 0; sl < I->ops_slots && hit < 0; ++sl)
      if (I->slot_kind[sl] == kind && I->slot_ops[sl].size() == (size_t)n_ops &&
          memcmp(I->slot_ops[sl].data(), I->pending.data(), sizeof(DevOp) * n_ops) == 0)
        hit = sl;
  }
  if (n_ops > 0 && hit >= 0)
  {
    char *dst = I->d_ops + (size_t)hit * I->ops_slot_bytes;
    if (!fat) ro.ops = reinterpret_cast<const DevOp *>(dst);
    else
    {
      d_irec = reinterpret_cast<const IssueRec *>(dst);
      d_xrec = reinterpret_cast<const ExecRec *>(dst + sizeof(IssueRec) * (n_ops + (n_ops & 1)));
      q.last_dest = I->pending[n_ops - 1].dest;
    }
    q.n_ops = fat ? n_ops + (n_ops & 1) : n_ops;
  }
  else if (n_ops > 0)
  {
    char *dst = I->d_ops + (size_t)I->ops_slot * I->ops_slot_bytes;
    // one or two operations of the lane-per-pattern nucleotide kernel travel in the kernel arguments (phyhip_nt2.hpp):
    // no staging, no copy command, and the device slots keep the long lists they cache
    const bool in_args = fat && (I->soa || I->perm) && I->args_recs && n_ops <= 2;
    IssueRec   arg_ir[2];
    ExecRec    arg_xr[2];
    if (!in_args)
    {
      new_slot = I->ops_slot;
      I->slot_kind[new_slot] = -1; // the slot's old content is gone; it holds the new list only once the copy was issued
    }
    void *st = nullptr;
    if (!fat)
    {
      rc = I->ring.alloc(sizeof(DevOp) * n_ops, I->stream, &st);
      if (rc) return rc;
      memcpy(st, I->pending.data(), sizeof(DevOp) * n_ops);
      HIPCHK(hipMemcpyAsync(dst, st, sizeof(DevOp) * n_ops, hipMemcpyHostToDevice, I->stream));
      ro.ops = reinterpret_cast<const DevOp *>(dst);
    }
    else
    {
      // one record pair per operation, all address arithmetic done here once.  The kernel alternates two
      // register sets, so an odd list is padded with a re-execution of its last operation (idempotent: same
      // inputs, same output, same address) whose forwarding flags are computed for its own position.
      const int    n_rec = n_ops + (n_ops & 1);
      const size_t ib = sizeof(IssueRec) * n_rec, xb = sizeof(ExecRec) * n_rec;
      if (!in_args)
      {
        rc = I->ring.alloc(ib + xb, I->stream, &st);
        if (rc) return rc;
      }
      IssueRec *ir = in_args ? arg_ir : reinterpret_cast<IssueRec *>(st);
      ExecRec  *xr = in_args ? arg_xr : reinterpret_cast<ExecRec *>((char *)st + ib);
      const size_t   bufbytes = buf_elems(I) * sizeof(double);
      // spare word of the data descriptors: byte offset of the child's matrix (natural table, or the MFMA
      // A-fragment table for the 20-state kernel)
      const unsigned matbytes = I->perm ? (unsigned)(kAaMat * sizeof(double))
                                        : (unsigned)((size_t)I->C * I->S * I->S * sizeof(double));
      auto desc = [](const void *base, size_t bytes, unsigned x) {
        Desc d;
        d.base = (unsigned long long)(uintptr_t)base; d.bytes = (unsigned)bytes; d.x = x;
        return d;
      };
      auto at = [&](int k) -> const DevOp & { return I->pending[std::min(k, n_ops - 1)]; };
      for (int k = 0; k < n_rec; ++k)
      {
        const DevOp &o  = at(k);
        const int    e1 = (k >= 1 && !big_take) ? at(k - 1).dest : -1;
        const int    e2 = (k >= 2 && I->prefetch_dist == 2) ? at(k - 2).dest : -1;
        unsigned     fl = 0;
        auto child = [&](int c, unsigned tipbit, unsigned f1bit, unsigned f2bit, Desc &data, Desc &scale, Desc &tip,
                         unsigned pmoff) {
          const bool t = c < I->tips;
          const bool f1 = !t && c == e1, f2 = !t && !f1 && c == e2;
          const bool ld = !t && !f1 && !f2 && !I->no_loads;
          if (t) fl |= tipbit;
          if (f1) fl |= f1bit;
          if (f2) fl |= f2bit;
          const size_t b = ld ? (size_t)(c - I->tips) : 0;
          data  = desc(I->d_partials + b * buf_elems(I), ld ? bufbytes : 0, pmoff);
          scale = desc(I->d_scales + b * scale_elems(I), ld ? scale_elems(I) * 4 : 0, 0);
          tip   = desc(I->d_tipcodes + (size_t)(t ? c : 0) * I->Ppad, t ? (size_t)I->Ppad : 0, 0);
          // lane-per-pattern nucleotide kernel and the 20-state kernel: ONE auxiliary dword load per child -- the scale
          // descriptor of a tip child points at its tip row instead (spare word 1: the kernel then reads the aligned dword
          // holding the byte)
          if (I->soa && t) scale = desc(I->d_tipcodes + (size_t)c * I->Ppad, (size_t)I->Ppad, 1);
          if (I->perm && t) scale = desc(I->d_tipmasks + (size_t)c * I->Ppad, (size_t)I->Ppad * 4, 1); // (the mask itself)
        };
        child(o.c1, kOpTip1, kOpF11, kOpF12, ir[k].c1_data, ir[k].c1_scale, ir[k].c1_tip, (unsigned)o.pm1 * matbytes);
        child(o.c2, kOpTip2, kOpF21, kOpF22, ir[k].c2_data, ir[k].c2_scale, ir[k].c2_tip, (unsigned)o.pm2 * matbytes);
        const size_t b = (size_t)(o.dest - I->tips);
        xr[k].dst_data  = desc(I->d_partials + b * buf_elems(I), bufbytes, fl);
        xr[k].dst_scale = desc(I->d_scales + b * scale_elems(I), scale_elems(I) * 4, 0);
      }
      // (reading short lists straight from the pinned staging memory instead was measured: no gain)
      if (in_args)
      {
        q.recs_in_args = 1; q.n_real_ops = n_ops;
        q.arg_ir[0] = ir[0]; q.arg_ir[1] = ir[1];
        q.arg_xr[0] = xr[0]; q.arg_xr[1] = xr[1];
      }
      else
      {
        HIPCHK(hipMemcpyAsync(dst, st, ib + xb, hipMemcpyHostToDevice, I->stream));
        d_irec = reinterpret_cast<const IssueRec *>(dst);
        d_xrec = reinterpret_cast<const ExecRec *>(dst + ib);
      }
      q.last_dest = I->pending[n_ops - 1].dest;
    }
    q.n_ops = fat ? n_ops + (n_ops & 1) : n_ops;
    if (new_slot >= 0)
    {
      I->slot_ops[new_slot]  = I->pending;
      I->slot_kind[new_slot] = kind;
      I->ops_slot = (new_slot + 1) % I->ops_slots;
    }
  }
  // small nucleotide alignments: the resident short-launch evaluator (resident_nt2_kernel) may take the call
  const bool rt_grid = resident_short_eligible(I);
  if (ee && ee->eigen)
  { // Update_Eigen_Lr fused behind the queued partial update(s): no sums, the products go to d_dot
    q.edge_eval = 2; q.e_parent = ee->parent; q.e_child = ee->child; q.e_pm = 0; q.dot_out = I->d_dot;
    memcpy(q.m_evec, I->h_evec.data(), 16 * sizeof(double)); memcpy(q.m_ivec, I->h_ivec.data(), 16 * sizeof(double));
    if (n_ops == 0 && I->args_recs) { q.recs_in_args = 1; q.n_real_ops = 0; }
    if (q.recs_in_args)
    {
      auto untouched = [&](int idx) {
        if (idx < I->tips) return false;
        for (const DevOp &o : I->pending)
          if (o.dest == idx) return false;
        return true;
      };
      q.e_prefetch = (untouched(ee->parent) ? 1 : 0) | (untouched(ee->child) ? 2 : 0);
    }
    // completion as an evaluation's: every workgroup fences its stores and posts an (empty) record the caller waits for -- the
    // stream is clean when phyhip_update_eigen_lr returns, and the resident workgroups can take the call (the only
    // instances that come here: phyhip_update_eigen_lr)
    q.host_blocks = I->h_blocks; q.host_tag = ++I->seq; q.warn = I->h_warn;
    host_sum_n    = I->grid_nt2;
  }
  else if (ee)
  {
    I->warn_current = false;
    q.edge_eval = 1; q.e_parent = ee->parent; q.e_child = ee->child; q.e_pm = ee->pm;
    if (n_ops == 0 && fat && I->soa && I->args_recs) { q.recs_in_args = 1; q.n_real_ops = 0; } // (evaluation-only short launch)
    if (q.recs_in_args)
    { // short launch: the kernel fetches the sides of the evaluation edge that no queued operation writes up front
      auto untouched = [&](int idx) {
        if (idx < I->tips) return false;
        for (const DevOp &o : I->pending)
          if (o.dest == idx) return false;
        return true;
      };
      q.e_prefetch = (untouched(ee->parent) ? 1 : 0) | (untouched(ee->child) ? 2 : 0);
    }
    const int nblk = I->soa ? I->grid_nt2 : (I->perm ? I->grid_aa : (fat ? I->grid_nt : I->grid));
    // (fusing on large grids was measured for one-operation launches too: 61.6 vs 41.9 us per SPR candidate at cfg5)
    fused_sum = !I->class_axis && fuse_reduce(I, nblk) && !(I->host_sum && ee->to_host && !ee->dev_out);
    if (fused_sum)
    { // the traversal kernel's last workgroup finishes the sum and reports to the host
      q.tickets = I->d_tickets; q.result = ee->dev_out ? ee->dev_out : I->d_result;
      q.result_host = ee->to_host ? I->h_result : nullptr; q.warn_host = I->h_warn;
      q.seq = ee->to_host ? ++I->seq : 0ull;
      q.warn_out = ee->warn_out;
    }
    if (!fused_sum && ee->to_host && !ee->dev_out && I->host_sum && !I->class_axis)
    { // the workgroups post their sums to the host, which adds them (wait_result) -- at every grid size: on small grids
      // this replaces the ticket draw of the fused sum (block sum written through, atomic, fence, re-read: ~3 us of
      // dependent memory round trips inside a ~10 us kernel), on large ones the second launch
      q.host_blocks = I->h_blocks; q.host_tag = ++I->seq;
      q.warn        = I->h_warn;   // raised straight in host-mapped memory
      *I->h_warn    = 0;
      host_sum_n    = nblk;
      if (I->eig_api_no && I->api_no == I->eig_api_no + 1 && (I->soa || I->perm))
      { // (the kernels that honour it; eig_api_no is only set for small alignments with the resident evaluator enabled)
        q.fence_post   = 1;
        I->fenced_eval = true;
      }
    }
    if (I->want_site_outputs) { q.site_lnl = I->d_site_lnl; q.site_lk = I->d_site_lk; q.site_cat = I->d_site_cat; }
  }
  // ---- small nucleotide alignments: the resident short-launch evaluator (resident_nt2_kernel) ------------------------
  static const bool rtdbg = kDiag && getenv("PHYHIP_RESIDENT_DEBUG") != nullptr; // (diag build: why an evaluation was launched)
  if (rtdbg && ee)
    fprintf(stderr, "rt: grid_ok %d (res %d spin %d hs %d soa %d co %d cls %d g2 %d abl %d grp %d) hsn %d args %d fresh %d site %d prof %d skip %d dirty_prev %d touched %d\n",
            (int)rt_grid, (int)I->resident, (int)I->spin_wait, (int)I->host_sum, (int)I->soa, I->co != nullptr, (int)I->class_axis, I->grid_nt2,
            I->ablate, I->nt_groups, host_sum_n, q.recs_in_args, q.n_fresh, (int)I->want_site_outputs, (int)I->prof, (int)I->rt_skip,
            (int)I->dirty_prev, (int)I->touched_call);
  if (rt_grid && host_sum_n > 0)
  { // every evaluation of such an instance completes its stores before it posts: the stream is clean once the scalar is back
    q.fence_post   = 1;
    I->fenced_eval = true;
  }
  if (rt_grid && host_sum_n > 0 && q.recs_in_args && q.n_fresh <= 4 && q.n_up == 0 && !I->prof && !I->rt_skip)
  {
    bool clean = !I->dirty_prev && !I->touched_call;
    if (clean && I->clean_after)
    { // the report of the last Update_Eigen_Lr (bounded wait, else the ordinary launch)
      volatile unsigned long long *stamp = reinterpret_cast<volatile unsigned long long *>(I->h_result + 3);
      struct timespec t0;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      for (long it = 1; *stamp < I->clean_after && clean; ++it)
      {
        __builtin_ia32_pause();
        if ((it & 255) == 0 && ns_since(t0) > 200000.0) clean = false;
      }
      if (clean) { __atomic_thread_fence(__ATOMIC_ACQUIRE); I->clean_after = 0; ++I->clean_epoch; }
    }
    if (!clean) ++I->rt.n_busy;
    else
    {
      Resident  &R = I->rt;
      // what the workgroups are launched with: everything of the launch form's arguments that does not change per call
      TreeParams sq = base_params(I);
      sq.host_blocks = I->h_blocks; sq.warn = I->h_warn; sq.fence_post = 1; sq.recs_in_args = 1; sq.edge_eval = 1;
      sq.br_len_mult = I->br_len_mult; sq.l_min = I->l_min; sq.l_max = I->l_max; sq.pmats_rw = I->d_pmats;
      sq.dot_out = I->d_dot;
      if (I->want_site_outputs) { sq.site_lnl = I->d_site_lnl; sq.site_lk = I->d_site_lk; sq.site_cat = I->d_site_cat; }
      if (!R.launched || R.grid != I->grid_nt2 || memcmp(&I->rt_static, &sq, sizeof sq) != 0 || resident_gone(R))
      {
        if (R.launched && (R.grid != I->grid_nt2 || memcmp(&I->rt_static, &sq, sizeof sq) != 0)) resident_stop(R);
        ResidentCtl r;
        hipStream_t st;
        if ((rc = resident_prepare(I, R, I->grid_nt2, kResidentNtWords, R.seq, r, &st))) return rc;
#define NT2RES(c_, g_)                                                                                                      \
  hipLaunchKernelGGL((resident_nt2_kernel<c_, g_>), dim3(I->grid_nt2), dim3(64), 0, st, sq, r, (const double *)I->d_pmats,   \
                     (const uint8_t *)I->d_tipcodes, (const double *)I->d_evec, (const double *)I->d_ivec,                    \
                     (const double *)I->d_eval, (const double *)I->d_catr);                                                   \
  break;
        switch (I->C * 8 + I->nt_groups)
        {
          case 1 * 8 + 1: NT2RES(1, 1)
          case 2 * 8 + 1: NT2RES(2, 1)
          case 2 * 8 + 2: NT2RES(2, 2)
          case 3 * 8 + 1: NT2RES(3, 1)
          case 4 * 8 + 1: NT2RES(4, 1)
          case 4 * 8 + 2: NT2RES(4, 2)
          default: return fail(PHYHIP_ERROR_GENERAL, "resident evaluator: no kernel for %d categories in %d groups", I->C, I->nt_groups);
        }
#undef NT2RES
        HIPCHK(hipGetLastError());
        memcpy(&I->rt_static, &sq, sizeof sq);
        resident_launched(R, I->grid_nt2);
      }
      unsigned long long words[kResidentNtWords];
      memset(words, 0, sizeof words);
      const bool changed = I->clean_epoch != I->rt_epoch; // the stream ran something since the last command
      words[0] = q.host_tag;
      words[1] = (unsigned long long)q.n_real_ops | (changed ? 4ull : 0ull) | ((unsigned long long)q.n_fresh << 4) |
                 ((unsigned long long)q.e_prefetch << 8) | (q.edge_eval == 2 ? 1ull << 10 : 0ull);
      words[2] = (unsigned long long)(unsigned)q.e_parent | ((unsigned long long)(unsigned)q.e_child << 32);
      words[3] = (unsigned long long)(unsigned)q.e_pm | ((unsigned long long)(unsigned)q.last_dest << 32);
      for (int k = 0; k < q.n_fresh; ++k)
      {
        words[4 + k / 2] |= (unsigned long long)(unsigned)q.fresh_idx[k] << (32 * (k & 1));
        memcpy(&words[6 + k], &q.fresh_len[k], 8);
      }
      auto put = [&](int k, const Desc &d) { words[k] = d.base; words[k + 1] = (unsigned long long)d.bytes | ((unsigned long long)d.x << 32); };
      for (int o = 0; o < q.n_real_ops; ++o)
      {
        put(10 + o * 12, q.arg_ir[o].c1_data); put(12 + o * 12, q.arg_ir[o].c2_data);
        put(14 + o * 12, q.arg_ir[o].c1_scale); put(16 + o * 12, q.arg_ir[o].c2_scale);
        put(18 + o * 12, q.arg_xr[o].dst_data); put(20 + o * 12, q.arg_xr[o].dst_scale);
      }
      // kept until the answer is in: an evaluation nobody answers is launched the ordinary way (phyhip_calculate_edge_log_likelihoods)
      I->rt_ops = I->pending; I->rt_pm_idx = I->pm_idx; I->rt_pm_len = I->pm_len;
      resident_send(I, R, words, kResidentNtWords); // (every sector the workgroups wait for carries the command's number)
      I->rt_epoch = I->clean_epoch;
      I->host_sum_n = host_sum_n; I->host_sum_ns = 1;
      if (fold_pm)
      {
        for (int m : I->pm_idx) I->pm_slot[m] = -1;
        I->pm_idx.clear();
        I->pm_len.clear();
      }
      I->pending.clear();
      std::fill(I->mat_in_queue.begin(), I->mat_in_queue.end(), 0);
      return 0;
    }
  }
  // ---- large nucleotide alignments: the large-grid resident evaluator (resident_big_kernel) -------------------------------
  // (launches of such an instance do not fence their stores before they post -- with megabytes of results in the L2s a
  // write-back per wave costs more than the launch; whether the stream is idle again is found by querying it, big_clean)
  if (big_take)
  {
    if (!(host_sum_n > 0 && q.recs_in_args && q.n_fresh <= 4 && q.n_up == 0))
      return fail(PHYHIP_ERROR_GENERAL, "large-grid resident evaluator: an evaluation it cannot take (%d records, %d matrices)", host_sum_n, q.n_fresh);
    {
      Resident &R = I->rb;
      unsigned long long words[kBigWords];
      memset(words, 0, sizeof words);
      const bool changed = I->clean_epoch != I->rt_epoch; // the stream ran something since the last command
      const bool dsum = host_sum_n > I->big_device_sum;
      words[0] = q.host_tag;
      words[1] = (unsigned long long)q.n_real_ops | (changed ? kBigChanged : 0ull) | ((unsigned long long)q.n_fresh << 4) |
                 ((unsigned long long)q.e_prefetch << 8) | (q.edge_eval == 2 ? kBigEigen : 0ull) | (dsum ? kBigDeviceSum : 0ull) |
                 (dsum && big_sum_by_group(I, host_sum_n) ? kBigGroupSum : 0ull);
      words[2] = (unsigned long long)(unsigned)q.e_parent | ((unsigned long long)(unsigned)q.e_child << 32);
      words[3] = (unsigned long long)(unsigned)q.e_pm | ((unsigned long long)(unsigned)q.last_dest << 32);
      for (int k = 0; k < q.n_fresh; ++k)
      {
        words[4 + k / 2] |= (unsigned long long)(unsigned)q.fresh_idx[k] << (32 * (k & 1));
        memcpy(&words[6 + k], &q.fresh_len[k], 8);
      }
      auto put = [&](int k, const Desc &d) { words[k] = d.base; words[k + 1] = (unsigned long long)d.bytes | ((unsigned long long)d.x << 32); };
      for (int o = 0; o < q.n_real_ops; ++o)
      {
        put(10 + o * 12, q.arg_ir[o].c1_data); put(12 + o * 12, q.arg_ir[o].c2_data);
        put(14 + o * 12, q.arg_ir[o].c1_scale); put(16 + o * 12, q.arg_ir[o].c2_scale);
        put(18 + o * 12, q.arg_xr[o].dst_data); put(20 + o * 12, q.arg_xr[o].dst_scale);
      }
      // kept until the answer is in: an evaluation nobody answers is launched the ordinary way (flush_and_wait)
      I->rt_ops = I->pending; I->rt_pm_idx = I->pm_idx; I->rt_pm_len = I->pm_len;
      resident_send(I, R, words, kBigWords);
      I->rt_epoch = I->clean_epoch;
      I->host_sum_n = dsum ? 1 : host_sum_n; I->host_sum_ns = 1;
      I->fenced_eval = true; // (nothing went onto the stream: it is as idle as it was found)
      if (fold_pm)
      {
        for (int m : I->pm_idx) I->pm_slot[m] = -1;
        I->pm_idx.clear();
        I->pm_len.clear();
      }
      I->pending.clear();
      std::fill(I->mat_in_queue.begin(), I->mat_in_queue.end(), 0);
      return 0;
    }
  }
  if (!big_try) big_release(I); // (what follows needs the wave slots the large-grid resident workgroups hold, if there are any)
  I->touched_call = true; // (everything below goes onto the stream)
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (I->prof)
  {
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, I->stream));
  }
  const unsigned long long hp1 = hp_now();
  rc = dispatch_shape(I, [&](auto s, auto cp) {
    constexpr int S_ = decltype(s)::value, CP_ = decltype(cp)::value;
    if constexpr (S_ == 4 && CP_ <= 4)
    {
      if (I->soa)
      { // lane-per-pattern kernel, instantiated on the exact category count
#ifdef PHYHIP_DIAG
        if ((I->ablate & 8) && I->C == 4 && I->nt_groups <= 2)
        { // PHYHIP_ABLATE=8: cycle stamps of one wave, printed to stderr (diagnostics; costs a sync)
          unsigned long long *&d_dbg = I->d_dbg;
          if (!d_dbg) HIPCHK(hipMalloc((void **)&d_dbg, 64 * 8 * 8));
          if (I->nt_groups == 2)
            hipLaunchKernelGGL((traverse_nt2_kernel<4, 2, true>), dim3(I->grid_nt2), dim3(64), 0, I->stream, q, d_irec, d_xrec, ro.pmats, ro.tip_codes, d_dbg);
          else
            hipLaunchKernelGGL((traverse_nt2_kernel<4, 1, true>), dim3(I->grid_nt2), dim3(64), 0, I->stream, q, d_irec, d_xrec, ro.pmats, ro.tip_codes, d_dbg);
          static int printed = 0;
          if (printed++ == 5)
          {
            unsigned long long h[64 * 8];
            HIPCHK(hipMemcpyAsync(h, d_dbg, sizeof h, hipMemcpyDeviceToHost, I->stream));
            HIPCHK(hipStreamSynchronize(I->stream));
            for (int k = 0; k < 64 && k < q.n_ops; ++k)
            {
              fprintf(stderr, "step %2d:", k);
              for (int i = 1; i < 7; ++i) fprintf(stderr, " %6lld", (long long)(h[k * 8 + i] - h[k * 8 + i - 1]));
              if (k + 1 < 64) fprintf(stderr, "  | next %6lld", (long long)(h[(k + 1) * 8] - h[k * 8 + 6]));
              fprintf(stderr, "  | load issue %6lld of segment 4", (long long)(h[k * 8 + 7] - h[k * 8 + 3]));
              fprintf(stderr, "\n");
            }
          }
          return 0;
        }
#endif
#define NT2LAUNCH(c_, g_, a_)                                                                                               \
  hipLaunchKernelGGL((traverse_nt2_kernel<c_, g_, false, a_>), dim3(I->grid_nt2), dim3(64), 0, I->stream, q, d_irec, d_xrec,  \
                     ro.pmats, ro.tip_codes, (unsigned long long *)nullptr);
#define NT2CASE(c_, g_)                                                                                                     \
  if (!q.recs_in_args && I->prefetch_dist == 1)                                                                             \
  {                                                                                                                         \
    hipLaunchKernelGGL((traverse_nt2_kernel<c_, g_, false, 0, 1>), dim3(I->grid_nt2), dim3(64), 0, I->stream, q, d_irec, d_xrec, \
                       ro.pmats, ro.tip_codes, (unsigned long long *)nullptr);                                              \
  }                                                                                                                         \
  else if (!q.recs_in_args) { NT2LAUNCH(c_, g_, 0) }                                                                        \
  else if (q.n_real_ops == 1) { NT2LAUNCH(c_, g_, 1) }                                                                      \
  else if (q.n_real_ops == 2) { NT2LAUNCH(c_, g_, 2) }                                                                      \
  else { NT2LAUNCH(c_, g_, 3) }                                                                                             \
  return 0;
        switch (I->C * 8 + I->nt_groups)
        {
          case 1 * 8 + 1: NT2CASE(1, 1)
          case 2 * 8 + 1: NT2CASE(2, 1)
          case 2 * 8 + 2: NT2CASE(2, 2)
          case 3 * 8 + 1: NT2CASE(3, 1)
          case 4 * 8 + 1: NT2CASE(4, 1)
          case 4 * 8 + 2: NT2CASE(4, 2)
          case 4 * 8 + 4: NT2CASE(4, 4)
          default: break;
        }
#undef NT2CASE
#undef NT2LAUNCH
      }
    }
    if constexpr (S_ == 4 && (CP_ == 8 || kDiag))
    { // first-generation lane = (pattern, category) pipeline: the production kernel for 5..8 categories
      if (!I->generic_nt)
      {
        if (I->prefetch_dist == 1)
        {
          hipLaunchKernelGGL((traverse_nt_kernel<CP_, 0, 1>), dim3(I->grid_nt), dim3(I->block_nt), 0, I->stream, q, d_irec, d_xrec, ro.pmats,
                             ro.tip_codes);
          return 0;
        }
#ifdef PHYHIP_DIAG
        if constexpr (CP_ == 4)
        {
          switch (I->ablate)
          {
#define ABLCASE(a) case a: hipLaunchKernelGGL((traverse_nt_kernel<CP_, a>), dim3(I->grid_nt), dim3(I->block_nt), 0, I->stream, q, d_irec, d_xrec, ro.pmats, ro.tip_codes); return 0;
            ABLCASE(1) ABLCASE(2) ABLCASE(3) ABLCASE(6) ABLCASE(7)
#undef ABLCASE
            default: break;
          }
        }
#endif
        hipLaunchKernelGGL((traverse_nt_kernel<CP_>), dim3(I->grid_nt), dim3(I->block_nt), 0, I->stream, q, d_irec, d_xrec, ro.pmats,
                           ro.tip_codes);
        return 0;
      }
    }
    if constexpr (S_ == 20 && CP_ <= 4)
    {
      if (I->perm)
      {
        const dim3 blk(64 * (I->aa_nw + 1));
#define AACASE(c_)                                                                                                          \
  case c_:                                                                                                                  \
    if (q.recs_in_args)                                                                                                     \
      hipLaunchKernelGGL((traverse_aa_kernel<c_, false, 0, true>), dim3(I->grid_aa), blk, 0, I->stream, q, d_irec, d_xrec,    \
                         (const double *)I->d_afrag, I->nmat, (const uint32_t *)I->d_tipmasks, (unsigned long long *)nullptr); \
    else                                                                                                                    \
    hipLaunchKernelGGL((traverse_aa_kernel<c_>), dim3(I->grid_aa), blk, 0, I->stream, q, d_irec, d_xrec,                      \
                       (const double *)I->d_afrag, I->nmat, (const uint32_t *)I->d_tipmasks, (unsigned long long *)nullptr); \
    return 0;
#ifdef PHYHIP_DIAG
        if (I->C == 4 && I->ablate >= 256)
        { // PHYHIP_ABLATE = 256 + bits: timing-only ablations of the 20-state kernel (results invalid)
#define AAABL(a_) case a_: hipLaunchKernelGGL((traverse_aa_kernel<4, false, a_>), dim3(I->grid_aa), blk, 0, I->stream, q, d_irec, d_xrec, (const double *)I->d_afrag, I->nmat, (const uint32_t *)I->d_tipmasks, (unsigned long long *)nullptr); return 0;
          switch (I->ablate - 256)
          {
            AAABL(1) AAABL(2) AAABL(4) AAABL(8) AAABL(9) AAABL(16) AAABL(18) AAABL(5) AAABL(13) AAABL(31) AAABL(27)
            default: break;
          }
#undef AAABL
        }
        if ((I->ablate & 8) && I->ablate < 256 && I->C == 4)
        { // PHYHIP_ABLATE=8: cycle stamps of one consumer wave, printed to stderr (diagnostics; costs a sync per launch)
          unsigned long long *&d_dbg = I->d_dbg;
          if (!d_dbg) HIPCHK(hipMalloc((void **)&d_dbg, 64 * 8 * 8));
          if (I->ablate & 128) // (stamps of the bare skeleton: every ablation on)
            hipLaunchKernelGGL((traverse_aa_kernel<4, true, 31>), dim3(I->grid_aa), blk, 0, I->stream, q, d_irec, d_xrec,
                               (const double *)I->d_afrag, I->nmat, (const uint32_t *)I->d_tipmasks, d_dbg);
          else
          hipLaunchKernelGGL((traverse_aa_kernel<4, true>), dim3(I->grid_aa), blk, 0, I->stream, q, d_irec, d_xrec,
                             (const double *)I->d_afrag, I->nmat, (const uint32_t *)I->d_tipmasks, d_dbg);
          static int printed = 0;
          if (printed++ == 5)
          {
            unsigned long long h[64 * 8];
            HIPCHK(hipMemcpyAsync(h, d_dbg, sizeof h, hipMemcpyDeviceToHost, I->stream));
            HIPCHK(hipStreamSynchronize(I->stream));
            for (int k = 0; k < 64 && k < q.n_ops; ++k)
            {
              fprintf(stderr, "step %2d:", k);
              for (int i = 1; i < 7; ++i) fprintf(stderr, " %6lld", (long long)(h[k * 8 + i] - h[k * 8 + i - 1]));
              if (k + 1 < 64) fprintf(stderr, "  | next %6lld", (long long)(h[(k + 1) * 8] - h[k * 8 + 6]));
              fprintf(stderr, "\n");
            }
          }
          return 0;
        }
#endif
        switch (I->C)
        {
          AACASE(1) AACASE(2) AACASE(3) AACASE(4)
          default: break;
        }
#undef AACASE
      }
    }
    hipLaunchKernelGGL((traverse_kernel<S_, CP_>), dim3(I->grid), dim3(256), 0, I->stream, q, ro.ops, ro.pmats, ro.tip_codes,
                       ro.code_masks);
    return 0;
  });
  if (rc) return rc;
  if (I->prof)
  {
    HIPCHK(hipEventRecord(e1, I->stream));
    I->prof_pairs.emplace_back(e0, e1);
    I->prof_updates += (double)n_ops * (double)I->P;
    // Minimum traffic of this launch if nothing but the kernel's own register forwarding saved a byte: every result is
    // written once; a child is read unless it is a tip (1 byte per pattern) or the result of one of the previous two
    // operations (forwarded in registers -- exactly the flags computed for the operation records above).
    {
      const double rec = (double)I->C * I->S * 8.0 + 4.0;
      double       rd = 0.0, wr = (double)n_ops * rec;
      for (int k = 0; k < n_ops; ++k)
      {
        const DevOp &o  = I->pending[k];
        const int    e1 = k >= 1 ? I->pending[k - 1].dest : -1;
        const int    e2 = (k >= 2 && fat && I->prefetch_dist == 2) ? I->pending[k - 2].dest : -1;
        for (int c : {o.c1, o.c2})
          rd += c < I->tips ? 1.0 : ((fat && (c == e1 || c == e2)) ? 0.0 : rec);
      }
      if (ee)
      { // root edge: both sides unless just produced, pattern weight in; per-pattern outputs out
        for (int c : {ee->parent, ee->child})
          rd += c < I->tips ? 1.0 : ((fat && n_ops > 0 && c == I->pending[n_ops - 1].dest) ? 0.0 : rec);
        rd += 8.0;
        wr += 4.0 + (I->want_site_outputs ? 16.0 + 8.0 * I->C : 0.0);
      }
      I->prof_rd_bytes += rd * (double)I->P;
      I->prof_wr_bytes += wr * (double)I->P;
    }
  }
  if (kDiag) { const unsigned long long hp2 = hp_now(); g_hp.prep += hp1 - hp0; g_hp.launch += hp2 - hp1; ++g_hp.n_launch; }
  HIPCHK(hipGetLastError());
  I->host_sum_n = host_sum_n; I->host_sum_ns = 1;
  if (ee && !ee->eigen && !fused_sum && !host_sum_n && !I->class_axis) // (class axis: the combination kernel follows, no sum here)
  {
    double *out = ee->dev_out ? ee->dev_out : I->d_result;
    const int nsum = I->soa ? I->grid_nt2 : (I->perm ? I->grid_aa : (fat ? I->grid_nt : I->grid));
    hipLaunchKernelGGL(final_reduce_kernel, dim3(1), dim3(256), 0, I->stream, (const double *)I->d_block, nsum, 1,
                       nsum, out, ee->to_host ? I->h_result : (double *)nullptr, I->d_warn, I->h_warn,
                       ee->to_host ? ++I->seq : 0ull, ee->warn_out);
    HIPCHK(hipGetLastError());
  }
  if (fold_pm)
  {
    for (int m : I->pm_idx) I->pm_slot[m] = -1;
    I->pm_idx.clear();
    I->pm_len.clear();
  }
  I->pending.clear();
  std::fill(I->mat_in_queue.begin(), I->mat_in_queue.end(), 0);
  // an evaluation the large-grid resident workgroups would have taken, had the stream been known to be idle: say when it is
  if (big_fit && host_sum_n > 0 && (rc = stamp_stream(I))) return rc;
  return 0;
}

int flush(Instance *I, const EdgeEval *ee)
{
  const int rc = flush_impl(I, ee);
  if (rc)
  { // a failed launch leaves no half-queued state behind: the operations are dropped (the caller gets the error and
    // PhyML's glue exits on it), queued matrix rebuilds stay queued, no device slot claims a list it never received
    I->pending.clear();
    std::fill(I->mat_in_queue.begin(), I->mat_in_queue.end(), 0);
  }
  return rc;
}

int flush_sync(Instance *I)
{
  int rc = flush(I, nullptr);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(I->stream));
  return 0;
}

int check_partial_index(const Instance *I, int idx, bool allow_tip)
{
  if (idx < 0 || idx >= I->nbuf || (!allow_tip && idx < I->tips))
    return fail(PHYHIP_ERROR_OUT_OF_RANGE, "partials buffer index %d out of range [%d,%d)", idx, allow_tip ? 0 : I->tips, I->nbuf);
  return 0;
}

// Wait until the final reduction has published evaluation `seq` in host-mapped memory.  Spinning on the
// sequence word avoids the stream-synchronise wake-up latency; after 2 ms of spinning (elapsed time, checked every
// 256 polls) fall back to it: evaluations of very large alignments take milliseconds and must not burn a core.
// The host's side of the final sum on large grids: poll the {sum, tag} records the workgroups posted (they arrive roughly in
// launch order), then add them exactly as final_reduce_kernel does -- 256 strided accumulators, then a binary tree -- so
// that the value does not depend on which path produced it.
int wait_host_sum(Instance *I)
{
  const int                n   = I->host_sum_n * I->host_sum_ns, per = I->host_sum_n;
  const unsigned long long tag = I->seq;
  volatile HostBlock      *hb  = I->h_blocks;
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  bool   synced = false;
  // final_reduce_kernel's order -- 256 strided accumulators per sum, then a binary tree -- with the records taken as they
  // arrive, front to back: accumulator t receives records t, t + 256, ... in that order either way, and one sequential pass
  // over the records costs a fraction of 256 strided ones (thousands of records per evaluation on large grids)
  double acc[2][256];
  for (int k = 0; k < I->host_sum_ns; ++k)
    for (int t = 0; t < 256; ++t) acc[k][t] = 0.0;
  for (int i = 0, k = 0, j = 0; i < n; ++i)
  {
    long it = 0;
    while (hb[i].tag != tag)
    {
      __builtin_ia32_pause();
      if ((++it & 255) == 0 && !synced)
      {
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const long waited = (t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec);
        if (I->r_inflight)
        { // no answer from the resident workgroups (they may have left just before the command arrived): the caller
          // retires them and launches the evaluation the ordinary way
          if (waited > 30000L && resident_gone(*I->r_inflight)) return kResidentSilent; // they left as the command arrived
          if (waited > 100000L && ns_since(I->r_inflight->t_launch) > 20e6) return kResidentSilent; // (a first launch loads code: ms)
          continue;
        }
        if (waited > 2000000L || !I->spin_wait)
        {
          HIPCHK(hipStreamSynchronize(I->stream));
          synced = true;
          it = 0;
        }
      }
      else if (synced && it > 100000000L)
        return fail(PHYHIP_ERROR_GENERAL, "evaluation %llu finished without posting block sum %d", tag, i);
    }
    acc[k][j & 255] += hb[i].sum; // (the record is ONE 16-byte store of the device: the sum is there when the tag is)
    if (++j == per) { j = 0; ++k; }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  for (int k = 0; k < I->host_sum_ns; ++k)
  {
    for (int off = 128; off > 0; off >>= 1)
      for (int t = 0; t < off; ++t) acc[k][t] += acc[k][t + off];
    I->h_result[k] = acc[k][0];
  }
  I->host_sum_n    = 0;
  if (I->r_inflight) I->r_inflight->ns_wait += ns_since(I->r_inflight->t_cmd);
  I->r_inflight    = nullptr;
  I->warn_current  = true;
  *reinterpret_cast<volatile unsigned long long *>(I->h_result + 2) = tag;
  return 0;
}

int wait_result_impl(Instance *I);
int wait_result(Instance *I)
{
  const unsigned long long t0 = hp_now();
  const int rc = wait_result_impl(I);
  if (kDiag) { const unsigned long long t1 = hp_now(); g_hp.wait += t1 - t0; ++g_hp.n_wait; if (!g_hp.t_first) g_hp.t_first = t0; g_hp.t_last = t1; }
  return rc;
}
int wait_result_impl(Instance *I)
{
  if (I->host_sum_n > 0) return wait_host_sum(I);
  if (I->spin_wait)
  {
    volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(I->h_result + 2);
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (long it = 0;; ++it)
    {
      if (*flag == I->seq)
      {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        I->warn_current = true;
        return 0;
      }
      __builtin_ia32_pause();
      if ((it & 255) == 255)
      {
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 2000000L) break;
      }
    }
  }
  HIPCHK(hipStreamSynchronize(I->stream));
  if (*reinterpret_cast<volatile unsigned long long *>(I->h_result + 2) != I->seq)
  { // the stream drained without the hand-over (a faulted launch): do not return a stale scalar, re-arm the ticket counter
    (void)hipMemsetAsync(I->d_tickets, 0, sizeof(unsigned) * (1 + kTicketGroups), I->stream);
    return fail(PHYHIP_ERROR_GENERAL, "evaluation %llu finished without handing its result over", I->seq);
  }
  I->warn_current = true;
  return 0;
}

int collect_profile(Instance *I)
{
  for (auto &pr : I->prof_pairs)
  {
    HIPCHK(hipEventSynchronize(pr.second));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, pr.first, pr.second));
    I->prof_ms += ms;
    I->prof_n += 1;
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  I->prof_pairs.clear();
  for (auto &pr : I->prof_aux)
  {
    HIPCHK(hipEventSynchronize(pr.b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, pr.a, pr.b));
    I->prof_aux_ms[pr.kind] += ms;
    I->prof_aux_n[pr.kind] += 1;
    (void)hipEventDestroy(pr.a);
    (void)hipEventDestroy(pr.b);
  }
  I->prof_aux.clear();
  return 0;
}

// HIP events around one launch of an eigen-basis kernel while the instance is being profiled (bench.py's K3 / K4 lines)
struct AuxProf
{
  Instance  *I;
  int        kind;
  hipEvent_t a = nullptr, b = nullptr;
  AuxProf(Instance *I_, int kind_) : I(I_), kind(kind_)
  {
    if (!I->prof) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
    (void)hipEventRecord(a, I->stream);
  }
  ~AuxProf()
  {
    if (!a) return;
    (void)hipEventRecord(b, I->stream);
    I->prof_aux.push_back({a, b, kind});
  }
};

// A combination / dLk kernel's sums go to the host as posted records (host-side final sum) when the instance allows it
void host_sum_finish(Instance *I, FinishParams &f, int grid, int ns)
{
  if (!I->host_sum || (size_t)grid * ns > I->h_blocks_cap) return; // (keeps the ticket path set up by the caller)
  f.host_blocks = I->h_blocks; f.host_tag = f.seq; f.stride = grid; f.warn = I->h_warn;
  *I->h_warn     = 0;
  I->host_sum_n  = grid; I->host_sum_ns = ns;
}

// the +I share of a mixture evaluation (phyhip_set_mixture_invariant_sites) into the combination kernel's parameters
template <typename Q> void fill_mixture_invariant(const Instance *I, Q &q)
{
  q.invar_model = I->mix_invar_model; q.pinvar = I->mix_pinvar; q.invar = I->d_invar;
  for (int s = 0; s < 20; ++s) q.pi_inv[s] = I->mix_pi_inv[s];
}

// Where a mixture evaluation's sums go: to the host (mo == nullptr: host-side final sum or the ticket path, then the caller
// waits), or -- one shard of a sharded evaluation -- into device memory next to the warning flag, nobody waiting (the
// collective follows, phyhip_shard.hpp)
struct MixOut
{
  double *dev_out, *warn_out;
};
static void mix_finish_setup(Instance *I0, FinishParams &fin, int grid, int ns, const MixOut *mo)
{
  fin.block_sums = I0->d_block; fin.stride = grid; fin.warn = I0->d_warn; fin.tickets = I0->d_tickets; fin.warn_host = I0->h_warn;
  if (mo)
  {
    fin.result = mo->dev_out; fin.result_host = nullptr; fin.seq = 0; fin.warn_out = mo->warn_out;
    return;
  }
  fin.result = I0->d_result; fin.result_host = I0->h_result; fin.seq = ++I0->seq;
  host_sum_finish(I0, fin, grid, ns);
}

#include "phyhip_shard.hpp"

} // namespace

extern "C" {

const char *phyhip_get_last_error(void) { return g_err.c_str(); }

// Frees everything an instance owns (also a partially built one: every pointer starts as nullptr).
static void release_instance(Instance *I)
{
  if (getenv("PHYHIP_RESIDENT_STATS"))
    for (const Resident *R : {&I->rd, &I->rt, &I->rb})
      if (R->n_cmd || R->n_busy)
        fprintf(stderr, "resident %s: %llu commands, %llu launches, %llu unanswered, %llu evaluations launched because the stream was busy; "
                        "%.2f us from command to answer\n",
                R == &I->rd ? "dLk evaluator" : (R == &I->rt ? "short-launch evaluator" : "large-grid evaluator"), R->n_cmd, R->n_launch,
                R->n_silent, R->n_busy, R->n_cmd ? R->ns_wait * 1e-3 / (double)R->n_cmd : 0.0);
  resident_free(I->rd);
  resident_free(I->rt);
  big_release(I);
  const bool big_used = I->rb.n_cmd > 0;
  resident_free(I->rb);
  if (I->d_big_stamps && big_used)
  { // where the last command's time went, per workgroup, relative to workgroup 0 seeing it (wall-clock ticks of 10 ns)
    std::vector<unsigned long long> h((size_t)8 * I->big_wgs);
    if (hipMemcpy(h.data(), I->d_big_stamps, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
    {
      const char *names[6] = {"command seen", "after the barrier", "wave 0 through", "all waves through", "ticket drawn", "final sum posted"};
      const unsigned long long t0 = h[0];
      for (int k = 0; k < 6; ++k)
      {
        double mn = 1e30, mx = -1e30, sum = 0.0; int n = 0;
        for (int w = 0; w < I->big_wgs; ++w)
        {
          const unsigned long long v = h[(size_t)w * 8 + k];
          if (!v || v < t0) continue;
          const double d = (double)(v - t0) * 1e6 / ((double)(I->wall_khz > 0 ? I->wall_khz : 100000) * 1e3);
          mn = std::min(mn, d); mx = std::max(mx, d); sum += d; ++n;
        }
        if (n) fprintf(stderr, "  big resident, last command: %-18s min %7.2f  mean %7.2f  max %7.2f us after workgroup 0 saw it (%d workgroups)\n",
                       names[k], mn, sum / n, mx, n);
      }
    }
  }
  if (I->stream) (void)hipStreamSynchronize(I->stream);
  void *ptrs[] = {I->d_partials, I->d_scales, I->d_tipcodes, I->d_masks, I->d_pmats, I->d_wght, I->d_invar, I->d_model,
                  I->d_site_lnl, I->d_site_lk, I->d_site_cat, I->d_fact, I->d_dot, I->d_block, I->d_result, I->d_warn, I->d_ops,
                  I->d_pmscratch, I->d_afrag, I->d_tickets, I->d_mixexpl, I->d_dbg, I->d_tipmasks, I->d_tile_sums, I->d_big_tickets, I->d_big_stamps, I->d_big_recs};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (I->h_result) (void)hipHostFree(I->h_result);
  if (I->h_warn) (void)hipHostFree(I->h_warn);
  if (I->h_blocks) (void)hipHostFree(I->h_blocks);
  if (I->ev_sync) (void)hipEventDestroy(I->ev_sync);
  if (I->ev_big) (void)hipEventDestroy(I->ev_big);
  I->ring.destroy();
  if (I->own_stream && I->stream) (void)hipStreamDestroy(I->stream);
  delete I;
}

static int build_instance(Instance *I, const hipDeviceProp_t &prop);

int phyhip_create_instance(int tipCount, int partialsBufferCount, int compactBufferCount, int stateCount,
                           int patternCount, int eigenBufferCount, int matrixBufferCount, int categoryCount,
                           int scaleBufferCount, const int *resourceList, int resourceCount, long preferenceFlags,
                           long requirementFlags, phyhip_instance_details *returnInfo)
{
  (void)compactBufferCount; (void)eigenBufferCount; (void)scaleBufferCount; (void)preferenceFlags;
  if (tipCount < 2 || partialsBufferCount <= tipCount || patternCount < 1 || matrixBufferCount < 1 || categoryCount < 1)
    return fail(PHYHIP_ERROR_OUT_OF_RANGE, "bad instance dimensions");
  if (stateCount != 4 && stateCount != 20)
    return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "stateCount %d: only 4 (nt) and 20 (aa) are built", stateCount);
  if (categoryCount > 8) return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "categoryCount %d > 8", categoryCount);
  if ((double)patternCount * categoryCount * stateCount * 8.0 >= 2147483648.0)
    return fail(PHYHIP_ERROR_OUT_OF_RANGE, "one partials buffer must stay below 2 GiB (shard the patterns across devices)");
  if ((double)matrixBufferCount * categoryCount * stateCount * stateCount * 8.0 >= 2147483648.0)
    return fail(PHYHIP_ERROR_OUT_OF_RANGE, "transition-matrix table must stay below 2 GiB");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(PHYHIP_ERROR_NO_RESOURCE, "no HIP device visible: libphyhip has no CPU fallback");
  if (resourceList && (resourceCount > 1 || (resourceCount == 1 && (requirementFlags & PHYHIP_FLAG_SHARDED))))
  { // sharded instance: one per-device instance per entry of the resource list + the RCCL communicators
    for (int g = 0; g < resourceCount; ++g)
      if (resourceList[g] < 0 || resourceList[g] >= ndev)
        return fail(PHYHIP_ERROR_NO_RESOURCE, "device %d not present (%d visible)", resourceList[g], ndev);
    return create_group(tipCount, partialsBufferCount, stateCount, patternCount, matrixBufferCount, categoryCount, resourceList,
                        resourceCount, returnInfo, requirementFlags & (PHYHIP_FLAG_CLASS_AXIS | PHYHIP_FLAG_GENERIC_LOOP));
  }
  int dev = 0;
  if (resourceList && resourceCount > 0) dev = resourceList[0];
  else if (const char *e = getenv("PHYHIP_DEVICE")) dev = atoi(e);
  if (dev < 0 || dev >= ndev) return fail(PHYHIP_ERROR_NO_RESOURCE, "device %d not present (%d visible)", dev, ndev);
  HIPCHK(hipSetDevice(dev));
  g_cur_dev = dev;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, dev));

  const bool class_axis = (requirementFlags & PHYHIP_FLAG_CLASS_AXIS) != 0;
  if (class_axis && !((stateCount == 20 && categoryCount <= 4) || (stateCount == 4 && (categoryCount == 1 || categoryCount == 2 || categoryCount == 4))))
    return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "the class axis is built for 20 states x up to 4 classes and 4 states x 1, 2 or 4 classes "
                                                "(one instance per class otherwise)");
  const bool generic_loop = (requirementFlags & PHYHIP_FLAG_GENERIC_LOOP) != 0;
  if (generic_loop && class_axis)
    return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "the generic loop (PHYHIP_FLAG_GENERIC_LOOP) is built for plain instances, not for the class axis");
  Instance *I = new Instance();
  I->generic_loop = generic_loop;
  I->class_axis = class_axis; I->NE = class_axis ? categoryCount : 1;
  I->dev = dev; I->tips = tipCount; I->nbuf = partialsBufferCount; I->S = stateCount; I->C = categoryCount;
  I->CP = next_pow2(categoryCount); I->P = patternCount; I->nmat = matrixBufferCount;
  {
    const int rc = build_instance(I, prop);
    if (rc < 0)
    { // a failed allocation half-way must not leak what was allocated before it
      release_instance(I);
      return rc;
    }
  }

  if (returnInfo)
  {
    memset(returnInfo, 0, sizeof *returnInfo);
    returnInfo->resourceNumber = dev;
    snprintf(returnInfo->resourceName, sizeof returnInfo->resourceName, "%s", prop.name);
    snprintf(returnInfo->implName, sizeof returnInfo->implName, "phyhip-%s", prop.gcnArchName);
    returnInfo->computeUnits   = prop.multiProcessorCount;
    returnInfo->globalMemBytes = (long long)prop.totalGlobalMem;
  }
  return InstanceTable::add(I);
}

// Device memory, staging, launch geometry and environment switches of a new instance.
static int build_instance(Instance *I, const hipDeviceProp_t &prop)
{
  int rc = 0;
  (void)prop;
  HIPCHK(hipStreamCreateWithFlags(&I->stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&I->ev_sync, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&I->ev_big, hipEventDisableTiming));

  I->perm = (I->S == 20) && (I->C <= 4) && !I->generic_loop &&
            (I->class_axis || !(diag_env("PHYHIP_GENERIC_AA") && atoi(diag_env("PHYHIP_GENERIC_AA"))));
  I->soa  = (I->S == 4) && (I->C <= 4) && !I->generic_loop &&
            (I->class_axis || (!(diag_env("PHYHIP_NT_SOA") && atoi(diag_env("PHYHIP_NT_SOA")) == 0) &&
                               !(diag_env("PHYHIP_GENERIC_NT") && atoi(diag_env("PHYHIP_GENERIC_NT")))));
  I->Ppad = I->perm ? ((I->P + 15) / 16) * 16 : (I->soa ? ((I->P + 63) / 64) * 64 : I->P);
  // category groups of the lane-per-pattern kernel (phyhip_nt2.hpp): split a pattern over 2 lanes while the
  // alignment is too short to give every SIMD two waves of 64 patterns
  I->nt_groups = 1;
  // measured (us per traversal, G=2 / G=1): 50 k 195 / 206, 75 k 301 / 363, 125 k 444 / 456, 250 k 837 / 848, 1 M 3293 / 3246
  // (round 3, non-temporal result stores: 50 k 165 / 198, 125 k 386 / 377, 1 M 3188 / 3186 -- the crossover moved to ~100 k)
  if (I->soa && I->C % 2 == 0 && I->Ppad / 64 <= 1600) I->nt_groups = 2;
  if (const char *e = diag_env("PHYHIP_NT_GROUPS"))
  {
    const int g = atoi(e);
    if (g >= 1 && g <= 4 && I->C % g == 0 && 64 % g == 0 && !(I->C == 2 && g == 4)) I->nt_groups = g;
  }
  if (I->soa && I->class_axis) I->nt_groups = I->C; // one class per lane: scaling and evaluation are per class
  I->grid_nt2 = (int)(I->Ppad / (64 / I->nt_groups));
  const size_t n_int = (size_t)(I->nbuf - I->tips);
  const size_t be    = buf_elems(I);
  HIPCHK(hipMalloc((void **)&I->d_partials, n_int * be * sizeof(double)));
  HIPCHK(hipMemset(I->d_partials, 0, n_int * be * sizeof(double)));
  HIPCHK(hipMalloc((void **)&I->d_scales, n_int * scale_elems(I) * sizeof(int)));
  HIPCHK(hipMemset(I->d_scales, 0, n_int * scale_elems(I) * sizeof(int)));
  if (I->perm)
  {
    const size_t fb = (size_t)I->nmat * kAaMat * sizeof(double);
    HIPCHK(hipMalloc((void **)&I->d_afrag, fb));
    HIPCHK(hipMemset(I->d_afrag, 0, fb));
    // one consumer wave per wave-tile (16 / aa_cb(C) patterns x all categories) + one loader wave per workgroup; enough
    // workgroups to give every CU one before any gets a second (the LDS ring allows one workgroup per CU at a time)
    const long long ntiles = I->Ppad / (16 / aa_cb(I->C));
    const long long cus    = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    I->aa_nw = (int)std::min<long long>(kAaMaxCons, std::max<long long>(1, (ntiles + cus - 1) / cus));
    if (const char *e = diag_env("PHYHIP_AA_NW")) { const int v = atoi(e); if (v >= 1 && v <= kAaMaxCons) I->aa_nw = v; }
    I->grid_aa = (int)((ntiles + I->aa_nw - 1) / I->aa_nw);
  }
  HIPCHK(hipMalloc((void **)&I->d_tipcodes, (size_t)I->tips * I->Ppad));
  HIPCHK(hipMemset(I->d_tipcodes, 0, (size_t)I->tips * I->Ppad));
  if (I->perm)
  {
    HIPCHK(hipMalloc((void **)&I->d_tipmasks, (size_t)I->tips * I->Ppad * sizeof(uint32_t)));
    HIPCHK(hipMemset(I->d_tipmasks, 0, (size_t)I->tips * I->Ppad * sizeof(uint32_t)));
  }
  HIPCHK(hipMalloc((void **)&I->d_masks, 256 * sizeof(uint32_t)));
  HIPCHK(hipMalloc((void **)&I->d_pmats, (size_t)I->nmat * I->C * I->S * I->S * sizeof(double)));
  HIPCHK(hipMemset(I->d_pmats, 0, (size_t)I->nmat * I->C * I->S * I->S * sizeof(double)));
  HIPCHK(hipMalloc((void **)&I->d_wght, I->P * sizeof(double)));
  HIPCHK(hipMalloc((void **)&I->d_invar, I->P * sizeof(short)));
  HIPCHK(hipMemset(I->d_invar, 0xff, I->P * sizeof(short)));
  {
    std::vector<double> ones((size_t)I->P, 1.0);
    HIPCHK(hipMemcpy(I->d_wght, ones.data(), I->P * sizeof(double), hipMemcpyHostToDevice));
  }
  const size_t NE = (size_t)I->NE; // pi | catw | catr | eval | evec | ivec, the eigen data once per eigen system
  const size_t model_doubles = NE * 2 * I->S + 2 * I->C + NE * 2 * (size_t)I->S * I->S;
  HIPCHK(hipMalloc((void **)&I->d_model, model_doubles * sizeof(double)));
  HIPCHK(hipMemset(I->d_model, 0, model_doubles * sizeof(double)));
  I->d_pi = I->d_model; I->d_catw = I->d_pi + NE * I->S; I->d_catr = I->d_catw + I->C; I->d_eval = I->d_catr + I->C;
  I->d_evec = I->d_eval + NE * I->S; I->d_ivec = I->d_evec + NE * (size_t)I->S * I->S;
  I->h_rates.assign(I->C, 1.0);
  I->h_model.assign(model_doubles, 0.0);
  I->h_model_set.assign(model_doubles, 0);
  I->h_eval.assign(NE * I->S, 0.0);
  I->h_evec.assign(NE * (size_t)I->S * I->S, 0.0); I->h_ivec.assign(NE * (size_t)I->S * I->S, 0.0);
  HIPCHK(hipMalloc((void **)&I->d_site_lnl, I->P * sizeof(double)));
  HIPCHK(hipMalloc((void **)&I->d_site_lk, I->P * sizeof(double)));
  HIPCHK(hipMalloc((void **)&I->d_site_cat, (size_t)I->P * I->C * sizeof(double)));
  HIPCHK(hipMalloc((void **)&I->d_fact, I->P * NE * sizeof(int))); // class axis: [class][pattern]
  HIPCHK(hipMemset(I->d_fact, 0, I->P * NE * sizeof(int)));
  HIPCHK(hipMalloc((void **)&I->d_dot, be * sizeof(double)));
  HIPCHK(hipMemset(I->d_dot, 0, be * sizeof(double)));

  I->grid = (int)(((long long)I->P * I->CP + 255) / 256);
  // large-grid resident evaluator (phyhip_big.hpp): a workgroup of big_nw waves per CU at the traversal kernel's register
  // budget (two waves per SIMD with two lanes per pattern, one otherwise); dLk in up to 2 048 one-wave virtual blocks
  I->cus     = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  I->n_vdlk  = I->grid_nt2; // (a dLk tile = the patterns of a traversal tile: dlk_tile)
  I->big_nw  = std::max(1, big_waves_per_workgroup(I->C, I->nt_groups));
  // (as many workgroups as there are tiles, up to one per CU: the tiles spread over the CUs first -- wave w of workgroup b takes
  // tile w * workgroups + b -- and fill a CU's wave slots only then)
  I->big_wgs = std::max(1, std::min(I->cus, I->grid_nt2));
  if (const char *e = diag_env("PHYHIP_BIG_DEVICE_SUM")) I->big_device_sum = atoi(e);
  if (const char *e = diag_env("PHYHIP_BIG_GROUP_SUM")) I->big_group_sum = atoi(e) != 0;
  // The pipelined nucleotide kernel is instruction-issue bound per CU, so CU-level balance matters more than
  // workgroup size: one-wave workgroups let the dispatcher spread e.g. 3125 waves as 12-13 per CU instead of
  // 3-4 four-wave groups (measured: 100 taxa x 50 000 patterns, 288 -> 25x us).
  if (const char *e = diag_env("PHYHIP_NT2_DIST")) I->nt2_dist = atoi(e) == 1 ? 1 : 2;
  I->block_nt = 64;
  if (const char *e = diag_env("PHYHIP_BLOCK")) { int b = atoi(e); if (b == 64 || b == 128 || b == 256) I->block_nt = b; }
  I->grid_nt = (int)(((long long)I->P * I->CP + I->block_nt - 1) / I->block_nt);
  {
    // distance-2 prefetch needs 168 VGPRs (3 waves/SIMD), distance-1 fits 4 waves/SIMD: prefer the deeper
    // pipeline unless that would leave a nearly empty second residency round
    const long long waves = ((long long)I->P * I->CP + 63) / 64, simds = 4LL * prop.multiProcessorCount;
    if (!diag_env("PHYHIP_DIST") && waves > 3 * simds && waves <= 4 * simds) I->prefetch_dist = 1;
    if (I->soa) I->prefetch_dist = I->nt2_dist;
    if (I->perm) I->prefetch_dist = 1; // the 20-state kernels forward only the previous result
  }
  HIPCHK(hipMalloc((void **)&I->d_block,
                   (size_t)2 * std::max(std::max(I->grid, I->grid_nt), std::max(I->grid_aa, I->grid_nt2)) * sizeof(double)));
  HIPCHK(hipMalloc((void **)&I->d_result, 2 * sizeof(double)));
  HIPCHK(hipHostMalloc((void **)&I->h_result, 4 * sizeof(double), hipHostMallocMapped));
  memset(I->h_result, 0, 4 * sizeof(double));
  if (const char *e = diag_env("PHYHIP_SPIN")) I->spin_wait = atoi(e) != 0;
  HIPCHK(hipMalloc((void **)&I->d_warn, sizeof(int)));
  HIPCHK(hipMemset(I->d_warn, 0, sizeof(int)));
  HIPCHK(hipHostMalloc((void **)&I->h_warn, sizeof(int), hipHostMallocMapped));
  *I->h_warn = 0;
  {
    // traversal grids (one sum), dLk grids (two sums, <= 2048 workgroups), mixture combination grids ((P + 255) / 256)
    const size_t nb = std::max<size_t>((size_t)std::max(std::max(I->grid, I->grid_nt), std::max(I->grid_aa, 2 * I->grid_nt2)), 2 * 2048);
    HIPCHK(hipHostMalloc((void **)&I->h_blocks, nb * sizeof(HostBlock), hipHostMallocMapped));
    memset(I->h_blocks, 0, nb * sizeof(HostBlock));
    I->h_blocks_cap = nb;
  }
  if (const char *e = getenv("PHYHIP_HOST_SUM")) I->host_sum = atoi(e) != 0;
  if (const char *e = getenv("PHYHIP_RESIDENT")) I->resident = atoi(e) != 0;
  if (const char *e = getenv("PHYHIP_RESIDENT_IDLE_US")) I->resident_idle_us = atof(e);
  if (const char *e = diag_env("PHYHIP_RESIDENT_DIRECT")) I->resident_direct = atoi(e);

  I->pm_scratch_cap = std::min(std::max(I->nmat, 16), 4096);
  HIPCHK(hipMalloc(&I->d_pmscratch, (size_t)I->pm_scratch_cap * 16 + 64));
  I->ops_cap = 2 * I->nbuf + 8;
  I->ops_slot_bytes = (size_t)(I->ops_cap + 2) * (sizeof(IssueRec) + sizeof(ExecRec));
  HIPCHK(hipMalloc((void **)&I->d_ops, (size_t)I->ops_slots * I->ops_slot_bytes));
  size_t chunk = std::max<size_t>(64 * 1024, std::max(I->ops_slot_bytes,
                                                       (size_t)I->C * I->S * I->S * sizeof(double) * 4));
  rc = I->ring.init(chunk);
  if (rc) return rc;
  I->ring.before_rotate = [I]() { return I->up_idx.empty() ? 0 : flush_uploads(I); };
  I->mat_in_queue.assign(I->nmat, 0);
  I->up_slot.assign(I->nmat, -1);
  I->pm_slot.assign(I->nmat, -1);
  I->slot_ops.assign(I->ops_slots, std::vector<DevOp>());
  I->slot_kind.assign(I->ops_slots, -1);
  if (const char *e = diag_env("PHYHIP_GENERIC_NT")) I->generic_nt = atoi(e) != 0;
  if (I->generic_loop) I->generic_nt = true;
#ifdef PHYHIP_DIAG
  if (const char *e = diag_env("PHYHIP_ABLATE")) I->ablate = atoi(e);
  if (const char *e = diag_env("PHYHIP_NOLOADS")) I->no_loads = atoi(e) != 0;
#endif
  if (const char *e = diag_env("PHYHIP_EAGER_PMAT")) I->eager_pmats = atoi(e) != 0;
  if (const char *e = diag_env("PHYHIP_ARG_UPLOADS")) I->arg_uploads = atoi(e) != 0;
  if (const char *e = diag_env("PHYHIP_FUSE_EIGEN")) I->fuse_eigen = atoi(e) != 0;
  if (const char *e = diag_env("PHYHIP_ARGS_RECS")) I->args_recs = atoi(e) != 0;
  if (const char *e = diag_env("PHYHIP_SPLIT_REDUCE")) // see fuse_reduce()
  {
    I->split_reduce        = atoi(e) != 0;
    I->split_reduce_forced = true;
  }
  if (const char *e = diag_env("PHYHIP_PM_COPY")) I->pm_copy = atoi(e) != 0;
  if (const char *e = diag_env("PHYHIP_FOLD_PMATS")) I->fold_pmats = atoi(e) != 0;
  HIPCHK(hipMalloc((void **)&I->d_mixexpl, sizeof(double) * kMaxMixClasses * 2 * 20));
  HIPCHK(hipMalloc((void **)&I->d_tickets, sizeof(unsigned) * (1 + kTicketGroups)));
  HIPCHK(hipMemset(I->d_tickets, 0, sizeof(unsigned) * (1 + kTicketGroups)));
  if (const char *e = diag_env("PHYHIP_DIST"))
    if (!I->perm && !I->soa) I->prefetch_dist = atoi(e) == 1 ? 1 : 2;

  // codes 0..S-1 are the single states
  for (int s = 0; s < I->S; ++s)
  {
    I->masks.push_back(1u << s);
    I->mask_code[1u << s] = s;
  }
  if (I->S > 8)
  { // ... and "every state" (a gap, a hidden character of the leave-one-out loop, phyhip_set_tip_partials_at_pattern): present
    // from the start, so that the first one costs no upload of the table (a synchronisation of the stream)
    const uint32_t all = (1u << I->S) - 1u;
    I->mask_code[all] = (int)I->masks.size();
    I->masks.push_back(all);
  }
  I->masks_dirty = true;

  return 0;
}

int phyhip_finalize_instance(int instance)
{
  if (kDiag && getenv("PHYHIP_HOSTPROF") && g_hp.n_wait)
  {
    struct timespec c0, c1;
    clock_gettime(CLOCK_MONOTONIC, &c0);
    const unsigned long long r0 = hp_now();
    do clock_gettime(CLOCK_MONOTONIC, &c1); while ((c1.tv_sec - c0.tv_sec) * 1000000000L + (c1.tv_nsec - c0.tv_nsec) < 5000000L);
    fprintf(stderr, "hostprof: %.3f cycles per ns\n", (double)(hp_now() - r0) / 5e6);
    fprintf(stderr, "hostprof (cycles): launches %llu waits %llu | per launch: prep %.0f launch %.0f | per wait: %.0f | span per wait %.0f\n",
            g_hp.n_launch, g_hp.n_wait, (double)g_hp.prep / g_hp.n_launch, (double)g_hp.launch / g_hp.n_launch,
            (double)g_hp.wait / g_hp.n_wait, (double)(g_hp.t_last - g_hp.t_first) / g_hp.n_wait);
    g_hp = HostProf();
  }
  if (Group *G = get_group(instance))
  {
    forget_group(instance);
    release_group(G);
    return PHYHIP_SUCCESS;
  }
  GET_INST(I, instance);
  (void)hipStreamSynchronize(I->stream);
  collect_profile(I);
  (void)InstanceTable::remove(instance);
  if (I->co) release_collective(I->co);
  release_instance(I);
  return PHYHIP_SUCCESS;
}

// ---- inputs --------------------------------------------------------------------------------------

static int set_tip_codes(Instance *I, int tip, const std::vector<uint8_t> &codes)
{
  int rc = flush_sync(I);
  if (rc) return rc;
  HIPCHK(hipMemcpy(I->d_tipcodes + (size_t)tip * I->Ppad, codes.data(), (size_t)I->P, hipMemcpyHostToDevice));
  if (I->d_tipmasks)
  {
    std::vector<uint32_t> m((size_t)I->P);
    for (long long p = 0; p < I->P; ++p) m[(size_t)p] = I->masks[codes[(size_t)p]];
    HIPCHK(hipMemcpy(I->d_tipmasks + (size_t)tip * I->Ppad, m.data(), (size_t)I->P * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  return upload_masks(I);
}

static int code_for_mask(Instance *I, uint32_t m, int *code)
{
  // (no state allowed: the reference builds no such tip vector, src/lk.c:26-161, and the fragment-major 20-state kernel reads a
  // zero mask word as padding)
  if (m == 0) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "a tip vector with no state allowed (all zeros)");
  auto it = I->mask_code.find(m);
  if (it != I->mask_code.end())
  {
    *code = it->second;
    return 0;
  }
  if (I->masks.size() >= 256) return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "more than 256 distinct tip state sets");
  *code = (int)I->masks.size();
  I->masks.push_back(m);
  I->mask_code[m] = *code;
  I->masks_dirty  = true;
  return 0;
}

int phyhip_set_tip_partials(int instance, int tipIndex, const double *inPartials)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long lo, long long) { return phyhip_set_tip_partials(id, tipIndex, inPartials + lo * G->S); });
  GET_INST(I, instance);
  if (tipIndex < 0 || tipIndex >= I->tips) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "tip index %d", tipIndex);
  std::vector<uint8_t> codes((size_t)I->P);
  for (long long p = 0; p < I->P; ++p)
  {
    uint32_t m = 0;
    for (int s = 0; s < I->S; ++s)
    {
      const double x = inPartials[(size_t)p * I->S + s];
      if (x == 1.0) m |= 1u << s;
      else if (x != 0.0)
        return fail(PHYHIP_ERROR_OUT_OF_RANGE, "tip %d pattern %lld state %d: partial %g is not 0 or 1", tipIndex, p, s, x);
    }
    int code = (int)m; // S <= 8: the byte stored on the device is the allowed-state mask itself
    if (I->S > 8)
    {
      int rc = code_for_mask(I, m, &code);
      if (rc) return rc;
    }
    codes[(size_t)p] = (uint8_t)code;
  }
  return set_tip_codes(I, tipIndex, codes);
}

// One pattern of one tip rewritten in place (leave-one-out cross-validation hides a character, optimises the pendant edge
// and restores it: src/cv.c:51-118, src/mixt.c:4225-4258 -> Init_Partial_Lk_Tips_Double_One_Character, src/lk.c:2092).  In
// stream order behind whatever was queued against the old state; no host synchronisation.
int phyhip_set_tip_partials_at_pattern(int instance, int tipIndex, int pattern, const double *inPartials)
{
  if (Group *G = get_group(instance))
  {
    for (size_t g = 0; g < G->sub.size(); ++g)
      if (pattern >= G->lo[g] && pattern < G->lo[g] + G->n[g])
        return phyhip_set_tip_partials_at_pattern(G->sub_id[g], tipIndex, (int)(pattern - G->lo[g]), inPartials);
    return fail(PHYHIP_ERROR_OUT_OF_RANGE, "pattern %d", pattern);
  }
  GET_INST(I, instance);
  if (tipIndex < 0 || tipIndex >= I->tips) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "tip index %d", tipIndex);
  if (pattern < 0 || pattern >= I->P) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "pattern %d", pattern);
  uint32_t m = 0;
  for (int s = 0; s < I->S; ++s)
  {
    const double x = inPartials[s];
    if (x == 1.0) m |= 1u << s;
    else if (x != 0.0) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "tip %d pattern %d state %d: partial %g is not 0 or 1", tipIndex, pattern, s, x);
  }
  int code = (int)m; // S <= 8: the byte stored on the device is the allowed-state mask itself
  int rc   = 0;
  if (I->S > 8 && (rc = code_for_mask(I, m, &code))) return rc;
  if ((rc = flush(I, nullptr))) return rc; // queued operations read the old state
  void *st = nullptr;
  if ((rc = I->ring.alloc(16, I->stream, &st))) return rc;
  *reinterpret_cast<uint8_t *>(st) = (uint8_t)code;
  *(reinterpret_cast<uint32_t *>(st) + 1) = m;
  I->stream_dirty = true; I->touched_call = true;
  HIPCHK(hipMemcpyAsync(I->d_tipcodes + (size_t)tipIndex * I->Ppad + pattern, st, 1, hipMemcpyHostToDevice, I->stream));
  if (I->d_tipmasks)
    HIPCHK(hipMemcpyAsync(I->d_tipmasks + (size_t)tipIndex * I->Ppad + pattern, reinterpret_cast<uint32_t *>(st) + 1, 4,
                          hipMemcpyHostToDevice, I->stream));
  return upload_masks(I);
}

int phyhip_set_tip_states(int instance, int tipIndex, const int *inStates)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long lo, long long) { return phyhip_set_tip_states(id, tipIndex, inStates + lo); });
  GET_INST(I, instance);
  if (tipIndex < 0 || tipIndex >= I->tips) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "tip index %d", tipIndex);
  std::vector<uint8_t> codes((size_t)I->P);
  const uint32_t       full = (I->S == 32) ? 0xffffffffu : ((1u << I->S) - 1u);
  for (long long p = 0; p < I->P; ++p)
  {
    const int st = inStates[p];
    int       code;
    if (I->S <= 8) code = (st < 0 || st >= I->S) ? (int)full : (1 << st);
    else if (st < 0 || st >= I->S)
    {
      int rc = code_for_mask(I, full, &code);
      if (rc) return rc;
    }
    else code = st;
    codes[(size_t)p] = (uint8_t)code;
  }
  return set_tip_codes(I, tipIndex, codes);
}

int phyhip_set_partials(int instance, int bufferIndex, const double *inPartials)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long lo, long long) { return phyhip_set_partials(id, bufferIndex, inPartials + lo * G->C * G->S); });
  GET_INST(I, instance);
  int rc = check_partial_index(I, bufferIndex, false);
  if (rc) return rc;
  rc = flush_sync(I);
  if (rc) return rc;
  if (!I->perm && !I->soa)
  {
    HIPCHK(hipMemcpy(I->d_partials + (size_t)(bufferIndex - I->tips) * buf_elems(I), inPartials, buf_elems(I) * sizeof(double),
                     hipMemcpyHostToDevice));
    return PHYHIP_SUCCESS;
  }
  std::vector<double> tmp(buf_elems(I), 0.0);
  for (long long p = 0; p < I->P; ++p)
    for (int c = 0; c < I->C; ++c)
      for (int s = 0; s < I->S; ++s)
        tmp[dev_off(I, p, c, s)] = inPartials[((size_t)p * I->C + c) * I->S + s];
  HIPCHK(hipMemcpy(I->d_partials + (size_t)(bufferIndex - I->tips) * buf_elems(I), tmp.data(), buf_elems(I) * sizeof(double),
                   hipMemcpyHostToDevice));
  return PHYHIP_SUCCESS;
}

int phyhip_set_pattern_weights(int instance, const double *w)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long lo, long long) { return phyhip_set_pattern_weights(id, w + lo); });
  GET_INST(I, instance);
  int rc = flush_sync(I);
  if (rc) return rc;
  HIPCHK(hipMemcpy(I->d_wght, w, I->P * sizeof(double), hipMemcpyHostToDevice));
  return PHYHIP_SUCCESS;
}

static int small_upload(Instance *I, double *dst, const double *src, size_t n)
{
  const size_t off = (size_t)(dst - I->d_model);
  if (I->h_model_set[off] && !memcmp(I->h_model.data() + off, src, n * sizeof(double))) return PHYHIP_SUCCESS;
  // A real change (every step of Round_Optimize's model optimisation): the queued work that still belongs to the old
  // values is launched, then the new block follows it IN STREAM ORDER through the pinned staging ring -- no host
  // synchronisation (round 1 drained the stream here).
  int rc = flush(I, nullptr);
  if (rc) return rc;
  void *st = nullptr;
  if ((rc = I->ring.alloc(n * sizeof(double), I->stream, &st))) return rc;
  memcpy(st, src, n * sizeof(double));
  HIPCHK(hipMemcpyAsync(dst, st, n * sizeof(double), hipMemcpyHostToDevice, I->stream));
  memcpy(I->h_model.data() + off, src, n * sizeof(double));
  I->h_model_set[off] = 1;
  return PHYHIP_SUCCESS;
}

int phyhip_set_category_rates(int instance, const double *r)
{
  if (Group *G = get_group(instance)) return group_each(G, [&](int id, long long, long long) { return phyhip_set_category_rates(id, r); });
  GET_INST(I, instance);
  // queued matrix rebuilds ride in kernel arguments built from these host shadows (flush_impl): launch what still belongs
  // to the old model before the shadow changes
  if (memcmp(I->h_rates.data(), r, sizeof(double) * I->C) != 0)
  {
    int rc = flush(I, nullptr);
    if (rc) return rc;
  }
  I->h_rates.assign(r, r + I->C);
  return small_upload(I, I->d_catr, r, I->C);
}

int phyhip_set_category_weights(int instance, int idx, const double *w)
{
  if (Group *G = get_group(instance)) return group_each(G, [&](int id, long long, long long) { return phyhip_set_category_weights(id, idx, w); });
  GET_INST(I, instance);
  if (idx != 0) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "categoryWeightsIndex must be 0");
  return small_upload(I, I->d_catw, w, I->C);
}

int phyhip_set_state_frequencies(int instance, int idx, const double *pi)
{
  if (Group *G = get_group(instance)) return group_each(G, [&](int id, long long, long long) { return phyhip_set_state_frequencies(id, idx, pi); });
  GET_INST(I, instance);
  // (class axis: idx is the class, as BEAGLE's stateFrequenciesIndex selects a frequency buffer)
  if (idx < 0 || idx >= I->NE) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "stateFrequenciesIndex %d (0..%d)", idx, I->NE - 1);
  return small_upload(I, I->d_pi + (size_t)idx * I->S, pi, I->S);
}

int phyhip_set_eigen_decomposition(int instance, int idx, const double *evec, const double *ivec, const double *eval)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long, long long) { return phyhip_set_eigen_decomposition(id, idx, evec, ivec, eval); });
  GET_INST(I, instance);
  if (idx < 0 || idx >= I->NE) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "eigenIndex %d (0..%d)", idx, I->NE - 1);
  const size_t SS = (size_t)I->S * I->S;
  if (memcmp(I->h_eval.data() + (size_t)idx * I->S, eval, sizeof(double) * I->S) || memcmp(I->h_evec.data() + idx * SS, evec, sizeof(double) * SS) ||
      memcmp(I->h_ivec.data() + idx * SS, ivec, sizeof(double) * SS))
  { // (as in phyhip_set_category_rates: queued rebuilds are launched with the model they were queued under)
    int rc0 = flush(I, nullptr);
    if (rc0) return rc0;
  }
  std::copy(eval, eval + I->S, I->h_eval.begin() + (size_t)idx * I->S);
  std::copy(evec, evec + SS, I->h_evec.begin() + idx * SS);
  std::copy(ivec, ivec + SS, I->h_ivec.begin() + idx * SS);
  int rc = small_upload(I, I->d_evec + idx * SS, evec, SS);
  if (rc) return rc;
  rc = small_upload(I, I->d_ivec + idx * SS, ivec, SS);
  if (rc) return rc;
  return small_upload(I, I->d_eval + (size_t)idx * I->S, eval, I->S);
}

int phyhip_set_phyml_options(int instance, double l_min, double l_max, double br_len_mult, int apply_lk_scaling)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long, long long) { return phyhip_set_phyml_options(id, l_min, l_max, br_len_mult, apply_lk_scaling); });
  GET_INST(I, instance);
  const int sc = apply_lk_scaling ? 1 : 0;
  if (I->l_min == l_min && I->l_max == l_max && I->br_len_mult == br_len_mult && I->apply_scaling == sc) return PHYHIP_SUCCESS;
  int rc = flush(I, nullptr);
  if (rc) return rc;
  I->l_min = l_min; I->l_max = l_max; I->br_len_mult = br_len_mult; I->apply_scaling = sc;
  return PHYHIP_SUCCESS;
}

int phyhip_set_invariant_sites(int instance, int invar_model, double pinvar, const short *invar)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long lo, long long) { return phyhip_set_invariant_sites(id, invar_model, pinvar, invar ? invar + lo : nullptr); });
  GET_INST(I, instance);
  if (!invar && invar_model) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "invar_model set but invar == NULL");
  const bool same_sites = !invar || (I->h_invar_set && !memcmp(I->h_invar.data(), invar, I->P * sizeof(short)));
  if (same_sites && I->invar_model == (invar_model ? 1 : 0) && I->pinvar == pinvar) return PHYHIP_SUCCESS;
  int rc = same_sites ? flush(I, nullptr) : flush_sync(I); // the two scalars travel in the kernel arguments
  if (rc) return rc;
  I->invar_model = invar_model ? 1 : 0;
  I->pinvar      = pinvar;
  if (!same_sites)
  {
    HIPCHK(hipMemcpy(I->d_invar, invar, I->P * sizeof(short), hipMemcpyHostToDevice));
    I->h_invar.assign(invar, invar + I->P);
    I->h_invar_set = true;
  }
  return PHYHIP_SUCCESS;
}

// ---- transition matrices ---------------------------------------------------------------------------

static int matrices_touch(Instance *I, const int *idx, int count)
{
  for (int i = 0; i < count; ++i)
  {
    if (idx[i] < 0 || idx[i] >= I->nmat) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "matrix index %d", idx[i]);
    if (I->mat_in_queue[idx[i]])
    { // a queued operation still reads the old matrix: launch the queue first (stream order does the rest)
      int rc = flush(I, nullptr);
      if (rc) return rc;
      break;
    }
  }
  return 0;
}

int phyhip_update_transition_matrices(int instance, int eigenIndex, const int *probabilityIndices,
                                      const int *firstDerivativeIndices, const int *secondDerivativeIndices,
                                      const double *edgeLengths, int count)
{
  if (Group *G = get_group(instance)) // (a whole-tree batch launches the rebuild at once: on the shard's helper thread)
    return group_parallel(G, [&](int g) {
      return phyhip_update_transition_matrices(G->sub_id[g], eigenIndex, probabilityIndices, firstDerivativeIndices,
                                               secondDerivativeIndices, edgeLengths, count);
    });
  GET_INST_RES(I, instance);
  if (eigenIndex != 0) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "eigenIndex must be 0");
  if (firstDerivativeIndices || secondDerivativeIndices)
    return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "derivative matrices are not used by PhyML's path (see phyhip_calculate_eigen_lnl_dlnl)");
  if (count <= 0) return PHYHIP_SUCCESS;
  int rc = matrices_touch(I, probabilityIndices, count);
  if (rc) return rc;
  // Deferred like the partial updates: SPR refreshes three matrices per regraft candidate (src/spr.c:643-646);
  // they are rebuilt by ONE pmat_kernel launch right before the traversal kernel that reads them.
  for (int i = 0; i < count; ++i)
    if (I->up_slot[probabilityIndices[i]] >= 0)
    { // an upload of the same matrix is still queued: it must land before the rebuild
      if ((rc = flush_uploads(I))) return rc;
      break;
    }
  for (int i = 0; i < count; ++i)
  {
    const int m = probabilityIndices[i];
    if (I->pm_slot[m] >= 0) I->pm_len[I->pm_slot[m]] = edgeLengths[i]; // last length wins
    else
    {
      I->pm_slot[m] = (int)I->pm_idx.size();
      I->pm_idx.push_back(m);
      I->pm_len.push_back(edgeLengths[i]);
    }
  }
  // A whole-tree batch (Update_All_PMat, src/lk.c:500-512) is launched now rather than with the traversal: the device
  // rebuilds the matrices while the host walks the tree and fills the operation list.
  if (count >= kEagerPmBatch && I->eager_pmats) return flush_pmats(I);
  I_call.leave_queued_only();
  return PHYHIP_SUCCESS;
}

int phyhip_set_transition_matrix(int instance, int matrixIndex, const double *inMatrix, double paddedValue)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long, long long) { return phyhip_set_transition_matrix(id, matrixIndex, inMatrix, paddedValue); });
  (void)paddedValue;
  GET_INST_RES(I, instance);
  int rc = matrices_touch(I, &matrixIndex, 1);
  if (rc) return rc;
  if (I->pm_slot[matrixIndex] >= 0 && (rc = flush_pmats(I))) return rc; // keep rebuild-then-upload order
  const size_t bytes = (size_t)I->C * I->S * I->S * sizeof(double);
  void        *st    = nullptr;
  rc = I->ring.alloc(bytes, I->stream, &st);
  if (rc) return rc;
  memcpy(st, inMatrix, bytes);
  // queued: the matrices set since the last launch travel together (SPR sets three per candidate, src/spr.c:643-646)
  if (I->up_slot[matrixIndex] >= 0) I->up_src[I->up_slot[matrixIndex]] = (const double *)st; // last upload wins
  else
  {
    I->up_slot[matrixIndex] = (int)I->up_idx.size();
    I->up_idx.push_back(matrixIndex);
    I->up_src.push_back((const double *)st);
  }
  if ((int)I->up_idx.size() >= 4 * kUploadBatch && (rc = flush_uploads(I))) return rc;
  return PHYHIP_SUCCESS;
}

int phyhip_get_transition_matrix(int instance, int matrixIndex, double *outMatrix)
{
  if (Group *G = get_group(instance)) return phyhip_get_transition_matrix(G->sub_id[0], matrixIndex, outMatrix); // replicated
  GET_INST(I, instance);
  if (matrixIndex < 0 || matrixIndex >= I->nmat) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "matrix index %d", matrixIndex);
  int rc = flush_sync(I);
  if (rc) return rc;
  HIPCHK(hipMemcpy(outMatrix, I->d_pmats + (size_t)matrixIndex * I->C * I->S * I->S, (size_t)I->C * I->S * I->S * sizeof(double),
                   hipMemcpyDeviceToHost));
  return PHYHIP_SUCCESS;
}

// ---- hot path --------------------------------------------------------------------------------------

int phyhip_update_partials(int instance, const phyhip_operation *ops, int n, int cumulativeScaleIndex)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long, long long) { return phyhip_update_partials(id, ops, n, cumulativeScaleIndex); });
  (void)cumulativeScaleIndex;
  GET_INST_RES(I, instance);
  for (int i = 0; i < n; ++i)
  {
    const phyhip_operation &o = ops[i];
    int rc = check_partial_index(I, o.destinationPartials, false);
    if (rc) return rc;
    if ((rc = check_partial_index(I, o.child1Partials, true))) return rc;
    if ((rc = check_partial_index(I, o.child2Partials, true))) return rc;
    if (o.child1TransitionMatrix < 0 || o.child1TransitionMatrix >= I->nmat || o.child2TransitionMatrix < 0 ||
        o.child2TransitionMatrix >= I->nmat)
      return fail(PHYHIP_ERROR_OUT_OF_RANGE, "operation %d: matrix index out of range", i);
    if ((int)I->pending.size() >= I->ops_cap)
    {
      rc = flush(I, nullptr);
      if (rc) return rc;
    }
    I->pending.push_back(DevOp{o.destinationPartials, o.child1Partials, o.child2Partials, o.child1TransitionMatrix,
                               o.child2TransitionMatrix, 0});
    I->mat_in_queue[o.child1TransitionMatrix] = 1;
    I->mat_in_queue[o.child2TransitionMatrix] = 1;
  }
  I_call.leave_queued_only();
  return PHYHIP_SUCCESS;
}

// An evaluation the host waits for (edge sum, or the empty records of a small alignment's Update_Eigen_Lr): queue, launch or
// hand to the resident short-launch evaluator, wait.  An evaluation the resident workgroups do not answer is launched.
static int flush_and_wait(Instance *I, EdgeEval &ee, bool flushed = false)
{
  int rc = flushed ? 0 : flush(I, &ee);
  if (rc) return rc;
  Resident  *const by = I->r_inflight;
  const bool by_resident = by != nullptr;
  rc = wait_result(I);
  if (rc == kResidentSilent)
  { // the resident workgroups had left: retire them for good (no late record can arrive after this), put the evaluation
    // back in the queue and launch it
    ++by->n_silent;
    resident_stop(*by);
    if (by == &I->rb) big_release(I);
    I->r_inflight = nullptr; I->host_sum_n = 0;
    I->pending = I->rt_ops;
    for (const DevOp &o : I->pending) { I->mat_in_queue[o.pm1] = 1; I->mat_in_queue[o.pm2] = 1; }
    for (size_t k = 0; k < I->rt_pm_idx.size(); ++k)
      if (I->pm_slot[I->rt_pm_idx[k]] < 0)
      {
        I->pm_slot[I->rt_pm_idx[k]] = (int)I->pm_idx.size();
        I->pm_idx.push_back(I->rt_pm_idx[k]);
        I->pm_len.push_back(I->rt_pm_len[k]);
      }
    I->rt_skip = true;
    rc = flush(I, &ee);
    I->rt_skip = false;
    if (rc) return rc;
    rc = wait_result(I);
  }
  if (rc) return rc;
  if (I->fenced_eval)
  { // every store of this evaluation -- and so everything queued before it -- is in memory
    I->fenced_eval = false; I->stream_dirty = false; I->clean_after = 0;
    // (a kernel ran, or the small evaluators' dot_prod was rewritten by another set of workgroups: they re-read.  Not the
    // large-grid evaluator's: the wave that evaluates a tile's dLk is the one that wrote its products, phyhip_big.hpp)
    if (!by_resident || (ee.eigen && by != &I->rb)) ++I->clean_epoch;
  }
  return 0;
}

int phyhip_calculate_edge_log_likelihoods(int instance, const int *parent, const int *child, const int *pm, const int *d1,
                                          const int *d2, const int *cw, const int *sf, const int *cs, int count,
                                          double *outSum, double *outD1, double *outD2)
{
  (void)cs;
  Group *G = get_group(instance);
  GET_INST_RES(I, G ? G->sub_id[0] : instance);
  if (count != 1) return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "count must be 1");
  if (d1 || d2 || outD1 || outD2)
    return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "derivatives: use phyhip_calculate_eigen_lnl_dlnl (PhyML's dLk path)");
  if ((cw && cw[0] != 0) || (sf && sf[0] != 0)) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "weights/frequencies index must be 0");
  if (G) return group_edge_lnl(G, parent[0], child[0], pm[0], outSum);
  if (I->class_axis) return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "class-axis instance: use phyhip_calculate_class_mixture_log_likelihood");
  int rc = check_partial_index(I, parent[0], true);
  if (rc) return rc;
  if ((rc = check_partial_index(I, child[0], true))) return rc;
  if (pm[0] < 0 || pm[0] >= I->nmat) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "matrix index %d", pm[0]);
  if (I->co)
  { // one process per GPU: this rank's shard sum stays on the device and goes through the all-reduce
    EdgeEval ee{parent[0], child[0], pm[0], I->d_red + 1, false, I->d_red};
    if ((rc = flush(I, &ee))) return rc;
    if ((rc = reduce_and_publish(*I->co, 2, I))) return rc;
    *outSum = I->h_result[0];
    return PHYHIP_SUCCESS;
  }
  EdgeEval ee{parent[0], child[0], pm[0], nullptr, true, nullptr};
  if ((rc = flush_and_wait(I, ee))) return rc;
  *outSum = I->h_result[0];
  return PHYHIP_SUCCESS;
}

int phyhip_calculate_edge_log_likelihoods_device(int instance, int parent, int child, int pm, double *deviceOut)
{
  if (get_group(instance))
    return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "a sharded instance reduces inside phyhip_calculate_edge_log_likelihoods");
  GET_INST(I, instance);
  int rc = check_partial_index(I, parent, true);
  if (rc) return rc;
  if ((rc = check_partial_index(I, child, true))) return rc;
  if (pm < 0 || pm >= I->nmat) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "matrix index %d", pm);
  EdgeEval ee{parent, child, pm, deviceOut, false, nullptr};
  return flush(I, &ee);
}

static int mixture_lnl_impl(const int *instances, int count, const int *parent, const int *child, const int *pm,
                            const double *classProba, const double *rMatWeight, const double *eFrqWeight, double rMatWeightSum,
                            double eFrqWeightSum, double sumProbas, double *outLnL, const MixOut *mo)
{
  if (count < 1 || count > kMaxMixClasses) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "1..%d mixture classes", kMaxMixClasses);
  Instance *I0 = nullptr;
  MixParams q;
  memset(&q, 0, sizeof q);
  for (int k = 0; k < count; ++k)
  {
    GET_INST(I, instances[k]);
    if (k == 0) I0 = I;
    if (I->C != 1 || I->P != I0->P || I->dev != I0->dev)
      return fail(PHYHIP_ERROR_OUT_OF_RANGE, "mixture class %d: needs one category, the same pattern count and the same device", k);
    int rc = check_partial_index(I, parent[k], true);
    if (rc) return rc;
    if ((rc = check_partial_index(I, child[k], true))) return rc;
    if (pm[k] < 0 || pm[k] >= I->nmat) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "matrix index %d", pm[k]);
    // the class's own edge evaluation: leaves unscaled_site_lk_cat and fact_sum_scale in its device arrays, no host sync
    EdgeEval ee{parent[k], child[k], pm[k], I->d_result, false, nullptr};
    if ((rc = flush(I, &ee))) return rc;
    if (I != I0)
    { // the combination runs on the first instance's stream, after every class stream
      HIPCHK(hipEventRecord(I->ev_sync, I->stream));
      HIPCHK(hipStreamWaitEvent(I0->stream, I->ev_sync, 0));
    }
    q.site_cat[k] = I->d_site_cat; q.fact[k] = I->d_fact;
    q.proba[k] = classProba[k]; q.r_w[k] = rMatWeight[k]; q.e_w[k] = eFrqWeight[k];
  }
  q.count = count; q.P = I0->P; q.r_sum = rMatWeightSum; q.e_sum = eFrqWeightSum; q.sum_probas = sumProbas;
  q.wght = I0->d_wght; q.site_lnl = I0->d_site_lnl; q.cat_stride = 1;
  fill_mixture_invariant(I0, q);
  const int grid = (int)((I0->P + 255) / 256);
  mix_finish_setup(I0, q.fin, grid, 1, mo);
  hipLaunchKernelGGL(mixture_combine_kernel, dim3(grid), dim3(256), 0, I0->stream, q);
  HIPCHK(hipGetLastError());
  if (mo) return PHYHIP_SUCCESS;
  int rc = wait_result(I0);
  if (rc) return rc;
  *outLnL = I0->h_result[0];
  return PHYHIP_SUCCESS;
}

int phyhip_calculate_mixture_log_likelihood(const int *instances, int count, const int *parent, const int *child, const int *pm,
                                            const double *classProba, const double *rMatWeight, const double *eFrqWeight,
                                            double rMatWeightSum, double eFrqWeightSum, double sumProbas, double *outLnL)
{
  if (count >= 1 && get_group(instances[0]))
  { // class instances that are sharded instances: every shard combines its own patterns, ONE all-reduce of {warning, lnL}
    std::vector<Group *> Gs;
    int rc = mixture_groups(instances, count, Gs);
    if (rc) return rc;
    Group *G0 = Gs[0];
    rc = group_parallel(G0, [&](int g) -> int {
      int ids[kMaxMixClasses];
      for (int k = 0; k < count; ++k) ids[k] = Gs[k]->sub_id[g];
      double      *slot = shard_slot(G0->co->ctx[G0->ctx_of[g]], G0->k_of[g]);
      const MixOut mo{slot + 1, slot};
      return mixture_lnl_impl(ids, count, parent, child, pm, classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum, sumProbas,
                              nullptr, &mo);
    });
    if (rc) return rc;
    if ((rc = reduce_and_publish(*G0->co, 2, G0->sub[0]))) return rc;
    *outLnL        = G0->sub[0]->h_result[0];
    G0->last_warn  = *G0->sub[0]->h_warn;
    G0->warn_valid = true;
    return PHYHIP_SUCCESS;
  }
  return mixture_lnl_impl(instances, count, parent, child, pm, classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum, sumProbas,
                          outLnL, nullptr);
}

static int mixture_dlnl_impl(const int *instances, int count, const int *left, const int *right, double *l, const double *classProba,
                             const double *rMatWeight, const double *eFrqWeight, double rMatWeightSum, double eFrqWeightSum,
                             double sumProbas, double *outLnL, double *outDLnL, const MixOut *mo)
{
  if (count < 1 || count > kMaxMixClasses) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "1..%d mixture classes", kMaxMixClasses);
  if (std::isnan(*l)) return fail(PHYHIP_ERROR_FLOATING_POINT, "branch length is NaN");
  Instance *I0 = nullptr;
  std::vector<double> expl;
  std::vector<Instance *> cls;
  for (int k = 0; k < count; ++k)
  {
    GET_INST(I, instances[k]);
    if (k == 0)
    {
      I0 = I;
      if (*l < I->l_min) *l = I->l_min; // src/lk.c:672-673 (dLk clamps before diverting to MIXT_dLk)
      else if (*l > I->l_max) *l = I->l_max;
      expl.assign((size_t)count * 2 * I->S, 0.0);
    }
    if (I->C != 1 || I->P != I0->P || I->dev != I0->dev || I->S != I0->S)
      return fail(PHYHIP_ERROR_OUT_OF_RANGE, "mixture class %d: needs one category, the same shape and the same device", k);
    int rc = check_partial_index(I, left[k], true);
    if (rc) return rc;
    if ((rc = check_partial_index(I, right[k], true))) return rc;
    if ((rc = flush(I, nullptr))) return rc; // queued partial updates write the scale vectors read below
    // src/mixt.c:3056-3114
    const double rr  = 1.0 * I->br_len_mult * I->h_rates[0];
    double       len = (*l) * rr;
    if (len < I->l_min) len = I->l_min;
    else if (len > I->l_max) len = I->l_max;
    for (int s = 0; s < I->S; ++s)
    {
      const double ev = I->h_eval[s], ex = exp(ev * len);
      expl[(size_t)k * 2 * I->S + 2 * s]     = ex;
      expl[(size_t)k * 2 * I->S + 2 * s + 1] = ex * ev * rr;
    }
    cls.push_back(I);
  }
  // expl pairs of all classes: staged copy into the first instance's matrix scratch area
  void        *st = nullptr;
  const size_t eb = expl.size() * sizeof(double);
  int rc = I0->ring.alloc(eb, I0->stream, &st);
  if (rc) return rc;
  memcpy(st, expl.data(), eb);
  HIPCHK(hipMemcpyAsync(I0->d_mixexpl, st, eb, hipMemcpyHostToDevice, I0->stream));
  auto launch = [&](auto s_) -> int {
    constexpr int S_ = decltype(s_)::value;
    MixDlkParams<S_> q;
    memset(&q, 0, sizeof q);
    for (int k = 0; k < count; ++k)
    {
      Instance *I = cls[k];
      if (I != I0)
      {
        HIPCHK(hipEventRecord(I->ev_sync, I->stream));
        HIPCHK(hipStreamWaitEvent(I0->stream, I->ev_sync, 0));
      }
      q.dot[k] = I->d_dot;
      q.scale_l[k] = left[k] < I->tips ? nullptr : I->d_scales + (size_t)(left[k] - I->tips) * I->Ppad;
      q.scale_r[k] = right[k] < I->tips ? nullptr : I->d_scales + (size_t)(right[k] - I->tips) * I->Ppad;
      q.proba[k] = classProba[k]; q.r_w[k] = rMatWeight[k]; q.e_w[k] = eFrqWeight[k];
    }
    q.count = count; q.P = I0->P; q.r_sum = rMatWeightSum; q.e_sum = eFrqWeightSum; q.sum_probas = sumProbas;
    q.expl = I0->d_mixexpl; q.wght = I0->d_wght; q.dot_stride = S_;
    fill_mixture_invariant(I0, q);
    const int grid = (int)((I0->P + 255) / 256);
    mix_finish_setup(I0, q.fin, grid, 2, mo);
    hipLaunchKernelGGL((mixture_dlk_kernel<S_>), dim3(grid), dim3(256), 0, I0->stream, q);
    HIPCHK(hipGetLastError());
    return 0;
  };
  if (I0->S == 4) rc = launch(std::integral_constant<int, 4>());
  else if (I0->S == 20) rc = launch(std::integral_constant<int, 20>());
  else return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "mixtures: 4 or 20 states");
  if (rc) return rc;
  if (mo) return PHYHIP_SUCCESS;
  if ((rc = wait_result(I0))) return rc;
  *outLnL = I0->h_result[0];
  if (outDLnL) *outDLnL = I0->h_result[1];
  return PHYHIP_SUCCESS;
}

int phyhip_calculate_mixture_eigen_lnl_dlnl(const int *instances, int count, const int *left, const int *right, double *l,
                                            const double *classProba, const double *rMatWeight, const double *eFrqWeight,
                                            double rMatWeightSum, double eFrqWeightSum, double sumProbas, double *outLnL,
                                            double *outDLnL)
{
  if (count >= 1 && get_group(instances[0]))
  { // sharded class instances: ONE all-reduce of {warning, lnL, dlnL}
    std::vector<Group *> Gs;
    int rc = mixture_groups(instances, count, Gs);
    if (rc) return rc;
    Group *G0 = Gs[0];
    std::vector<double> ls(G0->sub.size(), *l); // every shard clamps its own copy the same way
    rc = group_parallel(G0, [&](int g) -> int {
      int ids[kMaxMixClasses];
      for (int k = 0; k < count; ++k) ids[k] = Gs[k]->sub_id[g];
      double      *slot = shard_slot(G0->co->ctx[G0->ctx_of[g]], G0->k_of[g]);
      const MixOut mo{slot + 1, slot};
      return mixture_dlnl_impl(ids, count, left, right, &ls[g], classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum,
                               sumProbas, nullptr, nullptr, &mo);
    });
    if (rc) return rc;
    *l = ls[0];
    if ((rc = reduce_and_publish(*G0->co, 3, G0->sub[0]))) return rc;
    *outLnL = G0->sub[0]->h_result[0];
    if (outDLnL) *outDLnL = G0->sub[0]->h_result[1];
    G0->last_warn  = *G0->sub[0]->h_warn;
    G0->warn_valid = true;
    return PHYHIP_SUCCESS;
  }
  return mixture_dlnl_impl(instances, count, left, right, l, classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum, sumProbas,
                           outLnL, outDLnL, nullptr);
}

int phyhip_set_mixture_invariant_sites(int instance, int invar_model, double pinvar, const short *invar, const double *piInvariantClass)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long lo, long long) {
      return phyhip_set_mixture_invariant_sites(id, invar_model, pinvar, invar ? invar + lo : nullptr, piInvariantClass);
    });
  GET_INST(I, instance);
  if (invar_model && (!invar || !piInvariantClass)) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "invar_model set but invar / frequencies missing");
  const bool same_sites = !invar_model || (I->h_invar_set && !memcmp(I->h_invar.data(), invar, I->P * sizeof(short)));
  if (!same_sites)
  { // the per-pattern table shares the slot of phyhip_set_invariant_sites: a mixture's class instances run without +I
    int rc = flush_sync(I);
    if (rc) return rc;
    HIPCHK(hipMemcpy(I->d_invar, invar, I->P * sizeof(short), hipMemcpyHostToDevice));
    I->h_invar.assign(invar, invar + I->P);
    I->h_invar_set = true;
  }
  I->mix_invar_model = invar_model ? 1 : 0;
  I->mix_pinvar      = pinvar;
  if (invar_model) for (int s = 0; s < I->S; ++s) I->mix_pi_inv[s] = piInvariantClass[s];
  return PHYHIP_SUCCESS;
}

// ---- mixtures on the class axis of ONE instance --------------------------------------------------------------------

static int class_mixture_lnl_impl(int instance, int parent, int child, int pm, const double *classProba, const double *rMatWeight,
                                  const double *eFrqWeight, double rMatWeightSum, double eFrqWeightSum, double sumProbas,
                                  double *outLnL, const MixOut *mo)
{
  GET_INST(I, instance);
  if (!I->class_axis) return fail(PHYHIP_ERROR_GENERAL, "instance %d was not created with PHYHIP_FLAG_CLASS_AXIS", instance);
  int rc = check_partial_index(I, parent, true);
  if (rc) return rc;
  if ((rc = check_partial_index(I, child, true))) return rc;
  if (pm < 0 || pm >= I->nmat) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "matrix index %d", pm);
  // ONE traversal launch for all classes (queued updates + the per-class edge likelihoods and scale exponents) ...
  EdgeEval ee{parent, child, pm, nullptr, false, nullptr};
  if ((rc = flush(I, &ee))) return rc;
  // ... and the site loop of MIXT_Lk (src/mixt.c:1027-1135) over them
  MixParams q;
  memset(&q, 0, sizeof q);
  for (int k = 0; k < I->C; ++k)
  {
    q.site_cat[k] = I->d_site_cat + k; q.fact[k] = I->d_fact + (size_t)k * I->P;
    q.proba[k] = classProba[k]; q.r_w[k] = rMatWeight[k]; q.e_w[k] = eFrqWeight[k];
  }
  q.count = I->C; q.P = I->P; q.r_sum = rMatWeightSum; q.e_sum = eFrqWeightSum; q.sum_probas = sumProbas;
  q.wght = I->d_wght; q.site_lnl = I->d_site_lnl; q.cat_stride = I->C;
  fill_mixture_invariant(I, q);
  const int grid = (int)((I->P + 255) / 256);
  mix_finish_setup(I, q.fin, grid, 1, mo);
  hipLaunchKernelGGL(mixture_combine_kernel, dim3(grid), dim3(256), 0, I->stream, q);
  HIPCHK(hipGetLastError());
  if (mo) return PHYHIP_SUCCESS;
  if ((rc = wait_result(I))) return rc;
  *outLnL = I->h_result[0];
  return PHYHIP_SUCCESS;
}

int phyhip_calculate_class_mixture_log_likelihood(int instance, int parent, int child, int pm, const double *classProba,
                                                  const double *rMatWeight, const double *eFrqWeight, double rMatWeightSum,
                                                  double eFrqWeightSum, double sumProbas, double *outLnL)
{
  if (Group *G = get_group(instance))
  {
    int rc = group_parallel(G, [&](int g) -> int {
      double      *slot = shard_slot(G->co->ctx[G->ctx_of[g]], G->k_of[g]);
      const MixOut mo{slot + 1, slot};
      return class_mixture_lnl_impl(G->sub_id[g], parent, child, pm, classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum,
                                    sumProbas, nullptr, &mo);
    });
    if (rc) return rc;
    if ((rc = reduce_and_publish(*G->co, 2, G->sub[0]))) return rc;
    *outLnL       = G->sub[0]->h_result[0];
    G->last_warn  = *G->sub[0]->h_warn;
    G->warn_valid = true;
    return PHYHIP_SUCCESS;
  }
  return class_mixture_lnl_impl(instance, parent, child, pm, classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum, sumProbas,
                                outLnL, nullptr);
}

static int class_mixture_dlnl_impl(int instance, int left, int right, double *l, const double *classProba, const double *rMatWeight,
                                   const double *eFrqWeight, double rMatWeightSum, double eFrqWeightSum, double sumProbas,
                                   double *outLnL, double *outDLnL, const MixOut *mo)
{
  GET_INST(I, instance);
  if (!I->class_axis) return fail(PHYHIP_ERROR_GENERAL, "instance %d was not created with PHYHIP_FLAG_CLASS_AXIS", instance);
  if (std::isnan(*l)) return fail(PHYHIP_ERROR_FLOATING_POINT, "branch length is NaN");
  if (*l < I->l_min) *l = I->l_min; // src/lk.c:672-673 (dLk clamps before diverting to MIXT_dLk)
  else if (*l > I->l_max) *l = I->l_max;
  int rc = check_partial_index(I, left, true);
  if (rc) return rc;
  if ((rc = check_partial_index(I, right, true))) return rc;
  if ((rc = flush(I, nullptr))) return rc; // queued partial updates write the scale vectors read below
  std::vector<double> expl((size_t)I->C * 2 * I->S);
  for (int k = 0; k < I->C; ++k)
  { // src/mixt.c:3056-3114
    const double rr  = 1.0 * I->br_len_mult * I->h_rates[k];
    double       len = (*l) * rr;
    if (len < I->l_min) len = I->l_min;
    else if (len > I->l_max) len = I->l_max;
    for (int s = 0; s < I->S; ++s)
    {
      const double ev = I->h_eval[(size_t)k * I->S + s], ex = exp(ev * len);
      expl[(size_t)k * 2 * I->S + 2 * s]     = ex;
      expl[(size_t)k * 2 * I->S + 2 * s + 1] = ex * ev * rr;
    }
  }
  void        *st = nullptr;
  const size_t eb = expl.size() * sizeof(double);
  if ((rc = I->ring.alloc(eb, I->stream, &st))) return rc;
  memcpy(st, expl.data(), eb);
  HIPCHK(hipMemcpyAsync(I->d_mixexpl, st, eb, hipMemcpyHostToDevice, I->stream));
  auto launch = [&](auto s_) {
    constexpr int S_ = decltype(s_)::value;
    MixDlkParams<S_> q;
    memset(&q, 0, sizeof q);
    for (int k = 0; k < I->C; ++k)
    {
      q.dot[k]     = I->d_dot + (size_t)k * I->S;
      q.scale_l[k] = left < I->tips ? nullptr : I->d_scales + (size_t)(left - I->tips) * scale_elems(I) + (size_t)k * I->Ppad;
      q.scale_r[k] = right < I->tips ? nullptr : I->d_scales + (size_t)(right - I->tips) * scale_elems(I) + (size_t)k * I->Ppad;
      q.proba[k] = classProba[k]; q.r_w[k] = rMatWeight[k]; q.e_w[k] = eFrqWeight[k];
    }
    q.count = I->C; q.P = I->P; q.r_sum = rMatWeightSum; q.e_sum = eFrqWeightSum; q.sum_probas = sumProbas;
    q.expl = I->d_mixexpl; q.wght = I->d_wght; q.dot_stride = I->C * I->S;
    fill_mixture_invariant(I, q);
    const int grid = (int)((I->P + 255) / 256);
    mix_finish_setup(I, q.fin, grid, 2, mo);
    hipLaunchKernelGGL((mixture_dlk_kernel<S_>), dim3(grid), dim3(256), 0, I->stream, q);
  };
  if (I->S == 4) launch(std::integral_constant<int, 4>());
  else launch(std::integral_constant<int, 20>());
  HIPCHK(hipGetLastError());
  if (mo) return PHYHIP_SUCCESS;
  if ((rc = wait_result(I))) return rc;
  *outLnL = I->h_result[0];
  if (outDLnL) *outDLnL = I->h_result[1];
  return PHYHIP_SUCCESS;
}

int phyhip_calculate_class_mixture_eigen_lnl_dlnl(int instance, int left, int right, double *l, const double *classProba,
                                                  const double *rMatWeight, const double *eFrqWeight, double rMatWeightSum,
                                                  double eFrqWeightSum, double sumProbas, double *outLnL, double *outDLnL)
{
  if (Group *G = get_group(instance))
  {
    std::vector<double> ls(G->sub.size(), *l);
    int rc = group_parallel(G, [&](int g) -> int {
      double      *slot = shard_slot(G->co->ctx[G->ctx_of[g]], G->k_of[g]);
      const MixOut mo{slot + 1, slot};
      return class_mixture_dlnl_impl(G->sub_id[g], left, right, &ls[g], classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum,
                                     sumProbas, nullptr, nullptr, &mo);
    });
    if (rc) return rc;
    *l = ls[0];
    if ((rc = reduce_and_publish(*G->co, 3, G->sub[0]))) return rc;
    *outLnL = G->sub[0]->h_result[0];
    if (outDLnL) *outDLnL = G->sub[0]->h_result[1];
    G->last_warn  = *G->sub[0]->h_warn;
    G->warn_valid = true;
    return PHYHIP_SUCCESS;
  }
  return class_mixture_dlnl_impl(instance, left, right, l, classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum, sumProbas,
                                 outLnL, outDLnL, nullptr);
}

int phyhip_get_site_log_likelihoods(int instance, double *out)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long lo, long long) { return phyhip_get_site_log_likelihoods(id, out + lo); });
  GET_INST(I, instance);
  int rc = flush_sync(I);
  if (rc) return rc;
  HIPCHK(hipMemcpy(out, I->d_site_lnl, I->P * sizeof(double), hipMemcpyDeviceToHost));
  return PHYHIP_SUCCESS;
}

int phyhip_get_site_outputs(int instance, double *c_lnL_sorted, double *cur_site_lk, double *unscaled, int *fact)
{
  if (Group *G = get_group(instance))
  {
    const bool cls = G->sub[0]->class_axis; // fact_sum_scale of a class-axis instance is [class][pattern]: per-class rows
    std::vector<int> tmp;
    return group_each(G, [&](int id, long long lo, long long n) {
      if (cls && fact) tmp.resize((size_t)n * G->C);
      const int rc = phyhip_get_site_outputs(id, c_lnL_sorted ? c_lnL_sorted + lo : nullptr, cur_site_lk ? cur_site_lk + lo : nullptr,
                                             unscaled ? unscaled + lo * G->C : nullptr, !fact ? nullptr : (cls ? tmp.data() : fact + lo));
      if (rc == PHYHIP_SUCCESS && cls && fact)
        for (int k = 0; k < G->C; ++k) memcpy(fact + (size_t)k * G->P + lo, tmp.data() + (size_t)k * n, sizeof(int) * (size_t)n);
      return rc;
    });
  }
  GET_INST(I, instance);
  int rc = flush_sync(I);
  if (rc) return rc;
  if (c_lnL_sorted) HIPCHK(hipMemcpy(c_lnL_sorted, I->d_site_lnl, I->P * sizeof(double), hipMemcpyDeviceToHost));
  if (cur_site_lk) HIPCHK(hipMemcpy(cur_site_lk, I->d_site_lk, I->P * sizeof(double), hipMemcpyDeviceToHost));
  if (unscaled) HIPCHK(hipMemcpy(unscaled, I->d_site_cat, (size_t)I->P * I->C * sizeof(double), hipMemcpyDeviceToHost));
  if (fact) HIPCHK(hipMemcpy(fact, I->d_fact, I->P * I->NE * sizeof(int), hipMemcpyDeviceToHost)); // class axis: [class][pattern]
  return PHYHIP_SUCCESS;
}

int phyhip_get_partials(int instance, int bufferIndex, int scaleIndex, double *out)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long lo, long long) { return phyhip_get_partials(id, bufferIndex, scaleIndex, out + lo * G->C * G->S); });
  (void)scaleIndex;
  GET_INST(I, instance);
  int rc = check_partial_index(I, bufferIndex, false);
  if (rc) return rc;
  if ((rc = flush_sync(I))) return rc;
  if (!I->perm && !I->soa)
  {
    HIPCHK(hipMemcpy(out, I->d_partials + (size_t)(bufferIndex - I->tips) * buf_elems(I), buf_elems(I) * sizeof(double),
                     hipMemcpyDeviceToHost));
    return PHYHIP_SUCCESS;
  }
  std::vector<double> tmp(buf_elems(I));
  HIPCHK(hipMemcpy(tmp.data(), I->d_partials + (size_t)(bufferIndex - I->tips) * buf_elems(I), buf_elems(I) * sizeof(double),
                   hipMemcpyDeviceToHost));
  for (long long p = 0; p < I->P; ++p)
    for (int c = 0; c < I->C; ++c)
      for (int s = 0; s < I->S; ++s) out[((size_t)p * I->C + c) * I->S + s] = tmp[dev_off(I, p, c, s)];
  return PHYHIP_SUCCESS;
}

int phyhip_get_class_scale_factors(int instance, int bufferIndex, int classIndex, int *out)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long lo, long long) { return phyhip_get_class_scale_factors(id, bufferIndex, classIndex, out + lo); });
  GET_INST(I, instance);
  int rc = check_partial_index(I, bufferIndex, false);
  if (rc) return rc;
  if (classIndex < 0 || classIndex >= (I->class_axis ? I->C : 1)) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "class index %d", classIndex);
  if ((rc = flush_sync(I))) return rc;
  HIPCHK(hipMemcpy(out, I->d_scales + (size_t)(bufferIndex - I->tips) * scale_elems(I) + (size_t)classIndex * I->Ppad, I->P * sizeof(int),
                   hipMemcpyDeviceToHost));
  return PHYHIP_SUCCESS;
}

int phyhip_get_scale_factors(int instance, int bufferIndex, int *out)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long lo, long long) { return phyhip_get_scale_factors(id, bufferIndex, out + lo); });
  GET_INST(I, instance);
  int rc = check_partial_index(I, bufferIndex, false);
  if (rc) return rc;
  if ((rc = flush_sync(I))) return rc;
  HIPCHK(hipMemcpy(out, I->d_scales + (size_t)(bufferIndex - I->tips) * scale_elems(I), I->P * sizeof(int), hipMemcpyDeviceToHost));
  return PHYHIP_SUCCESS;
}

int phyhip_set_scale_factors(int instance, int bufferIndex, const int *in)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long lo, long long) { return phyhip_set_scale_factors(id, bufferIndex, in + lo); });
  GET_INST(I, instance);
  int rc = check_partial_index(I, bufferIndex, false);
  if (rc) return rc;
  if ((rc = flush_sync(I))) return rc;
  HIPCHK(hipMemcpy(I->d_scales + (size_t)(bufferIndex - I->tips) * scale_elems(I), in, I->P * sizeof(int), hipMemcpyHostToDevice));
  return PHYHIP_SUCCESS;
}

int phyhip_get_numerical_warning(int instance, int *out)
{
  if (Group *G = get_group(instance))
  {
    if (!G->warn_valid) return fail(PHYHIP_ERROR_GENERAL, "no evaluation yet");
    *out = G->last_warn;
    return PHYHIP_SUCCESS;
  }
  GET_INST_RES(I, instance);
  if (!(I->warn_current && I->pending.empty()))
  { // an evaluation whose result the host did not wait for (device-side output) may still be running
    int rc = flush_sync(I);
    if (rc) return rc;
  }
  else
  { // a query: it queues nothing and is not a step of the call sequence the resident evaluators watch
    I_call.leave_query();
  }
  *out = *I->h_warn; // written by the final reduction of the last edge evaluation, ahead of its sequence number
  return PHYHIP_SUCCESS;
}

// ---- eigen basis -------------------------------------------------------------------------------------

int phyhip_update_eigen_lr(int instance, int left, int rght)
{
  if (Group *G = get_group(instance)) return group_parallel(G, [&](int g) { return phyhip_update_eigen_lr(G->sub_id[g], left, rght); });
  GET_INST_RES(I, instance);
  int rc = check_partial_index(I, left, true);
  if (rc) return rc;
  if ((rc = check_partial_index(I, rght, true))) return rc;
  // Small nucleotide alignments (the resident short-launch evaluator's range, 2 048 patterns): the queued partial update(s) and
  // the products are ONE launch of the lane-per-pattern kernel (TreeParams::edge_eval 2) -- or, mostly, one command of the
  // resident workgroups.  Measured by chain (1 Update_Eigen_Lr + 5 dLk, tools/gpu_fuse_eigen_cross.sh): 40.3 vs 43.8 us at 382
  // patterns, 53.7 vs 56.2 at 2 048; WITHOUT the resident evaluator the fused launch loses to eigen_lr_kernel at every size
  // (+2 us: it stores dot_prod 16 bytes per lane at a 64-byte stride), beyond 4 096 patterns by 4-7 us -- so nowhere else.
  // Large alignments: the same, when the large-grid resident workgroups (phyhip_big.hpp) can take it -- launched, the fused form
  // loses to eigen_lr_kernel there, as a resident command it is the partial update and the products in one trip.
  const bool big_eig = I->NE == 1 && I->C <= 4 && I->fuse_eigen && big_eligible(I) && I->pending.size() <= 2 && I->pm_idx.size() <= 4 &&
                       I->up_idx.empty() && I->args_recs && I->fold_pmats && (I->rb.launched || I->big_streak >= 1) && big_ready(I);
  if (I->NE == 1 && I->C <= 4 && I->fuse_eigen && (resident_short_eligible(I) || big_eig))
  {
    AuxProf  ap(I, 0);
    EdgeEval ee{left, rght, 0, nullptr, false, nullptr};
    ee.eigen = true;
    const bool   wc = I->warn_current;
    const double keep = I->h_result[0];
    // the workgroups (launched or resident) post empty records once their products are in memory
    if ((rc = flush_and_wait(I, ee))) return rc;
    I->h_result[0] = keep; I->warn_current = wc;
    I->eig_api_no = 0;
    return PHYHIP_SUCCESS;
  }
  if ((rc = flush(I, nullptr))) return rc;
  big_release(I, false);
  I->touched_call = true;
  EigenParams e;
  e.t = base_params(I); e.ro = base_ro(I, nullptr); e.left = left; e.rght = rght; e.r_e_vect = I->d_evec; e.l_e_vect = I->d_ivec; e.dot_prod = I->d_dot;
  const bool report = I->resident && I->S == 4 && I->host_sum && I->spin_wait && I->grid <= kResidentMaxGrid && !I->co; // (see eigen_eval)
  e.tickets = report ? I->d_tickets : nullptr;
  e.stamp_host = reinterpret_cast<unsigned long long *>(I->h_result + 3); e.stamp = report ? ++I->stamp_seq : 0ull;
  AuxProf ap(I, 0);
  rc = dispatch_shape(I, [&](auto s, auto cp) {
    constexpr int S_ = decltype(s)::value, CP_ = decltype(cp)::value;
    const size_t lds = sizeof(double) * 2 * (size_t)(I->class_axis ? I->C : 1) * S_ * S_; // the eigenvectors, staged per workgroup
    const int egrid = (int)(((long long)I->P * I->CP * kEigenSplit<S_> + 255) / 256);
    if (I->class_axis) hipLaunchKernelGGL((eigen_lr_kernel<S_, CP_, true>), dim3(egrid), dim3(256), lds, I->stream, e);
    else hipLaunchKernelGGL((eigen_lr_kernel<S_, CP_, false>), dim3(egrid), dim3(256), lds, I->stream, e);
    return 0;
  });
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  if (report) { I->stream_dirty = false; I->clean_after = e.stamp; }
  else if (big_eligible(I) && !I->prof && (rc = stamp_stream(I))) return rc; // (large instance: the dLk calls that follow can be served resident)
  I->eig_api_no = report ? I->api_no : 0;
  return PHYHIP_SUCCESS;
}

// dev_out != nullptr (sharded evaluation): the two sums stay in device memory (dev_out[0..1]), the warning flag goes to
// *warn_out as a double, nothing is handed to the host and the call does not wait.
static int eigen_eval(Instance *I, double l, bool deriv, double *lnl, double *dlnl, double *dev_out = nullptr,
                      double *warn_out = nullptr)
{
  if ((size_t)I->C * 2 * I->S > (size_t)kMaxExpl) return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "expl table too large");
  int rc = flush(I, nullptr);
  if (rc) return rc;
  DlkParams q;
  memset(&q, 0, sizeof q);
  q.dot_prod = I->d_dot; q.wght = I->d_wght; q.fact = I->d_fact; q.cat_w = I->d_catw; q.pi = I->d_pi; q.invar = I->d_invar;
  q.P = I->P; q.C = I->C; q.invar_model = I->invar_model; q.apply_scaling = I->apply_scaling; q.with_derivative = deriv ? 1 : 0;
  // grid-stride kernel.  Measured (us per dLk incl. launch and host hand-over): 512 workgroups + fused final sum win up to
  // ~100 MB of dot_prod (20 states x 100 k patterns: 33 vs 48); beyond, filling every wave slot (2048 workgroups, separate
  // final sum) streams better (4 states x 1 M patterns: 45 vs 64; the one-workgroup-per-256-lanes form took 75)
  const size_t dot_bytes = (size_t)I->P * I->C * I->S * sizeof(double);
  int          dgrid = std::min(I->grid, dot_bytes > (size_t)100 << 20 ? 2048 : 512);
  if (const char *e = diag_env("PHYHIP_DLK_GRID")) dgrid = std::min(I->grid, std::max(1, atoi(e)));
  const bool hsum  = !dev_out && I->host_sum;
  // Large nucleotide alignments with the scalar wanted on the host: the evaluation is cut into one-wave virtual blocks
  // (dlk_tile) -- what the large-grid resident workgroups serve (phyhip_big.hpp) and, launched, dlk64_kernel: the same doubles
  const bool big = hsum && big_shape(I);
  if (big) dgrid = I->n_vdlk;
  q.pinvar = I->pinvar; q.fin.block_sums = I->d_block; q.fin.stride = dgrid; q.fin.warn = I->d_warn;
  const bool fused = !hsum && fuse_reduce(I, dgrid);
  if (hsum)
  { // both sums posted to the host per workgroup (see flush_impl)
    q.fin.host_blocks = I->h_blocks; q.fin.host_tag = ++I->seq; q.fin.warn = I->h_warn;
    *I->h_warn = 0;
  }
  if (fused)
  {
    q.fin.tickets = I->d_tickets; q.fin.result = dev_out ? dev_out : I->d_result;
    q.fin.result_host = dev_out ? nullptr : I->h_result; q.fin.warn_host = I->h_warn;
    q.fin.seq = dev_out ? 0ull : ++I->seq; q.fin.warn_out = warn_out;
  }
  for (int c = 0; c < I->C; ++c)
  {
    if (deriv)
    { // src/lk.c:688-726
      const double rr  = I->h_rates[c] * I->br_len_mult;
      double       len = l * rr;
      if (len < I->l_min) len = I->l_min;
      else if (len > I->l_max) len = I->l_max;
      for (int s = 0; s < I->S; ++s)
      {
        const double ev = I->h_eval[s], ex = exp(ev * len);
        q.expl[c * 2 * I->S + 2 * s]     = ex;
        q.expl[c * 2 * I->S + 2 * s + 1] = ex * ev * rr;
      }
    }
    else
    { // src/lk.c:594-602
      double len = (l > 0.0 ? l : 0.0) * I->h_rates[c];
      len *= I->br_len_mult;
      if (len < I->l_min) len = I->l_min;
      else if (len > I->l_max) len = I->l_max;
      for (int s = 0; s < I->S; ++s) q.expl[c * I->S + s] = exp(I->h_eval[s] * len);
    }
  }
  // Small alignment, scalar wanted on the host: hand the evaluation to the resident workgroups (resident_dlk_kernel) when
  // nothing of this instance is still running on its stream -- they are not ordered with it.  Right after Update_Eigen_Lr
  // the products are a few microseconds away: poll the stream that long, else launch as usual.
  // (4 states only: a 20-state command takes four 512-byte reads per poll instead of one and the round trip loses to the launch, 14.1-14.9 against
  // 12.4-12.5 us at 2 000 patterns -- measured, tools/gpu_resident_ab2.sh)
  if (kDiag && getenv("PHYHIP_RESIDENT_DEBUG") && big)
    fprintf(stderr, "big dLk: eligible %d | dirty %d dirty_prev %d touched %d clean_after %llu stamp %llu streak %d launched %d\n", (int)big_eligible(I),
            (int)I->stream_dirty, (int)I->dirty_prev, (int)I->touched_call, I->clean_after, *reinterpret_cast<volatile unsigned long long *>(I->h_result + 3),
            I->big_streak, (int)I->rb.launched);
  if (big && big_eligible(I) && big_ready(I))
  {
    I->stream_dirty = false; // (found idle; a dLk leaves nothing on the stream)
    const int brc = big_ensure(I);
    if (brc < 0) return brc;
    if (brc == 0)
    {
      Resident          &R = I->rb;
      unsigned long long words[kBigWords];
      memset(words, 0, sizeof words);
      const int  n_expl = I->C * (deriv ? 2 : 1) * I->S;
      const bool changed = I->clean_epoch != I->rt_epoch; // kernels ran on the stream since the last command
      const bool dsum = dgrid > I->big_device_sum;
      words[0] = q.fin.host_tag;
      words[1] = kBigDlk | (q.with_derivative ? kBigDeriv : 0ull) | (q.invar_model ? kBigInvar : 0ull) | (q.apply_scaling ? kBigScaling : 0ull) |
                 (changed ? kBigChanged : 0ull) | (dsum ? kBigDeviceSum : 0ull) | (dsum && big_sum_by_group(I, dgrid) ? kBigGroupSum : 0ull);
      memcpy(&words[2], &q.pinvar, 8);
      memcpy(&words[3], q.expl, sizeof(double) * (size_t)n_expl);
      resident_send(I, R, words, kBigWords);
      I->rb_dlk_api = I->api_no; I->rt_epoch = I->clean_epoch;
      I->host_sum_n = dsum ? 1 : dgrid; I->host_sum_ns = 2;
      rc = wait_result(I);
      if (rc == 0)
      {
        *lnl = I->h_result[0];
        if (dlnl) *dlnl = I->h_result[1];
        return PHYHIP_SUCCESS;
      }
      if (rc != kResidentSilent) return rc;
      // nobody there: make sure of it, then the ordinary launch below repeats the evaluation -- under a NEW tag: workgroups
      // that had started on the command may have posted the (single) record of a device-side final sum under the old one,
      // where the launched form's first tile record goes
      ++R.n_silent;
      resident_stop(R);
      big_release(I);
      I->r_inflight = nullptr; I->host_sum_n = 0;
      q.fin.host_tag = ++I->seq;
    }
  }
  else if (big && big_eligible(I)) { ++I->rb.n_busy; I->big_streak = 0; }
  if (big) big_release(I, false); // (launched on the stream: behind the resident workgroups' exit, if there are any)
  // (20 states, re-measured in round 4 with compact commands -- exp() values only, two 512-byte reads instead of four, ONE
  // polling workgroup, the others on the device-memory mailbox: 10.7 us from command to answer on the recorded proteic search,
  // 18.2 against 17.5 us per scalar-returning call, 13.2 against 13.4 us per dLk at 2 000 patterns -- still no gain: removed)
  if (!big && hsum && I->resident && I->S == 4 && dgrid <= kResidentMaxGrid && I->spin_wait)
  {
    bool idle = !I->stream_dirty;
    if (idle && I->clean_after)
    { // the report of the last Update_Eigen_Lr: a few microseconds away at most (bounded wait, then the ordinary launch)
      volatile unsigned long long *stamp = reinterpret_cast<volatile unsigned long long *>(I->h_result + 3);
      struct timespec t0;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      for (long it = 1; *stamp < I->clean_after && idle; ++it)
      {
        __builtin_ia32_pause();
        if ((it & 255) == 0 && ns_since(t0) > 200000.0) idle = false;
      }
      if (idle) { __atomic_thread_fence(__ATOMIC_ACQUIRE); I->clean_after = 0; ++I->clean_epoch; }
    }
    if (!idle) ++I->rd.n_busy;
    if (idle)
    {
      DlkParams qs = q; // what stays the same from call to call
      qs.with_derivative = 0; qs.invar_model = 0; qs.apply_scaling = 0; qs.pinvar = 0.0; qs.fin.host_tag = 0;
      memset(qs.expl, 0, sizeof qs.expl);
      const DlkParams &o = I->r_static;
      Resident  &R = I->rd;
      const bool same = R.launched && R.grid == dgrid && o.dot_prod == qs.dot_prod && o.wght == qs.wght && o.fact == qs.fact &&
                        o.cat_w == qs.cat_w && o.pi == qs.pi && o.invar == qs.invar && o.P == qs.P && o.C == qs.C &&
                        o.fin.host_blocks == qs.fin.host_blocks && o.fin.stride == qs.fin.stride && o.fin.warn == qs.fin.warn;
      if (!same)
      {
        resident_stop(R);
        if ((rc = resident_launch_dlk(I, qs, dgrid, R.seq))) return rc;
      }
      else if (resident_gone(R))
      { // the workgroups have left (idle)
        if ((rc = resident_launch_dlk(I, qs, dgrid, R.seq))) return rc;
      }
      unsigned long long words[kResidentWords];
      const int          n_words = 3 + I->C * 2 * I->S;
      const bool         changed = I->api_no != R.api_no + 1; // something else was called since the last command
      words[0] = q.fin.host_tag;
      words[1] = (q.with_derivative ? 1u : 0u) | (q.invar_model ? 2u : 0u) | (q.apply_scaling ? 4u : 0u) | (changed ? 8u : 0u);
      memcpy(&words[2], &q.pinvar, 8);
      memcpy(&words[3], q.expl, sizeof(double) * (size_t)(n_words - 3));
      resident_send(I, R, words, n_words);
      I->host_sum_n = dgrid; I->host_sum_ns = 2;
      rc = wait_result(I);
      if (rc == 0)
      {
        *lnl = I->h_result[0];
        if (dlnl) *dlnl = I->h_result[1];
        return PHYHIP_SUCCESS;
      }
      if (rc != kResidentSilent) return rc;
      // nobody there: make sure of it (after this no resident workgroup can still write a record), then the ordinary
      // launch below repeats the evaluation under the same tag
      ++R.n_silent;
      resident_stop(R);
      I->r_inflight = nullptr; I->host_sum_n = 0;
    }
  }
  {
  AuxProf ap(I, 1);
  rc = dispatch_shape(I, [&](auto s, auto cp) {
    constexpr int S_ = decltype(s)::value, CP_ = decltype(cp)::value;
    const unsigned long long h1 = hp_now();
    // (the last-workgroup sum instead of 2 x 3 126 records was measured for this launch: 37 against 28-33 us per call -- a ticket
    // per one-wave workgroup is thousands of atomics at the memory side)
    if (big) { if constexpr (S_ == 4 && CP_ <= 4) launch_dlk64<CP_>(I, q, dgrid); }
    else hipLaunchKernelGGL((dlk_kernel<S_, CP_>), dim3(dgrid), dim3(256), 0, I->stream, q);
    if (kDiag) { g_hp.launch += hp_now() - h1; ++g_hp.n_launch; }
    return 0;
  });
  }
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  if (hsum) { I->host_sum_n = dgrid; I->host_sum_ns = 2; }
  if (big && big_eligible(I) && !I->prof && (rc = stamp_stream(I))) return rc; // (the next one can go to the resident workgroups)
  if (!fused && !hsum)
  {
    hipLaunchKernelGGL(final_reduce_kernel, dim3(1), dim3(256), 0, I->stream, (const double *)I->d_block, dgrid, 2, dgrid,
                       dev_out ? dev_out : I->d_result, dev_out ? (double *)nullptr : I->h_result, I->d_warn, I->h_warn,
                       dev_out ? 0ull : ++I->seq, warn_out);
    HIPCHK(hipGetLastError());
  }
  if (dev_out)
  {
    I->warn_current = false;
    return PHYHIP_SUCCESS;
  }
  if ((rc = wait_result(I))) return rc;
  *lnl = I->h_result[0];
  if (dlnl) *dlnl = I->h_result[1];
  return PHYHIP_SUCCESS;
}

// dLk / eigen-basis Lk on the shards + the collective (count 3: warning, lnL, dlnL)
static int group_eigen_eval(Group *G, double l, bool deriv, double *lnl, double *dlnl)
{
  int rc = group_parallel(G, [&](int g) -> int {
    double *slot = shard_slot(G->co->ctx[G->ctx_of[g]], G->k_of[g]);
    return eigen_eval(G->sub[g], l, deriv, nullptr, nullptr, slot + 1, slot);
  });
  if (rc) return rc;
  rc = reduce_and_publish(*G->co, 3, G->sub[0]);
  if (rc) return rc;
  *lnl = G->sub[0]->h_result[0];
  if (dlnl) *dlnl = G->sub[0]->h_result[1];
  G->last_warn  = *G->sub[0]->h_warn;
  G->warn_valid = true;
  return PHYHIP_SUCCESS;
}

static int rank_eigen_eval(Instance *I, double l, bool deriv, double *lnl, double *dlnl)
{
  int rc = eigen_eval(I, l, deriv, nullptr, nullptr, I->d_red + 1, I->d_red);
  if (rc) return rc;
  if ((rc = reduce_and_publish(*I->co, 3, I))) return rc;
  *lnl = I->h_result[0];
  if (dlnl) *dlnl = I->h_result[1];
  return PHYHIP_SUCCESS;
}

int phyhip_calculate_eigen_lnl_dlnl(int instance, double *l, double *outLnL, double *outDLnL)
{
  Group *G = get_group(instance);
  GET_INST_RES(I, G ? G->sub_id[0] : instance);
  I_call.leave_untouched(); // (queues nothing by itself; flush() says so if it does)
  if (std::isnan(*l)) return fail(PHYHIP_ERROR_FLOATING_POINT, "branch length is NaN"); // src/lk.c:671
  if (*l < I->l_min) *l = I->l_min;                                                     // src/lk.c:673-674
  else if (*l > I->l_max) *l = I->l_max;
  if (G) return group_eigen_eval(G, *l, true, outLnL, outDLnL);
  if (I->co) return rank_eigen_eval(I, *l, true, outLnL, outDLnL);
  return eigen_eval(I, *l, true, outLnL, outDLnL);
}

int phyhip_calculate_eigen_lnl(int instance, double l, double *outLnL)
{
  if (Group *G = get_group(instance)) return group_eigen_eval(G, l, false, outLnL, nullptr);
  GET_INST_RES(I, instance);
  I_call.leave_untouched();
  if (I->co) return rank_eigen_eval(I, l, false, outLnL, nullptr);
  return eigen_eval(I, l, false, outLnL, nullptr);
}

// ---- multi-GPU: one process per GPU ------------------------------------------------------------------------------

int phyhip_comm_get_unique_id(char *outId)
{
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == PHYHIP_UNIQUE_ID_BYTES, "ncclUniqueId size");
  NCCLCHK(ncclGetUniqueId(&id));
  memcpy(outId, &id, sizeof id);
  return PHYHIP_SUCCESS;
}

int phyhip_comm_init_rank(int instance, int nranks, int rank, const char *uniqueId)
{
  if (get_group(instance)) return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "a sharded instance already owns its communicators");
  GET_INST(I, instance);
  if (I->co) return fail(PHYHIP_ERROR_GENERAL, "instance %d already has a communicator", instance);
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "rank %d of %d", rank, nranks);
  int rc = flush_sync(I);
  if (rc) return rc;
  ncclUniqueId id;
  memcpy(&id, uniqueId, sizeof id);
  DevCtx c;
  c.dev = I->dev; c.stream = I->stream; c.nsub = 1;
  NCCLCHK(ncclCommInitRank(&c.comm, nranks, id, rank));
  HIPCHK(hipMalloc((void **)&c.d_red, sizeof(double) * kRedStride));
  HIPCHK(hipMemset(c.d_red, 0, sizeof(double) * kRedStride));
  I->co = new Collective();
  I->co->ctx.push_back(c);
  I->co->nranks = nranks;
  I->d_red      = c.d_red;
  if ((rc = warm_up_collective(*I->co))) return rc; // (collective: every rank is inside phyhip_comm_init_rank here)
  return PHYHIP_SUCCESS;
}

int phyhip_comm_size(int instance, int *outRanks)
{
  if (Group *G = get_group(instance))
  {
    *outRanks = G->co->nranks;
    return PHYHIP_SUCCESS;
  }
  GET_INST(I, instance);
  *outRanks = I->co ? I->co->nranks : 1;
  return PHYHIP_SUCCESS;
}

int phyhip_get_shard_range(int instance, int shard, int *outDevice, int *outFirstPattern, int *outPatternCount)
{
  if (Group *G = get_group(instance))
  {
    if (shard < 0 || shard >= (int)G->sub.size()) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "shard %d of %d", shard, (int)G->sub.size());
    if (outDevice) *outDevice = G->sub[shard]->dev;
    if (outFirstPattern) *outFirstPattern = (int)G->lo[shard];
    if (outPatternCount) *outPatternCount = (int)G->n[shard];
    return (int)G->sub.size();
  }
  GET_INST(I, instance);
  if (shard != 0) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "shard %d of 1", shard);
  if (outDevice) *outDevice = I->dev;
  if (outFirstPattern) *outFirstPattern = 0;
  if (outPatternCount) *outPatternCount = (int)I->P;
  return 1;
}

int phyhip_get_dot_prod(int instance, double *out)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long lo, long long) { return phyhip_get_dot_prod(id, out + lo * G->C * G->S); });
  GET_INST(I, instance);
  int rc = flush_sync(I);
  if (rc) return rc;
  HIPCHK(hipMemcpy(out, I->d_dot, (size_t)I->P * I->C * I->S * sizeof(double), hipMemcpyDeviceToHost));
  return PHYHIP_SUCCESS;
}

// ---- plumbing --------------------------------------------------------------------------------------------

int phyhip_set_stream(int instance, void *hipStream)
{
  if (get_group(instance)) return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "a sharded instance owns one stream per device");
  GET_INST(I, instance);
  int rc = flush_sync(I);
  if (rc) return rc;
  if (I->own_stream && I->stream) (void)hipStreamDestroy(I->stream);
  I->stream     = (hipStream_t)hipStream;
  I->own_stream = false;
  if (I->co) I->co->ctx[0].stream = I->stream;
  return PHYHIP_SUCCESS;
}

int phyhip_synchronize(int instance)
{
  if (Group *G = get_group(instance)) return group_each(G, [&](int id, long long, long long) { return phyhip_synchronize(id); });
  GET_INST(I, instance);
  const int rc = flush_sync(I);
  if (rc == 0) { I->stream_dirty = false; I->clean_after = 0; ++I->clean_epoch; } // nothing queued is left
  return rc;
}

int phyhip_profile(int instance, int enable)
{
  if (Group *G = get_group(instance)) return group_each(G, [&](int id, long long, long long) { return phyhip_profile(id, enable); });
  GET_INST(I, instance);
  int rc = flush_sync(I);
  if (rc) return rc;
  collect_profile(I);
  I->prof = enable != 0;
  I->prof_ms = 0.0; I->prof_n = 0; I->prof_updates = 0.0; I->prof_rd_bytes = 0.0; I->prof_wr_bytes = 0.0;
  I->prof_aux_ms[0] = I->prof_aux_ms[1] = 0.0; I->prof_aux_n[0] = I->prof_aux_n[1] = 0;
  return PHYHIP_SUCCESS;
}

int phyhip_profile_read_eigen(int instance, double *outEigenLrMs, int *outEigenLrLaunches, double *outDlkMs, int *outDlkLaunches)
{
  if (get_group(instance)) return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "per-kernel profile of a sharded instance");
  GET_INST(I, instance);
  int rc = flush_sync(I);
  if (rc) return rc;
  if ((rc = collect_profile(I))) return rc;
  if (outEigenLrMs) *outEigenLrMs = I->prof_aux_ms[0];
  if (outEigenLrLaunches) *outEigenLrLaunches = I->prof_aux_n[0];
  if (outDlkMs) *outDlkMs = I->prof_aux_ms[1];
  if (outDlkLaunches) *outDlkLaunches = I->prof_aux_n[1];
  return PHYHIP_SUCCESS;
}

int phyhip_profile_read(int instance, double *ms, int *launches, double *updates)
{
  if (Group *G = get_group(instance))
  { // slowest shard's kernel time, its launch count, site-updates of all shards
    double tms = 0.0, tup = 0.0;
    int    tn  = 0;
    const int rc = group_each(G, [&](int id, long long, long long) {
      double m = 0.0, u = 0.0;
      int    k = 0;
      const int r = phyhip_profile_read(id, &m, &k, &u);
      if (m > tms) { tms = m; tn = k; }
      tup += u;
      return r;
    });
    if (ms) *ms = tms;
    if (launches) *launches = tn;
    if (updates) *updates = tup;
    return rc;
  }
  GET_INST(I, instance);
  int rc = flush_sync(I);
  if (rc) return rc;
  if ((rc = collect_profile(I))) return rc;
  if (ms) *ms = I->prof_ms;
  if (launches) *launches = I->prof_n;
  if (updates) *updates = I->prof_updates;
  return PHYHIP_SUCCESS;
}

int phyhip_get_resident_stats(int instance, long long out[8])
{
  if (Group *G = get_group(instance))
  { // (sharded instances hand their sums to the collective on the device: never resident)
    for (int k = 0; k < 8; ++k) out[k] = 0;
    return PHYHIP_SUCCESS;
  }
  GET_INST_RES(I, instance);
  I_call.leave_query();
  for (int k = 0; k < 8; ++k) out[k] = 0;
  int k = 0;
  for (const Resident *R : {&I->rd, &I->rt})
  {
    out[k++] = (long long)R->n_cmd; out[k++] = (long long)R->n_launch; out[k++] = (long long)R->n_silent; out[k++] = (long long)R->n_busy;
  }
  return PHYHIP_SUCCESS;
}

int phyhip_get_big_resident_stats(int instance, long long out[4])
{
  for (int k = 0; k < 4; ++k) out[k] = 0;
  if (get_group(instance)) return PHYHIP_SUCCESS; // (sharded instances: never resident)
  GET_INST_RES(I, instance);
  I_call.leave_query();
  const Resident *R = &I->rb;
  out[0] = (long long)R->n_cmd; out[1] = (long long)R->n_launch; out[2] = (long long)R->n_silent; out[3] = (long long)R->n_busy;
  return PHYHIP_SUCCESS;
}

int phyhip_profile_read_traffic(int instance, double *outReadBytes, double *outWriteBytes)
{
  if (Group *G = get_group(instance))
  {
    double r = 0.0, w = 0.0;
    const int rc = group_each(G, [&](int id, long long, long long) {
      double a = 0.0, b = 0.0;
      const int k = phyhip_profile_read_traffic(id, &a, &b);
      r += a; w += b;
      return k;
    });
    if (outReadBytes) *outReadBytes = r;
    if (outWriteBytes) *outWriteBytes = w;
    return rc;
  }
  GET_INST(I, instance);
  if (outReadBytes) *outReadBytes = I->prof_rd_bytes;
  if (outWriteBytes) *outWriteBytes = I->prof_wr_bytes;
  return PHYHIP_SUCCESS;
}

} // extern "C"
