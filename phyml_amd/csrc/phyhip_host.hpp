// phyhip_host.hpp -- what the translation units of libphyhip.so's host side share: the instance, the staging ring, the
// instance table with the choke point of the resident protocol, and the declarations of each unit's functions.
//   phyhip.hip           the C ABI: life cycle, inputs, transition matrices, the queueing entry points, getters, plumbing
//   phyhip_queue.hip     the deferred queue turned into launches (flush_impl) and the waits for their scalars
//   phyhip_resident.hip  host side of the resident evaluators (small and large-grid)
//   phyhip_eigen.hip     Update_Eigen_Lr / dLk entry points
//   phyhip_mixture.hip   mixtures (class instances, class axis)
//   phyhip_shard.hip     pattern shards over several devices, the one collective
//   phyhip_big.hip       instantiations of resident_big_kernel (its own compile flags)
// Internal: nothing here is part of the ABI (include/phyhip.h).
#pragma once
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

namespace phyhip_host
{
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
inline HostProf g_hp;
static inline unsigned long long hp_now() { return kDiag ? __builtin_ia32_rdtsc() : 0ull; }

inline thread_local std::string g_err;

int fail(int code, const char *fmt, ...); // (phyhip.hip)

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

constexpr int kPushCmdsDefault = 3;    // uncached device memory (see Instance::push_cmds; measured against 1 / 2 / host memory, profiles/r04_latency.md)
constexpr int kResidentDirect = 16;    // up to this many resident workgroups poll the host themselves, above that workgroup 0 relays (measured: 8 / 16 / 32, docs/history/tools/gpu_direct_ab.sh)
// Host side of one set of resident workgroups (see resident_dlk_kernel / resident_nt2_kernel)
struct Resident
{
  // The command record as the device reads it.  Either host-mapped memory, polled over the link (a read round trip per poll:
  // only a few workgroups may do that, the others get the commands relayed through the mailbox) -- or, where the host can store
  // straight into device memory (large BAR), DEVICE memory the host pushes each command into, 64 bytes per store: every
  // workgroup then polls it locally and nobody relays.  `shadow` is the host's copy of a pushed record (what it last wrote).
  ResidentCmd        *cmd = nullptr;
  ResidentCmd        *shadow = nullptr;
  bool                pushed = false;
  double             *up_area = nullptr; // pushed records only: device memory for a command's host-computed matrices (ResidentCtl::up_area)
  unsigned long long *report = nullptr; // host-mapped: the generation whose workgroups have left (written by workgroup 0)
  unsigned long long *mail = nullptr;  // device mailbox (word 0: what workgroup 0 decided; relayed commands)
  hipStream_t         stream[2] = {nullptr, nullptr};
  unsigned long long  gen = 0, seq = 0; // launch generation; commands issued
  unsigned long long  api_no = 0;       // entry-point call of the last command
  bool                launched = false;
  int                 grid = 0;
  struct timespec     t_launch = {0, 0}, t_cmd = {0, 0};
  unsigned long long  n_cmd = 0, n_launch = 0, n_silent = 0, n_busy = 0; // phyhip_get_resident_stats
  double              ns_wait = 0.0;   // PHYHIP_RESIDENT_STATS: host time from a command's last word to its answer, summed
};

// The in-step child of a queued operation (DevOp::pad bits 1 / 2, phyhip_nt2.hpp INL): the virtual tip x tip result it reads is
// computed inside its own step from tips a, b and matrices pmA, pmB (a < 0: the operation has none)
struct InlineDef
{
  int a, b, pmA, pmB;
  bool operator==(const InlineDef &o) const { return a == o.a && b == o.b && pmA == o.pmA && pmB == o.pmB; }
};

struct Instance
{
  Collective *co         = nullptr; // one-process-per-GPU mode: communicator attached by phyhip_comm_init_rank
  double     *d_red      = nullptr; // ... and this shard's {warning, lnL, dlnL} reduction buffer (owned by co)
  int         dev        = 0;
  hipStream_t stream     = nullptr;
  bool        own_stream = true;
  int         tips = 0, nbuf = 0, S = 0, C = 0, CP = 0, nmat = 0;
  int         nmat_all = 0;    // nmat + two snapshot slots per internal buffer (virtual buffers: shadow_slot)
  long long   P = 0, Ppad = 0; // Ppad: patterns per buffer as allocated (P, or P rounded up to 16 when perm)
  bool        class_axis = false; // categories are the classes of a mixture (PHYHIP_FLAG_CLASS_AXIS; TreeParams::class_axis)
  bool        generic_loop = false; // PHYHIP_FLAG_GENERIC_LOOP: the reference's generic loop (`--cov`): plain kernel, no all-ones shortcut
  int         NE = 1;          // eigen systems / frequency vectors held: C with the class axis, else 1
  bool        perm = false;    // 20-state buffers in the MFMA fragment-major layout (phyhip_aa.hpp)
  int         aa_nw = 1;               // 20 states: wave-tiles per workgroup of traverse_aa_kernel (= its consumer waves in the one-tile-per-wave forms)
  bool        aa_d2 = false;           // ... list-form launches load two operations ahead (at most kAaMaxConsD2 consumer waves per workgroup)
  int         aa_nt = 1;               // ... wave-tiles per consumer wave in list-form launches (2: alignments with enough tiles)
  bool        soa = false;     // 4-state buffers pattern-minor, lane-per-pattern kernel (phyhip_nt2.hpp)
  int         grid_nt2 = 0;
  int         nt_groups = 1; // lanes per pattern in the lane-per-pattern nucleotide kernel
  // whole-tree traversals with two wave shapes (traverse_nt2_mixed_kernel): mix_n2 two-lane workgroups (a multiple of the CU
  // count) + mix_n4 four-lane ones for the remaining patterns; 0: one shape
  int         mix_n2 = 0, mix_n4 = 0;
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
  int          push_cmds = 0;  // resident command records in device memory, pushed by the host (1 hipMalloc, 2 fine-grained, 3 uncached; 0: host memory)
  double       resident_idle_us = 1000.0; // PHYHIP_RESIDENT_IDLE_US: the workgroups leave after this long without a command
  Resident     rd, rt;                // the dLk evaluator (resident_dlk_kernel) and the short-launch one (resident_nt2_kernel)
  // large-grid resident evaluator (resident_big_kernel, phyhip_big.hpp): nucleotide instances of more than kResidentMaxGrid tiles
  Resident     rb;
  TreeParams   rb_static;             // what its workgroups were launched with
  double      *d_tile_sums = nullptr; // [2][max(grid_nt2, n_vdlk)] tile sums of commands whose final sum runs on the device
  unsigned    *d_big_tickets = nullptr;
  HostBlock   *d_big_recs = nullptr;  // [2][kBigGroupWgs] partial sums per workgroup (BigCtl::wg_recs)
  bool         big_group_sum = true;  // (diag: PHYHIP_BIG_GROUP_SUM=0 keeps the per-tile sums and the tickets)
  bool         big_oneshot = true;    // (diag: PHYHIP_BIG_ONE_SHOT=0 launches such evaluations the old way: pmat_kernel + one-wave workgroups)
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
  int                 rt_up_n = 0;      // ... and the host-computed matrices it carried (ResidentCtl::up_area)
  int                 rt_up_idx[kArgUp] = {0, 0, 0};
  double              rt_up_val[kArgUp][64];
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
  unsigned long long model_epoch = 1; // bumped whenever the model block on the device changes (resident 20-state evaluator: its staged eigen system)
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
  // Virtual buffers: a whole-tree traversal does not STORE the results of its tip x tip operations -- each
  // is recomputed in registers right in front of the operation that consumes it (two matrix columns and a product per
  // pattern: cheaper than the 16-byte-per-entry write and the read back) -- so the buffer's memory is stale until something
  // needs it: then the defining operation is queued again, storing (devirtualise).  virt[b] != 0 says so, vdef[b] is that
  // operation; the tips and the two matrices it reads may not change while virt[b] (matrices_touch, the tip setters).
  int                                    virt_min_ops = 16; // lists at least this long leave tip x tip results virtual (0: never)
  int                                    n_virtual = 0;
  bool                                   virt_inline = true; // (diag: PHYHIP_VIRT_INLINE=0 re-issues every virtual child as a step of its own)
  std::vector<InlineDef>                 pending_inl;        // per entry of `pending` as rewrite_pending left it (or empty: none has one)
  std::vector<std::vector<InlineDef>>    slot_inl;           // ... of the lists the device ring slots hold (content cache)
  std::vector<unsigned char>             virt;              // [nbuf]
  std::vector<DevOp>                     vdef;              // [nbuf]
  std::vector<unsigned char>             keep_real_flag;    // [nbuf]: buffer is in keep_real
  std::vector<int>                       keep_real;         // buffers the caller reads from MEMORY right after the next launch (devirtualise): that launch leaves them stored
  unsigned long long                     n_virt_skipped = 0, n_virt_recomputed = 0, n_virt_material = 0; // phyhip_get_virtual_stats
  std::vector<int>                       pm_idx;    // queued device-side matrix rebuilds (index, edge length)
  std::vector<double>                    pm_len;
  std::vector<int>                       pm_slot;   // matrix index -> position in pm_idx, or -1
  std::vector<int>                       pm_shadow; // per queued rebuild: slot that receives the matrix's OLD value first, or -1
  int                                    n_pm_shadow = 0; // ... how many of them ask for one (such a list is rebuilt by pmat_kernel)
  std::vector<std::vector<DevOp>>        slot_ops;  // what each device ring slot currently holds (content cache)
  std::vector<int>                       slot_kind; // 0 slim, 1 fat dist 1, 2 fat dist 2
  std::vector<unsigned char>             mat_in_queue; // matrix index referenced by a queued op
  std::vector<int>                       up_idx;       // host-computed matrices waiting for their upload launch
  std::vector<const double *>            up_src;       // ... their copies in pinned staging memory
  std::vector<int>                       up_slot;      // per matrix: position in up_idx or -1
  std::vector<int>                       up_shadow;    // per queued upload: slot that receives the old value first, or -1
  int                                    n_up_shadow = 0; // ... how many ask for one (such uploads go through upload_matrices_kernel)
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
  std::vector<hipEvent_t> prof_spare;      // events of collected launches, reused (creating one costs more than recording it)
  std::vector<ProfPair> prof_aux;          // while profiling: kind 0 eigen_lr_kernel (K3), 1 dlk_kernel (K4), 2 the collective path (local sum, all-reduce, publish)
  double     prof_aux_ms[3] = {0.0, 0.0, 0.0};
  int        prof_aux_n[3]  = {0, 0, 0};
  double     prof_ms = 0.0, prof_updates = 0.0;
  double     prof_rd_bytes = 0.0, prof_wr_bytes = 0.0; // traffic model of the profiled launches (phyhip_profile_read_traffic)
  int        prof_n = 0;
  char       prof_kernel[96] = {0}; // the traversal kernel of the last profiled launch, as a profiler names it (phyhip_profile_read_kernel)
};

// The calling thread's current device is ASKED, not remembered: a host application (or another library in its process) may
// call hipSetDevice between two calls of this ABI, and a remembered value would then send the next launch to the wrong device
// without any error.  hipGetDevice reads a thread-local of the runtime (tens of nanoseconds).
inline thread_local int g_cur_dev = -1; // (what this library last set: only a hint for the cases below that do not ask)
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

void big_release(Instance *I, bool restart_streak = true); // (phyhip_resident.hip)

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
  // (a fixed array of atomic pointers: the lookup of every entry-point call takes no lock -- hundreds of thousands of calls
  // per tree search, a hundred in front of every full traversal)
  static constexpr int                  kCapacity = 1 << 16;
  static inline std::mutex              mu_;  // add / remove
  static inline std::atomic<Instance *> tab_[kCapacity];
  static inline int                     used_ = 0; // slots ever handed out
  static Instance *at(int id) { return (id >= 0 && id < kCapacity) ? tab_[id].load(std::memory_order_acquire) : nullptr; }
  template <bool> friend class Entered;

 public:
  static int add(Instance *I) // phyhip_create_instance (-1: the table is full)
  {
    std::lock_guard<std::mutex> lk(mu_);
    for (int i = 0; i < used_; ++i)
      if (!tab_[i].load(std::memory_order_relaxed))
      {
        tab_[i].store(I, std::memory_order_release);
        return i;
      }
    if (used_ == kCapacity) return -1;
    tab_[used_].store(I, std::memory_order_release);
    return used_++;
  }
  static Instance *remove(int id) // phyhip_finalize_instance: out of the table, the caller frees it
  {
    std::lock_guard<std::mutex> lk(mu_);
    if (id < 0 || id >= used_) return nullptr;
    return tab_[id].exchange(nullptr, std::memory_order_acq_rel);
  }
  static Instance *wiring(int id) { return at(id); } // a sharded group attaching / detaching its sub-instances: no call runs
};

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

inline int next_pow2(int x)
{
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}

inline size_t buf_elems(const Instance *I) { return I->perm ? aa_buf_elems(I->Ppad, I->C) : (size_t)I->Ppad * I->C * I->S; }
// ints per partials buffer in the scale table: one exponent per pattern, or per (class, pattern) with the class axis
inline size_t scale_elems(const Instance *I) { return (size_t)I->Ppad * (I->class_axis ? I->C : 1); }

// element offset of (pattern, category, state) inside a device partials buffer of a non-host layout
inline size_t dev_off(const Instance *I, long long p, int c, int s)
{
  if (I->perm) return aa_off(p, I->C, c, s);
  return ((size_t)(c * 2 + (s >> 1)) * I->Ppad + (size_t)p) * 2 + (size_t)(s & 1); // pattern-minor, state pairs of 16 bytes
}

inline TreeParams base_params(Instance *I)
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

inline RO base_ro(Instance *I, const DevOp *ops)
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
  CASE(4, 16) CASE(4, 32) CASE(4, 64) CASE(20, 16) CASE(20, 32) CASE(20, 64) // (more than 8 categories: the plain kernels only)
#undef CASE
  return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "no kernel for %d states x %d categories", I->S, I->C);
}

inline int upload_masks(Instance *I)
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

// ---- phyhip_queue.hip ----------------------------------------------------------------------------------------------------
int  flush_uploads(Instance *I);
int  flush_pmats(Instance *I);
bool fuse_reduce(const Instance *I, int nblocks);
int  flush_impl(Instance *I, const EdgeEval *ee);
int  flush(Instance *I, const EdgeEval *ee);
int  flush_sync(Instance *I);
int  flush_and_wait(Instance *I, EdgeEval &ee, bool flushed = false);
int  check_partial_index(const Instance *I, int idx, bool allow_tip);
int  wait_host_sum(Instance *I);
int  wait_result(Instance *I);
int  wait_result_impl(Instance *I);
int  collect_profile(Instance *I);
void devirtualise(Instance *I, int buf);       // queue (in front) the storing operation that makes buffer `buf` real again
void devirtualise_all(Instance *I);
void devirtualise_matrix(Instance *I, int m);  // ... for every virtual buffer whose definition reads matrix m
void devirtualise_tip(Instance *I, int tip);   // ... reads tip row `tip`
// snapshot slot w (0: the definition's first matrix, 1: its second) of internal buffer b in the matrix tables
// (the launch in between has happened: nothing is asked to stay stored any more)
inline void keep_real_clear(Instance *I)
{
  for (int b : I->keep_real) I->keep_real_flag[b] = 0;
  I->keep_real.clear();
}
inline int shadow_slot(const Instance *I, int b, int w) { return I->nmat + 2 * (b - I->tips) + w; }
void rewrite_pending(Instance *I, const EdgeEval *ee, bool may_virtualise);

// ---- resident evaluators: host side -----------------------------------------------------------------------------
constexpr int kResidentMaxGrid = 64;   // resident evaluator: alignments of up to this many dLk workgroups
// 20 states, one eigen system, categories on their own axis: eigen-basis evaluations described by the edge length, the expl table
// built by the workgroups (DlkParams::from_len, phyhip_kernels.hpp) -- launched and resident alike
inline bool dlk_from_len(const Instance *I) { return I->S == 20 && I->NE == 1 && !I->class_axis && I->C <= 8; }
constexpr int kResidentSilent = -4242; // wait_host_sum: the resident workgroups did not answer (not an error)
static double ns_since(const struct timespec &t0)
{
  struct timespec t1;
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec);
}

// ---- phyhip_resident.hip -------------------------------------------------------------------------------------------------
void       resident_stop(Resident &R);
void       resident_free(Resident &R);
int        resident_prepare(Instance *I, Resident &R, int grid, int n_words, unsigned long long served, ResidentCtl &r, hipStream_t *st,
                            bool in_order = false);
void       resident_launched(Resident &R, int grid);
bool       resident_gone(const Resident &R);
void       resident_send(Instance *I, Resident &R, const unsigned long long *words, int n_words);
bool       resident_push_uploads(Resident &R, int n, const double (*vals)[64], int doubles_each);
int        resident_launch_dlk(Instance *I, const DlkParams &qs, int dgrid, unsigned long long served);
bool       big_clean(Instance *I);
int        stamp_stream(Instance *I);
bool       big_ready(Instance *I);
int        big_launch(Instance *I, const TreeParams &sq);
int        big_one_shot(Instance *I, const unsigned long long *words, int n_words);
TreeParams big_static_params(Instance *I);
int        big_ensure(Instance *I);



// instances whose short launches (SPR candidates, Lk(b), Update_Eigen_Lr) the resident workgroups of resident_nt2_kernel may take
inline bool resident_short_eligible(const Instance *I)
{
  return I->resident && I->spin_wait && I->host_sum && I->soa && !I->co && !I->class_axis && I->grid_nt2 <= kResidentMaxGrid &&
         !I->ablate && I->nt_groups <= 2;
}

// ... and the resident form of the 20-state kernel (phyhip_aa.hpp, RES): instances whose launched form has one wave-tile per workgroup
inline bool resident_aa_eligible(const Instance *I)
{
  return I->resident && I->spin_wait && I->host_sum && I->perm && !I->co && !I->class_axis && I->aa_nw == 1 && I->aa_nt == 1 && !I->ablate &&
         I->grid_aa <= kResidentMaxGrid * kAaMaxCons2 && !(I->rt.n_silent > 16 && I->rt.n_silent * 4 > I->rt.n_cmd);
}

// ---- the large-grid resident evaluator (phyhip_big.hpp): host side ---------------------------------------------------
// One instance per device at a time: the resident workgroups fill the device (a workgroup per CU at the register budget of
// the traversal kernel), so a second set could not start before the first has left.
inline std::atomic<Instance *> g_big_owner[64];

// Instances whose dLk runs in the traversal's tiles (dlk_tile / dlk64_kernel) -- a property of the instance alone, so that
// its evaluations return the same doubles whether the resident workgroups are enabled or not ...
inline bool big_shape(const Instance *I)
{
  return I->host_sum && I->soa && !I->co && !I->class_axis && I->grid_nt2 > kResidentMaxGrid && I->nt_groups <= 2 && !I->ablate;
}
// ... and whether those workgroups may serve it
// (an instance whose commands keep going unanswered -- more than 16 and more than a quarter of them -- stops asking: every such
// command costs the wait, the workgroups' exit and a launch)
inline bool big_eligible(const Instance *I)
{
  return big_shape(I) && I->resident && I->spin_wait && I->dev >= 0 && I->dev < 64 && !(I->rb.n_silent > 16 && I->rb.n_silent * 4 > I->rb.n_cmd);
}

// The final sum through one partial sum per workgroup (phyhip_big.hpp, kBigGroupSum): the tiles of a workgroup are one
// accumulator of final_reduce_kernel's order only when there are exactly as many workgroups as accumulators
inline bool big_sum_by_group(const Instance *I, int tiles)
{
  return I->big_group_sum && I->big_wgs == kBigGroupWgs && tiles <= kBigGroupWgs * kBigGroupTiles;
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
inline void host_sum_finish(Instance *I, FinishParams &f, int grid, int ns)
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
inline void mix_finish_setup(Instance *I0, FinishParams &fin, int grid, int ns, const MixOut *mo)
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

} // namespace phyhip_host
