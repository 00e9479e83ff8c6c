// phyhip_resident.hip -- host side of the resident evaluators: launching, commanding and releasing the persistent workgroups
// (libphyhip.so, gfx950 only; the units and what they share: phyhip_host.hpp)
#include "phyhip_host.hpp"

#include <immintrin.h>

namespace phyhip_host
{

// "Everything queued on this stream before me has finished and is in memory": one thread, one store into host-mapped
// memory.  Launched behind evaluations of large nucleotide instances that the resident workgroups could not take (stream
// not known to be idle, first call of a streak): the host finds the stream idle again without synchronising it -- and without
// the launched kernel's waves writing back their L2 before they post (megabytes of dirty lines at these sizes).
static __global__ void stream_stamp_kernel(unsigned long long *stamp_host, unsigned long long stamp)
{
  __hip_atomic_store(stamp_host, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- writing the command record ------------------------------------------------------------------------------------------
// A pushed record (device memory behind the BAR, write-combining for the host) is written 64 bytes at a time -- two sectors,
// payload and number together -- with MOVDIR64B where the CPU has it: one store, one write on the link, never seen half.
// Without it: the payload words of the line, a store fence, the sector numbers, a store fence (posted writes stay in order).
static bool cpu_has_movdir64b()
{
  static const int have = [] {
    unsigned a = 0, b = 0, c = 0, d = 0;
    __asm__ volatile("cpuid" : "=a"(a), "=b"(b), "=c"(c), "=d"(d) : "a"(7), "c"(0));
    if (diag_env("PHYHIP_PUSH_NO_MOVDIR")) return 0; // (diag: the two-pass form on a CPU that has the instruction)
    return (int)((c >> 28) & 1u);
  }();
  return have != 0;
}
__attribute__((target("movdir64b"))) static void store64_direct(void *dst, const void *src) { _movdir64b(dst, src); }
static void push_lines(Resident &R, int first_sector, int last_sector) // sectors of ResidentCmd, counted from the control sector
{
  const char *src = reinterpret_cast<const char *>(R.shadow);
  char       *dst = reinterpret_cast<char *>(R.cmd);
  const int   l0 = first_sector / 2, l1 = last_sector / 2;
  if (cpu_has_movdir64b())
    for (int l = l0; l <= l1; ++l) store64_direct(dst + 64 * l, src + 64 * l);
  else
  {
    for (int l = l0; l <= l1; ++l)
      for (int h = 0; h < 2; ++h)
        for (int k = 0; k < kResidentPay; ++k)
          reinterpret_cast<volatile unsigned long long *>(dst + 64 * l + 32 * h)[k] = reinterpret_cast<const unsigned long long *>(src + 64 * l + 32 * h)[k];
    _mm_sfence();
    for (int l = l0; l <= l1; ++l)
      for (int h = 0; h < 2; ++h)
        reinterpret_cast<volatile unsigned long long *>(dst + 64 * l + 32 * h)[kResidentPay] =
            reinterpret_cast<const unsigned long long *>(src + 64 * l + 32 * h)[kResidentPay];
  }
  _mm_sfence();
}
// A command's host-computed matrices into ResidentCtl::up_area, in front of the command itself (posted writes arrive in order)
bool resident_push_uploads(Resident &R, int n, const double (*vals)[64], int doubles_each)
{
  if (!R.pushed || !R.up_area) return false;
  alignas(64) double line[8];
  char *dst = reinterpret_cast<char *>(R.up_area);
  for (int m = 0; m < n; ++m)
    for (int l = 0; l * 8 < doubles_each; ++l)
    {
      if (cpu_has_movdir64b())
      {
        memcpy(line, vals[m] + l * 8, 64);
        store64_direct(dst + (size_t)m * 512 + 64 * l, line);
      }
      else
        for (int k = 0; k < 8; ++k) reinterpret_cast<volatile double *>(dst + (size_t)m * 512 + 64 * l)[k] = vals[m][l * 8 + k];
    }
  _mm_sfence();
  return true;
}

// control word 0 (generation in charge) / 1 (leave)
static void set_ctl(Resident &R, int word, unsigned long long v)
{
  if (R.pushed)
  {
    R.shadow->ctl.w[word] = v;
    push_lines(R, 0, 0);
  }
  else __atomic_store_n(&R.cmd->ctl.w[word], v, __ATOMIC_RELEASE);
}

// Tell the resident workgroups (if any) to leave and wait until they have.
void resident_stop(Resident &R)
{
  if (!R.cmd || !R.launched) return;
  set_ctl(R, 1, 1ull);
  for (hipStream_t st : R.stream)
    if (st) (void)hipStreamSynchronize(st);
  set_ctl(R, 1, 0ull);
  R.launched = false;
}

void resident_free(Resident &R)
{
  if (R.cmd)
  { // (also a generation that was only told to leave, big_release: nobody may still be polling the record when it is freed)
    set_ctl(R, 1, 1ull);
    for (hipStream_t st : R.stream)
      if (st) (void)hipStreamSynchronize(st);
    R.launched = false;
  }
  resident_stop(R);
  if (R.cmd) (void)(R.pushed ? hipFree(R.cmd) : hipHostFree(R.cmd));
  if (R.up_area) (void)hipFree(R.up_area);
  if (R.shadow) free(R.shadow);
  if (R.report) (void)hipHostFree(R.report);
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
int resident_prepare(Instance *I, Resident &R, int grid, int n_words, unsigned long long served, ResidentCtl &r, hipStream_t *st,
                     bool in_order)
{
  if (!R.cmd)
  {
    const size_t bytes = (sizeof(ResidentCmd) + 63) & ~size_t(63);
    if (I->push_cmds)
    { // device memory the host stores into (checked at instance creation: large BAR); the record starts on a 64-byte line
      hipError_t e = I->push_cmds == 1 ? hipMalloc((void **)&R.cmd, bytes)
                                       : hipExtMallocWithFlags((void **)&R.cmd, bytes, I->push_cmds == 2 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
      if (e == hipSuccess && posix_memalign((void **)&R.shadow, 64, bytes) == 0)
      {
        // (zeroed on the instance's own stream and waited for THERE: a device-wide synchronisation would also wait for other
        // instances' resident workgroups, which only leave after their idle time -- a mixture of 64 class instances met that
        // stall on every instance's first use)
        HIPCHK(hipMemsetAsync(R.cmd, 0, bytes, I->stream));
        // (the matrices a command may bring along, same kind of memory; without it such commands are launched)
        const size_t ub = sizeof(double) * 64 * kArgUp;
        const hipError_t e2 = I->push_cmds == 1 ? hipMalloc((void **)&R.up_area, ub)
                                                : hipExtMallocWithFlags((void **)&R.up_area, ub, I->push_cmds == 2 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
        if (e2 != hipSuccess) { (void)hipGetLastError(); R.up_area = nullptr; }
        else HIPCHK(hipMemsetAsync(R.up_area, 0, ub, I->stream));
        HIPCHK(hipStreamSynchronize(I->stream));
        memset(R.shadow, 0, bytes);
        R.pushed = true;
      }
      else
      {
        if (e == hipSuccess) (void)hipFree(R.cmd);
        else (void)hipGetLastError(); // (the record falls back to host memory: no stale error for the next launch check to report)
        R.cmd = nullptr; R.shadow = nullptr;
      }
    }
    if (!R.pushed)
    {
      HIPCHK(hipHostMalloc((void **)&R.cmd, bytes, hipHostMallocMapped | hipHostMallocCoherent)); // (polled from the device while it changes)
      memset(R.cmd, 0, bytes);
    }
    HIPCHK(hipHostMalloc((void **)&R.report, 64, hipHostMallocMapped | hipHostMallocCoherent));
    memset(R.report, 0, 64);
    HIPCHK(hipMalloc((void **)&R.mail, sizeof(ResidentCmd)));
    HIPCHK(hipMemsetAsync(R.mail, 0, sizeof(ResidentCmd), I->stream));
    HIPCHK(hipStreamSynchronize(I->stream));
    for (hipStream_t &s2 : R.stream) HIPCHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  }
  ++R.n_launch;
  ++R.gen;
  set_ctl(R, 0, R.gen);
  r.cmd = R.cmd; r.gen = R.gen; r.start_seq = served; r.n_sectors = (n_words + kResidentPay - 1) / kResidentPay;
  r.mail = R.mail; r.report = R.report; r.up_area = R.up_area;
  r.relay = (!R.pushed && grid > I->resident_direct) ? 1 : 0; // (a pushed record is polled locally: by everybody)
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
void resident_launched(Resident &R, int grid)
{
  R.grid = grid; R.launched = true;
  clock_gettime(CLOCK_MONOTONIC, &R.t_launch);
}

// Have the workgroups of the current generation left?  (Workgroup 0 reports it, see ResidentCmd::report.)
bool resident_gone(const Resident &R)
{
  return R.cmd && __atomic_load_n(R.report, __ATOMIC_ACQUIRE) == R.gen;
}

// The command: payload words into their sectors, each sector's number last (see ResidentCmd)
void resident_send(Instance *I, Resident &R, const unsigned long long *words, int n_words)
{
  ++R.seq;
  ResidentCmd *const rec = R.pushed ? R.shadow : R.cmd;
  int               nsec = 0;
  for (int l = 0; l * kResidentPay < n_words; ++l, ++nsec)
  {
    ResidentSector &sc = rec->sec[l];
    for (int k = 0; k < kResidentPay && l * kResidentPay + k < n_words; ++k) sc.w[k] = words[l * kResidentPay + k];
    if (R.pushed) sc.seq = R.seq;
    else __atomic_store_n(&sc.seq, R.seq, __ATOMIC_RELEASE);
  }
  if (R.pushed) push_lines(R, 1, nsec); // (sector 0 is control; it shares its line with the first command sector)
  R.api_no = I->api_no;
  ++R.n_cmd;
  clock_gettime(CLOCK_MONOTONIC, &R.t_cmd);
  I->r_inflight = &R;
}

int resident_launch_dlk(Instance *I, const DlkParams &qs, int dgrid, unsigned long long served)
{
  ResidentCtl r;
  hipStream_t st;
  int rc = resident_prepare(I, I->rd, dgrid, qs.from_len ? 4 : 3 + I->C * 2 * I->S, served, r, &st);
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

// Tell the large-grid resident workgroups (if any) to leave -- without waiting for it: whatever this instance launches next
// needs their wave slots, and gets them as they go.  (A new generation number is all it takes: workgroup 0 sees it at its
// next poll of the host record and passes it on through the mailbox.)
// The instance's stream is ordered behind their exit: the resident workgroups do not write back what they store while they
// stay (a write-back per command of megabytes of dirty lines cost more than the command: measured, round 4) -- the end of
// their kernel does, and kernels and copies of this instance that come later must find it in memory.
void big_release(Instance *I, bool restart_streak)
{
  if (restart_streak) I->big_streak = 0;
  Resident &R = I->rb;
  if (R.cmd && R.launched)
  {
    hipStream_t st = R.stream[0]; // (where every generation of these workgroups runs, resident_prepare)
    ++R.gen;
    set_ctl(R, 0, R.gen);
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
bool big_clean(Instance *I)
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
int stamp_stream(Instance *I)
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
bool big_ready(Instance *I)
{
  if (!big_eligible(I) || I->prof || I->rt_skip) return false;
  Instance *owner = g_big_owner[I->dev].load();
  if (owner && owner != I) return false;
  if (!big_clean(I)) return false;
  return true;
}

// device memory of the final sums (both forms of the kernel), on first use
static int big_alloc(Instance *I)
{
  if (I->d_tile_sums) return 0;
  const size_t n = (size_t)std::max(I->grid_nt2, I->n_vdlk);
  HIPCHK(hipMalloc((void **)&I->d_tile_sums, 2 * n * sizeof(double)));
  HIPCHK(hipMalloc((void **)&I->d_big_tickets, sizeof(unsigned) * (1 + kTicketGroups)));
  HIPCHK(hipMemsetAsync(I->d_big_tickets, 0, sizeof(unsigned) * (1 + kTicketGroups), I->stream));
  HIPCHK(hipMalloc((void **)&I->d_big_recs, sizeof(HostBlock) * 2 * kBigGroupWgs));
  HIPCHK(hipMemsetAsync(I->d_big_recs, 0, sizeof(HostBlock) * 2 * kBigGroupWgs, I->stream)); // (tag 0: no evaluation's)
  if (getenv("PHYHIP_RESIDENT_STATS"))
  {
    HIPCHK(hipMalloc((void **)&I->d_big_stamps, sizeof(unsigned long long) * 16 * (size_t)I->big_wgs));
    HIPCHK(hipMemsetAsync(I->d_big_stamps, 0, sizeof(unsigned long long) * 16 * (size_t)I->big_wgs, I->stream));
  }
  HIPCHK(hipStreamSynchronize(I->stream));
  return 0;
}

static void big_fill_args(Instance *I, const TreeParams &sq, BigArgs &a)
{
  memset(&a, 0, sizeof a);
  a.t = sq;
  a.b.n_tiles = I->grid_nt2; a.b.n_vdlk = I->n_vdlk; a.b.tile_sums = I->d_tile_sums; a.b.tickets = I->d_big_tickets; a.b.dot_prod = I->d_dot;
  a.b.stamps = I->d_big_stamps; a.b.wg_recs = I->d_big_recs;
  a.pmats = I->d_pmats; a.tip_codes = I->d_tipcodes;
}

// The launched form of the same kernel: ONE evaluation, its command in the arguments, on the instance's own stream (ordered
// with everything else there: no idle stream needed).  The answer arrives as after a resident command: one {sum, tag} record
// per sum in the host blocks.
int big_one_shot(Instance *I, const unsigned long long *words, int n_words)
{
  int rc = big_alloc(I);
  if (rc) return rc;
  BigArgs a;
  big_fill_args(I, big_static_params(I), a);
  a.n_one_shot = n_words;
  memcpy(a.one_shot, words, sizeof(unsigned long long) * (size_t)n_words);
  if (launch_resident_big(I->C, I->nt_groups, I->big_wgs, I->stream, a) != 0)
    return fail(PHYHIP_ERROR_GENERAL, "large-grid evaluator: no kernel for %d categories in %d groups", I->C, I->nt_groups);
  HIPCHK(hipGetLastError());
  return 0;
}

int big_launch(Instance *I, const TreeParams &sq)
{
  Resident &R = I->rb;
  {
    const int rc = big_alloc(I); // (the stream is idle: this evaluation was about to bypass it)
    if (rc) return rc;
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
  big_fill_args(I, sq, a);
  a.r = r;
  if (launch_resident_big(I->C, I->nt_groups, I->big_wgs, st, a) != 0)
    return fail(PHYHIP_ERROR_GENERAL, "large-grid resident evaluator: no kernel for %d categories in %d groups", I->C, I->nt_groups);
  HIPCHK(hipGetLastError());
  memcpy(&I->rb_static, &sq, sizeof sq);
  resident_launched(R, I->big_wgs);
  return 0;
}

// The launch arguments of the resident workgroups: everything of a short launch's TreeParams that does not change per call
TreeParams big_static_params(Instance *I)
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
int big_ensure(Instance *I)
{
  Resident        &R = I->rb;
  const TreeParams sq = big_static_params(I);
  const bool       same = R.launched && memcmp(&I->rb_static, &sq, sizeof sq) == 0;
  if (same && !resident_gone(R)) return 0;
  if (R.launched && !same) big_release(I), I->big_streak = 2; // (parameters changed: a new generation with the new ones)
  if (++I->big_streak < 2) return 1;
  return big_launch(I, sq);
}


} // namespace phyhip_host

