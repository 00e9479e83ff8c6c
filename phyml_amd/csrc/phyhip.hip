// phyhip.hip -- host side of libphyhip.so: instance table, device memory, the deferred operation
// queue and the C ABI declared in include/phyhip.h.  gfx950 only; no CPU fallback: every entry point
// fails with PHYHIP_ERROR_NO_RESOURCE when no device is visible.
#include "phyhip_host.hpp"

namespace phyhip_host
{

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

} // namespace phyhip_host

using namespace phyhip_host;

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
  const bool aa_res_used = I->perm && I->rt.n_cmd > 0;
  resident_free(I->rd);
  resident_free(I->rt);
  big_release(I);
  const bool big_used = I->rb.n_cmd > 0;
  resident_free(I->rb);
  if (I->d_big_stamps && aa_res_used)
  { // resident 20-state evaluator: workgroup 0's phases, mean per command
    unsigned long long h[32];
    if (hipMemcpy(h, I->d_big_stamps, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[7])
    {
      const char *bn[5] = {"operands read", "products", "row sums", "divisions", "stored"};
      for (int k = 0; k < 5; ++k)
        fprintf(stderr, "    inside the matrix rebuild (wave 1, two units), after its start: %-16s %6.2f us\n", bn[k], (double)h[16 + k] * 1e6 / ((double)(I->wall_khz > 0 ? I->wall_khz : 100000) * 1e3) / (double)h[7]);
      const char *in[7] = {"consumer wave 0 starts", "operations done", "evaluation edge's products done", "sums written", "fenced", "first operation's tables in the ring", "loader has published"};
      for (int k = 0; k < 7; ++k)
        fprintf(stderr, "    inside the evaluation, after its start: %-34s %6.2f us\n", in[k], (double)h[8 + k] * 1e6 / ((double)(I->wall_khz > 0 ? I->wall_khz : 100000) * 1e3) / (double)h[7]);
      const char  *names[5] = {"command seen -> parsed", "-> eigen system, exponentials", "-> matrices built", "-> global copies, records", "-> evaluated, fenced, posted"};
      const double us = 1e6 / ((double)(I->wall_khz > 0 ? I->wall_khz : 100000) * 1e3) / (double)h[7];
      for (int k = 0; k < 5; ++k) fprintf(stderr, "  20-state resident, workgroup 0, mean of %llu commands: %-32s %6.2f us\n", h[7], names[k], (double)h[k] * us);
    }
  }
  if (I->d_big_stamps && big_used)
  { // where the last command's time went, per workgroup, relative to workgroup 0 seeing it (wall-clock ticks of 10 ns)
    std::vector<unsigned long long> h((size_t)16 * I->big_wgs);
    if (hipMemcpy(h.data(), I->d_big_stamps, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
    {
      const char *names[6] = {"command seen", "after the barrier", "wave 0 through", "all waves through", "ticket drawn", "final sum posted"};
      const unsigned long long t0 = h[0];
      for (int k = 0; k < 6; ++k)
      {
        double mn = 1e30, mx = -1e30, sum = 0.0; int n = 0;
        for (int w = 0; w < I->big_wgs; ++w)
        {
          const unsigned long long v = h[(size_t)w * 16 + k];
          if (!v || v < t0) continue;
          const double d = (double)(v - t0) * 1e6 / ((double)(I->wall_khz > 0 ? I->wall_khz : 100000) * 1e3);
          mn = std::min(mn, d); mx = std::max(mx, d); sum += d; ++n;
        }
        if (n) fprintf(stderr, "  big resident, last command: %-18s min %7.2f  mean %7.2f  max %7.2f us after workgroup 0 saw it (%d workgroups)\n",
                       names[k], mn, sum / n, mx, n);
      }
      for (int k = 1; k < 6; ++k)
      { // the same per workgroup as a mean over ITS commands, after its own `command seen'
        double mn = 1e30, mx = -1e30, sum = 0.0; int n = 0;
        for (int w = 0; w < I->big_wgs; ++w)
        {
          const unsigned long long c = h[(size_t)w * 16 + 8], v = h[(size_t)w * 16 + 8 + k];
          if (!c || !v) continue;
          const double d = (double)v / (double)c * 1e6 / ((double)(I->wall_khz > 0 ? I->wall_khz : 100000) * 1e3);
          mn = std::min(mn, d); mx = std::max(mx, d); sum += d; ++n;
        }
        if (n) fprintf(stderr, "  big resident, mean of %llu commands: %-18s min %7.2f  mean %7.2f  max %7.2f us after the workgroup saw it (%d workgroups)\n",
                       h[8], names[k], mn, sum / n, mx, n);
      }
    }
  }
  if (I->stream) (void)hipStreamSynchronize(I->stream);
  for (hipEvent_t e : I->prof_spare) (void)hipEventDestroy(e);
  I->prof_spare.clear();
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
  // (src/cl.c:1262-1263 accepts any -c n >= 1; the lane = (pattern, category) kernels hold a pattern's categories in one wave)
  if (categoryCount > kMaxCategories) return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "categoryCount %d > %d", categoryCount, kMaxCategories);
  if ((double)patternCount * categoryCount * stateCount * 8.0 >= 2147483648.0)
    return fail(PHYHIP_ERROR_OUT_OF_RANGE, "one partials buffer must stay below 2 GiB (shard the patterns across devices)");
  // (instances whose buffers can be virtual carry two snapshot slots per internal buffer behind the caller's matrices)
  const bool can_virtualise = categoryCount <= 4 && !(requirementFlags & (PHYHIP_FLAG_CLASS_AXIS | PHYHIP_FLAG_GENERIC_LOOP));
  const int  matrixSlots = matrixBufferCount + (can_virtualise ? 2 * (partialsBufferCount - tipCount) : 0);
  if ((double)matrixSlots * categoryCount * stateCount * stateCount * 8.0 >= 2147483648.0)
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
  I->nmat_all = matrixSlots;
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
  const int id = InstanceTable::add(I);
  if (id < 0)
  {
    release_instance(I);
    return fail(PHYHIP_ERROR_OUT_OF_RANGE, "too many instances");
  }
  return id;
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
  // (round 5, virtual buffers + two wave shapes, docs/history/tools/gpu_r5m.sh: 125 000 patterns 319 / 333, 250 000 650 / 622, 500 000 1275 / 1207 -- the
  // crossover sits between two and four residency rounds of two-lane waves: up to 131 072 patterns)
  if (I->soa && I->C % 2 == 0 && I->Ppad / 64 <= 2048) I->nt_groups = 2;
  if (const char *e = diag_env("PHYHIP_NT_GROUPS"))
  {
    const int g = atoi(e);
    if (g >= 1 && g <= 4 && I->C % g == 0 && 64 % g == 0 && !(I->C == 2 && g == 4)) I->nt_groups = g;
  }
  if (I->soa && I->class_axis) I->nt_groups = I->C; // one class per lane: scaling and evaluation are per class
  I->grid_nt2 = (int)(I->Ppad / (64 / I->nt_groups));
  if (I->soa && I->C == 4 && I->nt_groups == 2 && !I->class_axis && !(diag_env("PHYHIP_NT_MIXED") && atoi(diag_env("PHYHIP_NT_MIXED")) == 0))
  { // two wave shapes for whole-tree traversals (phyhip_nt2.hpp): full rounds of two-lane waves, the rest in four-lane waves --
    // where there is at least one full round and the rest fits one small wave per CU
    const int       cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const long long n2 = (I->P / 32 / cus) * cus, rest = I->P - 32 * n2, n4 = (rest + 15) / 16;
    if (n2 >= cus && rest > 0 && n4 <= cus) { I->mix_n2 = (int)n2; I->mix_n4 = (int)n4; }
  }
  const size_t n_int = (size_t)(I->nbuf - I->tips);
  const size_t be    = buf_elems(I);
  HIPCHK(hipMalloc((void **)&I->d_partials, n_int * be * sizeof(double)));
  HIPCHK(hipMemset(I->d_partials, 0, n_int * be * sizeof(double)));
  HIPCHK(hipMalloc((void **)&I->d_scales, n_int * scale_elems(I) * sizeof(int)));
  HIPCHK(hipMemset(I->d_scales, 0, n_int * scale_elems(I) * sizeof(int)));
  if (I->perm)
  {
    const size_t fb = (size_t)I->nmat_all * kAaMat * sizeof(double);
    HIPCHK(hipMalloc((void **)&I->d_afrag, fb));
    HIPCHK(hipMemset(I->d_afrag, 0, fb));
    // one consumer wave per wave-tile (16 / aa_cb(C) patterns x all categories) + one loader wave per workgroup; enough
    // workgroups to give every CU one before any gets a second (the LDS ring allows one workgroup per CU at a time)
    const long long ntiles = I->Ppad / (16 / aa_cb(I->C));
    const long long cus    = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    I->aa_nw = (int)std::min<long long>(kAaMaxCons, std::max<long long>(1, (ntiles + cus - 1) / cus));
    if (const char *e = diag_env("PHYHIP_AA_NW")) { const int v = atoi(e); if (v >= 1 && v <= kAaMaxCons) I->aa_nw = v; }
    // (diag build, PHYHIP_AA_NT=2: list-form launches with two wave-tiles per consumer wave -- phyhip_aa.hpp: one read of the A
    // operand and one set of per-step scalar work for both, at most 2 x kAaMaxCons2 tiles per workgroup.  Built on the round-5
    // verdict's advice and measured: slower, cfg3 432-435 against 368-378 us, 100 000 patterns 3.25 against 3.05-3.10 ms
    // (profiles/r06_aa_kernel.md) -- five two-tile consumers per CU put four tile-steps on one SIMD where ten one-tile consumers
    // put three)
    // (diag build, PHYHIP_AA_D2=1: list-form launches of workgroups with at most kAaMaxConsD2 consumer waves request their children
    // two operations ahead -- measured: no change, cfg3 366-377 us either way)
    I->aa_d2 = false;
    if (const char *e = diag_env("PHYHIP_AA_D2")) I->aa_d2 = atoi(e) != 0 && I->aa_nw <= kAaMaxConsD2;
    I->aa_nt = 1;
    if (const char *e = diag_env("PHYHIP_AA_NT")) { const int v = atoi(e); if (v == 1 || v == 2) I->aa_nt = v; }
    if (I->aa_nt == 2) I->aa_nw = std::min(I->aa_nw, 2 * kAaMaxCons2);
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
  HIPCHK(hipMalloc((void **)&I->d_pmats, (size_t)I->nmat_all * I->C * I->S * I->S * sizeof(double)));
  HIPCHK(hipMemset(I->d_pmats, 0, (size_t)I->nmat_all * I->C * I->S * I->S * sizeof(double)));
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
  if (const char *e = diag_env("PHYHIP_BIG_ONE_SHOT")) I->big_oneshot = atoi(e) != 0;
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
    if (I->perm) I->prefetch_dist = 2; // (the 20-state kernel loads one operation ahead, but forwards the last TWO results in registers)
  }
  HIPCHK(hipMalloc((void **)&I->d_block,
                   (size_t)2 * std::max(std::max(I->grid, I->grid_nt), std::max(I->grid_aa, std::max(I->grid_nt2, I->mix_n2 + I->mix_n4))) * sizeof(double)));
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
  {
    // Where the host can store straight into device memory (large BAR), the resident evaluators' command records live there
    // and the host pushes each command (phyhip_resident.hip): every workgroup polls locally, nobody relays.
    int large_bar = 0;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, I->dev) != hipSuccess) large_bar = 0;
    I->push_cmds = large_bar ? kPushCmdsDefault : 0;
    if (const char *e = diag_env("PHYHIP_PUSH_CMDS")) I->push_cmds = large_bar ? atoi(e) : 0;
  }

  I->pm_scratch_cap = std::min(std::max(I->nmat, 16), 4096);
  HIPCHK(hipMalloc(&I->d_pmscratch, (size_t)I->pm_scratch_cap * 16 + 64));
  I->ops_cap = 2 * I->nbuf + 8;
  // (a launched list may be longer than the queue: operations that read a virtual buffer get its definition in front of them,
  // at most two per operation -- rewrite_pending)
  // ... and devirtualise() puts up to one storing definition per internal buffer in front of the queue: 3 x (queue + internal
  // buffers) records bound every rewritten list (flush_impl refuses a longer one instead of overrunning the slot)
  I->ops_slot_bytes = (size_t)(3 * (I->ops_cap + (I->nbuf - I->tips)) + 2) * (sizeof(IssueRec) + sizeof(ExecRec));
  I->virt.assign(I->nbuf, 0);
  I->keep_real_flag.assign(I->nbuf, 0);
  I->vdef.assign(I->nbuf, DevOp{0, 0, 0, 0, 0, 0});
  HIPCHK(hipMalloc((void **)&I->d_ops, (size_t)I->ops_slots * I->ops_slot_bytes));
  size_t chunk = std::max<size_t>(64 * 1024, std::max(I->ops_slot_bytes,
                                                       (size_t)I->C * I->S * I->S * sizeof(double) * 4));
  rc = I->ring.init(chunk);
  if (rc) return rc;
  I->ring.before_rotate = [I]() { return I->up_idx.empty() ? 0 : flush_uploads(I); };
  I->mat_in_queue.assign(I->nmat_all, 0);
  I->up_slot.assign(I->nmat, -1);
  I->pm_slot.assign(I->nmat, -1);
  I->slot_ops.assign(I->ops_slots, std::vector<DevOp>());
  I->slot_inl.assign(I->ops_slots, std::vector<InlineDef>());
  if (const char *e = diag_env("PHYHIP_VIRT_INLINE")) I->virt_inline = atoi(e) != 0;
  if (const char *e = diag_env("PHYHIP_VIRT_MIN_OPS")) I->virt_min_ops = std::max(0, atoi(e)); // (diag: phyhip_set_virtual_buffers' threshold from the environment)
  I->slot_kind.assign(I->ops_slots, -1);
  if (const char *e = diag_env("PHYHIP_GENERIC_NT")) I->generic_nt = atoi(e) != 0;
  if (I->generic_loop) I->generic_nt = true;
  if (I->C > 8) I->generic_nt = true; // (9 .. 64 categories: the plain lane = (pattern, category) kernel for either state count)
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
  // what is queued runs on the old row -- and may leave tip x tip results virtual; those, and the ones that already are, are
  // stored before the row changes
  int rc = flush(I, nullptr);
  if (rc) return rc;
  devirtualise_tip(I, tip);
  if ((rc = flush_sync(I))) return rc;
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
  devirtualise_tip(I, tipIndex);           // ... and virtual buffers defined on it are stored before it changes
  if ((rc = flush(I, nullptr))) return rc;
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
  devirtualise(I, bufferIndex); // (the scale vector of the last update stays what the reference has)
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
  ++I->model_epoch;
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
  if (I->apply_scaling != sc)
  { // (the rescaling rule is part of a virtual buffer's definition)
    devirtualise_all(I);
    if ((rc = flush(I, nullptr))) return rc;
  }
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

// Called before the matrices idx[0..count) change.  shadow != nullptr: the caller's rebuild / upload can move a matrix's old value
// into a snapshot slot first ([count] slots, filled here; -1: none) -- whole-tree batches of device-built matrices and uploads.
static int matrices_touch(Instance *I, const int *idx, int count, std::vector<int> *shadow = nullptr)
{
  for (int i = 0; i < count; ++i)
    if (idx[i] < 0 || idx[i] >= I->nmat) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "matrix index %d", idx[i]);
  bool queued_reader = false;
  for (int i = 0; i < count && !queued_reader; ++i) queued_reader = I->mat_in_queue[idx[i]] != 0;
  // does a virtual buffer's definition read one of them?  (a scan of the buffer flags: only while some are virtual)
  auto depends = [&]() {
    for (int b = I->tips; b < I->nbuf && I->n_virtual > 0; ++b)
      if (I->virt[b])
        for (int i = 0; i < count; ++i)
          if (I->vdef[b].pm1 == idx[i] || I->vdef[b].pm2 == idx[i]) return true;
    return false;
  };
  // a queued operation still reads an old matrix -- or may read a virtual buffer through a definition that is about to move to
  // a snapshot slot: launch the queue first (stream order does the rest).  That launch may itself leave tip x tip results
  // virtual that are defined on the old values: they are looked at below, after it.
  if (queued_reader || (shadow && !I->pending.empty() && depends()))
  {
    int rc = flush(I, nullptr);
    if (rc) return rc;
  }
  if (I->n_virtual == 0) return 0;
  const unsigned long long stored_before = I->n_virt_material;
  // A virtual buffer is defined on the matrices as they are.  Before one changes: pmat_kernel / upload_matrices_kernel move the
  // old value into the buffer's own snapshot slot, and the definition reads it there from now on (a full traversal that follows
  // recomputes every buffer anyway; the host route rewrites a tree's matrices one call at a time) -- or the buffer is stored first.
  if (shadow)
  {
    std::vector<int> where; // matrix -> position in idx (the last one wins, as in the queue); short lists are searched instead
    if (count > 4)
    {
      where.assign(I->nmat, -1);
      for (int i = 0; i < count; ++i) where[idx[i]] = i;
    }
    auto pos = [&](int pm) {
      if (pm >= I->nmat) return -1; // (already a snapshot slot)
      if (count > 4) return where[pm];
      for (int i = count - 1; i >= 0; --i)
        if (idx[i] == pm) return i;
      return -1;
    };
    for (int b = I->tips; b < I->nbuf && I->n_virtual > 0; ++b)
    {
      if (!I->virt[b]) continue;
      DevOp &d = I->vdef[b];
      const int  i1 = pos(d.pm1), i2 = pos(d.pm2);
      if (i1 < 0 && i2 < 0) continue;
      if (shadow->empty()) shadow->assign(count, -1);
      // (one snapshot per matrix and call; a second dependant of a matrix, or a definition that reads it twice, is stored instead)
      const bool free1 = i1 < 0 || ((*shadow)[i1] < 0 && I->pm_slot[d.pm1] < 0 && I->up_slot[d.pm1] < 0);
      const bool free2 = i2 < 0 || ((*shadow)[i2] < 0 && I->pm_slot[d.pm2] < 0 && I->up_slot[d.pm2] < 0);
      if (!free1 || !free2 || (i1 >= 0 && i2 >= 0 && d.pm1 == d.pm2)) { devirtualise(I, b); continue; }
      if (i1 >= 0) { (*shadow)[i1] = shadow_slot(I, b, 0); d.pm1 = shadow_slot(I, b, 0); }
      if (i2 >= 0) { (*shadow)[i2] = shadow_slot(I, b, 1); d.pm2 = shadow_slot(I, b, 1); }
    }
  }
  else
    for (int i = 0; i < count && I->n_virtual > 0; ++i) devirtualise_matrix(I, idx[i]);
  if (I->n_virt_material != stored_before)
  { // (stored on the old values: now)
    int rc = flush(I, nullptr);
    if (rc) return rc;
  }
  return 0;
}

int phyhip_update_transition_matrices(int instance, int eigenIndex, const int *probabilityIndices,
                                      const int *firstDerivativeIndices, const int *secondDerivativeIndices,
                                      const double *edgeLengths, int count)
{
  if (Group *G = get_group_nodrain(instance))
  {
    // a short list on a group with helper threads is only recorded (Group::deferred): validated here as the shards would
    if (G->defers() && count > 0 && count < kEagerPmBatch && eigenIndex == 0 && !firstDerivativeIndices && !secondDerivativeIndices && probabilityIndices && edgeLengths)
    {
      for (int i = 0; i < count; ++i)
        if (probabilityIndices[i] < 0 || probabilityIndices[i] >= G->nmat) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "matrix index %d", probabilityIndices[i]);
      GroupDeferred d;
      d.kind = 1; d.idx.assign(probabilityIndices, probabilityIndices + count); d.val.assign(edgeLengths, edgeLengths + count);
      G->deferred.push_back(std::move(d));
      if (G->deferred.size() >= 4096 && group_drain(G)) return group_take_drain_error(G); // (a caller that only ever queues)
      return PHYHIP_SUCCESS;
    }
    int rc = group_drain(G);
    if (rc) return group_take_drain_error(G);
    // (a whole-tree batch launches the rebuild at once: on the shard's helper thread)
    return group_parallel(G, [&](int g) {
      return phyhip_update_transition_matrices(G->sub_id[g], eigenIndex, probabilityIndices, firstDerivativeIndices,
                                               secondDerivativeIndices, edgeLengths, count);
    });
  }
  GET_INST_RES(I, instance);
  if (eigenIndex != 0) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "eigenIndex must be 0");
  if (firstDerivativeIndices || secondDerivativeIndices)
    return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "derivative matrices are not used by PhyML's path (see phyhip_calculate_eigen_lnl_dlnl)");
  if (count <= 0) return PHYHIP_SUCCESS;
  std::vector<int> shadow;
  const bool batch = count >= kEagerPmBatch;
  int rc = matrices_touch(I, probabilityIndices, count, batch ? &shadow : nullptr);
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
      I->pm_shadow.push_back(-1);
    }
    if (!shadow.empty() && shadow[i] >= 0) { I->pm_shadow[I->pm_slot[m]] = shadow[i]; ++I->n_pm_shadow; }
  }
  // A whole-tree batch (Update_All_PMat, src/lk.c:500-512) is launched now rather than with the traversal: the device
  // rebuilds the matrices while the host walks the tree and fills the operation list.
  if (count >= kEagerPmBatch && I->eager_pmats) return flush_pmats(I);
  I_call.leave_queued_only();
  return PHYHIP_SUCCESS;
}

int phyhip_set_transition_matrix(int instance, int matrixIndex, const double *inMatrix, double paddedValue)
{
  if (Group *G = get_group_nodrain(instance))
  {
    if (G->defers() && inMatrix)
    {
      if (matrixIndex < 0 || matrixIndex >= G->nmat) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "matrix index %d", matrixIndex);
      GroupDeferred d;
      d.kind = 2; d.idx.assign(1, matrixIndex); d.val.assign(inMatrix, inMatrix + (size_t)G->C * G->S * G->S);
      G->deferred.push_back(std::move(d));
      if (G->deferred.size() >= 4096 && group_drain(G)) return group_take_drain_error(G); // (a caller that only ever queues)
      return PHYHIP_SUCCESS;
    }
    int rc = group_drain(G);
    if (rc) return group_take_drain_error(G);
    return group_each(G, [&](int id, long long, long long) { return phyhip_set_transition_matrix(id, matrixIndex, inMatrix, paddedValue); });
  }
  (void)paddedValue;
  GET_INST_RES(I, instance);
  // (a virtual buffer defined on the old value: the upload kernel moves that into the buffer's snapshot slot first -- the host
  // route rewrites a whole tree's matrices one call at a time, and a stored buffer per call would be a launch per call)
  std::vector<int> shadow;
  int rc = matrices_touch(I, &matrixIndex, 1, &shadow);
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
    I->up_shadow.push_back(-1);
  }
  if (!shadow.empty() && shadow[0] >= 0) { I->up_shadow[I->up_slot[matrixIndex]] = shadow[0]; ++I->n_up_shadow; }
  if ((int)I->up_idx.size() >= 4 * kUploadBatch && (rc = flush_uploads(I))) return rc;
  I_call.leave_queued_only(); // (a flush above -- flush_pmats, flush_uploads, also through the staging ring's rotation -- says so itself)
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
  if (Group *G = get_group_nodrain(instance))
  {
    if (G->defers() && n > 0 && ops)
    { // (any list: a whole tree's is then queued on the shards side by side, by their helper threads) -- validated here as the shards would
      for (int i = 0; i < n; ++i)
      {
        const phyhip_operation &o = ops[i];
        if (o.destinationPartials < G->tips || o.destinationPartials >= G->nbuf || o.child1Partials < 0 || o.child1Partials >= G->nbuf ||
            o.child2Partials < 0 || o.child2Partials >= G->nbuf)
          return fail(PHYHIP_ERROR_OUT_OF_RANGE, "operation %d: partials buffer index out of range", i);
        if (o.child1TransitionMatrix < 0 || o.child1TransitionMatrix >= G->nmat || o.child2TransitionMatrix < 0 || o.child2TransitionMatrix >= G->nmat)
          return fail(PHYHIP_ERROR_OUT_OF_RANGE, "operation %d: matrix index out of range", i);
        if (o.destinationPartials == o.child1Partials || o.destinationPartials == o.child2Partials)
          return fail(PHYHIP_ERROR_OUT_OF_RANGE, "operation %d: buffer %d is both destination and child", i, o.destinationPartials);
      }
      GroupDeferred d;
      d.kind = 0; d.ops.assign(ops, ops + n);
      G->deferred.push_back(std::move(d));
      G->deferred_ops += (size_t)n;
      if (G->deferred.size() >= 4096 && group_drain(G)) return group_take_drain_error(G); // (a caller that only ever queues)
      return PHYHIP_SUCCESS;
    }
    int rc = group_drain(G);
    if (rc) return group_take_drain_error(G);
    return group_each(G, [&](int id, long long, long long) { return phyhip_update_partials(id, ops, n, cumulativeScaleIndex); });
  }
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
    // (Update_Partial_Lk writes one side of an edge from the two OTHER edges at that node, src/lk.c:1282: never in place.  The
    // pipelined kernels rely on it: an odd list runs its last operation twice, which must give the same result)
    if (o.destinationPartials == o.child1Partials || o.destinationPartials == o.child2Partials)
      return fail(PHYHIP_ERROR_OUT_OF_RANGE, "operation %d: buffer %d is both destination and child", i, o.destinationPartials);
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

int phyhip_calculate_edge_log_likelihoods(int instance, const int *parent, const int *child, const int *pm, const int *d1,
                                          const int *d2, const int *cw, const int *sf, const int *cs, int count,
                                          double *outSum, double *outD1, double *outD2)
{
  (void)cs;
  Group *G = get_group_nodrain(instance);
  if (G && count == 1 && group_combines_on_host(G, true))
  { // every shard answers as a plain instance (its resident evaluators included), the shard sums are added here in shard order;
    // the queue-only calls recorded since the last evaluation (Group::deferred) are replayed by the same job
    std::vector<double> part(G->sub.size(), 0.0);
    std::vector<int>    warn(G->sub.size(), 0);
    int rc = group_parallel(G, [&](int g) -> int {
      int r = group_replay(G, g);
      if (r) return r;
      r = phyhip_calculate_edge_log_likelihoods(G->sub_id[g], parent, child, pm, d1, d2, cw, sf, cs, count, &part[g], outD1, outD2);
      return r ? r : phyhip_get_numerical_warning(G->sub_id[g], &warn[g]);
    });
    G->deferred.clear(); G->deferred_ops = 0;
    if (rc) return rc;
    double sum = 0.0;
    for (double v : part) sum += v;
    *outSum = sum;
    return group_collect_warning(G, warn);
  }
  if (G && group_drain(G)) return group_take_drain_error(G); // (the collective route: the shards' queues first)
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
  devirtualise(I, bufferIndex);
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
  devirtualise(I, bufferIndex);
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
  devirtualise(I, bufferIndex);
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
  devirtualise(I, bufferIndex);
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
  while (I->prof && I->prof_spare.size() < 256)
  { // a supply of events for the launches to come (flush_impl)
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    I->prof_spare.push_back(e);
  }
  I->prof_ms = 0.0; I->prof_n = 0; I->prof_updates = 0.0; I->prof_rd_bytes = 0.0; I->prof_wr_bytes = 0.0; I->prof_kernel[0] = 0;
  for (int k = 0; k < 3; ++k) { I->prof_aux_ms[k] = 0.0; I->prof_aux_n[k] = 0; }
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
  { // (sharded instances: what their shards' evaluators served -- host-combined short calls, Group::host_combine)
    for (int k = 0; k < 8; ++k) out[k] = 0;
    for (int id : G->sub_id)
    {
      long long o[8];
      const int rc = phyhip_get_resident_stats(id, o);
      if (rc) return rc;
      for (int k = 0; k < 8; ++k) out[k] += o[k];
    }
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
  if (Group *G = get_group(instance))
  {
    for (int id : G->sub_id)
    {
      long long o[4];
      const int rc = phyhip_get_big_resident_stats(id, o);
      if (rc) return rc;
      for (int k = 0; k < 4; ++k) out[k] += o[k];
    }
    return PHYHIP_SUCCESS;
  }
  GET_INST_RES(I, instance);
  I_call.leave_query();
  const Resident *R = &I->rb;
  out[0] = (long long)R->n_cmd; out[1] = (long long)R->n_launch; out[2] = (long long)R->n_silent; out[3] = (long long)R->n_busy;
  return PHYHIP_SUCCESS;
}

int phyhip_set_virtual_buffers(int instance, int minOperations)
{
  if (Group *G = get_group(instance)) return group_each(G, [&](int id, long long, long long) { return phyhip_set_virtual_buffers(id, minOperations); });
  GET_INST(I, instance);
  if (minOperations < 0) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "minOperations %d", minOperations);
  I->virt_min_ops = minOperations;
  if (minOperations == 0 && I->n_virtual > 0)
  {
    devirtualise_all(I);
    return flush(I, nullptr);
  }
  return PHYHIP_SUCCESS;
}

int phyhip_get_virtual_stats(int instance, long long out[4])
{
  if (Group *G = get_group(instance)) return phyhip_get_virtual_stats(G->sub_id[0], out);
  GET_INST_RES(I, instance);
  I_call.leave_query();
  out[0] = I->n_virtual; out[1] = (long long)I->n_virt_skipped; out[2] = (long long)I->n_virt_recomputed; out[3] = (long long)I->n_virt_material;
  return PHYHIP_SUCCESS;
}

int phyhip_profile_read_collective(int instance, double *outMs, int *outCount, int *outRanks)
{
  Group *G = get_group(instance);
  GET_INST(I, G ? G->sub_id[0] : instance);
  int rc = flush_sync(I);
  if (rc) return rc;
  if ((rc = collect_profile(I))) return rc;
  if (outMs) *outMs = I->prof_aux_ms[2];
  if (outCount) *outCount = I->prof_aux_n[2];
  const Collective *co = G ? G->co : I->co;
  if (outRanks) *outRanks = co ? co->nranks : 1;
  return PHYHIP_SUCCESS;
}

int phyhip_profile_read_kernel(int instance, char *outName, int capacity)
{
  Group *G = get_group(instance);
  GET_INST_RES(I, G ? G->sub_id[0] : instance);
  I_call.leave_query();
  if (!outName || capacity < 1) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "no room for the kernel name");
  snprintf(outName, (size_t)capacity, "%s", I->prof_kernel);
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
