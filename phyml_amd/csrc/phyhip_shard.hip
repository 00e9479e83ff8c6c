// phyhip_shard.hip -- the multi-GPU side: pattern shards, the one collective (see phyhip_shard.hpp)
// (libphyhip.so, gfx950 only; the units and what they share: phyhip_host.hpp)
#include "phyhip_host.hpp"

namespace phyhip_host
{

// slot 0 += slots 1..n of one device, fixed order (several shards on one device: tests on a single-GPU box, or more
// shards than devices)
__global__ void shard_local_sum_kernel(double *slots, int n, int count)
{
  const int t = threadIdx.x;
  if (t >= count) return;
  double v = 0.0;
  for (int k = 1; k <= n; ++k) v += slots[k * kRedStride + t];
  slots[t] = v;
}

// after the all-reduce: {warning, lnL, dlnL} -> host-mapped result block + sequence number (what the fused final sum
// of a single-device evaluation does itself)
__global__ void shard_publish_kernel(const double *red, double *result_host, int *warn_host, unsigned long long seq)
{
  if (threadIdx.x != 0) return;
  // (written through at system scope and acknowledged, then the sequence number: the order a release fence gives without
  // its write-back of the L2 -- see finish_sums)
  unsigned long long b1, b2;
  const double       r1 = red[1], r2 = red[2];
  __builtin_memcpy(&b1, &r1, 8);
  __builtin_memcpy(&b2, &r2, 8);
  __hip_atomic_store(warn_host, red[0] != 0.0 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(result_host), b1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(result_host + 1), b2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_s_waitcnt(0);
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(result_host + 2), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Per-device local sums, ONE all-reduce (count doubles) on every device's stream, result of the first device to the
// host through I0's result block.  I0 must live on ctx[0] and run on its stream.
int reduce_and_publish(Collective &co, int count, Instance *I0)
{
  int rc = 0;
  hipEvent_t pa = nullptr, pb = nullptr; // (profiled instance: HIP events around the whole collective path on the first device's stream)
  if (I0->prof)
  {
    if ((rc = set_dev(co.ctx[0].dev))) return rc;
    if (hipEventCreate(&pa) != hipSuccess || hipEventCreate(&pb) != hipSuccess) pa = pb = nullptr;
    if (pa) (void)hipEventRecord(pa, co.ctx[0].stream);
  }
  for (auto &c : co.ctx)
    if (c.nsub > 1)
    {
      if ((rc = set_dev(c.dev))) return rc;
      hipLaunchKernelGGL(shard_local_sum_kernel, dim3(1), dim3(64), 0, c.stream, c.d_red, c.nsub, count);
      HIPCHK(hipGetLastError());
    }
  const bool grouped = co.ctx.size() > 1;
  if (grouped) NCCLCHK(ncclGroupStart());
  for (auto &c : co.ctx)
  {
    if ((rc = set_dev(c.dev))) return rc;
    NCCLCHK(ncclAllReduce(c.d_red, c.d_red, (size_t)count, ncclDouble, ncclSum, c.comm, c.stream));
  }
  if (grouped) NCCLCHK(ncclGroupEnd());
  if ((rc = set_dev(co.ctx[0].dev))) return rc;
  hipLaunchKernelGGL(shard_publish_kernel, dim3(1), dim3(64), 0, co.ctx[0].stream, (const double *)co.ctx[0].d_red, I0->h_result,
                     I0->h_warn, ++I0->seq);
  HIPCHK(hipGetLastError());
  if (pa)
  {
    (void)hipEventRecord(pb, co.ctx[0].stream);
    I0->prof_aux.push_back({pa, pb, 2});
  }
  return wait_result(I0);
}

// One all-reduce of zeros on every communicator of `co`, waited for: RCCL builds its channels, proxies and kernels on the
// FIRST collective of a communicator (tens of milliseconds) -- that belongs to communicator creation, not to the caller's
// first likelihood evaluation.
int warm_up_collective(Collective &co)
{
  int        rc      = 0;
  const bool grouped = co.ctx.size() > 1;
  if (grouped) NCCLCHK(ncclGroupStart());
  for (auto &c : co.ctx)
  {
    if ((rc = set_dev(c.dev))) return rc;
    NCCLCHK(ncclAllReduce(c.d_red, c.d_red, (size_t)kRedStride, ncclDouble, ncclSum, c.comm, c.stream));
  }
  if (grouped) NCCLCHK(ncclGroupEnd());
  for (auto &c : co.ctx)
  {
    if ((rc = set_dev(c.dev))) return rc;
    HIPCHK(hipStreamSynchronize(c.stream));
  }
  return 0;
}

void release_collective(Collective *co)
{
  if (!co) return;
  for (auto &c : co->ctx)
  {
    (void)hipSetDevice(c.dev);
    g_cur_dev = c.dev;
    if (c.stream) (void)hipStreamSynchronize(c.stream);
    if (c.comm && co->own_comms) (void)ncclCommDestroy(c.comm);
    if (c.d_red) (void)hipFree(c.d_red);
  }
  delete co;
}

std::mutex           g_groups_mu;
std::vector<Group *> g_groups;

Group *get_group_nodrain(int id)
{
  if (id < kGroupBase) return nullptr;
  std::lock_guard<std::mutex> lk(g_groups_mu);
  const int k = id - kGroupBase;
  if (k >= (int)g_groups.size()) return nullptr;
  return g_groups[k];
}
Group *get_group(int id)
{
  Group *G = get_group_nodrain(id);
  if (G && !G->deferred.empty()) (void)group_drain(G); // (a failure is kept in Group::drain_rc for the caller's group_each / group_parallel)
  return G;
}

// the recorded queue-only calls, on shard g through its plain entry points (its helper thread, or the calling thread)
int group_replay(Group *G, int g)
{
  const int id = G->sub_id[g];
  for (const GroupDeferred &d : G->deferred)
  {
    int rc = PHYHIP_SUCCESS;
    if (d.kind == 0) rc = phyhip_update_partials(id, d.ops.data(), (int)d.ops.size(), 0);
    else if (d.kind == 1) rc = phyhip_update_transition_matrices(id, 0, d.idx.data(), nullptr, nullptr, d.val.data(), (int)d.idx.size());
    else rc = phyhip_set_transition_matrix(id, d.idx[0], d.val.data(), 0.0);
    if (rc) return rc;
  }
  return PHYHIP_SUCCESS;
}

int group_drain(Group *G)
{
  if (G->deferred.empty()) return PHYHIP_SUCCESS;
  const int rc = group_parallel(G, [&](int g) -> int { return group_replay(G, g); });
  G->deferred.clear();
  G->deferred_ops = 0;
  if (rc) { G->drain_rc = rc; G->drain_err = g_err; }
  return rc;
}
void forget_group(int id)
{
  std::lock_guard<std::mutex> lk(g_groups_mu);
  g_groups[id - kGroupBase] = nullptr;
}

void release_group(Group *G)
{
  for (ShardWorker *w : G->workers)
  {
    w->stop();
    delete w;
  }
  G->workers.clear();
  release_collective(G->co); // drains every device's stream and destroys the communicators while the streams still exist
  G->co = nullptr;
  for (int id : G->sub_id)
    if (id >= 0)
    {
      Instance *I = InstanceTable::wiring(id);
      if (I && !I->own_stream) I->stream = nullptr; // shared stream: owned by the device's first shard
      (void)phyhip_finalize_instance(id);
    }
  delete G;
}

// the sharded class instances of one mixture: same shard layout, same devices (mixtures on sharded instances)
int mixture_groups(const int *instances, int count, std::vector<Group *> &Gs)
{
  if (count > kMaxMixClasses) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "1..%d mixture classes", kMaxMixClasses);
  Gs.clear();
  for (int k = 0; k < count; ++k)
  {
    Group *G = get_group(instances[k]);
    if (!G) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "mixture class %d: sharded and plain class instances cannot be mixed", k);
    if (k > 0 && (G->n != Gs[0]->n || G->ctx_of != Gs[0]->ctx_of || G->k_of != Gs[0]->k_of))
      return fail(PHYHIP_ERROR_OUT_OF_RANGE, "mixture class %d: another shard layout than class 0", k);
    Gs.push_back(G);
  }
  return 0;
}

int create_group(int tipCount, int partialsBufferCount, int stateCount, int patternCount, int matrixBufferCount,
                        int categoryCount, const int *resourceList, int resourceCount, phyhip_instance_details *returnInfo,
                        long classAxisFlag)
{
  if (patternCount < resourceCount)
    return fail(PHYHIP_ERROR_OUT_OF_RANGE, "%d patterns cannot be sharded over %d devices", patternCount, resourceCount);
  Group *G = new Group();
  G->S = stateCount; G->C = categoryCount; G->tips = tipCount; G->nbuf = partialsBufferCount; G->nmat = matrixBufferCount;
  G->P = patternCount;
  G->co = new Collective();
  const long long base = patternCount / resourceCount, rem = patternCount % resourceCount;
  for (int g = 0; g < resourceCount; ++g)
  {
    const long long lo = g * base + std::min<long long>(g, rem), n = base + (g < rem ? 1 : 0);
    const int       dev = resourceList[g];
    phyhip_instance_details det;
    const int id = phyhip_create_instance(tipCount, partialsBufferCount, 0, stateCount, (int)n, 1, matrixBufferCount, categoryCount,
                                          0, &dev, 1, 0, classAxisFlag, &det);
    if (id < 0)
    {
      release_group(G);
      return id;
    }
    if (g == 0 && returnInfo) *returnInfo = det;
    Instance *I = InstanceTable::wiring(id);
    int ci = -1;
    for (size_t k = 0; k < G->co->ctx.size(); ++k)
      if (G->co->ctx[k].dev == dev) ci = (int)k;
    if (ci < 0)
    {
      DevCtx c;
      c.dev = dev; c.stream = I->stream;
      G->co->ctx.push_back(c);
      ci = (int)G->co->ctx.size() - 1;
    }
    else
    { // a second shard on a device runs on the first one's stream: the local sum is then ordered by the stream alone
      (void)hipStreamDestroy(I->stream);
      I->stream = G->co->ctx[ci].stream;
      I->own_stream = false;
    }
    G->sub_id.push_back(id); G->sub.push_back(I); G->lo.push_back(lo); G->n.push_back(n);
    G->ctx_of.push_back(ci); G->k_of.push_back(G->co->ctx[ci].nsub++);
  }
  const int    nctx = (int)G->co->ctx.size();
  std::vector<int>        devs(nctx);
  std::vector<ncclComm_t> comms(nctx);
  for (int k = 0; k < nctx; ++k) devs[k] = G->co->ctx[k].dev;
  {
    const ncclResult_t r = ncclCommInitAll(comms.data(), nctx, devs.data());
    if (r != ncclSuccess)
    {
      release_group(G);
      return fail(PHYHIP_ERROR_GENERAL, "ncclCommInitAll over %d device(s) failed: %s", nctx, ncclGetErrorString(r));
    }
  }
  G->co->nranks = nctx;
  for (int k = 0; k < nctx; ++k) G->co->ctx[k].comm = comms[k]; // (all of them first: release_group destroys what ctx holds)
  for (int k = 0; k < nctx; ++k)
  {
    DevCtx &c = G->co->ctx[k];
    hipError_t e = hipSetDevice(c.dev);
    g_cur_dev = c.dev;
    const size_t bytes = sizeof(double) * kRedStride * (size_t)(1 + c.nsub);
    if (e == hipSuccess) e = hipMalloc((void **)&c.d_red, bytes);
    if (e == hipSuccess) e = hipMemset(c.d_red, 0, bytes);
    if (e != hipSuccess)
    {
      release_group(G);
      return fail(PHYHIP_ERROR_OUT_OF_MEMORY, "reduction buffer: %s", hipGetErrorString(e));
    }
  }
  if (warm_up_collective(*G->co))
  {
    release_group(G);
    return PHYHIP_ERROR_GENERAL;
  }
  if (const char *e = getenv("PHYHIP_SHARD_HOST_COMBINE")) G->host_combine = atoi(e);
  {
    const char *e = getenv("PHYHIP_SHARD_THREADS");
    if (e ? atoi(e) != 0 : nctx > 1)
      for (int g = 0; g < resourceCount; ++g)
      {
        ShardWorker *w = new ShardWorker();
        w->th = std::thread([w] { w->run(); });
        G->workers.push_back(w);
      }
  }
  std::lock_guard<std::mutex> lk(g_groups_mu);
  for (size_t i = 0; i < g_groups.size(); ++i)
    if (!g_groups[i])
    {
      g_groups[i] = G;
      return kGroupBase + (int)i;
    }
  g_groups.push_back(G);
  return kGroupBase + (int)g_groups.size() - 1;
}

// Does this evaluation go shard by shard through the plain entry points, the calling thread adding the shard sums
// (Group::host_combine)?  queue_counts: an edge evaluation -- short when no shard has more than two operations queued (what the
// resident evaluators and the records-in-the-arguments launches take).
bool group_combines_on_host(const Group *G, bool queue_counts)
{
  if (G->host_combine <= 0) return false;
  if (G->host_combine >= 2) return true;
  // (shards that share a device -- a test configuration -- would take the large-grid resident workgroups from each other, one
  // set fills the device: 1.9 ms instead of 0.18 per SPR candidate at 2 x 100 000 patterns on one GPU, measured)
  if (G->co->ctx.size() < G->sub.size() && big_shape(G->sub[0])) return false;
  if (!queue_counts) return true;
  for (const Instance *I : G->sub)
    if (I->pending.size() + G->deferred_ops > 2) return false;
  return true;
}

// after a host-combined evaluation: the warning flag of src/lk.c:847-851 over all shards -- each shard's own last evaluation, read
// by the thread that ran it (its device is current there: from the calling thread every shard would cost a device switch)
int group_collect_warning(Group *G, const std::vector<int> &shard_warn)
{
  int w = 0;
  for (int ws : shard_warn) w = std::max(w, ws);
  G->last_warn  = w;
  G->warn_valid = true;
  return PHYHIP_SUCCESS;
}

// Lk(b) / Lk(NULL) on a sharded instance: every shard's traversal + edge evaluation (no host synchronisation), then the
// collective.  Launches go out shard by shard from the one host thread; the devices run concurrently.
int group_edge_lnl(Group *G, int parent, int child, int pm, double *out)
{
  int rc = group_parallel(G, [&](int g) -> int {
    Instance *I = G->sub[g];
    int r;
    if ((r = check_partial_index(I, parent, true)) || (r = check_partial_index(I, child, true))) return r;
    if (pm < 0 || pm >= I->nmat) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "matrix index %d", pm);
    double  *slot = shard_slot(G->co->ctx[G->ctx_of[g]], G->k_of[g]);
    EdgeEval ee{parent, child, pm, slot + 1, false, slot};
    return flush(I, &ee);
  });
  if (rc) return rc;
  rc = reduce_and_publish(*G->co, 2, G->sub[0]);
  if (rc) return rc;
  *out          = G->sub[0]->h_result[0];
  G->last_warn  = *G->sub[0]->h_warn;
  G->warn_valid = true;
  return PHYHIP_SUCCESS;
}


} // namespace phyhip_host

using namespace phyhip_host;

extern "C" {

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

} // extern "C"
