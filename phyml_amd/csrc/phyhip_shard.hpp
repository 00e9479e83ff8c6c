// phyhip_shard.hpp -- the multi-GPU side of libphyhip.so: pattern shards and the ONE
// collective of the path, an RCCL all-reduce of the per-shard log-likelihood (src/lk.c:856: c_lnL is a plain sum over
// patterns; src/lk.c:744-745: so are c_lnL and c_dlnL of dLk).  Included by phyhip.hip inside its anonymous namespace.
//
// Two ways in, one mechanism:
//   * single process, G devices (what a C host like PhyML uses): phyhip_create_instance with a resource list of G
//     devices returns ONE instance id backed by G per-device instances over contiguous pattern ranges
//     [g*P/G, (g+1)*P/G) and one communicator per device from ncclCommInitAll.  Setters slice per-pattern inputs and
//     replicate the model; getters concatenate; an evaluation launches every shard, all-reduces on every device's stream
//     inside one ncclGroupStart/End, and hands device 0's copy to the host.
//   * one process per GPU (MPI-style hosts, bench.py under torchrun): phyhip_comm_init_rank attaches a communicator built
//     from a broadcast ncclUniqueId to a plain instance holding that rank's shard; the same evaluation entry points then
//     return the all-reduced value on every rank.
//
// The reduced vector is {numerical warning, lnL[, dlnL]}: count 2 for Lk, 3 for dLk -- the warning flag of
// src/lk.c:847-851 rides as one more double so that no second collective or extra synchronisation exists.
// The payload is 16-24 bytes: the collective is pure latency; the xGMI links' bandwidth never matters here.
#pragma once

// (<rccl/rccl.h> is included by phyhip.hip at global scope)

#define NCCLCHK(call)                                                                                        \
  do                                                                                                         \
  {                                                                                                          \
    ncclResult_t r_ = (call);                                                                                \
    if (r_ != ncclSuccess)                                                                                   \
      return fail(PHYHIP_ERROR_GENERAL, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

constexpr int kRedStride = 4; // doubles per reduction slot: {warning, lnL, dlnL, pad}

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

struct DevCtx
{
  int         dev    = 0;
  hipStream_t stream = nullptr;    // the stream every shard on this device runs on (= its first shard's own stream)
  ncclComm_t  comm   = nullptr;
  double     *d_red  = nullptr;    // [(1 + nsub)][kRedStride]: slot 0 is the all-reduce buffer
  int         nsub   = 0;
};

struct Collective
{
  std::vector<DevCtx> ctx;
  int                 nranks = 0; // communicator size (all processes)
  bool                own_comms = true;
};

static int set_dev(int dev) { return make_current(dev); }

// where shard number `k` of its device writes {warning, lnL, dlnL}
static double *shard_slot(const DevCtx &c, int k) { return c.d_red + (size_t)(c.nsub > 1 ? k + 1 : 0) * kRedStride; }

// Per-device local sums, ONE all-reduce (count doubles) on every device's stream, result of the first device to the
// host through I0's result block.  I0 must live on ctx[0] and run on its stream.
static int reduce_and_publish(Collective &co, int count, Instance *I0)
{
  int rc = 0;
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
  return wait_result(I0);
}

// One all-reduce of zeros on every communicator of `co`, waited for: RCCL builds its channels, proxies and kernels on the
// FIRST collective of a communicator (tens of milliseconds) -- that belongs to communicator creation, not to the caller's
// first likelihood evaluation.
static int warm_up_collective(Collective &co)
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

static void release_collective(Collective *co)
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

// ---- single process, several devices ----------------------------------------------------------------------------

// One helper thread per shard: a launch costs the calling thread 5-10 us, so eight devices driven by PhyML's single
// thread would start their traversals up to ~100 us apart (a quarter of a 125 000-pattern shard's run time).  The main
// thread posts the same job to every helper, each issues its shard's launches on its own device, the main thread joins
// them and then issues the collective.  Helpers spin for ~0.1 ms after a job (evaluations come in bursts) and sleep on a
// condition variable otherwise.
struct ShardWorker
{
  std::thread                    th;
  std::mutex                     m;
  std::condition_variable        cv;
  std::atomic<int>               state{0}; // 0 idle, 1 job posted, 2 job done
  std::atomic<bool>              quit{false};
  std::function<int()>           job;
  int                            rc = 0;
  std::string                    err;

  void run()
  {
    for (;;)
    {
      int spins = 0;
      while (state.load(std::memory_order_acquire) != 1)
      {
        if (quit.load(std::memory_order_acquire)) return;
        if (++spins < 40000) __builtin_ia32_pause();
        else
        {
          std::unique_lock<std::mutex> lk(m);
          cv.wait_for(lk, std::chrono::milliseconds(2),
                      [&] { return state.load(std::memory_order_acquire) == 1 || quit.load(std::memory_order_acquire); });
          spins = 0;
        }
      }
      rc = job();
      if (rc < 0) err = g_err;
      state.store(2, std::memory_order_release);
    }
  }
  void post(std::function<int()> f)
  {
    job = std::move(f);
    state.store(1, std::memory_order_release);
    cv.notify_one();
  }
  int join()
  {
    while (state.load(std::memory_order_acquire) != 2) __builtin_ia32_pause();
    state.store(0, std::memory_order_release);
    if (rc < 0) g_err = err;
    return rc;
  }
  void stop()
  {
    quit.store(true, std::memory_order_release);
    cv.notify_one();
    if (th.joinable()) th.join();
  }
};

constexpr int kGroupBase = 1 << 20; // instance ids >= kGroupBase name sharded instances

struct Group
{
  int                     S = 0, C = 0, tips = 0, nbuf = 0, nmat = 0;
  long long               P = 0;
  std::vector<int>        sub_id;  // plain instance ids, in pattern order
  std::vector<Instance *> sub;
  std::vector<long long>  lo, n;   // pattern range of each shard
  std::vector<int>        ctx_of, k_of; // device context of each shard, and its number on that device
  Collective             *co = nullptr;
  int                     last_warn = 0;
  bool                    warn_valid = false;
  std::vector<ShardWorker *> workers; // one per shard when the shards sit on more than one device (or PHYHIP_SHARD_THREADS=1)
};

std::mutex           g_groups_mu;
std::vector<Group *> g_groups;

Group *get_group(int id)
{
  if (id < kGroupBase) return nullptr;
  std::lock_guard<std::mutex> lk(g_groups_mu);
  const int k = id - kGroupBase;
  if (k >= (int)g_groups.size()) return nullptr;
  return g_groups[k];
}
static void forget_group(int id)
{
  std::lock_guard<std::mutex> lk(g_groups_mu);
  g_groups[id - kGroupBase] = nullptr;
}

template <typename F> int group_each(Group *G, F &&f)
{
  for (size_t g = 0; g < G->sub_id.size(); ++g)
  {
    const int rc = f(G->sub_id[g], G->lo[g], G->n[g]);
    if (rc < 0) return rc;
  }
  return PHYHIP_SUCCESS;
}

// run f(g) for every shard g: on the helper threads when the group has them, else in the calling thread
template <typename F> int group_parallel(Group *G, F &&f)
{
  if (G->workers.empty())
  {
    for (size_t g = 0; g < G->sub.size(); ++g)
    {
      int rc = set_dev(G->sub[g]->dev);
      if (rc || (rc = f((int)g)) < 0) return rc;
    }
    return PHYHIP_SUCCESS;
  }
  for (size_t g = 0; g < G->sub.size(); ++g)
    G->workers[g]->post([G, g, &f]() -> int {
      const int rc = set_dev(G->sub[g]->dev);
      return rc ? rc : f((int)g);
    });
  int rc = PHYHIP_SUCCESS;
  for (size_t g = 0; g < G->sub.size(); ++g)
  {
    const int r = G->workers[g]->join();
    if (r < 0 && rc == PHYHIP_SUCCESS) rc = r;
  }
  return rc;
}

static void release_group(Group *G)
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
static int mixture_groups(const int *instances, int count, std::vector<Group *> &Gs)
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

static int create_group(int tipCount, int partialsBufferCount, int stateCount, int patternCount, int matrixBufferCount,
                        int categoryCount, const int *resourceList, int resourceCount, phyhip_instance_details *returnInfo,
                        long classAxisFlag = 0)
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

// Lk(b) / Lk(NULL) on a sharded instance: every shard's traversal + edge evaluation (no host synchronisation), then the
// collective.  Launches go out shard by shard from the one host thread; the devices run concurrently.
static int group_edge_lnl(Group *G, int parent, int child, int pm, double *out)
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
