// phyhip_shard.hpp -- the multi-GPU side of libphyhip.so: pattern shards and the ONE
// collective of the path, an RCCL all-reduce of the per-shard log-likelihood (src/lk.c:856: c_lnL is a plain sum over
// patterns; src/lk.c:744-745: so are c_lnL and c_dlnL of dLk).  Declarations: included by phyhip_host.hpp inside namespace phyhip_host; the functions live in phyhip_shard.hip.
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

#define NCCLCHK(call)                                                                                        \
  do                                                                                                         \
  {                                                                                                          \
    ncclResult_t r_ = (call);                                                                                \
    if (r_ != ncclSuccess)                                                                                   \
      return fail(PHYHIP_ERROR_GENERAL, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

constexpr int kRedStride = 4; // doubles per reduction slot: {warning, lnL, dlnL, pad}

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

inline int set_dev(int dev) { return make_current(dev); }

// where shard number `k` of its device writes {warning, lnL, dlnL}
inline double *shard_slot(const DevCtx &c, int k) { return c.d_red + (size_t)(c.nsub > 1 ? k + 1 : 0) * kRedStride; }

int  reduce_and_publish(Collective &co, int count, Instance *I0);
int  warm_up_collective(Collective &co);
void release_collective(Collective *co);

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

// A queue-only call on a sharded instance that has helper threads, kept until something needs the shards to have seen it
// (Group::deferred): 0 phyhip_update_partials, 1 phyhip_update_transition_matrices, 2 phyhip_set_transition_matrix
struct GroupDeferred
{
  int                           kind;
  std::vector<phyhip_operation> ops;
  std::vector<int>              idx;
  std::vector<double>           val;
};

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
  // Scalar-returning SHORT calls (Lk(b) behind at most two queued operations, dLk, Lk in the eigen basis: the call patterns of
  // src/spr.c:640-646 and src/optimiz.c:607-663) answered shard by shard as plain instances answer them -- each shard's resident
  // evaluators included -- and added by the calling thread in shard order, instead of a launch per shard + the collective
  // (~35 us of pure latency for 16-24 bytes the host is waiting for anyway).  The process owns every shard, so the host IS the
  // place where the shard sums meet; whole-tree evaluations keep the RCCL all-reduce, and so does the one-process-per-GPU form
  // for everything.  PHYHIP_SHARD_HOST_COMBINE: 0 never, 1 short calls (default), 2 every evaluation.
  int                     host_combine = 1;
  // The queue-only calls of a search (three matrix refreshes and a partial update per SPR candidate, src/spr.c:640-646) do no device
  // work, but entering a shard's entry point from the calling thread makes that shard's device current -- a device switch per shard
  // and call, ~1 us each: 8 shards x 4 calls would cost a candidate more than its evaluation.  With helper threads (every shard's
  // device stays current on its own thread) such calls are only RECORDED here, validated against the group's dimensions, and
  // replayed by each shard's helper thread in front of the next thing that needs them: the evaluation's own job (one round trip to
  // the helpers per scalar-returning call) or, for every other entry point, get_group() itself, which drains the list first.
  std::vector<GroupDeferred> deferred;
  size_t                  deferred_ops = 0; // partial updates among them (group_combines_on_host counts them)
  int                     drain_rc = 0;     // a replay that failed: returned (once) by the next group_each / group_parallel
  std::string             drain_err;
  bool defers() const { return !workers.empty(); }
};

Group *get_group(int id);          // (drains Group::deferred first: what the caller does next sees every queued call)
Group *get_group_nodrain(int id);  // the entry points that record, and the evaluations that replay inside their own jobs
int    group_replay(Group *G, int g);
int    group_drain(Group *G);
void   forget_group(int id);

inline int group_take_drain_error(Group *G)
{
  if (!G->drain_rc) return 0;
  const int rc = G->drain_rc;
  g_err = G->drain_err;
  G->drain_rc = 0;
  return rc;
}

template <typename F> int group_each(Group *G, F &&f)
{
  if (const int rc = group_take_drain_error(G)) return rc;
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
  if (const int rc = group_take_drain_error(G)) return rc;
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

void release_group(Group *G);
int  mixture_groups(const int *instances, int count, std::vector<Group *> &Gs);
int  create_group(int tipCount, int partialsBufferCount, int stateCount, int patternCount, int matrixBufferCount, int categoryCount,
                  const int *resourceList, int resourceCount, phyhip_instance_details *returnInfo, long classAxisFlag = 0);
int  group_edge_lnl(Group *G, int parent, int child, int pm, double *out);
bool group_combines_on_host(const Group *G, bool queue_counts);
int  group_collect_warning(Group *G, const std::vector<int> &shard_warn);
