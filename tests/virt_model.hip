// tests/virt_model.hip -- test infrastructure (CPU only, no device is touched): a property test of the virtual-buffer bookkeeping
// of libphyhip.so (phyml_amd/csrc/phyhip_queue.hip: rewrite_pending, devirtualise*).  It links the BUILT library, builds an
// Instance by hand, feeds it random queues and readers, and after every "launch" interprets the rewritten queue with the
// kernels' forwarding rules (last two results in registers, in-step children, non-storing operations) on symbolic values:
//   * a stale buffer must never be read from memory;
//   * what a launch leaves in memory -- or virtual, through its definition -- must be what the plain semantics (every
//     operation stored, in queue order) hold for that buffer;
//   * the evaluation edge sees the plain semantics' values.
// Matrices and tip rows change through the store-first route (devirtualise_matrix / devirtualise_tip + a launch); the snapshot
// route of the matrix setters needs a device and is covered by tests/test_gpu_virtual.py.
// Built and run by tests/test_virt_model.py:  virt_model <seed> <events> <tips> <soa 0|1> [in-step children 1|0]
#include "../phyml_amd/csrc/phyhip_host.hpp"

#include <cstdint>
#include <random>

using namespace phyhip_host;

static uint64_t H(uint64_t a, uint64_t b, uint64_t c, uint64_t d)
{
  uint64_t x = 0x9e3779b97f4a7c15ull;
  for (uint64_t v : {a, b, c, d})
  {
    x ^= v + 0x9e3779b97f4a7c15ull + (x << 6) + (x >> 2);
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 31;
  }
  return x;
}

struct Model
{
  Instance *I;
  std::vector<uint64_t> tipv, matv;     // versions of the tip rows / matrices
  std::vector<uint64_t> ref, dev;       // plain semantics / device memory per buffer
  std::vector<char>     written;        // the buffer has been the destination of a queued operation
  std::vector<DevOp>    queued;         // what the caller queued since the last launch (plain semantics are applied at launch)
  long                  n_launch = 0, n_instep = 0, n_nostore = 0, n_pad = 0;

  uint64_t tipval(int t) const { return H(1, (uint64_t)t, tipv[t], 0); }
  uint64_t def_value(const DevOp &d) const { return H(tipval(d.c1), tipval(d.c2), matv[d.pm1], matv[d.pm2]); }

  void die(const char *what, int a = -1, int b = -1) const
  {
    fprintf(stderr, "VIRT_MODEL FAIL: %s (%d, %d) after %ld launches\n", what, a, b, n_launch);
    exit(1);
  }

  // One launch.  Every queued operation carries its number in the upper bits of DevOp::pad (the library uses bits 0-2 and copies
  // the rest along), so the device's run of the REWRITTEN queue can be followed in lockstep with the plain semantics (every
  // operation stored, in queue order): `cur` is the plain value of every buffer at the point the device has reached.
  void launch(const EdgeEval *ee)
  {
    std::vector<uint64_t> cur = ref;
    size_t jp = 0;
    auto plain = [&](size_t upto) { // apply the queued operations jp .. upto - 1
      for (; jp < upto; ++jp)
      {
        const DevOp &o = queued[jp];
        const uint64_t v1 = o.c1 < I->tips ? tipval(o.c1) : cur[o.c1], v2 = o.c2 < I->tips ? tipval(o.c2) : cur[o.c2];
        cur[o.dest] = H(v1, v2, matv[o.pm1], matv[o.pm2]);
      }
    };
    rewrite_pending(I, ee, true);
    keep_real_clear(I); // (as flush_impl does)
    const std::vector<DevOp> &L = I->pending;
    const bool has_inl = !I->pending_inl.empty();
    if (has_inl && I->pending_inl.size() != L.size()) die("pending_inl is not parallel to pending");
    int      d1 = -1, d2 = -1; // destinations of the previous two operations
    uint64_t r1 = 0, r2 = 0;   // ... and their results (registers)
    for (size_t k = 0; k < L.size(); ++k)
    {
      const DevOp &o = L[k];
      const size_t tag = (size_t)((unsigned)o.pad >> 8);
      if ((o.pad & 6) && L.size() < 3) die("in-step child in a launch of the argument form", (int)k);
      if ((o.pad & 1) && L.size() < 3) die("non-storing operation in a launch of the argument form", (int)k);
      if (tag)
      { // a queued operation at its place: everything queued in front of it has happened in the plain semantics (also the
        // operations that left their place), it has not yet
        if (tag - 1 < jp || tag - 1 >= queued.size()) die("queued operations out of order", (int)k, (int)tag);
        plain(tag - 1);
      }
      else
      { // an operation the library put in: it belongs to the queued operation that follows it (a definition re-issued in front of
        // its reader: what left its place in front of that reader has happened), or to the end of the queue
        size_t nxt = 0;
        for (size_t j = k + 1; j < L.size() && !nxt; ++j) nxt = (size_t)((unsigned)L[j].pad >> 8);
        if (nxt && nxt - 1 < jp) die("queued operations out of order", (int)k, (int)nxt);
        plain(nxt ? nxt - 1 : queued.size());
      }
      auto child = [&](int c, int bit) -> uint64_t {
        uint64_t v;
        if (o.pad & bit)
        {
          if (!has_inl) die("in-step flag without a definition", (int)k);
          const InlineDef &d = I->pending_inl[k];
          if (d.a < 0 || d.a >= I->tips || d.b < 0 || d.b >= I->tips) die("in-step definition is not tip x tip", (int)k);
          ++n_instep;
          v = H(tipval(d.a), tipval(d.b), matv[d.pmA], matv[d.pmB]);
        }
        else if (c < I->tips) return tipval(c);
        else if (c == d1) v = r1;
        else if (c == d2) v = r2;
        else v = dev[c];
        if (v != cur[c]) die("an operation reads another value than the plain semantics hold at that point (stale memory, a stale register or a stale definition)", c, (int)k);
        return v;
      };
      const uint64_t v = H(child(o.c1, 2), child(o.c2, 4), matv[o.pm1], matv[o.pm2]);
      if (tag) plain(tag); // (the operation itself)
      if (v != cur[o.dest]) die("an operation computes another value than the plain semantics hold for its buffer at that point", o.dest, (int)k);
      if (o.pad & 1) ++n_nostore;
      else dev[o.dest] = v;
      d2 = d1; r2 = r1; d1 = o.dest; r1 = v;
    }
    plain(queued.size());
    if ((L.size() & 1) && L.size() >= 3)
    { // The pipelined kernels alternate two register sets: an odd LIST (three operations or more; shorter ones travel in the kernel
      // arguments and are not padded) runs its last operation once more, one position later -- its own result is now the newest
      // register, the previous operation's the second, and what was two steps back is gone: read from memory
      const size_t k = L.size() - 1;
      const DevOp &o = L[k];
      ++n_pad;
      auto child = [&](int c, int bit) -> uint64_t {
        uint64_t v;
        if (o.pad & bit)
        {
          const InlineDef &d = I->pending_inl[k];
          v = H(tipval(d.a), tipval(d.b), matv[d.pmA], matv[d.pmB]);
        }
        else if (c < I->tips) return tipval(c);
        else if (c == d2) v = r2; // (the operation in front of the last one)
        else v = dev[c];
        if (v != cur[c]) die("the re-run of an odd list's last operation reads a stale value", c, (int)k);
        return v;
      };
      const uint64_t v = H(child(o.c1, 2), child(o.c2, 4), matv[o.pm1], matv[o.pm2]);
      if (v != cur[o.dest]) die("the re-run of an odd list's last operation computes another value", o.dest, (int)k);
      if (!(o.pad & 1)) dev[o.dest] = v;
      r1 = v;
    }
    if (ee)
      for (int side : {ee->parent, ee->child})
      {
        if (side < I->tips) continue;
        const uint64_t seen = (side == d1) ? r1 : dev[side];
        if (seen != cur[side]) die("the evaluation sees another value than the plain semantics", side);
      }
    ref = cur;
    queued.clear();
    I->pending.clear();
    std::fill(I->mat_in_queue.begin(), I->mat_in_queue.end(), 0);
    ++n_launch;
    // memory and definitions after the launch
    int nv = 0;
    for (int b = I->tips; b < I->nbuf; ++b)
    {
      if (I->virt[b])
      {
        ++nv;
        const DevOp &d = I->vdef[b];
        if (d.c1 >= I->tips || d.c2 >= I->tips) die("a virtual buffer's definition is not tip x tip", b);
        if (def_value(d) != ref[b]) die("a virtual buffer's definition is not its plain value", b);
      }
      else if (written[b] && dev[b] != ref[b]) die("a buffer is neither stored nor virtual", b);
    }
    if (nv != I->n_virtual) die("n_virtual does not count the flags", nv, I->n_virtual);
  }
};

int main(int argc, char **argv)
{
  const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
  const int      events = argc > 2 ? atoi(argv[2]) : 2000, tips = argc > 3 ? atoi(argv[3]) : 12;
  const bool     soa = argc > 4 ? atoi(argv[4]) != 0 : true;
  const bool     in_step = argc > 5 ? atoi(argv[5]) != 0 : true; // 0: every virtual child re-issued as a step of its own (diag: PHYHIP_VIRT_INLINE=0)
  std::mt19937   rng(seed);
  auto rnd = [&](int n) { return (int)(rng() % (unsigned)n); };

  Instance *I = new Instance();
  I->tips = tips; I->nbuf = 3 * tips; I->nmat = 2 * tips; I->nmat_all = I->nmat + 2 * (I->nbuf - I->tips);
  I->S = soa ? 4 : 20; I->C = 4; I->soa = soa; I->perm = !soa; I->nt_groups = 2; I->prefetch_dist = 2;
  I->virt.assign(I->nbuf, 0);
  I->keep_real_flag.assign(I->nbuf, 0);
  I->vdef.assign(I->nbuf, DevOp{0, 0, 0, 0, 0, 0});
  I->mat_in_queue.assign(I->nmat_all, 0);
  I->virt_min_ops = 3 + rnd(6);
  I->virt_inline = in_step;

  Model M;
  M.I = I;
  M.tipv.assign(tips, 1); M.matv.assign(I->nmat_all, 1);
  M.ref.assign(I->nbuf, 0); M.dev.assign(I->nbuf, 0); M.written.assign(I->nbuf, 0);

  auto queue_op = [&](int dest, int c1, int c2) {
    DevOp o{dest, c1, c2, rnd(I->nmat), rnd(I->nmat), 0};
    M.queued.push_back(o);
    o.pad = (int)(M.queued.size() << 8); // (its number in this queue, above the library's bits)
    I->pending.push_back(o);
    I->mat_in_queue[o.pm1] = 1; I->mat_in_queue[o.pm2] = 1;
    M.written[dest] = 1;
  };
  auto any_child = [&](int not_this) {
    for (;;)
    {
      const int c = rnd(I->nbuf);
      if (c == not_this) continue;
      if (c >= tips && !M.written[c]) continue; // (never written: the reference would read garbage as well)
      return c;
    }
  };
  // a "traversal": cherries first, then operations that read earlier results -- the shape that leaves buffers virtual
  auto queue_traversal = [&](int n) {
    std::vector<int> made;
    for (int k = 0; k < n; ++k)
    {
      const int dest = tips + rnd(I->nbuf - tips);
      const int kind = rnd(4);
      // (an operation never reads its own destination: phyhip_update_partials refuses it)
      int from = made.empty() ? -1 : made[rnd((int)made.size())];
      if (from == dest) from = -1;
      if (kind == 0 || from < 0) queue_op(dest, rnd(tips), rnd(tips));
      else if (kind == 1) queue_op(dest, from, rnd(tips));
      else queue_op(dest, from, any_child(dest));
      if (std::find(made.begin(), made.end(), dest) == made.end()) made.push_back(dest);
    }
  };
  for (int ev = 0; ev < events; ++ev)
  {
    switch (rnd(10))
    {
      case 0: case 1: queue_traversal(3 + rnd(14)); break;                       // a long list
      case 2: queue_traversal(1 + rnd(2)); break;                                // a short one
      case 3:
      { // an evaluation
        EdgeEval ee{any_child(-1), any_child(-1), rnd(I->nmat), nullptr, true, nullptr};
        M.launch(&ee);
        break;
      }
      case 4: M.launch(nullptr); break;                                          // a launch without an evaluation
      case 5:
      { // a reader of memory (phyhip_get_partials, Update_Eigen_Lr as its own kernel): the buffer is stored after the launch
        const int b = tips + rnd(I->nbuf - tips);
        if (!M.written[b]) break;
        devirtualise(I, b);
        M.launch(nullptr);
        if (I->virt[b] || M.dev[b] != M.ref[b]) M.die("a reader finds its buffer virtual or stale", b);
        break;
      }
      case 6:
      { // a matrix changes (store-first route): what is queued runs on the old value, dependants are stored on it
        const int m = rnd(I->nmat);
        M.launch(nullptr);
        devirtualise_matrix(I, m);
        M.launch(nullptr);
        for (int b = tips; b < I->nbuf; ++b)
          if (I->virt[b] && (I->vdef[b].pm1 == m || I->vdef[b].pm2 == m)) M.die("a virtual buffer still reads a matrix that changes", b, m);
        ++M.matv[m];
        break;
      }
      case 7:
      { // a tip row changes
        const int t = rnd(tips);
        M.launch(nullptr);
        devirtualise_tip(I, t);
        M.launch(nullptr);
        for (int b = tips; b < I->nbuf; ++b)
          if (I->virt[b] && (I->vdef[b].c1 == t || I->vdef[b].c2 == t)) M.die("a virtual buffer still reads a tip row that changes", b, t);
        ++M.tipv[t];
        break;
      }
      case 8:
        if (rnd(4) == 0)
        { // everything stored (phyhip_set_virtual_buffers(0) / another threshold)
          I->virt_min_ops = 0;
          devirtualise_all(I);
          M.launch(nullptr);
          if (I->n_virtual != 0) M.die("devirtualise_all left virtual buffers");
          I->virt_min_ops = 1 + rnd(12);
        }
        break;
      default: queue_traversal(6 + rnd(20)); break;
    }
  }
  M.launch(nullptr);
  printf("VIRT_MODEL OK seed %u: %ld launches, %llu stores skipped, %ld in-step children, %ld non-storing re-issues, %llu stored on demand, %ld odd lists re-ran their last operation\n", seed,
         M.n_launch, (unsigned long long)I->n_virt_skipped, M.n_instep, M.n_nostore, (unsigned long long)I->n_virt_material, M.n_pad);
  return 0;
}
