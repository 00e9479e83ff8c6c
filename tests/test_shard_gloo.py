"""The N>1 host logic on CPU: world_size-2 gloo processes do what bench.py's ranks do around the engine -- take their
contiguous pattern range, regenerate exactly their columns of the shared alignment, receive a 128-byte communicator id
from rank 0, evaluate (the CPU oracle stands in for the device kernels: there is no GPU here), sum ONE scalar over the
ranks, and agree on the maximum elapsed time.  The result must equal the unsharded reference lnL.  The engine's own
sharded path (RCCL inside libphyhip.so) is covered on hardware by tests/test_gpu_shard.py."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, name, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import orc
    from phyml_amd import phyg, shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = phyg.load(os.path.join(ROOT, "tests", "golden", name + ".phyg"))
    P = int(d["n_pattern"][0])
    lo, hi = shard.shard_range(P, rank, world)
    m = orc.Model(d)
    n = int(d["n_otu"][0])
    tv, ds, amb = [], [], []
    for t in range(n):
        v, s, a = orc.init_tip(m.datatype, d["tip_chars"][t][lo:hi])
        tv.append(v); ds.append(s); amb.append(a)
    ot = orc.OracleTree(m, n, d["edge_left"], d["edge_rght"], d["edge_len"], d["wght"][lo:hi], tv, ds, amb,
                        invar=d["invar"][lo:hi], apply_scaling=int(d["apply_lk_scaling"][0]))
    ot.set_adjacency(d["node_v"], d["node_b"]); ot.tip_root = int(d["tip_root"][0])
    ids = [bytes(range(128)) if rank == 0 else None]      # stands for phyhip_comm_get_unique_id() on rank 0
    dist.broadcast_object_list(ids, src=0)
    assert ids[0] == bytes(range(128))
    t = torch.tensor([ot.lk(None)], dtype=torch.float64)
    part = float(t[0])
    dist.all_reduce(t)
    tt = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    assert float(tt[0]) == float(world)
    q.put((rank, lo, hi, part, float(t[0])))
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["nucleic_gtr_g4", "synth_aa_90x24"])
def test_two_shards_allreduce(name, golden):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    ref = golden(name)["lnL"][0]
    (r0, lo0, hi0, p0, tot0), (r1, lo1, hi1, p1, tot1) = res
    assert lo0 == 0 and hi0 == lo1 and hi1 == int(golden(name)["n_pattern"][0])
    assert tot0 == tot1                                    # every rank holds the same reduced value
    assert abs(tot0 - ref) / abs(ref) < 1e-12              # shard lnL is additive
    assert abs((p0 + p1) - tot0) <= 1e-9


def test_rank_local_generation_equals_slices_of_the_whole_alignment():
    """Every rank regenerates only its own columns (workloads.make(pattern_offset=lo)); together they are the alignment."""
    from phyml_amd import shard, workloads
    whole = workloads.make("cfg4_nt_100x1M", n_pattern=4001)["states"]
    for world in (2, 3, 8):
        for r in range(world):
            lo, hi = shard.shard_range(4001, r, world)
            part = workloads.make("cfg4_nt_100x1M", n_pattern=hi - lo, pattern_offset=lo)["states"]
            assert np.array_equal(part, whole[:, lo:hi])


def test_shard_ranges_cover():
    from phyml_amd import shard
    for P in (1, 7, 16, 50000, 1000003):
        for w in (1, 2, 3, 8):
            r = [shard.shard_range(P, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == P
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
