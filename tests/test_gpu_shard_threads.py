"""The multi-device code of libphyhip.so in the form a real multi-GPU node runs it -- ONE HELPER THREAD PER SHARD issuing that
shard's launches (phyhip_shard.hpp: ShardWorker; groups over more than one device always have them) -- on whatever devices the box
has: a repeated-device list with PHYHIP_SHARD_THREADS=1 exercises the helper threads, the per-shard virtual-buffer bookkeeping, the
sharded mixtures and the collective on a one-GPU box; with two or more devices the same tests run over distinct devices as well.
(What this cannot show on one GPU: peer placement and a communicator of more than one rank -- tools/multigpu_selfcheck.py prints
those first on a multi-GPU box, and runs here on the repeated list.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import orc  # noqa: F401
from gpu_common import synthetic_pair
from phyml_amd import replay

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _device_lists():
    import torch
    n = torch.cuda.device_count()
    lists = [[0, 0], [0, 0, 0, 0, 0]]
    if n >= 2:
        lists += [[0, 1], list(range(min(n, 8)))]
    return lists


@pytest.fixture(autouse=True)
def _threads(monkeypatch):
    monkeypatch.setenv("PHYHIP_SHARD_THREADS", "1")


@pytest.mark.parametrize("ns,P", [(4, 9000), (4, 700), (20, 260)])
def test_virtual_buffers_per_shard_under_helper_threads(ns, P):
    """Whole-tree traversals that leave tip x tip results virtual on every shard, then a search-like stream (short launches that
    materialise them, matrix and tip changes in between): every scalar and every buffer read back equal to the plain instance's
    (scalars to 1e-12 -- the shard sums are added in another order --, vectors bit for bit)."""
    for devs in _device_lists():
        t, ot, tree, st = synthetic_pair(34, P, ns, 4, seed=31, host_pmat=False, ambiguous_every=13, devices=devs)
        t1, _, _, _ = synthetic_pair(34, P, ns, 4, seed=31, host_pmat=False, ambiguous_every=13)
        try:
            for x in (t, t1):
                x.Set_Both_Sides(True)
            a, b = t.Lk(None), t1.Lk(None)
            assert abs(a - b) <= 1e-12 * abs(b)
            assert t.inst.virtual_stats()[0] > 0   # (first shard's counters: buffers did stay virtual)
            tr = replay.make_trace(34, tree.edge_left, tree.edge_rght, tree.edge_len, 50, seed=9, walk_every=3, opt_every=4, n_dlk=3)
            (u, u2), (v, v2) = t.Replay_Surface_Trace(tr), t1.Replay_Surface_Trace(tr)
            m = v != 0
            assert np.max(np.abs(u[m] - v[m]) / np.abs(v[m])) < 1e-12
            assert np.max(np.abs(u2 - v2) / np.maximum(1.0, np.abs(v2))) < 1e-9
            a, b = t.Lk(None), t1.Lk(None)
            assert abs(a - b) <= 1e-12 * abs(b)
            for e in (0, 7, 20):
                for side in (0, 1):
                    if (e, side) in ot.plk:
                        assert np.array_equal(t.partials(e, side), t1.partials(e, side)), (devs, e, side)
                        assert np.array_equal(t.scale_factors(e, side), t1.scale_factors(e, side))
        finally:
            t.close(); t1.close()


def test_recorded_queue_only_calls_reach_the_shards_in_order():
    """A group with helper threads records the queue-only calls (phyhip_shard.hpp: Group::deferred) and replays them on the shards' own
    threads in front of whatever needs them.  Thousands of them without an evaluation in between (the list drains itself), matrix
    refreshes and partial updates interleaved, a getter right behind a recorded update (drained by get_group), argument errors
    reported at the call: the sharded instance ends where the plain instance ends, buffer for buffer."""
    t, ot, tree, st = synthetic_pair(20, 900, 4, 4, seed=5, host_pmat=False, devices=[0, 0, 0])
    t1, _, _, _ = synthetic_pair(20, 900, 4, 4, seed=5, host_pmat=False)
    try:
        for x in (t, t1):
            x.Set_Both_Sides(True)
            x.Lk(None)
        order = []
        ot.post_order(ot.tip_root, ot.adj[ot.tip_root][0][0], ops=order)
        rng = np.random.default_rng(3)
        n_calls = 0
        while n_calls < 5000:
            e = int(rng.integers(0, ot.ne))
            l = float(10.0 ** rng.uniform(-3, 0))
            for x in (t, t1):
                x.inst.update_transition_matrices(np.array([e], np.int32), np.array([l]))
            n_calls += 1
            for (b, dd) in order[:: max(1, len(order) // 3)]:
                for x in (t, t1):
                    x.Update_Partial_Lk(b, dd)
                n_calls += 1
        b, dd = order[-1]
        side = 0 if dd == ot.el[b] else 1
        assert np.array_equal(t.partials(b, side), t1.partials(b, side))          # a getter behind recorded updates
        for (b, dd) in order:
            for x in (t, t1):
                x.Update_Partial_Lk(b, dd)
        e = ot.root_edge()
        a, c = t.Lk(e), t1.Lk(e)
        assert abs(a - c) <= 1e-12 * abs(c)
        with pytest.raises(Exception):
            t.inst.update_partials([(t.inst.nbuf + 5, 0, 0, 1, 1)])               # reported at the call, not at the replay
        assert abs(t.Lk(e) - c) <= 1e-12 * abs(c)
    finally:
        t.close(); t1.close()


def test_sharded_mixtures_under_helper_threads():
    """The mixture tests' sharded layouts (class instances and the class axis, lnL and dLk) once more with the helper threads on."""
    import test_gpu_mixture as tm
    lay = [x for x in tm.LAYOUTS if x[1] is not None and len(x[1]) > 1]
    assert lay
    for layout in lay:
        for fx in ("lg4x", "nt4"):
            tm.test_lg4x_mixture_on_device(False, layout, fx)
            tm.test_lg4x_mixture_dlk_on_device(False, layout, fx)
            tm.test_lg4x_mixture_on_the_class_axis(False, layout, fx)
            tm.test_lg4x_mixture_dlk_on_the_class_axis(False, layout, fx)


def test_real_search_on_a_sharded_instance_under_helper_threads(tmp_path):
    """PhyML's real spr.c / optimiz.c on three shards driven by helper threads: every scalar against the reference (check mode)."""
    import test_gpu_search as ts
    import torch
    devs = "0,1,0" if torch.cuda.device_count() >= 2 else "0,0,0"
    ts._cache.clear()  # (the cache key does not know the environment)
    chk = ts.run_search("search_nucleic_spr", "check", tmp_path, devices=devs)
    ts._cache.clear()
    assert chk["calls"]["Lk"] > 10000 and chk["worst_rel_lnL"] < 1e-10 and chk["worst_rel_dlnL"] < 1e-6, chk


def test_multigpu_selfcheck_runs_on_this_box():
    """tools/multigpu_selfcheck.py -- the first command for a multi-GPU box -- on every device the box has (a repeated list on one
    GPU): the sharded lnL equals the sum of the shards evaluated alone, and RCCL built one rank per distinct device."""
    import torch
    n = torch.cuda.device_count()
    devs = ",".join(str(d) for d in range(n)) if n >= 2 else "0,0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "multigpu_selfcheck.py"), "--devices", devs, "--patterns", "200000", "--steps", "5"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.rstrip().endswith("OK"), (r.stdout[-1500:], r.stderr[-1500:])
    out = json.loads(r.stdout[:r.stdout.rindex("}") + 1])
    assert out["lnL_rel_err_vs_shard_sum"] < 1e-12 and out["rccl_ranks"] == len(set(devs.split(",")))
