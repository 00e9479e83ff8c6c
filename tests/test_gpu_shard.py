"""Multi-GPU path of the C ABI on real hardware: pattern shards inside libphyhip.so, ONE RCCL all-reduce of
{warning, lnL[, dlnL]} per evaluation, and the 1 M-pattern configuration (BASELINE configs[3]) against the reference's
own shard log-likelihoods (tests/golden/make_cfg4.py: the real PhyML on the eight 125 000-pattern shards).

On a one-GPU box the sharded instance is exercised as (a) one shard + a one-rank communicator and (b) several shards on
device 0 (local fixed-order sum + the one-rank all-reduce) -- every slicing / concatenating entry point and the RCCL call
path run; with two or more devices visible the same tests also run on distinct devices."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import orc  # noqa: F401
from conftest import FIXTURES
from gpu_common import device_tree_from_golden, synthetic_pair
from phyml_amd import capi, lktree, synth, workloads


def _n_devices():
    import torch
    return torch.cuda.device_count()


def _layouts():
    lay = [("one_shard_forced", [0], True), ("two_shards_dev0", [0, 0], False), ("three_shards_dev0", [0, 0, 0], False)]
    if _n_devices() >= 2:
        lay += [("two_devices", [0, 1], False), ("two_devices_two_shards_each", [0, 1, 0, 1], False)]
    return lay


LAYOUTS = _layouts()


@pytest.mark.parametrize("layout", LAYOUTS, ids=[x[0] for x in LAYOUTS])
@pytest.mark.parametrize("name", FIXTURES)
def test_sharded_instance_matches_reference(name, layout, golden):
    """Everything the single-device parity tests check, through a sharded instance: lnL and lnL at several edges against
    the reference (1e-12), partial vectors and scale vectors bit-equal to the oracle, per-site outputs, the eigen-basis
    products and dLk triples (the count-3 all-reduce)."""
    _, devs, force = layout
    d = golden(name)
    if len(devs) > int(d["wght"].size):
        pytest.skip("fewer patterns than shards")
    t, ot = device_tree_from_golden(d, devices=devs, force_sharded=force)
    try:
        rng = t.inst.shard_ranges()
        assert len(rng) == len(devs) and [r[0] for r in rng] == devs
        assert rng[0][1] == 0 and sum(r[2] for r in rng) == t.P and all(a[1] + a[2] == b[1] for a, b in zip(rng, rng[1:]))
        assert t.inst.comm_size() == len(set(devs))
        t.Set_Both_Sides(True)
        lnl = t.Lk(None)
        ref = d["lnL"][0]
        assert abs(lnl - ref) / abs(ref) < 1e-12
        assert t.inst.numerical_warning() == 0
        w = d["wght"] > 0
        ot.lk(None, both_sides=True)
        for (e, side), p in ot.plk.items():
            assert np.array_equal(t.partials(e, side)[w], p[w]), (e, side)
            assert np.array_equal(t.scale_factors(e, side)[w], ot.scale[(e, side)][w])
        site, cur, cat, fact = t.inst.site_outputs()
        assert np.array_equal(fact[w], d["fact_sum_scale"][w])
        assert np.max(np.abs(site[w] - d["c_lnL_sorted"][w])) < 1e-10
        assert np.allclose(cat[w], d["unscaled_site_lk_cat"][w], rtol=1e-12, atol=0)
        got = np.array([t.Lk(e) for e in range(0, t.ne, 7)])
        exp = d["edge_lnL"][0:t.ne:7]
        assert np.max(np.abs(got - exp) / np.abs(exp)) < 1e-12
        for k, e in enumerate(d["eigen_edges"]):
            e = int(e)
            t.Set_Update_Eigen_Lr(True); t.Set_Use_Eigen_Lr(False)
            t.Lk(e)
            t.Set_Update_Eigen_Lr(False); t.Set_Use_Eigen_Lr(True)
            assert np.allclose(t.inst.get_dot_prod()[w], d[f"dot_prod_{e}"][w], rtol=1e-12, atol=1e-300)
            for j in range(3):
                l_in, lnl_ref, dlnl_ref = d["dlk_triples"][k, j]
                l_out, v = t.dLk(l_in, e)
                assert l_out == l_in
                assert abs(v - lnl_ref) / abs(lnl_ref) < 1e-12
                assert abs(t.c_dlnL - dlnl_ref) <= 1e-8 * max(1.0, abs(dlnl_ref))
            assert abs(t.Lk(e) - d[f"eig_lnL_{e}"][0]) / abs(d[f"eig_lnL_{e}"][0]) < 1e-12
            t.Set_Use_Eigen_Lr(False)
    finally:
        t.close()


@pytest.mark.parametrize("ns,taxa,P", [(4, 40, 3000), (20, 24, 700)])
def test_sharded_instance_spr_call_pattern_matches_single_device(ns, taxa, P, monkeypatch):
    """A seeded SPR / Br_Len_Opt call stream (phyml_amd/replay.py) through a sharded instance returns the scalars the
    single-device instance returns (1e-12; the shard sums are added in a different order) -- with the shards driven from
    the calling thread and from the per-shard helper threads (what a multi-device group uses; PHYHIP_SHARD_THREADS=1
    forces them for shards that share a device), and with the short calls answered shard by shard and added on the host
    (PHYHIP_SHARD_HOST_COMBINE=1, the default: each shard's resident evaluators serve them), through the collective
    only (0) and with every evaluation added on the host (2)."""
    from phyml_amd import replay
    res, served = [], []
    runs = ((None, "0", "1"), ([0, 0, 0], "0", "1"), ([0, 0, 0], "1", "1"), ([0, 0, 0], "0", "0"), ([0, 0, 0], "1", "0"), ([0, 0, 0], "1", "2"))
    for devs, threads, combine in runs:
        monkeypatch.setenv("PHYHIP_SHARD_THREADS", threads)
        monkeypatch.setenv("PHYHIP_SHARD_HOST_COMBINE", combine)
        t, ot, tree, st = synthetic_pair(taxa, P, ns, 4, seed=77, devices=devs, host_pmat=(ns == 4))
        try:
            t.Set_Both_Sides(True)
            t.Lk(None)
            tr = replay.make_trace(taxa, tree.edge_left, tree.edge_rght, tree.edge_len, 60, seed=5, walk_every=3, opt_every=4, n_dlk=4)
            res.append(t.Replay_Surface_Trace(tr))
            served.append(t.inst.resident_stats(0)[0] + t.inst.resident_stats(1)[0])
        finally:
            t.close()
    (a, a2) = res[0]
    m = a != 0
    assert m.any()
    for (b, b2) in res[1:]:
        assert np.max(np.abs(a[m] - b[m]) / np.abs(a[m])) < 1e-12
        assert np.max(np.abs(a2 - b2) / np.maximum(1.0, np.abs(a2))) < 1e-9
    assert np.array_equal(res[1][0], res[2][0]) and np.array_equal(res[1][1], res[2][1])  # threads change nothing
    assert np.array_equal(res[3][0], res[4][0]) and np.array_equal(res[3][1], res[4][1])
    # the shards' resident evaluators serve the host-combined short calls, and only those
    assert served[1] > 0 and served[2] > 0 and served[5] > 0, served
    assert served[3] == 0 and served[4] == 0, served


def test_numerical_warning_rides_in_the_all_reduce():
    """Site likelihoods that underflow with scaling off raise tree->numerical_warning (src/lk.c:847-851) inside the
    shards; the flag must come back through the collective (and be absent again on a tree that does not underflow)."""
    for devs in (None, [0, 0]):
        t, ot, *_ = synthetic_pair(40, 64, 4, 4, seed=9, devices=devs)
        try:
            t.Lk(None)
            assert t.inst.numerical_warning() == 0
        finally:
            t.close()
        t, ot, *_ = synthetic_pair(900, 64, 4, 4, seed=9, lmin=1.0, lmax=3.0, apply_scaling=0, devices=devs)
        try:
            lnl = t.Lk(None)
            assert np.isfinite(lnl)
            assert t.inst.numerical_warning() == 1
        finally:
            t.close()


def test_one_process_per_gpu_communicator_single_rank(golden):
    """phyhip_comm_get_unique_id / phyhip_comm_init_rank (what bench.py under torchrun and an MPI host use): with one rank
    the evaluation still runs shard-sum -> ncclAllReduce -> publish and must return the reference's lnL and dLk."""
    d = golden("nucleic_gtr_g4_inv")
    t, ot = device_tree_from_golden(d)
    try:
        t.inst.comm_init_rank(1, 0, capi.comm_get_unique_id())
        assert t.inst.comm_size() == 1
        t.Set_Both_Sides(True)
        lnl = t.Lk(None)
        assert abs(lnl - d["lnL"][0]) / abs(d["lnL"][0]) < 1e-12
        e = int(d["eigen_edges"][0])
        t.Set_Update_Eigen_Lr(True); t.Lk(e); t.Set_Update_Eigen_Lr(False)
        l_in, lnl_ref, dlnl_ref = d["dlk_triples"][0, 1]
        _, v = t.dLk(l_in, e)
        assert abs(v - lnl_ref) / abs(lnl_ref) < 1e-12 and abs(t.c_dlnL - dlnl_ref) <= 1e-8 * max(1.0, abs(dlnl_ref))
    finally:
        t.close()


def _cfg_tree(name, lo, n, devices=None, force_sharded=False):
    cfg = workloads.CONFIGS[name]
    wl = workloads.make(name, n_pattern=n, pattern_offset=lo)
    tree, st, blk = wl["tree"], wl["states"], wl["model"]
    t = lktree.LkTree(tree.n_otu, tree.edge_left, tree.edge_rght, tree.edge_len, n, cfg["ns"], 4, devices=devices,
                      force_sharded=force_sharded)
    t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"],
                float(blk["l_min"][0]), float(blk["l_max"][0]))
    t.Make_Tree_For_Lk(np.ones(n))
    t.set_tips(tip_states=st.astype(np.int32))
    return t, st


@pytest.mark.parametrize("g", [0, 3, 7])
def test_cfg4_shards_match_the_reference(g):
    """125 000-pattern shard g of the 1 M-pattern alignment: input checksum and lnL against the real reference's value."""
    exp = workloads.manifest()["expected"]["cfg4_nt_100x1M"]
    lo = g * 125000
    t, st = _cfg_tree("cfg4_nt_100x1M", lo, 125000)
    try:
        assert synth.states_checksum(st) == exp["shard_checksum"][g]
        lnl = t.Lk(None)
        assert abs(lnl - exp["shard_lnL"][g]) / abs(exp["shard_lnL"][g]) < 1e-12
    finally:
        t.close()


def test_cfg4_one_million_patterns_against_the_reference():
    """BASELINE configs[3] at full size, 100 taxa x 1 000 000 patterns: on one device, and as a sharded instance (8 shards
    like the 8-GPU run when 8 devices are visible, else 8 shards over the visible devices) -- lnL within 1e-6 of the sum
    of the reference's shard values (north-star gate; 1e-12 held), same value at another evaluation edge."""
    exp = workloads.manifest()["expected"]["cfg4_nt_100x1M"]
    ref = exp["lnL"]
    t, st = _cfg_tree("cfg4_nt_100x1M", 0, 1000000)
    try:
        lnl = t.Lk(None)
        assert abs(lnl - ref) / abs(ref) < 1e-6
        assert abs(lnl - ref) / abs(ref) < 1e-12
    finally:
        t.close()
    nd = _n_devices()
    devs = [g % nd for g in range(8)]
    t = None
    os.environ["PHYHIP_SHARD_THREADS"] = "1"   # the helper threads a real 8-device group runs with
    wl = workloads.make("cfg4_nt_100x1M")
    tree, blk = wl["tree"], wl["model"]
    t = lktree.LkTree(tree.n_otu, tree.edge_left, tree.edge_rght, tree.edge_len, 1000000, 4, 4, devices=devs)
    try:
        t.set_model(blk["pi"], blk["gamma_rr"], blk["gamma_r_proba"], blk["e_val"], blk["r_e_vect"], blk["l_e_vect"],
                    float(blk["l_min"][0]), float(blk["l_max"][0]))
        t.Make_Tree_For_Lk(np.ones(1000000))
        t.set_tips(tip_states=st.astype(np.int32))
        assert [r[2] for r in t.inst.shard_ranges()] == [125000] * 8
        lnl8 = t.Lk(None)
        assert abs(lnl8 - ref) / abs(ref) < 1e-12
        t.Set_Both_Sides(True)
        t.Lk(None)
        assert abs(t.Lk(17) - ref) / abs(ref) < 1e-11
    finally:
        t.close()
        os.environ.pop("PHYHIP_SHARD_THREADS", None)


@pytest.mark.parametrize("split,host_sum", [("0", "0"), ("1", "0"), ("0", "1")])
@pytest.mark.parametrize("name", ["nucleic_gtr_g4", "proteic_lg_g4"])
def test_the_three_final_sums_agree(name, split, host_sum, golden, monkeypatch):
    """The three ways an evaluation's block sums become the scalar -- fused into the kernel's last workgroup
    (PHYHIP_HOST_SUM=0, PHYHIP_SPLIT_REDUCE=0), a separate final_reduce_kernel (PHYHIP_SPLIT_REDUCE=1), posted to the host
    and added there (PHYHIP_HOST_SUM=1, the default for host-returning calls at every grid size) -- all give the golden lnL, dLk and warning flag."""
    monkeypatch.setenv("PHYHIP_SPLIT_REDUCE", split)
    monkeypatch.setenv("PHYHIP_HOST_SUM", host_sum)
    d = golden(name)
    t, ot = device_tree_from_golden(d)
    try:
        t.Set_Both_Sides(True)
        lnl = t.Lk(None)
        assert abs(lnl - d["lnL"][0]) / abs(d["lnL"][0]) < 1e-12
        e = int(d["eigen_edges"][0])
        t.Set_Update_Eigen_Lr(True); t.Lk(e); t.Set_Update_Eigen_Lr(False)
        l_in, lnl_ref, dlnl_ref = d["dlk_triples"][0, 0]
        _, v = t.dLk(l_in, e)
        assert abs(v - lnl_ref) / abs(lnl_ref) < 1e-12
        assert t.inst.numerical_warning() == 0
    finally:
        t.close()


def test_host_side_final_sum_is_bit_identical_to_the_device_one(monkeypatch):
    """Large grid (more than 512 workgroups, several host-side partial sums): the host adds the posted block sums in final_reduce_kernel's order, so the
    scalar is the same double whichever path produced it; the numerical-warning flag arrives through host-mapped memory."""
    vals = []
    for hs in ("0", "1"):
        monkeypatch.setenv("PHYHIP_HOST_SUM", hs)
        t, ot, *_ = synthetic_pair(12, 40000, 4, 4, seed=31)
        try:
            vals.append((t.Lk(None), t.inst.numerical_warning()))
            ref = ot.lk(None)
            assert abs(vals[-1][0] - ref) / abs(ref) < 1e-12
        finally:
            t.close()
    assert vals[0] == vals[1] and vals[0][1] == 0
    monkeypatch.setenv("PHYHIP_HOST_SUM", "1")
    t, ot, *_ = synthetic_pair(900, 40000, 4, 4, seed=9, lmin=1.0, lmax=3.0, apply_scaling=0)
    try:
        assert np.isfinite(t.Lk(None)) and t.inst.numerical_warning() == 1
    finally:
        t.close()


def test_device_output_path_on_the_callers_stream(golden):
    """phyhip_set_stream + Lk_Shard_Device / phyhip_calculate_edge_log_likelihoods_device: the evaluation runs on the
    caller's stream and leaves lnL in the caller's device memory without synchronising."""
    import torch
    d = golden("nucleic_gtr_g4")
    t, ot = device_tree_from_golden(d)
    try:
        s = torch.cuda.Stream()
        out = torch.zeros(2, dtype=torch.float64, device="cuda:0")
        t.inst.set_stream(s.cuda_stream)
        with torch.cuda.stream(s):
            t.Lk_Shard_Device(out.data_ptr())
            twice = out * 2.0               # ordered behind the evaluation by the stream alone
        s.synchronize()
        ref = d["lnL"][0]
        assert abs(float(out[0]) - ref) / abs(ref) < 1e-12
        assert float(twice[0]) == 2.0 * float(out[0])
        assert abs(t.Lk(None) - ref) / abs(ref) < 1e-12   # and the host-returning form still works on that stream
    finally:
        t.close()


def test_sharded_instances_come_and_go(monkeypatch):
    """Create / evaluate / finalize sharded instances repeatedly (communicators, helper threads, shared streams and
    reduction buffers are torn down with them), interleaved with a plain instance that must keep working."""
    monkeypatch.setenv("PHYHIP_SHARD_THREADS", "1")
    t0, ot, *_ = synthetic_pair(12, 500, 4, 4, seed=3)
    try:
        ref = ot.lk(None)
        for rep in range(4):
            t, _, *_ = synthetic_pair(12, 500, 4, 4, seed=3, devices=[0] * (1 + rep % 3), force_sharded=True)
            try:
                assert abs(t.Lk(None) - ref) / abs(ref) < 1e-12
            finally:
                t.close()
            assert abs(t0.Lk(None) - ref) / abs(ref) < 1e-12
    finally:
        t0.close()
