"""GPU: seeded random sweep over tree size, pattern count, category count, branch-length range, ambiguity density,
zero-weight patterns, scaling on/off and evaluation mode -- device vs oracle: lnL, every partial vector and scale vector
bit for bit, per-site log-likelihoods, lnL at random edges, and dLk on a random edge."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from gpu_common import synthetic_pair
from phyml_amd import synth


def _cfg(i, ns):
    h = synth.hash_u64(1234 + ns, i, np.arange(16))
    n = 4 + int(h[0] % (40 if ns == 4 else 24))
    P = 1 + int(h[1] % (700 if ns == 4 else 150))
    C = [1, 2, 3, 4, 4, 4, 5, 8][int(h[2] % 8)]
    lmax = [0.05, 0.3, 1.5, 6.0][int(h[3] % 4)]  # long branches force the 2^256 rescaling on deeper trees
    amb = [0, 0, 3, 7][int(h[4] % 4)]
    w = None
    if h[5] % 3 == 0:
        w = (synth.hash_u64(99, i, np.arange(P)) % 4).astype(np.float64)  # weights 0..3, zeros included
        if not np.any(w > 0):
            w[0] = 1.0
    scaling = 0 if (h[6] % 5 == 0 and lmax < 1.0) else 1
    both = bool(h[7] % 2)
    host_pmat = bool(h[8] % 2)
    return dict(n=n, P=P, C=C, lmax=lmax, amb=amb, w=w, scaling=scaling, both=both, host_pmat=host_pmat, seed=int(h[9] % 100000))


@pytest.mark.parametrize("ns", [4, 20])
@pytest.mark.parametrize("i", range(int(__import__("os").environ.get("PHYHIP_FUZZ_N", "30"))))
def test_random_configuration(ns, i):
    c = _cfg(i, ns)
    t, ot, tree, _ = synthetic_pair(c["n"], c["P"], ns, c["C"], seed=c["seed"], lmin=0.001, lmax=c["lmax"], wght=c["w"],
                                    apply_scaling=c["scaling"], host_pmat=c["host_pmat"], ambiguous_every=c["amb"])
    try:
        t.Set_Both_Sides(1 if c["both"] else 0)
        lnl = t.Lk(None)
        ref = ot.lk(None, both_sides=c["both"])
        tol = 1e-12 if c["host_pmat"] else 1e-9
        assert abs(lnl - ref) <= tol * abs(ref), (c, lnl, ref)
        w = ot.wght > 0
        if c["host_pmat"]:
            for (e, side), p in ot.plk.items():
                if not c["both"] and not np.any(p):
                    continue
                assert np.array_equal(t.partials(e, side)[w], p[w]), (c, e, side)
                assert np.array_equal(t.scale_factors(e, side)[w], ot.scale[(e, side)][w]), (c, e, side)
        site = t.inst.site_outputs()[0]
        assert np.max(np.abs(site[w] - ot.c_lnL_sorted[w])) < (1e-10 if c["host_pmat"] else 1e-7)
        if c["both"]:
            ne = 2 * c["n"] - 3
            for e in {int(x) % ne for x in synth.hash_u64(7, i, np.arange(3))}:
                a, b = t.Lk(e), ot.lk(e)
                assert abs(a - b) <= tol * abs(b), (c, e, a, b)
            e = int(synth.hash_u64(8, i, np.arange(1))[0]) % ne
            t.Set_Update_Eigen_Lr(1); t.Set_Use_Eigen_Lr(0)
            t.Lk(e); ot.lk(e); ot.update_eigen_lr(e)
            t.Set_Update_Eigen_Lr(0); t.Set_Use_Eigen_Lr(1)
            for l in (1e-4, 0.07, 0.9):
                lv, la = t.dLk(l, e)
                da = t.c_dlnL
                _, lb, db = ot.dlk(l)
                assert abs(la - lb) <= tol * abs(lb) and abs(da - db) <= 1e-7 * max(1.0, abs(db)), (c, e, l, la, lb, da, db)
    finally:
        t.close()
