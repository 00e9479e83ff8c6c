"""Every environment switch of the product library that selects another kernel or another route to the same numbers gives the
same numbers: golden lnL, partial vectors and scale vectors bit-equal to the oracle, and a seeded SPR / Br_Len_Opt call
stream equal to the default build's, scalar by scalar (tools/README.md lists the switches and what each was measured for)."""
import numpy as np
import pytest

import orc  # noqa: F401
from gpu_common import device_tree_from_golden, synthetic_pair

pytestmark = pytest.mark.gpu

SWITCHES = [{}, {"PHYHIP_NT_GROUPS": "1"}, {"PHYHIP_NT_GROUPS": "2"}, {"PHYHIP_NT_GROUPS": "4"}, {"PHYHIP_NT2_DIST": "1"},
            {"PHYHIP_NT2_DIST": "1", "PHYHIP_NT_GROUPS": "1"}, {"PHYHIP_FOLD_PMATS": "0"}, {"PHYHIP_ARGS_RECS": "0"},
            {"PHYHIP_RESIDENT": "0"}, {"PHYHIP_HOST_SUM": "0"}, {"PHYHIP_SPIN": "0"}, {"PHYHIP_EAGER_PMAT": "0"},
            {"PHYHIP_PM_COPY": "1"}, {"PHYHIP_GENERIC_NT": "1"}]
_stream = {}


def _ids(sw):
    return ",".join(f"{k[7:]}={v}" for k, v in sw.items()) or "default"


@pytest.mark.parametrize("sw", SWITCHES, ids=_ids)
def test_switch_gives_the_same_numbers(sw, golden, monkeypatch):
    for k, v in sw.items():
        monkeypatch.setenv(k, v)
    d = golden("nucleic_gtr_g4")
    t, ot = device_tree_from_golden(d)
    try:
        t.Set_Both_Sides(True)
        lnl = t.Lk(None)
        assert abs(lnl - d["lnL"][0]) / abs(d["lnL"][0]) < 1e-12
        w = d["wght"] > 0
        ot.lk(None, both_sides=True)
        for (e, side), p in ot.plk.items():
            assert np.array_equal(t.partials(e, side)[w], p[w]), (e, side)
            assert np.array_equal(t.scale_factors(e, side)[w], ot.scale[(e, side)][w]), (e, side)
    finally:
        t.close()
    # the call stream of a search: device-built matrices, partial updates, edge likelihoods, eigen products, dLk chains
    from phyml_amd import replay
    t, ot, tree, st = synthetic_pair(26, 900, 4, 4, seed=19, host_pmat=False, ambiguous_every=11)
    try:
        t.Set_Both_Sides(True)
        t.Lk(None)
        tr = replay.make_trace(26, tree.edge_left, tree.edge_rght, tree.edge_len, 40, seed=2, walk_every=3, opt_every=4, n_dlk=3)
        out = t.Replay_Surface_Trace(tr)
    finally:
        t.close()
    if not sw:
        _stream["default"] = out
    elif "default" in _stream:
        a, a2 = _stream["default"]
        if "PHYHIP_GENERIC_NT" in sw or "PHYHIP_HOST_SUM" in sw or sw.get("PHYHIP_NT_GROUPS") in ("1", "4"):
            # another kernel shape adds the patterns' contributions in another order: the same value to rounding
            m = a != 0
            assert np.max(np.abs(out[0][m] - a[m]) / np.abs(a[m])) < 1e-12
        else:
            assert np.array_equal(out[0], a) and np.array_equal(out[1], a2)
