"""GPU: per-call parity with the REAL reference over recorded tree searches (tests/golden/trace_*.phyg).

The device engine replays, through the C host layer's Replay_Surface_Trace (one C loop over the C ABI), the
likelihood-surface calls that PhyML's own SPR search / branch-length optimisation made (oracle/trace_driver.c), and
must return what the reference returned, call by call: every Lk(b) lnL and every dLk (lnL, dlnL) -- SURVEY 8b
"caller counterpart", gate 1e-6 relative (north star); measured ~1e-14."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from phyml_amd import lktree, phyg, replay

pytestmark = pytest.mark.gpu


def device_tree_from_recorded(d, host_pmat):
    n, P, S, C = int(d["n_otu"][0]), int(d["n_pattern"][0]), int(d["ns"][0]), int(d["ncatg"][0])
    t = lktree.LkTree(n, d["edge_left"], d["edge_rght"], d["edge_len"], P, S, C, host_pmat=host_pmat)
    t.set_model(d["pi"], d["gamma_rr"], d["gamma_r_proba"], d["e_val"], d["r_e_vect"], d["l_e_vect"], float(d["l_min"][0]),
                float(d["l_max"][0]), float(d["br_len_mult"][0]), int(d["apply_lk_scaling"][0]), int(d["invar_model"][0]),
                float(d["pinvar"][0]))
    t.Make_Tree_For_Lk(d["wght"], d["invar"])
    tv, _, _ = replay.tips_from_masks(d["tip_mask"], S)
    t.set_tips(tip_partials=tv)
    return t


@pytest.mark.parametrize("name", ["trace_nucleic_spr", "trace_proteic_spr", "trace_synth200_spr"])
@pytest.mark.parametrize("host_pmat", [True, False])
def test_device_reproduces_recorded_search(name, host_pmat):
    d = phyg.load(os.path.join(GOLDEN, name + ".phyg"))
    tr, ref_out, ref_out2 = replay.recorded_trace(d)
    n = int(d["n_otu"][0])
    # the recorder numbers buffers by first appearance; the instance must hold them (3n-6 edge sides + spares)
    assert int(d["trace_n_buffers"][0]) <= 3 * n - 6 + 4 and int(d["trace_n_matrices"][0]) <= 2 * n - 3 + 4
    t = device_tree_from_recorded(d, host_pmat)
    try:
        out, out2 = t.Replay_Surface_Trace(tr)
    finally:
        t.close()
    kinds = tr["kind"]
    sc = np.isin(kinds, (replay.EDGE_LNL, replay.DLK, replay.EIGEN_LNL))
    rel = np.abs(out[sc] - ref_out[sc]) / np.abs(ref_out[sc])
    assert rel.max() < (1e-12 if host_pmat else 1e-10), (name, host_pmat, rel.max(), int(np.argmax(rel)))
    dl = kinds == replay.DLK
    err = np.abs(out2[dl] - ref_out2[dl]) / np.maximum(1.0, np.abs(ref_out2[dl]))
    assert err.max() < 1e-8, (name, host_pmat, err.max())
