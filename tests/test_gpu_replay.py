"""Per-call parity of the SPR / Br_Len_Opt surface-call pattern (SURVEY 7.1 step 10b, cfg5's caller side):
the same seeded call stream is replayed through the C host layer on the GPU and through the CPU oracle; every
scalar a call returned (lnL of each regraft candidate, lnL and dlnL of each dLk) must agree."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from gpu_common import device_tree_from_golden
from phyml_amd import replay
from replay_oracle import OracleReplayer


@pytest.mark.parametrize("name,ncand", [("nucleic_gtr_g4", 120), ("nucleic_gtr_g4_inv", 60), ("synth_aa_90x24", 60),
                                        ("synth_nt_300x40", 80)])
def test_spr_call_stream_parity(name, ncand, golden):
    d = golden(name)
    t, ot = device_tree_from_golden(d, host_pmat=True)
    try:
        t.Set_Both_Sides(True)
        t.Lk(None)
        ot.lk(None, both_sides=True)
        tr = replay.make_trace(ot.n, d["edge_left"], d["edge_rght"], d["edge_len"], ncand, seed=11, walk_every=3, opt_every=4,
                               n_dlk=5)
        assert t.spare_p_lk_idx == replay.side_buffer_map(ot.n, d["edge_left"], d["edge_rght"])[1]
        got, got2 = t.Replay_Surface_Trace(tr)
        ref, ref2 = OracleReplayer(ot).run(tr)
        k = tr["kind"]
        lnl_calls = (k == replay.EDGE_LNL) | (k == replay.DLK)
        assert lnl_calls.sum() >= ncand
        assert np.max(np.abs(got[lnl_calls] - ref[lnl_calls]) / np.abs(ref[lnl_calls])) < 1e-11
        dl = k == replay.DLK
        assert np.max(np.abs(got2[dl] - ref2[dl]) / np.maximum(1.0, np.abs(ref2[dl]))) < 1e-8
        # the stream leaves the tree's own buffers intact: a fresh full evaluation still gives the reference lnL
        assert abs(t.Lk(None) - d["lnL"][0]) / abs(d["lnL"][0]) < 1e-12
    finally:
        t.close()
