"""Per-call parity of the CPU oracle with the REAL reference over recorded tree searches.

tests/golden/trace_*.phyg hold the first thousands of likelihood-surface calls PhyML's own SPR search and
branch-length optimisation made on its example alignments (recorded by oracle/trace_driver.c through symbol
interposition on the unmodified reference), with the scalar every Lk(b) / dLk returned.  Replaying the stream on
the oracle must reproduce every one of them: this pins the restatement -- partial updates through pointer-swapped
buffers, P-matrix refreshes at optimiser-chosen lengths, edge likelihoods, eigen-basis derivatives -- call by call
against what the reference computed inside a real search."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from phyml_amd import phyg, replay
import replay_oracle

TRACES = ["trace_nucleic_spr", "trace_proteic_spr", "trace_synth200_spr"]  # (the last: 200 taxa, 46 000 calls, round 4)


@pytest.mark.parametrize("name", TRACES)
def test_oracle_reproduces_recorded_search(name):
    d = phyg.load(os.path.join(GOLDEN, name + ".phyg"))
    tr, ref_out, ref_out2 = replay.recorded_trace(d)
    kinds = tr["kind"]
    assert (kinds == replay.UPDATE).sum() > 500 and (kinds == replay.EDGE_LNL).sum() > 300 and (kinds == replay.DLK).sum() > 300
    ot = replay_oracle.tree_from_recorded(d)
    out, out2 = replay_oracle.RecordedReplayer(ot).run(tr)
    sc = np.isin(kinds, (replay.EDGE_LNL, replay.DLK, replay.EIGEN_LNL))
    rel = np.abs(out[sc] - ref_out[sc]) / np.abs(ref_out[sc])
    assert rel.max() < 1e-12, (name, rel.max(), int(np.argmax(rel)))
    dl = kinds == replay.DLK
    err = np.abs(out2[dl] - ref_out2[dl]) / np.maximum(1.0, np.abs(ref_out2[dl]))
    assert err.max() < 1e-9, (name, err.max())
    # the search moved: likelihoods differ across the stream and improve over it
    lnl = ref_out[kinds == replay.EDGE_LNL]
    assert lnl.max() - lnl.min() > 1.0


@pytest.mark.parametrize("name", TRACES)
def test_recorded_stream_is_well_formed(name):
    """Every buffer a record reads was written by an earlier record, every matrix was set before its first use, every dLk
    follows an Update_Eigen_Lr, ids stay inside what an instance sized like the reference's slab holds."""
    d = phyg.load(os.path.join(GOLDEN, name + ".phyg"))
    tr, _, _ = replay.recorded_trace(d)
    n = int(d["n_otu"][0])
    written, mats, have_dot = set(), set(), False
    for k, a, b, c, dd, e in zip(tr["kind"], tr["a"], tr["b"], tr["c"], tr["d"], tr["e"]):
        if k == replay.SET_PMAT:
            mats.add(a)
        elif k == replay.UPDATE:
            assert a >= n and c in mats and e in mats
            for child in (b, dd):
                assert child < n or child in written
            written.add(a)
        elif k == replay.EDGE_LNL:
            assert a in written and (b < n or b in written) and c in mats
        elif k == replay.EIGEN_LR:
            assert a in written and (b < n or b in written)
            have_dot = True
        elif k in (replay.DLK, replay.EIGEN_LNL):
            assert have_dot
    assert max(written) < n + 3 * n - 2 and max(mats) < 2 * n - 1
    assert int(d["trace_n_buffers"][0]) == len(written) and int(d["trace_n_matrices"][0]) == len(mats)
