"""Child process of tests/test_gpu_switches.py: loads the DIAG build of the engine (PHYHIP_LIBDIR is set by the parent), runs
the golden fixture and a seeded SPR / Br_Len_Opt call stream under the environment it was given, and prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import orc  # noqa: E402,F401
import phyg  # noqa: E402
from gpu_common import device_tree_from_golden, synthetic_pair  # noqa: E402


def main():
    d = phyg.load(os.path.join(ROOT, "tests", "golden", "nucleic_gtr_g4.phyg"))
    t, ot = device_tree_from_golden(d)
    res = {}
    try:
        t.Set_Both_Sides(True)
        lnl = t.Lk(None)
        res["lnl_rel"] = abs(lnl - d["lnL"][0]) / abs(d["lnL"][0])
        w = d["wght"] > 0
        ot.lk(None, both_sides=True)
        ok = True
        for (e, side), p in ot.plk.items():
            ok = ok and np.array_equal(t.partials(e, side)[w], p[w]) and np.array_equal(t.scale_factors(e, side)[w], ot.scale[(e, side)][w])
        res["vectors_bit_equal"] = bool(ok)
    finally:
        t.close()
    from phyml_amd import replay
    t, ot, tree, st = synthetic_pair(26, 900, 4, 4, seed=19, host_pmat=False, ambiguous_every=11)
    try:
        t.Set_Both_Sides(True)
        t.Lk(None)
        tr = replay.make_trace(26, tree.edge_left, tree.edge_rght, tree.edge_len, 40, seed=2, walk_every=3, opt_every=4, n_dlk=3)
        a, a2 = t.Replay_Surface_Trace(tr)
    finally:
        t.close()
    res["stream"] = [float.hex(float(x)) for x in a]
    res["stream2"] = [float.hex(float(x)) for x in a2]
    # the same stream with host-computed matrices (the argument-upload route), and on an alignment of 282 tiles (the
    # large-grid resident evaluator's range)
    t, ot, tree, st = synthetic_pair(26, 900, 4, 4, seed=19, host_pmat=True, ambiguous_every=11)
    try:
        t.Set_Both_Sides(True)
        t.Lk(None)
        h, h2 = t.Replay_Surface_Trace(tr)
    finally:
        t.close()
    t, ot, tree, st = synthetic_pair(14, 9000, 4, 4, seed=21, host_pmat=False, ambiguous_every=11)
    try:
        t.Set_Both_Sides(True)
        t.Lk(None)
        trb = replay.make_trace(14, tree.edge_left, tree.edge_rght, tree.edge_len, 30, seed=4, walk_every=3, opt_every=4, n_dlk=3)
        b, b2 = t.Replay_Surface_Trace(trb)
    finally:
        t.close()
    # (first outputs: the lnL of every scalar-returning call; second outputs: dlnL of the dLk calls -- compared separately)
    res["stream_host"] = [float.hex(float(x)) for x in h]
    res["stream_host_d"] = [float.hex(float(x)) for x in h2]
    res["stream_big"] = [float.hex(float(x)) for x in b]
    res["stream_big_d"] = [float.hex(float(x)) for x in b2]
    res["sum_w"] = {"stream": 900.0, "stream_host": 900.0, "stream_big": 9000.0}
    # 20 states, device-built matrices: SPR candidates (the matrix rebuild inside the launch) and branch-length chains
    t, ot, tree, st = synthetic_pair(14, 300, 20, 4, seed=23, host_pmat=False, ambiguous_every=13)
    try:
        t.Set_Both_Sides(True)
        t.Lk(None)
        tra = replay.make_trace(14, tree.edge_left, tree.edge_rght, tree.edge_len, 30, seed=5, walk_every=3, opt_every=4, n_dlk=3)
        s20, s20d = t.Replay_Surface_Trace(tra)
    finally:
        t.close()
    res["stream_aa"] = [float.hex(float(x)) for x in s20]
    res["stream_aa_d"] = [float.hex(float(x)) for x in s20d]
    res["sum_w"]["stream_aa"] = 300.0
    # 20 states: the golden proteic fixture (generic kernel against the MFMA kernel)
    d = phyg.load(os.path.join(ROOT, "tests", "golden", "proteic_lg_g4.phyg"))
    t, ot = device_tree_from_golden(d)
    try:
        lnl = t.Lk(None)
        res["aa_lnl_rel"] = abs(lnl - d["lnL"][0]) / abs(d["lnL"][0])
    finally:
        t.close()
    # ... and its transition matrices built on the device: the whole tree's list, then a short one (the arguments route), every
    # double of them (src/models.c:257-326 is one chain of operations: whichever kernel builds them must give these bits)
    import hashlib
    t, ot = device_tree_from_golden(d, host_pmat=False)
    try:
        lnl = t.Lk(None)
        h = hashlib.sha256()
        for e in range(d["Pij_rr"].shape[0]):
            h.update(np.ascontiguousarray(t.inst.get_transition_matrix(e)).tobytes())
        short = [0, 3, 5]
        t.inst.update_transition_matrices(np.array(short, np.int32), np.array([0.013, 0.2, 1.7]))
        for e in short:
            h.update(np.ascontiguousarray(t.inst.get_transition_matrix(e)).tobytes())
        res["aa_device_matrices"] = {"lnL": float.hex(float(lnl)), "sha256": h.hexdigest()}
    finally:
        t.close()
    print("SWITCH_RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
