"""Helpers shared by the GPU parity tests: build a device LkTree (C host layer) from a golden dump."""
import numpy as np

import orc
from phyml_amd import lktree


def device_tree_from_golden(d, host_pmat=True):
    """Device tree with the reference's own neighbour order (node_v/node_b of the dump); tips come from the
    oracle's tip encoder.  host_pmat=True: the C host layer's own PMat() + upload (bit-exact route, src/lk.c:2360);
    False: device PMat from the eigen system (src/lk.c:2344)."""
    ot = orc.tree_from_golden(d)
    m = ot.m
    t = lktree.LkTree(ot.n, d["edge_left"], d["edge_rght"], d["edge_len"], ot.P, m.ns, m.ncatg,
                      node_v=d["node_v"], node_b=d["node_b"], host_pmat=host_pmat)
    t.tip_root = ot.tip_root
    t.set_model(m.pi, m.gamma_rr, m.gamma_r_proba, m.e_val, m.r_e_vect, m.l_e_vect, m.l_min, m.l_max, m.br_len_mult,
                int(d["apply_lk_scaling"][0]), m.invar_model, m.pinvar)
    t.Make_Tree_For_Lk(d["wght"], d["invar"])
    t.set_tips(tip_partials=ot.tip_vec)
    return t, ot
