"""Helpers shared by the GPU parity tests: build a device LkTree from a golden dump / a synthetic case."""
import numpy as np

import orc
from phyml_amd import lktree


def device_tree_from_golden(d, host_pmat=True):
    """Device tree with the reference's own neighbour order; tips come from the oracle's tip encoder and
    transition matrices from the oracle's PMat (bit-exact route) unless host_pmat is False."""
    ot = orc.tree_from_golden(d)
    m = ot.m
    hp = None
    if host_pmat:
        def hp(l):
            return orc.pmat_edge(l, m.ns, m.ncatg, m.gamma_rr, m.br_len_mult, m.l_min, m.l_max, m.r_e_vect, m.l_e_vect, m.e_val)
    t = lktree.LkTree(ot.n, ot.el, ot.er, ot.len, ot.P, m.ns, m.ncatg, adjacency=ot.adj, host_pmat=hp)
    t.tip_root = ot.tip_root
    t.set_model(m.pi, m.gamma_rr, m.gamma_r_proba, m.e_val, m.r_e_vect, m.l_e_vect, m.l_min, m.l_max, m.br_len_mult,
                int(d["apply_lk_scaling"][0]), m.invar_model, m.pinvar, d["invar"])
    t.set_data(d["wght"], tip_partials=ot.tip_vec)
    return t, ot
