"""ctypes binding of oracle/liboracle.so plus a small tree driver around it.  TEST INFRASTRUCTURE.

`OracleTree` evaluates a whole tree with the CPU restatement: it owns one numpy partial vector and
one scale vector per (edge, side) exactly like the reference's t_edge (p_lk_left/p_lk_rght,
sum_scale_left/rght), generates the update order of Post_Order_Lk / Pre_Order_Lk
(src/lk.c:282-393) and resolves children the way Set_All_Partial_Lk does (src/lk.c:2937-2986,
3212-3260).  It is written independently of the product's host code so the two check each other.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class Side(C.Structure):
    _fields_ = [("p_lk", C.c_void_p), ("sum_scale", C.c_void_p), ("is_tip", C.c_int),
                ("is_ambigu", C.c_void_p), ("d_state", C.c_void_p)]


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ROOT, "oracle", "liboracle.so")
        src = os.path.join(ROOT, "oracle", "phylk_oracle.c")
        if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.orc_edge_lnl.restype = C.c_double
        L.orc_lk_eigen.restype = C.c_double
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def init_tip(datatype, chars):
    P = len(chars)
    ns = 4 if datatype == 0 else 20
    v = np.zeros((P, ns)); ds = np.zeros(P, np.int16); amb = np.zeros(P, np.int16)
    ch = np.ascontiguousarray(chars, dtype=np.uint8)
    rc = lib().orc_init_tip(C.c_int(datatype), _p(ch), C.c_int(P), _p(v), _p(ds), _p(amb))
    if rc != 0:
        raise ValueError("unknown character state")
    return v, ds, amb


def pmat_edge(l, ns, ncatg, gamma_rr, br_len_mult, l_min, l_max, U, V, R):
    out = np.zeros((ncatg, ns, ns))
    lib().orc_update_pmat_edge(C.c_double(l), C.c_int(ns), C.c_int(ncatg), _p(f64(gamma_rr)), C.c_double(br_len_mult),
                               C.c_double(l_min), C.c_double(l_max), _p(f64(U)), _p(f64(V)), _p(f64(R)), _p(out))
    return out


class Model:
    """pi, rate classes and the eigen system -- a 'model block' as dumped by oracle/ref_driver.c."""

    def __init__(self, d):
        self.ns = int(d["ns"][0]); self.ncatg = int(d["ncatg"][0])
        self.pi = f64(d["pi"]); self.gamma_rr = f64(d["gamma_rr"]); self.gamma_r_proba = f64(d["gamma_r_proba"])
        self.e_val = f64(d["e_val"]); self.r_e_vect = f64(d["r_e_vect"]); self.l_e_vect = f64(d["l_e_vect"])
        self.l_min = float(d["l_min"][0]); self.l_max = float(d["l_max"][0])
        self.br_len_mult = float(d["br_len_mult"][0])
        self.invar_model = int(d["invar_model"][0]); self.pinvar = float(d["pinvar"][0])
        self.datatype = int(d["datatype"][0])


class OracleTree:
    def __init__(self, model: Model, n_otu, edge_left, edge_rght, edge_len, wght, tip_vec, tip_d_state, tip_is_ambigu,
                 invar=None, apply_scaling=1, arith=1, pmats=None):
        self.m = model
        self.n = int(n_otu)
        self.el = np.asarray(edge_left, dtype=np.int64); self.er = np.asarray(edge_rght, dtype=np.int64)
        self.len = f64(edge_len).copy()
        self.ne = len(self.el)
        self.P = len(wght)
        self.wght = f64(wght)
        self.tip_vec = [f64(t) for t in tip_vec]
        self.tip_ds = [np.ascontiguousarray(t, dtype=np.int16) for t in tip_d_state]
        self.tip_amb = [np.ascontiguousarray(t, dtype=np.int16) for t in tip_is_ambigu]
        self.invar = None if invar is None else np.ascontiguousarray(invar, dtype=np.int16)
        self.apply_scaling = apply_scaling
        self.arith = arith
        S, Cc = model.ns, model.ncatg
        self.plk = {}
        self.scale = {}
        for e in range(self.ne):
            for side, node in ((0, self.el[e]), (1, self.er[e])):
                if node >= self.n:
                    self.plk[(e, side)] = np.zeros((self.P, Cc * S))
                    self.scale[(e, side)] = np.zeros(self.P, np.int32)
        # node adjacency in edge order (the reference's v[]/b[] order is free; ours is edge order unless given)
        self.adj = [[] for _ in range(2 * self.n - 2)]
        for e in range(self.ne):
            self.adj[self.el[e]].append((int(self.er[e]), e))
            self.adj[self.er[e]].append((int(self.el[e]), e))
        self.pm = np.zeros((self.ne, Cc, S, S))
        if pmats is not None:
            self.pm[:] = pmats
        else:
            for e in range(self.ne):
                self.update_pmat(e)
        self.tip_root = 0
        # per-site outputs
        self.c_lnL_sorted = np.zeros(self.P); self.cur_site_lk = np.zeros(self.P)
        self.unscaled_site_lk_cat = np.zeros((self.P, Cc)); self.fact_sum_scale = np.zeros(self.P, np.int32)
        self.dot_prod = np.zeros((self.P, Cc * S))
        self.numerical_warning = 0
        self.n_updates = 0

    def set_adjacency(self, node_v, node_b):
        """Use the reference's own neighbour order (node_v/node_b of a golden dump)."""
        self.adj = [[(int(node_v[k, i]), int(node_b[k, i])) for i in range(3) if node_v[k, i] >= 0]
                    for k in range(len(node_v))]

    # --- a12 ---------------------------------------------------------------------------------
    def update_pmat(self, e):
        m = self.m
        self.pm[e] = pmat_edge(self.len[e], m.ns, m.ncatg, m.gamma_rr, m.br_len_mult, m.l_min, m.l_max,
                               m.r_e_vect, m.l_e_vect, m.e_val)

    # --- buffer resolution -------------------------------------------------------------------
    def _side(self, e, side):
        """orc_side for the subtree on `side` of edge e (tip -> tip vector)."""
        node = self.el[e] if side == 0 else self.er[e]
        s = Side()
        if node < self.n:
            s.p_lk = _p(self.tip_vec[node]); s.sum_scale = None; s.is_tip = 1
            s.is_ambigu = _p(self.tip_amb[node]); s.d_state = _p(self.tip_ds[node])
        else:
            s.p_lk = _p(self.plk[(e, side)]); s.sum_scale = _p(self.scale[(e, side)]); s.is_tip = 0
            s.is_ambigu = None; s.d_state = None
        return s

    def op_for(self, e, d):
        """(dest key, [(child edge, child side)]*2) for Update_Partial_Lk(tree, b=e, d)."""
        dest_side = 0 if d == self.el[e] else 1
        ch = []
        for (v, be) in self.adj[d]:
            if be != e:
                # d's neighbour across edge `be` sits on the other end of it
                ch.append((be, 1 if d == self.el[be] else 0))
        assert len(ch) == 2
        return (e, dest_side), ch

    def update_partial(self, e, d):
        if d < self.n:
            return
        dest, ch = self.op_for(e, d)
        m = self.m
        s1 = self._side(*ch[0]); s2 = self._side(*ch[1])
        lib().orc_update_partial(C.c_int(self.P), C.c_int(m.ncatg), C.c_int(m.ns), _p(self.wght),
                                 C.byref(s1), _p(self.pm[ch[0][0]]), C.byref(s2), _p(self.pm[ch[1][0]]),
                                 _p(self.plk[dest]), _p(self.scale[dest]), C.c_int(self.apply_scaling), C.c_int(self.arith))
        self.n_updates += 1

    # --- traversals (src/lk.c:282-393) ----------------------------------------------------------
    def post_order(self, a, d, ops=None):
        stack = [(a, d, 0)]
        # iterative form of the recursion, children in adjacency order
        order = []
        def rec(a, d):
            if d < self.n:
                return
            dir_e = None
            for (v, be) in self.adj[d]:
                if v != a:
                    rec(d, v)
                else:
                    dir_e = be
            order.append((dir_e, d))
        old = sys.getrecursionlimit(); sys.setrecursionlimit(max(old, 10 * self.n + 100))
        try:
            rec(a, d)
        finally:
            sys.setrecursionlimit(old)
        if ops is not None:
            ops.extend(order)
        else:
            for (e, dd) in order:
                self.update_partial(e, dd)

    def pre_order(self, a, d, ops=None):
        order = []
        def rec(a, d):
            if d < self.n:
                return
            for (v, be) in self.adj[d]:
                if v != a:
                    order.append((be, d))
                    rec(d, v)
        old = sys.getrecursionlimit(); sys.setrecursionlimit(max(old, 10 * self.n + 100))
        try:
            rec(a, d)
        finally:
            sys.setrecursionlimit(old)
        if ops is not None:
            ops.extend(order)
        else:
            for (e, dd) in order:
                self.update_partial(e, dd)

    def root_edge(self):
        # a_nodes[tip_root]->b[0]  (src/lk.c:578-579)
        return self.adj[self.tip_root][0][1]

    # --- a1 --------------------------------------------------------------------------------------
    def lk(self, e=None, both_sides=False, refresh_pmat=True):
        if e is None:
            if refresh_pmat:
                for k in range(self.ne):
                    self.update_pmat(k)
            r = self.tip_root
            v0 = self.adj[r][0][0]
            self.post_order(r, v0)
            if both_sides:
                self.pre_order(r, v0)
            e = self.root_edge()
        elif refresh_pmat:
            self.update_pmat(e)
        return self.edge_lnl(e)

    def edge_lnl(self, e):
        m = self.m
        left = self._side(e, 0); rght = self._side(e, 1)
        warn = C.c_int(0)
        v = lib().orc_edge_lnl(C.c_int(self.P), C.c_int(m.ncatg), C.c_int(m.ns), _p(self.wght),
                               C.byref(left), C.byref(rght), _p(self.pm[e]), _p(m.pi), _p(m.gamma_r_proba),
                               C.c_int(m.invar_model), C.c_double(m.pinvar), _p(self.invar),
                               C.c_int(self.apply_scaling), C.c_int(self.arith),
                               _p(self.c_lnL_sorted), _p(self.cur_site_lk), _p(self.unscaled_site_lk_cat),
                               _p(self.fact_sum_scale), C.byref(warn))
        self.numerical_warning = warn.value
        return v

    # --- a10 / a11 ---------------------------------------------------------------------------------
    def update_eigen_lr(self, e):
        m = self.m
        left = self._side(e, 0); rght = self._side(e, 1)
        lib().orc_update_eigen_lr(C.c_int(self.P), C.c_int(m.ncatg), C.c_int(m.ns), _p(self.wght), C.byref(left), C.byref(rght),
                                  _p(m.r_e_vect), _p(m.l_e_vect), _p(m.pi), _p(self.dot_prod), C.c_int(self.arith))

    def dlk(self, l):
        m = self.m
        lv = C.c_double(l); lnl = C.c_double(0); dlnl = C.c_double(0)
        lib().orc_dlk(C.byref(lv), C.c_int(self.P), C.c_int(m.ncatg), C.c_int(m.ns), _p(self.wght), _p(self.dot_prod),
                      _p(m.e_val), _p(m.gamma_rr), _p(m.gamma_r_proba), C.c_double(m.br_len_mult), C.c_double(m.l_min),
                      C.c_double(m.l_max), C.c_int(m.invar_model), C.c_double(m.pinvar), _p(self.invar), _p(m.pi),
                      _p(self.fact_sum_scale), C.c_int(self.apply_scaling), C.byref(lnl), C.byref(dlnl))
        return lv.value, lnl.value, dlnl.value

    def lk_eigen(self, l):
        m = self.m
        return lib().orc_lk_eigen(C.c_double(l), C.c_int(self.P), C.c_int(m.ncatg), C.c_int(m.ns), _p(self.wght),
                                  _p(self.dot_prod), _p(m.e_val), _p(m.gamma_rr), _p(m.gamma_r_proba),
                                  C.c_double(m.br_len_mult), C.c_double(m.l_min), C.c_double(m.l_max),
                                  C.c_int(m.invar_model), C.c_double(m.pinvar), _p(self.invar), _p(m.pi),
                                  _p(self.fact_sum_scale), C.c_int(self.apply_scaling))


def tree_from_golden(d, arith=1, use_dumped_pmats=False):
    """Build an OracleTree from a golden dump, taking the tip data through the oracle's own K6."""
    m = Model(d)
    n = int(d["n_otu"][0])
    tv, ds, amb = [], [], []
    for t in range(n):
        v, s, a = init_tip(m.datatype, d["tip_chars"][t])
        tv.append(v); ds.append(s); amb.append(a)
    pm = None
    if use_dumped_pmats and d["Pij_rr"].shape[0] == len(d["edge_len"]):
        pm = d["Pij_rr"]
    t = OracleTree(m, n, d["edge_left"], d["edge_rght"], d["edge_len"], d["wght"], tv, ds, amb,
                   invar=d["invar"], apply_scaling=int(d["apply_lk_scaling"][0]), arith=arith, pmats=pm)
    t.set_adjacency(d["node_v"], d["node_b"])
    t.tip_root = int(d["tip_root"][0])
    return t
