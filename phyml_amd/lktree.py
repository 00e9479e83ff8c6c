"""Host-side mirror of PhyML's likelihood surface on top of the phyhip C ABI (Python harness form).

Names and argument meaning follow the reference (src/lk.h:27-159): `Lk(b)`, `dLk(l, b)`,
`Update_Partial_Lk(b, d)`, `Update_PMat_At_Given_Edge(b)`, `Post_Order_Lk(a, d)`,
`Pre_Order_Lk(a, d)`, `Update_Eigen_Lr(b)`, `Set_Both_Sides`, so the parity tests read like calls
into lk.c.  All arithmetic happens in libphyhip.so on the GPU; this module only walks the tree and
translates (edge, side) to buffer indices -- the job `Set_All_Partial_Lk` does in the reference
(src/lk.c:2922-3195).

Buffer indices: tips 0..n-1 (as src/lk.c:2229), then one partials buffer per internal edge side, then
(optionally) spare buffers; one transition-matrix buffer per edge.
"""
from __future__ import annotations

import sys

import numpy as np

from . import capi


class LkTree:
    def __init__(self, n_otu, edge_left, edge_rght, edge_len, n_pattern, ns, ncatg, device=None, adjacency=None,
                 host_pmat=None):
        self.n = int(n_otu)
        self.el = np.asarray(edge_left, dtype=np.int64)
        self.er = np.asarray(edge_rght, dtype=np.int64)
        self.len = np.array(edge_len, dtype=np.float64)
        self.ne = len(self.el)
        self.P, self.S, self.C = int(n_pattern), int(ns), int(ncatg)
        # (edge, side) -> partials buffer index; side 0 = left, 1 = rght
        self.buf = {}
        nxt = self.n
        for e in range(self.ne):
            for side, node in ((0, self.el[e]), (1, self.er[e])):
                if node < self.n:
                    self.buf[(e, side)] = int(node)      # tip vector lives on the tip's own index
                else:
                    self.buf[(e, side)] = nxt
                    nxt += 1
        self.n_partials = nxt
        self.inst = capi.Instance(self.n, self.n_partials, self.S, self.P, self.ne, self.C, device=device)
        if adjacency is None:
            adj = [[] for _ in range(2 * self.n - 2)]
            for e in range(self.ne):
                adj[self.el[e]].append((int(self.er[e]), e))
                adj[self.er[e]].append((int(self.el[e]), e))
            self.adj = adj
        else:
            self.adj = adjacency
        self.tip_root = 0
        self.both_sides = False
        self.use_eigen_lr = False
        self.update_eigen_lr = False
        self.c_lnL = 0.0
        self.c_dlnL = 0.0
        self.host_pmat = host_pmat          # callable(edge_len) -> [C][S][S] for the bit-exact host route
        self.n_update_calls = 0

    def close(self):
        self.inst.close()

    # ---- model / data upload (create_beagle_instance + update_beagle_ras/efrqs/eigen) -------------
    def set_model(self, pi, gamma_rr, gamma_r_proba, e_val, r_e_vect, l_e_vect, l_min=1e-8, l_max=100.0,
                  br_len_mult=1.0, apply_lk_scaling=1, invar_model=0, pinvar=0.0, invar=None):
        i = self.inst
        i.set_state_frequencies(pi)
        i.set_category_rates(gamma_rr)
        i.set_category_weights(gamma_r_proba)
        i.set_eigen_decomposition(r_e_vect, l_e_vect, e_val)
        i.set_phyml_options(l_min, l_max, br_len_mult, apply_lk_scaling)
        i.set_invariant_sites(invar_model, pinvar, invar)

    def set_data(self, wght, tip_partials=None, tip_states=None):
        self.inst.set_pattern_weights(wght)
        for t in range(self.n):
            if tip_partials is not None:
                self.inst.set_tip_partials(t, tip_partials[t])
            else:
                self.inst.set_tip_states(t, tip_states[t])

    # ---- a12 -------------------------------------------------------------------------------------------
    def Update_PMat_At_Given_Edge(self, b):
        if self.host_pmat is not None:
            self.inst.set_transition_matrix(b, self.host_pmat(self.len[b]))
        else:
            self.inst.update_transition_matrices([b], [self.len[b]])

    def Update_All_PMat(self):
        if self.host_pmat is not None:
            for b in range(self.ne):
                self.inst.set_transition_matrix(b, self.host_pmat(self.len[b]))
        else:
            self.inst.update_transition_matrices(np.arange(self.ne), self.len)

    # ---- a2 / a5 -----------------------------------------------------------------------------------------
    def _op(self, b, d):
        dest = self.buf[(b, 0 if d == self.el[b] else 1)]
        ch = []
        for (v, be) in self.adj[d]:
            if be != b:
                ch.append((self.buf[(be, 1 if d == self.el[be] else 0)], be))
        assert len(ch) == 2
        return (dest, ch[0][0], ch[0][1], ch[1][0], ch[1][1])

    def Update_Partial_Lk(self, b, d):
        if d < self.n:                       # src/lk.c:1297
            return
        self.inst.update_partials([self._op(b, d)])
        self.n_update_calls += 1

    # ---- a13 ----------------------------------------------------------------------------------------------
    def _post(self, a, d, out):
        if d < self.n:
            return
        dir_e = None
        for (v, be) in self.adj[d]:
            if v != a:
                self._post(d, v, out)
            else:
                dir_e = be
        out.append((dir_e, d))

    def _pre(self, a, d, out):
        if d < self.n:
            return
        for (v, be) in self.adj[d]:
            if v != a:
                out.append((be, d))
                self._pre(d, v, out)

    def _walk(self, fn, a, d):
        out = []
        old = sys.getrecursionlimit()
        sys.setrecursionlimit(max(old, 10 * self.n + 100))
        try:
            fn(a, d, out)
        finally:
            sys.setrecursionlimit(old)
        return out

    def Post_Order_Lk(self, a, d):
        order = self._walk(self._post, a, d)
        self.inst.update_partials([self._op(b, dd) for (b, dd) in order])
        self.n_update_calls += len(order)

    def Pre_Order_Lk(self, a, d):
        order = self._walk(self._pre, a, d)
        self.inst.update_partials([self._op(b, dd) for (b, dd) in order])
        self.n_update_calls += len(order)

    def Set_Both_Sides(self, yesno):
        self.both_sides = bool(yesno)

    def Set_Use_Eigen_Lr(self, yesno):
        self.use_eigen_lr = bool(yesno)

    def Set_Update_Eigen_Lr(self, yesno):
        self.update_eigen_lr = bool(yesno)

    # ---- a1 ---------------------------------------------------------------------------------------------------
    def Lk(self, b=None):
        if b is None:
            self.Update_All_PMat()
            r = self.tip_root
            v0 = self.adj[r][0][0]
            self.Post_Order_Lk(r, v0)
            if self.both_sides:
                self.Pre_Order_Lk(r, v0)
            b = self.adj[r][0][1]
        elif not self.use_eigen_lr:
            self.Update_PMat_At_Given_Edge(b)
        if self.update_eigen_lr:
            self.Update_Eigen_Lr(b)
        if self.use_eigen_lr:
            self.c_lnL = self.inst.eigen_lnl(self.len[b])
        else:
            self.c_lnL = self.inst.edge_lnl(self.buf[(b, 0)], self.buf[(b, 1)], b)
        return self.c_lnL

    # ---- a10 / a11 -----------------------------------------------------------------------------------------------
    def Update_Eigen_Lr(self, b):
        self.inst.update_eigen_lr(self.buf[(b, 0)], self.buf[(b, 1)])

    def dLk(self, l, b):
        """Returns (clamped l, lnL); sets c_lnL and c_dlnL like src/lk.c:749-750."""
        if self.update_eigen_lr:
            self.Update_Eigen_Lr(b)
        l2, lnl, dlnl = self.inst.eigen_lnl_dlnl(l)
        self.c_lnL, self.c_dlnL = lnl, dlnl
        return l2, lnl

    # ---- download hooks ---------------------------------------------------------------------------------------------
    def partials(self, b, side):
        return self.inst.get_partials(self.buf[(b, side)])

    def scale_factors(self, b, side):
        return self.inst.get_scale_factors(self.buf[(b, side)])
