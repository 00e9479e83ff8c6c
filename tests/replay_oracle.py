"""Replays a surface-call stream (phyml_amd/replay.py) on the CPU oracle -- the checker side of the
per-call parity test for the SPR / branch-length-optimisation call pattern."""
import ctypes as C

import numpy as np

import orc
from phyml_amd import replay


class OracleReplayer:
    def __init__(self, ot: orc.OracleTree, n_spare=4):
        self.ot = ot
        m = ot.m
        buf, n_real = replay.side_buffer_map(ot.n, ot.el, ot.er)
        self.key_of = {idx: key for key, idx in buf.items() if idx >= ot.n}
        self.n_real, self.ne = n_real, ot.ne
        self.spare_plk = [np.zeros((ot.P, m.ncatg * m.ns)) for _ in range(n_spare)]
        self.spare_scale = [np.zeros(ot.P, np.int32) for _ in range(n_spare)]
        self.spare_pm = [np.zeros((m.ncatg, m.ns, m.ns)) for _ in range(n_spare)]

    def _pm(self, idx):
        return self.ot.pm[idx] if idx < self.ne else self.spare_pm[idx - self.ne]

    def _arrays(self, idx):
        if idx >= self.n_real:
            return self.spare_plk[idx - self.n_real], self.spare_scale[idx - self.n_real]
        key = self.key_of[idx]
        return self.ot.plk[key], self.ot.scale[key]

    def _side(self, idx):
        s = orc.Side()
        ot = self.ot
        if idx < ot.n:
            s.p_lk = orc._p(ot.tip_vec[idx]); s.sum_scale = None; s.is_tip = 1
            s.is_ambigu = orc._p(ot.tip_amb[idx]); s.d_state = orc._p(ot.tip_ds[idx])
        else:
            p, sc = self._arrays(idx)
            s.p_lk = orc._p(p); s.sum_scale = orc._p(sc); s.is_tip = 0; s.is_ambigu = None; s.d_state = None
        return s

    def run(self, tr):
        ot, m, L = self.ot, self.ot.m, orc.lib()
        n = len(tr["kind"])
        out = np.zeros(n); out2 = np.zeros(n)
        for i in range(n):
            k, a, b, c, d, e, x = (int(tr["kind"][i]), int(tr["a"][i]), int(tr["b"][i]), int(tr["c"][i]), int(tr["d"][i]),
                                   int(tr["e"][i]), float(tr["x"][i]))
            if k == replay.SET_PMAT:
                pm = orc.pmat_edge(x, m.ns, m.ncatg, m.gamma_rr, m.br_len_mult, m.l_min, m.l_max, m.r_e_vect, m.l_e_vect, m.e_val)
                self._pm(a)[:] = pm
            elif k == replay.UPDATE:
                dst, dsc = self._arrays(a)
                s1, s2 = self._side(b), self._side(d)
                L.orc_update_partial(C.c_int(ot.P), C.c_int(m.ncatg), C.c_int(m.ns), orc._p(ot.wght), C.byref(s1), orc._p(self._pm(c)),
                                     C.byref(s2), orc._p(self._pm(e)), orc._p(dst), orc._p(dsc), C.c_int(ot.apply_scaling), C.c_int(ot.arith))
            elif k == replay.EDGE_LNL:
                left, rght = self._side(a), self._side(b)
                warn = C.c_int(0)
                out[i] = L.orc_edge_lnl(C.c_int(ot.P), C.c_int(m.ncatg), C.c_int(m.ns), orc._p(ot.wght), C.byref(left), C.byref(rght),
                                        orc._p(self._pm(c)), orc._p(m.pi), orc._p(m.gamma_r_proba), C.c_int(m.invar_model),
                                        C.c_double(m.pinvar), orc._p(ot.invar), C.c_int(ot.apply_scaling), C.c_int(ot.arith),
                                        orc._p(ot.c_lnL_sorted), orc._p(ot.cur_site_lk), orc._p(ot.unscaled_site_lk_cat),
                                        orc._p(ot.fact_sum_scale), C.byref(warn))
            elif k == replay.EIGEN_LR:
                left, rght = self._side(a), self._side(b)
                L.orc_update_eigen_lr(C.c_int(ot.P), C.c_int(m.ncatg), C.c_int(m.ns), orc._p(ot.wght), C.byref(left), C.byref(rght),
                                      orc._p(m.r_e_vect), orc._p(m.l_e_vect), orc._p(m.pi), orc._p(ot.dot_prod), C.c_int(ot.arith))
            elif k == replay.DLK:
                _, out[i], out2[i] = ot.dlk(x)
            elif k == replay.EIGEN_LNL:
                out[i] = ot.lk_eigen(x)
        return out, out2


class RecordedReplayer(OracleReplayer):
    """Replays a stream recorded from a real PhyML run (oracle/trace_driver.c): buffer and matrix ids are whatever the
    recorder assigned (order of first appearance), so storage is allocated per id on first use."""

    def __init__(self, ot: orc.OracleTree):
        self.ot = ot
        self.bufs, self.pms = {}, {}

    def _pm(self, idx):
        m = self.ot.m
        if idx not in self.pms:
            self.pms[idx] = np.zeros((m.ncatg, m.ns, m.ns))
        return self.pms[idx]

    def _arrays(self, idx):
        m = self.ot.m
        if idx not in self.bufs:
            self.bufs[idx] = (np.zeros((self.ot.P, m.ncatg * m.ns)), np.zeros(self.ot.P, np.int32))
        return self.bufs[idx]


def tree_from_recorded(d, arith=1):
    """OracleTree carrying the model, weights and tips of a recorded trace (its own edge arrays are only used for
    sizing: the replay works at buffer level)."""
    m = orc.Model(d)
    n = int(d["n_otu"][0])
    tv, ds, amb = replay.tips_from_masks(d["tip_mask"], m.ns)
    return orc.OracleTree(m, n, d["edge_left"], d["edge_rght"], d["edge_len"], d["wght"], tv, ds, amb, invar=d["invar"],
                          apply_scaling=int(d["apply_lk_scaling"][0]), arith=arith)
