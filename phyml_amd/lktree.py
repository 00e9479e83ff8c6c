"""Python handle on the C host layer (libphyhip_lk.so, include/phyhip_lk.h).

The host side of the likelihood surface -- `Lk`, `dLk`, `Update_Partial_Lk`, `Post_Order_Lk`, ... with
the reference's names and argument meaning (src/lk.h:27-159) -- is plain C in
phyml_amd/csrc/host/phl_lk.c, as in the reference.  This module only mirrors its structs with ctypes
so that the parity tests and bench.py can drive it; no arithmetic and no tree logic lives here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LK_LIB_PATH = os.path.join(capi.LIB_DIR, "libphyhip_lk.so")


class t_node(C.Structure):
    pass


class t_edge(C.Structure):
    pass


t_node._fields_ = [("v", C.POINTER(t_node) * 3), ("b", C.POINTER(t_edge) * 3), ("num", C.c_int), ("tax", C.c_int)]
t_edge._fields_ = [("left", C.POINTER(t_node)), ("rght", C.POINTER(t_node)), ("num", C.c_int), ("l", C.c_double),
                   ("Pij_rr_idx", C.c_int), ("p_lk_left_idx", C.c_int), ("p_lk_rght_idx", C.c_int),
                   ("p_lk_tip_idx", C.c_int), ("update_partial_lk_left", C.c_short),
                   ("update_partial_lk_rght", C.c_short), ("Pij_rr", C.POINTER(C.c_double))]


class t_mod(C.Structure):
    _fields_ = [("ns", C.c_int), ("n_catg", C.c_int), ("pi", C.POINTER(C.c_double)),
                ("gamma_rr", C.POINTER(C.c_double)), ("gamma_r_proba", C.POINTER(C.c_double)),
                ("e_val", C.POINTER(C.c_double)), ("r_e_vect", C.POINTER(C.c_double)),
                ("l_e_vect", C.POINTER(C.c_double)), ("l_min", C.c_double), ("l_max", C.c_double),
                ("br_len_mult", C.c_double), ("invar", C.c_int), ("pinvar", C.c_double), ("use_m4mod", C.c_int)]


class t_tree(C.Structure):
    _fields_ = [("a_nodes", C.POINTER(C.POINTER(t_node))), ("a_edges", C.POINTER(C.POINTER(t_edge))),
                ("mod", C.POINTER(t_mod)), ("n_otu", C.c_int), ("n_pattern", C.c_int),
                ("wght", C.POINTER(C.c_double)), ("invar", C.POINTER(C.c_short)), ("b_inst", C.c_int),
                ("tip_root", C.c_int), ("both_sides", C.c_short), ("use_eigen_lr", C.c_short),
                ("update_eigen_lr", C.c_short), ("apply_lk_scaling", C.c_short), ("numerical_warning", C.c_short),
                ("host_pmat", C.c_short), ("c_lnL", C.c_double), ("old_lnL", C.c_double), ("c_dlnL", C.c_double),
                ("n_edges_traversed", C.c_int), ("spare_p_lk_idx", C.c_int), ("spare_Pij_idx", C.c_int),
                ("e_root", C.POINTER(t_edge)), ("do_alias_subpatt", C.c_short), ("update_alias_subpatt", C.c_short),
                ("alias_one_subpatt", C.c_void_p)]


_lib = None
_errors = []


@C.CFUNCTYPE(None, C.c_char_p)
def _exit_handler(msg):
    _errors.append(msg.decode(errors="replace"))


def load():
    global _lib
    if _lib is None:
        capi.load()  # libphyhip.so first (the host layer links against it)
        if not os.path.exists(LK_LIB_PATH):
            raise capi.PhyhipError(f"{LK_LIB_PATH} not built: run __graft_entry__.build()")
        L = C.CDLL(LK_LIB_PATH)
        L.Make_Tree_From_Edges.restype = C.POINTER(t_tree)
        L.Make_Model_Basic.restype = C.POINTER(t_mod)
        for f in ("Lk", "dLk", "Br_Len_Newton", "Update_Lk_At_Given_Edge"):
            getattr(L, f).restype = C.c_double
        # tests must survive the reference's print-and-Exit() convention
        L.Set_Exit_Handler(_exit_handler)
        _lib = L
    return _lib


def _raise_if_error():
    if _errors:
        msg = "".join(_errors)
        _errors.clear()
        raise capi.PhyhipError(msg.strip())


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class LkTree:
    """Owns a C `t_tree` + `t_mod`.  Methods are one-line forwards to the C functions of the same name."""

    def __init__(self, n_otu, edge_left, edge_rght, edge_len, n_pattern, ns, ncatg, device=None, node_v=None,
                 node_b=None, host_pmat=False, devices=None, force_sharded=False, class_axis=False, use_m4mod=False):
        L = load()
        self.L = L
        self.n, self.P, self.S, self.C = int(n_otu), int(n_pattern), int(ns), int(ncatg)
        el = np.ascontiguousarray(edge_left, dtype=np.int32); er = np.ascontiguousarray(edge_rght, dtype=np.int32)
        ln = np.ascontiguousarray(edge_len, dtype=np.float64)
        nv = None if node_v is None else np.ascontiguousarray(node_v, dtype=np.int32)
        nb = None if node_b is None else np.ascontiguousarray(node_b, dtype=np.int32)
        self.ne = len(el)
        self.tree = L.Make_Tree_From_Edges(self.n, el.ctypes.data_as(C.c_void_p), er.ctypes.data_as(C.c_void_p), _dp(ln),
                                           None if nv is None else nv.ctypes.data_as(C.c_void_p),
                                           None if nb is None else nb.ctypes.data_as(C.c_void_p))
        self.mod = L.Make_Model_Basic(self.S, self.C)
        self.tree.contents.mod = self.mod
        self.tree.contents.host_pmat = 1 if host_pmat else 0
        self.mod.contents.use_m4mod = 1 if use_m4mod else 0  # `phyml --cov`: the reference's generic loop (PHYHIP_FLAG_GENERIC_LOOP)
        self.device = -1 if device is None else int(device)
        # multi-GPU: pattern shards over `devices` inside libphyhip.so (one RCCL all-reduce per evaluation)
        self.devices = None if devices is None else [int(x) for x in devices]
        self.force_sharded = bool(force_sharded)
        self.class_axis = bool(class_axis)  # categories = classes of a mixture (PHYHIP_FLAG_CLASS_AXIS)
        self._made = False
        self.inst = None

    # --- model / data ------------------------------------------------------------------------------
    def set_model(self, pi, gamma_rr, gamma_r_proba, e_val, r_e_vect, l_e_vect, l_min=1e-8, l_max=100.0,
                  br_len_mult=1.0, apply_lk_scaling=1, invar_model=0, pinvar=0.0):
        m = self.mod.contents
        for dst, src, n in ((m.pi, pi, self.S), (m.gamma_rr, gamma_rr, self.C), (m.gamma_r_proba, gamma_r_proba, self.C),
                            (m.e_val, e_val, self.S), (m.r_e_vect, r_e_vect, self.S * self.S),
                            (m.l_e_vect, l_e_vect, self.S * self.S)):
            a = np.ascontiguousarray(src, dtype=np.float64).ravel()
            assert a.size == n
            C.memmove(dst, a.ctypes.data, 8 * n)
        m.l_min, m.l_max, m.br_len_mult = float(l_min), float(l_max), float(br_len_mult)
        m.invar, m.pinvar = int(invar_model), float(pinvar)
        self.tree.contents.apply_lk_scaling = int(apply_lk_scaling)
        if self._made:
            self.L.Update_Model_On_Device(self.tree)
            _raise_if_error()

    def Make_Tree_For_Lk(self, wght, invar=None):
        w = np.ascontiguousarray(wght, dtype=np.float64); assert w.size == self.P
        iv = None if invar is None else np.ascontiguousarray(invar, dtype=np.int16)
        ivp = None if iv is None else iv.ctypes.data_as(C.c_void_p)
        if self.devices is not None or self.class_axis:
            devs = self.devices if self.devices is not None else ([self.device] if self.device >= 0 else [])
            dv = (C.c_int * max(1, len(devs)))(*devs)
            self.L.Make_Tree_For_Lk_On_Devices(self.tree, self.P, _dp(w), ivp, dv, len(devs),
                                               (1 if self.force_sharded else 0) | (2 if self.class_axis else 0))
        else:
            self.L.Make_Tree_For_Lk(self.tree, self.P, _dp(w), ivp, self.device)
        _raise_if_error()
        self._made = True
        self.inst = _InstanceView(self.tree.contents.b_inst, self)

    def set_tips(self, tip_partials=None, tip_states=None, tip_chars=None):
        for t in range(self.n):
            if tip_chars is not None:  # compressed sequences as characters: the host layer's own encoders (src/lk.c:26-161)
                seq = bytes(np.ascontiguousarray(tip_chars[t], dtype=np.uint8)); assert len(seq) == self.P
                self.L.Init_Partial_Lk_Tips_Chars_One_Tip(self.tree, t, C.c_char_p(seq))
            elif tip_partials is not None:
                a = np.ascontiguousarray(tip_partials[t], dtype=np.float64); assert a.size == self.P * self.S
                self.L.Init_Partial_Lk_Tips_Double_One_Tip(self.tree, t, _dp(a))
            else:
                a = np.ascontiguousarray(tip_states[t], dtype=np.int32); assert a.size == self.P
                self.L.Init_Partial_Lk_Tips_States_One_Tip(self.tree, t, a.ctypes.data_as(C.c_void_p))
            _raise_if_error()

    def close(self):
        if self.tree is not None:
            if self._made:
                self.L.Free_Tree_Lk(self.tree)
            self.L.Free_Tree(self.tree)
            self.L.Free_Model(self.mod)
            self.tree = None

    # --- accessors ---------------------------------------------------------------------------------------
    def edge(self, e):
        return self.tree.contents.a_edges[e]

    def node(self, k):
        return self.tree.contents.a_nodes[k]

    @property
    def c_lnL(self):
        return self.tree.contents.c_lnL

    @property
    def c_dlnL(self):
        return self.tree.contents.c_dlnL

    @property
    def tip_root(self):
        return self.tree.contents.tip_root

    @tip_root.setter
    def tip_root(self, v):
        self.tree.contents.tip_root = int(v)

    # --- the surface (same names as src/lk.h) ------------------------------------------------------------------
    def Lk(self, b=None):
        v = self.L.Lk(None if b is None else self.edge(b), self.tree)
        _raise_if_error()
        return v

    def dLk(self, l, b):
        lv = C.c_double(l)
        v = self.L.dLk(C.byref(lv), self.edge(b), self.tree)
        _raise_if_error()
        return lv.value, v

    def Update_Partial_Lk(self, b, d):
        self.L.Update_Partial_Lk(self.tree, self.edge(b), self.node(d)); _raise_if_error()

    def set_root_edge(self, b):
        """Rooted input tree with the root ignored (tree->e_root; None: unrooted)."""
        self.tree.contents.e_root = self.edge(b) if b is not None else C.POINTER(t_edge)()

    def Update_Lk_At_Given_Edge(self, b):
        v = self.L.Update_Lk_At_Given_Edge(self.edge(b), self.tree); _raise_if_error()
        return v

    def Update_PMat_At_Given_Edge(self, b):
        self.L.Update_PMat_At_Given_Edge(self.edge(b), self.tree); _raise_if_error()

    def Post_Order_Lk(self, a, d):
        self.L.Post_Order_Lk(self.node(a), self.node(d), self.tree); _raise_if_error()

    def Pre_Order_Lk(self, a, d):
        self.L.Pre_Order_Lk(self.node(a), self.node(d), self.tree); _raise_if_error()

    def Update_All_Partial_Lk(self):
        self.L.Update_All_Partial_Lk(self.tree); _raise_if_error()

    def Update_Eigen_Lr(self, b):
        self.L.Update_Eigen_Lr(self.edge(b), self.tree); _raise_if_error()

    def Set_Both_Sides(self, yesno):
        self.L.Set_Both_Sides(int(bool(yesno)), self.tree)

    def Set_Use_Eigen_Lr(self, yesno):
        self.L.Set_Use_Eigen_Lr(int(bool(yesno)), self.tree)

    def Set_Update_Eigen_Lr(self, yesno):
        self.L.Set_Update_Eigen_Lr(int(bool(yesno)), self.tree)

    def Br_Len_Newton(self, b):
        lv = C.c_double(self.edge(b).contents.l)
        v = self.L.Br_Len_Newton(C.byref(lv), self.edge(b), self.tree)
        _raise_if_error()
        return lv.value, v

    def Replay_Surface_Trace(self, trace):
        """trace: dict of equal-length arrays kind,a,b,c,d,e (int32) and x (float64); returns (out, out2)."""
        n = len(trace["kind"])
        arr = {k: np.ascontiguousarray(trace[k], dtype=np.int32) for k in ("kind", "a", "b", "c", "d", "e")}
        x = np.ascontiguousarray(trace["x"], dtype=np.float64)
        out = np.zeros(n); out2 = np.zeros(n)
        ip = lambda v: v.ctypes.data_as(C.c_void_p)
        self.L.Replay_Surface_Trace(self.tree, n, ip(arr["kind"]), ip(arr["a"]), ip(arr["b"]), ip(arr["c"]), ip(arr["d"]),
                                    ip(arr["e"]), _dp(x), _dp(out), _dp(out2))
        _raise_if_error()
        return out, out2

    @property
    def spare_p_lk_idx(self):
        return self.tree.contents.spare_p_lk_idx

    @property
    def spare_Pij_idx(self):
        return self.tree.contents.spare_Pij_idx

    def Lk_Shard_Device(self, device_ptr):
        self.L.Lk_Shard_Device(self.tree, C.c_void_p(device_ptr)); _raise_if_error()

    # --- download hooks ---------------------------------------------------------------------------------------------
    def side_buffer(self, b, side):
        e = self.edge(b).contents
        return e.p_lk_left_idx if side == 0 else e.p_lk_rght_idx

    def partials(self, b, side):
        return self.inst.get_partials(self.side_buffer(b, side))

    def scale_factors(self, b, side):
        return self.inst.get_scale_factors(self.side_buffer(b, side))


class _InstanceView(capi.Instance):
    """capi.Instance methods on an instance id that the C host layer created (no ownership)."""

    def __init__(self, inst_id, tree: LkTree):
        self.L = capi.load()
        self.id = inst_id
        self.tips, self.S, self.P, self.C = tree.n, tree.S, tree.P, tree.C
        self.nmat = tree.ne + 4
        self.details = None

    def close(self):
        self.id = None
