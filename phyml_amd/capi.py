"""ctypes binding of libphyhip.so (the C ABI in include/phyhip.h).

Python is only the test / benchmark harness here; the product is the shared library.  Loading fails
loudly when the library has not been built (`python -c 'import __graft_entry__ as g; g.build()'`);
there is no fallback implementation.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PHYHIP_LIBDIR: another build of the same libraries (tools/build_diag.sh puts the -DPHYHIP_DIAG build in phyml_amd/lib_diag)
LIB_DIR = os.environ.get("PHYHIP_LIBDIR") or os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libphyhip.so")


class PhyhipError(RuntimeError):
    pass


class Operation(C.Structure):
    _fields_ = [("destinationPartials", C.c_int), ("destinationScaleWrite", C.c_int), ("destinationScaleRead", C.c_int),
                ("child1Partials", C.c_int), ("child1TransitionMatrix", C.c_int),
                ("child2Partials", C.c_int), ("child2TransitionMatrix", C.c_int)]


class InstanceDetails(C.Structure):
    _fields_ = [("resourceNumber", C.c_int), ("resourceName", C.c_char * 64), ("implName", C.c_char * 64),
                ("flags", C.c_long), ("computeUnits", C.c_int), ("globalMemBytes", C.c_longlong)]


# every symbol include/phyhip.h declares (tests check the library exports them all)
SYMBOLS = [
    "phyhip_create_instance", "phyhip_finalize_instance", "phyhip_get_last_error", "phyhip_set_tip_partials",
    "phyhip_set_tip_states", "phyhip_set_tip_partials_at_pattern", "phyhip_set_partials", "phyhip_set_pattern_weights", "phyhip_set_category_rates",
    "phyhip_set_category_weights", "phyhip_set_state_frequencies", "phyhip_set_eigen_decomposition",
    "phyhip_set_phyml_options", "phyhip_set_invariant_sites", "phyhip_update_transition_matrices",
    "phyhip_set_transition_matrix", "phyhip_get_transition_matrix", "phyhip_update_partials",
    "phyhip_calculate_edge_log_likelihoods", "phyhip_calculate_edge_log_likelihoods_device",
    "phyhip_get_site_log_likelihoods", "phyhip_get_site_outputs", "phyhip_get_partials", "phyhip_get_scale_factors",
    "phyhip_set_scale_factors", "phyhip_get_numerical_warning", "phyhip_update_eigen_lr",
    "phyhip_calculate_eigen_lnl_dlnl", "phyhip_calculate_eigen_lnl", "phyhip_get_dot_prod", "phyhip_set_stream",
    "phyhip_synchronize", "phyhip_profile", "phyhip_profile_read", "phyhip_calculate_mixture_log_likelihood",
    "phyhip_calculate_mixture_eigen_lnl_dlnl", "phyhip_comm_get_unique_id", "phyhip_comm_init_rank", "phyhip_comm_size",
    "phyhip_get_shard_range", "phyhip_profile_read_kernel", "phyhip_profile_read_collective", "phyhip_profile_read_traffic", "phyhip_profile_read_eigen", "phyhip_get_resident_stats", "phyhip_get_big_resident_stats", "phyhip_set_virtual_buffers", "phyhip_get_virtual_stats", "phyhip_calculate_class_mixture_log_likelihood",
    "phyhip_calculate_class_mixture_eigen_lnl_dlnl", "phyhip_get_class_scale_factors", "phyhip_set_mixture_invariant_sites",
]

FLAG_SHARDED = 1 << 40  # PHYHIP_FLAG_SHARDED
FLAG_CLASS_AXIS = 1 << 41  # PHYHIP_FLAG_CLASS_AXIS
UNIQUE_ID_BYTES = 128

_lib = None


def comm_get_unique_id() -> bytes:
    """phyhip_comm_get_unique_id (rank 0; broadcast the bytes to the other ranks)."""
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    _chk(load().phyhip_comm_get_unique_id(buf))
    return buf.raw


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PhyhipError(f"{LIB_PATH} not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
        L = C.CDLL(LIB_PATH)
        L.phyhip_get_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _chk(rc):
    if rc < 0:
        raise PhyhipError(f"phyhip error {rc}: {load().phyhip_get_last_error().decode()}")
    return rc


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Instance:
    """Thin object wrapper: one method per C entry point, numpy in / numpy out."""

    def __init__(self, tip_count, partials_buffer_count, state_count, pattern_count, matrix_buffer_count,
                 category_count, device=None, devices=None, force_sharded=False, class_axis=False):
        L = load()
        self.L = L
        self.tips, self.nbuf, self.S, self.P, self.nmat, self.C = (tip_count, partials_buffer_count, state_count,
                                                                  pattern_count, matrix_buffer_count, category_count)
        self.details = InstanceDetails()
        if devices is not None:  # sharded instance: one pattern range per listed device, RCCL all-reduce inside the library
            res, nres = (C.c_int * len(devices))(*[int(d) for d in devices]), len(devices)
        else:
            res, nres = ((C.c_int * 1)(device), 1) if device is not None else (None, 0)
        self.id = _chk(L.phyhip_create_instance(tip_count, partials_buffer_count, 0, state_count, pattern_count, 1,
                                                matrix_buffer_count, category_count, 0, res, nres,
                                                C.c_long(0), C.c_long((FLAG_SHARDED if force_sharded else 0) | (FLAG_CLASS_AXIS if class_axis else 0)),
                                                C.byref(self.details)))

    def close(self):
        if self.id is not None:
            self.L.phyhip_finalize_instance(self.id)
            self.id = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- inputs
    def set_tip_partials(self, tip, partials):
        a = _f64(partials); assert a.size == self.P * self.S
        _chk(self.L.phyhip_set_tip_partials(self.id, tip, _ptr(a)))

    def set_tip_partials_at_pattern(self, tip, pattern, partials):
        a = _f64(partials); assert a.size == self.S
        _chk(self.L.phyhip_set_tip_partials_at_pattern(self.id, int(tip), int(pattern), _ptr(a)))

    def set_tip_states(self, tip, states):
        a = np.ascontiguousarray(states, dtype=np.int32); assert a.size == self.P
        _chk(self.L.phyhip_set_tip_states(self.id, tip, _ptr(a)))

    def set_partials(self, buf, partials):
        a = _f64(partials); assert a.size == self.P * self.C * self.S
        _chk(self.L.phyhip_set_partials(self.id, buf, _ptr(a)))

    def set_pattern_weights(self, w):
        a = _f64(w); assert a.size == self.P
        _chk(self.L.phyhip_set_pattern_weights(self.id, _ptr(a)))

    def set_category_rates(self, r):
        a = _f64(r); assert a.size == self.C
        _chk(self.L.phyhip_set_category_rates(self.id, _ptr(a)))

    def set_category_weights(self, w):
        a = _f64(w); assert a.size == self.C
        _chk(self.L.phyhip_set_category_weights(self.id, 0, _ptr(a)))

    def set_state_frequencies(self, pi, index=0):
        a = _f64(pi); assert a.size == self.S
        _chk(self.L.phyhip_set_state_frequencies(self.id, int(index), _ptr(a)))

    def set_eigen_decomposition(self, evec, ivec, evals, index=0):
        a, b, c = _f64(evec), _f64(ivec), _f64(evals)
        assert a.size == self.S * self.S and b.size == self.S * self.S and c.size == self.S
        _chk(self.L.phyhip_set_eigen_decomposition(self.id, int(index), _ptr(a), _ptr(b), _ptr(c)))

    # -- mixtures on the class axis (instance created with class_axis=True)
    def class_mixture_log_likelihood(self, parent, child, pm, proba, r_w, e_w, r_sum, e_sum, sum_probas):
        n = self.C
        da = lambda v: (C.c_double * n)(*[float(x) for x in v])
        out = C.c_double(0.0)
        _chk(self.L.phyhip_calculate_class_mixture_log_likelihood(self.id, int(parent), int(child), int(pm), da(proba), da(r_w), da(e_w),
                                                                 C.c_double(r_sum), C.c_double(e_sum), C.c_double(sum_probas), C.byref(out)))
        return out.value

    def class_mixture_eigen_lnl_dlnl(self, left, right, l, proba, r_w, e_w, r_sum, e_sum, sum_probas):
        n = self.C
        da = lambda v: (C.c_double * n)(*[float(x) for x in v])
        lv, lnl, dlnl = C.c_double(l), C.c_double(0.0), C.c_double(0.0)
        _chk(self.L.phyhip_calculate_class_mixture_eigen_lnl_dlnl(self.id, int(left), int(right), C.byref(lv), da(proba), da(r_w), da(e_w),
                                                                 C.c_double(r_sum), C.c_double(e_sum), C.c_double(sum_probas),
                                                                 C.byref(lnl), C.byref(dlnl)))
        return lv.value, lnl.value, dlnl.value

    def get_class_scale_factors(self, buf, k):
        out = np.zeros(self.P, np.int32)
        _chk(self.L.phyhip_get_class_scale_factors(self.id, int(buf), int(k), _ptr(out)))
        return out

    def set_phyml_options(self, l_min=1e-8, l_max=100.0, br_len_mult=1.0, apply_lk_scaling=1):
        _chk(self.L.phyhip_set_phyml_options(self.id, C.c_double(l_min), C.c_double(l_max), C.c_double(br_len_mult),
                                             int(apply_lk_scaling)))

    def set_invariant_sites(self, invar_model, pinvar, invar):
        a = None if invar is None else np.ascontiguousarray(invar, dtype=np.int16)
        _chk(self.L.phyhip_set_invariant_sites(self.id, int(invar_model), C.c_double(pinvar), _ptr(a)))

    # -- matrices
    def update_transition_matrices(self, indices, lengths):
        i = np.ascontiguousarray(indices, dtype=np.int32); l = _f64(lengths); assert i.size == l.size
        _chk(self.L.phyhip_update_transition_matrices(self.id, 0, _ptr(i), None, None, _ptr(l), int(i.size)))

    def set_transition_matrix(self, idx, mat):
        a = _f64(mat); assert a.size == self.C * self.S * self.S
        _chk(self.L.phyhip_set_transition_matrix(self.id, int(idx), _ptr(a), C.c_double(0.0)))

    def get_transition_matrix(self, idx):
        out = np.zeros((self.C, self.S, self.S))
        _chk(self.L.phyhip_get_transition_matrix(self.id, int(idx), _ptr(out)))
        return out

    # -- hot path
    def update_partials(self, ops):
        """ops: iterable of (dest, child1, pm1, child2, pm2)."""
        ops = list(ops)
        arr = (Operation * len(ops))()
        for k, (d, c1, m1, c2, m2) in enumerate(ops):
            arr[k] = Operation(d, -1, -1, c1, m1, c2, m2)
        _chk(self.L.phyhip_update_partials(self.id, arr, len(ops), -1))

    def edge_lnl(self, parent, child, pm):
        out = C.c_double(0)
        p = (C.c_int * 1)(parent); c = (C.c_int * 1)(child); m = (C.c_int * 1)(pm); z = (C.c_int * 1)(0)
        _chk(self.L.phyhip_calculate_edge_log_likelihoods(self.id, p, c, m, None, None, z, z, None, 1, C.byref(out), None, None))
        return out.value

    def edge_lnl_device(self, parent, child, pm, device_ptr):
        _chk(self.L.phyhip_calculate_edge_log_likelihoods_device(self.id, parent, child, pm, C.c_void_p(device_ptr)))

    def site_log_likelihoods(self):
        out = np.zeros(self.P)
        _chk(self.L.phyhip_get_site_log_likelihoods(self.id, _ptr(out)))
        return out

    def site_outputs(self, n_fact=1):
        """n_fact: class-axis instances return fact_sum_scale as [class][pattern] (n_fact = class count)."""
        a = np.zeros(self.P); b = np.zeros(self.P); c = np.zeros((self.P, self.C)); f = np.zeros(self.P * n_fact, np.int32)
        _chk(self.L.phyhip_get_site_outputs(self.id, _ptr(a), _ptr(b), _ptr(c), _ptr(f)))
        return a, b, c, f

    def get_partials(self, buf):
        out = np.zeros((self.P, self.C * self.S))
        _chk(self.L.phyhip_get_partials(self.id, int(buf), -1, _ptr(out)))
        return out

    def get_scale_factors(self, buf):
        out = np.zeros(self.P, np.int32)
        _chk(self.L.phyhip_get_scale_factors(self.id, int(buf), _ptr(out)))
        return out

    def numerical_warning(self):
        w = C.c_int(0)
        _chk(self.L.phyhip_get_numerical_warning(self.id, C.byref(w)))
        return w.value

    # -- eigen basis
    def update_eigen_lr(self, left, rght):
        _chk(self.L.phyhip_update_eigen_lr(self.id, int(left), int(rght)))

    def eigen_lnl_dlnl(self, l):
        lv = C.c_double(l); a = C.c_double(0); b = C.c_double(0)
        _chk(self.L.phyhip_calculate_eigen_lnl_dlnl(self.id, C.byref(lv), C.byref(a), C.byref(b)))
        return lv.value, a.value, b.value

    def eigen_lnl(self, l):
        a = C.c_double(0)
        _chk(self.L.phyhip_calculate_eigen_lnl(self.id, C.c_double(l), C.byref(a)))
        return a.value

    def get_dot_prod(self):
        out = np.zeros((self.P, self.C * self.S))
        _chk(self.L.phyhip_get_dot_prod(self.id, _ptr(out)))
        return out

    # -- multi-GPU
    def comm_init_rank(self, nranks, rank, unique_id: bytes):
        assert len(unique_id) == UNIQUE_ID_BYTES
        _chk(self.L.phyhip_comm_init_rank(self.id, int(nranks), int(rank), C.c_char_p(unique_id)))

    def comm_size(self):
        n = C.c_int(0)
        _chk(self.L.phyhip_comm_size(self.id, C.byref(n)))
        return n.value

    def shard_ranges(self):
        """[(device, first pattern, pattern count)] of the instance's shards."""
        out, k, n = [], 0, 1
        while k < n:
            d, lo, cnt = C.c_int(0), C.c_int(0), C.c_int(0)
            n = _chk(self.L.phyhip_get_shard_range(self.id, k, C.byref(d), C.byref(lo), C.byref(cnt)))
            out.append((d.value, lo.value, cnt.value))
            k += 1
        return out

    # -- plumbing
    def set_stream(self, stream_handle):
        _chk(self.L.phyhip_set_stream(self.id, C.c_void_p(stream_handle)))

    def synchronize(self):
        _chk(self.L.phyhip_synchronize(self.id))

    def profile(self, enable):
        _chk(self.L.phyhip_profile(self.id, int(enable)))

    def profile_read(self):
        ms = C.c_double(0); n = C.c_int(0); u = C.c_double(0)
        _chk(self.L.phyhip_profile_read(self.id, C.byref(ms), C.byref(n), C.byref(u)))
        return ms.value, n.value, u.value

    def resident_stats(self, which=0):
        """(evaluations served by the resident workgroups, their launches, unanswered commands, evaluations launched instead)
        of the dLk evaluator; resident_stats(1): of the short-evaluation one; resident_stats(2): of the large-grid one"""
        if which == 2:  # the large-grid evaluator (phyhip_big.hpp)
            big = (C.c_longlong * 4)()
            _chk(self.L.phyhip_get_big_resident_stats(self.id, big))
            return tuple(int(v) for v in big)
        out = (C.c_longlong * 8)()
        _chk(self.L.phyhip_get_resident_stats(self.id, out))
        return tuple(int(v) for v in out[4 * which:4 * which + 4])

    def set_virtual_buffers(self, min_operations):
        """phyhip_set_virtual_buffers: traversal launches of at least this many operations leave tip x tip results virtual
        (0: never, and what is virtual is stored)"""
        _chk(self.L.phyhip_set_virtual_buffers(self.id, int(min_operations)))

    def virtual_stats(self):
        """(buffers virtual now, stores skipped, non-storing re-issues, storing re-issues)"""
        out = (C.c_longlong * 4)()
        _chk(self.L.phyhip_get_virtual_stats(self.id, out))
        return tuple(int(v) for v in out)

    def profile_read_eigen(self):
        """(ms, launches) of eigen_lr_kernel and of dlk_kernel since profile(1)"""
        a = C.c_double(0); an = C.c_int(0); b = C.c_double(0); bn = C.c_int(0)
        _chk(self.L.phyhip_profile_read_eigen(self.id, C.byref(a), C.byref(an), C.byref(b), C.byref(bn)))
        return (a.value, an.value), (b.value, bn.value)

    def profile_read_collective(self):
        """(ms, evaluations, ranks) of the collective path (local sum, all-reduce, publish) since profile(1)"""
        ms = C.c_double(0); n = C.c_int(0); r = C.c_int(0)
        _chk(self.L.phyhip_profile_read_collective(self.id, C.byref(ms), C.byref(n), C.byref(r)))
        return ms.value, n.value, r.value

    def profile_read_kernel(self):
        """name of the traversal kernel of the last profiled launch (template arguments included)"""
        buf = C.create_string_buffer(128)
        _chk(self.L.phyhip_profile_read_kernel(self.id, buf, 128))
        return buf.value.decode()

    def profile_read_traffic(self):
        r = C.c_double(0); w = C.c_double(0)
        _chk(self.L.phyhip_profile_read_traffic(self.id, C.byref(r), C.byref(w)))
        return r.value, w.value


def mixture_log_likelihood(instance_ids, parents, children, matrices, proba, r_mat_weight, e_frq_weight, r_sum, e_sum, sum_probas):
    """phyhip_calculate_mixture_log_likelihood: MIXT_Lk over class instances (one category each)."""
    L = load()
    n = len(instance_ids)
    ia = lambda v: (C.c_int * n)(*[int(x) for x in v])
    da = lambda v: (C.c_double * n)(*[float(x) for x in v])
    out = C.c_double(0.0)
    _chk(L.phyhip_calculate_mixture_log_likelihood(ia(instance_ids), n, ia(parents), ia(children), ia(matrices), da(proba),
                                                   da(r_mat_weight), da(e_frq_weight), C.c_double(r_sum), C.c_double(e_sum),
                                                   C.c_double(sum_probas), C.byref(out)))
    return out.value


def mixture_log_likelihood_classes(instance_ids, parents, children, matrices, proba, r_mat_weight, e_frq_weight, r_sum, e_sum, sum_probas):
    """phyhip_calculate_mixture_log_likelihood with class-axis instances among the entries: the per-class tables are longer than
    the instance list (an entry stands for its C classes)."""
    L = load()
    n = len(instance_ids)
    ia = lambda v: (C.c_int * len(v))(*[int(x) for x in v])
    da = lambda v: (C.c_double * len(v))(*[float(x) for x in v])
    out = C.c_double(0.0)
    _chk(L.phyhip_calculate_mixture_log_likelihood(ia(instance_ids), n, ia(parents), ia(children), ia(matrices), da(proba),
                                                   da(r_mat_weight), da(e_frq_weight), C.c_double(r_sum), C.c_double(e_sum),
                                                   C.c_double(sum_probas), C.byref(out)))
    return out.value


def mixture_eigen_lnl_dlnl(instance_ids, lefts, rights, l, proba, r_mat_weight, e_frq_weight, r_sum, e_sum, sum_probas):
    """phyhip_calculate_mixture_eigen_lnl_dlnl: MIXT_dLk over class instances; returns (clamped l, lnL, dlnL)."""
    L = load()
    n = len(instance_ids)
    ia = lambda v: (C.c_int * n)(*[int(x) for x in v])
    da = lambda v: (C.c_double * n)(*[float(x) for x in v])
    lv, lnl, dlnl = C.c_double(l), C.c_double(0.0), C.c_double(0.0)
    _chk(L.phyhip_calculate_mixture_eigen_lnl_dlnl(ia(instance_ids), n, ia(lefts), ia(rights), C.byref(lv), da(proba),
                                                   da(r_mat_weight), da(e_frq_weight), C.c_double(r_sum), C.c_double(e_sum),
                                                   C.c_double(sum_probas), C.byref(lnl), C.byref(dlnl)))
    return lv.value, lnl.value, dlnl.value
