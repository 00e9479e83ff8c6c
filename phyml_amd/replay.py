"""Seeded generator of SPR / branch-length-optimisation surface-call streams (the
trace-free fallback): the *call pattern* spr.c and optimiz.c drive through the likelihood surface, expressed at
buffer-index level so it can be replayed by `Replay_Surface_Trace` (device) and by the CPU oracle (tests).

Per regraft candidate (src/spr.c:643-646):   Update_PMat x2 (the two halves of the target edge) ->
Update_Partial_Lk(b_arrow, n_link) into a spare buffer -> Lk(b_arrow) = Update_PMat(pendant) + edge lnL.
Every `walk_every` candidates one real edge side is refreshed (the path update of spr.c:543).
Every `opt_every` candidates a Br_Len_Opt pattern follows (src/optimiz.c:622-632): Update_Eigen_Lr on the
candidate edge, then `n_dlk` dLk evaluations at different lengths.

The stream is a valid sequence of buffer operations on a tree whose partials are up to date on both sides; it
does not move subtrees (pruning/regrafting is caller-side pointer surgery, out of scope), so the scalars are
not the lnL of SPR-rearranged trees -- they are what the same calls return on either engine, which is what
per-call parity needs.
"""
from __future__ import annotations

import numpy as np

from . import synth

SET_PMAT, UPDATE, EDGE_LNL, EIGEN_LR, DLK, EIGEN_LNL = 0, 1, 2, 3, 4, 5


def side_buffer_map(n_otu, edge_left, edge_rght):
    """(edge, side) -> partials buffer index with the host layer's numbering (tips first, then internal
    edge sides in edge order; phl_lk.c Make_Tree_For_Lk)."""
    buf, nxt = {}, n_otu
    for e in range(len(edge_left)):
        l, r = int(edge_left[e]), int(edge_rght[e])
        if l < n_otu and r >= n_otu:
            l, r = r, l
        for side, node in ((0, l), (1, r)):
            if node < n_otu:
                buf[(e, side)] = node
            else:
                buf[(e, side)] = nxt
                nxt += 1
    return buf, nxt


def make_trace(n_otu, edge_left, edge_rght, edge_len, n_candidates, seed, walk_every=3, opt_every=0, n_dlk=8):
    el = np.asarray(edge_left); er = np.asarray(edge_rght); ln = np.asarray(edge_len, dtype=np.float64)
    ne = len(el)
    buf, n_real = side_buffer_map(n_otu, el, er)
    spare_buf, spare_mat = n_real, ne
    adj = [[] for _ in range(2 * n_otu - 2)]
    for e in range(ne):
        adj[int(el[e])].append((int(er[e]), e)); adj[int(er[e])].append((int(el[e]), e))
    internal_edges = [e for e in range(ne) if el[e] >= n_otu and er[e] >= n_otu]
    rec = {k: [] for k in ("kind", "a", "b", "c", "d", "e", "x")}

    def push(kind, a=0, b=0, c=0, d=0, e=0, x=0.0):
        for k, v in zip(("kind", "a", "b", "c", "d", "e", "x"), (kind, a, b, c, d, e, x)):
            rec[k].append(v)

    h = synth.hash_u64(seed, 77, np.arange(4 * n_candidates + 8))
    for i in range(n_candidates):
        prune = internal_edges[int(h[4 * i]) % len(internal_edges)]
        target = int(h[4 * i + 1]) % ne
        if target == prune:
            target = (target + 1) % ne
        u = float(int(h[4 * i + 2]) >> 11) / float(1 << 53)
        half = 0.5 * max(ln[target], 1e-4) * (0.5 + u)
        # Update_PMat on the two halves of the target edge, then the new node's partial into a spare buffer
        push(SET_PMAT, a=spare_mat, x=half)
        push(SET_PMAT, a=spare_mat + 1, x=max(ln[target], 1e-4) - half if ln[target] > half else half)
        push(UPDATE, a=spare_buf, b=buf[(target, 0)], c=spare_mat, d=buf[(target, 1)], e=spare_mat + 1)
        # Lk(b_arrow): refresh the pendant matrix, evaluate against the pruned subtree's partial
        pend = ln[prune] * (0.5 + u)
        push(SET_PMAT, a=spare_mat + 2, x=pend)
        push(EDGE_LNL, a=spare_buf, b=buf[(prune, 1)], c=spare_mat + 2)
        if walk_every and i % walk_every == walk_every - 1:
            # path update on a real edge side (recomputes the value it already holds)
            e = internal_edges[int(h[4 * i + 3]) % len(internal_edges)]
            d_node = int(el[e])
            ch = [(buf[(be, 1 if d_node == el[be] else 0)], be) for (v, be) in adj[d_node] if be != e]
            push(UPDATE, a=buf[(e, 0)], b=ch[0][0], c=ch[0][1], d=ch[1][0], e=ch[1][1])
        if opt_every and i % opt_every == opt_every - 1:
            # Br_Len_Opt on the candidate edge: eigen products once, then a series of dLk calls
            push(EIGEN_LR, a=spare_buf, b=buf[(prune, 1)])
            for j in range(n_dlk):
                push(DLK, x=pend * (0.25 + 0.25 * j))
    out = {k: np.array(v, dtype=np.float64 if k == "x" else np.int32) for k, v in rec.items()}
    return out


def recorded_trace(d):
    """The op stream of a trace recorded from a real PhyML run (oracle/trace_driver.c, tests/golden/trace_*.phyg) as the
    dict Replay_Surface_Trace takes, plus the scalars the reference returned: (trace, out, out2)."""
    tr = {k: np.ascontiguousarray(d["trace_" + k], dtype=np.int32) for k in ("kind", "a", "b", "c", "d", "e")}
    tr["x"] = np.ascontiguousarray(d["trace_x"], dtype=np.float64)
    return tr, np.asarray(d["trace_out"], dtype=np.float64), np.asarray(d["trace_out2"], dtype=np.float64)


def tips_from_masks(tip_mask, ns):
    """0/1 tip vectors [P][ns], digit states and ambiguity flags from allowed-state bit masks (one bit = unambiguous,
    src/lk.c:26-161)."""
    tip_mask = np.asarray(tip_mask)
    vecs, states, amb = [], [], []
    bits = (1 << np.arange(ns, dtype=np.int64))
    for t in range(tip_mask.shape[0]):
        m = tip_mask[t].astype(np.int64)
        v = ((m[:, None] & bits[None, :]) != 0).astype(np.float64)
        cnt = v.sum(axis=1)
        vecs.append(np.ascontiguousarray(v))
        amb.append((cnt != 1).astype(np.int16))
        states.append(np.where(cnt == 1, v.argmax(axis=1), 0).astype(np.int16))
    return vecs, states, amb


# ---- mixture models (class trees, src/mixt.c) -------------------------------------------------------------------------

def mixture_classes(d):
    """Per-class model blocks of a mixture dump (oracle/mixt_driver.c): each class tree is a plain single-category
    model whose branch lengths are multiplied by the class rate (src/lk.c:2298); returns (list of model dicts, list of
    coefficient factor tuples (proba, r_mat_weight, e_frq_weight))."""
    K = int(d["n_classes"][0])
    models, factors = [], []
    for k in range(K):
        g = lambda name: d[f"class{k}_{name}"]
        rate = float(g("rate")[0]) * float(g("own_gamma_rr")[0])
        models.append(dict(ns=d["ns"], ncatg=np.array([1.0]), pi=g("pi"), gamma_rr=np.array([rate]), gamma_r_proba=np.array([1.0]),
                           e_val=g("e_val"), r_e_vect=g("r_e_vect"), l_e_vect=g("l_e_vect"), l_min=g("l_min"), l_max=g("l_max"),
                           br_len_mult=g("br_len_mult"), invar_model=np.array([0.0]), pinvar=np.array([0.0]), datatype=d["datatype"]))
        factors.append((float(g("proba")[0]), float(g("r_mat_weight")[0]), float(g("e_frq_weight")[0])))
    return models, factors


def mixture_combine(unscaled, fact, factors, r_sum, e_sum, sum_probas, wght):
    """The site loop of MIXT_Lk after the per-class Lk_Core calls (src/mixt.c:1027-1135), no +I: class likelihoods are
    brought to a common scale with 2^-sum (sum capped at 1023, :1027-1034), weighted (:1048-1053, same operation order),
    floored at DBL_MIN (:1114-1117), logged and accumulated in site order (:1133).  Returns (lnL, per-site log-lk)."""
    P = len(wght)
    site_lk = np.zeros(P)
    for u, f, (proba, rw, ew) in zip(unscaled, fact, factors):
        s = np.minimum(np.asarray(f, dtype=np.float64), 1024.0)
        s = np.where(np.asarray(f) > 1024, 1023.0, s)
        x = np.asarray(u, dtype=np.float64) / np.power(2.0, s)
        site_lk = site_lk + x * proba * rw / r_sum * ew / e_sum / sum_probas
    site_lk = np.maximum(site_lk, np.finfo(np.float64).tiny)
    logs = np.log(site_lk)
    lnl = 0.0
    for p in range(P):
        lnl += wght[p] * logs[p]
    return lnl, logs


def mixture_dlk(dot_prods, facts, models, factors, r_sum, e_sum, sum_probas, wght, l):
    """MIXT_dLk (src/mixt.c:2962-3340), no +I, from the per-class eigen-basis products: returns (lnL, dlnL)."""
    P = len(wght)
    site_lk = np.zeros(P); site_dlk = np.zeros(P)
    l_min, l_max = float(models[0]["l_min"][0]), float(models[0]["l_max"][0])
    l = min(max(l, l_min), l_max)  # src/lk.c:672-673
    for dp, f, md, (proba, rw, ew) in zip(dot_prods, facts, models, factors):
        rr = 1.0 * float(md["br_len_mult"][0]) * float(md["gamma_rr"][0])  # src/mixt.c:3060-3063
        ln = min(max(l * rr, float(md["l_min"][0])), float(md["l_max"][0]))
        ev = np.asarray(md["e_val"], dtype=np.float64)
        ex = np.exp(ev * ln)
        dx = ex * ev * rr
        dp = np.asarray(dp, dtype=np.float64)
        # src/avx.c:250-276: even states accumulate in one lane pair, odd states in the other
        lk = (dp[:, 0::2] * ex[0::2]).sum(axis=1) + (dp[:, 1::2] * ex[1::2]).sum(axis=1)
        dlk = (dp[:, 0::2] * dx[0::2]).sum(axis=1) + (dp[:, 1::2] * dx[1::2]).sum(axis=1)
        s = np.where(np.asarray(f) > 1024, 1023.0, np.asarray(f, dtype=np.float64))
        mult = np.power(2.0, s)
        coef = lambda x: x * proba * rw / r_sum * ew / e_sum / sum_probas
        site_lk = site_lk + coef(lk / mult)
        site_dlk = site_dlk + coef(dlk / mult * 1.0)
    lnl = dlnl = 0.0
    for p in range(P):
        lnl += wght[p] * np.log(site_lk[p])
        dlnl += wght[p] * (site_dlk[p] / site_lk[p])
    return lnl, dlnl
