"""Bit-reproducible synthetic alignments and trees for the likelihood hot path.

SURVEY.md §8(d) asks for synthetic inputs that can be regenerated *identically* on the GPU box
(nothing of the reference travels there), so everything here is driven by an integer hash
(splitmix64 finaliser over (seed, stream, index) counters) -- no libm and no library PRNG in the
sampling path except one `exp` per (edge, rate class) that is quantised to a 32-bit threshold.

Conventions follow the reference's tree layout (src/make.c:418-423):
tips are nodes 0..n-1, internal nodes n..2n-3, an unrooted binary tree has 2n-3 edges and a tip
always sits on the *right* end of its edge.
"""
from __future__ import annotations

import dataclasses
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
NT_ALPHABET = "ACGT"
# state order of the reference's amino-acid tip table (src/lk.c:129-148)
AA_ALPHABET = "ARNDCQEGHILKMFPSTWYV"


def splitmix64(x):
    """splitmix64 output function on a uint64 array (wraps modulo 2**64)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hash_u64(seed: int, stream: int, idx):
    """Counter-based 64-bit hash of (seed, stream, idx)."""
    with np.errstate(over="ignore"):
        s = splitmix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF))
        s = splitmix64(s ^ np.uint64(stream & 0xFFFFFFFFFFFFFFFF))
        return splitmix64(s ^ (np.asarray(idx, dtype=np.uint64) * np.uint64(0xD6E8FEB86659FD93) & _M64))


@dataclasses.dataclass
class EdgeTree:
    """Unrooted binary tree as edge arrays (the form oracle/ref_driver.c dumps the reference's tree in)."""

    n_otu: int
    edge_left: np.ndarray  # int32 [2n-3]  (internal node unless n_otu == 2)
    edge_rght: np.ndarray  # int32 [2n-3]  (a tip is always on the right)
    edge_len: np.ndarray   # float64 [2n-3]
    names: list

    @property
    def n_edges(self):
        return 2 * self.n_otu - 3

    def neighbours(self):
        """node -> list of (neighbour node, edge id), in edge order."""
        nb = [[] for _ in range(2 * self.n_otu - 2)]
        for e in range(self.n_edges):
            a, b = int(self.edge_left[e]), int(self.edge_rght[e])
            nb[a].append((b, e))
            nb[b].append((a, e))
        return nb

    def to_newick(self, fmt="%.17g"):
        """Unrooted Newick with a trifurcation at the first internal node."""
        nb = self.neighbours()
        root = self.n_otu

        def sub(node, parent):
            if node < self.n_otu:
                return self.names[node]
            parts = []
            for (v, e) in nb[node]:
                if v != parent:
                    parts.append(sub(v, node) + ":" + (fmt % self.edge_len[e]))
            return "(" + ",".join(parts) + ")"

        import sys
        old = sys.getrecursionlimit()
        sys.setrecursionlimit(max(old, 4 * self.n_otu + 100))
        try:
            return sub(root, -1) + ";"
        finally:
            sys.setrecursionlimit(old)


def random_tree(n_otu: int, seed: int, lmin=0.02, lmax=0.15) -> EdgeTree:
    """Random sequential-insertion topology; branch lengths U(lmin,lmax) from the integer hash."""
    assert n_otu >= 3
    nxt_node = n_otu
    # edges as [a, b] with a internal
    edges = [[nxt_node, 0], [nxt_node, 1], [nxt_node, 2]]
    nxt_node += 1
    for t in range(3, n_otu):
        h = int(hash_u64(seed, 1, t))
        e = h % len(edges)
        a, b = edges[e]
        m = nxt_node
        nxt_node += 1
        edges[e] = [a, m]          # a -- m   (a is internal; m internal)
        edges.append([m, b])       # m -- b   (b may be a tip -> stays on the right)
        edges.append([m, t])       # m -- new tip
    ne = 2 * n_otu - 3
    assert len(edges) == ne
    u = hash_u64(seed, 2, np.arange(ne)) >> np.uint64(11)          # 53 random bits
    lens = lmin + (lmax - lmin) * (u.astype(np.float64) / float(1 << 53))
    el = np.array([a for a, _ in edges], dtype=np.int32)
    er = np.array([b for _, b in edges], dtype=np.int32)
    # tip on the right, always
    for k in range(ne):
        if el[k] < n_otu:
            el[k], er[k] = er[k], el[k]
    names = ["T%05d" % i for i in range(n_otu)]
    return EdgeTree(n_otu, el, er, lens, names)


_SIM_RATES = (0.1369537815, 0.4767518619, 1.0, 2.3862943611)


def simulate_states(tree: EdgeTree, n_sites: int, n_states: int, seed: int, rate_scale=1.0, site_offset=0) -> np.ndarray:
    """Evolve uniform-root states down the tree under an equal-rates (JC-like) process with 4 site-rate
    classes.  Returns uint8 [n_otu, n_sites] state indices.  Deterministic given (tree, seed); every column
    depends only on its own index, so columns [site_offset, site_offset + n_sites) of a longer alignment can be
    generated on their own (pattern shards of the multi-GPU configuration)."""
    nb = tree.neighbours()
    site = np.arange(site_offset, site_offset + n_sites, dtype=np.uint64)
    rate_cls = (hash_u64(seed, 10, site) >> np.uint64(62)).astype(np.int64)  # 0..3
    out = np.zeros((tree.n_otu, n_sites), dtype=np.uint8)
    root = tree.n_otu
    root_state = (hash_u64(seed, 11, site) % np.uint64(n_states)).astype(np.uint8)
    stack = [(root, -1, root_state)]
    mu = n_states / (n_states - 1.0)
    while stack:
        node, parent, st = stack.pop()
        if node < tree.n_otu:
            out[node] = st
            continue
        for (v, e) in nb[node]:
            if v == parent:
                continue
            # P(no event) per rate class, quantised to 32 bits
            thr = np.array([int(np.floor(np.exp(-mu * tree.edge_len[e] * r * rate_scale) * 4294967296.0))
                            for r in _SIM_RATES], dtype=np.uint64)
            h = hash_u64(seed, 100 + e, site)
            keep = (h & np.uint64(0xFFFFFFFF)) < thr[rate_cls]
            new = ((h >> np.uint64(32)) % np.uint64(n_states)).astype(np.uint8)
            stack.append((v, node, np.where(keep, st, new).astype(np.uint8)))
    return out


def states_checksum(states: np.ndarray) -> int:
    """Order-sensitive 64-bit checksum of a state matrix (to prove the box regenerated the same input)."""
    flat = np.ascontiguousarray(states, dtype=np.uint8).ravel().astype(np.uint64)
    idx = np.arange(flat.size, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = splitmix64(flat + (idx << np.uint64(8)))
        return int(np.bitwise_xor.reduce(h) ^ np.uint64(flat.size))


def states_to_chars(states: np.ndarray, n_states: int) -> np.ndarray:
    alpha = NT_ALPHABET if n_states == 4 else AA_ALPHABET
    lut = np.frombuffer(alpha.encode(), dtype=np.uint8)
    return lut[states]


def write_phylip(path: str, names, chars: np.ndarray):
    """Sequential PHYLIP, one line per taxon (what the reference's reader accepts, src/io.c:1052)."""
    n, L = chars.shape
    with open(path, "w") as f:
        f.write(f"{n} {L}\n")
        for i in range(n):
            f.write(names[i] + "  " + chars[i].tobytes().decode() + "\n")
