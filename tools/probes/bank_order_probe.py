#!/usr/bin/env python3
"""Prototype (CPU only): how far can a relabelling WITHIN equal-length classes cut the LDS bank conflicts of the ordered
recurrence's gather?  Cost model = graph.hip place_group (four planes: 16 bank groups of 16 bytes, lane sets of 16)."""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
import bench
from gcn_fmri_decoding_amd import graph as G

BLK = [0, 1, 1, 0, 1, 0, 0, 1, 2, 3, 3, 2, 3, 2, 2, 3]


def build(L):
    order = G.length_order(L)
    Lp = G.permute(L, order)
    ip, idx, _ = G.rescaled_laplacian_csr(Lp)
    return order, ip, idx


def sets_of(M, SQ):
    """set id of every vertex with a row (v < 4*SQ): (block, i, lane set)"""
    v = np.arange(4 * SQ)
    q, i = v >> 2, v & 3
    b, l = q >> 6, q & 63
    s = np.array(BLK)[l >> 2]
    return (b * 4 + i) * 4 + s, (i * SQ + q) % 16


def cost(ip, idx, SQ, lab=None):
    """lab: vertex -> label (None = identity).  Returns (lower bound, ideal) over all lane sets."""
    n = 4 * SQ
    nact = len(ip) - 1
    lens = np.diff(ip)
    if lab is None:
        lab = np.arange(nact)
    inv = np.empty_like(lab); inv[lab] = np.arange(len(lab))      # label -> vertex
    sid, cls = sets_of(nact, SQ)
    nsets = sid.max() + 1
    hist = np.zeros((nsets, 16), np.int64)
    rows = np.repeat(np.arange(nact), lens)                        # vertex of each entry
    rl = lab[rows]; cl = lab[idx]
    ok = rl < n
    np.add.at(hist, (sid[rl[ok]], cls[cl[ok]]), 1)
    # group length: max over the 64 lanes of (block, i)
    gid = sid >> 2
    glen = np.zeros(gid.max() + 1, np.int64)
    np.maximum.at(glen, gid[lab[np.arange(nact)][lab < n] if False else gid[np.arange(n)][:0]], 0) if False else None
    lab_len = np.zeros(n, np.int64); lab_len[lab[lab < n]] = lens[np.arange(nact)[lab < n]]
    np.maximum.at(glen, gid, lab_len)
    glen = (glen + 1) & ~1
    npos = np.where(glen <= 8, 8, np.where(glen <= 10, 10, 4 * ((glen + 3) // 4)))
    npos_set = npos[np.arange(nsets) >> 2]
    lb = np.maximum(npos_set, hist.max(axis=1)).sum()
    return int(lb), int(npos_set.sum()), hist


if __name__ == '__main__':
    Ls, _ = bench.load_graph(10000, 1, 0, 1, None)
    order, ip, idx = build(Ls[0])
    lens = np.diff(ip)
    nact = int((lens > 0).sum()); SQ = (nact + 3) // 4
    ip, idx = ip[:nact + 1], idx
    lb, ideal, hist = cost(ip, idx, SQ)
    print('active', nact, 'SQ', SQ, 'lower bound', lb, 'ideal', ideal, '(library: after 11449, ideal 6256)')
    # ---- local search: swap labels of two vertices of the same length inside one lane set; objective sum of squares of the
    # neighbour-class histograms
    rs = np.random.RandomState(0)
    lab = np.arange(nact)                       # vertex -> label
    sid, cls = sets_of(nact, SQ)
    nbr = [idx[ip[v]:ip[v + 1]] for v in range(nact)]
    # who references u: for symmetric graphs = nbr[u]
    members = {}
    for v in range(nact):
        members.setdefault(int(sid[v]), []).append(v)
    H = hist.copy()
    vert_at = np.arange(nact)                   # label -> vertex
    t0 = time.time()
    tries = acc = 0
    set_ids = list(members)
    for it in range(400000):
        S = set_ids[rs.randint(len(set_ids))]
        mem = members[S]
        a, b = rs.randint(len(mem)), rs.randint(len(mem))
        la, lb_ = mem[a], mem[b]                # labels
        if la == lb_:
            continue
        x, y = vert_at[la], vert_at[lb_]
        if lens[x] != lens[y]:
            continue
        tries += 1
        ca, cb = cls[la], cls[lb_]
        # sets that reference x (its neighbours' rows) lose class ca and gain cb; those referencing y the reverse
        sx = sid[lab[nbr[x]]]; sy = sid[lab[nbr[y]]]
        d = 0
        for s in sx:
            d += 2 * (H[s, cb] - H[s, ca]) + 2
            H[s, ca] -= 1; H[s, cb] += 1
        for s in sy:
            d += 2 * (H[s, ca] - H[s, cb]) + 2
            H[s, cb] -= 1; H[s, ca] += 1
        # x and y also swap sets only if in different sets -- same set here, their own rows stay in S
        if d < 0:
            acc += 1
            lab[x], lab[y] = lb_, la
            vert_at[la], vert_at[lb_] = y, x
        else:                                   # undo
            for s in sx:
                H[s, ca] += 1; H[s, cb] -= 1
            for s in sy:
                H[s, cb] += 1; H[s, ca] -= 1
        if it % 50000 == 0:
            npos_dummy = None
            print(it, 'accepted', acc, 'of', tries, 'sum max', int(H.max(axis=1).sum()), 'sumsq', int((H * H).sum()), '%.1fs' % (time.time() - t0), flush=True)
    lb2, ideal2, hist2 = cost(ip, idx, SQ, lab)
    assert np.array_equal(hist2, H)
    print('after local search: lower bound', lb2, 'ideal', ideal2)
