"""A third, independent reading of K9's path upgrade (reference: /root/reference/src/paf_data.cpp:750-792 `internal_shortest_path_recover`,
:795-921 `upgrade_edge_path_with_alt_path`) - plain Python, own shape, shares no code with oracle/ or with the product.

Why it exists: `paf_data.cpp:739+` cannot be compiled in this image (ankerl/unordered_dense.h is absent), so K9 is the one
stage for which "HIP == oracle" does not mean "== reference" (DESIGN.md section 2).  This module does NOT pin K9 to the reference
either; it is one more reading of the same source text, written differently, plus two things a reading cannot get wrong:

  * EXHAUSTIVE minimality: for every window DP whose window holds at most `brute_max` vertices, every a -> b path inside the
    window is enumerated (with the whitelist rule on the last hop), and no enumerated path may be smaller under the
    QRY_SCORE_MODE order (paf_data.hpp:142-159) than the path the product took; the product's path must be one of the enumerated
    ones and its distance the minimum;
  * the reference's own Debug asserts on the upgraded path (:913-918): first tail = src, last head = dest, consecutive edges
    chained, every edge an edge of the graph.

Inputs are the product's (or the emulation's) intermediates: the CSR graph, the forward Kahn order, the walk as recovered
from the heaps (`pathA`) and the upgraded path (`pathB`).  Everything up to `pathA` is pinned to the reference (K1 ... K8)."""
import numpy as np


class Dist:
    """PafDistance (paf_data.hpp:121-189) as a tuple with the QRY_SCORE_MODE order."""
    __slots__ = ("q", "r", "a", "nz", "tot")

    def __init__(self, q=0, r=0, a=0, nz=0, tot=0):
        self.q, self.r, self.a, self.nz, self.tot = q, r, a, nz, tot

    def __add__(self, o):
        return Dist(self.q + o.q, self.r + o.r, self.a + o.a, self.nz + o.nz, self.tot + o.tot)

    def less_qry(self, o):
        """operator< in QRY_SCORE_MODE for two real distances (neither is max())."""
        if self.q != o.q:
            return self.q < o.q
        if self.r != o.r:
            return self.r < o.r
        if self.a != o.a:
            return self.a < o.a
        return self.nz * (o.tot or 1) > o.nz * (self.tot or 1)

    def key(self):
        return (self.q, self.r, self.a, self.nz, self.tot)


class Graph:
    """One contig's graph from the product's arrays (local vertex ids; src = V - 2, dest = V - 1)."""

    def __init__(self, rowptr, col, wq, wr, fl, v_i, v_j, fwd_order):
        self.V = len(rowptr) - 1
        self.rowptr, self.col, self.wq, self.wr, self.fl = rowptr, col, wq, wr, fl
        self.v_i, self.v_j = v_i, v_j
        self.order = fwd_order                                        # position -> vertex
        self.pos = np.empty(self.V, np.int64)
        self.pos[fwd_order] = np.arange(self.V)
        self.src, self.dest = self.V - 2, self.V - 1

    def out(self, u):
        for e in range(int(self.rowptr[u]), int(self.rowptr[u + 1])):
            f = int(self.fl[e])
            yield int(self.col[e]), Dist(int(self.wq[e]), int(self.wr[e]), f & 3, (f >> 2) & 1, (f >> 3) & 1)

    def has_edge(self, u, v):
        return v in self.col[int(self.rowptr[u]):int(self.rowptr[u + 1])]


def _hop_allowed(g, u, v, b, wl):
    """:767-773: with a whitelist the hop INTO b must come from a record vertex whose second index is the whitelisted record."""
    if wl is None or v != b:
        return True
    if u == g.src or u == g.dest:
        return False
    return int(g.v_j[u]) == wl


def window_dp(g, a, b, wl=None):
    """internal_shortest_path_recover (:750-792) as written: first-wins relaxation over the forward order.  Returns the vertex list
    a ... b ([] when a == b) and the distance of b."""
    if a == b:
        return [], None
    dist, pre = {a: Dist()}, {a: -1}
    for p in range(int(g.pos[a]), int(g.pos[b])):
        u = int(g.order[p])
        if u not in dist:
            continue
        du = dist[u]
        for v, w in g.out(u):
            if not _hop_allowed(g, u, v, b, wl):
                continue
            nd = du + w
            if v not in dist or nd.less_qry(dist[v]):
                dist[v], pre[v] = nd, u
    assert b in dist, "window DP: b unreachable (the reference's Debug assert :783)"
    path, x = [b], b
    while x != a:
        x = pre[x]
        path.append(x)
    return path[::-1], dist[b]


def enumerate_paths(g, a, b, wl=None, limit=200000):
    """Every a -> b path (a DAG: all of them stay inside the window of topological positions); None when there are too many."""
    out, stack = [], [(a, [a], Dist())]
    pb = int(g.pos[b])
    while stack:
        u, path, d = stack.pop()
        for v, w in g.out(u):
            if int(g.pos[v]) > pb or not _hop_allowed(g, u, v, b, wl):
                continue
            if v == b:
                out.append((path + [v], d + w))
                if len(out) > limit:
                    return None
            else:
                stack.append((v, path + [v], d + w))
    return out


def check_conversion(g, pathA, pathB, brute_max=12, stats=None):
    """pathA / pathB: lists of (u, v) edges (the walk from the heaps / the product's upgraded path).  Returns a list of findings
    (empty = fine).  Follows :801-912 edge by edge, computing every alt path with `window_dp`; wherever the window is small,
    the result is also checked against the exhaustive enumeration."""
    bad = []
    stats = stats if stats is not None else {}
    src, dest = g.src, g.dest
    # ---- the reference's asserts on the result (:913-918) + every edge exists
    if not pathB or pathB[0][0] != src or pathB[-1][1] != dest:
        return ["upgraded path does not run src -> dest"]
    for (u0, v0), (u1, v1) in zip(pathB, pathB[1:]):
        if v0 != u1:
            bad.append("upgraded path is not chained at %d -> %d | %d -> %d" % (u0, v0, u1, v1))
    for u, v in pathB:
        if not g.has_edge(u, v):
            bad.append("upgraded path uses %d -> %d, which is not an edge" % (u, v))
    if bad:
        return bad
    # ---- this reading's own upgrade of pathA
    mine = []                                                         # vertex pairs

    def alt(a, b, wl, drop_last, fallback):
        vs, d = window_dp(g, a, b, wl)
        if not vs:
            mine.extend(fallback)
            return
        nwin = int(g.pos[b]) - int(g.pos[a]) + 1
        stats["dp"] = stats.get("dp", 0) + 1
        if nwin <= brute_max:
            allp = enumerate_paths(g, a, b, wl)
            if allp is not None:
                stats["brute"] = stats.get("brute", 0) + 1
                stats["brute_paths"] = stats.get("brute_paths", 0) + len(allp)
                if vs not in [p for p, _ in allp]:
                    bad.append("window %d -> %d: the DP's path is not an a -> b path of the graph" % (a, b))
                for p, dd in allp:
                    if dd.less_qry(d):
                        bad.append("window %d -> %d (wl %r): path %r is smaller than the DP's %r" % (a, b, wl, p, vs))
                        break
                best = min(allp, key=lambda t: (t[1].q, t[1].r, t[1].a))
                if (best[1].q, best[1].r, best[1].a) != (d.q, d.r, d.a):
                    bad.append("window %d -> %d: DP distance %r is not the minimum %r" % (a, b, d.key(), best[1].key()))
        es = list(zip(vs, vs[1:]))
        if drop_last:
            es = es[:-1]
        mine.extend(es)

    it, n = 0, len(pathA)
    if n < 2 or pathA[0][0] != src or pathA[-1][1] != dest:
        return ["walk does not run src -> dest"]
    while it < n:
        u, v = pathA[it]
        if u == src or (v != dest and int(g.v_i[v]) == int(g.v_j[v])):
            cont = src if u == src else mine[-1][1]
            y = int(g.v_j[v])
            if u == src and not (int(g.v_i[v]) == y and v != dest):
                return ["walk starts with src -> %d, not a record vertex" % v]
            nu, nv = pathA[it + 1]
            if nu != v:
                return ["walk not chained at edge %d" % it]
            if nv == dest or int(g.v_i[nv]) == int(g.v_j[nv]):
                alt(cont, nv, y, True, [(u, v)])
            else:
                alt(cont, nv, None, False, [(u, v), (nu, nv)])
                it += 1
        elif v == dest:
            cont = mine[-1][1]
            alt(cont, v, None, False, [])
        else:                                                         # v = (x, y), x != y: the edge stays (:866-873)
            mine.append((u, v))
        it += 1
    if mine != list(pathB):
        k = next((i for i, (x, y) in enumerate(zip(mine, pathB)) if x != y), min(len(mine), len(pathB)))
        bad.append("upgraded path differs from this reading at edge %d: product %r, here %r (lengths %d / %d)" % (
            k, pathB[k] if k < len(pathB) else None, mine[k] if k < len(mine) else None, len(pathB), len(mine)))
    return bad


def chain_invariants_batch(voff, ctgV, rowptr, col, cv_ctg, cv_roff, cv_la, cv_path, rec_off, R0=0):
    """The asserts of :913-918 on EVERY conversion of a batch, vectorised enough for C3 / the C5 share: returns (#conversions
    checked, #edges checked, list of findings)."""
    bad, nedges = [], 0
    for j in range(len(cv_ctg)):
        la = int(cv_la[j])
        if la <= 0:
            bad.append("conversion %d: no walk" % j)
            continue
        c = int(cv_ctg[j])
        N = int(rec_off[c + 1] - rec_off[c])
        cap = N + 2
        base = 6 * int(cv_roff[j]) + 2 * cap
        pb = cv_path[base: base + 2 * cap].reshape(-1, 2)
        V = int(ctgV[c])
        ends = np.nonzero(pb[:, 1] == V - 1)[0]
        if pb[0, 0] != V - 2 or len(ends) == 0:
            bad.append("conversion %d (contig %d): upgraded path does not run src -> dest" % (j, c))
            continue
        lb = int(ends[0]) + 1
        pb = pb[:lb].astype(np.int64)
        if not (pb[1:, 0] == pb[:-1, 1]).all():
            bad.append("conversion %d (contig %d): upgraded path not chained" % (j, c))
        vb = int(voff[c])
        r0, r1 = rowptr[vb + pb[:, 0]], rowptr[vb + pb[:, 0] + 1]
        for t in range(lb):                                           # rows are short on sparse graphs; dense rows: one vectorised test per edge
            if not (col[int(r0[t]):int(r1[t])] == pb[t, 1]).any():
                bad.append("conversion %d (contig %d): %d -> %d is not an edge" % (j, c, pb[t, 0], pb[t, 1]))
                break
        nedges += lb
    return len(cv_ctg), nedges, bad


def collect(fetch, rec_off):
    """The arrays this checker reads, through `fetch(name, dtype)` (DeviceResult.debug of a keep_debug solve, or the emulation's
    fetch).  Returns a dict; `graph_of(arr, c)` / `conversions_of(arr, c)` cut one contig out of it."""
    a = {"rec_off": np.asarray(rec_off, np.int64)}
    C = len(a["rec_off"]) - 1
    a["voff"] = fetch("voff", np.int64)[:C + 1]
    a["ctgV"] = fetch("ctgV", np.int32)[:C]
    VT = int(a["voff"][C])
    a["rowptr"] = fetch("csr_rowptr", np.int64)[:VT + 1]
    ET = int(a["rowptr"][VT]) if VT else 0
    a["col"] = fetch("csr_col", np.int32)[:ET]
    a["wq"] = fetch("csr_w_qry", np.int64)[:ET]
    a["wr"] = fetch("csr_w_ref", np.int32)[:ET]
    a["fl"] = fetch("csr_w_flags", np.uint8)[:ET]
    a["v_i"] = fetch("v_i", np.int32)[:VT]
    a["v_j"] = fetch("v_j", np.int32)[:VT]
    a["fwd_order"] = fetch("fwd_order", np.int32)[:VT]
    a["conv_off"] = fetch("conv_off", np.int64)[:C + 1]
    NCONV = int(a["conv_off"][C])
    a["cv_ctg"] = fetch("cv_ctg", np.int32)[:NCONV]
    a["cv_roff"] = fetch("cv_roff", np.int64)[:NCONV + 1]
    a["cv_la"] = fetch("cv_la", np.int32)[:NCONV]
    a["cv_path"] = fetch("cv_path", np.int32)
    return a


def graph_of(a, c):
    vb, V = int(a["voff"][c]), int(a["ctgV"][c])
    rp = a["rowptr"][vb:vb + V + 1]
    e0, e1 = int(rp[0]), int(rp[-1])
    return Graph(rp - e0, a["col"][e0:e1], a["wq"][e0:e1], a["wr"][e0:e1], a["fl"][e0:e1], a["v_i"][vb:vb + V], a["v_j"][vb:vb + V],
                 a["fwd_order"][vb:vb + V].astype(np.int64))


def conversions_of(a, c):
    """[(pathA, pathB)] of contig c, in conversion order; paths as lists of (u, v)."""
    out = []
    N = int(a["rec_off"][c + 1] - a["rec_off"][c])
    cap = N + 2
    V = int(a["ctgV"][c])
    for j in range(int(a["conv_off"][c]), int(a["conv_off"][c + 1])):
        la = int(a["cv_la"][j])
        base = 6 * int(a["cv_roff"][j])
        pa = a["cv_path"][base: base + 2 * la].reshape(-1, 2)
        pb = a["cv_path"][base + 2 * cap: base + 4 * cap].reshape(-1, 2)
        ends = np.nonzero(pb[:, 1] == V - 1)[0]
        lb = int(ends[0]) + 1 if len(ends) else 0
        out.append(([(int(u), int(v)) for u, v in pa], [(int(u), int(v)) for u, v in pb[:lb]]))
    return out
