#!/usr/bin/env python3
"""What a wave pays for the box walk of the marking variant on the Cornell box, and what other trees over the same
per-octant leaves would cost.  A wave walks until its slowest lane is done, so the cost of a query is the MAXIMUM of
64 rays' step counts, not their mean.  CPU only (rays captured from the CPU restatement).

  python tools/octant_tree_probe.py [search]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cudapathtracer_amd as hp
from orclib import Oracle, RNG_PHILOX, TRIG_PORTABLE

f32 = np.float32
sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, 20, 1024, 1024)
nodes, table = sc.octant_copies(True)
words = nodes.view(np.uint32)
nn = nodes.shape[1]


def current_tree(k):
    """(leaves [(lo, hi)], tree as nested lists of leaf indices) of octant copy k."""
    n = int(table[k])
    leaves, stack, top = [], [], []
    for i in range(n):
        while stack and stack[-1][0] == i:
            stack.pop()
        near, far = nodes[k, i, 0:3], nodes[k, i, 4:7]
        lo, hi = np.minimum(near, far), np.maximum(near, far)
        miss = int(words[k, i, 3]) // 32 - k * nn
        here = stack[-1][1] if stack else top
        if words[k, i, 7]:
            MASKS.setdefault(k, []).append(int(words[k, i, 7]))
            leaves.append((lo, hi))
            here.append(len(leaves) - 1)
        else:
            sub = []
            here.append(sub)
            stack.append((miss, sub))
    return leaves, top


MASKS = {}


def spec(trees):
    """The trees in the form the A/B build of the library reads from HPT_OCTANT_TREE_SPEC (leaves as hex masks)."""
    def one(k, t):
        return "%x" % MASKS[k][t] if isinstance(t, int) else "[" + ",".join(one(k, c) for c in t) + "]"
    return ";".join(one(k, trees[k]) for k in range(8))


def capture_rays():
    osc = Oracle().cornell(1024, 1024)
    rs = np.random.RandomState(1)
    pix = rs.choice(1024 * 1024, 40000, replace=False).astype(np.int32)
    O, D = [], []
    for b in range(0, 12):
        o, d = osc.bounce_rays(RNG_PHILOX, TRIG_PORTABLE, b, np.zeros_like(pix), pix)
        O.append(o); D.append(d)
    o, d = np.concatenate(O), np.concatenate(D)
    prim, t = osc.intersect(o, d)
    hit = prim >= 0
    p = (o[hit] + d[hit] * t[hit, None].astype(f32)).astype(f32)
    x = (rs.rand(len(p), 3) * [130, 0, 105] + [213, 548.7, 227]).astype(f32)
    v = x - p
    ln = np.linalg.norm(v, axis=1)
    ok = ln > 1e-3
    sd = (v[ok] / ln[ok, None]).astype(f32)
    so = (p[ok] + sd * f32(1e-3)).astype(f32)
    return (o, d), (so, sd)


def octant_of(d):
    return (d[:, 0] < 0).astype(int) | (d[:, 1] < 0).astype(int) << 1 | (d[:, 2] < 0).astype(int) << 2


def slab(lo, hi, o, inv):
    a, b = (lo - o) * inv, (hi - o) * inv
    return np.minimum(a, b).max(axis=1) <= np.maximum(a, b).min(axis=1)


class Pool:
    """Rays of one kind, split by octant, with the hit flags of every leaf box and a cache for unions."""
    def __init__(self, o, d, leaves_by_octant):
        keep = np.all(d != 0, axis=1)
        o, d = o[keep], d[keep]
        self.k = octant_of(d)
        self.n = len(o)
        self.idx = [np.flatnonzero(self.k == k) for k in range(8)]
        self.o = [o[i] for i in self.idx]
        self.inv = [(f32(1) / d[i]).astype(f32) for i in self.idx]
        self.leaves = leaves_by_octant
        self.cache = [dict() for _ in range(8)]
        rs = np.random.RandomState(7)
        self.groups = rs.randint(0, self.n, (3000, 64))

    def hit(self, k, leafset):
        key = tuple(sorted(leafset))
        c = self.cache[k]
        if key not in c:
            lo = np.min([self.leaves[k][i][0] for i in key], axis=0)
            hi = np.max([self.leaves[k][i][1] for i in key], axis=0)
            c[key] = slab(lo, hi, self.o[k], self.inv[k])
        return c[key]

    def steps(self, k, tree):
        """records tested per ray of octant k under `tree` (nested lists; ints are leaves)."""
        total = np.zeros(len(self.idx[k]), np.int32)

        def leafset(t):
            return [t] if isinstance(t, int) else [x for s in t for x in leafset(s)]

        def walk(children, reached):
            for c in children:
                total[reached] += 1
                if not isinstance(c, int):
                    walk(c, reached & self.hit(k, leafset(c)))
        walk(tree, np.ones(len(self.idx[k]), bool))
        return total

    def all_steps(self, trees):
        s = np.zeros(self.n, np.int32)
        for k in range(8):
            s[self.idx[k]] = self.steps(k, trees[k])
        return s

    def wave_cost(self, s):
        return s[self.groups].max(axis=1).mean()


def flat(tree):
    return [tree] if isinstance(tree, int) else [x for s in tree for x in flat(s)]


def show(t):
    return str(t).replace(" ", "")


if __name__ == "__main__":
    cur = [current_tree(k) for k in range(8)]
    leaves = [c[0] for c in cur]
    trees = [c[1] for c in cur]
    (o, d), (so, sd) = capture_rays()
    closest, shadow = Pool(o, d, leaves), Pool(so, sd, leaves)
    print("rays: %d closest, %d shadow; per octant %s / %s" % (closest.n, shadow.n, [len(i) for i in closest.idx], [len(i) for i in shadow.idx]))

    def report(name, trees):
        sc_, ss_ = closest.all_steps(trees), shadow.all_steps(trees)
        print("%-28s closest mean %.2f wave %.2f | shadow mean %.2f wave %.2f | wave sum %.2f" % (
            name, sc_.mean(), closest.wave_cost(sc_), ss_.mean(), shadow.wave_cost(ss_), closest.wave_cost(sc_) + shadow.wave_cost(ss_)))
        return closest.wave_cost(sc_) + shadow.wave_cost(ss_)
    report("current trees", trees)
    report("flat lists", [flat(t) for t in trees])
    if len(sys.argv) > 1 and sys.argv[1] == "search":
        import copy, random
        random.seed(3)

        which = sys.argv[2] if len(sys.argv) > 2 else "both"

        def objective(trees, k=None, base=None):
            v = 0.0
            if which in ("both", "closest"): v += closest.wave_cost(closest.all_steps(trees))
            if which in ("both", "shadow"): v += shadow.wave_cost(shadow.all_steps(trees))
            return v

        def moves(tree):
            """neighbouring trees: dissolve an interior node, group two siblings, move a child into a sibling group."""
            out = []

            def paths(t, p):
                yield p, t
                for i, c in enumerate(t):
                    if not isinstance(c, int):
                        yield from paths(c, p + [i])
            for p, lst in paths(tree, []):
                for i, c in enumerate(lst):
                    if not isinstance(c, int):            # dissolve c
                        nt = copy.deepcopy(tree); l = nt
                        for j in p: l = l[j]
                        l[i:i + 1] = copy.deepcopy(c)
                        out.append(nt)
                if len(lst) > 2:
                    for i in range(len(lst)):
                        for j in range(i + 1, len(lst)):  # group i and j
                            nt = copy.deepcopy(tree); l = nt
                            for q in p: l = l[q]
                            a, b = l[i], l[j]
                            del l[j]; del l[i]
                            l.append([a, b])
                            out.append(nt)
                for i in range(len(lst)):
                    for j in range(len(lst)):             # move child i into sibling group j
                        if i != j and not isinstance(lst[j], int) and len(lst) > 2:
                            nt = copy.deepcopy(tree); l = nt
                            for q in p: l = l[q]
                            a = l[i]
                            l[j].append(a)
                            del l[i]
                            out.append(nt)
            return out
        best = copy.deepcopy(trees)
        best_cost = objective(best)
        improved = True
        while improved:
            improved = False
            for k in range(8):
                cand = moves(best[k])
                random.shuffle(cand)
                for c in cand:
                    t = best[:k] + [c] + best[k + 1:]
                    cost = objective(t)
                    if cost < best_cost - 1e-9:
                        best, best_cost, improved = t, cost, True
                        print("octant %d -> %.3f %s" % (k, cost, show(c)), flush=True)
                        break
        report("searched trees", best)
        print("SPEC", spec(best))
        for k in range(8):
            print(k, show(trees[k]), "->", show(best[k]))


def heuristic_tree(lv, alpha, scene_area):
    """The upload's rule (smallest-union-area agglomeration, then interior nodes that do not pay are dissolved) with
    the pay-off threshold scaled by alpha (1 = what build_octant_copies does)."""
    def area(lo, hi):
        d = np.maximum(0.0, hi.astype(float) - lo.astype(float))
        return 2.0 * (d[0] * d[1] + d[1] * d[2] + d[0] * d[2])
    pool = [dict(lo=l[0], hi=l[1], kids=None, leaf=i) for i, l in enumerate(lv)]
    roots = list(range(len(pool)))
    while len(roots) > 1:
        best = None
        for x in range(len(roots)):
            for y in range(x + 1, len(roots)):
                lo = np.minimum(pool[roots[x]]["lo"], pool[roots[y]]["lo"]); hi = np.maximum(pool[roots[x]]["hi"], pool[roots[y]]["hi"])
                a = area(lo, hi)
                if best is None or a < best[0]:
                    best = (a, x, y, lo, hi)
        _, x, y, lo, hi = best
        pool.append(dict(lo=lo, hi=hi, kids=[roots[x], roots[y]], leaf=-1))
        del roots[y]; del roots[x]
        roots.append(len(pool) - 1)

    def keep(n, q_parent, out):
        if pool[n]["leaf"] >= 0:
            out.append(pool[n]["leaf"]); return
        q = min(1.0, area(pool[n]["lo"], pool[n]["hi"]) / scene_area)
        kids = []
        for c in pool[n]["kids"]:
            keep(c, q, kids)
        if q > alpha * q_parent * (1.0 - 1.0 / len(kids)):
            out.extend(kids)
        else:
            out.append(kids)
    top = []
    keep(roots[0], 1.0, top)
    return top


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "alpha":
    root_lo, root_hi = nodes[8, 0, 0:3], nodes[8, 0, 4:7]
    dd = np.maximum(0.0, np.maximum(root_lo, root_hi).astype(float) - np.minimum(root_lo, root_hi).astype(float))
    scene_area = 2.0 * (dd[0] * dd[1] + dd[1] * dd[2] + dd[0] * dd[2])
    for alpha in (1.0, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3):
        ts = [heuristic_tree(leaves[k], alpha, scene_area) for k in range(8)]
        report("alpha %.1f" % alpha, ts)
        if alpha in (1.0, 0.6):
            for k in range(8): print("   ", k, show(ts[k]))


def synthetic_rays(n, seed):
    """What an upload could draw without tracing anything: origins uniform over the triangles' area, directions
    cosine-distributed about the normal; shadow rays from the same origins to uniform points of the emitting triangles."""
    fl = sc.flat_arrays()
    tris, v12 = fl["tris"], fl["tri_v12"]
    rs = np.random.RandomState(seed)
    area = tris[:, 3].astype(float)
    # the area slot holds a pdf on the device tables only; the host array holds the triangle's area
    v0, e1, e2, nrm = tris[:, 0:3], tris[:, 4:7], tris[:, 8:11], tris[:, 12:15]
    a = 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1)
    t = rs.choice(len(tris), n, p=a / a.sum())
    u, v = rs.rand(n), rs.rand(n)
    su = np.sqrt(u)
    p = v0[t] + e1[t] * ((1 - su) * 0 + su * (1 - v))[:, None] + e2[t] * (su * v)[:, None]
    N = nrm[t]
    r1, r2 = rs.rand(n), rs.rand(n)
    z = np.sqrt(1 - r1); phi = 2 * np.pi * r2
    x, y = np.sqrt(r1) * np.cos(phi), np.sqrt(r1) * np.sin(phi)
    h = np.where(np.abs(N[:, [0]]) > 0.9, [[0.0, 1.0, 0.0]], [[1.0, 0.0, 0.0]])
    T = np.cross(h, N); T /= np.linalg.norm(T, axis=1, keepdims=True)
    B = np.cross(N, T)
    d = (T * x[:, None] + B * y[:, None] + N * z[:, None]).astype(f32)
    o = (p + N * 1e-3).astype(f32)
    mats = fl["tris"][:, 7].view(np.int32)
    em = np.flatnonzero(np.any(fl["materials"].reshape(len(fl["materials"]), -1)[mats][:, 0:3] > 0, axis=1)) if False else np.array([i for i in range(len(tris)) if abs(v0[i, 1] - 548.7) < 0.2 or abs(v0[i, 1] - 548.8) < 0.05 and False])
    lt = rs.choice(em, n)
    u, v = rs.rand(n), rs.rand(n); su = np.sqrt(u)
    x_ = v0[lt] + e1[lt] * (su * (1 - v))[:, None] + e2[lt] * (su * v)[:, None]
    sd = x_ - p; sd /= np.linalg.norm(sd, axis=1, keepdims=True)
    ok = (sd * N).sum(axis=1) > 0
    return (o, d), (o[ok], sd[ok].astype(f32))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "greedy":
    import copy
    (o2, d2), (so2, sd2) = synthetic_rays(int(sys.argv[2]) if len(sys.argv) > 2 else 16384, 5)
    c2, s2 = Pool(o2, d2, leaves), Pool(so2, sd2, leaves)
    print("synthetic: %d closest %d shadow" % (c2.n, s2.n), [len(i) for i in c2.idx], [len(i) for i in s2.idx])

    def obj(ts):
        return c2.wave_cost(c2.all_steps(ts)) + s2.wave_cost(s2.all_steps(ts))

    def dissolves(tree):
        out = []

        def paths(t, p):
            yield p, t
            for i, c in enumerate(t):
                if not isinstance(c, int):
                    yield from paths(c, p + [i])
        for p, lst in paths(tree, []):
            for i, c in enumerate(lst):
                if not isinstance(c, int):
                    nt = copy.deepcopy(tree); l = nt
                    for j in p: l = l[j]
                    l[i:i + 1] = copy.deepcopy(c)
                    out.append(nt)
        return out
    best = copy.deepcopy(trees); cost = obj(best)
    print("synthetic objective of the current trees %.3f" % cost)
    improved = True
    while improved:
        improved = False
        for k in range(8):
            for c in dissolves(best[k]):
                t = best[:k] + [c] + best[k + 1:]
                v = obj(t)
                if v < cost - 1e-9:
                    best, cost, improved = t, v, True
                    break
    print("synthetic objective after greedy dissolves %.3f" % cost)
    report("greedy dissolves (real rays)", best)
    print("SPEC", spec(best))
    for k in range(8): print("   ", k, show(best[k]))


def emax(steps_list, weights=None):
    """E[max of 64 draws] from the pooled distribution of the step counts."""
    s = np.concatenate(steps_list)
    h = np.bincount(s, minlength=80).astype(float)
    F = np.cumsum(h) / h.sum()
    return float(np.sum(1.0 - F ** 64))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "cdf":
    import copy
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    (o2, d2), (so2, sd2) = synthetic_rays(N, 5)
    c2, s2 = Pool(o2, d2, leaves), Pool(so2, sd2, leaves)
    print("synthetic: %d closest %d shadow" % (c2.n, s2.n))
    lam = 0.05
    evals = [0]

    def per_octant(ts, k):
        return c2.steps(k, ts[k]), s2.steps(k, ts[k])

    def obj_from(parts):
        c = [p[0] for p in parts]; s = [p[1] for p in parts]
        return emax(c) + emax(s) + lam * (np.concatenate(c).mean() + np.concatenate(s).mean())
    best = copy.deepcopy(trees)
    parts = [per_octant(best, k) for k in range(8)]
    cost = obj_from(parts)
    print("objective of the current trees %.3f (real-ray check of emax: %.2f + %.2f)" % (
        cost, emax([closest.all_steps(trees)]), emax([shadow.all_steps(trees)])))

    def all_moves(tree):
        out = []

        def paths(t, p):
            yield p, t
            for i, c in enumerate(t):
                if not isinstance(c, int):
                    yield from paths(c, p + [i])
        for p, lst in paths(tree, []):
            for i, c in enumerate(lst):
                if not isinstance(c, int):
                    nt = copy.deepcopy(tree); l = nt
                    for j in p: l = l[j]
                    l[i:i + 1] = copy.deepcopy(c)
                    out.append(nt)
            if len(lst) > 2:
                for i in range(len(lst)):
                    for j in range(i + 1, len(lst)):
                        nt = copy.deepcopy(tree); l = nt
                        for q in p: l = l[q]
                        a, b = l[i], l[j]
                        del l[j]; del l[i]
                        l.append([a, b])
                        out.append(nt)
                for i in range(len(lst)):
                    for j in range(len(lst)):
                        if i != j and not isinstance(lst[j], int):
                            nt = copy.deepcopy(tree); l = nt
                            for q in p: l = l[q]
                            a = l[i]
                            l[j].append(a)
                            del l[i]
                            out.append(nt)
        return out
    improved = True
    while improved:
        improved = False
        for k in range(8):
            again = True
            while again:
                again = False
                for c in all_moves(best[k]):
                    t = best[:k] + [c] + best[k + 1:]
                    np_ = parts[:k] + [per_octant(t, k)] + parts[k + 1:]
                    evals[0] += 1
                    v = obj_from(np_)
                    if v < cost - 1e-3:
                        best, cost, parts, improved, again = t, v, np_, True, True
                        break
    print("objective after the search %.3f, %d evaluations" % (cost, evals[0]))
    report("cdf search N=%d (real rays)" % N, best)
    print("SPEC", spec(best))
