"""Round-4 experiment (VERDICT r03 #2 ii): other ordering FAMILIES for the meshed HV grids of the wave-team kernels.

The team kernels are level-bound: a Newton iteration walks one round per wavefront and elimination level (+ a barrier),
and the level-scheduled minimum-degree plan of the 306- / 372-bus grids has 25 / 30 levels of which 15-19 are the dense
tail (one pivot per level).  This script evaluates, on the CPU and on the plan level only (no kernel needed: rounds per
wavefront follow from the item counts per level), what other orderings would give:

  * the elimination-tree height of the CURRENT order (is the level schedule tight?),
  * sequential minimum degree / minimum fill scheduled by elimination-tree depth,
  * nested dissection (spectral bisection + vertex separator from a vertex cover of the cut, leaves by minimum degree),
    on the whole graph and on the core that remains after stripping the degree <= 2 chains,
  * the level-scheduled multiple-elimination with FILL instead of degree as the key (several slacks, 8 tie seeds).

    python scripts/ordering_experiments.py > profiles/r04_ordering_experiments.txt

Result (profiles/r04_ordering_experiments.txt): nothing beats the searched minimum-degree plan; nested dissection is
35-41 levels with twice the fill.  The graphs' cores are not planar-like (random meshed stand-ins: the final clique of
the minimum-degree order has ~22 vertices), so separators are large.  Negative result, not built into plan.cpp.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opfgym_amd import capi, grids  # noqa: E402
from opfgym_amd.case import net_to_case  # noqa: E402
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)


def graph(code):
    case = net_to_case(grids.get_grid(code)[0])
    plan = capi.Plan(case)
    yp, yc = plan.array('Y_PTR'), plan.array('Y_COL')
    bt = np.asarray(case.bus_type)
    adj = {i: set() for i in range(case.nb) if bt[i] != 3}
    for i in adj:
        for e in range(yp[i], yp[i + 1]):
            j = int(yc[e])
            if j != i and bt[j] != 3:
                adj[i].add(j)
                adj[j].add(i)
    return plan, adj


def etree_stats(adj, order):
    """Sequential symbolic elimination in `order`, scheduled by elimination-tree depth (maximal parallelism): levels,
    fill blocks, update terms (d^2 + d per pivot) per level."""
    adj = {k: set(v) for k, v in adj.items()}
    items, fill = {}, 0
    depth = {v: 0 for v in order}
    for k in order:
        nb_, lk = adj[k], depth[k]
        d = len(nb_)
        items[lk] = items.get(lk, 0) + d * d + d
        for i in nb_:
            depth[i] = max(depth[i], lk + 1)
            for j in nb_:
                if i < j and j not in adj[i]:
                    adj[i].add(j)
                    adj[j].add(i)
                    fill += 2
        for i in nb_:
            adj[i].discard(k)
        adj[k] = set()
    nlev = max(items) + 1
    return nlev, fill, [items.get(lv, 0) for lv in range(nlev)]


def rounds(items, nw=4):
    return sum(-(-(-(-n // 64)) // nw) for n in items if n > 0)


def fill_count(adj, v):
    nb = list(adj[v])
    return sum(1 for a in range(len(nb)) for b in range(a + 1, len(nb)) if nb[b] not in adj[nb[a]])


def greedy(adj, key='degree'):
    adj = {k: set(v) for k, v in adj.items()}
    verts, order = set(adj), []
    while verts:
        k = min(verts, key=(lambda v: (len(adj[v]), v)) if key == 'degree' else (lambda v: (fill_count(adj, v), len(adj[v]), v)))
        for i in adj[k]:
            adj[i] |= adj[k] - {i}
            adj[i].discard(k)
        adj[k] = set()
        verts.discard(k)
        order.append(k)
    return order


def components(adj, verts):
    verts, comps = set(verts), []
    while verts:
        s = verts.pop()
        comp, st = {s}, [s]
        while st:
            for w in adj[st.pop()]:
                if w in verts:
                    verts.discard(w)
                    comp.add(w)
                    st.append(w)
        comps.append(comp)
    return comps


def nested_dissection(adj, verts, leaf=12):
    verts = set(verts)
    sub = lambda vs: {v: adj[v] & vs for v in vs}
    if len(verts) <= leaf:
        return greedy(sub(verts))
    comps = components(adj, verts)
    if len(comps) > 1:
        return [v for c in comps for v in nested_dissection(adj, c, leaf)]
    vs = sorted(verts)
    idx = {v: i for i, v in enumerate(vs)}
    lap = np.zeros((len(vs), len(vs)))
    for v in vs:
        for w in adj[v]:
            if w in idx:
                lap[idx[v], idx[w]] -= 1
                lap[idx[v], idx[v]] += 1
    f = np.linalg.eigh(lap)[1][:, 1]                       # Fiedler vector
    a = {vs[i] for i in range(len(vs)) if f[i] <= np.median(f)}
    b = verts - a
    cut = [(x, y) for x in a for y in adj[x] if y in b]
    sep, left = set(), set(cut)
    while left:                                            # vertex cover of the cut edges, greedy, then pruned
        deg = {}
        for x, y in left:
            deg[x] = deg.get(x, 0) + 1
            deg[y] = deg.get(y, 0) + 1
        v = max(deg, key=lambda x: (deg[x], -x))
        sep.add(v)
        left = {e for e in left if v not in e}
    for v in sorted(sep):
        if all((x in sep - {v} or y in sep - {v}) for x, y in cut):
            sep.discard(v)
    a -= sep
    b -= sep
    if not a or not b:
        return greedy(sub(verts))
    return nested_dissection(adj, a, leaf) + nested_dissection(adj, b, leaf) + sorted(sep)


def strip_then(adj, inner):
    adj2 = {k: set(v) for k, v in adj.items()}
    verts, order = set(adj2), []
    while True:
        low = [v for v in verts if len(adj2[v]) <= 2]
        if not low:
            break
        k = min(low, key=lambda v: (len(adj2[v]), v))
        for i in adj2[k]:
            adj2[i] |= adj2[k] - {i}
            adj2[i].discard(k)
        adj2[k] = set()
        verts.discard(k)
        order.append(k)
    return order + inner({v: adj2[v] & verts for v in verts})


def level_scheduled(adj0, metric, slack, fslack, seed):
    """plan.cpp's multiple elimination with independent sets as levels; metric 'fill': candidates by the fill their
    elimination creates (level takes every independent vertex within `fslack` of the minimum) instead of by degree."""
    adj = {k: set(v) for k, v in adj0.items()}
    alive, levels = set(adj), []

    def h(v):
        x = (v * 2654435761 + seed * 40503) & 0xffffffff
        x ^= x >> 15
        x = (x * 2246822519) & 0xffffffff
        return x ^ (x >> 13)
    while alive:
        cand = sorted(alive)
        key = {v: len(adj[v]) for v in cand} if metric == 'degree' else {v: fill_count(adj, v) for v in cand}
        cap = max(2, min(key.values()) + slack) if metric == 'degree' else min(key.values()) + fslack
        dmin = min(len(adj[v]) for v in alive)
        cand.sort(key=lambda v: (key[v], len(adj[v]), h(v) if seed else v))
        blocked, piv = set(), []
        for k in cand:
            if key[k] > cap:
                break
            if metric != 'degree' and len(adj[k]) > max(2, dmin + slack + 10):
                continue
            if k not in blocked:
                piv.append(k)
                blocked |= adj[k] | {k}
        items = sum(len(adj[k]) ** 2 + len(adj[k]) for k in piv)
        uterms = sum(len(adj[k]) for k in piv)
        new = [(i, j) for k in piv for i in adj[k] for j in adj[k] if i < j and j not in adj[i]]
        for i, j in new:
            adj[i].add(j)
            adj[j].add(i)
        for k in piv:
            for j in adj[k]:
                adj[j].discard(k)
            adj[k] = set()
            alive.discard(k)
        levels.append((len(piv), items, uterms))
    m = 0
    for p, _, _ in reversed(levels):
        if p != 1 or m >= 32:
            break
        m += 1
    m = m if m >= 4 else 0
    rb = lambda n: -(-(-(-n // 64)) // 4) if n > 0 else 0
    b = sum(rb(it) for _, it, _ in levels)
    c = sum(rb(u) for _, _, u in (levels[:-m] if m else levels)) + (1 if m else 0)
    return len(levels), b, c, m, sum(it for _, it, _ in levels)


if __name__ == '__main__':
    for code in ('1-HV-mixed--0-sw', '1-HV-urban--0-sw'):
        plan, adj = graph(code)
        info = plan.info
        print(f'== {code}: {len(adj)} non-slack buses, {sum(len(v) for v in adj.values()) // 2} edges; the plan in use: {info["n_levels"]} levels, '
              f'{info["n_fill"]} fill blocks, {info["team_rounds_4"]} rounds / {info["team_barriers_4"]} barriers per wavefront of a team of four, dense tail {info["tail_m"]}')
        order = [int(b) for b in plan.array('PIV_BUS')]
        nlev, fill, items = etree_stats(adj, order)
        print(f'   elimination-tree height of that order: {nlev} (= its level count: the schedule is tight); update terms per level: {items}')
        print('   orderings scheduled by elimination-tree depth (levels | fill blocks | update terms | factorisation rounds per wavefront of four):')
        cands = {'minimum degree, sequential': greedy(adj), 'minimum fill, sequential': greedy(adj, 'fill'),
                 'nested dissection, leaves <= 12': nested_dissection(adj, set(adj), 12),
                 'nested dissection, leaves <= 6': nested_dissection(adj, set(adj), 6),
                 'chains stripped, then nested dissection (8)': strip_then(adj, lambda core: nested_dissection(core, set(core), 8)),
                 'chains stripped, then minimum fill': strip_then(adj, lambda core: greedy(core, 'fill'))}
        for name, o in cands.items():
            nlev, fill, items = etree_stats(adj, o)
            print(f'     {name:46s} {nlev:3d} | {fill:5d} | {sum(items):6d} | {rounds(items):3d}')
        print('   level-scheduled multiple elimination, best of 8 tie seeds (levels, B rounds, C rounds, tail, update terms):')
        for metric, slack, fslack in (('degree', 2, 0), ('degree', 3, 0), ('fill', 2, 0), ('fill', 2, 1), ('fill', 2, 2), ('fill', 2, 4)):
            res = sorted((level_scheduled(adj, metric, slack, fslack, seed) for seed in range(8)), key=lambda r: r[1] + r[2])
            print(f'     key {metric:6s} slack {slack} fill-slack {fslack}: best {res[0]}  (B + C = {res[0][1] + res[0][2]}), worst B + C = {res[-1][1] + res[-1][2]}')
