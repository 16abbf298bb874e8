"""Grid sources: textbook cases and a synthetic SimBench-like generator.

SimBench (third party; `sb.get_simbench_net`, build_simbench_net.py:11) and its
time-series data are not available in this environment, so the BASELINE
configurations run on synthetic grids with the element counts SURVEY.md §8
lists for the named SimBench codes, and synthetic 35 136-step profiles in
SimBench's factored form (a few relative profile types × per-unit peak).
Every generator is deterministic in its `seed`.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from . import net as ppn
from .net import Net

N_STEPS = 24 * 4 * 366          # data_split.py:13 hard-codes this horizon


# --------------------------------------------------------------------------
# textbook cases
# --------------------------------------------------------------------------
def two_bus(p_mw=0.8, q_mvar=0.3, r_ohm=0.5, x_ohm=0.8, vn_kv=10.0, vm_slack=1.0) -> Net:
    """One slack, one line without charging, one PQ load: closed-form solvable
    (|V2|² is the root of a quadratic) — KAT for the NR restatement."""
    net = Net('two_bus', sn_mva=1.0)
    b0 = ppn.create_bus(net, vn_kv, min_vm_pu=0.95, max_vm_pu=1.05)
    b1 = ppn.create_bus(net, vn_kv, min_vm_pu=0.95, max_vm_pu=1.05)
    ppn.create_ext_grid(net, b0, vm_pu=vm_slack)
    ppn.create_line_from_parameters(net, b0, b1, 1.0, r_ohm, x_ohm, 0.0, 0.3,
                                    max_loading_percent=100.0)
    ppn.create_load(net, b1, p_mw, q_mvar)
    return ppn.finalize(net)


def two_bus_closed_form(p_mw, q_mvar, r_ohm, x_ohm, vn_kv, vm_slack=1.0):
    """|V2| in p.u. for `two_bus`: with S=P+jQ drawn at bus 2 over Z=R+jX from
    a stiff source E:  V2⁴ + (2(PR+QX) − E²)·V2² + (P²+Q²)(R²+X²) = 0."""
    zb = vn_kv ** 2 / 1.0
    r, x = r_ohm / zb, x_ohm / zb
    a = 2 * (p_mw * r + q_mvar * x) - vm_slack ** 2
    c = (p_mw ** 2 + q_mvar ** 2) * (r ** 2 + x ** 2)
    v2 = (-a + np.sqrt(a * a - 4 * c)) / 2
    return np.sqrt(v2)


_CASE9_BRANCH = [  # from, to, r, x, b  (p.u. on 100 MVA) — WSCC 3-machine 9-bus
    (1, 4, 0.0, 0.0576, 0.0), (4, 5, 0.017, 0.092, 0.158),
    (5, 6, 0.039, 0.17, 0.358), (3, 6, 0.0, 0.0586, 0.0),
    (6, 7, 0.0119, 0.1008, 0.209), (7, 8, 0.0085, 0.072, 0.149),
    (8, 2, 0.0, 0.0625, 0.0), (8, 9, 0.032, 0.161, 0.306),
    (9, 4, 0.01, 0.085, 0.176)]


def case9() -> Net:
    """WSCC 3-machine 9-bus system (Anderson & Fouad): loads 90+j30 / 100+j35 /
    125+j50 MVA, PV set-points 163 and 85 MW at 1.025 p.u., slack 1.04 p.u.
    Public textbook data typed in by hand (bus numbering 1-3 generators, 4-9
    network); the three step-up transformers are plain reactances."""
    vn, base, f_hz = 345.0, 100.0, 50.0
    net = Net('case9', f_hz=f_hz, sn_mva=base)
    for _ in range(9):
        ppn.create_bus(net, vn, min_vm_pu=0.9, max_vm_pu=1.1)
    zb = vn ** 2 / base
    for fb, tb, r, x, b in _CASE9_BRANCH:
        c_nf = b / (2 * np.pi * f_hz * zb) * 1e9
        ppn.create_line_from_parameters(net, fb - 1, tb - 1, 1.0, r * zb, x * zb,
                                        c_nf, 1.0, max_loading_percent=100.0)
    ppn.create_ext_grid(net, 0, vm_pu=1.04)
    ppn.create_gen(net, 1, 163.0, vm_pu=1.025, min_q_mvar=-300.0, max_q_mvar=300.0)
    ppn.create_gen(net, 2, 85.0, vm_pu=1.025, min_q_mvar=-300.0, max_q_mvar=300.0)
    ppn.create_load(net, 4, 90.0, 30.0)
    ppn.create_load(net, 6, 100.0, 35.0)
    ppn.create_load(net, 8, 125.0, 50.0)
    return ppn.finalize(net)


def case9_opf() -> Net:
    """case9 with what a pandapower OPF case carries: generator active-power ranges and quadratic
    generation costs (MATPOWER case9 gencost).  Stand-in for `pp.networks.case_ieee30()` of
    examples/non_simbench_net.py (the IEEE 30-bus data is not available offline)."""
    net = case9()
    net.gen['min_p_mw'] = [10.0, 10.0]
    net.gen['max_p_mw'] = [300.0, 270.0]
    net.gen['min_min_p_mw'] = net.gen['min_p_mw']
    net.gen['max_max_p_mw'] = net.gen['max_p_mw']
    net.gen['controllable'] = True
    net.ext_grid['min_p_mw'] = 10.0
    net.ext_grid['max_p_mw'] = 250.0
    for et, el, c2, c1, c0 in (('ext_grid', 0, 0.11, 5.0, 150.0), ('gen', 0, 0.085, 1.2, 600.0),
                               ('gen', 1, 0.1225, 1.0, 335.0)):
        ppn.create_poly_cost(net, el, et, cp1_eur_per_mw=c1, cp2_eur_per_mw2=c2, cp0_eur=c0)
    return ppn.finalize(net)


# Published load-flow result of the WSCC 3-machine 9-bus system (Anderson &
# Fouad, "Power System Control and Stability", fig. 2.19; the same numbers the
# survey quotes in SURVEY.md §8c), re-indexed to the bus numbering used above:
# magnitudes to 3 decimals, angles to 0.1°, generator outputs to 0.1 MW/MVAr.
# Sanity KAT (tolerances 1e-3 p.u. / 0.06° / 0.06 MVA), not a 1e-6 pin.
CASE9_VM = np.array([1.040, 1.025, 1.025, 1.026, 1.013, 1.032, 1.016, 1.026, 0.996])
CASE9_VA_DEG = np.array([0.0, 9.3, 4.7, -2.2, -3.7, 2.0, 0.7, 3.7, -4.0])
CASE9_SLACK_PQ = (71.6, 27.0)
CASE9_GEN_Q = (6.7, -10.9)


# --------------------------------------------------------------------------
# synthetic SimBench-like grids
# --------------------------------------------------------------------------
def _df(cols: dict, n: int) -> pd.DataFrame:
    return pd.DataFrame({k: (np.full(n, v) if np.isscalar(v) or v is None else v)
                         for k, v in cols.items()}, index=np.arange(n))


def _feeder_tree(rng, roots, n_nodes, first_id, n_feeders, lateral_prob=0.15):
    """Radial feeders hanging off `roots` (busbars): returns edge list and the
    end bus of every feeder (for the open ring ties)."""
    sizes = np.full(n_feeders, n_nodes // n_feeders)
    sizes[: n_nodes - sizes.sum()] += 1
    edges, ends = [], []
    nxt = first_id
    for k, size in enumerate(sizes):
        prev = roots[k % len(roots)]
        chain = []
        for _ in range(size):
            if chain and len(chain) > 2 and rng.random() < lateral_prob:
                parent = chain[rng.integers(1, len(chain) - 1)]   # lateral stub
            else:
                parent = prev
            edges.append((parent, nxt))
            if parent == prev:
                prev = nxt
            chain.append(nxt)
            nxt += 1
        ends.append(prev)
    return edges, ends


def _units_on(rng, candidates, n):
    """Spread n units over candidate buses (first one per bus, then repeats)."""
    cand = np.array(candidates)
    if n <= len(cand):
        return np.sort(rng.choice(cand, n, replace=False))
    extra = rng.choice(cand, n - len(cand), replace=True)
    return np.sort(np.concatenate([cand, extra]))


def _radial_grid(name, seed, *, vn_hv, vn_lv, n_trafos, trafo, n_lv_nodes,
                 n_feeders, line_type, len_range, n_loads, n_sgens, n_storage,
                 load_peak, sgen_peak, big_sgens, ext_vm, n_ties):
    rng = np.random.default_rng(seed)
    nb = 1 + n_trafos + n_lv_nodes
    net = Net(name, f_hz=50.0, sn_mva=1.0)
    vn = np.full(nb, float(vn_lv))
    vn[0] = vn_hv
    net['bus'] = _df(dict(name=None, vn_kv=vn, type='b', in_service=True,
                          min_vm_pu=np.nan, max_vm_pu=np.nan), nb)
    busbars = list(range(1, 1 + n_trafos))
    edges, ends = _feeder_tree(rng, busbars, n_lv_nodes, 1 + n_trafos, n_feeders)
    ties = [(ends[2 * k], ends[2 * k + 1]) for k in range(min(n_ties, len(ends) // 2))]
    fb = np.array([e[0] for e in edges] + [t[0] for t in ties])
    tb = np.array([e[1] for e in edges] + [t[1] for t in ties])
    nl = len(fb)
    r_km, x_km, c_nf, imax = line_type
    net['line'] = _df(dict(
        name=None, from_bus=fb, to_bus=tb,
        length_km=rng.uniform(len_range[0], len_range[1], nl),
        r_ohm_per_km=r_km, x_ohm_per_km=x_km, c_nf_per_km=c_nf, g_us_per_km=0.0,
        max_i_ka=imax, df=1.0, parallel=1,
        in_service=np.arange(nl) < len(edges),          # ring ties are open
        max_loading_percent=np.nan), nl)
    sn, vk, vkr, pfe, i0, shift = trafo
    net['trafo'] = _df(dict(
        name=None, hv_bus=0, lv_bus=np.array(busbars), sn_mva=sn, vn_hv_kv=vn_hv,
        vn_lv_kv=vn_lv, vk_percent=vk, vkr_percent=vkr, pfe_kw=pfe, i0_percent=i0,
        shift_degree=shift, tap_side='hv', tap_neutral=0.0, tap_pos=-1.0,
        tap_step_percent=1.5, parallel=1, df=1.0, in_service=True,
        max_loading_percent=np.nan), n_trafos)
    net['ext_grid'] = _df(dict(name=None, bus=0, vm_pu=ext_vm, va_degree=0.0,
                               in_service=True), 1)
    nodes = list(range(1 + n_trafos, nb))
    lbus = _units_on(rng, nodes, n_loads)
    p_l = rng.uniform(load_peak[0], load_peak[1], n_loads)
    q_l = p_l * np.tan(np.arccos(rng.uniform(0.93, 0.98, n_loads)))
    net['load'] = _df(dict(name=None, bus=lbus, p_mw=p_l, q_mvar=q_l, scaling=1.0,
                           in_service=True, controllable=False), n_loads)
    sbus = _units_on(rng, nodes, n_sgens)
    p_s = rng.uniform(sgen_peak[0], sgen_peak[1], n_sgens)
    nbig = min(big_sgens[0], n_sgens)
    p_s[rng.choice(n_sgens, nbig, replace=False)] = rng.uniform(
        big_sgens[1], big_sgens[2], nbig)
    net['sgen'] = _df(dict(name=None, bus=sbus, p_mw=p_s, q_mvar=0.0, scaling=1.0,
                           in_service=True, controllable=False), n_sgens)
    if n_storage:
        stb = _units_on(rng, nodes, n_storage)
        p_st = rng.uniform(0.1, 0.8, n_storage)
        net['storage'] = _df(dict(name=None, bus=stb, p_mw=p_st,
                                  q_mvar=0.0, scaling=1.0, in_service=True,
                                  controllable=False, min_p_mw=-p_st, max_p_mw=p_st,
                                  min_q_mvar=0.0, max_q_mvar=0.0), n_storage)
    profiles = synthetic_profiles(net, seed + 1)
    return net, profiles


def synthetic_mv_urban(seed: int = 0):
    """Stand-in for SimBench `1-MV-urban--0-sw` (BASELINE configs 2 and 4):
    144 buses (one 110 kV + 143 at 10 kV incl. two busbar sections), two
    110/10 kV 63 MVA transformers with 150° phase shift, 12 radial cable feeders
    with 6 open ring ties, 139 loads, 134 sgens, no storage."""
    return _radial_grid(
        'syn-1-MV-urban--0-sw', seed, vn_hv=110.0, vn_lv=10.0, n_trafos=2,
        trafo=(63.0, 18.0, 0.32, 22.0, 0.04, 150.0), n_lv_nodes=141,
        n_feeders=12, line_type=(0.122, 0.112, 304.0, 0.421),
        len_range=(0.2, 0.8), n_loads=139, n_sgens=134, n_storage=0,
        load_peak=(0.08, 0.45), sgen_peak=(0.03, 0.45), big_sgens=(20, 0.6, 2.5),
        ext_vm=1.025, n_ties=6)


def synthetic_lv_rural1(seed: int = 0):
    """Stand-in for SimBench `1-LV-rural1--0-sw` (BASELINE config 1): 15 buses
    (one 20 kV + 14 at 0.4 kV), one 160 kVA transformer, 13 lines, 13 loads,
    4 sgens."""
    return _radial_grid(
        'syn-1-LV-rural1--0-sw', seed, vn_hv=20.0, vn_lv=0.4, n_trafos=1,
        trafo=(0.16, 4.0, 1.46, 0.46, 0.2, 150.0), n_lv_nodes=13,
        n_feeders=4, line_type=(0.2067, 0.0804, 260.0, 0.27),
        len_range=(0.02, 0.12), n_loads=13, n_sgens=4, n_storage=0,
        load_peak=(0.002, 0.012), sgen_peak=(0.004, 0.02), big_sgens=(0, 0, 0),
        ext_vm=1.025, n_ties=0)


def synthetic_mv_small(seed: int = 0, n_nodes: int = 30, n_storage: int = 2):
    """Small MV grid with storage units for fast tests (not a BASELINE config)."""
    return _radial_grid(
        'syn-mv-small', seed, vn_hv=110.0, vn_lv=20.0, n_trafos=2,
        trafo=(40.0, 16.2, 0.34, 18.0, 0.05, 150.0), n_lv_nodes=n_nodes,
        n_feeders=4, line_type=(0.161, 0.117, 273.0, 0.362),
        len_range=(0.5, 2.5), n_loads=n_nodes - 3, n_sgens=n_nodes // 2,
        n_storage=n_storage, load_peak=(0.2, 0.9), sgen_peak=(0.1, 0.9),
        big_sgens=(4, 1.0, 3.0), ext_vm=1.02, n_ties=2)


def synthetic_hv(seed: int = 0, nb: int = 306, n_ext: int = 2, n_gen: int = 12,
                 name: str = 'syn-1-HV-mixed--0-sw', trafos_per_ext: int = 2,
                 load_share: float = 0.35, sgen_share: float = 0.3):
    """Stand-in for the SimBench HV grids (configs 3 and 5): a meshed 110 kV
    overhead-line network (ring backbone plus chords), `n_ext` 380/110 kV
    infeeds each with its own ext_grid, `n_gen` PV-controlled generators, loads
    and sgens at most stations (306 buses HV-mixed, 372
    HV-urban)."""
    rng = np.random.default_rng(seed)
    net = Net(name, f_hz=50.0, sn_mva=1.0)
    n_ehv = n_ext
    n_hv = nb - n_ehv
    vn = np.full(nb, 110.0)
    vn[:n_ehv] = 380.0
    net['bus'] = _df(dict(name=None, vn_kv=vn, type='b', in_service=True,
                          min_vm_pu=np.nan, max_vm_pu=np.nan), nb)
    hv = np.arange(n_ehv, nb)
    # backbone: stations on a ring of ~n_hv/3 nodes, the rest in short spurs/loops
    n_ring = max(6, n_hv // 3)
    ring = hv[:n_ring]
    edges = [(ring[i], ring[(i + 1) % n_ring]) for i in range(n_ring)]
    for _ in range(n_ring // 4):                                   # chords
        a, b = rng.choice(n_ring, 2, replace=False)
        if abs(a - b) > 1:
            edges.append((ring[a], ring[b]))
    rest = hv[n_ring:]
    attach = {}
    for b in rest:                                                 # spurs
        parent = rng.choice(np.concatenate([ring, np.array(list(attach), dtype=int)])
                            if attach and rng.random() < 0.5 else ring)
        edges.append((int(parent), int(b)))
        attach[int(b)] = int(parent)
    for b in rng.choice(rest, len(rest) // 6, replace=False):      # close loops
        other = int(rng.choice(ring))
        if other != attach[int(b)]:
            edges.append((int(b), other))
    edges = sorted({(min(a, b), max(a, b)) for a, b in edges if a != b})
    nl = len(edges)
    net['line'] = _df(dict(
        name=None, from_bus=np.array([e[0] for e in edges]),
        to_bus=np.array([e[1] for e in edges]),
        length_km=rng.uniform(3.0, 25.0, nl), r_ohm_per_km=0.0949,
        x_ohm_per_km=0.38, c_nf_per_km=9.2, g_us_per_km=0.0, max_i_ka=0.74,
        df=1.0, parallel=1, in_service=True, max_loading_percent=np.nan), nl)
    n_tr = n_ext * trafos_per_ext
    infeed = ring[np.linspace(0, n_ring, n_tr, endpoint=False, dtype=int)]
    net['trafo'] = _df(dict(
        name=None, hv_bus=np.arange(n_tr) % n_ehv, lv_bus=infeed, sn_mva=350.0,
        vn_hv_kv=380.0, vn_lv_kv=110.0, vk_percent=22.0, vkr_percent=0.257,
        pfe_kw=0.0, i0_percent=0.0, shift_degree=0.0, tap_side='hv',
        tap_neutral=0.0, tap_pos=0.0, tap_step_percent=1.25, parallel=1, df=1.0,
        in_service=True, max_loading_percent=np.nan), n_tr)
    net['ext_grid'] = _df(dict(name=None, bus=np.arange(n_ehv), vm_pu=1.0,
                               va_degree=0.0, in_service=True), n_ext)
    n_loads = int(n_hv * load_share)
    lbus = _units_on(rng, list(hv), n_loads)
    p_l = rng.uniform(3.0, 16.0, n_loads)
    net['load'] = _df(dict(name=None, bus=lbus, p_mw=p_l,
                           q_mvar=p_l * np.tan(np.arccos(rng.uniform(0.95, 0.99, n_loads))),
                           scaling=1.0, in_service=True, controllable=False), n_loads)
    n_sg = int(n_hv * sgen_share)
    sbus = _units_on(rng, list(hv), n_sg)
    net['sgen'] = _df(dict(name=None, bus=sbus, p_mw=rng.uniform(3.0, 30.0, n_sg),
                           q_mvar=0.0, scaling=1.0, in_service=True,
                           controllable=False), n_sg)
    gbus = np.sort(rng.choice(hv[~np.isin(hv, infeed)], n_gen, replace=False))
    net['gen'] = _df(dict(name=None, bus=gbus, p_mw=rng.uniform(10.0, 50.0, n_gen),
                          vm_pu=1.0, scaling=1.0, in_service=True, controllable=True,
                          min_q_mvar=np.nan, max_q_mvar=np.nan), n_gen)
    profiles = synthetic_profiles(net, seed + 1)
    return net, profiles


def synthetic_hv_mixed(seed: int = 0):
    return synthetic_hv(seed, nb=306, n_ext=2, n_gen=12, name='syn-1-HV-mixed--0-sw')


def synthetic_hv_urban(seed: int = 0):
    return synthetic_hv(seed, nb=372, n_ext=1, n_gen=42, name='syn-1-HV-urban--0-sw',
                        trafos_per_ext=4, load_share=0.3, sgen_share=0.2)


# --------------------------------------------------------------------------
# synthetic time-series profiles (SimBench factored form)
# --------------------------------------------------------------------------
def _relative_profiles(rng, n_types, kind, n_steps=N_STEPS):
    t = np.arange(n_steps)
    day = (t % 96) / 96.0
    week = (t // 96) % 7
    year = t / n_steps
    out = np.empty((n_steps, n_types))
    for k in range(n_types):
        if kind == 'load':
            ph = rng.uniform(-0.08, 0.08)
            shape = (0.35 + 0.3 * np.exp(-((day - 0.33 - ph) / 0.09) ** 2)
                     + 0.45 * np.exp(-((day - 0.79 - ph) / 0.11) ** 2))
            shape *= np.where(week >= 5, rng.uniform(0.7, 0.95), 1.0)
            shape *= 1.0 + 0.18 * np.cos(2 * np.pi * year)
            noise = rng.uniform(0.85, 1.15, n_steps)
            out[:, k] = np.clip(shape * noise, 0.05, None)
        elif kind == 'pv':
            width = 0.16 + 0.07 * np.sin(np.pi * year)
            sun = np.exp(-((day - 0.5) / width) ** 2) - 0.12
            cloud = rng.beta(4, 1.5, n_steps // 96 + 1)[t // 96]
            out[:, k] = np.clip(sun, 0.0, None) * cloud * rng.uniform(0.9, 1.0, n_steps)
        else:  # wind-like: slowly varying Weibull-ish
            steps = rng.normal(0, 1, n_steps // 8 + 2)
            slow = np.interp(t / 8.0, np.arange(len(steps)), np.cumsum(steps))
            slow = (slow - slow.min()) / (np.ptp(slow) + 1e-12)
            out[:, k] = np.clip(slow ** 1.5 * rng.uniform(0.9, 1.1, n_steps), 0.0, 1.0)
        out[:, k] /= out[:, k].max()
    return out


class Profiles(dict):
    """dict[(unit, column)] -> DataFrame[N_STEPS × n_units] with column labels
    equal to the unit table index (opf_env.py:343), plus the factored form
    `rel[(unit,col)]` [T × n_types], `typ` [n_units], `peak` [n_units] so a
    device can hold the small relative tables instead of the dense ones."""

    def __init__(self):
        super().__init__()
        self.rel, self.typ, self.peak = {}, {}, {}

    def add(self, key, rel, typ, peak, index):
        self.rel[key], self.typ[key], self.peak[key] = rel, np.asarray(typ), np.asarray(peak, float)
        self._index = getattr(self, '_index', {})
        self._index[key] = np.asarray(index)
        self[key] = pd.DataFrame(rel[:, typ] * self.peak[key][None, :], columns=index)


def factored_profile(profiles, key):
    """(rel [T,n_types], typ [n_cols], peak [n_cols]) such that column j of
    profiles[key] equals rel[:, typ[j]] * peak[j] bit for bit.  Uses the
    SimBench-style factored form when the dict carries it (and it still matches
    after profile repair), otherwise one type per column with peak 1."""
    df = profiles[key]
    rel = getattr(profiles, 'rel', {}).get(key)
    if rel is not None:
        index = profiles._index[key]
        pos = np.array([int(np.flatnonzero(index == c)[0]) for c in df.columns], dtype=int)
        typ, peak = profiles.typ[key][pos], profiles.peak[key][pos]
        if len(pos) == 0 or np.array_equal(rel[:, typ] * peak[None, :], df.to_numpy()):
            return np.ascontiguousarray(rel), typ.astype(np.int32), peak.astype(float)
    vals = np.ascontiguousarray(df.to_numpy(dtype=float))
    return vals, np.arange(vals.shape[1], dtype=np.int32), np.ones(vals.shape[1])


def synthetic_profiles(net, seed: int = 1, n_steps: int = N_STEPS) -> Profiles:
    rng = np.random.default_rng(seed)
    prof = Profiles()
    n_ld = len(net.load)
    rel_l = _relative_profiles(rng, 6, 'load', n_steps)
    typ_l = rng.integers(0, 6, n_ld)
    prof.add(('load', 'p_mw'), rel_l, typ_l, net.load.p_mw.to_numpy(float), net.load.index)
    rel_q = rel_l * rng.uniform(0.9, 1.0, rel_l.shape)
    rel_q /= rel_q.max(axis=0)
    prof.add(('load', 'q_mvar'), rel_q, typ_l, net.load.q_mvar.to_numpy(float), net.load.index)
    n_sg = len(net.sgen)
    rel_s = np.concatenate([_relative_profiles(rng, 3, 'pv', n_steps),
                            _relative_profiles(rng, 2, 'wind', n_steps)], axis=1)
    p_s = net.sgen.p_mw.to_numpy(float)
    big = p_s > np.quantile(p_s, 0.8) if n_sg else np.zeros(0, bool)
    typ_s = np.where(big, rng.integers(3, 5, n_sg), rng.integers(0, 3, n_sg))
    prof.add(('sgen', 'p_mw'), rel_s, typ_s, p_s, net.sgen.index)
    # SimBench's get_absolute_values carries these keys even for grids without
    # such units (empty tables), and build_simbench_net.py:67-81 relies on that
    if len(net.gen):
        rel_g = _relative_profiles(rng, 2, 'load', n_steps)
        prof.add(('gen', 'p_mw'), rel_g, rng.integers(0, 2, len(net.gen)),
                 net.gen.p_mw.to_numpy(float), net.gen.index)
    else:
        prof[('gen', 'p_mw')] = pd.DataFrame(index=np.arange(n_steps))
    if len(net.storage):
        n_st = len(net.storage)
        base = _relative_profiles(rng, 2, 'wind', n_steps) * 2.0 - 1.0   # charge/discharge
        prof.add(('storage', 'p_mw'), base, rng.integers(0, 2, n_st),
                 net.storage.p_mw.to_numpy(float), net.storage.index)
    else:
        prof[('storage', 'p_mw')] = pd.DataFrame(index=np.arange(n_steps))
    return prof


def synthetic_hv_small(seed: int = 0):
    """Small meshed HV grid with PV generators for fast tests (not a BASELINE config)."""
    return synthetic_hv(seed, nb=40, n_ext=1, n_gen=4, name='syn-hv-small', trafos_per_ext=2,
                        load_share=0.5, sgen_share=0.4)


def synthetic_hv_small_sw(seed: int = 0):
    """hv-small with a (closed) line switch at the from-bus of every line, switch index = line
    index — the shape of the SimBench '-sw' variants as far as the switch actuators of
    examples/network_reconfiguration.py need it."""
    from . import net as ppn
    net, prof = synthetic_hv_small(seed)
    for idx in net.line.index:
        ppn.create_switch(net, int(net.line.at[idx, 'from_bus']), int(idx), 'l', closed=True)
    return net, prof


def synthetic_mv_3w(seed: int = 0):
    """mv-small plus a 110/20/10 kV three-winding transformer in parallel to the two-winding ones: its mv
    terminal on the first busbar section, a 10 kV bus with a load and an sgen on its lv terminal (test grid
    for `Trafo3wOverloadConstraint`, constraints.py:164-172; not a BASELINE config)."""
    from . import net as ppn
    net, prof = synthetic_mv_small(seed)
    lv = ppn.create_bus(net, 10.0, min_vm_pu=np.nan, max_vm_pu=np.nan)
    ppn.create_transformer3w_from_parameters(
        net, 0, 1, lv, 110.0, 20.0, 10.0, 40.0, 25.0, 15.0, vk_hv_percent=10.4, vk_mv_percent=6.4, vk_lv_percent=10.2,
        vkr_hv_percent=0.28, vkr_mv_percent=0.32, vkr_lv_percent=0.35, pfe_kw=28.0, i0_percent=0.06,
        shift_mv_degree=150.0, shift_lv_degree=150.0, tap_side='hv', tap_neutral=0, tap_pos=1, tap_step_percent=1.2,
        max_loading_percent=12.0)
    li = ppn.create_load(net, lv, 2.4, 0.7)
    si = ppn.create_sgen(net, lv, 1.6, 0.0)
    ppn.finalize(net)
    rng = np.random.default_rng(seed + 7)
    for key, idx, peak in ((('load', 'p_mw'), li, 2.4), (('load', 'q_mvar'), li, 0.7), (('sgen', 'p_mw'), si, 1.6)):
        df = prof[key]
        src = df.columns[int(rng.integers(len(df.columns)))]
        df[idx] = df[src].to_numpy() / max(float(df[src].max()), 1e-9) * peak
        if hasattr(prof, 'rel'):
            prof.rel.pop(key, None)
    return net, prof


def synthetic_hv_large(seed: int = 0):
    """1 000-bus meshed HV grid (not a BASELINE config): its 8 788 LU blocks do not fit a CU's LDS — the workload of the
    memory-resident kernel."""
    return synthetic_hv(seed + 5, nb=1000, n_ext=2, n_gen=10, name='syn-hv-large')


GRIDS = {
    'hv-large': synthetic_hv_large,
    '1-LV-rural1--0-sw': synthetic_lv_rural1,
    '1-MV-urban--0-sw': synthetic_mv_urban,
    '1-HV-mixed--0-sw': synthetic_hv_mixed,
    '1-HV-urban--0-sw': synthetic_hv_urban,
    'mv-small': synthetic_mv_small,
    'hv-small': synthetic_hv_small,
    'hv-small-sw': synthetic_hv_small_sw,
    'mv-3w': synthetic_mv_3w,
}


def get_grid(code: str, seed: int = 0):
    """Return (net, profiles) for a SimBench code (synthetic stand-in)."""
    return GRIDS[code](seed)
