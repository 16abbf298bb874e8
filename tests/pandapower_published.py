"""Power-flow results that pandapower ITSELF published, as data (test code only).

pandapower is absent from this image and from /root/reference (pyproject.toml:32 names it as a third-party
dependency), so nothing here was produced by running it.  Every expected number below is a constant that
pandapower prints in its documentation or asserts in its own test-suite; the networks are the ones those
sources build.  Both are written down FROM MEMORY of the public sources — marked [3P, from memory] as in
SURVEY.md — under this rule: the constants were written down first, the network was then solved with
`oracle/pf_oracle.py`, and a vector was kept only if every constant of it came out to its printed digits
(an accidental agreement of 8-9 digits on several numbers is not plausible; a mis-remembered network or
constant simply does not reproduce and was left out — see `NOT_REPRODUCED`).

Sources
  DOCS   pandapower documentation, "A short introduction" / README minimal example (3 buses: 20 kV slack at
         1.02 p.u., std-type transformer "0.4 MVA 20/0.4 kV", 100 m "NAYY 4x50 SE", 0.1 MW / 0.05 Mvar load);
         the printed `net.res_bus` table.  Std-type parameters spelled out from pandapower's std-type library.
  TESTS  pandapower/test/loadflow/test_results.py with the networks of
         pandapower/test/loadflow/result_test_network_generator.py (`add_grid_connection`, `create_test_line`,
         `add_test_load_sgen`, `add_test_line`, `add_test_gen`, `add_test_enforce_qlims`, `add_test_trafo`,
         `add_test_trafo3w`); that file labels its constants "result values from powerfactory" and asserts them
         with v_tol = 1e-6 p.u., l_tol = 1e-3 %, i_tol = 1e-6 kA, s_tol = 5e-3 kW — the tolerances used here.
         (Constants in kW/kvar in the older revisions of that file are given in MW/Mvar here.)

Which element formulas each vector exercises is listed per case (`covers`) and summarised in the header of
oracle/pf_oracle.py.

HOW MUCH THIS PINS: the selection rule above makes the oracle the filter of its own pins — a vector that did
not reproduce was dropped, not debugged against the source (which is not in this image).  Read these vectors as
RECALLED-AND-SELF-CONSISTENT: strong evidence against a gross modelling error on the paths they cover (nine-digit agreement
of several numbers per network does not happen by accident), not an independent reproduction of a pandapower run.  The
vectors that did not reproduce stay visible as expected failures (`NOT_REPRODUCED`, tests/test_pandapower_published.py::
test_recalled_constants_the_oracle_does_not_reproduce) and, where a hand calculation decides which side is wrong, say so.
What no recalled number covers is pinned by equivalence instead: tests/metamorphic.py.
"""
import numpy as np

from opfgym_amd import net as N

V_TOL, L_TOL, I_TOL, S_TOL = 1e-6, 1e-3, 1e-6, 5e-6     # p.u., percent, kA, MW/Mvar (= 5e-3 kW)
DOC_TOL = 5e-7                                           # six printed decimals


def _test_line(net, b1, b2, **kw):
    """`create_test_line`: 12.2 km, 0.08 + j0.12 Ohm/km, 300 nF/km, 0.2 kA, df = 0.8."""
    return N.create_line_from_parameters(net, b1, b2, 12.2, r_ohm_per_km=0.08, x_ohm_per_km=0.12, c_nf_per_km=300,
                                         max_i_ka=.2, df=.8, **kw)


def _grid_connection(net, vn_kv=20.):
    """`add_grid_connection`: slack bus at 1.01 p.u., one test line to a second bus."""
    b1 = N.create_bus(net, vn_kv)
    N.create_ext_grid(net, b1, vm_pu=1.01)
    b2 = N.create_bus(net, vn_kv)
    l1 = _test_line(net, b1, b2)
    return b1, b2, l1


def _test_trafo(net, hv, lv, **kw):
    return N.create_transformer_from_parameters(
        net, hv, lv, vk_percent=5., vkr_percent=2., i0_percent=.4, pfe_kw=2, sn_mva=0.4, vn_hv_kv=22, vn_lv_kv=0.42,
        tap_neutral=5, tap_step_percent=1.25, tap_pos=3, shift_degree=150, tap_side='hv', **kw)


def docs_minimal_example():
    net = N.Net()
    b1, b2, b3 = N.create_bus(net, 20.), N.create_bus(net, .4), N.create_bus(net, .4)
    N.create_ext_grid(net, b1, vm_pu=1.02)
    N.create_load(net, b3, p_mw=.1, q_mvar=.05)
    # std type "0.4 MVA 20/0.4 kV"
    N.create_transformer_from_parameters(net, b1, b2, sn_mva=.4, vn_hv_kv=20., vn_lv_kv=.4, vk_percent=6.,
                                         vkr_percent=1.425, pfe_kw=1.35, i0_percent=.3375, shift_degree=150,
                                         tap_side='hv', tap_neutral=0, tap_pos=0, tap_step_percent=2.5)
    # std type "NAYY 4x50 SE"
    N.create_line_from_parameters(net, b2, b3, .1, r_ohm_per_km=.642, x_ohm_per_km=.083, c_nf_per_km=210, max_i_ka=.142)
    checks = [('res_bus', 'vm_pu', b1, 1.02, DOC_TOL), ('res_bus', 'vm_pu', b2, 1.008843, DOC_TOL),
              ('res_bus', 'vm_pu', b3, 0.964431, DOC_TOL), ('res_bus', 'va_degree', b1, 0.0, DOC_TOL),
              ('res_bus', 'va_degree', b2, -0.760126, DOC_TOL), ('res_bus', 'va_degree', b3, 0.115859, DOC_TOL),
              ('res_ext_grid', 'p_mw', 0, 0.107265, DOC_TOL), ('res_ext_grid', 'q_mvar', 0, 0.052675, DOC_TOL)]
    return N.finalize(net), checks, {}


def tests_load_sgen():
    net = N.Net()
    b1, b2, l1 = _grid_connection(net)
    N.create_load(net, b2, p_mw=1.2, q_mvar=1.1)
    N.create_sgen(net, b2, p_mw=.5, q_mvar=-.1)
    # units that are out of service must not change the result (the generator adds them for that purpose)
    N.create_load(net, b2, p_mw=1.2, q_mvar=1.1, in_service=False)
    N.create_sgen(net, b2, p_mw=.5, q_mvar=-.1, in_service=False)
    return N.finalize(net), [('res_bus', 'vm_pu', b2, 1.00477465, V_TOL)], {}


def tests_line():
    """Two parallel systems on the first line, a second line energised from the slack bus and open at its far end
    (line switch), a line out of service."""
    net = N.Net()
    b1, b2, l1 = _grid_connection(net)
    net.line.loc[l1, 'parallel'] = 2
    N.create_load(net, b2, p_mw=1.2, q_mvar=1.1)
    l2 = _test_line(net, b1, b2)
    N.create_switch(net, b2, l2, 'l', closed=False)
    b3 = N.create_bus(net, 20.)
    _test_line(net, b2, b3, in_service=False)
    checks = [('res_bus', 'vm_pu', b2, 1.007395422, V_TOL),
              ('res_line', 'loading_percent', l1, 14.578, L_TOL), ('res_line', 'loading_percent', l2, 8.385, L_TOL),
              # i_ka of the test, through loading = i_ka / (max_i_ka df parallel)
              ('i_ka', None, l1, 0.0466479, I_TOL), ('i_ka', None, l2, 0.0134154, I_TOL)]
    return N.finalize(net), checks, {}


def _gen_net(**gen_kw):
    net = N.Net()
    b1, b2, l1 = _grid_connection(net)
    b3 = N.create_bus(net, .4)          # (sic: the generator of the test puts a 0.4 kV bus behind the 20 kV test line)
    _test_line(net, b2, b3)
    N.create_load(net, b3, p_mw=1.2, q_mvar=1.1)
    g = N.create_gen(net, b3, p_mw=.5, vm_pu=1.0, **gen_kw)
    return net, b2, b3, g


def tests_gen():
    net, b2, b3, g = _gen_net()
    checks = [('res_bus', 'vm_pu', b2, 1.00584636, V_TOL), ('res_bus', 'vm_pu', b3, 1.0, V_TOL),
              ('res_gen', 'q_mvar', g, -0.260660, S_TOL), ('res_gen', 'p_mw', g, 0.5, S_TOL)]
    return N.finalize(net), checks, dict(enforce_q_lims=False)


def tests_enforce_qlims():
    net, b2, b3, g = _gen_net(min_q_mvar=-.2)
    checks = [('res_bus', 'vm_pu', b2, 1.00607194, V_TOL), ('res_bus', 'vm_pu', b3, 1.00045091, V_TOL),
              ('res_gen', 'q_mvar', g, -0.2, S_TOL)]
    return N.finalize(net), checks, dict(enforce_q_lims=True)


def tests_trafo():
    """Two parallel transformers with an off-nominal tap (hv side, position 3 of neutral 5, 1.25 % per step), a third
    one open at its lv side.  The test's constants for the lv bus and for the open transformer are NOT used: they
    were not recalled to their digits (see NOT_REPRODUCED)."""
    net = N.Net()
    b1, b2, l1 = _grid_connection(net)
    b3 = N.create_bus(net, .4)
    t1 = _test_trafo(net, b2, b3, parallel=2)
    t2 = _test_trafo(net, b2, b3)
    N.create_switch(net, b3, t2, 't', closed=False)
    N.create_load(net, b3, p_mw=.2, q_mvar=.05)
    checks = [('res_bus', 'vm_pu', b2, 1.010159155, V_TOL), ('res_trafo', 'loading_percent', t1, 28.7842, L_TOL)]
    return N.finalize(net), checks, {}


def tests_trafo3w():
    net = N.Net()
    b1, b2, l1 = _grid_connection(net)
    b3, b4 = N.create_bus(net, .6), N.create_bus(net, .4)
    N.create_load(net, b3, p_mw=.2, q_mvar=0.)
    N.create_load(net, b4, p_mw=.1, q_mvar=0.)
    t = N.create_transformer3w_from_parameters(
        net, b2, b3, b4, vn_hv_kv=22, vn_mv_kv=.64, vn_lv_kv=.42, sn_hv_mva=1, sn_mv_mva=.7, sn_lv_mva=.3,
        vk_hv_percent=1., vkr_hv_percent=.03, vk_mv_percent=.5, vkr_mv_percent=.02, vk_lv_percent=.25,
        vkr_lv_percent=.01, pfe_kw=.5, i0_percent=.1, tap_side='hv', tap_pos=2, tap_step_percent=1.25, tap_neutral=0)
    checks = [('res_bus', 'vm_pu', b2, 1.010117166, V_TOL), ('res_bus', 'vm_pu', b3, 0.955501331, V_TOL),
              ('res_bus', 'vm_pu', b4, 0.940630980, V_TOL), ('res_trafo3w', 'loading_percent', t, 37.21, 1e-2)]
    return N.finalize(net), checks, {}


CASES = {
    'docs_minimal_example': (docs_minimal_example, 'DOCS',
                             'trafo T-model from vk/vkr/pfe/i0 (tap neutral), LV cable, slack P/Q; angles without the '
                             'vector-group shift (calculate_voltage_angles auto-off below 70 kV)'),
    'tests_load_sgen': (tests_load_sgen, 'TESTS', 'line pi-model, load - sgen bus injection, in_service mask of units'),
    'tests_line': (tests_line, 'TESTS', 'parallel lines, df, line open at one end (auxiliary bus / charging current), '
                                        'line out of service, loading_percent and i_ka of lines'),
    'tests_gen': (tests_gen, 'TESTS', 'PV bus, reactive dispatch of a generator (res_gen.q_mvar)'),
    'tests_enforce_qlims': (tests_enforce_qlims, 'TESTS', 'enforce_q_lims outer loop: PV -> PQ at min_q_mvar'),
    'tests_trafo': (tests_trafo, 'TESTS', 'off-nominal tap on the hv side, parallel transformers, transformer open at '
                                          'one side, loading_percent of transformers (trafo_loading="current")'),
    'tests_trafo3w': (tests_trafo3w, 'TESTS', 'three-winding transformer: star equivalent from vk/vkr of the three pairs, '
                                              'magnetising branch, hv tap, loading = worst winding'),
}

# Recalled, written down, and NOT reproduced by the oracle — so either the network or the constant is
# mis-remembered, and nothing is asserted (kept as a record that the selection above is not a cherry-pick of
# whatever happened to agree):
NOT_REPRODUCED = {
    'tests_trafo lv bus': 'recalled vm_pu 0.9677532 at the 0.4 kV bus; oracle 0.980003 (hv bus and loading of the same '
                          'case reproduce to 5e-8 / 3e-4, so the constant is the doubtful part; hand calculation below)',
    'tests_trafo_tap': 'recalled 1.010114175 / 0.924072090; the network was not recalled well enough (oracle on the '
                       'guessed one: 1.010154 / 0.970778)',
    'tests_bus_bus_switch': 'recalled 0.982265380; the network carries ward / xward elements that are not modelled here',
    'tests_ext_grid, tests_shunt': 'no constant recalled to its digits',
}

# The tapped-transformer case by hand (VERDICT r05 #2c).  20 kV buses, 0.4 kV lv bus, system base 1 MVA.  Two transformers in
# parallel (the third is open at its lv side): sn 0.4 MVA, 22 / 0.42 kV, vk 5 %, vkr 2 %, hv tap at position 3 of neutral 5,
# 1.25 % per step -> tap factor 1 + (3 - 5) * 0.0125 = 0.975.
#   ratio to the bus bases: (22 * 0.975 / 0.42) / (20 / 0.4) = 51.0714 / 50 = 1.021429
#   no-load lv voltage: |V_hv| / ratio = 1.010159 / 1.021429 = 0.98897        (|V_hv| = the published 1.010159155, reproduced)
#   short-circuit impedance of ONE transformer on the lv bus base: 0.05 * (0.42 / 0.4)^2 * (1 / 0.4) = 0.13781 p.u.,
#     r = 0.02 * 1.1025 * 2.5 = 0.055125, x = sqrt(0.13781^2 - 0.055125^2) = 0.12631;  two in parallel: r = 0.02756, x = 0.06316
#   load 0.2 MW + j 0.05 Mvar = 0.2 + j 0.05 p.u.:  dV ~ (r P + x Q) / |V| = (0.005512 + 0.003158) / 0.985 = 0.00880
#   |V_lv| ~ 0.98897 - 0.00880 = 0.9802        -> the oracle's 0.980003; the recalled 0.9677532 would need 2.4 times the
#   voltage drop a 5 % transformer pair gives at half load: a mis-recalled constant, not an open question about the model.
TRAFO_LV_BY_HAND = 0.9802


def recalled_but_not_reproduced():
    """[(label, net builder, (table, column, index getter), recalled constant)] of NOT_REPRODUCED entries whose network IS
    recalled: asserted as EXPECTED FAILURES so that the gap stays in every test report."""
    def trafo_lv():
        net, checks, kw = tests_trafo()
        return net, ('res_bus', 'vm_pu', int(net.trafo.lv_bus.iloc[0])), 0.9677532, kw
    return [('tests_trafo lv bus', trafo_lv)]


def evaluate(net, checks):
    """[(label, got, expected, tol)] of a solved net (its res_* tables) against the published constants."""
    rows = []
    for tbl, col, idx, want, tol in checks:
        if tbl == 'i_ka':
            ln = net['line'].loc[idx]
            got = float(net['res_line'].loc[idx, 'loading_percent']) / 100.0 * float(ln['max_i_ka']) * float(ln['df']) \
                * float(ln['parallel'])
            rows.append((f'line[{idx}].i_ka', got, want, tol))
        else:
            rows.append((f'{tbl}.{col}[{idx}]', float(net[tbl].loc[idx, col]), want, tol))
    return rows


def failures(rows):
    return [(lab, got, want) for lab, got, want, tol in rows if not abs(got - want) <= tol or np.isnan(got)]
