"""A net shaped like a REAL pandapower 2.13 net (every column pandapower creates, its dtypes, None / NaN in object
columns, NaN tap data, the empty element tables) must go through both converters unchanged, and content that
`pp.runpp` (opf_env.py:703) would model but the converters do not must be REFUSED by both, not dropped."""
import copy

import numpy as np
import pytest

from opfgym_amd import grids
from opfgym_amd.case import net_to_case
from oracle import pd2ppc
from oracle import pf_oracle as po
from pp_schema import dress_as_pandapower, unmodelled_variants
from test_pd2ppc_differential import _compare, _random_net


def _nets():
    yield 'mv-small', grids.get_grid('mv-small')[0]
    yield 'hv-small-sw', grids.get_grid('hv-small-sw')[0]
    yield 'mv-3w', grids.get_grid('mv-3w')[0]
    yield '1-LV-rural1--0-sw', grids.get_grid('1-LV-rural1--0-sw')[0]
    for seed in (3, 7, 11, 19):
        yield f'random{seed}', _random_net(np.random.default_rng(seed))


@pytest.mark.parametrize('name,net', list(_nets()), ids=[n for n, _ in _nets()])
def test_full_schema_changes_nothing(name, net):
    plain = copy.deepcopy(net)
    dressed = dress_as_pandapower(copy.deepcopy(net), seed=1)
    assert len(dressed['trafo'].columns) >= 30 and dressed['line']['from_bus'].dtype == np.uint32
    try:
        c0 = net_to_case(plain)
    except ValueError:
        pytest.skip('no slack left')
    c1, ppc1 = _compare(dressed)
    assert np.array_equal(c0.f, c1.f) and np.array_equal(c0.t, c1.t) and np.array_equal(c0.bus_type, c1.bus_type)
    for a in ('yff', 'yft', 'ytf', 'ytt', 'kf', 'kt', 'vm_set', 'va_set', 'gs', 'bs'):
        assert np.array_equal(getattr(c0, a), getattr(c1, a), equal_nan=True), a
    ppc0 = pd2ppc.build_ppc(plain)
    for a in ('r', 'x', 'b', 'tap', 'shift', 'status', 'pd', 'qd', 'gs', 'bs', 'vm', 'va'):
        assert np.array_equal(getattr(ppc0, a), getattr(ppc1, a), equal_nan=True), a
    # and the oracle's plug-in solves it to the same tables
    ref, got = copy.deepcopy(plain), copy.deepcopy(dressed)
    try:
        po.runpp(ref, enforce_q_lims=True)
    except po.LoadflowNotConverged:
        return
    po.runpp(got, enforce_q_lims=True)
    for tbl in ('res_bus', 'res_line', 'res_trafo', 'res_ext_grid'):
        for col in ref[tbl].columns:
            assert np.array_equal(ref[tbl][col].to_numpy(float), got[tbl][col].to_numpy(float), equal_nan=True), (tbl, col)


@pytest.mark.parametrize('label,mutate', unmodelled_variants(), ids=[lbl for lbl, _ in unmodelled_variants()])
def test_unmodelled_content_is_refused(label, mutate):
    net = dress_as_pandapower(_random_net(np.random.default_rng(7)), seed=2)
    net_to_case(net)                      # the dressed net itself is fine
    pd2ppc.build_ppc(net)
    mutate(net)
    with pytest.raises(ValueError, match=label.split('.')[0]):
        net_to_case(net)
    with pytest.raises(ValueError, match=label.split('.')[-1].split('_')[0]):
        pd2ppc.build_ppc(net)


def test_nan_tap_neutral_means_no_tap_change():
    """pandapower `_calc_tap_from_dataframe`: tap_steps = step_percent * (pos - neutral) / 100 with NaN replaced
    by 0, so a transformer without a neutral position runs at its rated ratio."""
    net, _ = grids.get_grid('hv-small')
    base = copy.deepcopy(net)
    base.trafo['tap_pos'] = base.trafo['tap_neutral']
    net.trafo['tap_pos'] = net.trafo['tap_neutral'] + 3
    net.trafo['tap_neutral'] = np.nan
    c0, c1 = net_to_case(base), net_to_case(net)
    assert np.allclose(c0.ybus_dense(), c1.ybus_dense(), rtol=0, atol=1e-14)
    _compare(net)


@pytest.mark.gpu
@pytest.mark.parametrize('name,net', list(_nets()), ids=[n for n, _ in _nets()])
def test_full_schema_net_through_the_gpu_plugin(name, net):
    """The same dressed nets through the batch-1 `power_flow_solver(net)` seam (opf_env.py:53) on the GPU, against
    the oracle's tables."""
    from opfgym_amd import power_flow_solver
    dressed = dress_as_pandapower(copy.deepcopy(net), seed=1)
    ref = copy.deepcopy(dressed)
    try:
        po.runpp(ref, enforce_q_lims=True)
    except (po.LoadflowNotConverged, ValueError):
        pytest.skip('random case without a solution')
    power_flow_solver(dressed, enforce_q_lims=True)
    for tbl, cols, tol in (('res_bus', ('vm_pu', 'va_degree'), 1e-8), ('res_line', ('loading_percent',), 1e-6),
                           ('res_trafo', ('loading_percent',), 1e-6), ('res_ext_grid', ('p_mw', 'q_mvar'), 1e-6)):
        for col in cols:
            a, b = dressed[tbl][col].to_numpy(float), ref[tbl][col].to_numpy(float)
            assert a.shape == b.shape and np.allclose(a, b, rtol=0, atol=tol, equal_nan=True), (tbl, col)


@pytest.mark.gpu
def test_tap_step_degree_on_the_gpu():
    """VERDICT r02 probe: tap_pos = neutral + 2 with tap_step_degree = 1.5 on the HV/MV transformers, and an ideal
    phase shifter: the product's converter + kernel against the oracle's converter + solver."""
    from opfgym_amd import power_flow_solver
    for shifter in (False, True):
        net, _ = grids.get_grid('hv-small')
        net.trafo['tap_step_degree'] = 1.5
        net.trafo['tap_pos'] = net.trafo['tap_neutral'] + 2
        net.trafo['tap_phase_shifter'] = shifter
        if shifter:
            net.trafo['tap_step_percent'] = 0.0
        ref = copy.deepcopy(net)
        po.runpp(ref, enforce_q_lims=True)
        power_flow_solver(net, enforce_q_lims=True)
        for tbl, col, tol in (('res_bus', 'vm_pu', 1e-9), ('res_bus', 'va_degree', 1e-7), ('res_trafo', 'loading_percent', 1e-6),
                              ('res_ext_grid', 'p_mw', 1e-6), ('res_ext_grid', 'q_mvar', 1e-6)):
            assert np.allclose(net[tbl][col].to_numpy(float), ref[tbl][col].to_numpy(float), rtol=0, atol=tol, equal_nan=True), (shifter, tbl, col)
