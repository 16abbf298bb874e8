"""CPU: the native, parameterised definition builder (opfgym_amd/native_definition.py: rule tables + one
interpreter) against the definitions RECORDED from the reference's own classes (opfgym_amd/definitions/, written by
tests/golden/make_definitions.py): every recorded definition of a class with a recipe is rebuilt from its constructor
arguments and compared value for value — element tables, action / observation / state keys, surviving profile
columns.  And constructor arguments nobody recorded must work (they raised ImportError)."""
import json
import os

import numpy as np
import pytest

from opfgym_amd import definition, envs, native_definition, simbench_build


def _recorded():
    idx = json.load(open(os.path.join(definition.DEF_DIR, 'index.json')))
    for key, name in sorted(idx.items()):
        ref_path, items = json.loads(key)
        if native_definition.has_recipe(ref_path):
            yield name, ref_path, dict((k, v) for k, v in items)


@pytest.mark.parametrize('name,ref_path,kwargs', list(_recorded()), ids=[n for n, _, _ in _recorded()])
def test_native_builder_reproduces_the_recorded_definition(name, ref_path, kwargs):
    rec = definition.load(os.path.join(definition.DEF_DIR, name))
    grid_seed, prepare = int(kwargs.pop('__grid_seed', 0)), kwargs.pop('__prepare', None)
    nat = native_definition.build(ref_path, kwargs, grid_seed, getattr(simbench_build, prepare) if prepare else None)
    a, b = definition.tables_to_arrays(rec.net), definition.tables_to_arrays(nat.net)
    assert set(a) == set(b), sorted(set(a) ^ set(b))
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.shape == y.shape, (k, x.shape, y.shape)
        same = np.array_equal(x.astype(float), y.astype(float), equal_nan=True) if x.dtype.kind in 'fiub' else np.array_equal(x, y)
        assert same, (name, k)
    for kind in ('act_keys', 'obs_keys', 'state_keys'):
        ka, kb = getattr(rec, kind), getattr(nat, kind)
        assert [(u, c) for u, c, _ in ka] == [(u, c) for u, c, _ in kb], kind
        for (_, _, i), (_, _, j) in zip(ka, kb):
            assert np.array_equal(np.asarray(i), np.asarray(j)), kind
    assert len(rec.n_minus_one_keys) == len(nat.n_minus_one_keys)
    assert set(rec.profiles) == set(nat.profiles)
    for key in rec.profiles:
        assert list(rec.profiles[key].columns) == list(nat.profiles[key].columns), key
        assert np.array_equal(rec.profiles[key].to_numpy(), nat.profiles[key].to_numpy()), key


def test_unrecorded_constructor_arguments_build_natively():
    """cos_phi, load_scaling, thresholds ... are live parameters of the definition, not keys of a file."""
    if definition.reference_class('opfgym.envs.VoltageControl') is not None:
        pytest.skip('the reference is importable here')
    base = envs.VoltageControl(simbench_network_name='1-MV-urban--0-sw', batch_size=1, defer_device=True)
    e1 = envs.VoltageControl(simbench_network_name='1-MV-urban--0-sw', cos_phi=0.9, batch_size=1, defer_device=True)
    assert np.allclose(e1.net.sgen.max_s_mva, e1.net.sgen.max_max_p_mw / 0.9)
    assert not np.allclose(e1.net.sgen.max_s_mva, base.net.sgen.max_s_mva)
    e2 = envs.VoltageControl(simbench_network_name='1-MV-urban--0-sw', load_scaling=1.2, batch_size=1, defer_device=True)
    assert (e2.net.load.scaling == 1.2).all()
    assert np.allclose(e2.net.load.max_max_p_mw * 1.5, base.net.load.max_max_p_mw * 1.2)
    e3 = envs.VoltageControl(simbench_network_name='1-MV-urban--0-sw', min_sgen_power=1.0, batch_size=1, defer_device=True)
    assert 0 < e3.n_actions < base.n_actions
    e4 = envs.QMarket(simbench_network_name='mv-small', max_q_exchange=0.25, voltage_band=0.03, batch_size=1, defer_device=True)
    assert (e4.net.ext_grid.max_q_mvar == 0.25).all() and np.allclose(e4.net.bus.max_vm_pu, 1.03)
    e5 = envs.EcoDispatch(simbench_network_name='hv-small', max_price_eur_gwh=0.8, min_power=5.0, batch_size=1, defer_device=True)
    assert (e5.net.poly_cost.max_cp1_eur_per_mw == 0.8).all()
    e6 = envs.MaxRenewable(simbench_network_name='1-LV-rural1--0-sw', min_sgen_power=0.01, min_storage_power=0.01,
                           gen_scaling=0.9, batch_size=1, defer_device=True)
    assert (e6.net.sgen.scaling == 0.9).all() and len(e6.net.ext_grid) == 1
    e7 = envs.LoadShedding(simbench_network_name='mv-small', min_load_power=0.5, max_p_exchange=5.0, batch_size=1, defer_device=True)
    assert (e7.net.ext_grid.max_p_mw == 5.0).all()
    with pytest.raises(TypeError, match='unknown definition argument'):
        native_definition.build('opfgym.envs.VoltageControl', dict(simbench_network_name='mv-small', cosphi=0.9))
    with pytest.raises(ImportError, match='stand-in'):
        envs.VoltageControl(simbench_network_name='1-MV-semiurb--1-sw', batch_size=1, defer_device=True)


def test_voltage_control_refuses_a_grid_with_generators():
    with pytest.raises(AssertionError, match='without generators'):
        native_definition.build('opfgym.envs.VoltageControl', dict(simbench_network_name='hv-small'))
