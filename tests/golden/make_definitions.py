"""Record the problem definitions of the reference's environment classes (container only).

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_definitions.py

The product package does not restate the reference's `_define_opf` code (opfgym_amd/definition.py): where
`opfgym` is not importable it reads recorded definitions — DATA: the element tables after `_define_opf`,
the action / observation / state keys, the surviving profile columns.  This script produces them by
constructing the reference's OWN classes (imported from /root/reference under the throw-away stub packages of
tests/golden/_stubs, exactly as tests/golden/make_golden.py does) for every set of constructor arguments the
tests, the examples and bench.py use, and writes `opfgym_amd/definitions/<class>_<hash>.npz` + `index.json`.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, '_stubs'), ROOT, '/root/reference', HERE, os.path.join(ROOT, 'tests')]

import numpy as np  # noqa: E402

import simbench as stub_simbench  # noqa: E402  (the stub: serves this repo's synthetic grids)
import opfgym  # noqa: E402,F401  (the reference)
from opfgym_amd import definition, envs, grids, simbench_build  # noqa: E402
from scenarios import SCENARIOS  # noqa: E402

OUT = os.environ.get('OPFX_DEF_OUT') or definition.DEF_DIR
index = {}
last_code = {}
_get = stub_simbench.get_simbench_net


def remembering_get(code):
    last_code['code'] = code
    return _get(code)


stub_simbench.get_simbench_net = remembering_get


def record(ref_path, class_kwargs, grid_seed=0, prepare=None):
    """Stands in for definition.resolve while this script runs: build the reference class, save, return."""
    key = definition.request_key(ref_path, class_kwargs, grid_seed, prepare)
    if key in index:
        return definition.load(os.path.join(OUT, index[key]))
    cls = definition.reference_class(ref_path)
    assert cls is not None, ref_path
    stub_simbench.GRID_SEED = int(grid_seed)
    last_code.clear()
    prep = getattr(simbench_build, prepare) if prepare else None
    defn = definition.build_from_reference(cls, class_kwargs, prep)
    raw = None
    if defn.profiles:
        raw_net, raw = grids.get_grid(last_code['code'], int(grid_seed))
        if prep:
            prep(raw_net, raw)
    name = f'{ref_path.rsplit(".", 1)[1]}_{hashlib.sha1(key.encode()).hexdigest()[:10]}.npz'
    definition.save(defn, os.path.join(OUT, name), grid_code=last_code.get('code'), grid_seed=grid_seed,
                    raw_profiles=raw, prepare=prepare)
    index[key] = name
    print(f'{name}: {ref_path} {class_kwargs} seed={grid_seed} -> {len(defn.net.bus)} buses, '
          f'{sum(len(i) for _, _, i in defn.act_keys)} actions')
    return definition.load(os.path.join(OUT, name))          # what a reader of the file gets


definition.resolve = record
envs.definition.resolve = record

# every environment construction the repository performs without the reference at hand
REQUESTS = [(cls, dict(kw)) for cls, kw, _, _ in SCENARIOS.values()]
import bench  # noqa: E402
REQUESTS += [(c[0], dict(c[1])) for c in bench.CONFIGS.values()]
REQUESTS += [
    ('NetworkReconfiguration', dict(simbench_network_name='hv-small-sw', controllable_switch_idxs=(1, 3), grid_seed=26)),
    ('VoltageControl', dict(simbench_network_name='1-MV-urban--0-sw')),                       # examples/quickstart.py
    ('SecurityConstrained', dict(simbench_network_name='hv-small', n_minus_one_lines=(1, 2, 3))),
    ('MultiStageOpf', dict(simbench_network_name='1-LV-rural1--0-sw', steps_per_episode=3)),
    ('QMarket', dict(simbench_network_name='1-MV-urban--0-sw')),
    ('SecurityConstrainedVoltageControl', dict(simbench_network_name='1-HV-urban--0-sw', n_minus_one_lines='first8_non_islanding')),
]

if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    for f in os.listdir(OUT):
        if f.endswith('.npz'):
            os.remove(os.path.join(OUT, f))
    for cls, kw in REQUESTS:
        if kw.get('n_minus_one_lines') == 'first8_non_islanding':
            kw['n_minus_one_lines'] = (1, 3, 7)
        getattr(envs, cls)(batch_size=1, defer_device=True, seed=0, **kw)
    json.dump(index, open(os.path.join(OUT, 'index.json'), 'w'), indent=0, sort_keys=True)
    print(f'{len(index)} definitions')
