"""Stub of pandapower.networks for the golden generator: `case_ieee30` (examples/non_simbench_net.py)
is answered with this repo's OPF-ready 9-bus case — the fixture pins the reference class's own
logic (normal_around_mean sampling with std_dev columns, gen P actions), not a particular grid."""
from opfgym_amd import grids as _grids


def case_ieee30():
    return _grids.case9_opf()
