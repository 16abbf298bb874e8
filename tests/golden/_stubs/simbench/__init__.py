"""Throw-away stand-in for `simbench`: serves the repo's synthetic grids under
the SimBench codes (SimBench data is not available in this environment)."""
from opfgym_amd import grids as _grids

_PROFILES = {}
GRID_SEED = 0


def get_simbench_net(code):
    net, profiles = _grids.get_grid(code, GRID_SEED)
    _PROFILES[id(net)] = profiles
    return net


def profiles_are_missing(net):
    return False


def get_absolute_values(net, profiles_instead_of_study_cases=True):
    return _PROFILES[id(net)]
