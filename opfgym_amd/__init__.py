"""opfgym_amd — MI355X-native batched AC power-flow backend for opfgym's
reset()/step()/reward hot path.  Public names mirror the
reference package (`opfgym/__init__.py:2-6`)."""
from .reward import RewardFunction  # noqa: F401
from .constraints import Constraint  # noqa: F401
from .batched_env import (BatchedOpfEnv, MultiStageOpfEnv, PowerFlowNotAvailable,  # noqa: F401
                          SecurityConstrainedOpfEnv, StochasticObservation)
from .solver_plugin import BatchedPowerFlowSolver, power_flow_solver  # noqa: F401

OpfEnv = BatchedOpfEnv
from .vector_env import OpfVectorEnv, make_vec, register  # noqa: F401,E402
