"""Reward-function parameter objects for the fused device evaluator.

Host-side mirror of `/root/reference/opfgym/reward.py`: same class names and
constructor parameters.  The reward itself is computed on the GPU at the end
of `opfx_step` (csrc/opfx.hip, "reward" section) from these parameters; the
`__call__` here is the same scalar formula for host-side use (reward.py:61-98).
`estimate_reward_distribution` (reward.py:181-216) becomes ONE batched
reset+step instead of 3000 sequential ones.
"""
from __future__ import annotations

import copy

import numpy as np


def calculate_normalization_params(std_objective, mean_objective, std_penalty, mean_penalty, **kw):
    # reward.py:120-137
    return {'objective_factor': 1 / std_objective, 'objective_bias': -mean_objective / std_objective,
            'penalty_factor': 1 / std_penalty, 'penalty_bias': -mean_penalty / std_penalty}


def calculate_minmax01_params(min_objective, max_objective, min_penalty, max_penalty, **kw):
    # reward.py:140-158
    do, dp = max_objective - min_objective, max_penalty - min_penalty
    return {'objective_factor': 1 / do, 'objective_bias': -(min_objective / do),
            'penalty_factor': 1 / dp, 'penalty_bias': -(min_penalty / dp)}


def calculate_minmax11_params(min_objective, max_objective, min_penalty, max_penalty, **kw):
    # reward.py:161-178
    do, dp = (max_objective - min_objective) / 2, (max_penalty - min_penalty) / 2
    return {'objective_factor': 1 / do, 'objective_bias': -(min_objective / do + 1),
            'penalty_factor': 1 / dp, 'penalty_bias': -(min_penalty / dp + 1)}


def select_reward_scaler(reward_scaling: str):
    try:
        return {'minmax11': calculate_minmax11_params, 'minmax01': calculate_minmax01_params,
                'normalization': calculate_normalization_params}[reward_scaling]
    except KeyError:
        raise NotImplementedError('This reward scaling does not exist!')


def estimate_reward_distribution(env, num_samples: int = 3000, draws=None) -> dict:
    """reward.py:181-216 as one batch: random states, random actions, one
    power flow each; statistics of Σobjective and Σpenalty over converged rows.
    `draws`: explicit inputs to replay (see BatchedOpfEnv.sample_objective_penalty)."""
    objectives, penalties = env.sample_objective_penalty(num_samples, draws)
    objectives = objectives[~np.isnan(objectives)]
    penalties = penalties[~np.isnan(penalties)]
    return {
        'min_objective': objectives.min(), 'max_objective': objectives.max(),
        'min_penalty': penalties.min(), 'max_penalty': penalties.max(),
        'mean_objective': objectives.mean(), 'mean_penalty': penalties.mean(),
        'std_objective': np.std(objectives), 'std_penalty': np.std(penalties),
        'median_objective': np.median(objectives), 'median_penalty': np.median(penalties),
        'mean_abs_objective': np.abs(objectives).mean(), 'mean_abs_penalty': np.abs(penalties).mean()}


class RewardFunction:
    KIND = 0

    def __init__(self, penalty_weight: float = 0.5, clip_range=None, reward_scaling: str = None,
                 scaling_params: dict = None, env=None):
        self.penalty_weight = penalty_weight
        self.clip_range = clip_range
        self.scaling_params = self.prepare_reward_scaling(reward_scaling, scaling_params, env)
        self.valid_reward = 0.0
        self.invalid_penalty = 0.0
        self.invalid_objective_share = 1.0

    def prepare_reward_scaling(self, reward_scaling, scaling_params, env) -> dict:
        # reward.py:21-48
        if not isinstance(reward_scaling, str):
            return {'penalty_factor': 1, 'penalty_bias': 0, 'objective_factor': 1, 'objective_bias': 0}
        scaling_params = scaling_params or {}
        user = copy.copy(scaling_params)
        scaler = select_reward_scaler(reward_scaling)
        try:
            scaling_params.update(scaler(**scaling_params))
        except TypeError:
            scaling_params = estimate_reward_distribution(env, **scaling_params)
            scaling_params.update(scaler(**scaling_params))
        scaling_params.update(user)
        if np.isnan(scaling_params['penalty_bias']):
            scaling_params['penalty_bias'] = 0
        if np.isinf(scaling_params['penalty_factor']):
            scaling_params['penalty_factor'] = 1
        return scaling_params

    # scalar host formulas (reward.py:61-98) -----------------------------------
    def __call__(self, objective, penalty, valid):
        objective = self.adjust_objective(objective, valid)
        penalty = self.adjust_penalty(penalty, valid)
        objective = self.scale_objective(objective)
        penalty = self.scale_penalty(penalty)
        reward = self.compute_total_reward(objective, penalty)
        if self.clip_range:
            reward = self.clip_reward(reward)
        return reward

    def clip_reward(self, reward):
        return float(np.clip(reward, self.clip_range[0], self.clip_range[1]))

    def compute_total_reward(self, objective, penalty):
        if self.penalty_weight is None:
            return objective + penalty
        return objective * (1 - self.penalty_weight) + penalty * self.penalty_weight

    def scale_objective(self, objective):
        return objective * self.scaling_params['objective_factor'] + self.scaling_params['objective_bias']

    def scale_penalty(self, penalty):
        return penalty * self.scaling_params['penalty_factor'] + self.scaling_params['penalty_bias']

    def calculate_cost(self, penalty, valid):
        return 0.0 if valid else abs(penalty * self.scaling_params['penalty_factor'])

    def adjust_penalty(self, penalty, valid):
        return penalty

    def adjust_objective(self, objective, valid):
        return objective


class Summation(RewardFunction):
    KIND = 0


class Replacement(RewardFunction):
    KIND = 1

    def __init__(self, valid_reward: float = 1.0, **kwargs):
        super().__init__(**kwargs)
        if isinstance(valid_reward, str):
            # reward.py:237-239 calls an undefined helper (defect D2): numeric only
            raise NotImplementedError("Replacement(valid_reward=<str>) raises NameError in the "
                                      "reference as well; pass a number")
        self.valid_reward = valid_reward

    def adjust_objective(self, objective, valid):
        return objective + self.valid_reward if valid else 0.0


class Parameterized(RewardFunction):
    KIND = 2

    def __init__(self, valid_reward: float = 0.0, invalid_penalty: float = 0.5,
                 invalid_objective_share: float = 1.0, **kwargs):
        super().__init__(**kwargs)
        if isinstance(valid_reward, str) or isinstance(invalid_penalty, str):
            # reward.py:323-333 reads `offset` before assignment (defect D3): numeric only
            raise NotImplementedError('string heuristics raise UnboundLocalError in the reference')
        assert valid_reward >= 0, 'Valid reward must be >= 0'
        assert invalid_penalty >= 0, 'Invalid penalty must be >= 0'
        assert 0 <= invalid_objective_share <= 1, 'Objective share must be in [0, 1]'
        self.valid_reward = valid_reward
        self.invalid_penalty = invalid_penalty
        self.invalid_objective_share = invalid_objective_share

    def adjust_penalty(self, penalty, valid):
        return penalty + self.valid_reward if valid else penalty - self.invalid_penalty

    def adjust_objective(self, objective, valid):
        return objective if valid else objective * self.invalid_objective_share

    def calculate_cost(self, penalty, valid):
        return 0.0 if valid else super().calculate_cost(penalty, valid) + self.invalid_penalty


class OnlyObjective(RewardFunction):
    KIND = 3

    def __init__(self, **kwargs):
        super().__init__(penalty_weight=0.0, **kwargs)

    def adjust_penalty(self, penalty, valid):
        return 0.0


def load_reward_class(name: str):
    """util/import_class.py:6-16: class by (capitalised) name."""
    g = globals()
    for cand in (name, name.capitalize()):
        if cand in g and isinstance(g[cand], type) and issubclass(g[cand], RewardFunction):
            return g[cand]
    raise AttributeError(f'Class {name} not found in module opfgym_amd.reward!')
