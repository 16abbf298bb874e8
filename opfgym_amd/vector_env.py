"""Vector-environment face of the batched environments and the registration hook.

The reference registers its five benchmark environments with gymnasium
(opfgym/envs/__init__.py:12-35) and RL libraries then vectorise them by running
copies.  Here ONE batched environment already is the vector: `OpfVectorEnv` gives
it the shape of `gymnasium.vector.VectorEnv` (num_envs, single_* spaces, batched
reset/step, autoreset) without depending on gymnasium; `register()` adds the same
ids to gymnasium's registry when gymnasium is installed (it is not in this image).
"""
import numpy as np

from . import envs as _envs
from .batched_env import Box

ENV_IDS = {                      # opfgym/envs/__init__.py:12-35
    'MaxRenewable-v0': 'MaxRenewable', 'QMarket-v0': 'QMarket', 'VoltageControl-v0': 'VoltageControl',
    'EcoDispatch-v0': 'EcoDispatch', 'LoadShedding-v0': 'LoadShedding',
}


class OpfVectorEnv:
    """`num_envs` = batch size of the wrapped environment.

    autoreset_mode (names as in gymnasium >= 1.0):
      'same_step' (default) — sub-environments whose episode ended are reset inside the same
          `step()` call; the returned observation is the first one of the new episode and the
          last one of the old episode is in `info['final_obs']` (rows flagged by `info['_final_obs']`).
          With the single-step benchmark environments one call is therefore one full
          reset + step cycle for the whole batch: two kernel launches (running the reset inside the step's launch
          was built and measured in round 3 and is slower, DESIGN.md §4 k_reset).
      'next_step' — the reset happens in the NEXT `step()` call, whose action is ignored for
          those rows (reward 0, not terminated).
      'disabled' — the caller resets.
    Observations, rewards and flags are torch tensors on the environment's device, views of its
    persistent output buffers: valid until the next reset/step call (`as_numpy=True` copies them to
    the host instead).
    """

    def __init__(self, env, autoreset_mode='same_step', as_numpy=False):
        assert autoreset_mode in ('same_step', 'next_step', 'disabled')
        self.env = env
        self.num_envs = env.B
        self.autoreset_mode = autoreset_mode
        self.as_numpy = as_numpy
        self.single_observation_space = env.observation_space
        self.single_action_space = env.action_space
        so, sa = env.observation_space, env.action_space
        self.observation_space = Box(np.broadcast_to(so.low, (self.num_envs,) + so.shape).copy(),
                                     np.broadcast_to(so.high, (self.num_envs,) + so.shape).copy())
        self.action_space = Box(np.broadcast_to(sa.low, (self.num_envs,) + sa.shape).copy(),
                                np.broadcast_to(sa.high, (self.num_envs,) + sa.shape).copy())
        self._pending = None          # rows to reset at the start of the next step ('next_step')
        self.metadata = {'autoreset_mode': autoreset_mode}

    # per-instance state a partial reset has to put back for the rows it must not touch (sampling_mode: the
    # data source an episode started with under train_data='mixed' — multi-stage episodes keep sampling from it)
    _STATE = ('x', 'step_count', 'initial_obj', 'steps_dev', 'sampling_mode')

    # ---- helpers ----------------------------------------------------------------------
    def _out(self, x):
        return x.detach().cpu().numpy() if self.as_numpy and hasattr(x, 'detach') else x

    def _reset_rows(self, mask):
        """Reset the rows of `mask`, leave the others untouched; returns the reset observation
        (valid in the masked rows)."""
        env = self.env
        if bool(mask.all()):
            return env.reset()[0].clone()           # (a view of the output buffer: the next step overwrites it)
        keep = ~mask
        names = [n for n in self._STATE if getattr(env, n, None) is not None]
        saved = {n: getattr(env, n).clone() for n in names}
        obs_new = env.reset()[0].clone()
        for n, old in saved.items():
            getattr(env, n)[keep] = old[keep]
        return obs_new

    # ---- VectorEnv API ------------------------------------------------------------------
    def reset(self, seed=None, options=None):
        self._pending = None
        obs, info = self.env.reset(seed=seed, options=options)
        return self._out(obs), info

    def step(self, actions):
        env = self.env
        t = env.torch
        pending = self._pending
        self._pending = None
        if pending is not None and bool(pending.any()):
            # 'next_step': rows that finished in the previous call are reset now; their action
            # is ignored.  The whole batch is stepped (one launch), then the reset rows get their
            # freshly reset state and observation back and neutral outputs.
            obs_reset = self._reset_rows(pending)
            names = [n for n in self._STATE if getattr(env, n, None) is not None]
            fresh = {n: getattr(env, n).clone() for n in names}
            obs, reward, term, trunc, info = env.step(actions)
            obs, reward, term, trunc = obs.clone(), reward.clone(), term.clone(), trunc.clone()
            for n, v in fresh.items():
                getattr(env, n)[pending] = v[pending]
            obs[pending] = obs_reset[pending]
            reward[pending] = 0.0
            term[pending] = 0
            trunc[pending] = 0
            info = dict(info)
            done = (term.bool() | trunc.bool()) & ~pending
            self._pending = done
            return self._out(obs), self._out(reward), self._out(term), self._out(trunc), \
                {k: self._out(v) for k, v in info.items()}
        obs, reward, term, trunc, info = env.step(actions)
        info = dict(info)
        if self.autoreset_mode == 'same_step' and env.steps_per_episode == 1:
            # single-step episodes: every row ends on every step (opf_env.py:406-414) — no flag
            # inspection (a host sync) and no masking; only the observation buffer is reused by reset
            final_obs = obs.clone()
            if env.pf_for_obs:      # that reset runs a power flow through the same output buffers (opf_env.py:209-216)
                reward, term, trunc = reward.clone(), term.clone(), trunc.clone()
                info = {k: (v.clone() if hasattr(v, 'clone') else v) for k, v in info.items()}
            obs = env.reset()[0]
            info['final_obs'] = self._out(final_obs)
            info['_final_obs'] = self._out(term)
            return self._out(obs), self._out(reward), self._out(term), self._out(trunc), \
                {k: self._out(v) for k, v in info.items()}
        done = term.bool() | trunc.bool()
        if self.autoreset_mode == 'same_step' and bool(done.any()):
            final_obs = obs.clone()
            reward, term, trunc = reward.clone(), term.clone(), trunc.clone()
            info = {k: (v.clone() if hasattr(v, 'clone') else v) for k, v in info.items()}
            obs_reset = self._reset_rows(done)
            obs = t.where(done[:, None], obs_reset, final_obs)
            info['final_obs'] = self._out(final_obs)
            info['_final_obs'] = self._out(done)
        elif self.autoreset_mode == 'next_step':
            self._pending = done
        return self._out(obs), self._out(reward), self._out(term), self._out(trunc), \
            {k: self._out(v) for k, v in info.items()}

    def close(self):
        self.env.close()

    def __getattr__(self, name):
        return getattr(self.env, name)


def make_vec(env_id, num_envs, **kwargs):
    """`gymnasium.make_vec`-like constructor: make_vec('VoltageControl-v0', 8192, device='cuda:0')."""
    vec_kw = {k: kwargs.pop(k) for k in ('autoreset_mode', 'as_numpy') if k in kwargs}
    cls = getattr(_envs, ENV_IDS[env_id])
    return OpfVectorEnv(cls(batch_size=num_envs, **kwargs), **vec_kw)


def register():
    """Add the reference's ids (opfgym/envs/__init__.py:12-35) to gymnasium's registry with this
    package's batched environments as vector entry points.  No-op with a message when gymnasium
    is not installed."""
    try:
        from gymnasium.envs.registration import register as gym_register
    except ImportError:
        return False
    for env_id, cls in ENV_IDS.items():
        gym_register(id=env_id, entry_point=f'opfgym_amd.envs:{cls}',
                     vector_entry_point=f'opfgym_amd.vector_env:_vector_entry_{cls}')
    return True


def _make_entry(cls_name):
    def entry(num_envs=1, **kwargs):
        return make_vec(f'{cls_name}-v0', num_envs, **kwargs)
    entry.__name__ = f'_vector_entry_{cls_name}'
    return entry


for _cls in ENV_IDS.values():
    globals()[f'_vector_entry_{_cls}'] = _make_entry(_cls)
