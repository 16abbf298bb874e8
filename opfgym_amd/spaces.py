"""Observation / state / action spaces of ONE instance (opf_env.py:124-130, 792-826), gymnasium-shaped without gymnasium.
Split out of batched_env.py in round 6, no behaviour change."""
from __future__ import annotations

import numpy as np

class Box:
    """Minimal stand-in for gymnasium.spaces.Box (gymnasium is optional): bounds as
    float64 arrays, `shape`, and `sample()`.  `BatchedOpfEnv` exposes the per-instance
    spaces; a batch of actions is [B, *shape]."""

    def __init__(self, low, high, shape=None, seed=None):
        low, high = np.asarray(low, dtype=float), np.asarray(high, dtype=float)
        if shape is None:
            shape = np.broadcast(low, high).shape
        self.low = np.broadcast_to(low, shape).copy()
        self.high = np.broadcast_to(high, shape).copy()
        self.shape = tuple(shape)
        self._rng = np.random.default_rng(seed)

    def sample(self, batch=None):
        shape = self.shape if batch is None else (batch,) + self.shape
        return self.low + (self.high - self.low) * self._rng.random(shape)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape[-len(self.shape):] == self.shape and bool(((x >= self.low) & (x <= self.high)).all())


def get_obs_and_state_space(net, obs_or_state_keys, add_time_obs=False, add_mean_obs=False, seed=None,
                            bus_wise_obs=False):
    """opf_env.py:720-803: observation/state bounds from the constraint columns of the net."""
    lows, highs = [], []
    if add_time_obs:                                                       # :728-732
        lows.append(-np.ones(6)); highs.append(np.ones(6))
    for unit_type, column, idxs in obs_or_state_keys:
        if 'res_' in unit_type:
            unit_type = unit_type[4:]                                      # :735-737
        elif 'max_' in column or 'min_' in column:
            column = column[4:]                                            # :738-740
        df = net[unit_type]
        if column == 'va_degree':                                          # :742-746
            lo, hi = np.full(len(idxs), -30.0), np.full(len(idxs), 30.0)
        else:
            try:
                lo = df[f'min_min_{column}' if f'min_min_{column}' in df.columns else f'min_{column}'] \
                    .loc[idxs].to_numpy(float)
                hi = df[f'max_max_{column}' if f'max_max_{column}' in df.columns else f'max_{column}'] \
                    .loc[idxs].to_numpy(float)
            except KeyError:                                               # :757-761 lines / trafos
                lo = np.zeros(len(idxs))
                hi = df[f'max_{column}'].loc[idxs].to_numpy(float) * 1.5
            if column == 'vm_pu' or unit_type == 'ext_grid':               # :764-768
                diff = hi - lo
                lo, hi = lo - diff * 0.75, hi + diff * 0.75
        if not ('min' in column or 'max' in column) and 'scaling' in df.columns:   # :770-778
            sc = df['scaling'].loc[idxs].to_numpy(float)
            lo, hi = lo / sc, hi / sc
        if bus_wise_obs and unit_type == 'load':                           # :780-784
            buses = sorted(set(df.bus))
            bus_col = df.bus.loc[idxs].to_numpy() if len(idxs) == len(df) else df.bus.to_numpy()
            lo = np.array([lo[bus_col == b].sum() for b in buses])
            hi = np.array([hi[bus_col == b].sum() for b in buses])
        if len(lo) > 0 and len(lo) == len(hi):
            lows.append(lo); highs.append(hi)
    if add_mean_obs:                                                       # :791-797
        start = 1 if add_time_obs else 0
        lows.append(np.array([np.mean(l) for l in lows[start:] if len(l) > 1]))
        highs.append(np.array([np.mean(h) for h in highs[start:] if len(h) > 1]))
    assert not any(np.isnan(l).any() for l in lows) and not any(np.isnan(h).any() for h in highs)
    return Box(np.concatenate(lows) if lows else np.zeros(0), np.concatenate(highs) if highs else np.zeros(0),
               seed=seed)
