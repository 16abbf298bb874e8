"""Batches whose instances do not share one compiled plan: topology twins for bus-bus switch actuators (a closed coupler
changes the bus SET, so every switch state that occurs gets its own compiled twin of the environment) and rescue plans for a
breakdown of the static pivoting (`on_pivot_breakdown='resolve'`).

Split out of batched_env.py in round 6, no behaviour change: `TopologyMixin` is a mix-in of `BatchedOpfEnv`."""
from __future__ import annotations

import copy
import ctypes as C

import numpy as np

from . import capi
from .case import REF

class TopologyMixin:
    """See the module docstring."""

    MAX_RESCUE_PLANS = 8

    def _drop_rescue_envs(self, keep=0):
        """Destroy rescue environments (oldest first) until `keep` are left: the env handle before its context goes."""
        cache = getattr(self, '_rescue_envs', None) or {}
        while len(cache) > keep:
            _, _, h = cache.pop(next(iter(cache)))
            capi.lib().opfx_env_destroy(h)
        self._rescue_envs = cache

    def _bb_states_after(self, action, mode):
        """[B, n] states (0 / 1) of the bus-bus switch actuators AFTER this launch has applied `action` — opf_env.py:429-481
        for those columns alone, because the topology must be known before the launch: clip, absolute or incremental
        set-point, clamp, scaling, rounding.  mode 1 applies no action: the states in the store."""
        t, d = self.torch, self._bb_act
        as_d = lambda v: t.as_tensor(np.asarray(v, dtype=np.float64), device=self.device)
        prev = self.x[:, t.as_tensor(d['slots'], device=self.device)]
        if mode == 1 or action is None:
            return t.round(prev).to(t.int64)
        a = action[:, t.as_tensor(d['cols'], device=self.device)].clamp(0.0, 1.0)          # :429
        lo, hi, sc = as_d(d['lo']), as_d(d['hi']), as_d(d['sc'])
        incremental = bool(self.diff_action_step_size) and mode == 0
        if incremental:
            sp = (a * 2 - 1) * self.diff_action_step_size * (hi - lo) + prev * sc        # :452-458
        else:
            sp = a * (hi - lo) + lo                                                      # :461
        if (not self.autoscale_actions) or incremental:                                  # :464-470
            cl, ch = as_d(d['cl']), as_d(d['ch'])
            sp = t.where(t.isnan(ch), sp, t.minimum(sp, ch))
            sp = t.where(t.isnan(cl), sp, t.maximum(sp, cl))
        return t.round(sp / sc).clamp(0, 1).to(t.int64)                                  # :472-478

    def _topology_variant(self, states):
        """The twin of this environment on the topology `states` (one 0 / 1 per bus-bus switch actuator): the same net with
        those switches set, compiled to its own case, plan and descriptor; in it the switch columns are plain columns.
        Built on first use and kept."""
        key = tuple(int(v) for v in states)
        if key not in self._topology_variants:
            net = copy.deepcopy(self.net)
            for sw, v in zip(self._bb_switches, key):
                net['switch'].at[sw['index'], 'closed'] = bool(v)
            kw = dict(self._ctor)
            kw.update(self._ctor_kwargs)
            kw.update(batch_size=1, device=self.device_spec, defer_device=False, seed=None, _topology_fixed=True,
                      reward_function=self.reward_function if self.reward_function is not None else kw['reward_function'],
                      state_keys=self.state_keys, on_pivot_breakdown='ignore', copy_outputs=False)
            action_keys, observation_keys = kw.pop('action_keys'), kw.pop('observation_keys')
            var = type(self).__new__(type(self))                  # (the same class: its `_sampling_ops` decides the row layout)
            var.__dict__.update(self._pre_init_attrs)
            from .batched_env import BatchedOpfEnv                # (the class this mix-in belongs to; imported late: it imports this module)
            BatchedOpfEnv.__init__(var, net, action_keys, observation_keys, **kw)
            if var.store.n != self.store.n or var.n_obs_raw != self.n_obs_raw or var.n_constraints != self.n_constraints:
                raise RuntimeError('a topology variant laid its rows out differently from its parent')
            var._results_from = self._results_map(var)
            self._topology_variants[key] = var
        return self._topology_variants[key]

    def _results_map(self, var):
        """(positions in this environment's result bank, positions in the variant's) of the same physical quantities:
        bus voltages and angles through the net's bus numbers (a fused bus serves both of its net buses), branch
        loadings through the net elements, slack powers through the ext_grids; derived rows by position."""
        c, v = self.case, var.case
        nb, nbv = c.nb, v.nb
        mine, theirs = [], []
        inv = {}
        for net_bus, i in c.bus_lookup.items():
            inv.setdefault(i, net_bus)
        for i in range(nb):
            j = v.bus_lookup.get(inv[i], -1) if i in inv else -1
            if j >= 0:
                for off_m, off_v in ((0, 0), (nb, nbv)):
                    mine.append(off_m + i); theirs.append(off_v + j)
        vbr = {(int(k), int(e)): n for n, (k, e) in enumerate(zip(v.br_kind, v.br_elem))}
        for n, (k, e) in enumerate(zip(c.br_kind, c.br_elem)):
            if (int(k), int(e)) in vbr:
                mine.append(2 * nb + n); theirs.append(2 * nbv + vbr[(int(k), int(e))])
        ref_m, ref_v = np.flatnonzero(c.bus_type == REF), np.flatnonzero(v.bus_type == REF)
        for r, i in enumerate(ref_m):
            j = v.bus_lookup.get(inv.get(int(i), -1), -1)
            hit = np.flatnonzero(ref_v == j)
            if len(hit):
                for q in range(2):
                    mine.append(2 * nb + c.nbr + q * len(ref_m) + r); theirs.append(2 * nbv + v.nbr + q * len(ref_v) + int(hit[0]))
        base_m, base_v = 2 * nb + c.nbr + 2 * len(ref_m), 2 * nbv + v.nbr + 2 * len(ref_v)
        for i in range(nb):                                               # (reactive power of the generators per bus)
            j = v.bus_lookup.get(inv[i], -1) if i in inv else -1
            if j >= 0:
                mine.append(base_m + i); theirs.append(base_v + j)
        for k in range(min(len(self._xres), len(var._xres))):             # (derived rows: allocated in the same order)
            mine.append(base_m + nb + k); theirs.append(base_v + nbv + k)
        t = self.torch
        return (t.as_tensor(mine, dtype=t.int64, device=self.device), t.as_tensor(theirs, dtype=t.int64, device=self.device))

    def _launch_step_by_topology(self, action, mode, with_initial_obj):
        """One step for a batch whose instances sit on different topologies (bus-bus switch actuators): the rows are
        grouped by the switch states the action leaves them in, every group is stepped by the twin compiled for that
        topology (its rows gathered into a compact batch, as the pivot rescue does) and scattered back.  One host
        synchronisation per step (which topologies occur) and one launch per topology that occurs."""
        t, b = self.torch, self.buf
        states = self._bb_states_after(action, mode)
        weights = t.as_tensor([1 << k for k in range(states.shape[1])], dtype=t.int64, device=self.device)
        codes = (states * weights).sum(dim=1)
        for code in t.unique(codes).cpu().tolist():
            idx = (codes == code).nonzero().flatten()
            var = self._topology_variant([(code >> k) & 1 for k in range(states.shape[1])])
            n = int(idx.numel())
            x2 = self.x[idx].contiguous()
            tmp = {name: buf[idx].contiguous() for name, buf in b.items() if name != 'results'}
            tmp['results'] = t.full((n, var.n_results), float('nan'), dtype=t.float64, device=self.device)
            io = capi.StepIO()
            io.x = x2.data_ptr()
            act2 = action[idx].contiguous() if action is not None else None
            io.action = act2.data_ptr() if act2 is not None else None
            init2 = self.initial_obj[idx].contiguous() if with_initial_obj else None
            io.initial_obj = init2.data_ptr() if init2 is not None else None
            cnt2 = self.step_count[idx].contiguous() if self.steps_per_episode != 1 else None
            io.step_in_episode = cnt2.data_ptr() if cnt2 is not None else None
            io.outage = None
            for name, buf in tmp.items():
                setattr(io, name, buf.data_ptr())
            with t.cuda.device(self.device):
                # (the twin's own start: its topology decides whether it has a DC model; everything else is the parent's)
                capi.check(capi.lib().opfx_step(var._env_handle, n, C.byref(io), C.byref(var.solve_opts), mode,
                                                capi._stream()), 'opfx_step (topology variant)')
            self.x[idx] = x2
            for name, buf in tmp.items():
                if name != 'results':
                    b[name][idx] = buf
            mine, theirs = var._results_from
            rows = t.full((n, self.n_results), float('nan'), dtype=t.float64, device=self.device)
            rows[:, mine] = tmp['results'][:, theirs]
            b['results'][idx] = rows

    def _rescue_pivot_breakdown(self, action, mode, with_initial_obj, x_before):
        """Rows whose factorisation broke down (not converged, min_pivot < 1e-8) once more, on a plan that eliminates the
        buses named by `min_pivot_bus` last (see `on_pivot_breakdown`): gathered into a compact batch, stepped by a second
        environment object on that plan (same descriptor), scattered back.  GPU only — there is no CPU fallback."""
        t, b = self.torch, self.buf
        bad = (b['converged'] == 0) & (b['min_pivot'] < self.PIVOT_BREAKDOWN) & (b['min_pivot_bus'] >= 0)
        if not bool(bad.any()):                                            # (the one host synchronisation of this option)
            return
        idx = bad.nonzero().flatten()
        buses = tuple(sorted(set(b['min_pivot_bus'][idx].cpu().tolist())))
        if buses not in self._rescue_envs:
            self._drop_rescue_envs(keep=self.MAX_RESCUE_PLANS - 1)         # (each holds a plan, a context and a device copy)
            plan = capi.Plan(self.case, elim_last=buses, debug=self.debug)
            ctx = capi.Context(plan, self.device.index or 0, debug=self.debug)
            h = C.c_void_p()
            capi.check(capi.lib().opfx_env_create(ctx.handle, C.byref(self._env_desc), C.byref(h)), 'opfx_env_create (rescue plan)')
            self._rescue_envs[buses] = (plan, ctx, h)
        self._rescue_envs[buses] = self._rescue_envs.pop(buses)            # most recently used last
        _, _, handle = self._rescue_envs[buses]
        n = int(idx.numel())
        x2 = (x_before if x_before is not None else self.x)[idx].contiguous()
        tmp = {name: buf[idx].contiguous() for name, buf in b.items()}
        io = capi.StepIO()
        io.x = x2.data_ptr()
        act2 = action[idx].contiguous() if action is not None else None
        io.action = act2.data_ptr() if act2 is not None else None
        init2 = self.initial_obj[idx].contiguous() if with_initial_obj else None
        io.initial_obj = init2.data_ptr() if init2 is not None else None
        cnt2 = self.step_count[idx].contiguous() if self.steps_per_episode != 1 else None
        io.step_in_episode = cnt2.data_ptr() if cnt2 is not None else None
        io.outage = None
        for name, buf in tmp.items():
            setattr(io, name, buf.data_ptr())
        with t.cuda.device(self.device):
            capi.check(capi.lib().opfx_step(handle, n, C.byref(io), C.byref(self.solve_opts), mode, capi._stream()), 'opfx_step (rescue plan)')
        self.x[idx] = x2
        for name, buf in tmp.items():
            b[name][idx] = buf
        self.pivot_rescues += n
        self.pivot_rescues_recovered += int(tmp['converged'].sum().item())
