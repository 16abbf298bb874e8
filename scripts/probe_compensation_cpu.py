"""CPU experiment: iterations of a fixed-Jacobian (compensated) contingency solve on the 372-bus N-1 stand-in.
For sampled instances: base case Newton -> x0; for each contingency k: (a) full Newton from x0 (what the kernel does today),
(b) simplified Newton with the Jacobian of the CONTINGENCY system frozen at x0 (= J0 + rank-4 update, what Woodbury on the
base-case LU gives), (c) the same with the BASE-CASE Jacobian alone (no compensation).  Counts iterations to |F|inf < 1e-8."""
import sys, numpy as np, scipy.sparse as sp
from scipy.sparse.linalg import splu
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from env_cases import oracle_env, product_env
from oracle import pf_oracle as po, pd2ppc

env = product_env('sc_vc_hv_urban', defer_device=True)
orc = oracle_env('sc_vc_hv_urban', env)
rng = np.random.default_rng(0)
steps = rng.choice(env.train_steps, 3)

def jac(ybus, v, pv, pq):
    ib = ybus @ v
    dV = sp.diags(v); dI = sp.diags(ib); dVn = sp.diags(v / np.abs(v))
    dS_dVa = 1j * dV @ np.conj(dI - ybus @ dV)
    dS_dVm = dV @ np.conj(ybus @ dVn) + np.conj(dI) @ dVn
    pvpq = np.r_[pv, pq]
    J = sp.bmat([[dS_dVa[pvpq][:, pvpq].real, dS_dVm[pvpq][:, pq].real],
                 [dS_dVa[pq][:, pvpq].imag, dS_dVm[pq][:, pq].imag]], format='csc')
    return J

def mism(ybus, sbus, v, pv, pq):
    s = v * np.conj(ybus @ v) - sbus
    return np.r_[s[np.r_[pv, pq]].real, s[pq].imag]

def update(v, dx, pv, pq):
    va, vm = np.angle(v), np.abs(v)
    npv, npq = len(pv), len(pq)
    pvpq = np.r_[pv, pq]
    va[pvpq] += dx[:npv + npq]; vm[pq] += dx[npv + npq:]
    return vm * np.exp(1j * va)

res = {'newton': [], 'frozen_k': [], 'frozen_0': [], 'f0': []}
for s in (steps if "--fixed-jacobian" in sys.argv else []):
    orc.reset(int(s))
    a = rng.random(env.n_actions)
    net = orc.net
    # apply action through the oracle env, but run the power flows here
    from oracle import env_oracle
    env_oracle.apply_actions(net, orc.act_keys, a, autoscale=True)
    ppc = pd2ppc.build_ppc(net)
    sol = po.solve(ppc, enforce_q_lims=True)
    assert sol['converged']
    v0 = sol['V']; bt = sol['bus_type']
    pv = np.flatnonzero(bt == po.PV); pq = np.flatnonzero(bt == po.PQ)
    sbus = po.make_sbus(ppc)
    ybus0 = po.make_ybus(ppc, ppc.status)
    lines = [k for k in range(ppc.nbr) if ppc.br_table[k] == 'line']
    cont = env.contingencies if hasattr(env, 'contingencies') else None
    J0 = splu(jac(ybus0, v0, pv, pq))
    n_done = 0
    for k in lines:
        st = ppc.status.copy(); st[k] = 0
        if not pd2ppc.supplied_buses(ppc, st).all():
            continue
        yk = po.make_ybus(ppc, st)
        f0 = np.abs(mism(yk, sbus, v0, pv, pq)).max()
        # (a) full Newton
        v = v0.copy(); it = 0
        while np.abs(mism(yk, sbus, v, pv, pq)).max() >= 1e-8 and it < 30:
            dx = splu(jac(yk, v, pv, pq)).solve(-mism(yk, sbus, v, pv, pq)); v = update(v, dx, pv, pq); it += 1
        res['newton'].append(it)
        # (b) frozen contingency Jacobian at x0
        Jk = splu(jac(yk, v0, pv, pq)); v = v0.copy(); it = 0
        while np.abs(mism(yk, sbus, v, pv, pq)).max() >= 1e-8 and it < 60:
            v = update(v, Jk.solve(-mism(yk, sbus, v, pv, pq)), pv, pq); it += 1
        res['frozen_k'].append(it)
        # (c) frozen base-case Jacobian
        v = v0.copy(); it = 0
        while np.abs(mism(yk, sbus, v, pv, pq)).max() >= 1e-8 and it < 200:
            v = update(v, J0.solve(-mism(yk, sbus, v, pv, pq)), pv, pq); it += 1
            if not np.isfinite(np.abs(v).max()): it = 999; break
        res['frozen_0'].append(it)
        res['f0'].append(f0)
        n_done += 1
    print('instance', s, 'contingencies', n_done, flush=True)
for k in (("newton", "frozen_k", "frozen_0") if res["newton"] else ()):
    a = np.array(res[k])
    print(k, 'mean', a.mean().round(2), 'median', np.median(a), 'p90', np.percentile(a, 90), 'max', a.max(), 'hist', np.bincount(np.minimum(a, 20)).tolist())
if res['f0']: print('initial mismatch: median', np.median(res['f0']), 'max', np.max(res['f0']))


# ---- (d) full Newton from the base case's solution with the ANGLES moved by the DC (LODF) prediction of the outage ------------
def dc_predictor_experiment():
    """theta_start = theta_base + alpha w,  w = B'^-1 (e_f - e_t),  alpha = b (theta_f - theta_t) / (1 - b (w_f - w_t)) with the base
    case's AC angles: the rank-1 update of the DC power flow applied as a PREDICTOR to the AC solution (|V| kept)."""
    out = {'newton': [], 'predicted': [], 'f0': [], 'f0_pred': []}
    for s in steps:
        orc.reset(int(s))
        a = np.random.default_rng(int(s)).random(env.n_actions)
        net = orc.net
        env_oracle.apply_actions(net, orc.act_keys, a, autoscale=True)
        ppc = pd2ppc.build_ppc(net)
        sol = po.solve(ppc, enforce_q_lims=True)
        v0 = sol['V']; bt = sol['bus_type']
        pv = np.flatnonzero(bt == po.PV); pq = np.flatnonzero(bt == po.PQ)
        free = np.r_[pv, pq]; free.sort()
        sbus = po.make_sbus(ppc)
        bdc = (ppc.status > 0) / ppc.x / ppc.tap
        nl, nb = ppc.nbr, ppc.nb
        cft = sp.csr_matrix((np.r_[np.ones(nl), -np.ones(nl)], (np.r_[np.arange(nl), np.arange(nl)], np.r_[ppc.f, ppc.t])), (nl, nb))
        bbus = (cft.T @ sp.diags(bdc) @ cft).tocsc()
        lu = splu(bbus[free][:, free].tocsc())
        posf = -np.ones(nb, int); posf[free] = np.arange(len(free))
        for k in range(nl):
            if ppc.br_table[k] != 'line':
                continue
            st = ppc.status.copy(); st[k] = 0
            if not pd2ppc.supplied_buses(ppc, st).all():
                continue
            yk = po.make_ybus(ppc, st)
            f, t, b = int(ppc.f[k]), int(ppc.t[k]), bdc[k]
            u = np.zeros(len(free))
            if posf[f] >= 0: u[posf[f]] += 1.0
            if posf[t] >= 0: u[posf[t]] -= 1.0
            w = lu.solve(u)
            th = np.angle(v0)
            alpha = b * (th[f] - th[t]) / (1.0 - b * (u @ w))
            va = th.copy(); va[free] += alpha * w
            v_pred = np.abs(v0) * np.exp(1j * va)
            for key, vs in (('newton', v0), ('predicted', v_pred)):
                v = vs.copy(); it = 0
                while np.abs(mism(yk, sbus, v, pv, pq)).max() >= 1e-8 and it < 30:
                    dx = splu(jac(yk, v, pv, pq)).solve(-mism(yk, sbus, v, pv, pq)); v = update(v, dx, pv, pq); it += 1
                out[key].append(it)
            out['f0'].append(np.abs(mism(yk, sbus, v0, pv, pq)).max())
            out['f0_pred'].append(np.abs(mism(yk, sbus, v_pred, pv, pq)).max())
    for key in ('newton', 'predicted'):
        a = np.array(out[key])
        print('DC predictor:', key, 'mean', a.mean().round(3), 'hist', np.bincount(a).tolist())
    print('initial mismatch: base-case start median', np.median(out['f0']).round(3), 'max', np.max(out['f0']).round(2),
          '| predicted start median', np.median(out['f0_pred']).round(3), 'max', np.max(out['f0_pred']).round(2))


from oracle import env_oracle  # noqa: E402
dc_predictor_experiment()
