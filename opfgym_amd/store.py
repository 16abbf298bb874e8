"""The per-instance column store x[B, nx] and the reset programme's op builder (host side of `opfx_reset_desc`).

Split out of batched_env.py in round 6, no behaviour change.  `ColumnStore` hands out the slots of x: one
contiguous range per (table, column) of the reference net that is held per instance; `OpsBuilder` collects the `_sampling`
tails of the environments as vector ops on those slots (OPFX_OP_* in include/opfx.h)."""
from __future__ import annotations

import numpy as np

from . import capi

class ColumnStore:
    """Slot allocator for x: (table, column) -> contiguous range over the rows
    of that table, with the net's current values as the row template."""

    def __init__(self, net):
        self.net = net
        self.ranges = {}
        self.template = []
        self.n = 0
        self.dynamic = set()

    def slot(self, table, col, dynamic=False):
        key = (table, col)
        if key not in self.ranges:
            tbl = self.net[table]
            n = len(tbl)
            if col in tbl.columns:
                vals = np.array([float(v) if v is not None else np.nan
                                 for v in tbl[col].to_numpy()], dtype=float) if n else np.zeros(0)
            else:
                vals = np.zeros(n)
            self.ranges[key] = (self.n, n)
            self.template.append(vals)
            self.n += n
        if dynamic:
            self.dynamic.add(key)
        return self.ranges[key][0]

    def computed(self, table, col, values):
        """Slots of a column that the net's table does not hold as such (a quantity derived from several of its columns,
        constant per environment): the row template carries `values`."""
        key = (table, col)
        if key not in self.ranges:
            vals = np.asarray(values, dtype=float)
            assert len(vals) == len(self.net[table]), key
            self.ranges[key] = (self.n, len(vals))
            self.template.append(vals.copy())
            self.n += len(vals)
        return self.ranges[key][0]

    def rows(self, table, idxs):
        pos = self.net[table].index.get_indexer(np.asarray(idxs))
        if (pos < 0).any():
            raise KeyError(f'index not in net.{table}: {np.asarray(idxs)[pos < 0]}')
        return pos

    def slots(self, table, col, idxs, dynamic=False):
        return self.slot(table, col, dynamic) + self.rows(table, idxs)

    def row_template(self):
        return np.concatenate(self.template) if self.template else np.zeros(0)


class OpsBuilder:
    """Collects the `_sampling` tail of an environment as vector ops on x
    (see OPFX_OP_* in include/opfx.h)."""

    def __init__(self, store: ColumnStore):
        self.store = store
        self.ops = []          # (code, dst, a, n, c0, c1, c2) with c* numpy vectors or None
        self.n_uniform = 0
        self.uniform_runs = []     # (first column, count, source mask) of every uniform op
        self.n_normal = 0
        self.mode_mask = 7     # data sources under which the ops added next run ('mixed' sampling)

    def _emit(self, code, dst, a, c0=None, c1=None, c2=None):
        """dst/a: arrays of slots; split into runs where both are contiguous."""
        dst = np.asarray(dst, dtype=np.int64)
        a = np.asarray(a, dtype=np.int64)
        n = len(dst)
        if n == 0:
            return
        cut = np.flatnonzero((np.diff(dst) != 1) | (np.diff(a) != 1)) + 1
        for s, e in zip(np.r_[0, cut], np.r_[cut, n]):
            sl = slice(s, e)
            self.ops.append((code, int(dst[s]), int(a[s]), int(e - s),
                             None if c0 is None else np.broadcast_to(np.asarray(c0, float), (n,))[sl].copy(),
                             None if c1 is None else np.broadcast_to(np.asarray(c1, float), (n,))[sl].copy(),
                             None if c2 is None else np.broadcast_to(np.asarray(c2, float), (n,))[sl].copy(),
                             self.mode_mask))

    def _all(self, table, col, rows=None, dynamic=False):
        base = self.store.slot(table, col, dynamic)
        n = len(self.store.net[table])
        rows = np.arange(n) if rows is None else np.asarray(rows)
        return base + rows

    def set_const(self, table, col, values, rows=None):
        dst = self._all(table, col, rows, True)
        self._emit(capi.OP_SET_CONST, dst, dst, c0=values)

    def affine(self, table, col, src_col, c0, c1, rows=None):
        dst = self._all(table, col, rows, True)
        self._emit(capi.OP_AFFINE, dst, self._all(table, src_col, rows), c0=c0, c1=c1)

    def sqrt_diff(self, table, col, src_col, c0, rows=None):
        dst = self._all(table, col, rows, True)
        self._emit(capi.OP_SQRT_DIFF, dst, self._all(table, src_col, rows), c0=c0)

    def div(self, table, col, src_col, c0, rows=None):
        dst = self._all(table, col, rows, True)
        self._emit(capi.OP_DIV, dst, self._all(table, src_col, rows), c0=c0)

    def neg(self, table, col, src_col, rows=None):
        dst = self._all(table, col, rows, True)
        self._emit(capi.OP_NEG, dst, self._all(table, src_col, rows))

    def uniform(self, table, col, idxs, lo, hi, scale=1.0):
        """opf_env.py:266-284 `_sample_from_range`: one U[lo,hi] draw per row,
        divided by `scale`; consumes len(idxs) draws of the instance's draw
        vector, in order."""
        rows = self.store.rows(table, idxs)
        dst = self._all(table, col, rows, True)
        src = self.n_uniform + np.arange(len(rows))
        self.uniform_runs.append((self.n_uniform, len(rows), self.mode_mask))
        self.n_uniform += len(rows)
        self._emit(capi.OP_UNIFORM, dst, src, c0=lo, c1=hi, c2=scale)

    def uniform_columns(self, source: int):
        """Columns of the [B, n_uniform] draw matrix that a reset under data source `source` consumes, in
        the order the reference would draw them (the matrix has one fixed column per op whatever the
        source; the reference draws sequentially and only what the source needs)."""
        cols = [np.arange(s, s + n) for s, n, mask in self.uniform_runs if (mask >> source) & 1]
        return np.concatenate(cols) if cols else np.zeros(0, dtype=np.int64)


def _truncated_normal(ops, table, col, idxs, mean, scale, a, b):
    """opf_env.py:306-309: `scipy.stats.truncnorm.rvs(min_values, max_values, mean, std * diff)` per row.
    scipy reads its first two arguments as STANDARDISED bounds, so what the reference samples is
    mean + scale * Z with Z standard normal truncated to [min_values, max_values] (the raw numbers; defect
    D14, reproduced).  scipy draws from its own generator, which cannot be replayed; here Z comes from the
    instance's uniform draw u by the inverse CDF scipy itself applies to its uniforms, `truncnorm.ppf(u, a, b)`,
    as a device op that works in log space (OPFX_OP_TRUNCNORM): bounds like [10, 200] — every unit above ~8 MW —
    lie so far in the upper tail that Phi(a) == Phi(b) == 1.0 in double precision, and the plain
    Phi^-1(Phi(a) + u (Phi(b) - Phi(a))) returns +inf there."""
    n = len(np.asarray(a, dtype=float))
    ops.uniform(table, col, idxs, np.zeros(n), np.ones(n), 1.0)      # u itself, into the column
    rows = ops.store.rows(table, idxs)
    dst = ops._all(table, col, rows, True)
    ops._emit(capi.OP_TRUNCNORM, dst, dst, c0=np.asarray(a, dtype=float), c1=np.asarray(b, dtype=float))
    ops._emit(capi.OP_AFFINE, dst, dst, c0=np.asarray(scale, dtype=float), c1=np.asarray(mean, dtype=float))


def _normal_and_clip(ops, table, col, idxs, mean, std, lo, hi):
    """opf_env.py:311-315: N(mean, std) per row clipped to [lo, hi]; consumes len(idxs)
    standard-normal draws of the instance's draw vector, in order."""
    rows = ops.store.rows(table, idxs)
    dst = ops._all(table, col, rows, True)
    src = ops.n_normal + np.arange(len(rows))
    ops.n_normal += len(rows)
    ops._emit(capi.OP_NORMAL, dst, src, c0=mean, c1=std)
    ops._emit(capi.OP_CLIP, dst, dst, c0=lo, c1=hi)
