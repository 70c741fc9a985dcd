"""Minimal stand-in for the ``cvxopt`` names used by the reference's ``compute_cn_lr``
(/root/reference/src/breakpoint_graph.py:495-606).

TEST INFRASTRUCTURE ONLY.  cvxopt is not installed and cannot be fetched, so the reference's CN step
is run against this stub when golden fixtures are generated.  ``solvers.cp`` here is NOT cvxopt's
interior-point method: it minimises the same objective under the same equality constraints with
scipy's trust-constr followed by a null-space Newton polish, to ~1e-12.  Consequently CN parity
against real cvxopt is **unpinned**; goldens pin CN against this independent solver only.
"""
from __future__ import annotations

import types

import numpy as np


class matrix:
    def __init__(self, data, size=None, tc="d"):
        if isinstance(data, matrix):
            self.a = data.a.copy()
        elif size is not None and np.isscalar(data):
            self.a = np.full(size, float(data))
        else:
            arr = np.array(data, dtype=float)
            if arr.ndim == 1:
                arr = arr.reshape(-1, 1)
            self.a = arr
        self.size = self.a.shape

    # element access: linear index in column-major order as cvxopt does
    def __getitem__(self, i):
        return float(self.a.reshape(-1, order="F")[i])

    def __len__(self):
        return self.a.size

    def __iter__(self):
        return iter(self.a.reshape(-1, order="F").tolist())

    def _bin(self, o, f):
        ob = o.a if isinstance(o, matrix) else o
        return matrix(f(self.a, ob))

    def __add__(self, o): return self._bin(o, np.add)
    __radd__ = __add__
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return matrix((o.a if isinstance(o, matrix) else o) - self.a)
    def __mul__(self, o): return self._bin(o, np.multiply) if np.isscalar(o) else matrix(self.a @ o.a)
    def __rmul__(self, o): return matrix(o * self.a)
    def __neg__(self): return matrix(-self.a)
    def __pow__(self, p): return matrix(self.a ** p)

    @property
    def T(self):
        return matrix(self.a.T)


def mul(a, b):
    return matrix(a.a * b.a)


def log(x):
    return matrix(np.log(x.a))


def spdiag(x):
    return matrix(np.diag(x.a.reshape(-1)))


def _dot(a, b):
    return float(np.sum(a.a.reshape(-1) * b.a.reshape(-1)))


modeling = types.ModuleType("cvxopt.modeling")
modeling.dot = _dot


def _cp(F, G=None, h=None, dims=None, A=None, b=None, kktsolver=None, options=None):
    from scipy.optimize import minimize, LinearConstraint, Bounds
    from scipy.linalg import null_space
    _, x0 = F()
    n = len(x0)
    Am = A.a if A is not None else np.zeros((0, n))

    def fun(x):
        r = F(matrix(x))
        if r is None:
            return 1e300, np.zeros(n)
        f, Df = r
        return float(f), Df.a.reshape(-1)

    def hess(x):
        r = F(matrix(x), matrix([1.0]))
        return r[2].a

    x = np.ones(n)
    cons = [LinearConstraint(Am, 0.0, 0.0)] if Am.shape[0] else []
    res = minimize(fun, x, jac=True, hess=hess, method="trust-constr", constraints=cons,
                   bounds=Bounds(1e-9, np.inf), options=dict(gtol=1e-12, xtol=1e-14, maxiter=5000))
    x = np.maximum(res.x, 1e-9)
    # null-space Newton polish on the exactly feasible manifold
    N = null_space(Am) if Am.shape[0] else np.eye(n)
    if Am.shape[0]:
        x = N @ (N.T @ x)          # project onto A x = 0
    for _ in range(200):
        f, g = fun(x)
        H = hess(x)
        gz = N.T @ g
        Hz = N.T @ H @ N
        try:
            dz = -np.linalg.solve(Hz, gz)
        except np.linalg.LinAlgError:
            break
        dx = N @ dz
        t = 1.0
        while np.min(x + t * dx) <= 0 or fun(x + t * dx)[0] > f + 1e-4 * t * float(g @ dx):
            t *= 0.5
            if t < 1e-12:
                break
        if t < 1e-12:
            break
        x = x + t * dx
        if np.max(np.abs(dx) / np.maximum(np.abs(x), 1e-300)) < 1e-15:
            break
    f, g = fun(x)
    return {"status": "optimal", "x": matrix(x), "primal objective": f, "dual objective": f, "gap": 0.0,
            "relative gap": 0.0, "primal infeasibility": float(np.max(np.abs(Am @ x))) if Am.shape[0] else 0.0,
            "dual infeasibility": 0.0}


solvers = types.ModuleType("cvxopt.solvers")
solvers.cp = _cp
solvers.options = {}
