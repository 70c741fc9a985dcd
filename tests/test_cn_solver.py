"""coral_cn_solve (the Newton iteration of the CN assignment in libcoral_hip) against the numpy iteration it mirrors and against
the optimality conditions themselves, on chain-shaped balance problems like the ones compute_cn_lr builds (bg:495-606)."""
import numpy as np
import pytest

from coral_amd import breakpoint_graph as bg


def _problem(rng, n_seq, unsupported=0):
    """A path of sequence edges joined by concordant edges, some discordant edges between random nodes: variables = edges,
    one balance row per interior node (seq edge in, concordant / discordant edges out)."""
    nodes = 2 * n_seq
    n_conc = n_seq - 1
    n_disc = int(rng.integers(1, max(2, n_seq // 2)))
    n = n_seq + n_conc + n_disc
    A = np.zeros((nodes, n))
    for k in range(n_seq):
        A[2 * k, k] = 1
        A[2 * k + 1, k] = 1
    for k in range(n_conc):
        A[2 * k + 1, n_seq + k] = -1
        A[2 * k + 2, n_seq + k] = -1
    for k in range(n_disc):
        a, b = rng.integers(1, nodes - 1, 2)
        A[a, n_seq + n_conc + k] = -1
        A[b, n_seq + n_conc + k] = -1                 # (a == b: assigned, not accumulated, as the reference does)
    A = A[1:-1]                                       # the two end nodes have no balance row
    cov = 3.0
    length = rng.integers(1000, 200000, n_seq).astype(float)
    depth = rng.uniform(2, 80, n_seq)
    w_lin, w_log, w_inv = np.zeros(n), np.zeros(n), np.zeros(n)
    w_lin[:n_seq] = 0.5 * cov * length
    w_log[:n_seq] = -0.5
    w_inv[:n_seq] = 0.5 * (depth * length) ** 2 / (cov * length)
    w_lin[n_seq:] = cov
    w_log[n_seq:] = rng.integers(1, 200, n - n_seq).astype(float)
    if unsupported:
        w_log[n_seq:n_seq + unsupported] = 0.0        # concordant edges nobody supports: h == 0 there
    return w_inv, w_lin, w_log, A


@pytest.mark.parametrize("unsupported", [0, 2])
def test_native_newton_equals_the_numpy_iteration(unsupported, monkeypatch):
    rng = np.random.default_rng(7 + unsupported)
    for trial in range(25):
        w = _problem(rng, int(rng.integers(2, 40)), unsupported)
        monkeypatch.setenv("CORAL_CN_SOLVER", "python")
        want = bg.solve_cn_lr(*w)
        res_py = bg.solve_cn_lr.last_residual
        monkeypatch.setenv("CORAL_CN_SOLVER", "native")
        got = bg.solve_cn_lr(*w)
        assert bg.solve_cn_lr.last_residual < 1e-8 and res_py < 1e-8
        assert np.allclose(got, want, rtol=1e-9, atol=1e-12), trial
        assert (got > 0).all()


def test_native_solver_reports_a_singular_system_instead_of_guessing():
    """Dependent balance rows (the caller normally removes them): status 1, the general numpy path takes over."""
    import ctypes as C
    from coral_amd import _lib
    w_inv, w_lin, w_log, A = _problem(np.random.default_rng(1), 6)
    A = np.vstack([A, A[0]])
    x, nu, it = np.empty(len(w_lin)), np.empty(A.shape[0]), C.c_int32(0)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    a, b, c, d = f(w_inv), f(w_lin), f(w_log), f(A)
    rc = _lib.lib().coral_cn_solve(len(w_lin), A.shape[0], a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data, 200, x.ctypes.data,
                                   nu.ctypes.data, C.byref(it))
    assert rc == 1
    assert _lib.lib().coral_cn_solve(0, 0, None, None, None, None, 10, None, None, None) < 0


def test_native_row_selection_equals_the_numpy_one(monkeypatch):
    rng = np.random.default_rng(3)
    for trial in range(40):
        w = _problem(rng, int(rng.integers(2, 30)))
        A = w[3]
        extra = [A[int(rng.integers(0, len(A)))] for _ in range(int(rng.integers(0, 4)))]          # duplicated rows
        extra += [A[0] + A[1]] if len(A) > 1 else []                                               # a dependent combination
        B = np.vstack([A] + extra) if extra else A
        B = B[rng.permutation(len(B))]
        monkeypatch.setenv("CORAL_CN_SOLVER", "python")
        want = bg._independent_rows(B)
        monkeypatch.setenv("CORAL_CN_SOLVER", "native")
        assert bg._independent_rows(B) == want and len(want) == np.linalg.matrix_rank(B)
