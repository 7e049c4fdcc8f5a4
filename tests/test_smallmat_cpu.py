"""CPU tests of the host-side l x l dense algebra used by the PCA driver (smallmat.hpp)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def sm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("sm") / "libsm.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out,
                           os.path.join(HERE, "cpp", "smallmat_harness.cpp")])
    return C.CDLL(out)


def P(a):
    return C.c_void_p(a.ctypes.data)


@pytest.mark.parametrize("n", [1, 2, 5, 64, 96])
def test_sym_eig_desc(sm, n):
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n))
    A = B @ B.T + np.diag(rng.uniform(0, 3, n))
    if n >= 5:
        A[2, :] = A[:, 2] = 0.0            # a zero row/col: exercises the scale == 0 branch
    w = np.zeros(n)
    V = np.zeros((n, n))
    assert sm.t_sym_eig_desc(n, P(np.ascontiguousarray(A)), P(w), P(V)) == 0
    ref = np.linalg.eigvalsh(A)[::-1]
    np.testing.assert_allclose(w, ref, rtol=1e-12, atol=1e-12 * max(1.0, abs(ref).max()))
    np.testing.assert_allclose(V.T @ V, np.eye(n), atol=1e-12)
    np.testing.assert_allclose(A @ V, V * w[None, :], atol=1e-10 * max(1.0, abs(ref).max()))
    assert np.all(np.diff(w) <= 1e-12)


def test_sym_eig_clustered(sm):
    n = 64
    rng = np.random.default_rng(1)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.concatenate([np.full(10, 5.0), 5.0 + 1e-9 * np.arange(10), np.linspace(1, 2, n - 20)])
    A = (Q * lam[None, :]) @ Q.T
    w = np.zeros(n)
    V = np.zeros((n, n))
    assert sm.t_sym_eig_desc(n, P(np.ascontiguousarray(A)), P(w), P(V)) == 0
    np.testing.assert_allclose(np.sort(w), np.sort(lam), atol=1e-12)
    np.testing.assert_allclose(V.T @ V, np.eye(n), atol=1e-12)


@pytest.mark.parametrize("n,ld", [(1, 1), (7, 7), (40, 64), (64, 64)])
def test_chol_upper_inverse(sm, n, ld):
    rng = np.random.default_rng(7)
    B = rng.standard_normal((n + 5, n))
    G = np.zeros((ld, ld))
    G[:n, :n] = B.T @ B
    Rinv = np.full((ld, ld), 9.0)
    assert sm.t_chol_upper_inverse(n, ld, P(G), P(Rinv)) == 0
    R = np.linalg.cholesky(G[:n, :n]).T
    np.testing.assert_allclose(Rinv[:n, :n], np.linalg.inv(R), rtol=1e-9, atol=1e-11)
    assert np.all(np.tril(Rinv[:n, :n], -1) == 0) and np.all(Rinv[n:, :] == 0) and np.all(Rinv[:, n:] == 0)
    W = B @ Rinv[:n, :n]
    np.testing.assert_allclose(W.T @ W, np.eye(n), atol=1e-10)


def test_chol_rejects_indefinite(sm):
    G = np.array([[1.0, 2.0], [2.0, 1.0]])
    assert sm.t_chol_upper_inverse(2, 2, P(G), P(np.zeros((2, 2)))) == 1
