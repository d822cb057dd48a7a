"""The oracle's Lanczos iteration held to the reference's DRIVER (tests/golden/reference_lanczos.json: the text of
gpu_stokes_BrealLanczos_wrap, PSEv1/Brownian.cu:357-765, and of the helper kernels it launches, executed by
tests/golden/make_lanczos_fixture.py on dense positive definite operators; inputs and results only).

What this pins that the dense square root cannot: WHEN the iteration stops -- m_in - 1 vectors first, one more per pass, the
first m >= m_in whose step norm sqrt(|u_m - u_{m-1}|^2 / (psi.M psi / |psi|^2)) is within the tolerance -- hence the number of
near-field mat-vecs of every Brownian step, which the device reproduces vector for vector (tests/test_gpu_parity.py: equal m).
One deliberate difference, stated in the port and checked here: at a breakdown (|v| < 1e-8: psi lies in an invariant subspace)
the reference returns the PREVIOUS approximation and reports m - 1 (Brownian.cu:503-506,655-658), the port keeps the vector
that completes the subspace -- its answer is then exact."""
import json
import math
import os

import numpy as np
import pytest
import scipy.linalg

from oracle import pse_port as oracle

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_lanczos.json")) as f:
    CASES = json.load(f)["cases"]


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b))


@pytest.mark.parametrize("c", [c for c in CASES if "breakdown" not in c["name"]], ids=lambda c: c["name"])
def test_lanczos_driver(c):
    M, psi = np.array(c["M"]), np.array(c["psi"])
    calls = [0]

    def mv(v):
        calls[0] += 1
        return (M @ v.ravel()).reshape(v.shape)
    u, m = oracle.lanczos_sqrt(mv, psi, m_in=c["m_in"], tol=c["tol"])
    assert m == c["m"], (m, c["m"])
    # (rounding differences grow with the lost orthogonality of a plain Lanczos basis: 4.5e-9 at 17 vectors and condition 400)
    assert rel(math.sqrt(2.0 * c["T"] / c["dt"]) * u, c["vel"]) < (1e-9 if c["m"] <= 12 else 1e-7)
    # the reference spends one more product on psi.M psi (Brownian.cu:448); the port reads it off alpha_0 -- same number
    assert calls[0] == c["matvecs"] - 1
    # and the iteration stopped where the tolerance says, not later: against the dense square root
    exact = (scipy.linalg.sqrtm(M).real @ psi.ravel()).reshape(psi.shape)
    assert rel(u, exact) < 30 * c["tol"]


def test_breakdown_rule_is_the_one_deliberate_difference():
    c = [c for c in CASES if "breakdown" in c["name"]][0]
    M, psi = np.array(c["M"]), np.array(c["psi"])
    exact = math.sqrt(2.0 * c["T"] / c["dt"]) * (scipy.linalg.sqrtm(M).real @ psi.ravel()).reshape(psi.shape)
    u, m = oracle.lanczos_sqrt(lambda v: (M @ v.ravel()).reshape(v.shape), psi, m_in=c["m_in"], tol=c["tol"])
    assert m == c["m"] + 1 == 3                                       # three eigenvectors span psi
    assert rel(math.sqrt(2.0 * c["T"] / c["dt"]) * u, exact) < 1e-10
    assert rel(c["vel"], exact) > 1e-3                                # the reference stops one vector short
