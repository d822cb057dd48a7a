import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def make_suspension(n, phi=None, L=None, seed=12345, fseed=54321, xy=0.0):
    """Synthetic random-sphere suspension of SURVEY.md 8(d): uniform positions in a cubic box centred on the
    origin (a = 1), N(0,1) forces with zero mean."""
    if L is None:
        L = (4.0 * np.pi * n / (3.0 * phi)) ** (1.0 / 3.0)
    rng = np.random.default_rng(seed)
    f = rng.uniform(0.0, 1.0, (n, 3)) - 0.5
    pos = np.empty((n, 3))
    pos[:, 1] = f[:, 1] * L
    pos[:, 2] = f[:, 2] * L
    pos[:, 0] = f[:, 0] * L + xy * pos[:, 1]
    force = np.random.default_rng(fseed).normal(size=(n, 3))
    force -= force.mean(axis=0)
    return pos, force, (L, L, L, xy)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pse_port
    pse_port.lib()
    return pse_port


# Trajectories of TWO engines compared over several steps (a team against the single GPU).  The deterministic part of a step is
# double precision end to end: 1e-9.  With kT > 0 the near-field operator inside the Lanczos iteration reads its pair coefficients in
# SINGLE precision (the per-step pair list, pse_kernels.hip nb_store): rounding is discontinuous, so positions that differ by 1e-16
# (summation order) now and then round a coefficient the other way, that particle's velocity moves by ~1e-8, its neighbours' pairs
# follow, and within ~15 steps the two trajectories differ by the noise floor of the rounded (single-precision-accurate) coefficients (~1e-7 per step and
# particle at dt = 0.25) -- a single evaluation from identical positions still agrees to 1e-9 and better (tested next to this).
# Once positions differ by 1e-8 a second discontinuity takes part: a pair within that distance of rcut is inside the cutoff in one run
# and outside in the other (N nbar / 2 x 3 delta / rcut pairs per step: ~0.2 % per step at N = 40 000), and its term there is what the
# method truncates -- ~1e-4 of a velocity, 7e-5 of a position at dt = 0.25 (tools/soak_local.py shows both stages at N = 1e6: 6e-9
# after 25 steps, 2.6e-7 after 200, 1.3e-4 after 225).  The bound for Brownian steps is therefore the size of a few such events,
# not the noise floor: what it still catches is a wrong exchange or a missing contribution (1e-2 and more).
TRAJ_TOL_DETERMINISTIC = 1e-9
TRAJ_TOL_BROWNIAN = 1e-3


def to4(a, w=0.0):
    import torch
    out = np.zeros((a.shape[0], 4))
    out[:, :3] = a
    out[:, 3] = w
    return torch.tensor(out, dtype=torch.float64, device="cuda")
