"""The port held to the two pieces of the reference's step the kernel fixture had left to restatement (tests/golden/
reference_driver.json.gz, written by tests/golden/make_driver_fixture.py from the reference's text; inputs and results only):

  * the host wrapper gpu_stokes_CombinedMobilityBrownian_wrap (PSEv1/Brownian.cu:772-923): order of the kernels with and without
    temperature and the constants the host hands them (prefac, expfac, the gather weight quadW prefac, the (1, 1) sums);
  * the gather kernel gpu_stokes_Contract_kernel (PSEv1/Mobility.cu:325-477), run with real barriers for its tree reduction;
  * the particle-noise kernel gpu_stokes_BrownianGenerate_kernel (PSEv1/Brownian.cu:99-130)."""
import gzip
import json
import math
import os

import numpy as np
import pytest

from oracle import pse_port as oracle

HERE = os.path.dirname(os.path.abspath(__file__))
with gzip.open(os.path.join(HERE, "golden", "reference_driver.json.gz")) as f:
    FIX = json.load(f)

PI_AS_WRITTEN = 3.1415926536          # the literal of PSEv1/Brownian.cu:828 (a deliberate difference: the build uses exact pi)


@pytest.mark.parametrize("w", FIX["wrapper"], ids=lambda w: f"P{w['P']}-T{w['T']}")
def test_host_wrapper_order_and_constants(w):
    names = [c[0] for c in w["calls"]]
    wave = ["ZeroGrid"] * 3 + ["Spread"] + ["cufftExecC2C"] * 3 + ["Green"]
    tail = ["cufftExecC2C"] * 3 + ["Contract", "Mreal", "LinearCombination"]
    if w["T"] > 0:     # noise enters between the Green operator and the inverse transform; M_real^{1/2} psi is added last
        assert names == ["BrownianGenerate"] + wave + ["BrownianGridGenerate"] + tail + ["BrealLanczos", "LinearCombination"]
    else:
        assert names == ["BrownianGenerate"] + wave + tail
    calls = {c[0]: c[1:] for c in w["calls"]}
    assert [c[1] for c in w["calls"] if c[0] == "cufftExecC2C"] == ["forward"] * 3 + ["inverse"] * 3
    c = 2.0 * w["xi"] ** 2 / w["eta"]
    P, prefac, expfac = calls["Spread"]
    assert P == w["P"] and abs(expfac - c) < 1e-15 * c
    assert abs(prefac - (c / PI_AS_WRITTEN) ** 1.5) < 1e-15 * prefac              # as written ...
    assert abs(prefac - (c / math.pi) ** 1.5) < 2e-11 * prefac                    # ... and what exact pi changes
    quadW = w["gridh"][0] * w["gridh"][1] * w["gridh"][2]
    assert abs(calls["Contract"][1] - quadW * prefac) < 1e-15 * prefac and calls["Contract"][2] == expfac   # Brownian.cu:872
    assert all(c[1:] == [1.0, 1.0] for c in w["calls"] if c[0] == "LinearCombination")                     # parts are plainly added
    if w["T"] > 0:
        assert calls["BrownianGridGenerate"] == [w["T"], 1e-3, quadW] and calls["BrealLanczos"] == [1e-3, w["T"], 1e-3]
    # the port's weights use the same two numbers: a particle sitting exactly on a node gives prefac at that node
    p = dict(grid=(8, 8, 8), xi=w["xi"], eta=w["eta"], P=w["P"] if w["P"] <= 8 else 8, h=tuple(w["gridh"]))
    box = tuple(8 * h for h in w["gridh"]) + (0.0,)
    node = np.array([[-0.5 * box[0] + 3 * p["h"][0], -0.5 * box[1] + 4 * p["h"][1], -0.5 * box[2] + 2 * p["h"][2]]])
    wts, _ = oracle._weights(node, box, p)
    assert abs(wts.max() - (c / math.pi) ** 1.5) < 1e-14 * prefac


@pytest.mark.parametrize("c", FIX["contract"], ids=lambda c: f"P{c['P']}-" + "x".join(map(str, c["grid"])))
def test_gather_against_the_contract_kernel(c):
    grid, box = tuple(c["grid"]), tuple(c["box"])
    p = dict(grid=grid, xi=c["xi"], eta=c["eta"], P=c["P"], h=tuple(box[a] / grid[a] for a in range(3)))
    u = oracle.gather(np.array(c["ugrid"]), np.array(c["pos"]), box, p)
    ref = np.array(c["vel"])
    assert np.abs(u - ref[:, :3]).max() < 1e-13 * np.abs(ref[:, :3]).max()
    assert np.all(ref[:, 3] == 7.0)                                               # d_vel.w is kept (Mobility.cu:473-475)


def test_particle_noise_kernel():
    """gpu_stokes_BrownianGenerate_kernel (PSEv1/Brownian.cu:99-130): psi of a particle is keyed by its GLOBAL index (not its place in
    the group) and by timestep + seed, uniform on (-sqrt 3, sqrt 3) (the reference's literal 1.73205080757: 3e-12 off), .w kept,
    particles outside the group untouched."""
    c = FIX["psi"]
    got = np.array(c["psi"])
    want = oracle.psi_particles(c["n_total"], c["seed"], c["timestep"])
    mem = np.array(c["members"])
    assert np.abs(got[mem, :3] - want[mem]).max() < 1e-11
    others = np.setdiff1d(np.arange(c["n_total"]), mem)
    assert np.all(got[others, :3] == -9.0) and np.all(got[:, 3] == 4.5)
