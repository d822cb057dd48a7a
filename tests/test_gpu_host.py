"""GPU tests of the host layer: the C++ Stokes class (module _PSEv1) and the mirrored Python UI drive the same C-ABI
step as the raw engine, and the example script runs."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import make_suspension, to4

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ui_run_matches_raw_engine_and_oracle(oracle):
    import torch
    import pse_amd
    from pse_amd import integrate, shear_function, variant
    from pse_amd.system import System
    n = 1000
    pos, _, box = make_suspension(n, phi=0.1)
    dt, seed = 1e-2, 3
    system = System(pos, box, dt=dt)
    f = shear_function.sine(dt=dt, shear_rate=2.0, shear_freq=5.0)
    system.box_tilt_variant = variant.shear_variant(f, 100, max_strain=0.5)
    integ = integrate.PSEv1(group=system.all(), T=1.0, seed=seed, xi=0.5, error=1e-3, function_form=f)
    assert abs(integ.rcut - oracle.select_params(box, 0.5, 1e-3)["rcut"]) < 1e-12

    # the same three steps through the raw engine, with the box tilt and shear rate taken from the oracle's formulas
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=oracle.hash_seed(seed))
    rp, rv = to4(pos), to4(np.zeros((n, 3)), 1.0)
    rF = torch.zeros((n, 4), dtype=torch.float64, device="cuda")
    ra = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); ri = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    sf = oracle.SinShear(2.0, 5.0, 0, dt)
    m = 2
    cur_xy = 0.0
    for t in range(3):
        xy = oracle.variant_value(sf, t, 100, -0.5, 0.5)
        if xy != cur_xy:
            fl = torch.floor((rp[:, 0] - xy * rp[:, 1]) / box[0] + 0.5)
            rp[:, 0] -= fl * box[0]; ri[:, 0] += fl.to(torch.int32)
            eng.set_box(box[0], box[1], box[2], xy); cur_xy = xy
        m = eng.step(rp, rv, ra, ri, rF, 1.0, dt, t, shear_rate=sf.shear_rate(t), lanczos_m=m)
    system.run(3)
    assert system.timestep == 3
    assert np.abs(system.pos.cpu().numpy() - rp.cpu().numpy()).max() < 1e-12
    assert np.array_equal(system.image.cpu().numpy(), ri.cpu().numpy())
    assert integ.cpp_method.lanczosIterations() == m
    assert abs(system.box[3] - oracle.variant_value(sf, 2, 100, -0.5, 0.5)) < 1e-15


def test_set_params_and_stop_shear():
    from pse_amd import integrate, shear_function
    from pse_amd.system import System
    system = System.create_lattice_sc(a=6.4, n=6, dt=1e-3)
    integ = integrate.PSEv1(group=system.all(), T=0.0, seed=1, xi=0.5, error=1e-3,
                            function_form=shear_function.steady(dt=1e-3, shear_rate=1.0))
    p0 = system.pos.clone()
    system.run(1)
    moved = (system.pos - p0)[:, 0].abs().max().item()
    assert moved > 1e-3                    # pure affine shear (kT = 0, no forces): dx = rate * y * dt
    integ.stop_shear()
    p1 = system.pos.clone()
    system.run(1)
    assert (system.pos - p1).abs().max().item() == 0.0
    integ.set_params(T=1.0)
    system.run(1)
    assert (system.pos - p1).abs().max().item() > 0.0


def test_example_script_runs():
    env = dict(os.environ, PSE_EXAMPLE_STEPS="5")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "run.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ran 5 steps" in r.stdout


def test_hoomd_style_script_runs_on_the_shim():
    """compat/hoomd: the calls of the reference's example script (import hoomd, create_lattice, mode_standard, group.all,
    PSEv1.integrate.PSEv1, hoomd.run) on the stand-in package."""
    env = dict(os.environ)
    env.pop("PSE_EXAMPLE_STEPS", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "run_hoomd_api.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "hoomd-style run done: 20 steps" in r.stdout


def test_shim_driven_run_matches_raw_engine(oracle):
    """SURVEY 8 f3 with a result: the calls of the reference's example script made on the `hoomd` stand-in (lattice, mode_standard,
    group.all, PSEv1.integrate.PSEv1 with a steady shear, hoomd.run), then the same steps through the raw C-ABI engine --
    positions, image flags and the Lanczos count must agree."""
    import math
    import torch
    import pse_amd
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    import hoomd
    import hoomd.PSEv1
    from pse_amd import context as pctx
    hoomd.context.initialize('')
    dt, nrun, n1, a = 1e-3, 5, 7, 5.0
    hoomd.init.create_lattice(unitcell=hoomd.lattice.sc(a=a), n=n1)
    system = pctx.current
    p0 = system.pos.cpu().numpy()[:, :3].copy()
    n, box = p0.shape[0], tuple(system.box)
    assert n == n1 ** 3 and abs(box[0] - n1 * a) < 1e-12 and box[3] == 0.0
    function_form = hoomd.PSEv1.shear_function.steady(dt=dt, shear_rate=0.5)
    hoomd.md.integrate.mode_standard(dt=dt)
    pse = hoomd.PSEv1.integrate.PSEv1(group=hoomd.group.all(), seed=1, T=1.0, xi=0.5, error=1E-3, function_form=function_form)
    os.environ.pop("PSE_EXAMPLE_STEPS", None)
    hoomd.run(nrun)
    # the raw engine: no box_resize in the script, so the box stays untilted and the steady rate enters the integration only
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=oracle.hash_seed(1))
    rp, rv = to4(p0), to4(np.zeros((n, 3)), 1.0)
    rF = torch.zeros((n, 4), dtype=torch.float64, device="cuda")
    ra = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); ri = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    m = 2
    for t in range(nrun):
        m = eng.step(rp, rv, ra, ri, rF, 1.0, dt, t, shear_rate=0.5, lanczos_m=m)
    assert system.timestep == nrun
    assert np.abs(system.pos.cpu().numpy()[:, :3] - rp.cpu().numpy()[:, :3]).max() < 1e-12
    assert np.array_equal(system.image.cpu().numpy(), ri.cpu().numpy())
    assert pse.cpp_method.lanczosIterations() == m
    assert np.abs(system.pos.cpu().numpy()[:, :3] - p0).max() > 1e-3     # it did move


def test_force_provider_and_trajectory(tmp_path):
    """The steps either side of the path: a soft-repulsion force provider feeding net_force and a
    trajectory writer.  Overlapping random spheres at kT = 0 are pushed apart by M.F_repulsion; frames are written."""
    import numpy as np
    import torch
    from pse_amd import dump, forces, integrate
    from pse_amd.system import System
    rng = np.random.default_rng(2)
    n, L = 800, 24.0
    pos = rng.uniform(-L / 2, L / 2, size=(n, 3))
    s = System(pos, (L, L, L, 0.0), dt=2e-3)
    pse = integrate.PSEv1(group=s.all(), T=0.0, seed=3, xi=0.5, error=1e-3)
    forces.HarmonicRepulsion(pse, k=50.0, sigma=2.0)
    traj = dump.Trajectory(s, str(tmp_path / "traj.npz"), period=10)

    def overlap_energy():
        p = s.pos[:, :3]
        d = p[:, None, :] - p[None, :, :]
        d -= L * torch.round(d / L)
        r = d.norm(dim=2) + 10.0 * torch.eye(n, device="cuda", dtype=torch.float64)
        return float((torch.clamp(2.0 - r, min=0.0) ** 2).sum())

    e0 = overlap_energy()
    s.run(40)
    e1 = overlap_energy()
    assert e0 > 0.0 and e1 < 0.5 * e0, (e0, e1)
    assert float(s.net_force[:, :3].abs().max()) > 0.0
    fr = dump.load(traj.write())
    assert list(fr["timestep"]) == [0, 10, 20, 30] and fr["position"].shape == (4, n, 3) and fr["box"].shape == (4, 4)
    assert np.abs(fr["position"][0] - pos).max() < 1e-12
