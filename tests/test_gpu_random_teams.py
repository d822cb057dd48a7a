"""Slab teams on configurations nobody picked by hand: seeded random non-cubic sheared boxes, support sizes 4..13, two to four
ranks, both far-field modes -- every rank of the in-process team against the single-GPU engine (itself held to the oracle on the same
kind of configurations by tests/test_gpu_random_configs.py)."""
import math

import numpy as np
import pytest

from conftest import to4

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


SPECIAL = {
    16: (4, (240, 40, 32), 1e-3, "slab"),        # mixed-radix x pass above 200 points (two kz columns per workgroup) on the transposed layout
    17: (8, (256, 32, 40), 1e-3, "slab"),        # eight ranks, radix-16 x pass
    18: (3, (96, 36, 45), 1e-4, "slab"),         # odd Nz: the gather leaves the binned kernel on a slab window
    19: (2, (64, 24, 24), 1e-6, "slab"),         # P = 13 on 24 nodes: spread and gather by the generic kernels
    20: (4, (128, 48, 32), 1e-5, "slab"),        # P = 11 under shear: two + six halo planes of a 32-plane slab
    21: (4, (120, 40, 40), 1e-3, "slab"),        # 120 = 8 5 3
    22: (4, (64, 24, 24), 1e-6, "slab"),
    23: (2, (72, 36, 32), 1e-5, "slab"),         # two ranks forced to slabs at P = 11
}


def config(seed):
    rng = np.random.default_rng(9000 + seed)
    world = [2, 3, 4][seed % 3]
    err = [1e-3, 1e-4, 1e-5, 1e-6][seed % 4]
    s = math.sqrt(-math.log(err))
    nx = int(rng.choice([48, 60, 72, 96])) if world != 4 else int(rng.choice([64, 80, 96]))     # slabs of whole planes, wider than a support
    grid = (nx, int(rng.choice([36, 48] if world == 3 else [32, 36, 40, 48])), int(rng.choice([32, 36, 40, 48])))   # Nx, Ny: multiples of the rank count
    mode = [None, "slab", "replicated"][(seed // 3) % 3]
    if seed in SPECIAL:                                                 # shapes the draw above cannot produce
        world, grid, err, mode = SPECIAL[seed]
        s = math.sqrt(-math.log(err))
    h = rng.uniform(0.65, 0.9, 3)
    box = tuple(float(grid[a] * h[a]) for a in range(3)) + (float(rng.uniform(-0.45, 0.45)) if seed % 2 or seed == 20 else 0.0,)
    xi = float(0.9 * min(math.pi / (2.0 * h[a] * s) for a in range(3)))
    n = int(rng.integers(1500, 5000))
    f = rng.uniform(-0.5, 0.5, (n, 3))
    if seed % 5 == 0:
        f[: n // 2, 0] = rng.uniform(-0.5, -0.2, n // 2)                # crowd one end: unequal row blocks
    pos = np.empty((n, 3))
    pos[:, 1] = f[:, 1] * box[1]
    pos[:, 2] = f[:, 2] * box[2]
    pos[:, 0] = f[:, 0] * box[0] + box[3] * pos[:, 1]
    return dict(world=world, err=err, grid=grid, box=box, xi=xi, pos=pos, force=rng.normal(size=(n, 3)), n=n, mode=mode,
                seed=int(rng.integers(1, 2 ** 31)))


@pytest.mark.parametrize("seed", range(24))
def test_random_team(seed, monkeypatch):
    import pse_amd
    from pse_amd.sharded import LoopbackSimulation
    c = config(seed)
    if c["mode"]:
        monkeypatch.setenv("PSE_WAVE_MODE", c["mode"])
    else:
        monkeypatch.delenv("PSE_WAVE_MODE", raising=False)
    kw = dict(xi=c["xi"], error=c["err"], seed=c["seed"], grid=c["grid"])
    ref = pse_amd.Engine(c["n"], c["box"], **kw)
    i = ref.info()
    if i["ncell_x"] < c["world"]:
        pytest.skip("fewer cell layers than ranks")
    try:
        sim = LoopbackSimulation(c["n"], c["box"], c["world"], **kw)
    except pse_amd.PSEError as e:                                       # a refused decomposition must say why
        assert "slab" in str(e) or "cell" in str(e) or "planes" in str(e), str(e)
        pytest.skip(f"decomposition refused: {e}")
    sim.load(c["pos"], c["force"])
    pos, force = to4(c["pos"]), to4(c["force"])
    for parts in (2, 1, 3):
        u_ref = ref.mobility(pos, force, parts=parts).cpu().numpy()[:, :3]
        for r, v in enumerate(sim.mobility(parts=parts)):
            assert rel(v.cpu().numpy()[:, :3], u_ref) < 1e-11, (seed, parts, r, i["P"], rel(v.cpu().numpy()[:, :3], u_ref))
    v_ref, m_ref = ref.brownian_velocity(pos, force, 0.8, 1e-3, 3 + seed)
    vels, m = sim.brownian_velocity(0.8, 1e-3, 3 + seed)
    assert m == m_ref
    for r in range(c["world"]):
        assert rel(vels[r].cpu().numpy()[:, :3], v_ref.cpu().numpy()[:, :3]) < 1e-10, (seed, r)
