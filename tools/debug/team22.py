import sys, os, math
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import to4
import pse_amd
from pse_amd.sharded import LoopbackSimulation
os.environ["PSE_WAVE_MODE"] = "slab"

def rel(a, b): return np.linalg.norm(a - b) / np.linalg.norm(b)

def run(world, grid, err, xy=0.0, n=3000, hh=0.8):
    rng = np.random.default_rng(1)
    s = math.sqrt(-math.log(err))
    box = tuple(grid[a] * hh for a in range(3)) + (xy,)
    xi = 0.9 * math.pi / (2 * hh * s)
    f = rng.uniform(-0.5, 0.5, (n, 3)); pos = np.empty((n, 3)); pos[:, 1] = f[:, 1] * box[1]; pos[:, 2] = f[:, 2] * box[2]; pos[:, 0] = f[:, 0] * box[0] + xy * pos[:, 1]
    force = rng.normal(size=(n, 3))
    kw = dict(xi=xi, error=err, seed=3, grid=grid)
    ref = pse_amd.Engine(n, box, **kw)
    i = ref.info()
    try:
        sim = LoopbackSimulation(n, box, world, **kw)
    except pse_amd.PSEError as e:
        print(world, grid, err, "refused:", e); return
    sim.load(pos, force)
    out = []
    for parts in (2, 1):
        u = ref.mobility(to4(pos), to4(force), parts=parts).cpu().numpy()[:, :3]
        out.append([float("%.1e" % rel(v.cpu().numpy()[:, :3], u)) for v in sim.mobility(parts=parts)])
    print("world", world, "grid", grid, "err", err, "P", i["P"], "xy", xy, "ncell_x", i["ncell_x"], "wave", out[0], "real", out[1], flush=True)

run(4, (64, 24, 24), 1e-6)
run(2, (64, 24, 24), 1e-6)
run(4, (64, 32, 32), 1e-6)
run(4, (64, 24, 24), 1e-3)
run(4, (64, 24, 24), 1e-5)
run(4, (96, 24, 24), 1e-6)
run(4, (128, 24, 24), 1e-6)
run(8, (128, 32, 32), 1e-6)
run(8, (128, 24, 24), 1e-6)
run(4, (64, 24, 24), 1e-6, xy=0.3)
