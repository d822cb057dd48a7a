import numpy as np, math, sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from conftest import make_suspension, to4
import pse_amd, torch
n, phi, grid, err = 1000000, 0.1, 256, 1e-3
pos, force, box = make_suspension(n, phi=phi)
L = box[0]; xi = math.pi*grid/(2*L*math.sqrt(-math.log(err)))
eng = pse_amd.Engine(n, box, xi=xi, error=err, seed=1, grid=(grid,)*3)
dpos, dF = to4(pos, 1.0), to4(force); vel = to4(np.zeros((n,3)), 1.0)
eng.set_timing(True)
acc = {}
m = 2
for it in range(6):
    _, m = eng.brownian_velocity(dpos, dF, 1.0, 1e-3, it, vel=vel, lanczos_m=m)
    if it >= 2:
        for k, v in eng.info().items():
            if k.startswith('t_'): acc[k] = acc.get(k, 0) + v/4
print('tile variant', os.environ.get('PSE_SPREAD_TILE'), {k: round(v,3) for k,v in acc.items() if v>0}, 'm', m)
