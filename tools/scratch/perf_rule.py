"""The metric point with the reference's own parameter rule (xi = 0.5 -> 360^3 grid) instead of the 256^3 override."""
import numpy as np, math, sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from conftest import make_suspension, to4
import pse_amd, torch
n, phi = 1000000, 0.1
pos, force, box = make_suspension(n, phi=phi)
eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=1)
i = eng.info(); print({k: i[k] for k in ('Nx', 'Ny', 'Nz', 'P', 'rcut', 'eta')})
dpos, dF = to4(pos, 1.0), to4(force); vel = to4(np.zeros((n,3)), 1.0)
accel = torch.zeros((n,3), dtype=torch.float64, device='cuda'); image = torch.zeros((n,3), dtype=torch.int32, device='cuda')
m = 2
for it in range(3): m = eng.step(dpos, vel, accel, image, dF, 1.0, 1e-3, it, lanczos_m=m)
eng.set_timing(True)
m = eng.step(dpos, vel, accel, image, dF, 1.0, 1e-3, 9, lanczos_m=m)
print({k: round(v,3) for k,v in eng.info().items() if k.startswith('t_') and v>0}, 'm', m)
eng.set_timing(False)
torch.cuda.synchronize(); t0=time.time()
for it in range(10): m = eng.step(dpos, vel, accel, image, dF, 1.0, 1e-3, 20+it, lanczos_m=m)
torch.cuda.synchronize(); t=(time.time()-t0)/10
print('reference-rule grid: %.3f ms/step -> %.1f steps/s' % (t*1e3, 1/t))
