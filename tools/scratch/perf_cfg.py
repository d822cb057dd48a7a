"""BASELINE configs 2, 3 and 5 on one GPU (config 5 = config 3 under oscillatory shear, tilt xy = 0.3 here)."""
import numpy as np, math, sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from conftest import make_suspension, to4
import pse_amd, torch
def run(name, n, phi, grid, kT, xy=0.0, err=1e-3):
    pos, force, box = make_suspension(n, phi=phi, xy=xy)
    xi = math.pi*grid/(2*box[0]*math.sqrt(-math.log(err)))
    eng = pse_amd.Engine(n, box, xi=xi, error=err, seed=1, grid=(grid,)*3)
    dpos, dF = to4(pos, 1.0), to4(force); vel = to4(np.zeros((n,3)), 1.0)
    accel = torch.zeros((n,3), dtype=torch.float64, device='cuda'); image = torch.zeros((n,3), dtype=torch.int32, device='cuda')
    for it in range(3): eng.mobility(dpos, dF, vel=vel)
    torch.cuda.synchronize(); t0=time.time()
    for it in range(10): eng.mobility(dpos, dF, vel=vel)
    torch.cuda.synchronize(); tm=(time.time()-t0)/10
    out = '%s: N=%d phi=%.2f grid %d^3 rcut %.2f  M.F %.3f ms (%.0f evals/s)' % (name, n, phi, grid, eng.info()['rcut'], tm*1e3, 1/tm)
    if kT > 0:
        m = 2
        for it in range(3): m = eng.step(dpos, vel, accel, image, dF, kT, 1e-3, it, shear_rate=1.0 if xy else 0.0, lanczos_m=m)
        torch.cuda.synchronize(); t0=time.time()
        for it in range(10): m = eng.step(dpos, vel, accel, image, dF, kT, 1e-3, 10+it, shear_rate=1.0 if xy else 0.0, lanczos_m=m)
        torch.cuda.synchronize(); t=(time.time()-t0)/10
        out += '  step %.3f ms (%.1f steps/s, %.3g particle-steps/s, m=%d)' % (t*1e3, 1/t, n/t, m)
    print(out)
    del eng; torch.cuda.empty_cache()
run('config 2', 65536, 0.10, 64, 0.0)
run('config 3', 1048576, 0.20, 256, 1.0)
run('config 5', 1048576, 0.20, 256, 1.0, xy=0.3)
