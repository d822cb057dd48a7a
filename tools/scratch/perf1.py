import numpy as np, math, sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from conftest import make_suspension, to4
import pse_amd, torch
def run(n, phi, grid, err=1e-3, kT=1.0, steps=5):
    pos, force, box = make_suspension(n, phi=phi)
    L = box[0]
    xi = math.pi*grid/(2*L*math.sqrt(-math.log(err)))   # xi from the grid
    t0=time.time()
    eng = pse_amd.Engine(n, box, xi=xi, error=err, seed=1, grid=(grid,)*3)
    print('create %.2fs'%(time.time()-t0), {k:v for k,v in eng.info().items() if k in ('Nx','P','rcut','eta','ncell_x','device_bytes')})
    dpos, dF = to4(pos, 1.0), to4(force)
    vel = to4(np.zeros((n,3)), 1.0)
    eng.set_timing(True)
    for it in range(3):
        eng.mobility(dpos, dF, vel=vel)
    i = eng.info(); print('M.F phases ms:', {k: round(v,3) for k,v in i.items() if k.startswith('t_') and v>0})
    m = 2
    for it in range(3):
        _, m = eng.brownian_velocity(dpos, dF, kT, 1e-3, it, vel=vel, lanczos_m=m)
    i = eng.info(); print('Brownian phases ms:', {k: round(v,3) for k,v in i.items() if k.startswith('t_') and v>0}, 'm', m, i['lanczos_matvecs'])
    eng.set_timing(False)
    torch.cuda.synchronize(); t0=time.time()
    for it in range(steps): eng.mobility(dpos, dF, vel=vel)
    torch.cuda.synchronize(); t=(time.time()-t0)/steps
    print('M.F  %.3f ms/eval  -> %.1f evals/s'%(t*1e3, 1/t))
    accel = torch.zeros((n,3), dtype=torch.float64, device='cuda'); image = torch.zeros((n,3), dtype=torch.int32, device='cuda')
    torch.cuda.synchronize(); t0=time.time()
    for it in range(steps): m = eng.step(dpos, vel, accel, image, dF, kT, 1e-3, 100+it, lanczos_m=m)
    torch.cuda.synchronize(); t=(time.time()-t0)/steps
    print('step %.3f ms/step -> %.1f steps/s, %.3g particle-steps/s, m=%d'%(t*1e3, 1/t, n/t, m))
run(65536, 0.1, 64)
run(1000000, 0.1, 256)
