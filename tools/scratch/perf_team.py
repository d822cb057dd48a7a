import numpy as np, math, sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from conftest import make_suspension
import pse_amd, torch
from pse_amd.sharded import LoopbackSimulation
n, phi, grid, err = 1000000, 0.1, 256, 1e-3
pos, force, box = make_suspension(n, phi=phi)
L = box[0]; xi = math.pi*grid/(2*L*math.sqrt(-math.log(err)))
for G in (1, 2, 4, 8):
    sim = LoopbackSimulation(n, box, G, xi=xi, error=err, seed=1, grid=(grid,)*3)
    sim.load(pos, force)
    m = 2
    for it in range(3): m = sim.step(1.0, 1e-3, it, lanczos_m=m)
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(5): m = sim.step(1.0, 1e-3, 10+it, lanczos_m=m)
    torch.cuda.synchronize(); t = (time.time()-t0)/5
    for e in sim.engines: e.set_timing(True)
    m = sim.step(1.0, 1e-3, 20, lanczos_m=m)
    i = sim.engines[0].info()
    print('G=%d  %.3f ms/step (all ranks on one GPU)  per-rank %.3f ms  m=%s  rank0 phases:' % (G, t*1e3, t*1e3/G, m), {k: round(v,3) for k,v in i.items() if k.startswith('t_') and v>0})
    del sim
    torch.cuda.empty_cache()
