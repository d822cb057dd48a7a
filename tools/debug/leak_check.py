import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import make_suspension, to4
import pse_amd
pos, force, box = make_suspension(20000, phi=0.1)
free0 = torch.cuda.mem_get_info()[0]
for it in range(int(os.environ.get("CYCLES", "40"))):
    eng = pse_amd.Engine(20000, box, xi=0.5, error=1e-3, seed=it)
    eng.brownian_velocity(to4(pos), to4(force), 1.0, 1e-3, it)
    if it % 3 == 0:
        from pse_amd.sharded import LoopbackSimulation
        sim = LoopbackSimulation(20000, box, 2, xi=0.5, error=1e-3, seed=1); sim.load(pos, force); sim.mobility(); sim.team.close()
        for e in sim.engines: e.close()
    eng.close()
torch.cuda.synchronize(); torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
print("free before %.1f MB after %.1f MB, delta %.1f MB" % (free0/1e6, free1/1e6, (free0-free1)/1e6))
assert free0 - free1 < 400e6
print("ok")
