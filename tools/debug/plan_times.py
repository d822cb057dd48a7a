import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import pse_amd
from conftest import make_suspension
grids = [(256, 32, 32), (360, 32, 32), (512, 32, 32), (240, 32, 32), (64, 48, 40), (32, 360, 36), (36, 256, 30), (32, 512, 256), (32, 256, 512), (32, 36, 360), (36, 36, 180), (32, 40, 270)]
for g in grids:
    pos, force, box = make_suspension(1200, L=24.0, xy=0.1)
    t0 = time.time()
    e = pse_amd.Engine(1200, box, xi=0.5, error=1e-3, seed=1, grid=g, P=4)
    t1 = time.time()
    print(os.environ.get("TAG"), g, "create %.2f s" % (t1 - t0), flush=True)
    del e
