import os, sys, math, statistics
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch
from conftest import make_suspension, to4
import pse_amd
n = 200000
pos, force, box = make_suspension(n, phi=0.1)
eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=9)
dpos, dF = to4(pos, 1.0), to4(force)
v = torch.zeros_like(dpos)
for busy in (False, True):
    ts = []
    for it in range(40):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if busy:
            eng.mobility(dpos, dF, vel=v)
        a.record(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print("busy" if busy else "idle", "empty event pair: median %.2f us, min %.2f, max %.2f" % (statistics.median(ts), ts[0], ts[-1]))
