"""A physical run on top of the path: 20 000 soft-repulsive spheres at phi = 0.2 under oscillatory Lees-Edwards shear with
Brownian motion (box tilt follows the wrapped strain, PSEv1/VariantShearFunction.cc:34-43), 1000 steps; prints a health
line per 100 steps (finite positions, particles inside the sheared cell, Lanczos vectors, largest force)."""
import numpy as np, math, sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pse_amd import integrate, shear_function, variant, forces
from pse_amd.system import System
rng = np.random.default_rng(5)
n, phi = 20000, 0.2
L = (4*math.pi*n/(3*phi))**(1/3)
pos = rng.uniform(-L/2, L/2, size=(n,3))
s = System(pos, (L,L,L,0.0), dt=1e-3)
ff = shear_function.sine(dt=1e-3, shear_rate=1.0, shear_freq=1.0)
s.box_tilt_variant = variant.shear_variant(ff, 2000, max_strain=0.5)
pse = integrate.PSEv1(group=s.all(), T=1.0, seed=11, xi=0.5, error=1e-3, function_form=ff)
forces.HarmonicRepulsion(pse, k=200.0, sigma=2.0)
t0=time.time()
for blk in range(10):
    s.run(100)
    p = s.pos[:, :3]
    ok = bool(torch.isfinite(p).all())
    fx = (p[:,0] - s.box[3]*p[:,1])/L
    print(blk, 'finite', ok, 'max|frac|', float(fx.abs().max()), float((p[:,1]/L).abs().max()), 'xy', round(s.box[3],4), 'm', pse.cpp_method.lanczosIterations(), 'maxF', float(s.net_force[:,:3].abs().max()))
torch.cuda.synchronize(); print('1000 steps in %.2f s' % (time.time()-t0))
