"""The reference's examples/run.py (1000 spheres on a simple-cubic lattice, L = 64, oscillatory shear, kT = 1,
xi = 0.5, error = 1e-3) written against this package.  HOOMD calls map as:
  hoomd.init.create_lattice(unitcell=hoomd.lattice.sc(a), n)  -> System.create_lattice_sc(a, n, dt)
  hoomd.md.integrate.mode_standard(dt)                        -> the dt of the System
  hoomd.PSEv1.shear_function.sine / integrate.PSEv1 / hoomd.run -> same names below
"""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pse_amd import integrate, shear_function, variant
from pse_amd.system import System

dt = 1e-3
tf = 1e0
nrun = int(os.environ.get("PSE_EXAMPLE_STEPS", tf / dt))

N = 1000
L = 64
n = math.ceil(N ** (1.0 / 3.0))
a = L / n

system = System.create_lattice_sc(a=a, n=n, dt=dt)
function_form = shear_function.sine(dt=dt, shear_rate=1.0, shear_freq=1.0)
# Lees-Edwards: deform the box with the wrapped strain of the same function (the shipped script applies the shear
# velocity but never tilts the box)
system.box_tilt_variant = variant.shear_variant(function_form, nrun, max_strain=0.5)
pse = integrate.PSEv1(group=system.all(), seed=1, T=1.0, xi=0.5, error=1e-3, function_form=function_form)
system.run(nrun)
print("ran", nrun, "steps; Lanczos vectors in the last step:", pse.cpp_method.lanczosIterations())
