"""A script in the reference's own idiom (`import hoomd`, `hoomd.PSEv1.integrate.PSEv1`, `hoomd.run`) running on the
`hoomd` stand-in of compat/ -- the same calls the reference's examples/run.py makes (which runs unchanged with
PYTHONPATH=compat), here with steady shear on a smaller lattice."""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "compat"))
import hoomd
from hoomd import _hoomd      # noqa: F401
from hoomd.md import _md      # noqa: F401
import hoomd.PSEv1

hoomd.context.initialize('')
dt = 1e-3
nrun = 20
N, L = 512, 48
n = math.ceil(N ** (1.0 / 3.0))
hoomd.init.create_lattice(unitcell=hoomd.lattice.sc(a=L / n), n=n)
function_form = hoomd.PSEv1.shear_function.steady(dt=dt, shear_rate=0.5)
hoomd.md.integrate.mode_standard(dt=dt)
pse = hoomd.PSEv1.integrate.PSEv1(group=hoomd.group.all(), seed=1, T=1.0, xi=0.5, error=1E-3, function_form=function_form)
hoomd.run(nrun)
print("hoomd-style run done:", nrun, "steps, Lanczos vectors", pse.cpp_method.lanczosIterations())
