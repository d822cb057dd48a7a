"""`hoomd` stand-in for the reference's example script: only the calls `examples/run.py` of
stochasticHydroTools/PSE makes, mapped onto this package, so that script runs UNCHANGED with

    PYTHONPATH=<repo>/compat python /path/to/PSE/examples/run.py

  hoomd.context.initialize('')                                   -> no-op (the GPU context is torch's)
  hoomd.init.create_lattice(unitcell=hoomd.lattice.sc(a=a), n=n) -> pse_amd.system.System.create_lattice_sc
  hoomd.md.integrate.mode_standard(dt=dt)                         -> the time step of the current System
  hoomd.group.all()                                               -> System.all()
  hoomd.PSEv1.integrate / shear_function / variant                -> pse_amd.integrate / shear_function / variant
  hoomd.run(nsteps)                                               -> System.run
  hoomd._hoomd, hoomd.md._md                                      -> empty modules (the script imports them, nothing more)

It is not HOOMD: no force fields, no file output, no other integrators.  Set PSE_EXAMPLE_STEPS to shorten hoomd.run().
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from pse_amd import context as _pse_context          # noqa: E402
from . import _hoomd, context, group, init, lattice, md   # noqa: E402,F401


def run(tsteps):
    """hoomd.run: advance the current system; like HOOMD the argument may be a float (the script passes tf / dt)."""
    if _pse_context.current is None:
        raise RuntimeError("hoomd.run before hoomd.init.create_lattice")
    n = int(os.environ.get("PSE_EXAMPLE_STEPS", round(float(tsteps))))
    _pse_context.current.run(n)
