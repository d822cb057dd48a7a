"""hoomd.md.integrate.mode_standard(dt): the time step of the run (read by PSEv1 when it is constructed)."""
from pse_amd import context as _ctx


class mode_standard:
    def __init__(self, dt):
        if _ctx.current is None:
            raise RuntimeError("mode_standard before hoomd.init.create_lattice")
        self.dt = float(dt)
        _ctx.current.dt = self.dt
