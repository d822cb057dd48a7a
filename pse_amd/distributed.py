"""Simulation drivers over the C-ABI: single GPU, and the multi-GPU sharding of the PSE step.

`make_simulation(world=1)` returns `Simulation` (one handle, everything on one GPU).
"""
import numpy as np

from .engine import Engine


def _to4(a, w=0.0):
    import torch
    out = np.zeros((a.shape[0], 4))
    out[:, :3] = a
    out[:, 3] = w
    return torch.tensor(out, dtype=torch.float64, device="cuda")


class Simulation:
    """Single-GPU suspension state + engine: what HOOMD's ParticleData + Stokes hold together."""

    def __init__(self, n, box, **kw):
        self.n = n
        self.engine = Engine(n, box, **kw)

    def describe(self):
        return "1 GPU"

    def load(self, pos, force, mass=1.0):
        import torch
        self.pos = _to4(pos, 0.0)
        self.force = _to4(force, 0.0)
        self.vel = _to4(np.zeros((self.n, 3)), mass)
        self.accel = torch.zeros((self.n, 3), dtype=torch.float64, device="cuda")
        self.image = torch.zeros((self.n, 3), dtype=torch.int32, device="cuda")

    def info(self):
        return self.engine.info()

    def set_timing(self, on):
        self.engine.set_timing(on)

    def phase_times(self):
        return {k: v for k, v in self.engine.info().items() if k.startswith("t_")}

    def mobility(self):
        return self.engine.mobility(self.pos, self.force, vel=self.vel)

    def step(self, kT, dt, timestep, shear_rate=0.0, lanczos_m=2):
        return self.engine.step(self.pos, self.vel, self.accel, self.image, self.force, kT, dt, timestep,
                                shear_rate=shear_rate, lanczos_m=lanczos_m)


def make_simulation(n, box, world=1, rank=0, **kw):
    import os
    if world == 1 and not os.environ.get("PSE_FORCE_SHARDED"):   # the env var runs the RCCL driver with one rank (tests)
        return Simulation(n, box, **kw)
    from .sharded import ShardedSimulation
    return ShardedSimulation(n, box, world=world, rank=rank, **kw)
