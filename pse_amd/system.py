"""Particle data + run loop: the minimum of HOOMD that `examples/run.py` touches (create_lattice, group.all,
mode_standard(dt), run).  Arrays live on the GPU in HOOMD's Scalar4 layout with Scalar = double."""
import math

import numpy as np

from . import context


class Group:
    """hoomd.group.all(): the index array handed to the integrator as d_group_members (PSEv1/Stokes.cc:461)."""

    def __init__(self, system, members=None):
        import torch
        self.system = system
        self.members = None if members is None else torch.as_tensor(np.asarray(members, dtype=np.int32), device="cuda")

    def __len__(self):
        return self.system.n if self.members is None else int(self.members.shape[0])


class System:
    def __init__(self, pos, box, dt=1e-3, mass=1.0):
        import torch
        pos = np.asarray(pos, dtype=np.float64)
        self.n = pos.shape[0]
        self.box = tuple(float(b) for b in box) + ((0.0,) if len(box) == 3 else ())
        self.dt = float(dt)
        self.timestep = 0
        p4 = np.zeros((self.n, 4)); p4[:, :3] = pos
        self.pos = torch.tensor(p4, dtype=torch.float64, device="cuda")
        v4 = np.zeros((self.n, 4)); v4[:, 3] = mass
        self.vel = torch.tensor(v4, dtype=torch.float64, device="cuda")
        self.net_force = torch.zeros((self.n, 4), dtype=torch.float64, device="cuda")
        self.accel = torch.zeros((self.n, 3), dtype=torch.float64, device="cuda")
        self.image = torch.zeros((self.n, 3), dtype=torch.int32, device="cuda")
        self.integrators = []
        self.forces = []          # force providers (pse_amd.forces): fill net_force before the integrators run
        self.analyzers = []       # e.g. pse_amd.dump.Trajectory
        self.box_tilt_variant = None    # a variant.shear_variant: Lees-Edwards box deformation
        context.current = self

    @classmethod
    def create_lattice_sc(cls, a, n, dt=1e-3):
        """hoomd.init.create_lattice(unitcell=hoomd.lattice.sc(a), n): n^3 particles on a simple-cubic lattice."""
        g = (np.arange(n) + 0.5) * a - 0.5 * n * a
        pos = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
        L = n * a
        return cls(pos, (L, L, L, 0.0), dt=dt)

    def all(self):
        return Group(self)

    def run(self, nsteps):
        for _ in range(int(nsteps)):
            if self.box_tilt_variant is not None:
                xy = self.box_tilt_variant.get_value(self.timestep)
                if xy != self.box[3]:
                    self._set_tilt(xy)
            for a in self.analyzers:
                a.analyze(self.timestep)
            if self.forces:
                self.net_force.zero_()
                for f in self.forces:
                    f.compute(self.timestep)
            for integ in self.integrators:
                integ.update(self.timestep)
            self.timestep += 1

    def _set_tilt(self, xy):
        # changing the tilt re-labels images: keep every particle inside the new primary cell
        Lx, Ly, Lz, _ = self.box
        self.box = (Lx, Ly, Lz, float(xy))
        import torch
        f = torch.floor((self.pos[:, 0] - xy * self.pos[:, 1]) / Lx + 0.5)
        self.pos[:, 0] -= f * Lx
        self.image[:, 0] += f.to(torch.int32)
        for integ in self.integrators:
            integ.set_box(self.box)
