"""Force providers: what fills `net_force` before the integrator consumes it (HOOMD's force computes in the reference,
PSEv1/Stokes.cc:447; its example script has none).  SURVEY.md 8 f4: the step either side of the hot path, kept minimal."""


class HarmonicRepulsion:
    """F_i = sum_j k (sigma - r)(r_i - r_j)/r for minimum-image pairs with r < sigma (sigma = 2a: contact of unit spheres).
    Evaluated on the integrator's own cell list; sigma must not exceed the hydrodynamic real-space cutoff."""

    def __init__(self, integrator, k, sigma=2.0):
        self.integrator, self.k, self.sigma = integrator, float(k), float(sigma)
        integrator.system.forces.append(self)

    def compute(self, timestep):
        s, g = self.integrator.system, self.integrator.group
        m = g.members
        self.integrator.cpp_method.pairRepulsion(s.pos.data_ptr(), s.net_force.data_ptr(), 0 if m is None else m.data_ptr(),
                                                 len(g), self.k, self.sigma, True)


class ConstantForce:
    """The same force on every particle of the group (gravity / sedimentation); its mean is what the k = 0 mode drops."""

    def __init__(self, system, fx=0.0, fy=0.0, fz=0.0):
        import torch
        self.system = system
        self.f = torch.tensor([fx, fy, fz, 0.0], dtype=torch.float64, device="cuda")
        system.forces.append(self)

    def compute(self, timestep):
        self.system.net_force += self.f
