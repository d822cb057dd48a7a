"""Mirror of the reference's integrator UI (PSEv1/integrate.py:15-123): `PSEv1(group, T, seed, xi, error,
function_form, max_strain, nlist_type)`, `.set_params`, `.stop_shear`, backed by the C++ `Stokes` class over the C-ABI."""
import math

from . import _PSEv1, context, shear_function


class _const_or_variant:
    @staticmethod
    def make(T):
        return T.cpp_variant if hasattr(T, "cpp_variant") else _PSEv1.VariantConst(float(T))


class PSEv1:
    def __init__(self, group, T, seed=0, xi=0.5, error=0.001, function_form=None, max_strain=0.5, nlist_type="cell",
                 grid=None, P=0, rcut=0.0):
        self.group = group
        self.system = group.system
        # real-space cutoff from the error estimate of the spectral sums (integrate.py:47, Stokes.cc:135)
        self.rcut = math.sqrt(-math.log(error)) / xi
        if nlist_type.upper() not in ("CELL", "TREE", "STENCIL"):                       # integrate.py:58-78
            context.msg.error("Invalid neighborlist method specified. Valid options are: cell, tree, stencil. \n")
            raise RuntimeError("Error constructing neighborlist")
        # every nlist_type maps to the engine's own cell list: the three HOOMD builders produce the same neighbour set
        s = self.system
        self._T = _const_or_variant.make(T)
        self.cpp_method = _PSEv1.Stokes(s.n, s.box[0], s.box[1], s.box[2], s.box[3], self._T, int(seed), float(xi),
                                        float(error), s.dt)                                # integrate.py:86
        if function_form is not None:                                                   # integrate.py:90-94
            self._shear = function_form
        else:
            self._shear = shear_function.steady(dt=0)
        self.cpp_method.setShear(self._shear.cpp_function, max_strain)
        if grid is not None or P or rcut:
            g = grid or (0, 0, 0)
            self.cpp_method.setOverrides(g[0], g[1], g[2], int(P), float(rcut))
        self.cpp_method.setParams()                                                     # integrate.py:96
        s.integrators.append(self)

    def set_params(self, T=None, function_form=None, max_strain=0.5):                  # integrate.py:108-118
        if T is not None:
            self._T = _const_or_variant.make(T)
            self.cpp_method.setT(self._T)
        if function_form is not None:
            self._shear = function_form
            self.cpp_method.setShear(function_form.cpp_function, max_strain)

    def stop_shear(self, max_strain=0.5):                                              # integrate.py:121-123
        self._shear = shear_function.steady(dt=0)
        self.cpp_method.setShear(self._shear.cpp_function, max_strain)

    def set_box(self, box):
        self.cpp_method.setBox(*box)

    def update(self, timestep):
        s = self.system
        g = self.group.members
        n = len(self.group)
        self.cpp_method.integrateStepOne(int(timestep), s.pos.data_ptr(), s.vel.data_ptr(), s.accel.data_ptr(),
                                         s.image.data_ptr(), s.net_force.data_ptr(),
                                         0 if g is None else g.data_ptr(), n)
