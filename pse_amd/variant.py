"""Mirror of PSEv1/variant.py:15-32: the wrapped-strain variant that drives the box tilt under shear.
(The reference file raises NameError on construction -- it calls `_variant.__init__` without importing it,
PSEv1/variant.py:24 -- so only its intent can be mirrored.)"""
from . import _PSEv1, context


class shear_variant:
    def __init__(self, function_form, total_timestep, max_strain=0.5):
        if total_timestep <= 0:
            context.msg.error("Cannot create a shear_variant with 0 or negative points\n")
            raise RuntimeError("Error creating variant")
        self._keep = function_form
        self.cpp_variant = _PSEv1.VariantShearFunction(function_form.cpp_function, int(total_timestep), -max_strain, max_strain)

    def get_value(self, timestep):
        return self.cpp_variant.getValue(int(timestep))
