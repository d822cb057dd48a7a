"""Mirror of the reference's Python shear-function UI (PSEv1/shear_function.py:10-114): same class names, arguments,
validation messages and accessors, backed by the C++ classes of the `_PSEv1` module."""
from . import _PSEv1, context


class _shear_function:
    def __init__(self, zero="now"):                                        # shear_function.py:15-28
        self.cpp_function = None
        if zero == "now":
            self._offset = context.current_timestep()
        else:
            if zero < 0:
                context.msg.error("Cannot create a shear_function variant with a negative zero\n")
                raise RuntimeError("Error creating shear function")
            if zero > context.current_timestep():
                context.msg.error("Cannot create a shear_function variant with a zero in the future\n")
                raise RuntimeError("Error creating shear function")
            self._offset = zero

    def get_shear_rate(self, timestep):                                    # shear_function.py:30-40
        return self.cpp_function.getShearRate(timestep)

    def get_strain(self, timestep):
        return self.cpp_function.getStrain(timestep)

    def get_offset(self):
        return self.cpp_function.getOffset()


class steady(_shear_function):                                             # shear_function.py:44-52
    def __init__(self, dt, shear_rate=0, zero="now"):
        _shear_function.__init__(self, zero)
        self.cpp_function = _PSEv1.SteadyShearFunction(shear_rate, self._offset, dt)


class sine(_shear_function):                                               # shear_function.py:54-72
    def __init__(self, dt, shear_rate, shear_freq, zero="now"):
        if shear_rate <= 0:
            context.msg.error("Shear rate must be positive (use steady class instead for zero shear)\n")
            raise RuntimeError("Error creating shear function")
        if shear_freq <= 0:
            context.msg.error("Shear frequency must be positive (use steady class instead for steady shear)\n")
            raise RuntimeError("Error creating shear function")
        _shear_function.__init__(self, zero)
        self.cpp_function = _PSEv1.SinShearFunction(shear_rate, shear_freq, self._offset, dt)


class chirp(_shear_function):                                              # shear_function.py:74-86
    def __init__(self, dt, amplitude, omega_0, omega_f, periodT, zero="now"):
        _shear_function.__init__(self, zero)
        self.cpp_function = _PSEv1.ChirpShearFunction(amplitude, omega_0, omega_f, periodT, self._offset, dt)


class tukey_window(_shear_function):                                       # shear_function.py:88-102
    def __init__(self, dt, periodT, tukey_param, zero="now"):
        if tukey_param <= 0 or tukey_param > 1:
            context.msg.error("Tukey parameter must be within (0, 1]")
            raise RuntimeError("Error creating Tukey window function")
        _shear_function.__init__(self, zero)
        self.cpp_function = _PSEv1.TukeyWindowFunction(periodT, tukey_param, self._offset, dt)


class windowed(_shear_function):                                           # shear_function.py:104-114
    def __init__(self, function_form, window):
        _shear_function.__init__(self, "now")   # zero is not used by the windowed class
        self._keep = (function_form, window)    # keep Python subclasses of ShearFunction alive
        self.cpp_function = _PSEv1.WindowedFunction(function_form.cpp_function, window.cpp_function)
