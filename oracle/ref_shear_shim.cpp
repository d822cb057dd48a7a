// Test infrastructure only.  A C-ABI shim around the REFERENCE's own shear classes, compiled from the reference's source where it
// lies (/root/reference/PSEv1/SpecificShearFunction.h and ShearFunction.h, included -- not copied) by `make -C oracle ref` into
// oracle/_ref/libpse_ref_shear.so.  These headers need nothing but pybind11 (they include it under HOOMD's vendored path
// <hoomd/extern/pybind/include/pybind11/pybind11.h>: the Makefile points that path at the pybind11 installed in the image -- the real
// library, not a stand-in); everything else of the reference needs HOOMD, CUDA, cuFFT or LAPACKE and cannot be built here.
// tests/test_reference_pin.py holds the oracle's restatement and the product's host classes to THIS library where it exists
// (the build container; the built .so travels to the GPU box with the snapshot).
#include "SpecificShearFunction.h"   // from -I $(REF)/PSEv1

#include <memory>

namespace {
std::shared_ptr<ShearFunction> make(int kind, const double *a) {
    switch (kind) {
        case 0: return std::make_shared<SinShearFunction>(a[0], a[1], (unsigned int)a[2], a[3]);
        case 1: return std::make_shared<SteadyShearFunction>(a[0], (unsigned int)a[1], a[2]);
        case 2: return std::make_shared<ChirpShearFunction>(a[0], a[1], a[2], a[3], (unsigned int)a[4], a[5]);
        case 3: return std::make_shared<TukeyWindowFunction>(a[0], a[1], (unsigned int)a[2], a[3]);
        default: return nullptr;
    }
}
}  // namespace

extern "C" {
// kind: 0 sine (max_shear_rate, frequency, offset, dt), 1 steady (shear_rate, offset, dt), 2 chirp (amp, omega_0, omega_f, periodT,
// offset, dt), 3 Tukey window (periodT, tukey_param, offset, dt).  what: 0 getShearRate, 1 getStrain, 2 getOffset.  Returns 0 on success.
int pse_ref_shear(int kind, const double *args, unsigned int timestep, int what, double *out) {
    std::shared_ptr<ShearFunction> f = make(kind, args);
    if (!f || !out) return 1;
    *out = what == 0 ? f->getShearRate(timestep) : (what == 1 ? f->getStrain(timestep) : (double)f->getOffset());
    return 0;
}
// WindowedFunction(base, window) of two of the above
int pse_ref_shear_windowed(int kind_base, const double *args_base, int kind_win, const double *args_win, unsigned int timestep, int what,
                           double *out) {
    std::shared_ptr<ShearFunction> b = make(kind_base, args_base), w = make(kind_win, args_win);
    if (!b || !w || !out) return 1;
    WindowedFunction f(b, w);
    *out = what == 0 ? f.getShearRate(timestep) : (what == 1 ? f.getStrain(timestep) : (double)f.getOffset());
    return 0;
}
}
