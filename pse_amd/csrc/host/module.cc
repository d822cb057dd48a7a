// pybind11 module _PSEv1: the classes the reference registers at PSEv1/module.cc:13-22 (export_Stokes,
// export_ShearFunction, export_ShearFunctionWrap, export_SpecificShearFunction, export_VariantShearFunction), with the
// same Python names.  Device arrays are passed as integer addresses (torch tensor .data_ptr()).
#include <pybind11/pybind11.h>

#include "ShearFunction.h"
#include "Stokes.h"

namespace py = pybind11;
using namespace pse_host;

// trampoline so Python subclasses can override (the reference's ShearFunctionWrap binds the base class methods and has
// no trampoline, so overriding silently does nothing there -- SURVEY.md 2.4-10)
class ShearFunctionWrap : public ShearFunction {
public:
    using ShearFunction::ShearFunction;
    double getShearRate(unsigned int t) override { PYBIND11_OVERRIDE(double, ShearFunction, getShearRate, t); }
    double getStrain(unsigned int t) override { PYBIND11_OVERRIDE(double, ShearFunction, getStrain, t); }
    unsigned int getOffset() override { PYBIND11_OVERRIDE(unsigned int, ShearFunction, getOffset, ); }
};

template <class T>
static T *ptr(std::uintptr_t a) { return reinterpret_cast<T *>(a); }

PYBIND11_MODULE(_PSEv1, m) {
    py::class_<ShearFunction, ShearFunctionWrap, std::shared_ptr<ShearFunction>>(m, "ShearFunction")
        .def(py::init<>())
        .def("getShearRate", &ShearFunction::getShearRate)
        .def("getStrain", &ShearFunction::getStrain)
        .def("getOffset", &ShearFunction::getOffset);
    m.attr("ShearFunctionWrap") = m.attr("ShearFunction");
    py::class_<SinShearFunction, ShearFunction, std::shared_ptr<SinShearFunction>>(m, "SinShearFunction")
        .def(py::init<double, double, unsigned int, double>());
    py::class_<SteadyShearFunction, ShearFunction, std::shared_ptr<SteadyShearFunction>>(m, "SteadyShearFunction")
        .def(py::init<double, unsigned int, double>());
    py::class_<ChirpShearFunction, ShearFunction, std::shared_ptr<ChirpShearFunction>>(m, "ChirpShearFunction")
        .def(py::init<double, double, double, double, unsigned int, double>());
    py::class_<TukeyWindowFunction, ShearFunction, std::shared_ptr<TukeyWindowFunction>>(m, "TukeyWindowFunction")
        .def(py::init<double, double, unsigned int, double>());
    py::class_<WindowedFunction, ShearFunction, std::shared_ptr<WindowedFunction>>(m, "WindowedFunction")
        .def(py::init<std::shared_ptr<ShearFunction>, std::shared_ptr<ShearFunction>>());

    py::class_<Variant, std::shared_ptr<Variant>>(m, "Variant").def("getValue", &Variant::getValue);
    py::class_<VariantConst, Variant, std::shared_ptr<VariantConst>>(m, "VariantConst").def(py::init<double>());
    py::class_<VariantShearFunction, Variant, std::shared_ptr<VariantShearFunction>>(m, "VariantShearFunction")
        .def(py::init<std::shared_ptr<ShearFunction>, unsigned int, double, double>())
        .def("wrapValue", &VariantShearFunction::wrapValue);

    py::class_<Stokes, std::shared_ptr<Stokes>>(m, "Stokes")
        .def(py::init([](unsigned int n_total, double Lx, double Ly, double Lz, double xy, std::shared_ptr<Variant> T,
                         unsigned int seed, double xi, double error, double dt) {
            return std::make_shared<Stokes>(n_total, BoxDim{Lx, Ly, Lz, xy}, T, seed, xi, error, dt);
        }))
        .def("setT", &Stokes::setT)
        .def("setParams", &Stokes::setParams)
        .def("setShear", &Stokes::setShear)
        .def("setDeltaT", &Stokes::setDeltaT)
        .def("setOverrides", &Stokes::setOverrides)
        .def("setBox", [](Stokes &s, double Lx, double Ly, double Lz, double xy) { s.setBox(BoxDim{Lx, Ly, Lz, xy}); })
        .def("integrateStepOne", [](Stokes &s, unsigned int timestep, std::uintptr_t pos, std::uintptr_t vel, std::uintptr_t accel,
                                    std::uintptr_t image, std::uintptr_t force, std::uintptr_t group, unsigned int n) {
            s.integrateStepOne(timestep, ParticleArrays{ptr<pse_double4>(pos), ptr<pse_double4>(vel), ptr<pse_double3>(accel),
                                                       ptr<pse_int3>(image), ptr<const pse_double4>(force),
                                                       ptr<const unsigned int>(group), n});
        })
        .def("integrateStepTwo", &Stokes::integrateStepTwo)
        .def("pairRepulsion", [](Stokes &s, std::uintptr_t pos, std::uintptr_t force, std::uintptr_t group, unsigned int n, double k,
                                 double sigma, bool accumulate) {
            s.pairRepulsion(ptr<const pse_double4>(pos), ptr<pse_double4>(force), ptr<const unsigned int>(group), n, k, sigma, accumulate);
        })
        .def("lanczosIterations", &Stokes::lanczosIterations)
        .def("hashedSeed", &Stokes::hashedSeed)
        .def("info", [](const Stokes &s) {
            const pse_info i = s.info();
            py::dict d;
            d["Nx"] = i.Nx; d["Ny"] = i.Ny; d["Nz"] = i.Nz; d["P"] = i.P; d["rcut"] = i.rcut; d["xi"] = i.xi; d["eta"] = i.eta;
            d["gaussm"] = i.gaussm; d["self_mobility"] = i.self_mobility; d["lanczos_m"] = i.lanczos_m;
            return d;
        });
}
