// The part of the C-ABI (include/pse_amd.h) that never touches a device: the error text, the parameter rule of
// Stokes::setParams without a handle (PSEv1/Stokes.cc:129-236,319) and the Lanczos tridiagonal square root
// (LAPACKE_spteqr + the host loops at PSEv1/Brownian.cu:540-582).  Plain C++: linked into libpse_amd.so, and -- with
// pse_params.cpp and a stub of the device entry points -- into the sanitizer build that the CPU tests run against.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "pse_err.h"

namespace pse {

std::string &error_text() {
    static thread_local std::string text;
    return text;
}

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    error_text() = buf;
    return code;
}

void fill_info(const Derived &d, pse_info *o) {
    memset(o, 0, sizeof *o);
    o->Nx = d.Nx; o->Ny = d.Ny; o->Nz = d.Nz; o->P = d.P;
    o->rcut = d.rcut; o->xi = d.xi; o->eta = d.eta; o->gaussm = d.gaussm; o->lambda = d.lambda;
    o->self_mobility = d.self; o->hx = d.hx; o->hy = d.hy; o->hz = d.hz;
}

// The spreading Gaussian exp(-c r^2), c = 2 xi^2 / eta, has its width from the SMALLEST grid spacing (the reference's rule makes the
// three spacings equal up to the rounding of the grid sizes, PSEv1/Stokes.cc:147-214); spread and gather build its P values per axis
// by a product recurrence whose factors reach exp(c h^2 P) and whose values fall to exp(-c h^2 P^2 / 4).  A grid or box override whose
// coarsest spacing puts those outside the double range would turn into NaN velocities: refused.
int gaussian_fits(const Derived &d, double hx, double hy, double hz) {
    const double c = 2.0 * d.xi * d.xi / d.eta, hmax = std::max(hx, std::max(hy, hz));
    const double worst = c * hmax * hmax * std::max((double)d.P, 0.25 * d.P * d.P);
    if (!(worst < 700.0))
        return fail(PSE_ERR_INVALID, "grid spacings (%g, %g, %g) too unequal for the spreading Gaussian of P = %d points, eta = %g: "
                    "exp(+-%.0f) over its support on the coarsest axis", hx, hy, hz, d.P, d.eta, worst);
    return 0;
}

}  // namespace pse

using namespace pse;

extern "C" const char *pse_last_error(void) { return error_text().c_str(); }

extern "C" int pse_host_select_params(const pse_params *p, pse_info *info) {
    if (!p || !info) return fail(PSE_ERR_INVALID, "null argument");
    Derived d;
    std::string e = select_params(Box{p->Lx, p->Ly, p->Lz, p->xy}, p->xi, p->error, p->max_strain, p->Nx, p->Ny, p->Nz,
                                  p->P, p->rcut, d);
    if (!e.empty()) return fail(PSE_ERR_INVALID, "%s", e.c_str());
    if (int rc = gaussian_fits(d, d.hx, d.hy, d.hz)) return rc;
    fill_info(d, info);
    return 0;
}

extern "C" int pse_host_lanczos_sqrt_e1(int m, const double *alpha, const double *beta, double *t) {
    if (m < 1 || m > 4096 || !alpha || !beta || !t) return fail(PSE_ERR_INVALID, "bad argument");
    std::vector<double> tv;
    if (!lanczos_sqrt_e1(m, alpha, beta, tv)) return fail(PSE_ERR_NUMERIC, "tridiagonal eigen-solve did not converge");
    std::copy(tv.begin(), tv.end(), t);
    return 0;
}
