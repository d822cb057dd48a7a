// Device entry points of include/pse_amd.h for the CPU SANITIZER build only (python -m pse_amd.build --asan; never part of
// libpse_amd.so): GPU AddressSanitizer is not available on the MI355X pool, so the host-side code -- the parameter rule and the
// real-space table builder (pse_params.cpp), the tridiagonal solver, and the C++ host classes (csrc/host/) -- is
// exercised under -fsanitize=address,undefined against this stand-in.  The six calls the host classes make (pse_create,
// pse_destroy, pse_set_box, pse_get_info, pse_step, pse_pair_repulsion) keep a small host object that runs the REAL parameter
// rule and table builder; every entry point that would need a device returns PSE_ERR_HIP.  No test takes a number from here.
#include <cmath>
#include <vector>

#include "pse_err.h"

using namespace pse;

struct pse_handle {
    pse_params par;
    Derived d;
    pse_info info;
    std::vector<double> coef;
    int n_intervals = 0;
    unsigned long long steps = 0;
};
struct pse_team { int unused; };

static int no_device(const char *what) { return fail(PSE_ERR_HIP, "%s: sanitizer build, there is no device behind this library", what); }

extern "C" {

int pse_create(const pse_params *p, pse_handle **out) {
    if (!p || !out) return fail(PSE_ERR_INVALID, "null argument");
    *out = nullptr;
    if (p->n_max == 0) return fail(PSE_ERR_INVALID, "n_max must be positive");
    if (!(std::fabs(p->xy) <= 0.5 * (1.0 + 1e-9))) return fail(PSE_ERR_INVALID, "tilt xy = %g outside [-0.5, 0.5]", p->xy);
    pse_handle *h = new pse_handle();
    h->par = *p;
    std::string e = select_params(Box{p->Lx, p->Ly, p->Lz, p->xy}, p->xi, p->error, p->max_strain, p->Nx, p->Ny, p->Nz, p->P, p->rcut, h->d);
    if (!e.empty()) { delete h; return fail(PSE_ERR_INVALID, "%s", e.c_str()); }
    if (int rc = gaussian_fits(h->d, h->d.hx, h->d.hy, h->d.hz)) { delete h; return rc; }
    build_realspace_table(h->d.xi, h->d.rcut, h->coef, h->n_intervals);
    fill_info(h->d, &h->info);
    *out = h;
    return 0;
}
int pse_destroy(pse_handle *h) { delete h; return 0; }
int pse_set_box(pse_handle *h, double Lx, double Ly, double Lz, double xy) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    if (!(Lx > 0 && Ly > 0 && Lz > 0)) return fail(PSE_ERR_INVALID, "box lengths must be positive");
    if (!(std::fabs(xy) <= 0.5 * (1.0 + 1e-9))) return fail(PSE_ERR_INVALID, "tilt xy = %g outside [-0.5, 0.5]", xy);
    if (int rc = gaussian_fits(h->d, Lx / h->d.Nx, Ly / h->d.Ny, Lz / h->d.Nz)) return rc;
    h->par.Lx = Lx; h->par.Ly = Ly; h->par.Lz = Lz; h->par.xy = xy;
    h->info.hx = Lx / h->d.Nx; h->info.hy = Ly / h->d.Ny; h->info.hz = Lz / h->d.Nz;
    return 0;
}
int pse_get_info(pse_handle *h, pse_info *info) {
    if (!h || !info) return fail(PSE_ERR_INVALID, "null argument");
    *info = h->info;
    return 0;
}
int pse_step(pse_handle *h, pse_double4 *, pse_double4 *, pse_double3 *, pse_int3 *, const pse_double4 *, const unsigned int *,
             unsigned int N, double kT, double dt, unsigned int, double, int *lanczos_m) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    if (N == 0 || N > h->par.n_max) return fail(PSE_ERR_INVALID, "N = %u outside (0, n_max = %u]", N, h->par.n_max);
    if (kT < 0 || !(dt > 0)) return fail(PSE_ERR_INVALID, "need kT >= 0 and dt > 0");
    ++h->steps;
    if (lanczos_m && *lanczos_m < 2) *lanczos_m = 2;
    return 0;   // nothing is integrated: the arrays are device pointers and there is no device
}
int pse_pair_repulsion(pse_handle *h, const pse_double4 *, pse_double4 *, const unsigned *, unsigned N, double, double sigma, int) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    if (N == 0 || N > h->par.n_max) return fail(PSE_ERR_INVALID, "N = %u outside (0, n_max = %u]", N, h->par.n_max);
    if (!(sigma > 0.0) || sigma > h->d.rcut) return fail(PSE_ERR_INVALID, "repulsion range %.4f outside (0, rcut = %.4f]", sigma, h->d.rcut);
    return 0;
}

int pse_set_stream(pse_handle *, void *) { return no_device("pse_set_stream"); }
int pse_set_timing(pse_handle *, int) { return no_device("pse_set_timing"); }
int pse_set_async(pse_handle *, int) { return no_device("pse_set_async"); }
int pse_set_timestep_offset(pse_handle *, const unsigned int *) { return no_device("pse_set_timestep_offset"); }
int pse_debug_last_gate(pse_handle *, int *) { return no_device("pse_debug_last_gate"); }
int pse_set_neighbor_skin(pse_handle *, double) { return no_device("pse_set_neighbor_skin"); }
int pse_neighbor_stats(pse_handle *, double *, unsigned long long *, unsigned long long *) { return no_device("pse_neighbor_stats"); }
int pse_mobility(pse_handle *, const pse_double4 *, const pse_double4 *, pse_double4 *, const unsigned int *, unsigned int, int) { return no_device("pse_mobility"); }
int pse_brownian_velocity(pse_handle *, const pse_double4 *, const pse_double4 *, pse_double4 *, const unsigned int *, unsigned int, double,
                          double, unsigned int, int *) { return no_device("pse_brownian_velocity"); }
int pse_sqrt_mreal(pse_handle *, const pse_double4 *, const pse_double4 *, pse_double4 *, const unsigned int *, unsigned int, double, int *) { return no_device("pse_sqrt_mreal"); }
int pse_random_psi(pse_handle *, pse_double4 *, const unsigned int *, unsigned int, unsigned int) { return no_device("pse_random_psi"); }
int pse_eval_realspace(pse_handle *, const double *, int, double *, double *) { return no_device("pse_eval_realspace"); }
int pse_debug_copy_grid(pse_handle *, int, double *) { return no_device("pse_debug_copy_grid"); }
int pse_debug_spread(pse_handle *, const pse_double4 *, const pse_double4 *, const unsigned int *, unsigned int) { return no_device("pse_debug_spread"); }
int pse_debug_kvector(pse_handle *, int, const int *, double *) { return no_device("pse_debug_kvector"); }
int pse_debug_grid_placement(pse_handle *, int *, float *, float *) { return no_device("pse_debug_grid_placement"); }
int pse_debug_vq_roundtrip(int, const double *, double *) { return no_device("pse_debug_vq_roundtrip"); }
int pse_debug_matvec_ms(pse_handle *, int, float *) { return no_device("pse_debug_matvec_ms"); }
int pse_set_lanczos_extra(pse_handle *, int) { return no_device("pse_set_lanczos_extra"); }
int pse_brownian_velocity_part(pse_handle *, const pse_double4 *, const pse_double4 *, pse_double4 *, const unsigned int *, unsigned int, double,
                               double, unsigned int, int, int *) { return no_device("pse_brownian_velocity_part"); }
int pse_integrate(pse_handle *, pse_double4 *, const pse_double4 *, pse_double3 *, pse_int3 *, const pse_double4 *, const unsigned int *, unsigned int,
                  double, double) { return no_device("pse_integrate"); }
int pse_team_redistribute_local(pse_team *, pse_double4 *const *, pse_double4 *const *, pse_double3 *const *, pse_int3 *const *, pse_double4 *const *,
                                unsigned int *const *, unsigned int *const *) { return no_device("pse_team_redistribute_local"); }
int pse_team_set_lanczos_extra(pse_team *, int) { return no_device("pse_team_set_lanczos_extra"); }
int pse_team_unique_id(void *) { return no_device("pse_team_unique_id"); }
int pse_team_create(pse_handle **, int, const void *, pse_team **) { return no_device("pse_team_create"); }
int pse_team_create_transport(pse_handle *, const pse_transport *, pse_team **) { return no_device("pse_team_create_transport"); }
int pse_team_destroy(pse_team *) { return 0; }
int pse_team_debug_solo(pse_team *, int) { return no_device("pse_team_debug_solo"); }
int pse_team_step_local(pse_team *, pse_double4 *const *, pse_double4 *const *, pse_double3 *const *, pse_int3 *const *, const pse_double4 *const *,
                        unsigned int *const *, unsigned int *const *, double, double, unsigned int, double, int, int *) { return no_device("pse_team_step_local"); }
int pse_local_layout(pse_handle *, int *, int *, int *, int *, int *) { return no_device("pse_local_layout"); }
int pse_team_local_status(pse_team *, int *) { return no_device("pse_team_local_status"); }
int pse_team_set_diag(pse_team *, int) { return no_device("pse_team_set_diag"); }
int pse_team_get_diag(pse_team *, pse_team_diag *) { return no_device("pse_team_get_diag"); }
int pse_team_mobility(pse_team *, const pse_double4 *const *, const pse_double4 *const *, pse_double4 *const *, const unsigned int *, unsigned int, int) { return no_device("pse_team_mobility"); }
int pse_team_brownian_velocity(pse_team *, const pse_double4 *const *, const pse_double4 *const *, pse_double4 *const *, const unsigned int *,
                               unsigned int, double, double, unsigned int, int *) { return no_device("pse_team_brownian_velocity"); }
int pse_team_step(pse_team *, pse_double4 *const *, pse_double4 *const *, pse_double3 *const *, pse_int3 *const *, const pse_double4 *const *,
                  const unsigned int *, unsigned int, double, double, unsigned int, double, int *) { return no_device("pse_team_step"); }

}  // extern "C"
