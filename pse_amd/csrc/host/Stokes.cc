#include "Stokes.h"

#include <stdexcept>

namespace pse_host {

static void check(int status, const char *what) {
    if (status != PSE_OK) throw std::runtime_error(std::string(what) + ": " + pse_last_error());
}

Stokes::Stokes(unsigned int n_total, BoxDim box, std::shared_ptr<Variant> T, unsigned int seed, double xi, double error, double dt)
    : m_n_total(n_total), m_box(box), m_T(T), m_seed(seed), m_xi(xi), m_error(error), m_deltaT(dt) {
    // hash the user's seed so that it is unlikely to be a low positive integer (PSEv1/Stokes.cc:102)
    m_seed = m_seed * 0x12345677u + 0x12345u;
    m_seed ^= (m_seed >> 16);
    m_seed *= 0x45679u;
    m_shear_func = std::make_shared<SteadyShearFunction>(0.0, 0u, 0.0);   // integrate.py:93-94 default
}

Stokes::~Stokes() {
    if (m_h) pse_destroy(m_h);
}

void Stokes::setParams() {
    if (m_h) { pse_destroy(m_h); m_h = nullptr; }
    m_m_Lanczos = 2;   // "try two Lanczos iterations to start" (PSEv1/Stokes.cc:131-132)
    pse_params p{};
    p.n_max = m_n_total;
    p.Lx = m_box.Lx; p.Ly = m_box.Ly; p.Lz = m_box.Lz; p.xy = m_box.xy;
    p.xi = m_xi; p.error = m_error; p.max_strain = m_max_strain; p.seed = m_seed;
    p.Nx = m_Nx; p.Ny = m_Ny; p.Nz = m_Nz; p.P = m_P; p.rcut = m_rcut;
    p.device = -1; p.n_slabs = 1; p.slab_rank = 0;
    check(pse_create(&p, &m_h), "Error initializing Stokes");
}

void Stokes::setBox(BoxDim box) {
    m_box = box;
    if (m_h) check(pse_set_box(m_h, box.Lx, box.Ly, box.Lz, box.xy), "Stokes::setBox");
}

void Stokes::integrateStepOne(unsigned int timestep, const ParticleArrays &p) {
    if (!m_h) throw std::runtime_error("Stokes::setParams() has not been called");
    if (p.group_size == 0) return;                                       // Stokes.cc:443-444
    const double shear_rate = m_shear_func->getShearRate(timestep);      // Stokes.cc:473
    check(pse_step(m_h, p.pos, p.vel, p.accel, p.image, p.net_force, p.group_members, p.group_size,
                   m_T->getValue(timestep), m_deltaT, timestep, shear_rate, &m_m_Lanczos),
          "Stokes::integrateStepOne");
}

void Stokes::pairRepulsion(const pse_double4 *pos, pse_double4 *force, const unsigned int *group, unsigned int n, double k,
                           double sigma, bool accumulate) {
    if (!m_h) throw std::runtime_error("Stokes::setParams() has not been called");
    if (n == 0) return;
    check(pse_pair_repulsion(m_h, pos, force, group, n, k, sigma, accumulate ? 1 : 0), "Stokes::pairRepulsion");
}

pse_info Stokes::info() const {
    pse_info i{};
    if (m_h) pse_get_info(m_h, &i);
    return i;
}

}  // namespace pse_host
