// Host integrator class of the PSE method, HOOMD-free: the counterpart of `class Stokes : IntegrationMethodTwoStep`
// (PSEv1/Stokes.h:86-161, PSEv1/Stokes.cc:85-530).  It owns a pse_handle and calls the C-ABI (include/pse_amd.h);
// the particle arrays stay with the caller (HOOMD's ParticleData in the reference, PSEv1/Stokes.cc:454-461).
#pragma once
#include <memory>
#include <string>

#include "../../../include/pse_amd.h"
#include "ShearFunction.h"

namespace pse_host {

struct BoxDim {   // the part of HOOMD's BoxDim the path uses: lengths and the xy tilt factor
    double Lx, Ly, Lz, xy;
};

// device pointers of the particle data the step touches (ArrayHandle acquisitions at PSEv1/Stokes.cc:454-461)
struct ParticleArrays {
    pse_double4 *pos; pse_double4 *vel; pse_double3 *accel; pse_int3 *image; const pse_double4 *net_force;
    const unsigned int *group_members;   // may be null: all particles
    unsigned int group_size;
};

class Stokes {
public:
    // (sysdef, group, T, seed, nlist, xi, error) of the reference become (n_total, box, T, seed, xi, error): the
    // neighbour list is internal to the engine (PSEv1/Stokes.cc:85-111)
    Stokes(unsigned int n_total, BoxDim box, std::shared_ptr<Variant> T, unsigned int seed, double xi, double error, double dt);
    ~Stokes();
    void setT(std::shared_ptr<Variant> T) { m_T = T; }                                     // Stokes.h:106-109
    void setShear(std::shared_ptr<ShearFunction> f, double max_strain) { m_shear_func = f; m_max_strain = max_strain; }  // Stokes.h:118-121
    void setDeltaT(double dt) { m_deltaT = dt; }
    // explicit overrides of the parameter rule (0 = reference rule); must precede setParams
    void setOverrides(int Nx, int Ny, int Nz, int P, double rcut) { m_Nx = Nx; m_Ny = Ny; m_Nz = Nz; m_P = P; m_rcut = rcut; }
    void setParams();                                                                      // Stokes.cc:129-424
    void setBox(BoxDim box);                                                               // per-step box under shear
    void integrateStepOne(unsigned int timestep, const ParticleArrays &p);                 // Stokes.cc:429-523
    void integrateStepTwo(unsigned int) {}                                                 // Stokes.cc:528-530
    // force provider on the integrator's own cell list (the reference takes net_force from HOOMD, Stokes.cc:447)
    void pairRepulsion(const pse_double4 *pos, pse_double4 *force, const unsigned int *group, unsigned int n, double k,
                       double sigma, bool accumulate);
    pse_info info() const;
    int lanczosIterations() const { return m_m_Lanczos; }
    unsigned int hashedSeed() const { return m_seed; }
    pse_handle *handle() const { return m_h; }
private:
    unsigned int m_n_total;
    BoxDim m_box;
    std::shared_ptr<Variant> m_T;
    unsigned int m_seed;
    double m_xi, m_error, m_deltaT;
    std::shared_ptr<ShearFunction> m_shear_func;
    double m_max_strain = 0.5;
    int m_Nx = 0, m_Ny = 0, m_Nz = 0, m_P = 0;
    double m_rcut = 0.0;
    int m_m_Lanczos = 2;                                                                   // Stokes.cc:132
    pse_handle *m_h = nullptr;
};

}  // namespace pse_host
