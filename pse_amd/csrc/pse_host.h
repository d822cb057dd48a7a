// Internal host-side declarations shared by the parameter/table builder and the C-ABI driver.
#pragma once
#include <string>
#include <vector>

namespace pse {

// Degree of the per-interval polynomials of the real-space (smooth part) table and interval count per unit r.
constexpr int RS_DEG = 9;           // 10 coefficients per function
constexpr int RS_NCOEF = RS_DEG + 1;
constexpr int RS_PER_UNIT = 8;      // intervals of width 1/8 (in units of the particle radius)

struct Box {
    double Lx, Ly, Lz, xy;
};

// Everything Stokes::setParams derives (PSEv1/Stokes.cc:129-319).
struct Derived {
    double xi, error, max_strain;
    double rcut;          // Stokes.cc:135
    int kmax;             // Stokes.cc:138
    int Nx, Ny, Nz;       // Stokes.cc:143-199
    double lambda;        // Stokes.cc:217-219
    double gaussm;        // Stokes.cc:225-228
    int P;                // Stokes.cc:229-233
    double eta;           // Stokes.cc:234-236
    double hx, hy, hz;    // Stokes.cc:222
    double self;          // Stokes.cc:319
};

// Returns empty string on success, otherwise an error message.
std::string select_params(const Box &box, double xi, double error, double max_strain,
                          int Nx, int Ny, int Nz, int P, double rcut, Derived &out);

// Real-space table: for interval k (r in [k/8,(k+1)/8)) and local t = 2*(8r-k)-1 in [-1,1):
//   f_w(r) = sum_q coef[k][q] t^q,  g_w(r) = sum_q coef[k][RS_NCOEF+q] t^q
// are the *smooth* (free-space wave) parts; the device adds the analytic RPY branch:
//   f = f_RPY - f_w, g = g_RPY - g_w   (replaces m_ewaldC1, PSEv1/Stokes.cc:334-422).
void build_realspace_table(double xi, double rcut, std::vector<double> &coef, int &n_intervals);

// Symmetric tridiagonal eigen-decomposition (implicit QL), replaces LAPACKE_spteqr (PSEv1/Brownian.cu:540).
// d[0..n) diagonal, e[0..n-1) off-diagonal. On return d = eigenvalues, z (n x n, row-major) = eigenvectors in
// columns. Returns false if it fails to converge.
bool tridiag_eigen(int n, std::vector<double> &d, std::vector<double> &e, std::vector<double> &z);

// t = T^{1/2} e_1 for the Lanczos tridiagonal T(alpha[0..m), beta[1..m)) (PSEv1/Brownian.cu:563-582).
bool lanczos_sqrt_e1(int m, const double *alpha, const double *beta, std::vector<double> &t);

}  // namespace pse
