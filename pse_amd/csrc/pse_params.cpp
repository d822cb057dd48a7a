// Host-side set-up of the PSE engine: parameter selection, real-space function table, and the small
// tridiagonal eigen-solver the Lanczos driver needs.  Replaces Stokes::setParams
// (PSEv1/Stokes.cc:129-424) and LAPACKE_spteqr (PSEv1/Brownian.cu:540,673).
#include "pse_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <thread>

namespace pse {

typedef long double ld;
static const ld PI_L = 3.14159265358979323846264338327950288L;

// Smallest 2^a 3^b 5^c >= n within [8, 4096]  (PSEv1/Stokes.cc:147-199); 0 if none.
static int next235(int n) {
    int best = 0;
    for (long p2 = 1; p2 <= 4096; p2 *= 2)
        for (long p3 = p2; p3 <= 4096; p3 *= 3)
            for (long p5 = p3; p5 <= 4096; p5 *= 5)
                if (p5 >= 8 && p5 >= n && (best == 0 || p5 < best)) best = (int)p5;
    return best;
}

std::string select_params(const Box &box, double xi, double error, double max_strain,
                          int Nx, int Ny, int Nz, int P, double rcut, Derived &o) {
    if (!(xi > 0.0)) return "xi must be positive";
    if (!(error > 0.0 && error < 1.0)) return "error must be in (0,1)";
    if (!(box.Lx > 0 && box.Ly > 0 && box.Lz > 0)) return "box lengths must be positive";
    if (max_strain < 0) return "max_strain must be non-negative";
    if (P < 0) return "P override must be positive (0: the rule of Stokes.cc:225-233)";
    if (Nx < 0 || Ny < 0 || Nz < 0) return "grid override must be positive (0: the rule of Stokes.cc:143-199)";
    if (rcut < 0) return "rcut override must be positive (0: the rule of Stokes.cc:135)";
    o.xi = xi; o.error = error; o.max_strain = max_strain;
    const double s = std::sqrt(-std::log(error));
    o.rcut = rcut > 0 ? rcut : s / xi;                                   // Stokes.cc:135
    o.kmax = int(2.0 * s * xi) + 1;                                      // Stokes.cc:138
    const double L[3] = {box.Lx, box.Ly, box.Lz};
    int over[3] = {Nx, Ny, Nz}, n[3];
    for (int a = 0; a < 3; ++a) {
        if (over[a] > 0) { n[a] = over[a]; continue; }
        n[a] = next235(int(o.kmax * L[a] / M_PI) + 1);                   // Stokes.cc:143-199
        if (n[a] == 0) return "requested FFT grid exceeds 4096 nodes in one dimension; reduce xi";
    }
    o.Nx = n[0]; o.Ny = n[1]; o.Nz = n[2];
    if (o.Nx < 2 || o.Ny < 2 || o.Nz < 2) return "grid must have at least 2 nodes per dimension";
    const double g = max_strain, g2 = g * g;
    o.lambda = 1.0 + g2 / 2.0 + g * std::sqrt(1.0 + g2 / 4.0);          // Stokes.cc:217-219
    int i = 0;                                                           // Stokes.cc:225-228 (integer counter: no drift)
    while (std::erfc((1.0 + 0.01 * i) / std::sqrt(2.0 * o.lambda)) > error) ++i;
    o.gaussm = 1.0 + 0.01 * i;
    o.P = P > 0 ? P : int(o.gaussm * o.gaussm / M_PI) + 1;               // Stokes.cc:229
    o.P = std::min(o.P, std::min(o.Nx, std::min(o.Ny, o.Nz)));          // Stokes.cc:231-233
    o.hx = box.Lx / o.Nx; o.hy = box.Ly / o.Ny; o.hz = box.Lz / o.Nz;    // Stokes.cc:222
    const double w = o.P * o.hx / 2.0;                                   // Stokes.cc:234
    o.eta = (2.0 * w / o.gaussm) * (2.0 * w / o.gaussm) * xi * xi;       // Stokes.cc:236
    if (!(o.eta < 1.0)) {
        char buf[256];
        snprintf(buf, sizeof buf, "grid too coarse for this xi: eta = %.4f >= 1 makes the k-space factor grow "
                 "(need P*h*xi/gaussm < 1); use a finer grid or a smaller xi", o.eta);
        return buf;
    }
    const ld xl = xi, sp = sqrtl(PI_L);
    o.self = (double)((1.0L + 4.0L * sp * xl * erfcl(2.0L * xl) - expl(-4.0L * xl * xl)) / (4.0L * sp * xl));  // Stokes.cc:319
    return "";
}

// ---- real-space table ------------------------------------------------------------------------------------
// 16-point Gauss-Legendre, positive half
static const ld GLX[8] = {
    0.0950125098376374401853193354249581L, 0.2816035507792589132304605014604961L,
    0.4580167776572273863424194429835776L, 0.6178762444026437484466717640487910L,
    0.7554044083550030338951011948474423L, 0.8656312023878317438804678977123931L,
    0.9445750230732325760779884155346083L, 0.9894009349916499325961541734503326L};
static const ld GLW[8] = {
    0.1894506104550684962853967232082831L, 0.1826034150449235888667636679692199L,
    0.1691565193950025381893120790303600L, 0.1495959888165767320815017305474786L,
    0.1246289712555338720524762821920164L, 0.0951585116824927848099251076022462L,
    0.0622535239386478928628438369943777L, 0.0271524594117540948517805724560182L};

static ld sph_j0(ld x) { return fabsl(x) < 1e-4L ? 1.0L - x * x / 6.0L + x * x * x * x / 120.0L : sinl(x) / x; }
static ld sph_j1_over_x(ld x) {
    if (fabsl(x) < 0.05L) {
        ld x2 = x * x;
        return 1.0L / 3.0L - x2 / 30.0L + x2 * x2 / 840.0L - x2 * x2 * x2 / 45360.0L + x2 * x2 * x2 * x2 / 3991680.0L;
    }
    return (sinl(x) / x - cosl(x)) / (x * x);
}

// Free-space wave part of the RPY-Ewald pair functions,
//   M_wave(r) = (1/(2pi)^3) Int (6 pi / k^2) H(k) sinc^2(k a) (I - kk) e^{ik.r} d^3k,  H = (1+k^2/4xi^2) e^{-k^2/4xi^2}
// (k-space factor: PSEv1/Helper.cu:326 x PSEv1/Mobility.cu:290), after the angular integration:
//   f_w = (3/pi) Int_0^inf H sinc^2 [ j0(kr) - j1(kr)/(kr) ] dk,   g_w = (3/pi) Int_0^inf H sinc^2 2 j1(kr)/(kr) dk.
static void wave_fg(ld r, ld xi, ld &f, ld &g) {
    const ld kmax = 2.0L * xi * sqrtl(48.0L);
    ld dk = PI_L / (r + 3.0L);
    const int np = (int)ceill(kmax / dk);
    dk = kmax / np;
    ld sf = 0, sg = 0;
    for (int p = 0; p < np; ++p) {
        const ld c = (p + 0.5L) * dk, h = 0.5L * dk;
        for (int q = 0; q < 16; ++q) {
            const ld k = q < 8 ? c - h * GLX[7 - q] : c + h * GLX[q - 8];
            const ld w = q < 8 ? GLW[7 - q] : GLW[q - 8];
            const ld k2 = k * k / (4.0L * xi * xi);
            const ld H = (1.0L + k2) * expl(-k2);
            ld s = sph_j0(k); s *= s;
            const ld x = k * r, j = sph_j1_over_x(x);
            sf += w * h * H * s * (sph_j0(x) - j);
            sg += w * h * H * s * 2.0L * j;
        }
    }
    f = 3.0L / PI_L * sf;
    g = 3.0L / PI_L * sg;
}

void build_realspace_table(double xi, double rcut, std::vector<double> &coef, int &n_intervals) {
    n_intervals = (int)std::ceil(rcut * RS_PER_UNIT) + 1;
    coef.assign((size_t)n_intervals * 2 * RS_NCOEF, 0.0);
    const int n = RS_NCOEF;
    // Chebyshev -> monomial conversion matrix T_q(t) = sum_p C[q][p] t^p
    ld C[RS_NCOEF][RS_NCOEF] = {};
    C[0][0] = 1;
    C[1][1] = 1;
    for (int q = 2; q < n; ++q)
        for (int p = 0; p < n; ++p) C[q][p] = (p > 0 ? 2 * C[q - 1][p - 1] : 0) - C[q - 2][p];
    auto fit = [&](int k) {
        ld fv[RS_NCOEF], gv[RS_NCOEF], tn[RS_NCOEF];
        for (int m = 0; m < n; ++m) {
            tn[m] = cosl(PI_L * (m + 0.5L) / n);  // Chebyshev nodes of the first kind
            const ld r = (k + 0.5L * (tn[m] + 1.0L)) / RS_PER_UNIT;
            wave_fg(r, xi, fv[m], gv[m]);
        }
        ld cf[RS_NCOEF], cg[RS_NCOEF];
        for (int q = 0; q < n; ++q) {
            ld a = 0, b = 0;
            for (int m = 0; m < n; ++m) {
                const ld Tq = cosl(q * PI_L * (m + 0.5L) / n);
                a += fv[m] * Tq; b += gv[m] * Tq;
            }
            cf[q] = a * (q == 0 ? 1.0L : 2.0L) / n;
            cg[q] = b * (q == 0 ? 1.0L : 2.0L) / n;
        }
        double *out = &coef[(size_t)k * 2 * n];
        for (int p = 0; p < n; ++p) {
            ld a = 0, b = 0;
            for (int q = p; q < n; ++q) { a += cf[q] * C[q][p]; b += cg[q] * C[q][p]; }
            out[p] = (double)a; out[n + p] = (double)b;
        }
    };
    // the quadrature is ~1e4 long-double transcendentals per node: spread the intervals over a few host threads
    const int nt = std::max(1, std::min(8, (int)std::thread::hardware_concurrency()));
    std::vector<std::thread> pool;
    for (int w = 0; w < nt; ++w)
        pool.emplace_back([&, w] { for (int k = w; k < n_intervals; k += nt) fit(k); });
    for (auto &th : pool) th.join();
}

// ---- symmetric tridiagonal eigen-solver (implicit QL with Wilkinson shifts) ------------------------------
bool tridiag_eigen(int n, std::vector<double> &d, std::vector<double> &e_in, std::vector<double> &z) {
    std::vector<double> e(n, 0.0);
    for (int i = 0; i + 1 < n; ++i) e[i] = e_in[i];
    z.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) z[(size_t)i * n + i] = 1.0;
    for (int l = 0; l < n; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
            }
            if (m != l) {
                if (iter++ == 200) return false;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i], b = c * e[i];
                    e[i + 1] = (r = std::hypot(f, g));
                    if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
                    s = f / r; c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    d[i + 1] = g + (p = s * r);
                    g = c * r - b;
                    for (int k = 0; k < n; ++k) {
                        f = z[(size_t)k * n + i + 1];
                        z[(size_t)k * n + i + 1] = s * z[(size_t)k * n + i] + c * f;
                        z[(size_t)k * n + i] = c * z[(size_t)k * n + i] - s * f;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p; e[l] = g; e[m] = 0.0;
            }
        } while (m != l);
    }
    return true;
}

bool lanczos_sqrt_e1(int m, const double *alpha, const double *beta, std::vector<double> &t) {
    std::vector<double> d(alpha, alpha + m), e(std::max(m - 1, 0)), z;
    for (int i = 0; i + 1 < m; ++i) e[i] = beta[i + 1];
    if (!tridiag_eigen(m, d, e, z)) return false;
    t.assign(m, 0.0);
    // t = Z sqrt(Lambda) Z^T e1 ; (Z^T e1)_j = Z[0][j]   (PSEv1/Brownian.cu:563-582)
    for (int j = 0; j < m; ++j) {
        const double s = std::sqrt(std::max(d[j], 0.0)) * z[j];
        for (int i = 0; i < m; ++i) t[i] += z[(size_t)i * m + j] * s;
    }
    return true;
}

}  // namespace pse
