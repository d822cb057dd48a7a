/*
 * pse_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked or called by the product path).
 *
 * CPU restatement of the mathematics of the reference PSE hot path
 * (stochasticHydroTools/PSE, HOOMD plugin PSEv1) as a *direct* O(N^2) periodic
 * Rotne-Prager-Yamakawa mobility evaluation: Ewald-split real-space image sum +
 * direct structure-factor k-sum + self term, all fp64 (long double inside the
 * radial functions).  Units as in the reference: particle radius a = 1
 * (PSEv1/Stokes.cc:314-316), mobility in units of 1/(6 pi eta a).
 *
 * What each function follows in the reference:
 *   pse_oracle_fg_real      real-space RPY-Ewald functions "Imrr"/"rr" tabulated at
 *                           PSEv1/Stokes.cc:334-406 (three branches r>2a, r==2a, r<2a).
 *                           Not a transcription: re-derived as the double
 *                           sphere-surface average of the Hasimoto-split biharmonic
 *                           kernel (see the comment above W0()), which collapses the
 *                           three branches into one expression.
 *   pse_oracle_fg_wave_quad the defining Fourier integral of the same functions
 *                           (k-space factor PSEv1/Helper.cu:326 x PSEv1/Mobility.cu:290),
 *                           by Gauss-Legendre quadrature -- an independent route used to
 *                           pin the closed form.
 *   pse_oracle_self         PSEv1/Stokes.cc:319.
 *   pse_oracle_mobility_direct / _dense
 *                           U = M.F with M = M_real + M_wave as defined by
 *                           PSEv1/Mobility.cu:669-677 (pair formula), PSEv1/Helper.cu:300-327
 *                           (sheared wave vectors, exact pi, and the Hasimoto factor) and
 *                           PSEv1/Mobility.cu:283-295 (sinc^2 RPY factor, transverse projector),
 *                           but with the grid/cutoff approximations removed: all images
 *                           and all wave vectors to a 1e-14 truncation.
 *
 * Parity pin: the reference has no tests or golden vectors and cannot
 * be built here, but it holds its real-space functions, self term, parameter rule, seed
 * hash, k-space factor and shear formulas as plain expression text.
 * tests/golden/make_reference_fixture.py reads that text from /root/reference in the build
 * container, evaluates it (50-digit and fp64) and commits the numbers as
 * tests/golden/reference_arithmetic.json; tests/test_reference_pin.py holds this oracle to
 * them (pse_oracle_fg_real vs PSEv1/Stokes.cc:348-406 to 2e-14 on all three branches incl.
 * r < 0.25, where the fixture is independent of the quadrature below).  On top of that the
 * known-answer values KAT-1..KAT-5 of SURVEY.md section 8(c), xi-independence and the
 * quadrature route in tests/test_oracle.py.
 *
 * Box convention (HOOMD triclinic, PSEv1/Mobility.cu:223-230): lattice vectors
 *   a1 = (Lx,0,0), a2 = (xy*Ly, Ly, 0), a3 = (0,0,Lz); box[4] = {Lx,Ly,Lz,xy}.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef long double ld;
#define PI_L 3.14159265358979323846264338327950288L

/* ------------------------------------------------------------------------------------------
 * Radial functions.
 *
 * Hasimoto split of the biharmonic Green's function r = B_w(r) + B_r(r) with the smooth part
 *   B_w(s) = s erf(xi s) + exp(-xi^2 s^2)/(xi sqrt(pi))          (FT: -8 pi/k^4 (1+k^2/4xi^2) e^{-k^2/4xi^2}).
 * The RPY tensor is the double surface average over two spheres of radius a of the Oseen tensor
 * (1/8 pi eta)(lap I - grad grad) r; averaging commutes with the derivatives, and for an even
 * radial function phi the double-shell average is a second difference of its second "s-weighted"
 * antiderivative:   <<phi>>(r) = [W(r+2a) - 2 W(r) + W(r-2a)] / (4 a^2 r),   W'' = s phi(s).
 * With Bbar = <<B_w>>:  f_w = (3a/4)(Bbar'' + Bbar'/r),  g_w = (3a/2) Bbar'/r   [units 1/(6 pi eta a)],
 * and the real-space functions are  f_r = f_RPY - f_w,  g_r = g_RPY - g_w  for every r > 0
 * (overlapping pairs included: f_RPY, g_RPY switch branch at r = 2a, f_w, g_w do not).
 * ------------------------------------------------------------------------------------------ */
static ld W0(ld s, ld xi) {
    ld x2 = xi * xi, x3 = x2 * xi, x4 = x2 * x2;
    return erfl(xi * s) * (s * s * s * s / 12.0L - 1.0L / (16.0L * x4))
         + expl(-x2 * s * s) / sqrtl(PI_L) * (s * s * s / (12.0L * xi) - s / (24.0L * x3));
}
static ld W1(ld s, ld xi) { /* W0' */
    ld x2 = xi * xi, x3 = x2 * xi;
    return s * s * s / 3.0L * erfl(xi * s) + expl(-x2 * s * s) * (2.0L * x2 * s * s - 1.0L) / (6.0L * x3 * sqrtl(PI_L));
}
static ld W2(ld s, ld xi) { /* W0'' = s B_w(s) */
    return s * s * erfl(xi * s) + s * expl(-xi * xi * s * s) / (xi * sqrtl(PI_L));
}

static void fg_rpy(ld r, ld *f, ld *g) { /* free-space RPY, a = 1 */
    if (r > 2.0L) { ld r3 = r * r * r; *f = 0.75L / r + 0.5L / r3; *g = 1.5L / r - 1.0L / r3; }
    else          { *f = 1.0L - 9.0L * r / 32.0L; *g = 1.0L - 3.0L * r / 16.0L; }
}

/* composite Gauss-Legendre (16 points per panel) of the wave-part Fourier integral */
static const ld GLX[8] = {
    0.0950125098376374401853193354249581L, 0.2816035507792589132304605014604961L,
    0.4580167776572273863424194429835776L, 0.6178762444026437484466717640487910L,
    0.7554044083550030338951011948474423L, 0.8656312023878317438804678977123931L,
    0.9445750230732325760779884155346083L, 0.9894009349916499325961541734503326L };
static const ld GLW[8] = {
    0.1894506104550684962853967232082831L, 0.1826034150449235888667636679692199L,
    0.1691565193950025381893120790303600L, 0.1495959888165767320815017305474786L,
    0.1246289712555338720524762821920164L, 0.0951585116824927848099251076022462L,
    0.0622535239386478928628438369943777L, 0.0271524594117540948517805724560182L };

static ld j0s(ld x) { return fabsl(x) < 1e-4L ? 1.0L - x * x / 6.0L + x * x * x * x / 120.0L : sinl(x) / x; }
static ld j1ox(ld x) { /* j1(x)/x, stable at 0 */
    if (fabsl(x) < 0.05L) { ld x2 = x * x; return 1.0L / 3.0L - x2 / 30.0L + x2 * x2 / 840.0L - x2 * x2 * x2 / 45360.0L + x2 * x2 * x2 * x2 / 3991680.0L; }
    return (sinl(x) / x - cosl(x)) / (x * x);
}
static void fg_wave_quad_l(ld r, ld xi, ld *f, ld *g) {
    ld kmax = 2.0L * xi * sqrtl(48.0L);              /* exp(-k^2/4xi^2) < 1.5e-21 */
    ld dk = PI_L / (r + 3.0L);                       /* <= half a period of the fastest factor */
    int np = (int)ceill(kmax / dk);
    dk = kmax / np;
    ld sf = 0, sg = 0;
    for (int p = 0; p < np; ++p) {
        ld c = (p + 0.5L) * dk, h = 0.5L * dk;
        for (int q = 0; q < 16; ++q) {
            ld k = (q < 8) ? c - h * GLX[7 - q] : c + h * GLX[q - 8];
            ld w = (q < 8) ? GLW[7 - q] : GLW[q - 8];
            ld k2 = k * k / (4.0L * xi * xi);
            ld H = (1.0L + k2) * expl(-k2);
            ld s = j0s(k); s *= s;
            ld x = k * r;
            ld j = j1ox(x);
            sf += w * h * H * s * (j0s(x) - j);
            sg += w * h * H * s * 2.0L * j;
        }
    }
    *f = 3.0L / PI_L * sf; *g = 3.0L / PI_L * sg;
}

static void fg_wave_l(ld r, ld xi, ld *f, ld *g) {
    if (r < 0.25L) { fg_wave_quad_l(r, xi, f, g); return; } /* closed form cancels like eps/(xi^4 r^3) */
    ld D0 = W0(r + 2.0L, xi) - 2.0L * W0(r, xi) + W0(r - 2.0L, xi);
    ld D1 = W1(r + 2.0L, xi) - 2.0L * W1(r, xi) + W1(r - 2.0L, xi);
    ld D2 = W2(r + 2.0L, xi) - 2.0L * W2(r, xi) + W2(r - 2.0L, xi);
    *f = 3.0L / (16.0L * r) * (D2 - D1 / r + D0 / (r * r));
    *g = 3.0L / (8.0L * r * r) * (D1 - D0 / r);
}

/* real-space functions f = coefficient of (I - rr), g = coefficient of rr  (PSEv1/Stokes.cc:348-406) */
void pse_oracle_fg_real(double r, double xi, double *f, double *g) {
    ld fw, gw, f0, g0;
    fg_wave_l(r, xi, &fw, &gw); fg_rpy(r, &f0, &g0);
    *f = (double)(f0 - fw); *g = (double)(g0 - gw);
}
/* free-space wave part by quadrature (independent of the closed form) */
void pse_oracle_fg_wave_quad(double r, double xi, double *f, double *g) {
    ld fw, gw; fg_wave_quad_l(r, xi, &fw, &gw); *f = (double)fw; *g = (double)gw;
}
void pse_oracle_fg_wave(double r, double xi, double *f, double *g) {
    ld fw, gw; fg_wave_l(r, xi, &fw, &gw); *f = (double)fw; *g = (double)gw;
}
/* self mobility (PSEv1/Stokes.cc:319), a = 1 */
double pse_oracle_self(double xi_) {
    ld xi = xi_, sp = sqrtl(PI_L);
    return (double)((1.0L + 4.0L * sp * xi * erfcl(2.0L * xi) - expl(-4.0L * xi * xi)) / (4.0L * sp * xi));
}

/* ------------------------------------------------------------------------------------------
 * Direct Ewald sum.
 * ------------------------------------------------------------------------------------------ */
typedef struct { double re, im; } cpx;
static inline cpx cmul(cpx a, cpx b) { cpx c = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; return c; }

/* real-space + self part of U = M.F:  out[3N] (+=).  rc = real cutoff.  If dense != NULL, the
 * 3N x 3N matrix is accumulated instead (row-major), and force/out are ignored. */
static void real_part(int N, const double *pos, const double *force, const double *box, double xi, double rc,
                      double *out, double *dense) {
    double Lx = box[0], Ly = box[1], Lz = box[2], xy = box[3];
    int n1 = (int)ceil(rc * sqrt(1.0 + xy * xy) / Lx) + 1, n2 = (int)ceil(rc / Ly) + 1, n3 = (int)ceil(rc / Lz) + 1;
    double self = pse_oracle_self(xi);
    #pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < N; ++i) {
        double u[3] = { 0, 0, 0 };
        if (dense) for (int a = 0; a < 3; ++a) dense[(size_t)(3 * i + a) * 3 * N + 3 * i + a] += self;
        else for (int a = 0; a < 3; ++a) u[a] = self * force[3 * i + a];
        for (int j = 0; j < N; ++j) {
            double d0[3] = { pos[3 * i] - pos[3 * j], pos[3 * i + 1] - pos[3 * j + 1], pos[3 * i + 2] - pos[3 * j + 2] };
            /* reduce to the home cell in lattice coordinates, then scan images */
            double s2 = round(d0[1] / Ly); d0[1] -= s2 * Ly; d0[0] -= s2 * xy * Ly;
            double s1 = round(d0[0] / Lx); d0[0] -= s1 * Lx;
            d0[2] -= round(d0[2] / Lz) * Lz;
            for (int a = -n1; a <= n1; ++a) for (int b = -n2; b <= n2; ++b) for (int c = -n3; c <= n3; ++c) {
                if (i == j && a == 0 && b == 0 && c == 0) continue;
                double r[3] = { d0[0] + a * Lx + b * xy * Ly, d0[1] + b * Ly, d0[2] + c * Lz };
                double r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
                if (r2 >= rc * rc || r2 == 0.0) continue;
                double rr = sqrt(r2), f, g;
                pse_oracle_fg_real(rr, xi, &f, &g);
                if (dense) {
                    for (int p = 0; p < 3; ++p) for (int q = 0; q < 3; ++q)
                        dense[(size_t)(3 * i + p) * 3 * N + 3 * j + q] += (p == q ? f : 0.0) + (g - f) * r[p] * r[q] / r2;
                } else {
                    const double *F = force + 3 * j;
                    double rdF = (r[0] * F[0] + r[1] * F[1] + r[2] * F[2]) / r2;
                    for (int p = 0; p < 3; ++p) u[p] += f * F[p] + (g - f) * rdF * r[p];
                }
            }
        }
        if (!dense) for (int a = 0; a < 3; ++a) out[3 * i + a] += u[a];
    }
}

/* wave-space part by direct structure factors; kc = |k| cutoff */
static void wave_part(int N, const double *pos, const double *force, const double *box, double xi, double kc,
                      double *out, double *dense) {
    double Lx = box[0], Ly = box[1], Lz = box[2], xy = box[3];
    double V = Lx * Ly * Lz, tp = 2.0 * M_PI;
    int m1 = (int)ceil(kc * Lx / tp) + 1, m3 = (int)ceil(kc * Lz / tp) + 1;
    int m2 = (int)ceil(kc * Ly / tp * (1.0 + fabs(xy))) + 1 + (int)ceil(fabs(xy) * m1 * Ly / Lx);
    int w1 = 2 * m1 + 1, w2 = 2 * m2 + 1, w3 = m3 + 1;
    cpx *e1 = malloc(sizeof(cpx) * (size_t)N * w1), *e2 = malloc(sizeof(cpx) * (size_t)N * w2), *e3 = malloc(sizeof(cpx) * (size_t)N * w3);
    for (int j = 0; j < N; ++j) {
        double x = pos[3 * j], y = pos[3 * j + 1], z = pos[3 * j + 2];
        /* k = m1 b1 + m2 b2 + m3 b3, b1 = 2pi(1/Lx, -xy/Lx, 0), b2 = 2pi(0,1/Ly,0), b3 = 2pi(0,0,1/Lz)  (Helper.cu:307-312) */
        double p1 = tp * (x - xy * y) / Lx, p2 = tp * y / Ly, p3 = tp * z / Lz;
        for (int m = -m1; m <= m1; ++m) { e1[(size_t)j * w1 + m + m1].re = cos(m * p1); e1[(size_t)j * w1 + m + m1].im = sin(m * p1); }
        for (int m = -m2; m <= m2; ++m) { e2[(size_t)j * w2 + m + m2].re = cos(m * p2); e2[(size_t)j * w2 + m + m2].im = sin(m * p2); }
        for (int m = 0; m <= m3; ++m)   { e3[(size_t)j * w3 + m].re = cos(m * p3); e3[(size_t)j * w3 + m].im = sin(m * p3); }
    }
    size_t nout = dense ? (size_t)9 * N * N : (size_t)3 * N;
    #pragma omp parallel
    {
        double *acc = calloc(nout, sizeof(double));
        cpx *E12 = malloc(sizeof(cpx) * N), *E = malloc(sizeof(cpx) * N);
        #pragma omp for collapse(2) schedule(dynamic, 4)
        for (int a = -m1; a <= m1; ++a) for (int b = -m2; b <= m2; ++b) {
            double kx = tp * a / Lx, ky = tp * (b / Ly - xy * a / Lx);
            if (kx * kx + ky * ky >= kc * kc) continue;
            for (int j = 0; j < N; ++j) E12[j] = cmul(e1[(size_t)j * w1 + a + m1], e2[(size_t)j * w2 + b + m2]);
            for (int c = 0; c <= m3; ++c) {
                /* half space: c > 0, or c == 0 and (b > 0 or (b == 0 and a > 0)); weight 2 */
                if (c == 0 && (b < 0 || (b == 0 && a <= 0))) continue;
                double kz = tp * c / Lz, k2 = kx * kx + ky * ky + kz * kz;
                if (k2 >= kc * kc) continue;
                double k = sqrt(k2), q = k2 / (4.0 * xi * xi), sk = sin(k) / k;
                /* Helper.cu:326 (without the 1/Ng FFT normalisation) x Mobility.cu:290, x 2/V */
                double coef = 2.0 / V * 6.0 * M_PI * (1.0 + q) * exp(-q) / k2 * sk * sk;
                double kh[3] = { kx / k, ky / k, kz / k };
                cpx S[3] = { { 0, 0 }, { 0, 0 }, { 0, 0 } };
                for (int j = 0; j < N; ++j) {
                    E[j] = cmul(E12[j], e3[(size_t)j * w3 + c]);
                    if (!dense) for (int p = 0; p < 3; ++p) { S[p].re += force[3 * j + p] * E[j].re; S[p].im -= force[3 * j + p] * E[j].im; }
                }
                if (!dense) {
                    double kSr = kh[0] * S[0].re + kh[1] * S[1].re + kh[2] * S[2].re;
                    double kSi = kh[0] * S[0].im + kh[1] * S[1].im + kh[2] * S[2].im;
                    for (int p = 0; p < 3; ++p) { S[p].re -= kh[p] * kSr; S[p].im -= kh[p] * kSi; }
                    for (int i = 0; i < N; ++i) for (int p = 0; p < 3; ++p)
                        acc[3 * i + p] += coef * (S[p].re * E[i].re - S[p].im * E[i].im);
                } else {
                    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) {
                        double cr = coef * (E[i].re * E[j].re + E[i].im * E[j].im); /* Re(E_i conj E_j) */
                        for (int p = 0; p < 3; ++p) for (int s = 0; s < 3; ++s)
                            acc[(size_t)(3 * i + p) * 3 * N + 3 * j + s] += cr * ((p == s ? 1.0 : 0.0) - kh[p] * kh[s]);
                    }
                }
            }
        }
        #pragma omp critical
        { double *dst = dense ? dense : out; for (size_t t = 0; t < nout; ++t) dst[t] += acc[t]; }
        free(acc); free(E12); free(E);
    }
    free(e1); free(e2); free(e3);
}

static void cutoffs(double xi, double tol, double *rc, double *kc) {
    double s = sqrt(-log(tol));
    *rc = 1.15 * s / xi + 2.0; *kc = 1.15 * 2.0 * xi * s;
}

/* U = M.F, pos/force/vel are [N][3] doubles.  parts: 1 = real+self, 2 = wave, 3 = both. */
int pse_oracle_mobility_direct(int N, const double *pos, const double *force, const double *box,
                               double xi, double tol, int parts, double *vel, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    double rc, kc; cutoffs(xi, tol, &rc, &kc);
    memset(vel, 0, sizeof(double) * 3 * N);
    if (parts & 1) real_part(N, pos, force, box, xi, rc, vel, NULL);
    if (parts & 2) wave_part(N, pos, force, box, xi, kc, vel, NULL);
    return 0;
}

/* dense 3N x 3N mobility matrix (row-major) for small N */
int pse_oracle_mobility_dense(int N, const double *pos, const double *box, double xi, double tol, int parts,
                              double *M, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    double rc, kc; cutoffs(xi, tol, &rc, &kc);
    memset(M, 0, sizeof(double) * 9 * (size_t)N * N);
    if (parts & 1) real_part(N, pos, NULL, box, xi, rc, NULL, M);
    if (parts & 2) wave_part(N, pos, NULL, box, xi, kc, NULL, M);
    return 0;
}

/* real-space part truncated at rcut with minimum image only -- exactly the sum the product's
 * near-field kernel performs (PSEv1/Mobility.cu:594-687: pairs with r < rcut, self term), but with
 * the closed-form functions instead of a table. Requires rcut <= half the shortest box width. */
/* One pair's term of the near-field sum.  rounded == 0: f F + (g - f) (r.F) r / r^2 in double precision.  rounded != 0: the term as the
 * build's Lanczos mat-vecs apply it (pse_amd/csrc/pse_kernels.hip, pair_coef / nb_pack: the 16-byte records of the per-step pair list
 * carry pair coefficients of single-precision accuracy): fr = f rounded to a multiple of 2^-24, s = r sqrt|h| with h = (g - f) / r^2 (the
 * root taken in single precision) rounded to 22-bit mantissas under the exponent of its largest component, term fr F + sgn(h) (s.F) s
 * accumulated in double precision.  The reference's own pair coefficients are single precision throughout (PSEv1/Mobility.cu:661-677
 * with Scalar = float); this restates the build's rounding operation by operation so that M_real^{1/2} psi can be compared at 1e-9
 * instead of 1e-7. */
static void pair_term(const double r[3], double r2, double f, double g, const double *F, int rounded, double u[3]) {
    if (!rounded) {
        double rdF = (r[0] * F[0] + r[1] * F[1] + r[2] * F[2]) / r2;
        for (int p = 0; p < 3; ++p) u[p] += f * F[p] + (g - f) * rdF * r[p];
        return;
    }
    double h = (g - f) / r2;
    double hs = (double)sqrtf((float)fabs(h));
    /* the 16-byte pair record of the device (pse_kernels.hip pair_coef): fr a signed 26-bit integer in units of 2^-24, s three signed
     * 22-bit mantissas under the exponent of its largest component */
    double sv[3] = { r[0] * hs, r[1] * hs, r[2] * hs }, s[3];
    double m = fmax(fabs(sv[0]), fmax(fabs(sv[1]), fabs(sv[2])));
    int e = 0;
    if (m > 0.0) (void)frexp(m, &e);
    if (e < -100) e = -100;
    if (e > 100) e = 100;
    for (int p = 0; p < 3; ++p) {
        double q = rint(ldexp(sv[p], 21 - e));
        if (q > 2097151.0) q = 2097151.0;
        if (q < -2097151.0) q = -2097151.0;
        s[p] = ldexp(q, e - 21);
    }
    double fq = rint(f * 16777216.0);
    if (fq > 33554431.0) fq = 33554431.0;
    if (fq < -33554431.0) fq = -33554431.0;
    double fr = fq * 5.9604644775390625e-08;
    double sd = s[0] * F[0] + s[1] * F[1] + s[2] * F[2];
    if (h < 0.0) sd = -sd;
    for (int p = 0; p < 3; ++p) u[p] += fr * F[p] + sd * s[p];
}

static int mreal_cutoff(int N, const double *pos, const double *force, const double *box,
                        double xi, double rcut, double *vel, int nthreads, int rounded) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    double Lx = box[0], Ly = box[1], Lz = box[2], xy = box[3];
    double self = pse_oracle_self(xi);
    #pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < N; ++i) {
        double u[3] = { self * force[3 * i], self * force[3 * i + 1], self * force[3 * i + 2] };
        for (int j = 0; j < N; ++j) {
            if (j == i) continue;
            double r[3] = { pos[3 * i] - pos[3 * j], pos[3 * i + 1] - pos[3 * j + 1], pos[3 * i + 2] - pos[3 * j + 2] };
            double s2 = round(r[1] / Ly); r[1] -= s2 * Ly; r[0] -= s2 * xy * Ly;
            r[0] -= round(r[0] / Lx) * Lx; r[2] -= round(r[2] / Lz) * Lz;
            /* the lattice-reduced vector is not always the shortest image when xy != 0: scan x neighbours */
            double best[3] = { r[0], r[1], r[2] }, b2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
            for (int b = -1; b <= 1; ++b) for (int a = -1; a <= 1; ++a) {
                double c[3] = { r[0] + a * Lx + b * xy * Ly, r[1] + b * Ly, r[2] };
                double c2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
                if (c2 < b2) { b2 = c2; best[0] = c[0]; best[1] = c[1]; best[2] = c[2]; }
            }
            if (b2 >= rcut * rcut || b2 == 0.0) continue;
            double f, g; pse_oracle_fg_real(sqrt(b2), xi, &f, &g);
            pair_term(best, b2, f, g, force + 3 * j, rounded, u);
        }
        vel[3 * i] = u[0]; vel[3 * i + 1] = u[1]; vel[3 * i + 2] = u[2];
    }
    return 0;
}

int pse_oracle_mreal_cutoff(int N, const double *pos, const double *force, const double *box,
                            double xi, double rcut, double *vel, int nthreads) {
    return mreal_cutoff(N, pos, force, box, xi, rcut, vel, nthreads, 0);
}
/* the same sum with the rounded pair coefficients of the build's Lanczos mat-vecs (pair_term) */
int pse_oracle_mreal_cutoff_rounded(int N, const double *pos, const double *force, const double *box,
                                double xi, double rcut, double *vel, int nthreads) {
    return mreal_cutoff(N, pos, force, box, xi, rcut, vel, nthreads, 1);
}

int pse_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* rows of the cutoff near-field sum for a subset of particles (full-size parity checks: O(nrows * N)) */
int pse_oracle_mreal_cutoff_rows(int N, const double *pos, const double *force, const double *box, double xi, double rcut,
                                 int nrows, const int *rows, double *vel, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    double Lx = box[0], Ly = box[1], Lz = box[2], xy = box[3];
    double self = pse_oracle_self(xi);
    #pragma omp parallel for schedule(dynamic, 1)
    for (int q = 0; q < nrows; ++q) {
        int i = rows[q];
        double u[3] = { self * force[3 * i], self * force[3 * i + 1], self * force[3 * i + 2] };
        for (int j = 0; j < N; ++j) {
            if (j == i) continue;
            double r[3] = { pos[3 * i] - pos[3 * j], pos[3 * i + 1] - pos[3 * j + 1], pos[3 * i + 2] - pos[3 * j + 2] };
            double s2 = round(r[1] / Ly); r[1] -= s2 * Ly; r[0] -= s2 * xy * Ly;
            r[0] -= round(r[0] / Lx) * Lx; r[2] -= round(r[2] / Lz) * Lz;
            double b2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
            if (b2 >= rcut * rcut || b2 == 0.0) continue;   /* rcut <= half the perpendicular widths: the reduced vector is the image */
            double f, g; pse_oracle_fg_real(sqrt(b2), xi, &f, &g);
            const double *F = force + 3 * j;
            double rdF = (r[0] * F[0] + r[1] * F[1] + r[2] * F[2]) / b2;
            for (int p = 0; p < 3; ++p) u[p] += f * F[p] + (g - f) * rdF * r[p];
        }
        vel[3 * q] = u[0]; vel[3 * q + 1] = u[1]; vel[3 * q + 2] = u[2];
    }
    return 0;
}
