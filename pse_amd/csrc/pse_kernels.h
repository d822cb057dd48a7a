// Launch wrappers of the HIP kernels (defined in pse_kernels.hip).  All take the stream explicitly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pse_device.h"

namespace pse {

// ---- particle binning ------------------------------------------------------------------------------------
void launch_cell_keys(const double4 *pos, const unsigned *group, int N, DBox box, DCells nc, unsigned *keys,
                      unsigned *vals, hipStream_t s);
size_t sort_pairs_temp_bytes(int N, int end_bit);
void sort_pairs(void *temp, size_t temp_bytes, const unsigned *keys_in, unsigned *keys_out, const unsigned *vals_in,
                unsigned *vals_out, int N, int end_bit, hipStream_t s);
// pos_s[i] = wrapped position of particle perm[i] (w = tag as double), vec_s[i] = vec[tag].xyz; cell bounds
void launch_permute(const double4 *pos, const double4 *vec, const unsigned *group, const unsigned *perm,
                    const unsigned *keys_sorted, int N, DBox box, double4 *pos_s, double4 *vec_s, unsigned *tag_s,
                    int ncell, int *cell_off, hipStream_t s);
void launch_permute_vec(const double4 *vec, const unsigned *tag_s, int N, double4 *vec_s, hipStream_t s);

// ---- near field (K9) -------------------------------------------------------------------------------------
// per-step pair list in ELL layout (slot-major): entry (slot, i) at slot*stride + i
struct NbList {
    unsigned *j;
    double *f;            // transverse coefficient f(r)
    double *dx, *dy, *dz; // sqrt(|h|) (r_i - r_j), h = (g - f)/r^2; sign(h) is bit 31 of j
    int *cnt;             // true neighbour count per particle (may exceed cap: that particle falls back to the cells)
    int cap;
    size_t stride;
};
enum { MREAL_CELLS = 0, MREAL_BUILD_LIST = 1, MREAL_USE_LIST = 2 };
// out = M_real . vec (+ self). mode: cells only / cells + write the pair list / use the pair list
// rows [lo, hi) of the mat-vec (the whole vector is read; multi-GPU ranks each take a row range)
void launch_mreal(const double4 *pos_s, const double4 *vec_s, double4 *out_s, int lo, int hi, const int *cell_off,
                  DBox box, DCells nc, double rcut, double self, const double *coef, NbList nb, int mode, hipStream_t s);

// ---- far field (K2-K8) -----------------------------------------------------------------------------------
// scratch of the fast far-field path (rebuilt every call): support offsets and the per-particle separable weights
struct SpreadWork {
    double4 *d0_s;              // [N] offset of the support origin from the particle, grid units
    double *wtab;               // [N][P^2 + P] (+ padding)
};
// true if the caller must zero the grids first (atomic fallback: P outside 4..8 or a grid smaller than two tiles)
bool spread_needs_zero(const DGrid &G);
void launch_spread(const double4 *pos_s, const double4 *f_s, int4 *sup_s, int N, const int *cell_off, DCells nc,
                   double *gx, double *gy, double *gz, DGrid G, DBox box, SpreadWork w, hipStream_t s);
struct ScaleArgs {
    double xi, eta;
    int noise;               // add k-space Brownian noise (K6)
    double noise_fac;        // sqrt(2 kT / (dt h^3))
    uint32_t seed, timestep;
    int transposed;          // layout [Nx][ny_local][Nzh] (slab mode, after the transpose) instead of [nx_local][Ny][Nzh]
    int y0, nyl;             // slab of y rows in transposed layout
};
void launch_scale(double2 *X, double2 *Y, double2 *Z, DGrid G, DBox box, ScaleArgs a, hipStream_t s);
// fused forward-x FFT + scale + inverse-x FFT on [3][Nx][Ny][Nzh] (Nx a power of two, 16..512); tw[m] = exp(-2 pi i m/Nx)
bool xfuse_supported(int Nx);
void launch_xfft_scale(double2 *X, double2 *Y, double2 *Z, DGrid G, DBox box, ScaleArgs a, const double2 *tw, hipStream_t s);
// the fast path reads sup_s / wtab written by launch_spread of the same step
void launch_gather(const double4 *pos_s, const int4 *sup_s, const double *wtab, const int *cell_off, DCells nc, int N,
                   const double *gx, const double *gy, const double *gz, DGrid G, DBox box, double4 *u_s, hipStream_t s);

// ---- slab decomposition helpers
void launch_slab_pack(double2 *cgrid, double2 *buf, int nxl, int Ny, int Nzh, int nyl, int unpack, hipStream_t s);
void launch_add_inplace(double *a, const double *b, size_t n, hipStream_t s);

// ---- vector kernels (K10-K14) ----------------------------------------------------------------------------
void launch_psi(double4 *psi_s, const unsigned *tag_s, int N, uint32_t seed, uint32_t timestep, hipStream_t s);
// scal layout (device doubles): [0..127] alpha, [128..255] beta, [256] psi norm, [257] scratch
constexpr int LZ_ALPHA = 0, LZ_BETA = 128, LZ_NORM = 256, LZ_TMP = 257, LZ_NSCAL = 264;
constexpr int LZ_NPART = 1024;  // partial-sum slots
void launch_lz_start(const double4 *psi_s, double4 *V0, double4 *partial_ws, double *scal, double *partials, int N,
                     hipStream_t s);
// one Lanczos iteration j given w = M.V[j] in `w`: fills alpha[j], beta[j+1], V[j+1]
void launch_lz_iter(double4 *w, const double4 *Vj, const double4 *Vjm1, double4 *Vjp1, int j, double *scal,
                    double *partials, int N, hipStream_t s);
// out_s = scale * sum_q t[q] V[q]
// rows [lo, hi)
void launch_basis_combine(const double4 *V, size_t stride, const double *t_dev, int m, const double *scal,
                          double scale, int use_norm, double4 *out_s, int lo, int hi, hipStream_t s);
// distributed Lanczos iteration on the own rows: (a) w -= beta v_{j-1}, partial v_j.w and w.w -> scal[LZ_TMP..+1];
// [all-reduce of the two scalars by the caller]; (c) alpha, beta, V[j+1]
void launch_lzd_a(double4 *w, const double4 *Vj, const double4 *Vjm1, int j, double *scal, double *partials, int lo, int hi,
                  hipStream_t s);
void launch_lzd_c(const double4 *w, const double4 *Vj, double4 *Vjp1, int j, double *scal, int lo, int hi, hipStream_t s);
void launch_sum_rows(const double4 *a, const double4 *b, const double4 *c, double4 *out, int lo, int hi, hipStream_t s);
void launch_pick(const int *cell_off, const int *idx, int n, int *out, hipStream_t s);
// vel[tag].xyz = a + b + c (each may be null), keep w
void launch_scatter_sum(const double4 *a, const double4 *b, const double4 *c, const unsigned *tag_s, int N,
                        double4 *vel, hipStream_t s);
void launch_integrate(double4 *pos, const double4 *vel, double3 *accel, int3 *image, const double4 *force,
                      const unsigned *group, int N, DBox box, double dt, double shear_rate, hipStream_t s);
void launch_eval_fg(const double *r, int n, const double *coef, double *f, double *g, hipStream_t s);

}  // namespace pse
