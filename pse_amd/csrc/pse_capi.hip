// C-ABI of the PSE engine (include/pse_amd.h): handle life cycle, rocFFT plans, the step driver that replaces
// gpu_stokes_step_one / gpu_stokes_CombinedMobilityBrownian_wrap / gpu_stokes_BrealLanczos_wrap
// (PSEv1/Stokes.cu:234-365, PSEv1/Brownian.cu:772-923, PSEv1/Brownian.cu:357-765).
//
// Differences from the reference driver that are deliberate: all workspaces are allocated once
// in pse_create (the reference cudaMalloc/cudaFree's ~109 N Scalar4 every step); Lanczos scalars stay on the
// device and the host is consulted only at convergence checks; wave vectors are computed inside the scaling
// kernel; failures are returned, never exit()ed.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pse_amd.h"
#include "pse_host.h"
#include "pse_kernels.h"

using namespace pse;

static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(x)                                                                                      \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) return fail(PSE_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define FFTCHK(x)                                                                                      \
    do {                                                                                               \
        rocfft_status s_ = (x);                                                                        \
        if (s_ != rocfft_status_success) return fail(PSE_ERR_FFT, "%s failed: rocfft status %d (%s:%d)", #x, (int)s_, __FILE__, __LINE__); \
    } while (0)

constexpr int M_MAX = 100;   // Lanczos basis cap (PSEv1/Brownian.cu:397)

struct Phase {
    hipEvent_t a = nullptr, b = nullptr;
};

struct pse_handle {
    pse_params par;
    Derived d;
    Box box;
    DBox dbox;
    DGrid G;
    DCells nc;
    double cell_gamma;  // tilt bound the cell grid was sized for
    int device = 0;
    hipStream_t stream = nullptr;
    int n_max = 0;
    // sorted particle state
    unsigned *keys = nullptr, *keys_s = nullptr, *vals = nullptr, *perm = nullptr, *tag_s = nullptr;
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int *cell_off = nullptr;
    int4 *sup_s = nullptr;    // support origin of each sorted particle (node indices)
    NbList nb = {};
    bool nb_valid = false;   // the pair list matches the current sorted positions
    size_t n_cells_alloc = 0;
    double4 *pos_s = nullptr, *f_s = nullptr, *uw_s = nullptr, *ur_s = nullptr, *ub_s = nullptr, *psi_s = nullptr, *w_s = nullptr;
    // real-space table
    double *coef = nullptr;
    int n_intervals = 0;
    // grids
    double *rgrid = nullptr;     // [3][nxl][Ny][Nz]
    double2 *cgrid = nullptr;    // [3][nxl][Ny][Nzh]
    rocfft_plan plan_fwd = nullptr, plan_inv = nullptr;
    rocfft_execution_info info_fwd = nullptr, info_inv = nullptr;
    void *fft_work = nullptr;
    size_t fft_work_bytes = 0;
    // Lanczos
    double4 *V = nullptr;        // [M_MAX + 1][n_max]
    double *scal = nullptr, *partials = nullptr, *t_dev = nullptr;
    // bookkeeping
    pse_info info;
    bool timing = false;
    Phase ph[11];
    unsigned long long bytes = 0;
    int sorted_N = 0;
};

static std::once_flag g_fft_once;

template <class T>
static int dmalloc(pse_handle *h, T **p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    HIPCHK(hipMalloc((void **)p, n * sizeof(T)));
    h->bytes += n * sizeof(T);
    return 0;
}
#define TRY(x)              \
    do {                    \
        int r_ = (x);       \
        if (r_) return r_;  \
    } while (0)

static void set_dbox(pse_handle *h) {
    h->dbox = DBox{h->box.Lx, h->box.Ly, h->box.Lz, h->box.xy, 1.0 / h->box.Lx, 1.0 / h->box.Ly, 1.0 / h->box.Lz};
}

// cells of at least rcut perpendicular width for tilts up to gamma; a dimension with fewer than 3 cells uses 1
static int set_cells(pse_handle *h, double gamma) {
    const double rc = h->d.rcut;
    const double wx = h->box.Lx / std::sqrt(1.0 + gamma * gamma), wy = h->box.Ly, wz = h->box.Lz;
    if (rc > 0.5 * wx * (1 + 1e-12) || rc > 0.5 * wy * (1 + 1e-12) || rc > 0.5 * wz * (1 + 1e-12))
        return fail(PSE_ERR_INVALID, "real-space cutoff %.4f exceeds half the box width (%.4f, %.4f, %.4f at tilt %.3f): "
                    "the minimum-image near field needs rcut <= L/2; increase xi", rc, wx, wy, wz, gamma);
    auto n = [&](double w) { int c = (int)std::floor(w / rc); if (c < 3) c = 1; if (c > 1024) c = 1024; return c; };
    h->nc = DCells{n(wx), n(wy), n(wz)};
    h->cell_gamma = gamma;
    return 0;
}

static int ts(pse_handle *h, int p) { if (h->timing) HIPCHK(hipEventRecord(h->ph[p].a, h->stream)); return 0; }
static int te(pse_handle *h, int p) { if (h->timing) HIPCHK(hipEventRecord(h->ph[p].b, h->stream)); return 0; }
enum { PH_SORT, PH_SPREAD, PH_FFTF, PH_SCALE, PH_FFTI, PH_GATHER, PH_REAL, PH_LANCZOS, PH_INTEG, PH_COMM, PH_TOTAL };

static int collect_times(pse_handle *h, unsigned mask) {
    if (!h->timing) return 0;
    HIPCHK(hipStreamSynchronize(h->stream));
    double *dst[11] = {&h->info.t_sort, &h->info.t_spread, &h->info.t_fft_fwd, &h->info.t_scale, &h->info.t_fft_inv,
                       &h->info.t_gather, &h->info.t_real, &h->info.t_lanczos, &h->info.t_integrate, &h->info.t_comm,
                       &h->info.t_total};
    for (int p = 0; p < 11; ++p) {
        *dst[p] = 0.0;
        if (mask & (1u << p)) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, h->ph[p].a, h->ph[p].b) == hipSuccess) *dst[p] = ms;
        }
    }
    return 0;
}

extern "C" const char *pse_last_error(void) { return g_err.c_str(); }

static void fill_info(const Derived &d, pse_info *o) {
    memset(o, 0, sizeof *o);
    o->Nx = d.Nx; o->Ny = d.Ny; o->Nz = d.Nz; o->P = d.P;
    o->rcut = d.rcut; o->xi = d.xi; o->eta = d.eta; o->gaussm = d.gaussm; o->lambda = d.lambda;
    o->self_mobility = d.self; o->hx = d.hx; o->hy = d.hy; o->hz = d.hz;
}

extern "C" int pse_host_select_params(const pse_params *p, pse_info *info) {
    if (!p || !info) return fail(PSE_ERR_INVALID, "null argument");
    Derived d;
    std::string e = select_params(Box{p->Lx, p->Ly, p->Lz, p->xy}, p->xi, p->error, p->max_strain, p->Nx, p->Ny, p->Nz,
                                  p->P, p->rcut, d);
    if (!e.empty()) return fail(PSE_ERR_INVALID, "%s", e.c_str());
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

extern "C" int pse_destroy(pse_handle *h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    if (h->plan_fwd) rocfft_plan_destroy(h->plan_fwd);
    if (h->plan_inv) rocfft_plan_destroy(h->plan_inv);
    if (h->info_fwd) rocfft_execution_info_destroy(h->info_fwd);
    if (h->info_inv) rocfft_execution_info_destroy(h->info_inv);
    void *ptrs[] = {h->keys, h->keys_s, h->vals, h->perm, h->tag_s, h->sort_tmp, h->cell_off, h->sup_s, h->nb.j, h->nb.fh, h->nb.dx, h->nb.dy, h->nb.dz, h->nb.cnt, h->pos_s,
                    h->f_s, h->uw_s, h->ur_s, h->ub_s, h->psi_s, h->w_s, h->coef, h->rgrid, h->cgrid, h->fft_work, h->V,
                    h->scal, h->partials, h->t_dev};
    for (void *p : ptrs) if (p) hipFree(p);
    for (auto &p : h->ph) { if (p.a) hipEventDestroy(p.a); if (p.b) hipEventDestroy(p.b); }
    delete h;
    return 0;
}

static int make_plans(pse_handle *h) {
    std::call_once(g_fft_once, [] { rocfft_setup(); });
    const DGrid &G = h->G;
    // rocFFT lengths are fastest-first: z, y, x.  Real grids [3][Nx][Ny][Nz] -> half spectra [3][Nx][Ny][Nzh].
    const size_t len[3] = {(size_t)G.Nz, (size_t)G.Ny, (size_t)G.Nx};
    FFTCHK(rocfft_plan_create(&h->plan_fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                              rocfft_precision_double, 3, len, 3, nullptr));
    FFTCHK(rocfft_plan_create(&h->plan_inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                              rocfft_precision_double, 3, len, 3, nullptr));
    size_t wf = 0, wi = 0;
    FFTCHK(rocfft_plan_get_work_buffer_size(h->plan_fwd, &wf));
    FFTCHK(rocfft_plan_get_work_buffer_size(h->plan_inv, &wi));
    h->fft_work_bytes = std::max(wf, wi);
    if (h->fft_work_bytes) TRY(dmalloc(h, (char **)&h->fft_work, h->fft_work_bytes));
    FFTCHK(rocfft_execution_info_create(&h->info_fwd));
    FFTCHK(rocfft_execution_info_create(&h->info_inv));
    if (h->fft_work_bytes) {
        FFTCHK(rocfft_execution_info_set_work_buffer(h->info_fwd, h->fft_work, h->fft_work_bytes));
        FFTCHK(rocfft_execution_info_set_work_buffer(h->info_inv, h->fft_work, h->fft_work_bytes));
    }
    FFTCHK(rocfft_execution_info_set_stream(h->info_fwd, h->stream));
    FFTCHK(rocfft_execution_info_set_stream(h->info_inv, h->stream));
    return 0;
}

static int create_impl(const pse_params *p, pse_handle *h) {
    h->par = *p;
    h->box = Box{p->Lx, p->Ly, p->Lz, p->xy};
    std::string e = select_params(h->box, p->xi, p->error, p->max_strain, p->Nx, p->Ny, p->Nz, p->P, p->rcut, h->d);
    if (!e.empty()) return fail(PSE_ERR_INVALID, "%s", e.c_str());
    if (p->n_max == 0) return fail(PSE_ERR_INVALID, "n_max must be positive");
    if (p->n_slabs > 1) return fail(PSE_ERR_INVALID, "slab decomposition is driven through the pse_slab_* entry points");
    const Derived &d = h->d;
    if (p->device >= 0) { HIPCHK(hipSetDevice(p->device)); h->device = p->device; }
    else HIPCHK(hipGetDevice(&h->device));
    h->n_max = (int)p->n_max;
    set_dbox(h);
    TRY(set_cells(h, std::max(std::fabs(p->xy), p->max_strain)));
    fill_info(d, &h->info);
    h->info.ncell_x = h->nc.nx; h->info.ncell_y = h->nc.ny; h->info.ncell_z = h->nc.nz;

    DGrid &G = h->G;
    G.Nx = d.Nx; G.Ny = d.Ny; G.Nz = d.Nz; G.Nzh = d.Nz / 2 + 1; G.P = d.P;
    G.x0 = 0; G.nxl = d.Nx;
    G.hx = d.hx; G.hy = d.hy; G.hz = d.hz;
    const double c = 2.0 * d.xi * d.xi / d.eta;
    G.expfac = c;                                      // PSEv1/Brownian.cu:829
    G.prefac = (c / M_PI) * std::sqrt(c / M_PI);       // PSEv1/Brownian.cu:828

    const size_t n = h->n_max;
    TRY(dmalloc(h, &h->keys, n)); TRY(dmalloc(h, &h->keys_s, n)); TRY(dmalloc(h, &h->vals, n));
    TRY(dmalloc(h, &h->perm, n)); TRY(dmalloc(h, &h->tag_s, n));
    h->sort_tmp_bytes = sort_pairs_temp_bytes((int)n, 32);
    TRY(dmalloc(h, (char **)&h->sort_tmp, h->sort_tmp_bytes));
    // size the cell arrays for zero tilt (most cells)
    {
        const double rc = d.rcut;
        auto cnt = [&](double w) { int c2 = (int)std::floor(w / rc); if (c2 < 3) c2 = 1; if (c2 > 1024) c2 = 1024; return (size_t)c2; };
        h->n_cells_alloc = cnt(h->box.Lx) * cnt(h->box.Ly) * cnt(h->box.Lz);
    }
    TRY(dmalloc(h, &h->cell_off, h->n_cells_alloc + 1));
    TRY(dmalloc(h, &h->sup_s, n));
    {   // per-step pair list for the Lanczos mat-vecs: capacity from the mean neighbour count at full occupancy
        const double vol = h->box.Lx * h->box.Ly * h->box.Lz;
        const double nbar = (double)n / vol * 4.18879020478639 * d.rcut * d.rcut * d.rcut;
        int cap = (int)std::ceil(1.5 * nbar + 16.0);
        cap = std::max(16, std::min(cap, 256));
        const double bytes = (double)cap * (double)n * 44.0;
        if (bytes > 32e9 || n >= ((size_t)1 << 27)) cap = 0;   // too large: mat-vecs always walk the cells
        h->nb.cap = cap; h->nb.stride = n;
        if (cap > 0) {
            TRY(dmalloc(h, &h->nb.j, (size_t)cap * n)); TRY(dmalloc(h, &h->nb.fh, (size_t)cap * n));
            TRY(dmalloc(h, &h->nb.dx, (size_t)cap * n)); TRY(dmalloc(h, &h->nb.dy, (size_t)cap * n));
            TRY(dmalloc(h, &h->nb.dz, (size_t)cap * n));
        }
        TRY(dmalloc(h, &h->nb.cnt, n));
    }
    TRY(dmalloc(h, &h->pos_s, n)); TRY(dmalloc(h, &h->f_s, n)); TRY(dmalloc(h, &h->uw_s, n)); TRY(dmalloc(h, &h->ur_s, n));
    TRY(dmalloc(h, &h->ub_s, n)); TRY(dmalloc(h, &h->psi_s, n)); TRY(dmalloc(h, &h->w_s, n));

    std::vector<double> coef;
    build_realspace_table(d.xi, d.rcut, coef, h->n_intervals);
    TRY(dmalloc(h, &h->coef, coef.size()));
    HIPCHK(hipMemcpy(h->coef, coef.data(), coef.size() * sizeof(double), hipMemcpyHostToDevice));

    const size_t nr = (size_t)G.nxl * G.Ny * G.Nz, ncx = (size_t)G.nxl * G.Ny * G.Nzh;
    TRY(dmalloc(h, &h->rgrid, 3 * nr));
    TRY(dmalloc(h, &h->cgrid, 3 * ncx));
    TRY(make_plans(h));

    TRY(dmalloc(h, &h->V, (size_t)(M_MAX + 1) * n));
    TRY(dmalloc(h, &h->scal, (size_t)LZ_NSCAL)); TRY(dmalloc(h, &h->partials, (size_t)2 * LZ_NPART));
    TRY(dmalloc(h, &h->t_dev, (size_t)M_MAX + 1));
    for (auto &ph : h->ph) { HIPCHK(hipEventCreate(&ph.a)); HIPCHK(hipEventCreate(&ph.b)); }
    h->info.device_bytes = h->bytes;
    return 0;
}

extern "C" int pse_create(const pse_params *p, pse_handle **out) {
    if (!p || !out) return fail(PSE_ERR_INVALID, "null argument");
    *out = nullptr;
    pse_handle *h = new pse_handle();
    int r = create_impl(p, h);
    if (r) { std::string keep = g_err; pse_destroy(h); g_err = keep; return r; }
    *out = h;
    return 0;
}

extern "C" int pse_set_box(pse_handle *h, double Lx, double Ly, double Lz, double xy) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    if (!(Lx > 0 && Ly > 0 && Lz > 0)) return fail(PSE_ERR_INVALID, "box lengths must be positive");
    const Box old = h->box;
    h->box = Box{Lx, Ly, Lz, xy};
    // the grid, P and eta were chosen for the creation box (the reference also fixes them in setParams and only
    // recomputes wave vectors per step, PSEv1/Stokes.cu:298); lengths may change only by re-deriving h
    h->d.hx = Lx / h->d.Nx; h->d.hy = Ly / h->d.Ny; h->d.hz = Lz / h->d.Nz;
    h->G.hx = h->d.hx; h->G.hy = h->d.hy; h->G.hz = h->d.hz;
    set_dbox(h);
    const double gamma = std::max(std::fabs(xy), h->par.max_strain);
    int r = set_cells(h, gamma);
    if (!r && (size_t)h->nc.nx * h->nc.ny * h->nc.nz > h->n_cells_alloc)
        r = fail(PSE_ERR_INVALID, "box grew beyond the cell-list capacity sized at creation");
    if (r) { h->box = old; set_dbox(h); return r; }
    h->info.ncell_x = h->nc.nx; h->info.ncell_y = h->nc.ny; h->info.ncell_z = h->nc.nz;
    h->info.hx = h->d.hx; h->info.hy = h->d.hy; h->info.hz = h->d.hz;
    return 0;
}

extern "C" int pse_set_stream(pse_handle *h, void *stream) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    h->stream = (hipStream_t)stream;
    FFTCHK(rocfft_execution_info_set_stream(h->info_fwd, h->stream));
    FFTCHK(rocfft_execution_info_set_stream(h->info_inv, h->stream));
    return 0;
}
extern "C" int pse_set_timing(pse_handle *h, int enabled) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    h->timing = enabled != 0;
    return 0;
}
extern "C" int pse_get_info(pse_handle *h, pse_info *info) {
    if (!h || !info) return fail(PSE_ERR_INVALID, "null argument");
    *info = h->info;
    return 0;
}

// ---- phases ---------------------------------------------------------------------------------------------------
static int check_n(pse_handle *h, unsigned N) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    if (N == 0 || (int)N > h->n_max) return fail(PSE_ERR_INVALID, "N = %u outside (0, n_max = %d]", N, h->n_max);
    HIPCHK(hipSetDevice(h->device));
    return 0;
}

// bin + sort + gather into cell order (positions change every step, so this runs every call)
static int prepare(pse_handle *h, const double4 *pos, const double4 *vec, const unsigned *group, int N) {
    TRY(ts(h, PH_SORT));
    const size_t ncell = (size_t)h->nc.nx * h->nc.ny * h->nc.nz;
    int bits = 1;
    while (((size_t)1 << bits) < ncell) ++bits;
    launch_cell_keys(pos, group, N, h->dbox, h->nc, h->keys, h->vals, h->stream);
    sort_pairs(h->sort_tmp, h->sort_tmp_bytes, h->keys, h->keys_s, h->vals, h->perm, N, bits, h->stream);
    launch_permute(pos, vec, group, h->perm, h->keys_s, N, h->dbox, h->pos_s, h->f_s, h->tag_s, (int)ncell,
                   h->cell_off, h->stream);
    h->sorted_N = N;
    h->nb_valid = false;
    TRY(te(h, PH_SORT));
    HIPCHK(hipGetLastError());
    return 0;
}

// wave-space part: spread -> FFT -> scale (+ noise) -> inverse FFT -> gather  (PSEv1/Brownian.cu:836-872)
static int wave(pse_handle *h, int N, bool noise, double kT, double dt, unsigned timestep) {
    const DGrid &G = h->G;
    const size_t nr = (size_t)G.nxl * G.Ny * G.Nz, ncx = (size_t)G.nxl * G.Ny * G.Nzh;
    double *gx = h->rgrid, *gy = h->rgrid + nr, *gz = h->rgrid + 2 * nr;
    double2 *cx = h->cgrid, *cy = h->cgrid + ncx, *cz = h->cgrid + 2 * ncx;
    TRY(ts(h, PH_SPREAD));
    if (spread_needs_zero(G)) HIPCHK(hipMemsetAsync(h->rgrid, 0, 3 * nr * sizeof(double), h->stream));
    launch_spread(h->pos_s, h->f_s, h->sup_s, N, h->cell_off, h->nc, gx, gy, gz, G, h->dbox, h->stream);
    TRY(te(h, PH_SPREAD));
    TRY(ts(h, PH_FFTF));
    { void *in[1] = {h->rgrid}, *out[1] = {h->cgrid}; FFTCHK(rocfft_execute(h->plan_fwd, in, out, h->info_fwd)); }
    TRY(te(h, PH_FFTF));
    TRY(ts(h, PH_SCALE));
    ScaleArgs a;
    a.xi = h->d.xi; a.eta = h->d.eta; a.noise = noise ? 1 : 0;
    a.noise_fac = noise ? std::sqrt(2.0 * kT / dt / (G.hx * G.hy * G.hz)) : 0.0;   // PSEv1/Brownian.cu:197
    a.seed = h->par.seed; a.timestep = timestep; a.transposed = 0; a.y0 = 0; a.nyl = G.Ny;
    launch_scale(cx, cy, cz, G, h->dbox, a, h->stream);
    TRY(te(h, PH_SCALE));
    TRY(ts(h, PH_FFTI));
    { void *in[1] = {h->cgrid}, *out[1] = {h->rgrid}; FFTCHK(rocfft_execute(h->plan_inv, in, out, h->info_inv)); }
    TRY(te(h, PH_FFTI));
    TRY(ts(h, PH_GATHER));
    launch_gather(h->pos_s, N, gx, gy, gz, G, h->dbox, h->uw_s, h->stream);
    TRY(te(h, PH_GATHER));
    HIPCHK(hipGetLastError());
    return 0;
}

// near-field mat-vec; build_list: also record the pair list so later mat-vecs of this step can reuse it
static int real(pse_handle *h, const double4 *vec_s, double4 *out_s, int N, bool build_list) {
    int mode = MREAL_CELLS;
    if (h->nb.cap > 0) {
        if (h->nb_valid) mode = MREAL_USE_LIST;
        else if (build_list) mode = MREAL_BUILD_LIST;
    }
    launch_mreal(h->pos_s, vec_s, out_s, N, h->cell_off, h->dbox, h->nc, h->d.rcut, h->d.self, h->coef, h->nb, mode,
                 h->stream);
    if (mode == MREAL_BUILD_LIST) h->nb_valid = true;
    return 0;
}

// M_real^{1/2} psi by Lanczos (PSEv1/Brownian.cu:357-765): psi_s in sorted order -> out_s = scale |psi| V t.
static int lanczos(pse_handle *h, const double4 *psi_s, double4 *out_s, int N, double tol, double scale, int *m_io) {
    const size_t stride = h->n_max;
    int m_in = m_io ? *m_io : 2;
    if (m_in < 1) m_in = 1;
    if (m_in > M_MAX) m_in = M_MAX;
    launch_lz_start(psi_s, h->V, nullptr, h->scal, h->partials, N, h->stream);
    std::vector<double> sc(LZ_NSCAL), t_prev, t_cur;
    int done = 0;                         // iterations launched so far
    int target = std::max(m_in, 2);       // first convergence check is at m = max(m_in, 2)   (Brownian.cu:465-466,606)
    int m_final = 0;
    double stepnorm = 1.0;
    int checked = 0;                      // largest m whose t has been evaluated
    while (true) {
        for (; done < target; ++done) {
            double4 *Vj = h->V + (size_t)done * stride;
            TRY(real(h, Vj, h->w_s, N, true));
            launch_lz_iter(h->w_s, Vj, done > 0 ? h->V + (size_t)(done - 1) * stride : nullptr,
                           h->V + (size_t)(done + 1) * stride, done, h->scal, h->partials, N, h->stream);
        }
        HIPCHK(hipMemcpyAsync(sc.data(), h->scal, LZ_NSCAL * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        const double *alpha = &sc[LZ_ALPHA], *beta = &sc[LZ_BETA];
        if (!(sc[LZ_NORM] > 0.0) || !std::isfinite(sc[LZ_NORM])) {   // psi == 0 -> result 0
            HIPCHK(hipMemsetAsync(out_s, 0, (size_t)N * sizeof(double4), h->stream));
            if (m_io) *m_io = m_in;
            h->info.lanczos_m = 0; h->info.lanczos_matvecs = done; h->info.lanczos_stepnorm = 0.0;
            return 0;
        }
        // walk m upward exactly as the reference's while loop does, one vector at a time
        for (int m = std::max(checked + 1, std::max(m_in - 1, 1)); m <= done && !m_final; ++m) {
            if (!std::isfinite(alpha[m - 1]) || !std::isfinite(beta[m]))
                return fail(PSE_ERR_NUMERIC, "Lanczos produced a non-finite coefficient at iteration %d", m - 1);
            if (!lanczos_sqrt_e1(m, alpha, beta, t_cur))
                return fail(PSE_ERR_NUMERIC, "tridiagonal eigen-solve failed at m = %d", m);
            if (beta[m] < 1e-8) { m_final = m; stepnorm = 0.0; break; }          // invariant subspace (Brownian.cu:503)
            if (!t_prev.empty() && (int)t_prev.size() == m - 1) {
                double s2 = t_cur[m - 1] * t_cur[m - 1];
                for (int q = 0; q < m - 1; ++q) s2 += (t_cur[q] - t_prev[q]) * (t_cur[q] - t_prev[q]);
                stepnorm = std::sqrt(s2 / alpha[0]);                               // Brownian.cu:719-724; psi.M.psi/|psi|^2 = alpha_0
                if (stepnorm <= tol || m >= M_MAX) { m_final = m; break; }
            }
            t_prev = t_cur;
            checked = m;
        }
        if (m_final) break;
        target = std::min(M_MAX, done + std::max(2, done / 4));
        if (done >= M_MAX) { m_final = M_MAX; break; }
    }
    if ((int)t_cur.size() != m_final) {
        if (!lanczos_sqrt_e1(m_final, &sc[LZ_ALPHA], &sc[LZ_BETA], t_cur))
            return fail(PSE_ERR_NUMERIC, "tridiagonal eigen-solve failed at m = %d", m_final);
    }
    HIPCHK(hipMemcpyAsync(h->t_dev, t_cur.data(), m_final * sizeof(double), hipMemcpyHostToDevice, h->stream));
    launch_basis_combine(h->V, stride, h->t_dev, m_final, h->scal, scale, 1, out_s, N, h->stream);   // Brownian.cu:716,739
    HIPCHK(hipStreamSynchronize(h->stream));   // t_cur is host memory that goes out of scope
    if (m_io) *m_io = m_final;
    h->info.lanczos_m = m_final; h->info.lanczos_matvecs = done; h->info.lanczos_stepnorm = stepnorm;
    return 0;
}

static int velocity(pse_handle *h, const double4 *pos, const double4 *force, double4 *vel, const unsigned *group, int N,
                    int parts, double kT, double dt, unsigned timestep, int *m_io, unsigned *mask) {
    TRY(prepare(h, pos, force, group, N));
    *mask |= 1u << PH_SORT;
    const bool noise = kT > 0.0;
    if (parts & 2) {
        TRY(wave(h, N, noise, kT, dt, timestep));
        *mask |= (1u << PH_SPREAD) | (1u << PH_FFTF) | (1u << PH_SCALE) | (1u << PH_FFTI) | (1u << PH_GATHER);
    }
    if (parts & 1) {
        TRY(ts(h, PH_REAL));
        TRY(real(h, h->f_s, h->ur_s, N, noise));
        TRY(te(h, PH_REAL));
        *mask |= 1u << PH_REAL;
    }
    if (noise) {
        TRY(ts(h, PH_LANCZOS));
        launch_psi(h->psi_s, h->tag_s, N, h->par.seed, timestep, h->stream);
        TRY(lanczos(h, h->psi_s, h->ub_s, N, h->d.error, std::sqrt(2.0 * kT / dt), m_io));
        TRY(te(h, PH_LANCZOS));
        *mask |= 1u << PH_LANCZOS;
    }
    launch_scatter_sum((parts & 2) ? h->uw_s : nullptr, (parts & 1) ? h->ur_s : nullptr, noise ? h->ub_s : nullptr,
                       h->tag_s, N, vel, h->stream);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int pse_mobility(pse_handle *h, const pse_double4 *pos, const pse_double4 *force, pse_double4 *vel,
                            const unsigned *group, unsigned N, int parts) {
    TRY(check_n(h, N));
    if (!pos || !force || !vel) return fail(PSE_ERR_INVALID, "null array");
    if (!(parts & 3)) return fail(PSE_ERR_INVALID, "parts must select real (1), wave (2) or both (3)");
    unsigned mask = 1u << PH_TOTAL;
    TRY(ts(h, PH_TOTAL));
    TRY(velocity(h, (const double4 *)pos, (const double4 *)force, (double4 *)vel, group, (int)N, parts, 0.0, 1.0, 0,
                 nullptr, &mask));
    TRY(te(h, PH_TOTAL));
    return collect_times(h, mask);
}

extern "C" int pse_brownian_velocity(pse_handle *h, const pse_double4 *pos, const pse_double4 *force, pse_double4 *vel,
                                     const unsigned *group, unsigned N, double kT, double dt, unsigned timestep,
                                     int *lanczos_m) {
    TRY(check_n(h, N));
    if (!pos || !force || !vel) return fail(PSE_ERR_INVALID, "null array");
    if (kT < 0 || !(dt > 0)) return fail(PSE_ERR_INVALID, "need kT >= 0 and dt > 0");
    unsigned mask = 1u << PH_TOTAL;
    TRY(ts(h, PH_TOTAL));
    TRY(velocity(h, (const double4 *)pos, (const double4 *)force, (double4 *)vel, group, (int)N, 3, kT, dt, timestep,
                 lanczos_m, &mask));
    TRY(te(h, PH_TOTAL));
    return collect_times(h, mask);
}

extern "C" int pse_step(pse_handle *h, pse_double4 *pos, pse_double4 *vel, pse_double3 *accel, pse_int3 *image,
                        const pse_double4 *net_force, const unsigned *group, unsigned N, double kT, double dt,
                        unsigned timestep, double shear_rate, int *lanczos_m) {
    TRY(check_n(h, N));
    if (!pos || !vel || !accel || !image || !net_force) return fail(PSE_ERR_INVALID, "null array");
    if (kT < 0 || !(dt > 0)) return fail(PSE_ERR_INVALID, "need kT >= 0 and dt > 0");
    unsigned mask = (1u << PH_TOTAL) | (1u << PH_INTEG);
    TRY(ts(h, PH_TOTAL));
    TRY(velocity(h, (const double4 *)pos, (const double4 *)net_force, (double4 *)vel, group, (int)N, 3, kT, dt, timestep,
                 lanczos_m, &mask));
    TRY(ts(h, PH_INTEG));
    launch_integrate((double4 *)pos, (const double4 *)vel, (double3 *)accel, (int3 *)image, (const double4 *)net_force,
                     group, (int)N, h->dbox, dt, shear_rate, h->stream);
    TRY(te(h, PH_INTEG));
    TRY(te(h, PH_TOTAL));
    HIPCHK(hipGetLastError());
    return collect_times(h, mask);
}

extern "C" int pse_sqrt_mreal(pse_handle *h, const pse_double4 *pos, const pse_double4 *psi, pse_double4 *out,
                              const unsigned *group, unsigned N, double tol, int *lanczos_m) {
    TRY(check_n(h, N));
    if (!pos || !psi || !out) return fail(PSE_ERR_INVALID, "null array");
    TRY(prepare(h, (const double4 *)pos, (const double4 *)psi, group, (int)N));   // f_s <- psi in sorted order
    HIPCHK(hipMemcpyAsync(h->psi_s, h->f_s, (size_t)N * sizeof(double4), hipMemcpyDeviceToDevice, h->stream));
    TRY(lanczos(h, h->psi_s, h->ub_s, (int)N, tol, 1.0, lanczos_m));
    launch_scatter_sum(h->ub_s, nullptr, nullptr, h->tag_s, (int)N, (double4 *)out, h->stream);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int pse_random_psi(pse_handle *h, pse_double4 *psi, const unsigned *group, unsigned N, unsigned timestep) {
    TRY(check_n(h, N));
    if (!psi) return fail(PSE_ERR_INVALID, "null array");
    // tags in group order, no sorting needed: reuse the key kernel's identity permutation
    std::vector<unsigned> tags(N);
    if (group) HIPCHK(hipMemcpy(tags.data(), group, N * sizeof(unsigned), hipMemcpyDeviceToHost));
    else for (unsigned i = 0; i < N; ++i) tags[i] = i;
    HIPCHK(hipMemcpy(h->tag_s, tags.data(), N * sizeof(unsigned), hipMemcpyHostToDevice));
    launch_psi(h->psi_s, h->tag_s, (int)N, h->par.seed, timestep, h->stream);
    launch_scatter_sum(h->psi_s, nullptr, nullptr, h->tag_s, (int)N, (double4 *)psi, h->stream);
    HIPCHK(hipStreamSynchronize(h->stream));
    h->sorted_N = 0;
    return 0;
}

extern "C" int pse_eval_realspace(pse_handle *h, const double *r_host, int n, double *f_host, double *g_host) {
    if (!h || !r_host || !f_host || !g_host || n <= 0) return fail(PSE_ERR_INVALID, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    for (int i = 0; i < n; ++i)
        if (!(r_host[i] > 0.0) || r_host[i] * RS_PER_UNIT >= h->n_intervals)
            return fail(PSE_ERR_INVALID, "r[%d] = %g outside the table range (0, %g)", i, r_host[i], (double)h->n_intervals / RS_PER_UNIT);
    double *buf = nullptr;
    HIPCHK(hipMalloc((void **)&buf, (size_t)3 * n * sizeof(double)));
    hipError_t e = hipMemcpy(buf, r_host, n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_eval_fg(buf, n, h->coef, buf + n, buf + 2 * n, h->stream);
        e = hipStreamSynchronize(h->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(f_host, buf + n, n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(g_host, buf + 2 * n, n * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(buf);
    if (e != hipSuccess) return fail(PSE_ERR_HIP, "pse_eval_realspace: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int pse_debug_copy_grid(pse_handle *h, int stage, double *host_out) {
    if (!h || !host_out) return fail(PSE_ERR_INVALID, "bad argument");
    (void)stage;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t nr = (size_t)h->G.nxl * h->G.Ny * h->G.Nz;
    HIPCHK(hipMemcpy(host_out, h->rgrid, 3 * nr * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}
