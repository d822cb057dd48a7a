// C-ABI of the PSE engine (include/pse_amd.h): handle life cycle, rocFFT plans, the step driver that replaces
// gpu_stokes_step_one / gpu_stokes_CombinedMobilityBrownian_wrap / gpu_stokes_BrealLanczos_wrap
// (PSEv1/Stokes.cu:234-365, PSEv1/Brownian.cu:772-923, PSEv1/Brownian.cu:357-765).
//
// Differences from the reference driver that are deliberate: all workspaces are allocated once
// in pse_create (the reference cudaMalloc/cudaFree's ~109 N Scalar4 every step); Lanczos scalars stay on the
// device and the host is consulted only at convergence checks; wave vectors are computed inside the scaling
// kernel; failures are returned, never exit()ed.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <rocfft/rocfft.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/pse_amd.h"
#include "pse_err.h"
#include "pse_host.h"
#include "pse_kernels.h"
#include "pse_local.h"

using namespace pse;

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

#define TRY(x)              \
    do {                    \
        int r_ = (x);       \
        if (r_) return r_;  \
    } while (0)

constexpr int M_MAX = 100;   // Lanczos basis cap (PSEv1/Brownian.cu:397)
constexpr int PH_COUNT_ = 13;   // timed phases (enum PH_* below)

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
    hipStream_t stream = nullptr;    // main chain: sort, near field, Lanczos, update (caller-visible ordering)
    hipStream_t wstream = nullptr;   // wave-space chain; == stream unless the two chains overlap (single GPU)
    hipStream_t side = nullptr;      // non-blocking stream behind wstream (an in-process team shares one among its members)
    hipStream_t side_owned = nullptr;
    int *bounds_host = nullptr;      // pinned: slab row boundaries on their way back from the device
    hipEvent_t ev_bounds = nullptr;
    bool bounds_pending = false;
    // developer switches, read from the environment ONCE, in pse_create (nothing on the call path consults the environment)
    struct Tuning {
        int cell_bz = 6;          // PSE_CELL_BZ: height of the z blocks of the cell storage order (0: plain x, y, z order)
        double skin = 0.4;        // PSE_SKIN: r_buff of the neighbour list kept across calls (0: off)
        int overlap = 1;          // PSE_OVERLAP: 1 (default since round 6) two chains for every call, 0 only for kT = 0, -1 never
        bool no_xfuse = false;    // PSE_NO_XFUSE: rocFFT for the x pass
        int own_y_pow2 = 1;       // PSE_OWN_Y_POW2=0: rocFFT's 2-D transforms at Ny = 256 instead of its 1-D z pass + k_yfft_regs (A/B)
        int own_y = 1;            // PSE_OWN_Y=0: rocFFT's 2-D (y, z) transforms also where the own y pass applies
        int own_z = 1;            // PSE_OWN_Z=0: rocFFT's 1-D z transforms also where k_zfft_rows applies (A/B)
        int yslab_regs = 1;       // PSE_YSLAB_REGS=0: a slab rank's y pass at Ny = 256, 512 by k_fft_cols instead of k_yfft_regs (A/B)
        int yfft_kb = 4;          // PSE_YFFT_KB: kz columns per workgroup of the own y pass (2, 4, 8)
        int wave_mode = 0;        // PSE_WAVE_MODE: 0 automatic, 1 slab, 2 replicated
        int spread_tz = 0, spread_nw = 0;   // PSE_SPREAD_TZ, PSE_SPREAD_NW
        int gather_bz = 0;        // PSE_GATHER_BZ=1|2: bins along z per gather workgroup (0: by the particles per bin)
        int xmix_runtime = 0;     // PSE_XMIX=1: runtime radix plan of the mixed x pass also where a compile-time plan exists
        int xfft_small_wide = 0;  // PSE_XFFT_SMALL_KB=8: the eight-column x pass also on small grids
        bool verbose = false;     // PSE_VERBOSE
        bool team_sstep = true;      // PSE_TEAM_SSTEP=0: teams run one Lanczos iteration per exchange (default: two, see lanczos_team)
        int team_sched[3] = {1, 2, 3};   // PSE_TEAM_SCHED=a,b,c: far-field exchange k of a team step is issued before Lanczos exchange sched[k]
        int vq = 1;                  // PSE_VQ=0: the pair-list mat-vec gathers the neighbours' rows as doubles (two gathers per pair) (A/B)
        int place_trials = 10;       // PSE_PLACE_TRIALS=K: the two grids are allocated up to K times and the pair the x + inverse y + z passes run fastest on is kept (0, 1: off)
        int lz_extra = 2;            // PSE_LANCZOS_EXTRA: iterations a queue-only Brownian call queues beyond the starting count (gated on the device-side decision)
    } tun;
    DCells bidx_nc = {0, 0, 0};      // cell grid the boundary-cell indices on the device belong to
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool overlap_all = false;   // fork also for Brownian steps (PSE_OVERLAP=1)
    bool side_on = false;       // this call runs the wave chain on the side stream
    hipEvent_t ev_scal = nullptr;   // the Lanczos scalars have reached the pinned host buffer
    int n_max = 0, n_pad = 0;
    // sorted particle state
    unsigned *keys = nullptr, *keys_s = nullptr, *vals = nullptr, *perm = nullptr, *tag_s = nullptr;
    void *sort_tmp = nullptr;
    int *cell_cnt = nullptr;
    size_t sort_tmp_bytes = 0;
    int *cell_off = nullptr;
    int *cnt_block = nullptr; // [far-field bin counts | the two flags of the kept neighbour list | cell counts]: zeroed by ONE memset per call
    size_t cnt_bins = 0;      // ints of the bin counts (incl. the sentinel)
    SpreadWork sw = {};       // far-field bins and the bin-ordered particle records (origins, prefac * force, separable weights)
    NbList nb = {};
    bool nb_valid = false;   // the pair list matches the current sorted positions
    bool lz_last_queued = false;   // the most recent Brownian call took its Lanczos decision on the device (pse_get_info then reads the host mirror)
    // neighbour list kept across steps (HOOMD's NeighborList with r_buff and a distance check every step, PSEv1/integrate.py:60,79)
    double skin = 0.0;           // r_buff; 0: cell walk every step
    double skin_max = 0.0;       // what the cell grid and the list capacity were sized for
    VerletList vl = {};
    double4 *pos_build = nullptr;   // sorted positions at the build
    int *flags_host = nullptr;      // pinned copy of vl.flags
    bool vl_valid = false;       // the list on the device belongs to the current order (perm) and box
    bool vl_use = false;         // this call runs on the kept list (no sort, no cell walk)
    bool vl_pending = false;     // this call's first cell pass writes the list
    int vl_N = 0; const unsigned *vl_group = nullptr; Box vl_box = {};
    bool nc_wide = false;        // the cell grid in use is the one sized for rcut + skin_max
    unsigned long long nlist_builds = 0, nlist_reuses = 0;
    // A list that is never reused costs a wider cell grid and its own writes at every build: after two builds in a row whose
    // list failed its first distance check (particles diffuse more than r_buff / 2 per call) the list is suspended for a while
    // -- builds then run exactly as with r_buff = 0 -- and tried again, with the pause doubling while it keeps failing.
    // Kept per kind of call -- [1]: integrating steps (pse_step), [0]: evaluations (pse_mobility, pse_brownian_velocity, ...): a
    // time-stepping loop that outruns the skin must not switch the list off for evaluations repeated at fixed positions.
    int vl_reused_since_build = 0, vl_misses[2] = {0, 0}, vl_suspend_left[2] = {0, 0}, vl_suspend_len[2] = {32, 32};
    int vl_kind = 0;             // kind of the call being prepared
    bool pv_is_f = false;        // the vector half of pv mirrors f_s (as the permute wrote it)
    bool w_is_mpsi = false;  // w_s already holds M_real psi_s (delivered by the pass that built the pair list)
    bool sums0_done = false; // ... and scal[LZ_TMP ..] the sums psi.psi, psi.M psi of Lanczos iteration 0
    CombineSink sink = {};   // where the final Lanczos combination of this call sends its rows (velocity() sets it; off: ub_s)
    bool tail_done = false;  // ... and it did: the sum of the three contributions has reached its destination
    bool async_mode = false; // pse_set_async: deterministic evaluations queue their work and return -- no flag read-back, capturable
    bool gated = false;      // this call decides between the kept list and a rebuild ON THE DEVICE: both chains are queued (gate_word)
    int *gate_word = nullptr;   // (cnt_block): flags[0] | flags[1] of this call, read by the kernels of both chains
    size_t n_cells_alloc = 0;
    float4 *posf_s = nullptr;   // single-precision copy of pos_s (cutoff pre-filter of the near field)
    double2 *pv = nullptr;      // [N][3] packed (position, force) records gathered by the drain of the near-field passes
    double4 *pos_s = nullptr, *f_s = nullptr, *uw_s = nullptr, *ur_s = nullptr, *ub_s = nullptr, *psi_s = nullptr, *w_s = nullptr;
    // real-space table
    double *coef = nullptr;
    int n_intervals = 0;
    // grids
    double *rgrid = nullptr;     // [3][nxl][Ny][Nz]
    double2 *cgrid = nullptr;    // [3][nxl][Ny][Nzh]
    rocfft_plan plan_fwd = nullptr, plan_inv = nullptr;      // single GPU: 3-D real transforms, batch 3
    rocfft_plan plan_x_fwd = nullptr, plan_x_inv = nullptr;  // slab mode: 1-D complex transforms along x (strided, in place)
    rocfft_execution_info info_fwd = nullptr, info_inv = nullptr;
    // slab decomposition (n_slabs > 1): x planes [x0, x0+nxl) of the real grid, y rows [y0, y0+nyl) of the transposed spectrum
    int n_slabs = 1, slab_rank = 0, nyl = 0, y0 = 0;
    double2 *sendbuf = nullptr, *recvbuf = nullptr;          // [3][n_slabs][nxl][nyl][Nzh] each
    int *d_bidx = nullptr, *d_bounds = nullptr;              // slab mode: cell indices / row offsets of the cell-slab boundaries
    std::vector<int> row_lo, first_end, last_begin;          // per rank: own rows [row_lo[r], row_lo[r+1]), first / last cell layer
    std::vector<int> first2_end, last2_begin;                // ... and its first / last TWO cell layers (two-step Lanczos of a team)
    double4 *w2_s = nullptr, *u_s = nullptr;                 // two-step Lanczos: w2 = M M v_j, u = M v_{j-1}
    double *sums_all = nullptr;                              // [n_slabs][LZ_NGRAM]: every rank's partial Lanczos sums (they travel with the ghost rows)
    double4 *utot_s = nullptr;                               // slab mode: summed velocity of the own rows, all-gathered
    bool xfuse = false;                                      // power-of-two Nx: fused x pass (k_xfft_scale)
    bool own_y = false;                                      // y transforms by k_fft_cols, rocFFT does the z transforms only
    bool own_y_slab = false;                                 // ... on a slab rank, with the all-to-all block layout as its output / input
    int lz_extra = -1;           // pse_set_lanczos_extra: gated iterations a queue-only call queues beyond its starting count (-1: PSE_LANCZOS_EXTRA)
    void *vq = nullptr;          // [n] 16-byte mirror of the newest Lanczos vector (single GPU; k_lz_update writes it, the pair-list mat-vec gathers from it)
    int place_tried = 0; float place_ms_first = 0, place_ms_kept = 0;   // place_grids: pairs tried, the probe's time on the first and on the kept one
    bool own_z = false;                                      // z transforms by k_zfft_rows (Nz = 256, 512): rocFFT is then off the path
    double2 *twiddle_z = nullptr, *twiddle_z_owned = nullptr;   // [Nz] exp(-2 pi i m / Nz)
    double2 *twiddle_y = nullptr;                            // [Ny] exp(-2 pi i m / Ny) (== twiddle when Ny == Nx)
    double2 *twiddle_y_owned = nullptr;
    int grid_slabs = 1;   // slabs the far-field grid is cut into: n_slabs, or 1 when every rank keeps the whole grid
    double2 *twiddle = nullptr;                              // [Nx] exp(-2 pi i m / Nx)
    void *fft_work = nullptr;
    size_t fft_work_bytes = 0;
    // Lanczos
    double4 *V = nullptr;        // [M_MAX + 1][n_max]
    double *scal = nullptr, *partials = nullptr;
    double *sc_host = nullptr;   // pinned host copy of scal, mapped: the Lanczos kernels write alpha, beta, the norm there as well
    double *sc_host_dev = nullptr;   // its device address
    int *bounds_host_dev = nullptr;  // device address of bounds_host (mapped pinned): k_pick writes the row boundaries straight to the host
    int npart_cap = 0;
    // owned-particle rank (pse_params.local_rows; pse_local.h): geometry, capacities and the buffers of the step's first exchange
    struct LocalPlan {
        bool on = false;
        LocalGeom g{};
        int rows_cap = 0;                    // c_own + 2 c_g
        double *send[2] = {nullptr, nullptr}, *recv[2] = {nullptr, nullptr};   // [left, right] messages: LOCAL_HDR + LOCAL_REC * c_x doubles
        size_t msg = 0;                      // doubles per message
        double4 *stage_w1 = nullptr, *stage_w2 = nullptr;   // the last layers of w1, w2 on their way to the right neighbour (c_g rows each)
        double4 *porig_s = nullptr; double *mass_s = nullptr; int3 *image_s = nullptr;
        LocalRows *rows = nullptr;           // device
        int *counters = nullptr;             // (inside cnt_block: zeroed with the cell counts)
        int *layer_cnt = nullptr;            // (likewise) kept particles per x layer of the cell grid: LOCAL_MAX_LAYERS ints
        size_t zero_ints = 0;                // ints of cnt_block a step starts from zero
        bool zeroed = false;                 // the last step's k_local_finish has cleared them (else: a memset in front of the step)
        int *err = nullptr, *err_host = nullptr;   // error word: device, and a pinned word the status call copies it to
        // pse_team_redistribute_local (allocated at its first call): records out / in (c_own each), the destination of every particle,
        // the ranks' count rows [G][row] (row = G counts, the rank's capacity, its particle count; an even number of ints), send offsets + fill counters
        double *rd_send = nullptr, *rd_recv = nullptr;
        int *rd_rows = nullptr, *rd_off = nullptr, *rd_host = nullptr;
    } loc;
    LzState *lz_state = nullptr;     // the device-side Lanczos decision of queue-only Brownian calls (pse_set_async)
    unsigned long long lz_seq = 0;   // calls that queued one (the decision kernel stamps the host mirror with it)
    const uint32_t *ts_off = nullptr;   // pse_set_timestep_offset: the noise of a Brownian call is drawn at timestep + *ts_off
    // bookkeeping
    pse_info info;
    bool timing = false;
    Phase ph[PH_COUNT_];
    unsigned long long bytes = 0;
    int sorted_N = 0;
    bool matvec_timed = false;
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
// The passes that READ the spectra (inverse z, both y passes, the x pass) run 5 - 20 % faster or slower depending on where the driver
// placed the two grids -- a property of the allocation that holds for the life of the buffers (round 5: the two speeds of the 512^3 x
// pass; round 6: at 256^3 the inverse y + z passes take 0.293 - 0.329 ms on six pairs allocated one after another in one process, the
// step 3.15 - 3.26 ms from process to process; at 512^3 the FIRST pair of a process is the slow one, 4.60 ms for the three passes
// against 3.98 - 4.05 on the next five: docs/HISTORY.md).  A planner's answer: with the first pair in place, allocate more
// pairs (all alive at once, so that they ARE elsewhere), time the x pass + the inverse y + z passes on each, keep the fastest and free the rest.
// Only for grids large enough to matter, only while the device has room for all the candidates; results do not depend on the choice.
static ScaleArgs scale_args(pse_handle *h, bool noise, double kT, double dt, unsigned timestep);
static int place_grids(pse_handle *h, size_t nr, size_t ncx) {
    const DGrid &G = h->G;
    const size_t bytes_r = 3 * nr * sizeof(double), bytes_c = 3 * ncx * sizeof(double2);
    int K = h->tun.place_trials;
    if (K < 2 || bytes_c < ((size_t)96 << 20)) return 0;
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    K = (int)std::min<size_t>((size_t)K, 1 + free_b / 4 / (bytes_r + bytes_c));   // the candidates may take a quarter of what is free
    if (K < 2) return 0;
    struct Cand { double *r = nullptr; double2 *c = nullptr; float t = 1e30f; } cand[12];
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    int best = 0, got = 0, rc = 0;
    for (int k = 0; k < K && rc == 0; ++k) {
        Cand &c = cand[k];
        if (k == 0) { c.r = h->rgrid; c.c = h->cgrid; }           // the pair the engine already holds
        else {
            if (hipMalloc((void **)&c.r, bytes_r) != hipSuccess) { (void)hipGetLastError(); c.r = nullptr; break; }
            if (hipMalloc((void **)&c.c, bytes_c) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(c.r); c.r = nullptr; break; }
        }
        ++got;
        auto probe = [&]() -> int {
            HIPCHK(hipMemsetAsync(c.r, 0, bytes_r, h->stream));
            HIPCHK(hipMemsetAsync(c.c, 0, bytes_c, h->stream));
            double *zr[3]; double2 *zs[3];
            for (int q = 0; q < 3; ++q) { zr[q] = c.r + q * nr + (size_t)G.hl * G.Ny * G.Nz; zs[q] = c.c + q * ncx; }
            const ScaleArgs sa = scale_args(h, false, 0.0, 1.0, 0);
            for (int it = 0; it < 4; ++it) {                      // (the first one untimed)
                HIPCHK(hipEventRecord(e0, h->stream));
                if (h->xfuse) launch_xfft_scale(c.c, c.c + ncx, c.c + 2 * ncx, G, h->dbox, sa, h->twiddle, h->stream);   // (zeros in, zeros out)
                launch_yfft(c.c, G, true, h->twiddle_y, h->stream, h->tun.yfft_kb);
                launch_zfft(zr, zs, G.Nx * G.Ny, G.Nz, G.Nzp, true, h->twiddle_z, h->stream);
                HIPCHK(hipEventRecord(e1, h->stream));
                HIPCHK(hipEventSynchronize(e1));
                float ms = 0;
                HIPCHK(hipEventElapsedTime(&ms, e0, e1));
                if (it && ms < c.t) c.t = ms;
            }
            return 0;
        };
        rc = probe();
        if (h->tun.verbose) fprintf(stderr, "grid placement %d: real %p spectra %p: x + inverse y + z passes %.4f ms\n", k, (void *)c.r, (void *)c.c, c.t);
        if (rc == 0 && c.t < cand[best].t) best = k;
        // about one pair in six is of the fast kind (5 - 8 % below the rest, which lie within 2 - 3 % of one another): once one has
        // shown up there is nothing more to find
        if (rc == 0 && got >= 3) {
            float worst = 0.0f;
            for (int q = 0; q < got; ++q) worst = std::max(worst, cand[q].t);
            if (cand[best].t < 0.95f * worst) break;
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc) best = 0;
    for (int k = 1; k < got; ++k) if (k != best) { (void)hipFree(cand[k].r); (void)hipFree(cand[k].c); }
    if (best != 0) { (void)hipFree(cand[0].r); (void)hipFree(cand[0].c); h->rgrid = cand[best].r; h->cgrid = cand[best].c; }
    h->place_tried = got; h->place_ms_first = cand[0].t; h->place_ms_kept = cand[best].t;
    if (h->tun.verbose) fprintf(stderr, "grid placement: kept %d of %d (%.4f ms; the first one %.4f)\n", best, got, cand[best].t, cand[0].t);
    return rc;
}
static void set_dbox(pse_handle *h) {
    h->dbox = DBox{h->box.Lx, h->box.Ly, h->box.Lz, h->box.xy, 1.0 / h->box.Lx, 1.0 / h->box.Ly, 1.0 / h->box.Lz};
}

// cells of at least rcut perpendicular width for tilts up to gamma; a dimension with fewer than 3 cells uses 1
static int cells_for(const Box &box, double rc, double gamma, int n_slabs, int bz_want, DCells &out, int xpad = 0) {
    const double wx = box.Lx / std::sqrt(1.0 + gamma * gamma), wy = box.Ly, wz = box.Lz;
    if (rc > 0.5 * wx * (1 + 1e-12) || rc > 0.5 * wy * (1 + 1e-12) || rc > 0.5 * wz * (1 + 1e-12))
        return fail(PSE_ERR_INVALID, "real-space cutoff %.4f exceeds half the box width (%.4f, %.4f, %.4f at tilt %.3f): "
                    "the minimum-image near field needs rcut <= L/2; increase xi", rc, wx, wy, wz, gamma);
    auto n = [&](double w) { int c = (int)std::floor(w / rc); if (c < 3) c = 1; if (c > 1024) c = 1024; return c; };
    out = DCells{n(wx), n(wy), n(wz), 0, 1, xpad};
    // Blocks of six cells along z in the storage order (pse_device.h; PSE_CELL_BZ=b at pse_create overrides, 0 = plain (x, y, z) order).
    // Measured at the metric point: the pair-list mat-vec 0.163 -> 0.153 ms (a wave's gathers come from ~95 cells instead of
    // ~130), the cell pass 0.60 -> 0.62 ms (lanes of a wave no longer walk the same z lines): about 1 % per step in four of four
    // paired bench runs.
    out.bz = (bz_want > 0 && out.nz >= 2 * bz_want) ? bz_want : out.nz;
    out.nzb = (out.nz + out.bz - 1) / out.bz;
    if (n_slabs > 1) {
        // cell slabs coincide with grid slabs: every rank owns ncx/G whole cell layers = one contiguous row range
        const int c = (int)std::floor(wx / rc) / n_slabs * n_slabs;
        if (c < 3 || c < n_slabs)
            return fail(PSE_ERR_INVALID, "box too small to split the near field over %d ranks: only %d cells of width >= rcut "
                        "fit along x", n_slabs, (int)std::floor(wx / rc));
        out.nx = std::min(c, 1024 / n_slabs * n_slabs);
    }
    return 0;
}
static double near_radius(const pse_handle *h) { return h->d.rcut + h->skin_max; }   // cells are as wide as the kept list reaches
static int set_cells(pse_handle *h, double gamma) {
    DCells nc;
    TRY(cells_for(h->box, near_radius(h), gamma, h->n_slabs, h->tun.cell_bz, nc, h->loc.on ? 1 : 0));
    h->nc = nc;
    h->cell_gamma = gamma;
    h->nc_wide = h->skin_max > 0.0;
    return 0;
}

enum { PH_SORT, PH_SPREAD, PH_FFTF, PH_SCALE, PH_FFTI, PH_GATHER, PH_REAL, PH_LANCZOS, PH_INTEG, PH_COMM, PH_TOTAL, PH_MATVEC, PH_RECORDS, PH_COUNT };
// Phase ranges for rocprofv3 --marker-trace (PSE_ROCTX=1): host-side roctx ranges around the launches of each phase (the
// reference has one HOOMD Profiler push/pop around the whole step, PSEv1/Stokes.cc:450-451,519-521).  libroctx64 is
// looked up at run time: no link dependency, nothing happens unless the switch is set.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char *e = getenv("PSE_ROCTX");
        if (!e || !atoi(e)) return;
        void *lib = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
        push = (int (*)(const char *))dlsym(lib, "roctxRangePushA");
        pop = (int (*)())dlsym(lib, "roctxRangePop");
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
static const Roctx &roctx() { static Roctx r; return r; }
static_assert(PH_COUNT == PH_COUNT_, "phase table");
static const char *const PH_NAME[PH_COUNT] = {"pse:sort", "pse:spread", "pse:fft_forward", "pse:kspace_scale", "pse:fft_inverse", "pse:gather",
                                              "pse:near_field", "pse:lanczos", "pse:integrate", "pse:exchange", "pse:total", "pse:matvec",
                                              "pse:far_records"};
static void range_push(int p) { if (roctx().push) roctx().push(PH_NAME[p]); }
static void range_pop() { if (roctx().pop) roctx().pop(); }
static int ts(pse_handle *h, int p) { range_push(p); if (h->timing) HIPCHK(hipEventRecord(h->ph[p].a, h->stream)); return 0; }
static int te(pse_handle *h, int p) { if (h->timing) HIPCHK(hipEventRecord(h->ph[p].b, h->stream)); range_pop(); return 0; }
static int tsw(pse_handle *h, int p) { range_push(p); if (h->timing) HIPCHK(hipEventRecord(h->ph[p].a, h->wstream)); return 0; }
static int tew(pse_handle *h, int p) { if (h->timing) HIPCHK(hipEventRecord(h->ph[p].b, h->wstream)); range_pop(); return 0; }

static int collect_times(pse_handle *h, unsigned mask) {
    if (!h->timing) return 0;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->side) HIPCHK(hipStreamSynchronize(h->side));
    double *dst[PH_COUNT] = {&h->info.t_sort, &h->info.t_spread, &h->info.t_fft_fwd, &h->info.t_scale, &h->info.t_fft_inv,
                             &h->info.t_gather, &h->info.t_real, &h->info.t_lanczos, &h->info.t_integrate, &h->info.t_comm,
                             &h->info.t_total, &h->info.t_matvec, &h->info.t_records};
    for (int p = 0; p < PH_COUNT; ++p) {
        *dst[p] = 0.0;
        if (mask & (1u << p)) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, h->ph[p].a, h->ph[p].b) == hipSuccess) *dst[p] = ms;
        }
    }
    return 0;
}

extern "C" int pse_destroy(pse_handle *h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (h->plan_fwd) rocfft_plan_destroy(h->plan_fwd);
    if (h->plan_inv) rocfft_plan_destroy(h->plan_inv);
    if (h->plan_x_fwd) rocfft_plan_destroy(h->plan_x_fwd);
    if (h->plan_x_inv) rocfft_plan_destroy(h->plan_x_inv);
    if (h->info_fwd) rocfft_execution_info_destroy(h->info_fwd);
    if (h->info_inv) rocfft_execution_info_destroy(h->info_inv);
    void *ptrs[] = {h->keys, h->keys_s, h->vals, h->perm, h->tag_s, h->sort_tmp, h->cell_off, h->cnt_block, h->sw.rec_t, h->sw.fb.off, h->sw.fb.rank_s, h->sw.fb.tmp, h->nb.data, h->nb.cnt, h->vl.idx, h->vl.cnt, h->pos_build, h->pos_s, h->posf_s, h->pv,
                    h->f_s, h->uw_s, h->ur_s, h->ub_s, h->psi_s, h->w_s, h->coef, h->rgrid, h->cgrid, h->sendbuf, h->recvbuf, h->d_bidx, h->d_bounds, h->utot_s, h->w2_s, h->u_s, h->sums_all, h->twiddle, h->twiddle_y_owned, h->twiddle_z_owned, h->fft_work, h->V,
                    h->scal, h->partials, h->lz_state, h->vq};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &p : h->ph) { if (p.a) (void)hipEventDestroy(p.a); if (p.b) (void)hipEventDestroy(p.b); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_scal) (void)hipEventDestroy(h->ev_scal);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->side_owned) (void)hipStreamDestroy(h->side_owned);
    if (h->bounds_host) (void)hipHostFree(h->bounds_host);
    if (h->ev_bounds) (void)hipEventDestroy(h->ev_bounds);
    if (h->sc_host) (void)hipHostFree(h->sc_host);
    if (h->flags_host) (void)hipHostFree(h->flags_host);
    {
        void *lp[] = {h->loc.send[0], h->loc.send[1], h->loc.recv[0], h->loc.recv[1], h->loc.stage_w1, h->loc.stage_w2, h->loc.porig_s, h->loc.mass_s,
                      h->loc.image_s, h->loc.rows, h->loc.err, h->loc.rd_send, h->loc.rd_recv, h->loc.rd_rows, h->loc.rd_off};
        for (void *q : lp) if (q) (void)hipFree(q);
        if (h->loc.err_host) (void)hipHostFree(h->loc.err_host);
        if (h->loc.rd_host) (void)hipHostFree(h->loc.rd_host);
    }
    delete h;
    return 0;
}

static int make_plans(pse_handle *h) {
    std::call_once(g_fft_once, [] { rocfft_setup(); });
    const DGrid &G = h->G;
    size_t work = 0, w = 0;
    h->xfuse = xfuse_supported(G.Nx) && !h->tun.no_xfuse;   // x axis by k_xfft_scale (also after the slab transpose)
    if (h->xfuse) {
        std::vector<double2> tw(G.Nx);
        for (int m = 0; m < G.Nx; ++m) {
            const long double ang = -2.0L * 3.14159265358979323846264338327950288L * m / G.Nx;
            tw[m] = make_double2((double)cosl(ang), (double)sinl(ang));
        }
        TRY(dmalloc(h, &h->twiddle, (size_t)G.Nx));
        HIPCHK(hipMemcpy(h->twiddle, tw.data(), G.Nx * sizeof(double2), hipMemcpyHostToDevice));
    }
    // the y transforms of a single GPU's grid by the own in-place pass where rocFFT's strided pass is slow (not a power of two)
    const bool z_ok = h->tun.own_z > 0 && zfft_supported(G.Nz);
    h->own_y = h->xfuse && h->grid_slabs == 1 && h->tun.own_y > 0 &&
               (yfft_supported(G.Ny) || (h->tun.own_y_pow2 && yfft_regs_supported(G.Ny, G.Nz, z_ok)));
    // a slab rank: the own y pass for ANY smooth Ny -- it writes the all-to-all blocks directly (no pack / unpack pass)
    h->own_y_slab = h->xfuse && h->grid_slabs > 1 && h->tun.own_y > 0 && yfft_possible(G.Ny);
    h->own_z = z_ok && (h->own_y || h->own_y_slab);   // (where rocFFT's plan is its 1-D z plan)
    if (h->own_z) {
        if (G.Nz == G.Nx) h->twiddle_z = h->twiddle;
        else {
            std::vector<double2> tw(G.Nz);
            for (int m = 0; m < G.Nz; ++m) {
                const long double ang = -2.0L * 3.14159265358979323846264338327950288L * m / G.Nz;
                tw[m] = make_double2((double)cosl(ang), (double)sinl(ang));
            }
            TRY(dmalloc(h, &h->twiddle_z_owned, (size_t)G.Nz));
            HIPCHK(hipMemcpy(h->twiddle_z_owned, tw.data(), G.Nz * sizeof(double2), hipMemcpyHostToDevice));
            h->twiddle_z = h->twiddle_z_owned;
        }
    }
    if (h->own_y || h->own_y_slab) {
        if (G.Ny == G.Nx) h->twiddle_y = h->twiddle;
        else {
            std::vector<double2> tw(G.Ny);
            for (int m = 0; m < G.Ny; ++m) {
                const long double ang = -2.0L * 3.14159265358979323846264338327950288L * m / G.Ny;
                tw[m] = make_double2((double)cosl(ang), (double)sinl(ang));
            }
            TRY(dmalloc(h, &h->twiddle_y_owned, (size_t)G.Ny));
            HIPCHK(hipMemcpy(h->twiddle_y_owned, tw.data(), G.Ny * sizeof(double2), hipMemcpyHostToDevice));
            h->twiddle_y = h->twiddle_y_owned;
        }
    }
    // real grid rows hold Nz doubles, spectrum rows Nzp >= Nz/2 + 1 complex numbers (padded to 128 bytes)
    auto real_plans = [&](size_t dims, const size_t *len, size_t batch) -> int {
        size_t rs[3] = {1, (size_t)G.Nz, (size_t)G.Ny * G.Nz}, cs[3] = {1, (size_t)G.Nzp, (size_t)G.Ny * G.Nzp};
        const size_t rdist = dims == 1 ? (size_t)G.Nz : (dims == 2 ? (size_t)G.Ny * G.Nz : (size_t)G.Nx * G.Ny * G.Nz);
        const size_t cdist = dims == 1 ? (size_t)G.Nzp : (dims == 2 ? (size_t)G.Ny * G.Nzp : (size_t)G.Nx * G.Ny * G.Nzp);
        rocfft_plan_description df = nullptr, di = nullptr;
        FFTCHK(rocfft_plan_description_create(&df));
        FFTCHK(rocfft_plan_description_create(&di));
        FFTCHK(rocfft_plan_description_set_data_layout(df, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr, nullptr,
                                                       dims, rs, rdist, dims, cs, cdist));
        FFTCHK(rocfft_plan_description_set_data_layout(di, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, nullptr, nullptr,
                                                       dims, cs, cdist, dims, rs, rdist));
        FFTCHK(rocfft_plan_create(&h->plan_fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                                  rocfft_precision_double, dims, len, batch, df));
        FFTCHK(rocfft_plan_create(&h->plan_inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                                  rocfft_precision_double, dims, len, batch, di));
        rocfft_plan_description_destroy(df);
        rocfft_plan_description_destroy(di);
        return 0;
    };
    if (h->own_y) {
        // 1-D real transforms along z of every row; the y transforms are k_fft_cols', the x transforms k_xfft_scale's
        const size_t len1[1] = {(size_t)G.Nz};
        TRY(real_plans(1, len1, (size_t)3 * G.Nx * G.Ny));
    } else if (h->xfuse && h->grid_slabs == 1) {
        // 2-D (y,z) real transforms of all 3 Nx planes in one batch
        const size_t len2[2] = {(size_t)G.Nz, (size_t)G.Ny};
        TRY(real_plans(2, len2, (size_t)3 * G.Nx));
    } else if (h->grid_slabs == 1) {
        // rocFFT lengths are fastest-first: z, y, x.  Real grids [3][Nx][Ny][Nz] -> half spectra [3][Nx][Ny][Nzp].
        const size_t len[3] = {(size_t)G.Nz, (size_t)G.Ny, (size_t)G.Nx};
        TRY(real_plans(3, len, 3));
    } else {
        // slab mode: 2-D (y,z) real transforms of the nxl local planes of one component, then (after the transpose)
        // 1-D complex transforms along x on [Nx][nyl][Nzp]: stride nyl*Nzp, one transform per (y,kz) column
        const size_t len2[2] = {(size_t)G.Nz, (size_t)G.Ny};
        const size_t len1[1] = {(size_t)G.Nz};
        if (h->own_y_slab) TRY(real_plans(1, len1, (size_t)G.nxl * G.Ny));   // z transforms only; the y transforms are k_fft_cols'
        else TRY(real_plans(2, len2, (size_t)G.nxl));
        const size_t lenx[1] = {(size_t)G.Nx};
        size_t stride[1] = {(size_t)h->nyl * G.Nzp};
        rocfft_plan_description desc = nullptr;
        FFTCHK(rocfft_plan_description_create(&desc));
        FFTCHK(rocfft_plan_description_set_data_layout(desc, rocfft_array_type_complex_interleaved,
                                                       rocfft_array_type_complex_interleaved, nullptr, nullptr, 1, stride, 1,
                                                       1, stride, 1));
        FFTCHK(rocfft_plan_create(&h->plan_x_fwd, rocfft_placement_inplace, rocfft_transform_type_complex_forward,
                                  rocfft_precision_double, 1, lenx, stride[0], desc));
        FFTCHK(rocfft_plan_create(&h->plan_x_inv, rocfft_placement_inplace, rocfft_transform_type_complex_inverse,
                                  rocfft_precision_double, 1, lenx, stride[0], desc));
        rocfft_plan_description_destroy(desc);
        FFTCHK(rocfft_plan_get_work_buffer_size(h->plan_x_fwd, &w)); work = std::max(work, w);
        FFTCHK(rocfft_plan_get_work_buffer_size(h->plan_x_inv, &w)); work = std::max(work, w);
    }
    FFTCHK(rocfft_plan_get_work_buffer_size(h->plan_fwd, &w)); work = std::max(work, w);
    FFTCHK(rocfft_plan_get_work_buffer_size(h->plan_inv, &w)); work = std::max(work, w);
    h->fft_work_bytes = work;
    if (work) TRY(dmalloc(h, (char **)&h->fft_work, work));
    FFTCHK(rocfft_execution_info_create(&h->info_fwd));
    FFTCHK(rocfft_execution_info_create(&h->info_inv));
    if (work) {
        FFTCHK(rocfft_execution_info_set_work_buffer(h->info_fwd, h->fft_work, work));
        FFTCHK(rocfft_execution_info_set_work_buffer(h->info_inv, h->fft_work, work));
    }
    FFTCHK(rocfft_execution_info_set_stream(h->info_fwd, h->wstream));
    FFTCHK(rocfft_execution_info_set_stream(h->info_inv, h->wstream));
    return 0;
}

static bool team_sstep(const pse_handle *h);
static int create_impl(const pse_params *p, pse_handle *h) {
    h->par = *p;
    h->box = Box{p->Lx, p->Ly, p->Lz, p->xy};
    std::string e = select_params(h->box, p->xi, p->error, p->max_strain, p->Nx, p->Ny, p->Nz, p->P, p->rcut, h->d);
    if (!e.empty()) return fail(PSE_ERR_INVALID, "%s", e.c_str());
    if (p->n_max == 0) return fail(PSE_ERR_INVALID, "n_max must be positive");
    TRY(gaussian_fits(h->d, h->d.hx, h->d.hy, h->d.hz));
    if (!(std::fabs(p->xy) <= 0.5 * (1.0 + 1e-9)))
        return fail(PSE_ERR_INVALID, "tilt xy = %g outside [-0.5, 0.5]: HOOMD flips the box there, and the sequential minimum image "
                    "(PSEv1/Mobility.cu:648) is exact only up to that tilt", p->xy);
    const Derived &d = h->d;
    if (p->device >= 0) { HIPCHK(hipSetDevice(p->device)); h->device = p->device; }
    else HIPCHK(hipGetDevice(&h->device));
    h->n_max = (int)p->n_max;
    set_dbox(h);
    h->n_slabs = std::max(1, p->n_slabs);
    h->loc.on = p->local_rows != 0;
    if (h->loc.on && h->n_slabs < 2) return fail(PSE_ERR_INVALID, "local_rows needs n_slabs >= 2 (an owned-particle rank is a member of a team)");
    (void)roctx();   // PSE_ROCTX: looked up here, once per process
    {   // the developer switches: the environment is read here and nowhere else
        auto ienv = [](const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; };
        auto &t = h->tun;
        t.cell_bz = std::min(16, std::max(0, ienv("PSE_CELL_BZ", 6)));   // n_cells_alloc pads every z line by up to 15 cells
        if (const char *v = getenv("PSE_SKIN")) t.skin = atof(v);
        t.overlap = ienv("PSE_OVERLAP", 1);   // (Brownian steps fork too: with the lighter pair-list mat-vec the far-field chain fits beside the Lanczos chain, 3.23 -> 3.11 ms; rounds 4-5: a loss)
        t.no_xfuse = getenv("PSE_NO_XFUSE") != nullptr;
        t.own_y = ienv("PSE_OWN_Y", 1); t.own_y_pow2 = ienv("PSE_OWN_Y_POW2", 1); t.yfft_kb = ienv("PSE_YFFT_KB", 4); t.own_z = ienv("PSE_OWN_Z", 1); t.yslab_regs = ienv("PSE_YSLAB_REGS", 1);
        if (const char *v = getenv("PSE_WAVE_MODE")) t.wave_mode = !strcmp(v, "slab") ? 1 : (!strcmp(v, "replicated") ? 2 : 0);
        t.spread_tz = ienv("PSE_SPREAD_TZ", 0); t.spread_nw = ienv("PSE_SPREAD_NW", 0);
        t.gather_bz = ienv("PSE_GATHER_BZ", 0); t.xmix_runtime = ienv("PSE_XMIX", 0) == 1; t.xfft_small_wide = ienv("PSE_XFFT_SMALL_KB", 2) == 8;
        t.verbose = ienv("PSE_VERBOSE", 0) > 0;
        t.team_sstep = ienv("PSE_TEAM_SSTEP", 1) > 0;
        t.lz_extra = std::max(0, std::min(32, ienv("PSE_LANCZOS_EXTRA", 2)));
        t.place_trials = std::max(0, std::min(12, ienv("PSE_PLACE_TRIALS", 10)));
        t.vq = ienv("PSE_VQ", 1);
        if (const char *v = getenv("PSE_TEAM_SCHED")) {
            int a = 1, b = 2, c = 3;
            if (sscanf(v, "%d,%d,%d", &a, &b, &c) == 3) { t.team_sched[0] = a; t.team_sched[1] = b; t.team_sched[2] = c; }
        }
        h->sw.force_tz = t.spread_tz; h->sw.force_nw = t.spread_nw; h->sw.force_bz = t.gather_bz;
    }
    {
        // Neighbour list across steps: on by default with the reference's r_buff = 0.4 (PSEv1/integrate.py:60), single GPU, table
        // in LDS, box wide enough for rcut + skin; PSE_SKIN overrides (0: off), pse_set_neighbor_skin may lower it later.
        double skin = h->tun.skin;
        const int nint = (int)std::ceil(d.rcut * RS_PER_UNIT) + 1;
        const double gam = std::max(std::fabs(p->xy), p->max_strain);
        const double wmin = std::min(std::min(h->box.Lx / std::sqrt(1.0 + gam * gam), h->box.Ly), h->box.Lz);
        if (!(skin > 0.0) || h->n_slabs > 1 || !mreal_table_in_lds(nint * 2 * RS_NCOEF) || d.rcut + skin > 0.5 * wmin)
            skin = 0.0;
        h->skin = h->skin_max = skin;
    }
    TRY(set_cells(h, std::max(std::fabs(p->xy), p->max_strain)));
    fill_info(d, &h->info);
    h->info.ncell_x = h->nc.nx; h->info.ncell_y = h->nc.ny; h->info.ncell_z = h->nc.nz;

    DGrid &G = h->G;
    G.Nx = d.Nx; G.Ny = d.Ny; G.Nz = d.Nz; G.Nzh = d.Nz / 2 + 1; G.P = d.P;
    G.Nzp = (G.Nzh + 7) & ~7;   // spectrum rows padded to 128 bytes: the x pass reads and writes whole aligned lines
    h->n_slabs = std::max(1, p->n_slabs);
    h->slab_rank = h->n_slabs > 1 ? p->slab_rank : 0;
    if (h->slab_rank < 0 || h->slab_rank >= h->n_slabs) return fail(PSE_ERR_INVALID, "slab_rank outside [0, n_slabs)");
    // Far field of a team: slab-decomposed (2 all-to-alls per evaluation), or kept whole on every rank with only the near
    // field and Lanczos sharded.  With two ranks the all-to-all is one xGMI link each way (101 MB per direction at 256^3:
    // ~2 ms, longer than the far field itself), so two ranks replicate the far field; PSE_WAVE_MODE=slab|replicated overrides.
    h->grid_slabs = h->n_slabs;
    if (h->n_slabs > 1) {
        const bool replicate = h->tun.wave_mode ? h->tun.wave_mode == 2 : h->n_slabs == 2;
        if (replicate && !h->loc.on) h->grid_slabs = 1;   // (an owned-particle rank has no particles to spread a whole grid with)
    }
    if (h->grid_slabs > 1) {
        if (d.Nx % h->grid_slabs || d.Ny % h->grid_slabs)
            return fail(PSE_ERR_INVALID, "slab decomposition needs Nx and Ny divisible by the number of ranks (%d x %d over %d)",
                        d.Nx, d.Ny, h->grid_slabs);
        if (d.Nx / h->grid_slabs < d.P)
            return fail(PSE_ERR_INVALID, "slabs of %d planes are thinner than the support P = %d", d.Nx / h->grid_slabs, d.P);
    }
    const int grid_rank = h->grid_slabs > 1 ? h->slab_rank : 0;
    G.nxl = d.Nx / h->grid_slabs; G.x0 = grid_rank * G.nxl;
    // the gather of a particle is done by the rank whose slab holds the particle; its support reaches (P-1)/2 planes
    // below and (P+1)/2 planes above that slab: copies of the neighbours' planes are stored around the own ones
    G.hl = h->grid_slabs > 1 ? (d.P - 1) / 2 : 0;
    G.nhalo = h->grid_slabs > 1 ? (d.P + 1) / 2 : 0;
    h->nyl = d.Ny / h->grid_slabs; h->y0 = grid_rank * h->nyl;
    G.hx = d.hx; G.hy = d.hy; G.hz = d.hz;
    const double c = 2.0 * d.xi * d.xi / d.eta;
    G.expfac = c;                                      // PSEv1/Brownian.cu:829
    G.prefac = (c / M_PI) * std::sqrt(c / M_PI);       // PSEv1/Brownian.cu:828

    // particle arrays are padded so that equal row chunks of every rank fit (all-gather inside Lanczos)
    const size_t n = ((size_t)h->n_max + h->n_slabs - 1) / h->n_slabs * h->n_slabs;
    h->n_pad = (int)n;
    TRY(dmalloc(h, &h->keys, n)); TRY(dmalloc(h, &h->keys_s, n)); TRY(dmalloc(h, &h->vals, n));
    TRY(dmalloc(h, &h->perm, n)); TRY(dmalloc(h, &h->tag_s, n));

    // size the cell arrays for zero tilt (most cells)
    {
        const double rc = d.rcut;
        auto cnt = [&](double w) { int c2 = (int)std::floor(w / rc); if (c2 < 3) c2 = 1; if (c2 > 1024) c2 = 1024; return (size_t)c2; };
        h->n_cells_alloc = cnt(h->box.Lx) * cnt(h->box.Ly) * (cnt(h->box.Lz) + 16) + cnt(h->box.Lx);   // + the padding of the last z block (bz <= 16), + one empty cell per x layer (xpad)
    }
    TRY(dmalloc(h, &h->cell_off, h->n_cells_alloc + 1));
    h->sort_tmp_bytes = cell_sort_temp_bytes(h->n_cells_alloc);
    TRY(dmalloc(h, (char **)&h->sort_tmp, h->sort_tmp_bytes));
    // the counters every call starts from zero, side by side: [bin counts | flags[0], flags[1] | cell counts] -- one memset
    // covers what a call needs (a call that checks the kept list stops after flags[0]: flags[1] is the mark of its build)
    const size_t nbins = (size_t)((d.Nx + 7) / 8) * ((d.Ny + 7) / 8) * ((d.Nz + 7) / 8);
    const bool fast_far = d.P >= 4 && d.P <= FAR_PMAX;
    // layout in ints: [0, B - 1) bin counts (incl. sentinel, padded) | B - 1: flags[0] | B: flags[1] | B + 4 ...: cell counts.  B and every
    // memset size are multiples of four ints: a memset that is not a multiple of 16 bytes takes two fill kernels.
    h->cnt_bins = ((fast_far ? nbins + 1 : 0) + 1 + 3) & ~(size_t)3;   // B
    constexpr size_t LAYER_INTS = (size_t)LOCAL_MAX_LAYERS * LOCAL_LAYER_SLOTS;   // (the per-layer counts of an owned-particle rank's sort)
    TRY(dmalloc(h, &h->cnt_block, h->cnt_bins + 4 + h->n_cells_alloc + 1 + 8 + LOCAL_NCOUNTER + 2 + LAYER_INTS + 4));
    h->loc.counters = h->cnt_block + ((h->cnt_bins + 4 + h->n_cells_alloc + 1 + 8 + 1) & ~(size_t)1);   // 8-byte aligned
    h->loc.layer_cnt = h->loc.counters + LOCAL_NCOUNTER;
    h->loc.zero_ints = ((size_t)(h->loc.layer_cnt - h->cnt_block) + LAYER_INTS + 3) & ~(size_t)3;
    if (h->nc.nx > LOCAL_MAX_LAYERS && p->local_rows) return fail(PSE_ERR_INVALID, "more than %d cell layers along x", LOCAL_MAX_LAYERS);
    h->vl.flags = h->cnt_block + h->cnt_bins - 1;
    h->gate_word = h->cnt_block + h->cnt_bins + 1;   // (B + 1: between flags[1] and the cell counts; no memset covers it alone)
    h->cell_cnt = h->cnt_block + h->cnt_bins + 4;
    if (fast_far) {   // fast far-field path: bin-ordered 64-byte records
        h->sw.fb.cnt = h->cnt_block;
        TRY(dmalloc(h, (char **)&h->sw.rec_t, (n + 64) * 64));   // 64-byte records (idle lanes read past the last one)
        TRY(dmalloc(h, &h->sw.fb.rank_s, n));
        TRY(dmalloc(h, &h->sw.fb.off, nbins + 1));
        h->sw.fb.tmp_bytes = bin_scan_temp_bytes(nbins);
        TRY(dmalloc(h, (char **)&h->sw.fb.tmp, h->sw.fb.tmp_bytes));
    }
    std::vector<double> coef;
    build_realspace_table(d.xi, d.rcut, coef, h->n_intervals);
    TRY(dmalloc(h, &h->coef, coef.size()));
    HIPCHK(hipMemcpy(h->coef, coef.data(), coef.size() * sizeof(double), hipMemcpyHostToDevice));
    TRY(dmalloc(h, &h->nb.cnt, n));
    {   // per-step pair list: capacity from the mean neighbour count
        const double vol = h->box.Lx * h->box.Ly * h->box.Lz;
        // particles of the whole box behind the capacity (an owned-particle rank holds its slab + four ghost layers of it, with slack)
        const double n_box = h->loc.on ? (double)n * h->n_slabs * (h->nc.nx / h->n_slabs) / (h->nc.nx / h->n_slabs + 4.0) : (double)n;
        const double nbar = n_box / vol * 4.18879020478639 * d.rcut * d.rcut * d.rcut;
        int cap = (int)std::ceil(1.5 * nbar + 16.0);
        cap = (std::max(16, std::min(cap, 256)) + 3) & ~3;   // whole groups of four slots
        const double bytes = (double)cap * (double)n * 20.0;
        if (bytes > 32e9 || n >= ((size_t)1 << 27)) cap = 0;   // too large: mat-vecs always walk the cells
        h->nb.cap = cap;
        if (cap > 0) {
            TRY(dmalloc(h, &h->nb.data, nb_list_bytes(n + 1024, cap)));   // + the padding of up to three row ranges to whole workgroups (RowMap)
        }
        if (cap == 0) h->skin = h->skin_max = 0.0;
        if (h->skin_max > 0.0) {
            const double rs = d.rcut + h->skin_max, nbar_v = n_box / vol * 4.18879020478639 * rs * rs * rs;
            h->vl.cap = (std::max(16, std::min((int)std::ceil(2.0 * nbar_v + 32.0), 512)) + 3) & ~3;   // 4 bytes per slot: generous
            h->vl.rskin = rs;
            TRY(dmalloc(h, (char **)&h->vl.idx, verlet_list_bytes(n + 1024, h->vl.cap)));   // + padding rows (the lanes past the last row of the build pass write there)
            TRY(dmalloc(h, &h->vl.cnt, n));
            TRY(dmalloc(h, &h->pos_build, n));
            HIPCHK(hipHostMalloc((void **)&h->flags_host, 2 * sizeof(int)));
        }
    }
    TRY(dmalloc(h, &h->pos_s, n)); TRY(dmalloc(h, &h->posf_s, n));
    TRY(dmalloc(h, &h->pv, 3 * n));   // packed (position, force) records of the near-field passes
    TRY(dmalloc(h, &h->f_s, n)); TRY(dmalloc(h, &h->uw_s, n)); TRY(dmalloc(h, &h->ur_s, n));
    TRY(dmalloc(h, &h->ub_s, n)); TRY(dmalloc(h, &h->psi_s, n)); TRY(dmalloc(h, &h->w_s, n));

    const size_t nr = (size_t)(G.nxl + G.hl + G.nhalo) * G.Ny * G.Nz, ncx = (size_t)G.nxl * G.Ny * G.Nzp;
    TRY(dmalloc(h, &h->rgrid, 3 * nr));
    TRY(dmalloc(h, &h->cgrid, 3 * ncx));
    if (h->grid_slabs > 1) { TRY(dmalloc(h, &h->sendbuf, 3 * ncx)); TRY(dmalloc(h, &h->recvbuf, 3 * ncx)); }
    if (h->n_slabs > 1) {
        HIPCHK(hipHostMalloc((void **)&h->bounds_host, ((size_t)5 * h->n_slabs + 1) * sizeof(int), hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void **)&h->bounds_host_dev, h->bounds_host, 0));
        HIPCHK(hipEventCreateWithFlags(&h->ev_bounds, hipEventDisableTiming));
        TRY(dmalloc(h, &h->d_bidx, (size_t)5 * h->n_slabs + 1)); TRY(dmalloc(h, &h->d_bounds, (size_t)5 * h->n_slabs + 1));
        TRY(dmalloc(h, &h->utot_s, n));
        TRY(dmalloc(h, &h->sums_all, (size_t)h->n_slabs * LZ_NGRAM));
        if (h->tun.team_sstep && h->nb.cap > 0) {
            TRY(dmalloc(h, &h->w2_s, n)); TRY(dmalloc(h, &h->u_s, n));
        }
    }
    if (h->loc.on) {
        // owned-particle rank: the row space [own | left ghosts | right ghosts] and the messages of the step's first exchange
        auto &L = h->loc;
        const int G_ = h->n_slabs, per = h->nc.nx / G_, depth = 2;
        if (per < depth + 1 || (G_ - 1) * per < 2 * depth)
            return fail(PSE_ERR_INVALID, "an owned-particle rank needs >= %d cell layers of width rcut per rank along x (%d ranks: %d layers each)",
                        G_ == 2 ? 2 * depth : depth + 1, G_, per);
        if (!team_sstep(h))
            return fail(PSE_ERR_INVALID, "an owned-particle rank needs the per-step pair list and the real-space table in LDS (rcut = %.3f, n_max = %d)",
                        d.rcut, h->n_max);
        if ((d.P / 2 + 1.0) / d.Nx > (double)depth / h->nc.nx)
            return fail(PSE_ERR_INVALID, "the spreading support (P = %d on %d planes) reaches beyond the %d ghost cell layers of an owned-particle rank", d.P, d.Nx, depth);
        const int c_g = (int)((double)h->n_max * depth / (per + 2 * depth)) / 256 * 256, c_own = (h->n_max - 2 * c_g) / 256 * 256;
        if (c_g < 256 || c_own < c_g)
            return fail(PSE_ERR_INVALID, "n_max = %d is too small a row capacity for an owned-particle rank (%d layers + 2 x %d ghost layers)", h->n_max, per, depth);
        L.g = LocalGeom{h->slab_rank, G_, per, depth, h->nc.nx, c_own, c_g, c_g};
        L.rows_cap = c_own + 2 * c_g;
        L.msg = (size_t)LOCAL_HDR + (size_t)LOCAL_REC * c_g;
        for (int q = 0; q < 2; ++q) { TRY(dmalloc(h, &L.send[q], L.msg)); TRY(dmalloc(h, &L.recv[q], L.msg)); HIPCHK(hipMemset(L.recv[q], 0, L.msg * sizeof(double))); }
        TRY(dmalloc(h, &L.stage_w1, (size_t)c_g)); TRY(dmalloc(h, &L.stage_w2, (size_t)c_g));
        TRY(dmalloc(h, &L.porig_s, (size_t)c_own)); TRY(dmalloc(h, &L.mass_s, (size_t)c_own)); TRY(dmalloc(h, &L.image_s, (size_t)c_own));
        TRY(dmalloc(h, &L.rows, 1));
        TRY(dmalloc(h, &L.err, 4));
        HIPCHK(hipMemset(L.err, 0, 4 * sizeof(int)));
        HIPCHK(hipMemset(L.rows, 0, sizeof(LocalRows)));
        HIPCHK(hipHostMalloc((void **)&L.err_host, 4 * sizeof(int)));
        L.err_host[0] = 0;
    }
    // The wave-space chain (spread -> FFTs -> gather) and the real-space chain (near field + Lanczos) only meet in the
    // final sum: on a single GPU they run on two streams, so latency-bound kernels of one chain fill the gaps of the
    // other and the Lanczos host checks do not stall the far field.  (Teams keep one stream: one RCCL communicator.)
    h->wstream = h->stream;
    // Deterministic evaluations (kT = 0) fork whenever no phase timing is requested: the far-field chain runs on a side stream next
    // to the near field (+3.5 % M.F evaluations per second).  Brownian steps stay on ONE stream by default since the end of round 3:
    // with the build pass at three workgroups per CU, the gather at six and the spread at twelve, every kernel fills the chip on
    // its own and a second chain buys nothing measurable (six paired runs: forked minus one stream between -1.2 % and +1.8 %;
    // PSE_OVERLAP=1 forks them too, -1 never forks).  With pse_set_timing on, everything runs on one stream: per-kernel
    // durations -- the roofline evidence -- are then those of the kernel alone.
    if (h->tun.overlap >= 0) {
        {   // PSE_SIDE_PRIORITY=low|high|default: the far-field lane at the lowest / highest / the default stream priority.  Default: the
            // default priority -- but LOW on an owned-particle rank, whose near field + Lanczos chain is the longer lane by ~100 us at
            // eight ranks: its mat-vecs then lose less to the transforms that run next to them (solo rank, metric point 0.724 -> 0.701 ms,
            // BASELINE config 4 2.72 -> 2.69)
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo: numerically largest = lowest priority
            const char *pr = getenv("PSE_SIDE_PRIORITY");
            if (!pr || !*pr) pr = h->loc.on ? "low" : "default";
            if (!strcmp(pr, "low") || !strcmp(pr, "high"))
                HIPCHK(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, !strcmp(pr, "low") ? lo : hi));
            else
                HIPCHK(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
        }
        h->side_owned = h->side;
        HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        h->overlap_all = h->tun.overlap > 0;    // Brownian steps fork unless PSE_OVERLAP=0
    }
    TRY(make_plans(h));
    if (h->tun.place_trials > 1 && h->own_z && h->own_y && h->grid_slabs == 1)
        TRY(place_grids(h, (size_t)(G.nxl + G.hl + G.nhalo) * G.Ny * G.Nz, (size_t)G.nxl * G.Ny * G.Nzp));

    TRY(dmalloc(h, &h->V, (size_t)(M_MAX + 1) * n));
    // single GPU: the newest Lanczos vector also in 16 bytes per row (vq_pack): what the pair-list mat-vec gathers its neighbours from
    if (h->n_slabs == 1 && !h->loc.on && h->tun.vq > 0) TRY(dmalloc(h, (char **)&h->vq, (size_t)16 * n));
    TRY(dmalloc(h, &h->scal, (size_t)LZ_NSCAL));
    TRY(dmalloc(h, &h->lz_state, 1));
    HIPCHK(hipMemset(h->lz_state, 0, sizeof(LzState)));
    HIPCHK(hipHostMalloc((void **)&h->sc_host, LZ_NSCAL * sizeof(double), hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer((void **)&h->sc_host_dev, h->sc_host, 0));
    memset(h->sc_host, 0, LZ_NSCAL * sizeof(double));
    HIPCHK(hipEventCreateWithFlags(&h->ev_scal, hipEventDisableTiming)); 
    if (!h->partials) {
        h->npart_cap = std::max(LZ_NPART, mreal_partials_needed((int)n + 1024));   // + the padding of the row ranges (RowMap)
        TRY(dmalloc(h, &h->partials, (size_t)LZ_NGRAM * h->npart_cap));
    }
    for (auto &ph : h->ph) { HIPCHK(hipEventCreate(&ph.a)); HIPCHK(hipEventCreate(&ph.b)); }
    h->info.device_bytes = h->bytes;
    if (h->tun.verbose) {
        // the parameter summary the reference prints at notice level 2 (PSEv1/Stokes.cc:241-252), one block on stderr
        fprintf(stderr, "--- NUFFT Hydrodynamics Statistics ---\nMx: %d\nMy: %d\nMz: %d\nrcut: %.6g\n"
                "Points per radius (x,y,z): %.4g, %.4g, %.4g\n--- Gaussian Spreading Parameters ---\ngauss_m: %.4g\ngauss_P: %d\n"
                "gauss_eta: %.6g\ngauss_w: %.6g\ngauss_gridh (x,y,z): %.6g, %.6g, %.6g\n"
                "--- engine ---\ncells: %d x %d x %d, ranks: %d (far-field slabs %d), device memory: %.2f GB\n",
                d.Nx, d.Ny, d.Nz, d.rcut, d.Nx / h->box.Lx, d.Ny / h->box.Ly, d.Nz / h->box.Lz, d.gaussm, d.P, d.eta,
                d.P * d.hx / 2.0, d.hx, d.hy, d.hz, h->nc.nx, h->nc.ny, h->nc.nz, h->n_slabs, h->grid_slabs, h->bytes / 1e9);
        fprintf(stderr, "grids at: real %p, spectra %p\n", (void *)h->rgrid, (void *)h->cgrid);   // (the x pass at 512^3 has two speeds that go with the allocation: tools/debug/mode_probe.py)
    }
    return 0;
}

extern "C" int pse_create(const pse_params *p, pse_handle **out) {
    if (!p || !out) return fail(PSE_ERR_INVALID, "null argument");
    *out = nullptr;
    pse_handle *h = new pse_handle();
    int r = create_impl(p, h);
    if (r) { std::string keep = error_text(); pse_destroy(h); error_text() = keep; return r; }
    *out = h;
    return 0;
}

extern "C" int pse_set_box(pse_handle *h, double Lx, double Ly, double Lz, double xy) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    if (!(Lx > 0 && Ly > 0 && Lz > 0)) return fail(PSE_ERR_INVALID, "box lengths must be positive");
    if (!(std::fabs(xy) <= 0.5 * (1.0 + 1e-9)))
        return fail(PSE_ERR_INVALID, "tilt xy = %g outside [-0.5, 0.5]: HOOMD flips the box there, and the sequential minimum image "
                    "(PSEv1/Mobility.cu:648) is exact only up to that tilt", xy);
    // Nothing of the handle changes unless every check passes: the candidate cell grid is computed on the side.
    // The grid, P and eta were chosen for the creation box (the reference also fixes them in setParams and only
    // recomputes wave vectors per step, PSEv1/Stokes.cu:298); lengths may change only by re-deriving h.
    const Box nb{Lx, Ly, Lz, xy};
    const double gamma = std::max(std::fabs(xy), h->par.max_strain);
    DCells nc, nc_narrow;
    TRY(cells_for(nb, near_radius(h), gamma, h->n_slabs, h->tun.cell_bz, nc, h->loc.on ? 1 : 0));
    // prepare() switches to the narrow grid (cells of width rcut: more of them) whenever the kept list is suspended or off
    TRY(cells_for(nb, h->d.rcut, gamma, h->n_slabs, h->tun.cell_bz, nc_narrow, h->loc.on ? 1 : 0));
    if (h->loc.on && nc.nx != h->nc.nx)
        return fail(PSE_ERR_INVALID, "an owned-particle rank cannot change its cell layers along x (%d -> %d): the slabs are the ownership", h->nc.nx, nc.nx);
    if ((size_t)std::max(cells_total(nc), cells_total(nc_narrow)) > h->n_cells_alloc)
        return fail(PSE_ERR_INVALID, "box grew beyond the cell-list capacity sized at creation");
    TRY(gaussian_fits(h->d, Lx / h->d.Nx, Ly / h->d.Ny, Lz / h->d.Nz));
    h->box = nb;
    h->nc = nc;
    h->cell_gamma = gamma;
    h->nc_wide = h->skin_max > 0.0;
    // the kept neighbour list is tied to the box it was built in: prepare() compares vl_box with the box of the call
    h->d.hx = Lx / h->d.Nx; h->d.hy = Ly / h->d.Ny; h->d.hz = Lz / h->d.Nz;
    h->G.hx = h->d.hx; h->G.hy = h->d.hy; h->G.hz = h->d.hz;
    set_dbox(h);
    h->info.ncell_x = h->nc.nx; h->info.ncell_y = h->nc.ny; h->info.ncell_z = h->nc.nz;
    h->info.hx = h->d.hx; h->info.hy = h->d.hy; h->info.hz = h->d.hz;
    return 0;
}

extern "C" int pse_set_neighbor_skin(pse_handle *h, double r_buff) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    if (!(r_buff >= 0.0)) return fail(PSE_ERR_INVALID, "r_buff must be >= 0");
    if (r_buff > h->skin_max * (1 + 1e-12))
        return fail(PSE_ERR_INVALID, "r_buff = %g exceeds %g, what the cell grid and the list capacity were sized for at creation "
                    "(0 here: the neighbour list is not kept on this handle -- slab rank, cutoff too large for the box or the table)",
                    r_buff, h->skin_max);
    h->skin = r_buff;
    h->vl_valid = false;
    for (int q = 0; q < 2; ++q) { h->vl_misses[q] = 0; h->vl_suspend_left[q] = 0; h->vl_suspend_len[q] = 32; }
    return 0;
}
extern "C" int pse_neighbor_stats(pse_handle *h, double *r_buff, unsigned long long *builds, unsigned long long *reuses) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    if (r_buff) *r_buff = h->skin;
    if (builds) *builds = h->nlist_builds;
    if (reuses) *reuses = h->nlist_reuses;
    return 0;
}

extern "C" int pse_set_stream(pse_handle *h, void *stream) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    h->stream = (hipStream_t)stream;
    if (!h->side_on) {
        h->wstream = h->stream;
        FFTCHK(rocfft_execution_info_set_stream(h->info_fwd, h->wstream));
        FFTCHK(rocfft_execution_info_set_stream(h->info_inv, h->wstream));
    }
    return 0;
}
extern "C" int pse_set_async(pse_handle *h, int enabled) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    h->async_mode = enabled != 0;
    h->vl_valid = false;
    return 0;
}
extern "C" int pse_set_timestep_offset(pse_handle *h, const unsigned int *device_word) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    h->ts_off = device_word;
    return 0;
}
extern "C" int pse_debug_last_gate(pse_handle *h, int *gate) {
    if (!h || !gate) return fail(PSE_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    *gate = -1;
    if (h->gated) HIPCHK(hipMemcpy(gate, h->gate_word, sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int pse_set_timing(pse_handle *h, int enabled) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    h->timing = enabled != 0;
    return 0;
}
extern "C" int pse_get_info(pse_handle *h, pse_info *info) {
    if (!h || !info) return fail(PSE_ERR_INVALID, "null argument");
    if (h->lz_last_queued && h->lz_seq > 0 && h->sc_host && h->sc_host[LZ_HOST_SEQ] > 0.0) {   // (not after a host-checked call: its numbers are in h->info already)
        // queue-only Brownian calls: what the device-side decision of the most recent COMPLETED call left in the host mirror (the
        // caller synchronises its stream first if it wants the call it has just queued)
        h->info.lanczos_m = (int)h->sc_host[LZ_HOST_M];
        h->info.lanczos_stepnorm = h->sc_host[LZ_HOST_STEPNORM];
        h->info.lanczos_status = (int)h->sc_host[LZ_HOST_STATUS];
    }
    h->info.lanczos_open_calls = h->sc_host ? (unsigned long long)h->sc_host[LZ_HOST_OPEN] : 0ull;
    *info = h->info;
    return 0;
}

// ---- team: the SPMD driver ---------------------------------------------------------------------------------------
// A team is the set of slab ranks this process drives.  Production multi-GPU: one rank per process, team of one
// member, data moved between processes by RCCL on the member's stream.  In-process loopback: all n_slabs ranks are
// members of one team on one device (used to test the decomposition on a single GPU); the "collectives" are device
// copies.  A single GPU is a team of one with n_slabs = 1 and no communication.  Every phase below runs for all local
// members, then the exchange; the phase code is identical in the three cases.
struct pse_team {
    std::vector<pse_handle *> m;
    int G = 1;                   // ranks in the decomposition
    ncclComm_t nccl = nullptr;   // set when members.size() == 1 and G > 1: ONE communicator for every exchange of the team
    // ... and ONE stream all its RCCL calls are issued on, in program order -- the same order on every rank (SPMD code, host-side
    // decisions taken from identical all-reduced numbers).  The two compute lanes (main: sort, near field, Lanczos, update; side:
    // the far-field chain) hand their buffers to it and take them back through events, so the lanes overlap while no two
    // collectives of one communicator are ever in flight in an order that could differ between ranks.
    hipStream_t comm = nullptr;
    std::vector<hipEvent_t> evs; // ring of events for those hand-overs
    size_t ev_next = 0;
    pse_transport cb = {};       // or: a transport supplied by the host program (host-staged; pse_team_create_transport)
    bool has_cb = false;
    double *stage = nullptr;     // pinned staging of the callback transport
    size_t stage_n = 0;
    // developer switch (pse_team_debug_solo): an in-process team queues the work of ONE member only -- its kernels on both lanes,
    // the copies that stand for what it receives -- so that the wall time of a call is that rank's critical path on a GPU of its
    // own (the other members' buffers keep what the last full call left there: the numbers are not meaningful, the timing is)
    int solo = -1;
    std::vector<pse_handle *> solo_m;
    bool lanes = true;           // two compute lanes (PSE_TEAM_LANES=0: one stream for everything, also the RCCL calls)
    int lz_extra = -1;           // pse_team_set_lanczos_extra: iterations an owned-particle step queues beyond its starting count (-1: the members' PSE_LANCZOS_EXTRA)
    bool debug_sync = getenv("PSE_DEBUG_SYNC") != nullptr;   // (read once, when the team is created)
    // self-diagnosis (pse_team_set_diag): every exchange of a call bracketed by events on the lane that issues it, the lanes'
    // spans, the host time the transport's callback took -- read after the call by pse_team_get_diag
    struct Diag {
        bool on = false;
        std::vector<hipEvent_t> ev;          // 2 per exchange slot + 4 for the lanes
        int n = 0;                           // exchanges of the last call
        int kind[PSE_DIAG_MAX]; int lane[PSE_DIAG_MAX]; double host_us[PSE_DIAG_MAX]; unsigned long long bytes[PSE_DIAG_MAX];
        bool side_used = false;
    } diag;
};
enum { DIAG_FIRST = 0, DIAG_LANCZOS = 1, DIAG_ALL_TO_ALL = 2, DIAG_HALO = 3, DIAG_ALL_GATHER = 4, DIAG_GHOST = 5 };
// brackets one exchange with two events on the stream of the lane that issues it (and measures the host time in between)
struct DiagScope {
    pse_team &T; int slot = -1; hipStream_t s; std::chrono::steady_clock::time_point t0;
    DiagScope(pse_team &team, int kind, bool wave_lane, unsigned long long bytes) : T(team) {
        if (!T.diag.on || T.diag.n >= PSE_DIAG_MAX || T.G == 1) return;
        pse_handle *h = T.solo >= 0 ? T.solo_m[0] : T.m[0];
        s = wave_lane ? h->wstream : h->stream;
        slot = T.diag.n++;
        T.diag.kind[slot] = kind; T.diag.lane[slot] = wave_lane ? 1 : 0; T.diag.bytes[slot] = bytes; T.diag.host_us[slot] = 0.0;
        (void)hipEventRecord(T.diag.ev[2 * slot], s);
        t0 = std::chrono::steady_clock::now();
    }
    ~DiagScope() {
        if (slot < 0) return;
        T.diag.host_us[slot] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        (void)hipEventRecord(T.diag.ev[2 * slot + 1], s);
    }
};
static const std::vector<pse_handle *> &act(const pse_team &T) { return T.solo >= 0 ? T.solo_m : T.m; }
#define NCCLCHK(x)                                                                                              \
    do {                                                                                                        \
        ncclResult_t r_ = (x);                                                                                  \
        if (r_ != ncclSuccess) return fail(PSE_ERR_COMM, "%s failed: %s (%s:%d)", #x, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

static bool loopback(const pse_team &T) { return T.G > 1 && !T.nccl && !T.has_cb; }
static bool remote(const pse_team &T) { return T.nccl || T.has_cb; }   // one member per process
static hipEvent_t team_event(pse_team &T) { hipEvent_t e = T.evs[T.ev_next]; T.ev_next = (T.ev_next + 1) % T.evs.size(); return e; }

static void diag_mark(pse_team &T, int which, hipStream_t s) {   // 0 / 1 main lane begin / end, 2 / 3 far-field lane begin / end
    if (T.diag.on) (void)hipEventRecord(T.diag.ev[2 * PSE_DIAG_MAX + which], s);
    if (T.diag.on && which == 2) T.diag.side_used = true;
}
static void diag_begin(pse_team &T) {
    if (!T.diag.on) return;
    T.diag.n = 0; T.diag.side_used = false;
    diag_mark(T, 0, (T.solo >= 0 ? T.solo_m[0] : T.m[0])->stream);
}
// One exchange of a process-per-rank team: a list of point-to-point transfers (counts in doubles; 0 = none) -- ONE RCCL group, or one
// call of the host program's transport.  Every exchange of the team goes through here -- also the partial sums of the Lanczos
// iteration, which travel as small blocks to every rank in the group of the ghost rows (no collective of another kind is ever
// posted) -- so what the RCCL path sends (buffers, counts, peers, order) is exactly what the callback transport sends -- and that
// one runs between real processes in the tests (two ranks cannot share a GPU under RCCL).
struct Xfer { const double *send; size_t ns; int to; double *recv; size_t nr; int from; };
static int team_exchange(pse_team &T, const std::vector<Xfer> &ops, bool wave_lane) {
    pse_handle *h = T.m[0];
    hipStream_t s = wave_lane ? h->wstream : h->stream;
    const int me = h->slab_rank;
    // a rank's sends to itself and its receives from itself pair up in order (the all-to-all's diagonal block): local copies in
    // every transport (nothing of the exchange depends on how a library treats a send to the calling rank)
    std::vector<const Xfer *> self_send, self_recv;
    for (const Xfer &x : ops) {
        if (x.ns && x.to == me) self_send.push_back(&x);
        if (x.nr && x.from == me) self_recv.push_back(&x);
    }
    if (self_send.size() != self_recv.size()) return fail(PSE_ERR_COMM, "unpaired self transfer");
    for (size_t q = 0; q < self_send.size(); ++q) {
        if (self_send[q]->ns != self_recv[q]->nr) return fail(PSE_ERR_COMM, "self transfer of unequal sizes");
        if (self_recv[q]->recv != self_send[q]->send)
            HIPCHK(hipMemcpyAsync(self_recv[q]->recv, self_send[q]->send, self_send[q]->ns * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    bool any = false;
    for (const Xfer &x : ops) any = any || (x.ns && x.to != me) || (x.nr && x.from != me);
    if (!any) return 0;
    if (T.nccl) {
        ncclComm_t comm = T.nccl;
        // hand-over lane -> communication stream (what the lane has queued so far produces the buffers) ...
        hipStream_t cs = T.comm ? T.comm : s;
        if (T.comm) { hipEvent_t e = team_event(T); HIPCHK(hipEventRecord(e, s)); HIPCHK(hipStreamWaitEvent(T.comm, e, 0)); }
        NCCLCHK(ncclGroupStart());
        ncclResult_t bad = ncclSuccess;   // a failing call must not leave the library inside an open group
        for (const Xfer &x : ops) if (bad == ncclSuccess && x.ns && x.to != me) bad = ncclSend(x.send, x.ns, ncclDouble, x.to, comm, cs);
        for (const Xfer &x : ops) if (bad == ncclSuccess && x.nr && x.from != me) bad = ncclRecv(x.recv, x.nr, ncclDouble, x.from, comm, cs);
        const ncclResult_t end = ncclGroupEnd();
        if (bad != ncclSuccess || end != ncclSuccess)
            return fail(PSE_ERR_COMM, "RCCL exchange failed: %s", ncclGetErrorString(bad != ncclSuccess ? bad : end));
        // ... and back: the lane goes on when the exchange has completed
        if (T.comm) { hipEvent_t e = team_event(T); HIPCHK(hipEventRecord(e, T.comm)); HIPCHK(hipStreamWaitEvent(s, e, 0)); }
        return 0;
    }
    // host-staged transport: device -> pinned host, the host program moves the bytes, pinned host -> device
    size_t need = 0;
    for (const Xfer &x : ops) need += (x.to == me ? 0 : x.ns) + (x.from == me ? 0 : x.nr);
    if (need > T.stage_n) {
        if (T.stage) (void)hipHostFree(T.stage);
        T.stage = nullptr; T.stage_n = 0;
        HIPCHK(hipHostMalloc((void **)&T.stage, (need + 1024) * sizeof(double), hipHostMallocDefault));
        T.stage_n = need + 1024;
    }
    std::vector<pse_host_xfer> hx;
    size_t off = 0;
    std::vector<std::pair<double *, const double *>> back;   // (device destination, host source) of what arrives
    std::vector<size_t> back_n;
    for (const Xfer &x : ops) {
        pse_host_xfer e{};
        if (x.ns && x.to != me) { e.send = T.stage + off; e.send_count = x.ns; e.send_to = x.to; HIPCHK(hipMemcpyAsync(T.stage + off, x.send, x.ns * sizeof(double), hipMemcpyDeviceToHost, s)); off += x.ns; }
        if (x.nr && x.from != me) { e.recv = T.stage + off; e.recv_count = x.nr; e.recv_from = x.from; back.push_back({x.recv, T.stage + off}); back_n.push_back(x.nr); off += x.nr; }
        if (e.send_count || e.recv_count) {
            if (!e.send_count) e.send_to = -1;
            if (!e.recv_count) e.recv_from = -1;
            hx.push_back(e);
        }
    }
    HIPCHK(hipStreamSynchronize(s));
    if (!hx.empty() && T.cb.exchange(T.cb.user, (int)hx.size(), hx.data())) return fail(PSE_ERR_COMM, "transport: exchange failed");
    for (size_t q = 0; q < back.size(); ++q)
        HIPCHK(hipMemcpyAsync(back[q].first, back[q].second, back_n[q] * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));   // the staging buffer is reused by the next exchange
    return 0;
}

// the device copies of an in-process exchange, batched into as few launches as the argument space allows
struct CopyBatch {
    CopyList l{};
    hipStream_t s;
    explicit CopyBatch(hipStream_t st) : s(st) { l.n = 0; }
    void add(double *dst, const double *src, size_t n) {
        if (!n || dst == src) return;
        while (n) {   // counts are 32-bit
            const size_t c = std::min<size_t>(n, 0x40000000u);
            l.src[l.n] = src; l.dst[l.n] = dst; l.cnt[l.n] = (unsigned)c;
            if (++l.n == 40) flush();
            src += c; dst += c; n -= c;
        }
    }
    void flush() { if (l.n) launch_copy_list(l, s); l.n = 0; }
    ~CopyBatch() { flush(); }
};

// all-to-all of equal blocks, nset sets at once (one group): member r sends block q of set c of send(r) to rank q,
// which stores it as block r of set c of recv(q); sets are set_stride doubles apart
template <class FS, class FR>
static int team_all_to_all(pse_team &T, FS send, FR recv, size_t blk_doubles, int nset, size_t set_stride) {
    if (T.G == 1) return 0;
    DiagScope dg(T, DIAG_ALL_TO_ALL, true, (unsigned long long)nset * (T.G - 1) * blk_doubles * sizeof(double));
    if (remote(T)) {
        pse_handle *h = T.m[0];
        std::vector<Xfer> ops;
        for (int c = 0; c < nset; ++c)
            for (int q = 0; q < T.G; ++q)
                ops.push_back(Xfer{send(h) + c * set_stride + (size_t)q * blk_doubles, blk_doubles, q,
                                   recv(h) + c * set_stride + (size_t)q * blk_doubles, blk_doubles, q});
        return team_exchange(T, ops, true);
    }
    CopyBatch cb(act(T)[0]->wstream);   // an in-process team shares one stream per lane
    for (int c = 0; c < nset; ++c)
        for (pse_handle *src : T.m)
            for (pse_handle *dst : act(T))
                cb.add(recv(dst) + c * set_stride + (size_t)src->slab_rank * blk_doubles,
                       send(src) + c * set_stride + (size_t)dst->slab_rank * blk_doubles, blk_doubles);
    return 0;
}
// gather halo: every rank stores copies of its left neighbour's last hl planes below its slab and of its right
// neighbour's first nhalo planes above it
static int team_halo_exchange(pse_team &T) {
    if (T.G == 1) return 0;
    DiagScope dg(T, DIAG_HALO, true, (unsigned long long)3 * (T.m[0]->G.hl + T.m[0]->G.nhalo) * T.m[0]->G.Ny * T.m[0]->G.Nz * sizeof(double));
    auto comp = [](pse_handle *h, int c) { return h->rgrid + (size_t)c * (h->G.nxl + h->G.hl + h->G.nhalo) * h->G.Ny * h->G.Nz; };
    if (remote(T)) {
        pse_handle *h = T.m[0];
        const DGrid &G = h->G;
        const size_t plane = (size_t)G.Ny * G.Nz;
        const int left = (h->slab_rank + T.G - 1) % T.G, right = (h->slab_rank + 1) % T.G;
        std::vector<Xfer> ops;
        for (int c = 0; c < 3; ++c) {
            double *own = comp(h, c) + plane * G.hl;
            // my first planes go left and arrive above the left neighbour's slab; my last planes go right and arrive below the right one's
            ops.push_back(Xfer{own, plane * G.nhalo, left, own + plane * G.nxl, plane * G.nhalo, right});
            ops.push_back(Xfer{own + plane * (G.nxl - G.hl), plane * G.hl, right, comp(h, c), plane * G.hl, left});
        }
        return team_exchange(T, ops, true);
    }
    auto member = [&](int r) { for (pse_handle *h : T.m) if (h->slab_rank == r) return h; return (pse_handle *)nullptr; };
    CopyBatch cb(act(T)[0]->wstream);
    for (pse_handle *dst : act(T)) {
        const DGrid &G = dst->G;
        const size_t plane = (size_t)G.Ny * G.Nz;
        pse_handle *L = member((dst->slab_rank + T.G - 1) % T.G), *R = member((dst->slab_rank + 1) % T.G);
        for (int c = 0; c < 3; ++c) {
            cb.add(comp(dst, c) + plane * (G.hl + G.nxl), comp(R, c) + plane * G.hl, plane * G.nhalo);
            cb.add(comp(dst, c), comp(L, c) + plane * (G.hl + G.nxl - G.hl), plane * G.hl);
        }
    }
    return 0;
}

// ghost rows of distributed vectors: for every buffer, a rank receives its right neighbour's first `depth` cell layers and its left
// neighbour's last `depth` layers (the near-field mat-vec of the own rows reads one layer besides the own rows; the two-step
// Lanczos block keeps two).  Row numbers are global, so a block lands at the position it was sent from.
static std::vector<Xfer> ghost_ops(const pse_handle *h, int G, std::initializer_list<double *> bufs, int depth = 1) {
    const std::vector<int> &lo = h->row_lo;
    const std::vector<int> &fe = depth == 1 ? h->first_end : h->first2_end, &lb = depth == 1 ? h->last_begin : h->last2_begin;
    auto cnt = [](int a, int b) { return (size_t)std::max(0, b - a) * 4; };
    const int r = h->slab_rank, L = (r + G - 1) % G, R = (r + 1) % G;
    std::vector<Xfer> ops;
    for (double *buf : bufs) {
        if (!buf) continue;
        // my first layers go left (the left neighbour's right ghost), my last layers go right
        ops.push_back(Xfer{buf + (size_t)lo[r] * 4, cnt(lo[r], fe[r]), L, buf + (size_t)lo[R] * 4, cnt(lo[R], fe[R]), R});
        ops.push_back(Xfer{buf + (size_t)lb[r] * 4, cnt(lb[r], lo[r + 1]), R, buf + (size_t)lb[L] * 4, cnt(lb[L], lo[L + 1]), L});
    }
    return ops;
}
// The partial Lanczos sums of a rank (n numbers at scal[LZ_TMP ..]) go to every rank -- point-to-point blocks in the SAME group as the
// ghost rows (round 4: no separate all-reduce collective; the kernels that consume the sums add the ranks' blocks in rank order)
static void sum_ops(const pse_handle *h, int G, int n, std::vector<Xfer> &ops) {
    for (int q = 0; q < G; ++q)
        ops.push_back(Xfer{h->scal + LZ_TMP, (size_t)n, q, h->sums_all + (size_t)q * LZ_NGRAM, (size_t)n, q});
}
// An exchange given as one transfer list per rank (+ an optional sum over the ranks), in any of the three transports.  In-process
// teams execute the receives as device copies: the k-th receive of dst from src pairs with the k-th send of src to dst -- the
// matching rule of the message transports.
template <class FOPS>
static int team_run_exchange(pse_team &T, FOPS make_ops, bool wave_lane, int kind = DIAG_GHOST) {
    if (T.G == 1) return 0;
    unsigned long long bytes = 0;
    if (T.diag.on) for (const Xfer &x : make_ops(T.solo >= 0 ? T.solo_m[0] : T.m[0])) bytes += (unsigned long long)x.ns * sizeof(double);
    DiagScope dg(T, kind, wave_lane, bytes);
    if (remote(T)) return team_exchange(T, make_ops(T.m[0]), wave_lane);
    std::vector<std::vector<Xfer>> all(T.G);
    for (pse_handle *h : T.m) all[h->slab_rank] = make_ops(h);
    CopyBatch cb(wave_lane ? act(T)[0]->wstream : act(T)[0]->stream);
    for (pse_handle *dst : act(T)) {
        const int d = dst->slab_rank;
        std::vector<size_t> taken(T.G, 0);   // sends of each source already paired with a receive of dst
        for (const Xfer &rx : all[d]) {
            if (!rx.nr) continue;
            const std::vector<Xfer> &so = all[rx.from];
            size_t &k = taken[rx.from];
            while (k < so.size() && !(so[k].ns && so[k].to == d)) ++k;
            if (k == so.size() || so[k].ns != rx.nr) return fail(PSE_ERR_COMM, "in-process exchange: unmatched transfer");
            cb.add(rx.recv, so[k].send, rx.nr);
            ++k;
        }
    }
    return 0;
}
template <class FB>
static int team_ghost_exchange(pse_team &T, FB buf) {
    return team_run_exchange(T, [&](pse_handle *h) { return ghost_ops(h, T.G, {buf(h)}); }, false);
}

// One exchange per Lanczos iteration: the three partial sums (all-reduce) and the ghost rows of y = M x (every rank receives its
// right neighbour's first cell layer and its left neighbour's last one) travel in ONE group.
template <class FB>
static int team_lanczos_exchange(pse_team &T, FB buf) {
    return team_run_exchange(T, [&](pse_handle *h) { auto ops = ghost_ops(h, T.G, {buf(h)}); sum_ops(h, T.G, 3, ops); return ops; }, false, DIAG_LANCZOS);
}

// every rank's own rows [row_lo[r], row_lo[r+1]) of buf become visible on every rank (blocks of different sizes)
template <class FB>
static int team_all_gather_rows(pse_team &T, FB buf) {
    if (T.G == 1) return 0;
    const std::vector<int> &lo = T.m[0]->row_lo;
    DiagScope dg(T, DIAG_ALL_GATHER, false, (unsigned long long)(T.G - 1) * (lo[T.m[0]->slab_rank + 1] - lo[T.m[0]->slab_rank]) * 32ull);
    if (remote(T)) {
        pse_handle *h = T.m[0];
        const int r = h->slab_rank;
        std::vector<Xfer> ops;
        for (int q = 0; q < T.G; ++q)
            if (q != r)
                ops.push_back(Xfer{buf(h) + (size_t)lo[r] * 4, (size_t)(lo[r + 1] - lo[r]) * 4, q,
                                   buf(h) + (size_t)lo[q] * 4, (size_t)(lo[q + 1] - lo[q]) * 4, q});
        return team_exchange(T, ops, false);
    }
    CopyBatch cb(act(T)[0]->stream);
    for (pse_handle *src : T.m)
        for (pse_handle *dst : act(T))
            if (src != dst) {
                const int r = src->slab_rank;
                cb.add(buf(dst) + (size_t)lo[r] * 4, buf(src) + (size_t)lo[r] * 4, (size_t)(lo[r + 1] - lo[r]) * 4);
            }
    return 0;
}

// ---- phases ---------------------------------------------------------------------------------------------------
static int check_n(pse_handle *h, unsigned N) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    if (N == 0 || (int)N > h->n_max) return fail(PSE_ERR_INVALID, "N = %u outside (0, n_max = %d]", N, h->n_max);
    HIPCHK(hipSetDevice(h->device));
    return 0;
}

// rows of the particle arrays this rank owns: the particles of its cell slab (contiguous in the cell-sorted order)
static void row_range(const pse_handle *h, int N, int &lo, int &hi) {
    if (h->n_slabs == 1) { lo = 0; hi = N; return; }
    lo = h->row_lo[h->slab_rank]; hi = h->row_lo[h->slab_rank + 1];
}

// after the sort: where the cell slabs begin in the sorted arrays (3G+1 ints; identical on every rank because the particle
// arrays are replicated).  The copy back is issued here and awaited only where the host first needs the numbers -- after the
// far-field chain has been queued, so the GPU does not idle meanwhile.
static int slab_bounds_issue(pse_handle *h) {
    const int G = h->n_slabs;
    if (G == 1) return 0;
    const int layer = h->nc.nzb * h->nc.ny * h->nc.bz, per = h->nc.nx / G;   // storage cells of one x layer
    std::vector<int> idx(5 * G + 1);
    for (int r = 0; r <= G; ++r) idx[r] = r * per * layer;
    for (int r = 0; r < G; ++r) {
        idx[G + 1 + r] = (r * per + 1) * layer;                       // end of the first layer, begin of the last
        idx[2 * G + 1 + r] = ((r + 1) * per - 1) * layer;
        idx[3 * G + 1 + r] = (r * per + std::min(2, per)) * layer;    // ... of the first / last two layers
        idx[4 * G + 1 + r] = ((r + 1) * per - std::min(2, per)) * layer;
    }
    if (h->bidx_nc.nx != h->nc.nx || h->bidx_nc.ny != h->nc.ny || h->bidx_nc.nz != h->nc.nz) {   // once per cell grid
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(h->d_bidx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
        h->bidx_nc = h->nc;
    }
    launch_pick(h->cell_off, h->d_bidx, (int)idx.size(), h->bounds_host_dev, h->stream);   // straight into the mapped host buffer: no copy to queue
    HIPCHK(hipEventRecord(h->ev_bounds, h->stream));
    h->bounds_pending = true;
    return 0;
}
static int slab_bounds_wait(pse_handle *h, int N) {
    if (!h->bounds_pending) return 0;
    h->bounds_pending = false;
    const int G = h->n_slabs;
    HIPCHK(hipEventSynchronize(h->ev_bounds));
    const int *out = h->bounds_host;
    h->row_lo.assign(out, out + G + 1);
    h->first_end.assign(out + G + 1, out + 2 * G + 1);
    h->last_begin.assign(out + 2 * G + 1, out + 3 * G + 1);
    h->first2_end.assign(out + 3 * G + 1, out + 4 * G + 1);
    h->last2_begin.assign(out + 4 * G + 1, out + 5 * G + 1);
    if (h->row_lo[0] != 0 || h->row_lo[G] != N) return fail(PSE_ERR_NUMERIC, "inconsistent cell offsets after the sort");
    return 0;
}

// Two Lanczos iterations per exchange (lanczos_team): needs the pair list, the table in LDS (the pass that builds the list then
// delivers M psi with it), at least two cell layers per rank and two ghost layers on either side that are not the rank's own.
static bool team_sstep(const pse_handle *h) {
    const int G = h->n_slabs, per = h->nc.nx / std::max(1, G);
    return G > 1 && h->tun.team_sstep && h->w2_s && h->nb.cap > 0 && mreal_table_in_lds(h->n_intervals * 2 * RS_NCOEF) && per >= 2 &&
           h->nc.nx >= per + 4;
}

// What a slab rank orders, gathers into cell order and keeps particle data for: its own cell layers and one ghost layer on either
// side (CellRanges in pse_kernels.h).  Everything, when the far field is replicated on every rank, when the slab and its ghosts
// cover all layers anyway, or when a support could reach further than a ghost layer is wide (a particle beyond the ghost layer
// must not touch the slab's planes: (P/2 + 1) node planes against the width of a cell layer).
static CellRanges slab_need(const pse_handle *h) {
    CellRanges r{};
    const int G = h->n_slabs, nx = h->nc.nx;
    if (G == 1 || h->grid_slabs == 1) return r;
    const int per = nx / G, layer = h->nc.nzb * h->nc.ny * h->nc.bz;
    const int depth = team_sstep(h) ? 2 : 1;   // ghost layers kept on either side
    if (per + 2 * depth >= nx) return r;
    if ((h->G.P / 2 + 1.0) / h->G.Nx > 1.0 / nx) return r;
    const int start = ((h->slab_rank * per - depth) % nx + nx) % nx, len = per + 2 * depth;
    if (start + len <= nx) { r.n = 1; r.c0[0] = start * layer; r.c1[0] = (start + len) * layer; }
    else { r.n = 2; r.c0[0] = start * layer; r.c1[0] = nx * layer; r.c0[1] = 0; r.c1[1] = (start + len - nx) * layer; }
    return r;
}

// bin + sort + gather into cell order (positions change every step, so this runs every call; every rank COUNTS all particles
// -- the state is replicated and the row offsets are global -- but orders and gathers only what slab_need() says)
// need_cells: the caller walks the cell list itself (pair repulsion): sort even if the neighbour list could be kept.
//
// With a neighbour skin the sort and the cell walk run only when the distance check says so (the reference keeps HOOMD's
// NeighborList with r_buff = 0.4 and setEvery(1, dist_check), PSEv1/integrate.py:60,79, and calls m_nlist->compute every
// step, Stokes.cc:433): the particles are gathered into the order of the last sort, every one is compared with where it
// was at the build, and one flag comes back to the host.  Kept: perm, the neighbour list; rebuilt as before: the far-field
// records, the per-step (f, h) pair list.
struct PrepExtra { bool psi; unsigned timestep; };   // psi: the pass also draws the particle noise of this step into psi_s
static int prepare(pse_handle *h, const double4 *pos, const double4 *vec, const unsigned *group, int N, bool defer_bounds = false,
                   bool need_cells = false, PrepExtra px = PrepExtra{false, 0}, bool with_real = true) {
    TRY(ts(h, PH_SORT));
    const FarBinArgs far = far_bin_args(h->G, h->sw);
    double4 *psi_out = px.psi ? h->psi_s : nullptr;
    h->nb_valid = false;
    h->w_is_mpsi = false; h->sums0_done = false;
    h->vl_use = false;
    h->vl_pending = false;
    h->pv_is_f = vec != nullptr && h->pv != nullptr;
    const bool same_box = h->vl_box.Lx == h->box.Lx && h->vl_box.Ly == h->box.Ly && h->vl_box.Lz == h->box.Lz && h->vl_box.xy == h->box.xy;
    h->gated = false;
    // (only when the call has a near-field part: the rebuild chain's list and build positions are written by real(); a far-field-only
    // call that took it would leave a list in the OLD order marked valid -- such a call rebuilds, ungated)
    if (h->async_mode && with_real && !px.psi && h->skin > 0.0 && h->vl_valid && !need_cells && h->vl_N == N && h->vl_group == group && same_box &&
        h->sorted_N == N && h->nc_wide && h->n_slabs == 1) {
        // Asynchronous mode, a deterministic evaluation, a kept list exists: the decision "reuse or rebuild" is taken on the device and
        // BOTH chains are queued -- the kernels of the chain not taken read the gate word and leave at once.  No read-back, nothing
        // host-side depends on the outcome, so the call can be captured into a hipGraph and replayed with any positions.
        //   ungated: gather into the order of the last build + distance check -> flags; decide -> gate word
        //   rebuild chain (gate != 0): zero the bin and cell counts, cell sort, gather into the new order, [real(): cell pass that
        //                              writes the list, positions of the build]
        //   reuse chain (gate == 0):   [real(): the kept-list pass]
        const int ncell = cells_total(h->nc);
        HIPCHK(hipMemsetAsync(h->cnt_block, 0, h->cnt_bins * sizeof(int), h->stream));   // bin counts + flags[0]
        launch_permute(pos, vec, group, h->perm, N, h->dbox, h->pos_s, h->posf_s, h->pv, h->f_s, h->tag_s, h->stream,
                       h->pos_build, 0.25 * h->skin * h->skin, h->vl.flags, CellRanges{}, nullptr, &far, nullptr, 0, 0);
        launch_gate_decide(h->vl.flags, h->gate_word, h->stream);
        const Gate rb{h->gate_word, 1};
        launch_gate_zero(rb, h->cnt_block, h->cnt_bins - 1, h->cell_cnt, (size_t)ncell + 1, h->stream);   // (not the flags)
        HIPCHK(cell_sort(pos, group, N, h->dbox, h->nc, h->keys, h->vals, h->keys_s, h->cell_cnt, ncell, h->sort_tmp, h->sort_tmp_bytes,
                         h->cell_off, h->perm, h->stream, CellRanges{}, SlabBook{}, true, rb));
        launch_permute(pos, vec, group, h->perm, N, h->dbox, h->pos_s, h->posf_s, h->pv, h->f_s, h->tag_s, h->stream, nullptr, 0.0, nullptr,
                       CellRanges{}, h->cell_off, &far, nullptr, 0, 0, rb);
        h->gated = true;
        h->vl.rskin = h->d.rcut + h->skin;
        h->sw.need = CellRanges{}; h->sw.cell_off = h->cell_off;
        ++h->nlist_builds;   // (counted as a build: which chain ran is known on the device only)
        TRY(te(h, PH_SORT));
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (h->skin > 0.0 && !h->async_mode && h->vl_valid && !need_cells && h->vl_N == N && h->vl_group == group && same_box && h->sorted_N == N) {
        HIPCHK(hipMemsetAsync(h->cnt_block, 0, h->cnt_bins * sizeof(int), h->stream));   // bin counts + flags[0] ([1] is the build's overflow mark)
        launch_permute(pos, vec, group, h->perm, N, h->dbox, h->pos_s, h->posf_s, h->pv, h->f_s, h->tag_s, h->stream,
                       h->pos_build, 0.25 * h->skin * h->skin, h->vl.flags, CellRanges{}, nullptr, &far, psi_out, h->par.seed, px.timestep, Gate{}, h->ts_off);
        HIPCHK(hipMemcpyAsync(h->flags_host, h->vl.flags, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->flags_host[0] == 0 && h->flags_host[1] == 0) {
            h->vl_use = true;
            ++h->nlist_reuses;
            ++h->vl_reused_since_build;
            h->vl_misses[h->vl_kind] = 0; h->vl_suspend_len[h->vl_kind] = 32;
            TRY(te(h, PH_SORT));
            return 0;
        }
        const int q = h->vl_kind;
        if (h->vl_reused_since_build == 0 && ++h->vl_misses[q] >= 2) {   // built twice for nothing: pause
            h->vl_suspend_left[q] = h->vl_suspend_len[q];
            h->vl_suspend_len[q] = std::min(2 * h->vl_suspend_len[q], 4096);
            h->vl_misses[q] = 1;                                     // one more miss after the pause suspends again
        }
    }
    h->vl_valid = false;
    ++h->nlist_builds;
    // (asynchronous mode: no host-side feedback, hence no suspension logic -- deterministic evaluations always keep the list, for the
    // device-side decision of the next call; Brownian calls, which cannot take that path, do not)
    const bool with_list = h->skin > 0.0 && (h->async_mode ? !px.psi : h->vl_suspend_left[h->vl_kind] == 0);
    if (h->skin > 0.0 && !h->async_mode && !with_list) --h->vl_suspend_left[h->vl_kind];
    if (h->skin_max > 0.0) {   // cells as wide as this build reaches: rcut + r_buff with the list, rcut without
        const bool wide = h->nc_wide;
        if (wide != with_list) {
            DCells nc;
            TRY(cells_for(h->box, h->d.rcut + (with_list ? h->skin_max : 0.0), h->cell_gamma, h->n_slabs, h->tun.cell_bz, nc));
            h->nc = nc; h->nc_wide = with_list;
            h->info.ncell_x = nc.nx; h->info.ncell_y = nc.ny; h->info.ncell_z = nc.nz;
        }
    }
    const int ncell = cells_total(h->nc);
    if ((size_t)ncell > h->n_cells_alloc)   // pse_create and pse_set_box check both cell grids: cannot happen, must not pass silently
        return fail(PSE_ERR_INVALID, "cell grid %d x %d x %d exceeds the capacity sized at creation", h->nc.nx, h->nc.ny, h->nc.nz);
    const CellRanges need = slab_need(h);
    h->sw.need = need; h->sw.cell_off = h->cell_off;
    SlabBook sb{};
    if (need.n > 0 && h->n_slabs <= 64) {   // where the particles of the layers this rank does not keep are counted (cell_sort)
        auto kept = [&](int c) { for (int q = 0; q < need.n; ++q) if (c >= need.c0[q] && c < need.c1[q]) return true; return false; };
        const int layer = h->nc.nzb * h->nc.ny * h->nc.bz, per = h->nc.nx / h->n_slabs;
        sb.n = h->n_slabs; sb.cells_per_slab = per * layer; sb.spread = std::max(1, std::min(1024, layer));
        for (int q = 0; q < h->n_slabs; ++q) {
            sb.book[q] = q * per * layer;   // (a slab kept whole has no foreign particles: never used)
            for (int l = 0; l < per; ++l)
                if (!kept((q * per + l) * layer)) { sb.book[q] = (q * per + l) * layer; break; }
        }
    }
    // one memset: the bin counts, both flags of the kept list, the cell counts (+ the sentinel)
    HIPCHK(hipMemsetAsync(h->cnt_block, 0, ((h->cnt_bins + 4 + (size_t)ncell + 1 + 3) & ~(size_t)3) * sizeof(int), h->stream));   // (rounded up into the padding)
    HIPCHK(cell_sort(pos, group, N, h->dbox, h->nc, h->keys, h->vals, h->keys_s, h->cell_cnt, ncell, h->sort_tmp, h->sort_tmp_bytes,
                     h->cell_off, h->perm, h->stream, need, sb, true));
    launch_permute(pos, vec, group, h->perm, N, h->dbox, h->pos_s, h->posf_s, h->pv, h->f_s, h->tag_s, h->stream, nullptr, 0.0, nullptr,
                   need, h->cell_off, &far, psi_out, h->par.seed, px.timestep, Gate{}, h->ts_off);
    h->sorted_N = N;
    if (with_list) {   // the first cell pass of this call writes the list (real())
        h->vl_pending = true; h->vl_N = N; h->vl_group = group;
        h->vl.rskin = h->d.rcut + h->skin;
        h->vl_reused_since_build = 0;
    }
    TRY(slab_bounds_issue(h));
    if (!defer_bounds) TRY(slab_bounds_wait(h, N));
    TRY(te(h, PH_SORT));
    HIPCHK(hipGetLastError());
    return 0;
}

static ScaleArgs scale_args(pse_handle *h, bool noise, double kT, double dt, unsigned timestep) {
    ScaleArgs a;
    const DGrid &G = h->G;
    a.xi = h->d.xi; a.eta = h->d.eta; a.noise = noise ? 1 : 0;
    a.noise_fac = noise ? std::sqrt(2.0 * kT / dt / (G.hx * G.hy * G.hz)) : 0.0;   // PSEv1/Brownian.cu:197
    a.seed = h->par.seed; a.timestep = timestep; a.ts_off = h->ts_off;
    a.transposed = h->grid_slabs > 1 ? 1 : 0; a.y0 = h->y0; a.nyl = h->grid_slabs > 1 ? h->nyl : G.Ny;
    a.runtime_plan = h->tun.xmix_runtime; a.wide_small = h->tun.xfft_small_wide;
    return a;
}

// wave-space part: spread -> FFT -> scale (+ noise) -> inverse FFT -> gather  (PSEv1/Brownian.cu:836-872), cut where a
// slab-decomposed team exchanges data: compute part k, exchange k, compute part k + 1, ...
//   part 0  records, spread, 2-D forward transforms (+ pack)         exchange 0  all-to-all (y rows of every x plane -> their rank)
//   part 1  x transforms + k-space scaling (+ noise)                 exchange 1  all-to-all back
//   part 2  (unpack +) 2-D inverse transforms                        exchange 2  plane halo of the gather
//   part 3  gather
// A single GPU, or a team that keeps the whole grid on every rank, has no exchanges: the parts simply follow one another.
struct WaveArgs { int N; bool noise; double kT, dt; unsigned timestep; };
static int wave_compute(pse_team &T, const WaveArgs &a, int part) {
    const int GS = T.m[0]->grid_slabs;   // 1: every rank transforms the whole grid (single GPU, or a team that replicates it)
    for (pse_handle *h : act(T)) {
        const DGrid &G = h->G;
        const size_t nr = (size_t)(G.nxl + G.hl + G.nhalo) * G.Ny * G.Nz, ncx = (size_t)G.nxl * G.Ny * G.Nzp;
        if (part == 0) {
            double *gx = h->rgrid, *gy = h->rgrid + nr, *gz = h->rgrid + 2 * nr;
            TRY(tsw(h, PH_RECORDS));
            HIPCHK(launch_far_records(h->pos_s, h->f_s, a.N, G, h->dbox, h->sw, h->wstream));
            TRY(tew(h, PH_RECORDS));
            TRY(tsw(h, PH_SPREAD));
            if (spread_needs_zero(G)) HIPCHK(hipMemsetAsync(h->rgrid, 0, 3 * nr * sizeof(double), h->wstream));
            HIPCHK(launch_spread(h->pos_s, h->f_s, a.N, gx, gy, gz, G, h->dbox, h->sw, h->wstream));
            TRY(tew(h, PH_SPREAD));
            TRY(tsw(h, PH_FFTF));
            double *zr[3]; double2 *zs[3];   // the rows of the own planes of each component
            for (int c = 0; c < 3; ++c) { zr[c] = h->rgrid + c * nr + (size_t)G.hl * G.Ny * G.Nz; zs[c] = h->cgrid + c * ncx; }
            if (GS == 1) {
                if (h->own_z) launch_zfft(zr, zs, G.Nx * G.Ny, G.Nz, G.Nzp, false, h->twiddle_z, h->wstream);
                else {
                    void *in[1] = {h->rgrid}, *out[1] = {h->cgrid};
                    FFTCHK(rocfft_execute(h->plan_fwd, in, out, h->info_fwd));
                }
                if (h->own_y) launch_yfft(h->cgrid, G, false, h->twiddle_y, h->wstream, h->tun.yfft_kb);
            } else {
                if (h->own_z) launch_zfft(zr, zs, G.nxl * G.Ny, G.Nz, G.Nzp, false, h->twiddle_z, h->wstream);
                else for (int c = 0; c < 3; ++c) {   // 2-D (y,z) transforms of the local planes, one component at a time
                    void *in[1] = {h->rgrid + c * nr + (size_t)G.hl * G.Ny * G.Nz}, *out[1] = {h->cgrid + c * ncx};
                    FFTCHK(rocfft_execute(h->plan_fwd, in, out, h->info_fwd));
                }
                if (h->own_y_slab) launch_yfft_slab(h->cgrid, h->sendbuf, G, h->nyl, false, h->twiddle_y, h->wstream, h->tun.yslab_regs > 0);
                else launch_slab_pack(h->cgrid, h->sendbuf, G.nxl, G.Ny, G.Nzp, h->nyl, 0, h->wstream);
            }
            TRY(tew(h, PH_FFTF));
        } else if (part == 1) {
            double2 *sp = GS == 1 ? h->cgrid : h->recvbuf;   // [3][Nx][nyl][Nzh] after the transpose
            TRY(tsw(h, PH_SCALE));
            const ScaleArgs sa = scale_args(h, a.noise, a.kT, a.dt, a.timestep);
            if (h->xfuse) {
                launch_xfft_scale(sp, sp + ncx, sp + 2 * ncx, G, h->dbox, sa, h->twiddle, h->wstream);
            } else {
                if (GS > 1)
                    for (int c = 0; c < 3; ++c) { void *io[1] = {sp + c * ncx}; FFTCHK(rocfft_execute(h->plan_x_fwd, io, nullptr, h->info_fwd)); }
                launch_scale(sp, sp + ncx, sp + 2 * ncx, G, h->dbox, sa, h->wstream);
                if (GS > 1)
                    for (int c = 0; c < 3; ++c) { void *io[1] = {sp + c * ncx}; FFTCHK(rocfft_execute(h->plan_x_inv, io, nullptr, h->info_inv)); }
            }
            TRY(tew(h, PH_SCALE));
        } else if (part == 2) {
            TRY(tsw(h, PH_FFTI));
            double *zr[3]; double2 *zs[3];
            for (int c = 0; c < 3; ++c) { zr[c] = h->rgrid + c * nr + (size_t)G.hl * G.Ny * G.Nz; zs[c] = h->cgrid + c * ncx; }
            if (GS == 1) {
                if (h->own_y) launch_yfft(h->cgrid, G, true, h->twiddle_y, h->wstream, h->tun.yfft_kb);
                if (h->own_z) launch_zfft(zr, zs, G.Nx * G.Ny, G.Nz, G.Nzp, true, h->twiddle_z, h->wstream);
                else {
                    void *in[1] = {h->cgrid}, *out[1] = {h->rgrid};
                    FFTCHK(rocfft_execute(h->plan_inv, in, out, h->info_inv));
                }
            } else {
                if (h->own_y_slab) launch_yfft_slab(h->cgrid, h->sendbuf, G, h->nyl, true, h->twiddle_y, h->wstream, h->tun.yslab_regs > 0);
                else launch_slab_pack(h->cgrid, h->sendbuf, G.nxl, G.Ny, G.Nzp, h->nyl, 1, h->wstream);
                if (h->own_z) launch_zfft(zr, zs, G.nxl * G.Ny, G.Nz, G.Nzp, true, h->twiddle_z, h->wstream);
                else for (int c = 0; c < 3; ++c) {
                    void *in[1] = {h->cgrid + c * ncx}, *out[1] = {h->rgrid + c * nr + (size_t)G.hl * G.Ny * G.Nz};
                    FFTCHK(rocfft_execute(h->plan_inv, in, out, h->info_inv));
                }
            }
            TRY(tew(h, PH_FFTI));
        } else {
            TRY(tsw(h, PH_GATHER));
            HIPCHK(launch_gather(h->pos_s, h->sw, a.N, h->rgrid, h->rgrid + nr, h->rgrid + 2 * nr, G, h->dbox, h->uw_s, h->wstream));
            TRY(tew(h, PH_GATHER));
            // every particle was gathered by the rank that owns its row (zeros elsewhere)
        }
    }
    return 0;
}
static int wave_exchange(pse_team &T, int k) {
    pse_handle *h0 = T.m[0];
    if (h0->grid_slabs == 1) return 0;
    const size_t blk = (size_t)h0->G.nxl * h0->nyl * h0->G.Nzp * 2, comp = blk * h0->grid_slabs;
    if (k == 0) {
        for (pse_handle *h : act(T)) TRY(tsw(h, PH_COMM));
        TRY(team_all_to_all(T, [&](pse_handle *h) { return (double *)h->sendbuf; }, [&](pse_handle *h) { return (double *)h->recvbuf; }, blk, 3, comp));
        for (pse_handle *h : act(T)) TRY(tew(h, PH_COMM));
        return 0;
    }
    if (k == 1)
        return team_all_to_all(T, [&](pse_handle *h) { return (double *)h->recvbuf; }, [&](pse_handle *h) { return (double *)h->sendbuf; }, blk, 3, comp);
    return team_halo_exchange(T);
}
// The far-field chain as the main lane's driver sees it: compute parts are queued as early as the data allows, exchange k of the
// chain is ISSUED when the driver reaches its slot -- just before Lanczos exchange number slot[k] -- so that the order of the
// collectives on the team's one communication stream is a fixed interleaving of the two lanes, the same on every rank.
struct WavePump {
    pse_team *T = nullptr;
    WaveArgs a{};
    int next = 4;                   // next compute part to queue (4: the chain is complete, or not part of this call)
    int slot[3] = {0, 0, 0};
    unsigned *mask = nullptr;
    bool forked = false;
    // the side lane may start from here (the sorted arrays exist); what the caller queues on the main lane afterwards runs NEXT to the chain
    int fork(pse_team &team) {
        for (pse_handle *h : act(team))
            if (h->side_on) {
                HIPCHK(hipEventRecord(h->ev_fork, h->stream));
                HIPCHK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
                if (h == act(team)[0]) diag_mark(team, 2, h->side);
            }
        forked = true;
        return 0;
    }
    int start(pse_team &team, const WaveArgs &args, const int sched[3], unsigned *m) {
        T = &team; a = args; mask = m;
        for (int k = 0; k < 3; ++k) slot[k] = sched[k];
        if (!forked) TRY(fork(team));
        *mask |= (1u << PH_SPREAD) | (1u << PH_FFTF) | (1u << PH_SCALE) | (1u << PH_FFTI) | (1u << PH_GATHER) | (1u << PH_RECORDS);
        if (T->m[0]->grid_slabs > 1) *mask |= 1u << PH_COMM;
        TRY(wave_compute(*T, a, 0));
        next = 1;
        if (T->m[0]->grid_slabs == 1) return upto(1 << 30);   // no exchanges: the whole chain is queued at once
        return 0;
    }
    // called before Lanczos exchange e is issued (and with a huge e to finish the chain)
    int upto(int e) {
        while (T && next <= 3 && slot[next - 1] <= e) {
            TRY(wave_exchange(*T, next - 1));
            TRY(wave_compute(*T, a, next));
            ++next;
        }
        return 0;
    }
    int drain() { return upto(1 << 30); }
};

// rows of a rank with `depth` ghost cell layers on either side: its own rows first, then its left neighbour's last layers and its
// right neighbour's first layers (duplicates dropped: with two ranks and one layer each, both ghosts are the same rows)
static int row_ranges(const pse_handle *h, int N, int depth, int rg[3][2]) {
    if (h->n_slabs == 1) { rg[0][0] = 0; rg[0][1] = N; return 1; }
    const int G = h->n_slabs, r = h->slab_rank, L = (r + G - 1) % G, R = (r + 1) % G;
    int n = 0;
    auto add = [&](int a, int b) {
        if (b <= a) return;
        for (int q = 0; q < n; ++q) if (rg[q][0] == a && rg[q][1] == b) return;
        rg[n][0] = a; rg[n][1] = b; ++n;
    };
    rg[0][0] = h->row_lo[r]; rg[0][1] = h->row_lo[r + 1]; n = 1;   // the own rows are range 0 even when empty (list rows start there)
    if (depth >= 1) {
        add((depth == 1 ? h->last_begin : h->last2_begin)[L], h->row_lo[L + 1]);
        add(h->row_lo[R], (depth == 1 ? h->first_end : h->first2_end)[R]);
    }
    return n;
}
static RowMap rank_rows(const pse_handle *h, int N, int depth) {
    int rg[3][2];
    const int n = row_ranges(h, N, depth, rg);
    return row_map(rg, n);
}

// near-field mat-vec out = M_real vec (PSEv1/Mobility.cu:594-687) on the rows this rank owns (+ `depth` ghost layers); vec must be
// valid on those rows and on the cell layer beyond them.  build_list: also record the pair list for later mat-vecs of this step.
static int real(pse_team &T, double4 *pse_handle::*vec, double4 *pse_handle::*out, size_t vec_off, size_t out_off, int N,
                bool build_list, bool with_psi = false, int depth = 0) {
    for (pse_handle *h : act(T)) {
        h->w_is_mpsi = false;
        int mode = MREAL_CELLS;
        if (h->nb.cap > 0) {
            if (h->nb_valid) mode = MREAL_USE_LIST;
            else if (build_list) mode = MREAL_BUILD_LIST;
        }
        // the kept neighbour list: used when this call runs on it, written by the first cell pass after a sort
        int vlm = VL_NONE;
        if (h->vl_use) vlm = VL_USE;
        else if (h->vl_pending && mode != MREAL_USE_LIST) vlm = VL_WRITE;
        const double4 *v = h->*vec + vec_off;
        if (h->gated && mode == MREAL_CELLS) {   // both chains, gated on the device-side decision of prepare()
            const RowMap rows = rank_rows(h, N, 0);
            const int nco = h->n_intervals * 2 * RS_NCOEF;
            const Gate rb{h->gate_word, 1}, ru{h->gate_word, 0};
            launch_mreal(h->pos_s, h->posf_s, v, h->*out + out_off, rows, h->cell_off, h->dbox, h->nc, h->d.rcut, h->d.self, h->coef, nco,
                         h->nb, MREAL_CELLS, h->stream, nullptr, nullptr, h->vl, VL_WRITE, nullptr, nullptr, 0, nullptr, rb);
            launch_gate_copy(rb, h->pos_build, h->pos_s, (size_t)N, h->stream);
            launch_mreal(h->pos_s, h->posf_s, v, h->*out + out_off, rows, h->cell_off, h->dbox, h->nc, h->d.rcut, h->d.self, h->coef, nco,
                         h->nb, MREAL_CELLS, h->stream, nullptr, nullptr, h->vl, VL_USE,
                         v == h->f_s && h->pv_is_f ? h->pv : nullptr, nullptr, 0, nullptr, ru);
            h->vl_valid = true; h->vl_box = h->box;
            continue;
        }
        const bool psi = with_psi && mode == MREAL_BUILD_LIST;
        launch_mreal(h->pos_s, h->posf_s, v, h->*out + out_off, rank_rows(h, N, mode == MREAL_BUILD_LIST ? depth : 0), h->cell_off, h->dbox,
                     h->nc, h->d.rcut, h->d.self, h->coef, h->n_intervals * 2 * RS_NCOEF, h->nb, mode, h->stream,
                     psi ? h->psi_s : nullptr, h->w_s, h->vl, vlm,
                     v == h->f_s && h->pv_is_f && (vlm == VL_USE || mode != MREAL_USE_LIST) ? h->pv : nullptr,   // packed (position, F) records
                     psi && h->n_slabs == 1 ? h->partials : nullptr, h->npart_cap, h->scal);   // single GPU: + the sums of Lanczos iteration 0
        h->sums0_done = psi && h->n_slabs == 1 && vlm != VL_USE && mreal_table_in_lds(h->n_intervals * 2 * RS_NCOEF);
        if (vlm == VL_WRITE) {   // the list now matches perm, pos_s and the box of this call
            HIPCHK(hipMemcpyAsync(h->pos_build, h->pos_s, (size_t)N * sizeof(double4), hipMemcpyDeviceToDevice, h->stream));
            h->vl_pending = false; h->vl_valid = true; h->vl_box = h->box;
        }
        if (mode == MREAL_BUILD_LIST) { h->nb_valid = true; h->w_is_mpsi = with_psi && (vlm == VL_USE || mreal_table_in_lds(h->n_intervals * 2 * RS_NCOEF)); }
    }
    return 0;
}

// rows a rank updates in a Lanczos iteration: its own and the ghost layers its next mat-vec reads
static int update_ranges(const pse_handle *h, int N, int rg[3][2]) { return row_ranges(h, N, 1, rg); }

// Queue-only Lanczos of a single GPU (pse_set_async): the iterations of the starting count are queued as the host-driven loop
// below queues them; then ONE launch takes the decision the host would take (k_lz_decide: the tridiagonal square roots of the two
// sizes m_in - 1 and m_in, the step norm), and `extra` more iterations follow, each with its own decision, every kernel of them
// gated on the outcome so far -- the reference iterates until the step norm passes (PSEv1/Brownian.cu:606-724); here whatever is
// not needed leaves at once.  The final combination takes m and its coefficients from the device.  Nothing is read back: the
// call can be captured into a hipGraph; m of a call reaches the host lazily (pse_get_info after a synchronisation, or the
// lanczos_m of the next call).  If the queue runs out before the step norm passes, the result uses the last size and
// pse_info.lanczos_status says 1.
static int lanczos_queued(pse_team &T, int N, double tol, double scale, int *m_io, const std::function<int()> &before_first_wait,
                          const std::function<int()> &before_combine) {
    pse_handle *h = T.m[0];
    const size_t stride = h->n_pad;
    const int m_in = std::min(std::max(m_io ? *m_io : 2, 1), M_MAX);
    const int target = std::max(m_in, 2);
    const int extra = std::max(0, std::min(h->lz_extra >= 0 ? h->lz_extra : h->tun.lz_extra, M_MAX - target));
    const int *stop = &h->lz_state->done;
    const double seq = (double)++h->lz_seq;
    h->lz_last_queued = true;
    int rg[3][2];
    const int nrg_all = update_ranges(h, N, rg);
    auto vec = [&](int q) -> const double4 * { return q == 0 ? h->psi_s : h->V + (size_t)q * stride; };
    // the vector part of iteration j: x_{j+1} from the sums that are still in place
    auto vector_part = [&](int j, const int *gate) {
        launch_lz_update(vec(j), h->w_s, j > 0 ? vec(j - 1) : nullptr, h->V + (size_t)(j + 1) * stride, j, h->scal, rg, nrg_all, h->stream,
                         nullptr, 0, h->sc_host_dev, gate, h->vq);   // (+ the mirror of x_{j+1}: the next mat-vec's gathers)
    };
    auto iteration = [&](int j, bool scalars_only, const int *gate) -> int {
        const bool have_y = j == 0 && h->w_is_mpsi;
        const bool fused = !have_y && h->nb.cap > 0 && h->nb_valid;
        if (!fused && !have_y) {
            if (gate) return fail(PSE_ERR_NUMERIC, "queued Lanczos: a gated iteration needs the pair list");
            TRY(real(T, j == 0 ? &pse_handle::psi_s : &pse_handle::V, &pse_handle::w_s, (size_t)j * stride, 0, N, true));
        }
        const double4 *vjm1 = j > 0 ? vec(j - 1) : nullptr;
        if (fused)
            launch_mreal_lanczos(h->pos_s, vec(j), h->w_s, row_map(0, N), h->cell_off, h->dbox, h->nc, h->d.rcut, h->d.self, h->coef, h->nb,
                                 LzFuse{vjm1, h->partials, h->npart_cap, nullptr, nullptr, nullptr}, h->scal, nullptr, nullptr, h->stream,
                                 h->vl_use ? h->vl : VerletList{}, 1, gate, DevRowArgs{}, j > 0 ? h->vq : nullptr);
        else if (!(j == 0 && h->sums0_done))
            launch_lz_dots(vec(j), h->w_s, vjm1, 0, N, h->partials, h->npart_cap, h->scal, h->stream);
        h->w_is_mpsi = false; h->sums0_done = false;
        if (scalars_only) launch_lz_update(vec(j), h->w_s, vjm1, h->V + (size_t)(j + 1) * stride, j, h->scal, rg, 0, h->stream, nullptr, 0,
                                           h->sc_host_dev, gate);
        else vector_part(j, gate);
        return 0;
    };
    for (int j = 0; j < target; ++j) TRY(iteration(j, j == target - 1, nullptr));
    LzDecide d{};
    d.m_lo = std::max(m_in - 1, 1); d.m_hi = target; d.done_iters = target; d.first = 1; d.last = extra == 0 ? 1 : 0; d.m_max = M_MAX; d.tol = tol;
    launch_lz_decide(d, h->scal, h->lz_state, h->sc_host_dev, seq, h->stream);
    int done = target;
    for (int x = 0; x < extra; ++x) {
        vector_part(done - 1, stop);
        TRY(iteration(done, true, stop));
        d.m_lo = d.m_hi = done + 1; d.done_iters = done + 1; d.pending_beta = done; d.first = 0; d.last = x == extra - 1 ? 1 : 0;
        launch_lz_decide(d, h->scal, h->lz_state, h->sc_host_dev, seq, h->stream);
        ++done;
    }
    TRY(te(h, PH_LANCZOS));
    if (before_first_wait) TRY(before_first_wait());
    if (before_combine) TRY(before_combine());
    launch_basis_combine(h->psi_s, h->V, stride, BasisCoef{}, 0, h->scal, scale, 1, h->ub_s, 0, N, h->stream, h->sink, h->lz_state);
    h->tail_done = h->sink.on;
    h->info.lanczos_matvecs = done; h->info.lanczos_exchanges = 0;
    // m of this call is known on the device only; what the host hands back is the most recent one that has reached it
    if (m_io) *m_io = h->sc_host[LZ_HOST_SEQ] > 0.0 && h->sc_host[LZ_HOST_M] >= 1.0 ? (int)h->sc_host[LZ_HOST_M] : m_in;
    HIPCHK(hipGetLastError());
    return 0;
}

// M_real^{1/2} psi by Lanczos (PSEv1/Brownian.cu:357-765): psi_s (sorted order, replicated on every rank) ->
// ub_s = scale |psi| V t on the rows this rank owns.  Scalars are replicated; vectors are valid on the own rows (+ the
// neighbouring cell layers for the vector the next mat-vec reads).
static int lanczos(pse_team &T, int N, double tol, double scale, int *m_io, const std::function<int()> &before_first_wait = nullptr,
                   WavePump *pump = nullptr, const std::function<int()> &before_combine = nullptr) {
    int n_exchanges = 0;
    pse_handle *h0 = T.m[0];
    if (h0->async_mode && T.G == 1 && T.solo < 0 && h0->nb.cap > 0 && !h0->timing &&
        lz_decide_supported(std::min(M_MAX, std::max(std::min(std::max(m_io ? *m_io : 2, 1), M_MAX), 2) + h0->tun.lz_extra)))
        return lanczos_queued(T, N, tol, scale, m_io, before_first_wait, before_combine);
    const size_t stride = h0->n_pad;
    for (pse_handle *h : act(T)) h->lz_last_queued = false;
    int m_in = m_io ? *m_io : 2;
    if (m_in < 1) m_in = 1;
    if (m_in > M_MAX) m_in = M_MAX;
    std::vector<double> t_prev, t_cur;
    double *sc = h0->sc_host;   // pinned: the read-back of the device scalars is one DMA, no staging copy
    int done = 0;                         // iterations launched so far
    int target = std::max(m_in, 2);       // first convergence check is at m = max(m_in, 2)   (Brownian.cu:465-466,606)
    int m_final = 0, checked = 0, pending_beta = 0;
    double stepnorm = 1.0;
    bool hook_done = false;
    while (true) {
        for (; done < target; ++done) {
            // iteration j = done on the unnormalised x_j (psi for j = 0, else parked in V[j]); see k_lz_update
            const bool have_y = done == 0 && h0->w_is_mpsi;      // M psi came with the pass that built the pair list
            const bool fused = !have_y && h0->nb.cap > 0 && h0->nb_valid;   // sums fused into the pair-list mat-vec
            const bool timed = done == 1 && fused;               // one pair-list mat-vec kernel per call is timed on its own
            if (!fused && !have_y) TRY(real(T, done == 0 ? &pse_handle::psi_s : &pse_handle::V, &pse_handle::w_s, (size_t)done * stride, 0, N, true));
            for (pse_handle *h : act(T)) {
                int lo, hi;
                row_range(h, N, lo, hi);
                const double4 *xj = done == 0 ? h->psi_s : h->V + (size_t)done * stride;
                const double4 *vjm1 = done > 1 ? h->V + (size_t)(done - 1) * stride : (done == 1 ? h->psi_s : nullptr);   // x_{j-1}, unnormalised
                if (fused) {
                    const bool ev = timed && h->timing;
                    launch_mreal_lanczos(h->pos_s, xj, h->w_s, row_map(lo, hi), h->cell_off, h->dbox, h->nc, h->d.rcut, h->d.self, h->coef,
                                         h->nb, LzFuse{vjm1, h->partials, h->npart_cap, nullptr, nullptr, nullptr}, h->scal,
                                         ev ? h->ph[PH_MATVEC].a : nullptr, ev ? h->ph[PH_MATVEC].b : nullptr, h->stream,
                                         h->vl_use ? h->vl : VerletList{}, 1, nullptr, DevRowArgs{}, done > 0 && T.G <= 1 ? h->vq : nullptr);
                    if (timed) h->matvec_timed = true;
                } else if (!(done == 0 && h->sums0_done)) {   // (iteration 0: the sums came with the pass that built the pair list)
                    launch_lz_dots(xj, h->w_s, vjm1, lo, hi, h->partials, h->npart_cap, h->scal, h->stream);
                }
                h->w_is_mpsi = false; h->sums0_done = false;
            }
            // the partial sums and the ghost rows of y = M x_j in one exchange; every rank then updates its own rows AND its ghost
            // rows (it holds x_j and x_{j-1} there from the previous iteration), so x_{j+1} needs no exchange of its own
            if (pump) TRY(pump->upto(n_exchanges));   // the far-field exchanges whose slot has come go first
            ++n_exchanges;
            TRY(team_lanczos_exchange(T, [](pse_handle *h) { return (double *)h->w_s; }));
            // The LAST iteration of a batch only derives its scalars (alpha_j, beta_j): whether x_{j+1} is needed at all is what the
            // check below decides -- in the steady state of a time-stepping loop (m_in = m) it is not, and the step saves one vector
            // pass and the reduction of |x_{j+1}| (0.04 ms at the metric point).  If the iteration goes on, the vector part follows.
            const bool scalars_only = done == target - 1;
            for (pse_handle *h : act(T)) {
                int rg[3][2];
                const int nrg = scalars_only ? 0 : update_ranges(h, N, rg);
                const double4 *xj = done == 0 ? h->psi_s : h->V + (size_t)done * stride;
                launch_lz_update(xj, h->w_s, done > 1 ? h->V + (size_t)(done - 1) * stride : (done == 1 ? h->psi_s : nullptr),
                                 h->V + (size_t)(done + 1) * stride, done, h->scal, rg, nrg, h->stream, h->sums_all, T.G > 1 ? T.G : 0, h == h0 ? h->sc_host_dev : nullptr,
                                 nullptr, T.G <= 1 ? h->vq : nullptr);
            }
        }
        // (alpha, beta and the norm are already on their way: the update kernels write them to the mapped host buffer as well)
        HIPCHK(hipEventRecord(h0->ev_scal, h0->stream));
        if (!hook_done) {   // independent work queued behind the read-back keeps the GPU busy while the host decides
            hook_done = true;
            for (pse_handle *h : act(T)) TRY(te(h, PH_LANCZOS));
            if (before_first_wait) TRY(before_first_wait());
        }
        HIPCHK(hipEventSynchronize(h0->ev_scal));
        if (T.solo >= 0) { m_final = done; t_cur.assign(m_final, 0.0); break; }   // timing only (pse_team_debug_solo)
        const double *alpha = &sc[LZ_ALPHA], *beta = &sc[LZ_BETA];   // alpha_0 .. alpha_{done-1}, beta_1 .. beta_{done-1}; beta_done not yet
        if (!(sc[LZ_NORM] > 0.0) || !std::isfinite(sc[LZ_NORM])) {   // psi == 0 -> result 0
            for (pse_handle *h : act(T)) {
                HIPCHK(hipMemsetAsync(h->ub_s, 0, (size_t)N * sizeof(double4), h->stream));
                h->info.lanczos_m = 0; h->info.lanczos_matvecs = done; h->info.lanczos_stepnorm = 0.0;
            }
            if (m_io) *m_io = m_in;
            return 0;
        }
        // |x_m| of the vector the previous batch ended on arrives with this batch's first mat-vec: its breakdown test comes first
        if (pending_beta && beta[pending_beta] < 1e-8) { m_final = pending_beta; stepnorm = 0.0; t_cur = t_prev; break; }
        pending_beta = 0;
        // walk m upward exactly as the reference's while loop does, one vector at a time
        for (int m = std::max(checked + 1, std::max(m_in - 1, 1)); m <= done && !m_final; ++m) {
            const bool last = m == done;                                          // beta_done = |x_done| does not exist yet
            if (!std::isfinite(alpha[m - 1]) || (!last && !std::isfinite(beta[m])))
                return fail(PSE_ERR_NUMERIC, "Lanczos produced a non-finite coefficient at iteration %d", m - 1);
            if (!lanczos_sqrt_e1(m, alpha, beta, t_cur))
                return fail(PSE_ERR_NUMERIC, "tridiagonal eigen-solve failed at m = %d", m);
            if (!last && beta[m] < 1e-8) { m_final = m; stepnorm = 0.0; break; }  // invariant subspace (Brownian.cu:503)
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
        if (done >= M_MAX) { m_final = M_MAX; break; }
        // not converged: the vector part of the last iteration (its sums are still in place), then the next batch; an x_done that
        // vanishes (invariant subspace) shows in the first sums of that batch
        for (pse_handle *h : act(T)) {
            int rg[3][2];
            const int nrg = update_ranges(h, N, rg);
            const int j = done - 1;
            const double4 *xj = j == 0 ? h->psi_s : h->V + (size_t)j * stride;
            launch_lz_update(xj, h->w_s, j > 1 ? h->V + (size_t)(j - 1) * stride : (j == 1 ? h->psi_s : nullptr),
                             h->V + (size_t)(j + 1) * stride, j, h->scal, rg, nrg, h->stream, h->sums_all, T.G > 1 ? T.G : 0, h == h0 ? h->sc_host_dev : nullptr,
                             nullptr, T.G <= 1 ? h->vq : nullptr);
        }
        pending_beta = done;
        target = std::min(M_MAX, done + std::max(2, done / 4));
    }
    if ((int)t_cur.size() != m_final) {
        if (!lanczos_sqrt_e1(m_final, &sc[LZ_ALPHA], &sc[LZ_BETA], t_cur))
            return fail(PSE_ERR_NUMERIC, "tridiagonal eigen-solve failed at m = %d", m_final);
    }
    if (before_combine) TRY(before_combine());   // (the far-field velocities must be in place if the combination adds them)
    for (pse_handle *h : act(T)) {
        BasisCoef tc{};   // the basis holds the unnormalised x_q: v_q = x_q / |x_q|, |x_0| = the norm of psi, |x_q| = beta_q
        for (int q = 0; q < m_final; ++q) tc.t[q] = t_cur[q] / (q == 0 ? sc[LZ_NORM] : sc[LZ_BETA + q]);
        int lo, hi;
        row_range(h, N, lo, hi);
        launch_basis_combine(h->psi_s, h->V, stride, tc, m_final, h->scal, scale, 1, h->ub_s, lo, hi, h->stream, h->sink);   // Brownian.cu:716,739
        h->tail_done = h->sink.on;
        h->info.lanczos_m = m_final; h->info.lanczos_matvecs = done; h->info.lanczos_stepnorm = stepnorm;
        h->info.lanczos_exchanges = T.G > 1 ? n_exchanges : 0;
    }
    if (m_io) *m_io = m_final;
    return 0;
}

// M_real^{1/2} psi for a team, TWO Lanczos iterations per exchange (k_lz_block in pse_kernels.hip has the algebra).  Per block:
// w1 = M v_j on the own rows + one ghost layer, w2 = M w1 on the own rows with the Gram sums fused in, ONE exchange (all-reduce
// of the sums + two ghost layers of w1 and w2 to either neighbour), then every rank forms v_{j+1}, v_{j+2} and M v_{j+1} on its own
// and its ghost rows.  An odd tail is a single step (scalars only unless the iteration goes on).  The alpha, beta the host checks
// are the reference's (PSEv1/Brownian.cu:440-521) to rounding, so m is -- the tests hold it to the port's.  The first w1 = M psi
// comes with the pass that built the pair list.  Exchanges per step at m = 7: 4 (one-step driver: 7).
static int lanczos_team(pse_team &T, int N, double tol, double scale, int *m_io, const std::function<int()> &before_first_wait,
                        WavePump *pump, const std::function<int()> &before_combine = nullptr) {
    pse_handle *h0 = T.m[0];
    const size_t stride = h0->n_pad;
    for (pse_handle *h : act(T)) h->lz_last_queued = false;
    int m_in = m_io ? *m_io : 2;
    m_in = std::min(std::max(m_in, 1), M_MAX);
    std::vector<double> t_prev, t_cur;
    double *sc = h0->sc_host;
    int done = 0, target = std::max(m_in, 2), m_final = 0, checked = 0, n_exchanges = 0, matvecs = 0;
    int half_pending = -1;                // j of a single step whose vectors have not been formed yet
    double stepnorm = 1.0;
    bool hook_done = false;
    auto vec = [&](pse_handle *h, int q) { return q == 0 ? h->psi_s : h->V + (size_t)q * stride; };
    auto block_args = [&](pse_handle *h, int j, bool full) {
        LzBlockArgs a{};
        a.q = vec(h, j); a.p = j > 0 ? vec(h, j - 1) : nullptr; a.w1 = h->w_s; a.w2 = full ? h->w2_s : nullptr;
        a.u = h->u_s; a.v1 = h->V + (size_t)(j + 1) * stride; a.v2 = full ? h->V + (size_t)(j + 2) * stride : nullptr;
        a.j = j;
        return a;
    };
    while (true) {
        while (done < target) {
            if (half_pending >= 0) {      // the iteration goes on past a single step: its vector part (w1 is there on two ghost layers)
                for (pse_handle *h : act(T)) {
                    int rg[3][2];
                    const int nrg = row_ranges(h, N, 2, rg);
                    launch_lz_block(block_args(h, half_pending, false), false, h->scal, rg, nrg, h->stream, h->sums_all, T.G, h == h0 ? h->sc_host_dev : nullptr);
                }
                half_pending = -1;
            }
            const int j = done;
            const bool full = target - done >= 2 && j + 2 <= M_MAX;
            for (pse_handle *h : act(T)) {
                const LzFuse lz{nullptr, h->partials, h->npart_cap, vec(h, j), j > 0 ? vec(h, j - 1) : nullptr, j > 0 ? h->u_s : nullptr};
                if (j == 0 && !h->w_is_mpsi) return fail(PSE_ERR_NUMERIC, "two-step Lanczos: M psi did not come with the pair list");
                if (!(j == 0 && h->w_is_mpsi)) {   // w1 = M v_j: own rows + one ghost layer for a block, own rows for a single step
                    launch_mreal_lanczos(h->pos_s, vec(h, j), h->w_s, rank_rows(h, N, full ? 1 : 0), h->cell_off, h->dbox, h->nc, h->d.rcut,
                                         h->d.self, h->coef, h->nb, lz, h->scal, nullptr, nullptr, h->stream, VerletList{}, full ? 0 : 3);
                    ++matvecs;
                } else if (!full) {
                    return fail(PSE_ERR_NUMERIC, "two-step Lanczos: a single step cannot start at j = 0");
                }
                h->w_is_mpsi = false;
                if (full) {                        // w2 = M w1 on the own rows, Gram sums fused
                    launch_mreal_lanczos(h->pos_s, h->w_s, h->w2_s, rank_rows(h, N, 0), h->cell_off, h->dbox, h->nc, h->d.rcut, h->d.self,
                                         h->coef, h->nb, lz, h->scal, nullptr, nullptr, h->stream, VerletList{}, 2);
                    ++matvecs;
                }
            }
            if (pump) TRY(pump->upto(n_exchanges));
            ++n_exchanges;
            TRY(team_run_exchange(T, [&](pse_handle *h) {
                                      auto ops = ghost_ops(h, T.G, {(double *)h->w_s, full ? (double *)h->w2_s : nullptr}, 2);
                                      sum_ops(h, T.G, LZ_NGRAM, ops);
                                      return ops; },
                                  false, DIAG_LANCZOS));
            for (pse_handle *h : act(T)) {
                int rg[3][2];
                const int nrg = full ? row_ranges(h, N, 2, rg) : 0;   // a single step derives its scalars only, for now
                launch_lz_block(block_args(h, j, full), full, h->scal, rg, nrg, h->stream, h->sums_all, T.G, h == h0 ? h->sc_host_dev : nullptr);
            }
            if (full) done += 2; else { done += 1; half_pending = j; }
        }
        // (alpha, beta and the norm are already on their way: the update kernels write them to the mapped host buffer as well)
        HIPCHK(hipEventRecord(h0->ev_scal, h0->stream));
        if (!hook_done) {   // independent work queued behind the read-back keeps the GPU busy while the host decides
            hook_done = true;
            for (pse_handle *h : act(T)) TRY(te(h, PH_LANCZOS));
            if (before_first_wait) TRY(before_first_wait());
        }
        HIPCHK(hipEventSynchronize(h0->ev_scal));
        if (T.solo >= 0) { m_final = done; t_cur.assign(m_final, 0.0); break; }   // timing only (pse_team_debug_solo)
        const double *alpha = &sc[LZ_ALPHA], *beta = &sc[LZ_BETA];   // alpha_0 .. alpha_{done-1}, beta_1 .. beta_done
        if (!(sc[LZ_NORM] > 0.0) || !std::isfinite(sc[LZ_NORM])) {   // psi == 0 -> result 0
            for (pse_handle *h : act(T)) {
                HIPCHK(hipMemsetAsync(h->ub_s, 0, (size_t)N * sizeof(double4), h->stream));
                h->info.lanczos_m = 0; h->info.lanczos_matvecs = matvecs; h->info.lanczos_stepnorm = 0.0;
            }
            if (m_io) *m_io = m_in;
            return 0;
        }
        // walk m upward exactly as the reference's while loop does, one vector at a time
        for (int m = std::max(checked + 1, std::max(m_in - 1, 1)); m <= done && !m_final; ++m) {
            if (!std::isfinite(alpha[m - 1]) || !std::isfinite(beta[m]))
                return fail(PSE_ERR_NUMERIC, "Lanczos produced a non-finite coefficient at iteration %d", m - 1);
            if (!lanczos_sqrt_e1(m, alpha, beta, t_cur))
                return fail(PSE_ERR_NUMERIC, "tridiagonal eigen-solve failed at m = %d", m);
            if (m < done && beta[m] < 1e-8) { m_final = m; stepnorm = 0.0; break; }  // invariant subspace (Brownian.cu:503)
            if (!t_prev.empty() && (int)t_prev.size() == m - 1) {
                double s2 = t_cur[m - 1] * t_cur[m - 1];
                for (int q = 0; q < m - 1; ++q) s2 += (t_cur[q] - t_prev[q]) * (t_cur[q] - t_prev[q]);
                stepnorm = std::sqrt(s2 / alpha[0]);                               // Brownian.cu:719-724; psi.M.psi/|psi|^2 = alpha_0
                if (stepnorm <= tol || m >= M_MAX) { m_final = m; break; }
            }
            if (m == done && beta[m] < 1e-8) { m_final = m; stepnorm = 0.0; break; }   // the one-step driver sees this with its next sums
            t_prev = t_cur;
            checked = m;
        }
        if (m_final) break;
        if (done >= M_MAX) { m_final = M_MAX; break; }
        target = std::min(M_MAX, done + std::max(2, done / 4));
    }
    if ((int)t_cur.size() != m_final) {
        if (!lanczos_sqrt_e1(m_final, &sc[LZ_ALPHA], &sc[LZ_BETA], t_cur))
            return fail(PSE_ERR_NUMERIC, "tridiagonal eigen-solve failed at m = %d", m_final);
    }
    if (half_pending >= 0 && m_final > half_pending + 1) return fail(PSE_ERR_NUMERIC, "two-step Lanczos: basis vector %d was never formed", half_pending + 1);
    if (before_combine) TRY(before_combine());   // (the far-field velocities must be in place if the combination adds them)
    for (pse_handle *h : act(T)) {
        BasisCoef tc{};   // the basis holds NORMALISED v_q here, except v_0 = psi / |psi|
        for (int q = 0; q < m_final; ++q) tc.t[q] = q == 0 ? t_cur[q] / sc[LZ_NORM] : t_cur[q];
        int lo, hi;
        row_range(h, N, lo, hi);
        launch_basis_combine(h->psi_s, h->V, stride, tc, m_final, h->scal, scale, 1, h->ub_s, lo, hi, h->stream, h->sink);   // Brownian.cu:716,739
        h->tail_done = h->sink.on;
        h->info.lanczos_m = m_final; h->info.lanczos_matvecs = matvecs; h->info.lanczos_stepnorm = stepnorm;
        h->info.lanczos_exchanges = n_exchanges;
    }
    if (m_io) *m_io = m_final;
    return 0;
}

// per-member argument pointers of a team call
struct Args {
    const double4 *pos; const double4 *force; double4 *vel;
};

// what pse_step does after the velocity is known (K15 gpu_stokes_step_one_kernel, PSEv1/Stokes.cu:137-192): handed to velocity()
// so that the final un-sort and the Euler update are one pass over the particles
struct StepArgs {
    double4 *pos; double4 *vel; double3 *accel; int3 *image; const double4 *force;
};
struct StepTail { const std::vector<StepArgs> *sa; double dt, shear_rate; };

static int velocity(pse_team &T, const std::vector<Args> &a, const unsigned *group, int N, int parts, double kT, double dt,
                    unsigned timestep, int *m_io, unsigned *mask, const StepTail *tail = nullptr) {
    for (pse_handle *h : T.m)
        if (h->loc.on) return fail(PSE_ERR_INVALID, "this handle is an owned-particle rank (pse_params.local_rows): drive it through pse_team_step_local");
    diag_begin(T);
    for (size_t r = 0; r < T.m.size(); ++r)
        if (T.solo < 0 || T.m[r]->slab_rank == T.solo)   // psi rides with the gather into cell order: the near-field pass that
            TRY(prepare(T.m[r], a[r].pos, a[r].force, group, N, true, false, PrepExtra{kT > 0.0, timestep}, (parts & 1) != 0));   // builds the pair list applies M_real to F and psi together
    *mask |= 1u << PH_SORT;
    const bool noise = kT > 0.0;
    const bool sstep = T.G > 1 && noise && (parts & 1) && team_sstep(T.m[0]);   // two Lanczos iterations per exchange
    for (pse_handle *h : act(T)) {   // where the wave chain of this call runs
        // The two chains share the chip whenever nothing is timed per kernel (with phase timing on, every kernel runs alone on one
        // stream: those durations are the roofline evidence).  A single GPU forks for deterministic evaluations only (Brownian
        // steps gain nothing there); a TEAM always forks: at 1 / G of the rows no kernel fills the chip, and the
        // far-field exchanges travel while the Lanczos iterations compute (one communicator, one communication stream: pse_team).
        const bool on = h->side && parts == 3 && !h->timing && (T.G > 1 ? T.lanes : (h->overlap_all || !noise));
        if (on != h->side_on || h->wstream != (on ? h->side : h->stream)) {
            h->side_on = on;
            h->wstream = on ? h->side : h->stream;
            FFTCHK(rocfft_execution_info_set_stream(h->info_fwd, h->wstream));
            FFTCHK(rocfft_execution_info_set_stream(h->info_inv, h->wstream));
        }
    }
    WavePump pump;
    const WaveArgs wa{N, noise, kT, dt, timestep};
    auto wave_start = [&]() -> int { return pump.start(T, wa, T.m[0]->tun.team_sched, mask); };
    // With noise on one stream the wave chain is queued BEHIND the Lanczos iterations: the host has to read their scalars
    // back before it can finish the Brownian part, and meanwhile the GPU works through the far field instead of idling.
    const bool wave_behind = noise && (parts & 2) && (parts & 1) && !T.m[0]->side_on;
    // Two lanes and noise: the near field + Lanczos chain is the longer lane, so its first pass is queued BEFORE the far-field chain
    // (which forks here and follows at once); a deterministic evaluation has the far field as its longer lane and keeps it first.
    const bool near_first = noise && parts == 3 && T.m[0]->side_on;
    if (near_first) TRY(pump.fork(T));
    if ((parts & 2) && !wave_behind && !near_first) TRY(wave_start());
    // the slab row boundaries are needed from here on (a host round trip); the first part of the far-field chain is queued
    for (pse_handle *h : act(T)) TRY(slab_bounds_wait(h, N));
    if (parts & 1) {
        for (pse_handle *h : act(T)) TRY(ts(h, PH_REAL));
        TRY(real(T, &pse_handle::f_s, &pse_handle::ur_s, 0, 0, N, noise, noise, sstep ? 1 : 0));
        for (pse_handle *h : act(T)) TRY(te(h, PH_REAL));
        *mask |= 1u << PH_REAL;
    }
    if (near_first) TRY(wave_start());
    const bool lz = noise && (parts & 1);   // the particle noise M_real^{1/2} psi belongs to the real-space half (the k-space noise to the wave half)
    if (lz) {
        for (pse_handle *h : act(T)) TRY(ts(h, PH_LANCZOS));   // closed inside the Lanczos driver, after the first batch of iterations
        for (pse_handle *h : act(T)) h->matvec_timed = false;
        const std::function<int()> hook = [&]() -> int {    // before the host first waits for the Lanczos scalars: everything else is queued
            if (wave_behind) TRY(wave_start());
            return pump.drain();
        };
        const double tol = T.m[0]->d.error, scale = std::sqrt(2.0 * kT / dt);
        // The final combination of the Lanczos vectors adds the far-field and the near-field velocity of its row and sends the sum
        // where the step wants it: one pass and one launch less than combination -> ub_s -> sum / un-sort.
        for (size_t r = 0; r < T.m.size(); ++r) {
            pse_handle *h = T.m[r];
            h->tail_done = false;
            h->sink = CombineSink{};
            if (parts == 3) h->sink = CombineSink{true, h->uw_s, h->ur_s, h->tag_s, T.G == 1 ? a[r].vel : nullptr, T.G == 1 ? nullptr : h->utot_s};
        }
        const std::function<int()> join = [&]() -> int {   // the gathered far-field velocity is needed by the combination
            TRY(pump.drain());
            for (pse_handle *h : act(T))
                if (h->side_on && h->sink.on) {
                    if (h == act(T)[0]) diag_mark(T, 3, h->side);
                    HIPCHK(hipEventRecord(h->ev_join, h->side));
                    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
                }
            return 0;
        };
        if (sstep) TRY(lanczos_team(T, N, tol, scale, m_io, hook, &pump, join));
        else TRY(lanczos(T, N, tol, scale, m_io, hook, &pump, join));
        for (pse_handle *h : act(T)) h->sink = CombineSink{};
        *mask |= 1u << PH_LANCZOS;
        if (T.m[0]->matvec_timed) *mask |= 1u << PH_MATVEC;
    }
    TRY(pump.drain());   // (kT = 0: the exchanges of the far-field chain are issued here, one after the other)
    if (T.G > 1) {
        // every rank has all three contributions for the rows it owns: add them, exchange the row blocks once
        for (pse_handle *h : act(T))
            if ((parts & 2) && h->side_on) {   // join: the gathered far-field velocity is needed now
                if (h == act(T)[0] && !(noise && h->tail_done)) diag_mark(T, 3, h->side);
                HIPCHK(hipEventRecord(h->ev_join, h->side));
                HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
            }
        for (pse_handle *h : act(T)) {
            if (noise && h->tail_done) continue;   // the Lanczos combination has written the summed rows already
            int lo, hi;
            row_range(h, N, lo, hi);
            launch_sum_rows((parts & 2) ? h->uw_s : nullptr, (parts & 1) ? h->ur_s : nullptr, lz ? h->ub_s : nullptr,
                            h->utot_s, lo, hi, h->stream, h->tag_s);   // the tags travel with the rows: no rank orders foreign rows
        }
        TRY(team_all_gather_rows(T, [](pse_handle *h) { return (double *)h->utot_s; }));
    }
    for (size_t r = 0; r < T.m.size(); ++r) {
        pse_handle *h = T.m[r];
        if (T.solo >= 0 && h->slab_rank != T.solo) continue;
        if ((parts & 2) && h->side_on && T.G == 1) {   // join
            HIPCHK(hipEventRecord(h->ev_join, h->side));
            HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
        }
        const double4 *ua = T.G > 1 ? h->utot_s : ((parts & 2) ? h->uw_s : nullptr);
        const double4 *ub = T.G > 1 ? nullptr : ((parts & 1) ? h->ur_s : nullptr), *uc = T.G > 1 ? nullptr : (lz ? h->ub_s : nullptr);
        const unsigned *tags = T.G > 1 ? nullptr : h->tag_s;     // a team: the tag travels in the fourth component of its row
        if (!(T.G == 1 && noise && h->tail_done)) launch_scatter_sum(ua, ub, uc, tags, N, a[r].vel, h->stream);   // (single GPU: the Lanczos combination has un-sorted the sum)
        if (tail) {
            // The Euler update stays its own pass in the CALLER's order (replicated state: every rank updates every particle,
            // bit-identically).  Fused into the un-sort it ran over the sorted rows and made five arrays scattered instead of
            // one: 152 us against 29 + 35 per rank at the metric point (profiles/r04_team8_notes.txt).
            const StepArgs &x = (*tail->sa)[r];
            TRY(ts(h, PH_INTEG));
            launch_integrate(x.pos, x.vel, x.accel, x.image, x.force, group, N, h->dbox, tail->dt, tail->shear_rate, h->stream);
            TRY(te(h, PH_INTEG));
        }
        HIPCHK(hipGetLastError());
    }
    diag_mark(T, 1, act(T)[0]->stream);
    return 0;
}

static int team_of_one(pse_handle *h, pse_team &T) {
    if (h->n_slabs > 1) return fail(PSE_ERR_INVALID, "this handle is a slab rank: drive it through a pse_team");
    T.m = {h}; T.G = 1;
    return 0;
}

static int do_mobility(pse_team &T, const std::vector<Args> &a, const unsigned *group, unsigned N, int parts) {
    for (pse_handle *h : T.m) TRY(check_n(h, N));
    for (auto &x : a) if (!x.pos || !x.force || !x.vel) return fail(PSE_ERR_INVALID, "null array");
    if (!(parts & 3)) return fail(PSE_ERR_INVALID, "parts must select real (1), wave (2) or both (3)");
    unsigned mask = 1u << PH_TOTAL;
    for (pse_handle *h : act(T)) TRY(ts(h, PH_TOTAL));
    TRY(velocity(T, a, group, (int)N, parts, 0.0, 1.0, 0, nullptr, &mask));
    for (pse_handle *h : act(T)) { TRY(te(h, PH_TOTAL)); TRY(collect_times(h, mask)); }
    return 0;
}

static int do_brownian(pse_team &T, const std::vector<Args> &a, const unsigned *group, unsigned N, double kT, double dt,
                       unsigned timestep, int *lanczos_m, int parts = 3) {
    for (pse_handle *h : act(T)) TRY(check_n(h, N));
    for (auto &x : a) if (!x.pos || !x.force || !x.vel) return fail(PSE_ERR_INVALID, "null array");
    if (kT < 0 || !(dt > 0)) return fail(PSE_ERR_INVALID, "need kT >= 0 and dt > 0");
    unsigned mask = 1u << PH_TOTAL;
    for (pse_handle *h : act(T)) TRY(ts(h, PH_TOTAL));
    TRY(velocity(T, a, group, (int)N, parts, kT, dt, timestep, lanczos_m, &mask));
    for (pse_handle *h : act(T)) { TRY(te(h, PH_TOTAL)); TRY(collect_times(h, mask)); }
    return 0;
}

static int do_step(pse_team &T, const std::vector<StepArgs> &sa, const unsigned *group, unsigned N, double kT, double dt,
                   unsigned timestep, double shear_rate, int *lanczos_m) {
    for (pse_handle *h : act(T)) TRY(check_n(h, N));
    std::vector<Args> a;
    for (auto &x : sa) {
        if (!x.pos || !x.vel || !x.accel || !x.image || !x.force) return fail(PSE_ERR_INVALID, "null array");
        a.push_back(Args{x.pos, x.force, x.vel});
    }
    if (kT < 0 || !(dt > 0)) return fail(PSE_ERR_INVALID, "need kT >= 0 and dt > 0");
    unsigned mask = (1u << PH_TOTAL) | (1u << PH_INTEG);
    for (pse_handle *h : act(T)) TRY(ts(h, PH_TOTAL));
    for (pse_handle *h : act(T)) h->vl_kind = 1;   // an integrating step: its own neighbour-list suspension state
    const StepTail tail{&sa, dt, shear_rate};
    const int rc_v = velocity(T, a, group, (int)N, 3, kT, dt, timestep, lanczos_m, &mask, &tail);
    for (pse_handle *h : act(T)) h->vl_kind = 0;
    if (rc_v) return rc_v;
    for (pse_handle *h : act(T)) {
        TRY(te(h, PH_TOTAL));
        HIPCHK(hipGetLastError());
        TRY(collect_times(h, mask));
    }
    return 0;
}

// ---- owned-particle team step (pse_team_step_local; kernels in pse_local.hip) ---------------------------------------------------
// Everything below only QUEUES work: the row counts of a step never reach the host (LocalRows lives in device memory, every
// launch covers the capacity), every exchange has the fixed size of its capacity, the Lanczos decision is k_lz_decide's.
static LocalRegions local_regions(const pse_handle *h) {
    const LocalGeom &g = h->loc.g;
    const int lc = layer_cells(h->nc), nx = g.nx;
    auto wrap = [&](int l) { return ((l % nx) + nx) % nx; };
    LocalRegions r{};
    const int l_own = g.rank * g.per, l_gl = wrap(l_own - g.depth), l_gr = wrap(l_own + g.per);
    r.c0[0] = l_own * lc; r.c1[0] = (l_own + g.per) * lc; r.base[0] = 0; r.cap[0] = g.c_own;
    r.c0[1] = l_gl * lc; r.c1[1] = (l_gl + g.depth) * lc; r.base[1] = g.c_own; r.cap[1] = g.c_g;
    r.c0[2] = l_gr * lc; r.c1[2] = (l_gr + g.depth) * lc; r.base[2] = g.c_own + g.c_g; r.cap[2] = g.c_g;
    r.c_first_end = (l_own + g.depth) * lc - 1;            // the empty last cell of layer l_own + depth - 1
    r.c_last_begin = (l_own + g.per - g.depth) * lc;
    r.c_gl_adj = wrap(l_own - 1) * lc;
    r.c_gr_adj = (l_gr + 1) * lc - 1;                      // the empty last cell of layer l_gr
    return r;
}

// M_real^{1/2} psi of an owned-particle team: the two-step blocks of lanczos_team (k_lz_block has the algebra), queue-only.  Blocks for
// the starting count, ONE decision on the device (sizes m_in - 1 and m_in), then ceil(PSE_LANCZOS_EXTRA / 2) further blocks, each
// with its decision, their kernels gated on the outcome so far (the exchanges of a gated block still travel: they carry no
// meaning then, and every rank takes the same decisions from the same sums).  Ghost rows move by position: the first c_g rows of
// w1, w2 go left as they lie, the rows of the last layers were parked by the mat-vecs (stage_w1, stage_w2: their first row is known
// on the device only) and go right; they arrive in the ghost regions of the neighbours' w1, w2.
static int lanczos_local(pse_team &T, double tol, int *m_io, WavePump *pump) {
    pse_handle *h0 = T.m[0];
    const size_t stride = h0->n_pad;
    const int m_in = std::min(std::max(m_io ? *m_io : 2, 1), M_MAX), target = std::max(m_in, 2);
    const int extra_blocks = ((T.lz_extra >= 0 ? T.lz_extra : h0->tun.lz_extra) + 1) / 2;
    if (target + 2 * extra_blocks > M_MAX || !lz_decide_supported(target + 2 * extra_blocks))
        return fail(PSE_ERR_INVALID, "owned-particle step: starting count %d of the Lanczos iteration is beyond what the device-side decision takes", m_in);
    int n_exchanges = 0, matvecs = 0, done = 0, half_pending = -1;
    auto vec = [&](pse_handle *h, int q) { return q == 0 ? h->psi_s : h->V + (size_t)q * stride; };
    auto block_args = [&](pse_handle *h, int j, bool full) {
        LzBlockArgs a{};
        a.q = vec(h, j); a.p = j > 0 ? vec(h, j - 1) : nullptr; a.w1 = h->w_s; a.w2 = full ? h->w2_s : nullptr;
        a.u = h->u_s; a.v1 = h->V + (size_t)(j + 1) * stride; a.v2 = full ? h->V + (size_t)(j + 2) * stride : nullptr;
        a.j = j;
        return a;
    };
    auto fuse = [&](pse_handle *h, int j) { return LzFuse{nullptr, h->partials, h->npart_cap, vec(h, j), j > 0 ? vec(h, j - 1) : nullptr, j > 0 ? h->u_s : nullptr}; };
    auto exchange = [&](bool full) -> int {
        if (pump) TRY(pump->upto(n_exchanges));
        ++n_exchanges;
        return team_run_exchange(T, [&](pse_handle *h) {
            const LocalGeom &g = h->loc.g;
            const int L = (g.rank + g.G - 1) % g.G, R = (g.rank + 1) % g.G;
            const size_t cnt = (size_t)g.c_g * 4;
            std::vector<Xfer> ops;
            // my first layers go left and arrive as the left neighbour's right ghosts; my last layers (parked) go right
            ops.push_back(Xfer{(double *)h->w_s, cnt, L, (double *)(h->w_s + g.c_own + g.c_g), cnt, R});
            ops.push_back(Xfer{(double *)h->loc.stage_w1, cnt, R, (double *)(h->w_s + g.c_own), cnt, L});
            if (full) {
                ops.push_back(Xfer{(double *)h->w2_s, cnt, L, (double *)(h->w2_s + g.c_own + g.c_g), cnt, R});
                ops.push_back(Xfer{(double *)h->loc.stage_w2, cnt, R, (double *)(h->w2_s + g.c_own), cnt, L});
            }
            sum_ops(h, T.G, LZ_NGRAM, ops);
            return ops; }, false, DIAG_LANCZOS);
    };
    auto full_block = [&](int j, bool gated) -> int {
        for (pse_handle *h : act(T)) {
            const int *gate = gated ? &h->lz_state->done : nullptr;
            const LocalRows *R = h->loc.rows;
            if (j == 0) {
                if (!h->w_is_mpsi) return fail(PSE_ERR_NUMERIC, "owned-particle step: M psi did not come with the pair list");
            } else {   // w1 = M v_j on the own rows + the adjacent ghost layers
                launch_mreal_lanczos(h->pos_s, vec(h, j), h->w_s, RowMap{}, h->cell_off, h->dbox, h->nc, h->d.rcut, h->d.self, h->coef, h->nb, fuse(h, j),
                                     h->scal, nullptr, nullptr, h->stream, VerletList{}, 0, gate,
                                     DevRowArgs{&R->own1, R, h->loc.stage_w1, h->loc.rows_cap, h->loc.g.c_g});
                ++matvecs;
            }
            h->w_is_mpsi = false;
            // w2 = M w1 on the own rows, Gram sums fused
            launch_mreal_lanczos(h->pos_s, h->w_s, h->w2_s, RowMap{}, h->cell_off, h->dbox, h->nc, h->d.rcut, h->d.self, h->coef, h->nb, fuse(h, j), h->scal,
                                 nullptr, nullptr, h->stream, VerletList{}, 2, gate, DevRowArgs{&R->own, R, h->loc.stage_w2, h->loc.g.c_own, h->loc.g.c_g});
            ++matvecs;
        }
        TRY(exchange(true));
        for (pse_handle *h : act(T))
            launch_lz_block(block_args(h, j, true), true, h->scal, nullptr, 0, h->stream, h->sums_all, T.G, h->sc_host_dev, &h->loc.rows->all,
                            h->loc.rows_cap, false, gated ? &h->lz_state->done : nullptr);
        return 0;
    };
    auto single = [&](int j, bool gated) -> int {   // an odd starting count ends on one iteration: scalars now, vectors only if the iteration goes on
        for (pse_handle *h : act(T)) {
            const LocalRows *R = h->loc.rows;
            launch_mreal_lanczos(h->pos_s, vec(h, j), h->w_s, RowMap{}, h->cell_off, h->dbox, h->nc, h->d.rcut, h->d.self, h->coef, h->nb, fuse(h, j), h->scal,
                                 nullptr, nullptr, h->stream, VerletList{}, 3, gated ? &h->lz_state->done : nullptr,
                                 DevRowArgs{&R->own, R, h->loc.stage_w1, h->loc.g.c_own, h->loc.g.c_g});
            ++matvecs;
        }
        TRY(exchange(false));
        for (pse_handle *h : act(T))
            launch_lz_block(block_args(h, j, false), false, h->scal, nullptr, 0, h->stream, h->sums_all, T.G, h->sc_host_dev, &h->loc.rows->all,
                            h->loc.rows_cap, true, gated ? &h->lz_state->done : nullptr);
        half_pending = j;
        return 0;
    };
    auto decide = [&](int m_lo, int m_hi, int done_iters, bool first, bool last) {
        for (pse_handle *h : act(T)) {
            LzDecide d{};
            d.m_lo = m_lo; d.m_hi = m_hi; d.done_iters = done_iters; d.have_last_beta = 1; d.first = first ? 1 : 0; d.last = last ? 1 : 0;
            // (pse_team_debug_solo: the other ranks' sums are stale, the step norm means nothing -- the decision is made to pass at the
            // starting count, as it does in the steady state of a time-stepping loop, so the gated block is timed as what it then is)
            d.normalised = 1; d.m_max = M_MAX; d.tol = T.solo >= 0 ? 1e300 : tol;
            launch_lz_decide(d, h->scal, h->lz_state, h->sc_host_dev, (double)h->lz_seq, h->stream);
        }
    };
    for (pse_handle *h : act(T)) { ++h->lz_seq; h->lz_last_queued = true; }
    while (target - done >= 2) { TRY(full_block(done, false)); done += 2; }
    if (target - done == 1) { TRY(single(done, false)); done += 1; }
    decide(std::max(m_in - 1, 1), target, done, true, extra_blocks == 0);
    for (int x = 0; x < extra_blocks; ++x) {
        if (half_pending >= 0) {   // the vector part of the single step (its sums are still in place)
            for (pse_handle *h : act(T))
                launch_lz_block(block_args(h, half_pending, false), false, h->scal, nullptr, 0, h->stream, h->sums_all, T.G, h->sc_host_dev,
                                &h->loc.rows->all, h->loc.rows_cap, false, &h->lz_state->done);
            half_pending = -1;
        }
        TRY(full_block(done, true));
        decide(done + 1, done + 2, done + 2, false, x == extra_blocks - 1);
        done += 2;
    }
    for (pse_handle *h : act(T)) { h->info.lanczos_matvecs = matvecs; h->info.lanczos_exchanges = n_exchanges; }
    // What the host hands back is the most recent m that has reached its mirror -- in ONE process only: between processes that
    // moment differs from rank to rank, and ranks that then start the next step from different counts queue different numbers of
    // exchanges (a hang).  A rank of a process team gets its starting count back unchanged; pse_get_info after a synchronisation
    // gives the m of the completed step -- the same on every rank, they all take the same decisions from the same sums.
    if (m_io) *m_io = !remote(T) && h0->sc_host[LZ_HOST_SEQ] > 0.0 && h0->sc_host[LZ_HOST_M] >= 1.0 ? (int)h0->sc_host[LZ_HOST_M] : m_in;
    return 0;
}

static int local_call(pse_team &T, const std::vector<LocalCaller> &ca, double kT, double dt, unsigned timestep, double shear_rate, int integrate,
                      int *m_io) {
    if (T.G < 2) return fail(PSE_ERR_INVALID, "pse_team_step_local needs a team of >= 2 ranks");
    for (size_t r = 0; r < T.m.size(); ++r) {
        pse_handle *h = T.m[r];
        if (!h->loc.on) return fail(PSE_ERR_INVALID, "pse_team_step_local: member %zu was not created with local_rows", r);
        if (h->loc.err_host[0]) return fail(PSE_ERR_INVALID, "owned-particle rank %d: an earlier step failed on the device (flags %d: 1 own rows, 2 ghost rows, "
                                            "4 message capacity exceeded, 8 a particle moved beyond the neighbour, 16 n_local above capacity)", h->slab_rank, h->loc.err_host[0]);
        const LocalCaller &c = ca[r];
        if (!c.pos || !c.vel || !c.accel || !c.image || !c.force || !c.tag || !c.n_local) return fail(PSE_ERR_INVALID, "null array");
        HIPCHK(hipSetDevice(h->device));
    }
    if (kT < 0 || !(dt > 0)) return fail(PSE_ERR_INVALID, "need kT >= 0 and dt > 0");
    const bool noise = kT > 0.0;
    unsigned mask = 0;
    auto stage = [&](const char *what) -> int {   // developer aid (PSE_DEBUG_SYNC, read when the team was created): wait and report after every stage
        if (!T.debug_sync) return 0;
        hipError_t e = hipDeviceSynchronize();
        fprintf(stderr, "pse local_call: %s: %s\n", what, hipGetErrorString(e));
        return e == hipSuccess ? 0 : fail(PSE_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    diag_begin(T);
    for (pse_handle *h : act(T)) {   // two lanes: the far-field chain next to the near field / Lanczos chain
        const bool on = h->side && !h->timing && T.lanes;
        if (on != h->side_on || h->wstream != (on ? h->side : h->stream)) {
            h->side_on = on;
            h->wstream = on ? h->side : h->stream;
            FFTCHK(rocfft_execution_info_set_stream(h->info_fwd, h->wstream));
            FFTCHK(rocfft_execution_info_set_stream(h->info_inv, h->wstream));
        }
    }
    // (1) classification of the own particles + the messages for both neighbours
    for (size_t r = 0; r < T.m.size(); ++r) {
        pse_handle *h = T.m[r];
        if (T.solo >= 0 && h->slab_rank != T.solo) continue;
        // (the counters of a step -- bins, cells, layers, messages -- were cleared by the last step's k_local_finish; a first step, or
        // one after a call that did not reach its end, clears them here)
        if (!h->loc.zeroed) HIPCHK(hipMemsetAsync(h->cnt_block, 0, h->loc.zero_ints * sizeof(int), h->stream));
        h->loc.zeroed = false;
        const LocalPool pool{h->keys, h->keys_s, h->perm, h->cell_cnt};
        launch_local_classify(ca[r], h->loc.g, h->dbox, h->nc, pool, h->loc.send[0], h->loc.send[1], h->loc.counters, h->loc.err, h->loc.layer_cnt, h->stream);
    }
    TRY(stage("classify"));
    // (2) ONE exchange: my left message goes left, what arrives from the right is the right neighbour's left message
    TRY(team_run_exchange(T, [&](pse_handle *h) {
        const LocalGeom &g = h->loc.g;
        const int L = (g.rank + g.G - 1) % g.G, R = (g.rank + 1) % g.G;
        std::vector<Xfer> ops;
        const size_t body = h->loc.msg - LOCAL_HDR;
        // the records, and in front of them -- one more transfer, 8 bytes -- the sender's two record counters as they lie in its memory
        ops.push_back(Xfer{h->loc.send[0] + LOCAL_HDR, body, L, h->loc.recv[1] + LOCAL_HDR, body, R});
        ops.push_back(Xfer{h->loc.send[1] + LOCAL_HDR, body, R, h->loc.recv[0] + LOCAL_HDR, body, L});
        ops.push_back(Xfer{(const double *)h->loc.counters, 1, L, h->loc.recv[1], 1, R});
        ops.push_back(Xfer{(const double *)h->loc.counters, 1, R, h->loc.recv[0], 1, L});
        return ops; }, false, DIAG_FIRST));
    TRY(stage("first exchange"));
    // (3) cell sort of what the rank keeps: own particles that stayed, arrivals, ghosts
    for (size_t r = 0; r < T.m.size(); ++r) {
        pse_handle *h = T.m[r];
        if (T.solo >= 0 && h->slab_rank != T.solo) continue;
        const LocalPool pool{h->keys, h->keys_s, h->perm, h->cell_cnt};
        const int ncell = cells_total(h->nc);
        if ((size_t)ncell > h->n_cells_alloc) return fail(PSE_ERR_INVALID, "cell grid exceeds the capacity sized at creation");
        launch_local_bin_incoming(h->loc.recv[0], h->loc.recv[1], h->loc.g, h->dbox, h->nc, pool, h->loc.err, h->loc.layer_cnt, h->stream);
        const LocalRegions rg = local_regions(h);
        launch_local_offsets(h->cell_cnt, h->loc.layer_cnt, h->cell_off, h->loc.g, rg, layer_cells(h->nc), h->loc.rows, h->loc.err, h->stream);
        launch_local_scatter(h->cell_off, h->loc.g, pool, h->vals, h->loc.rows, h->stream);
        const FarBinArgs far = far_bin_args(h->G, h->sw);
        const LocalSorted out{h->pos_s, h->posf_s, h->pv, h->f_s, h->tag_s, h->loc.porig_s, h->loc.mass_s, h->loc.image_s, noise ? h->psi_s : nullptr};
        launch_local_permute(ca[r], h->loc.recv[0], h->loc.recv[1], h->loc.g, h->dbox, h->cell_off, pool, h->vals, h->loc.rows, out, &far, h->par.seed,
                             timestep, h->ts_off, h->stream);
        h->sorted_N = 0; h->nb_valid = false; h->vl_valid = false; h->w_is_mpsi = false; h->pv_is_f = true;
        CellRanges need{};   // (the generic far-field kernels: rows outside these hold no particle)
        need.n = 3;
        for (int q = 0; q < 3; ++q) { need.c0[q] = rg.c0[q]; need.c1[q] = rg.c1[q] - 1; }
        h->sw.need = need; h->sw.cell_off = h->cell_off; h->sw.rows_local = 1;
    }
    // (4) the far-field chain (side lane), the near field and the Lanczos blocks (main lane)
    TRY(stage("sort + permute"));
    // The side lane forks from the sort, but the main lane's pass is queued FIRST: the near field + Lanczos chain is the longer of the
    // two (the far-field lane ends ~250 us earlier at eight ranks), and a dispatch that reaches the device behind the spread waits
    // for the spread's workgroups to leave the CUs.
    WavePump pump;
    const WaveArgs wa{T.m[0]->loc.rows_cap, noise, kT, dt, timestep};
    TRY(pump.fork(T));
    for (pse_handle *h : act(T)) {
        const LocalRows *R = h->loc.rows;
        const int nco = h->n_intervals * 2 * RS_NCOEF;
        if (noise)   // the pass that builds the pair list applies M_real to F and psi together, on the own rows + the adjacent ghost layers
            launch_mreal(h->pos_s, h->posf_s, h->f_s, h->ur_s, RowMap{}, h->cell_off, h->dbox, h->nc, h->d.rcut, h->d.self, h->coef, nco, h->nb,
                         MREAL_BUILD_LIST, h->stream, h->psi_s, h->w_s, VerletList{}, VL_NONE, h->pv, nullptr, 0, nullptr, Gate{},
                         DevRowArgs{&R->own1, R, h->loc.stage_w1, h->loc.rows_cap, h->loc.g.c_g});
        else
            launch_mreal(h->pos_s, h->posf_s, h->f_s, h->ur_s, RowMap{}, h->cell_off, h->dbox, h->nc, h->d.rcut, h->d.self, h->coef, nco, h->nb,
                         MREAL_CELLS, h->stream, nullptr, nullptr, VerletList{}, VL_NONE, h->pv, nullptr, 0, nullptr, Gate{},
                         DevRowArgs{&R->own, R, nullptr, h->loc.g.c_own, 0});
        h->nb_valid = noise; h->w_is_mpsi = noise;
    }
    TRY(stage("near field"));
    TRY(pump.start(T, wa, T.m[0]->tun.team_sched, &mask));
    TRY(stage("records + spread + forward transforms"));
    if (noise) TRY(lanczos_local(T, T.m[0]->d.error, m_io, &pump));
    TRY(stage("lanczos"));
    TRY(pump.drain());
    TRY(stage("far field"));
    // (5) join the lanes; the end of the step on the own rows, written to the caller's arrays
    for (size_t r = 0; r < T.m.size(); ++r) {
        pse_handle *h = T.m[r];
        if (T.solo >= 0 && h->slab_rank != T.solo) continue;
        if (h->side_on) {
            if (h == act(T)[0]) diag_mark(T, 3, h->side);
            HIPCHK(hipEventRecord(h->ev_join, h->side));
            HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
        }
        LocalFinish f{};
        f.rows = h->loc.rows; f.st = noise ? h->lz_state : nullptr;
        f.zero = h->cnt_block; f.n_zero = (int)h->loc.zero_ints;
        f.psi_s = h->psi_s; f.V = h->V; f.stride = h->n_pad; f.scal = h->scal; f.scale = noise ? std::sqrt(2.0 * kT / dt) : 0.0;
        f.uw_s = h->uw_s; f.ur_s = h->ur_s;
        f.porig_s = h->loc.porig_s; f.f_s = h->f_s; f.mass_s = h->loc.mass_s; f.image_s = h->loc.image_s; f.tag_s = h->tag_s;
        f.integrate = integrate; f.dt = dt; f.shear_rate = shear_rate;
        launch_local_finish(f, ca[r], h->dbox, h->loc.g.c_own, h->stream);
        HIPCHK(hipMemcpyAsync(h->loc.err_host, h->loc.err, sizeof(int), hipMemcpyDeviceToHost, h->stream));   // (read by the NEXT call: nothing waits)
        HIPCHK(hipGetLastError());
        h->loc.zeroed = true;
    }
    diag_mark(T, 1, act(T)[0]->stream);
    return 0;
}

// ---- single-handle C-ABI -------------------------------------------------------------------------------------------
extern "C" int pse_mobility(pse_handle *h, const pse_double4 *pos, const pse_double4 *force, pse_double4 *vel,
                            const unsigned *group, unsigned N, int parts) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    pse_team T;
    TRY(team_of_one(h, T));
    return do_mobility(T, {Args{(const double4 *)pos, (const double4 *)force, (double4 *)vel}}, group, N, parts);
}

extern "C" int pse_brownian_velocity(pse_handle *h, const pse_double4 *pos, const pse_double4 *force, pse_double4 *vel,
                                     const unsigned *group, unsigned N, double kT, double dt, unsigned timestep,
                                     int *lanczos_m) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    pse_team T;
    TRY(team_of_one(h, T));
    return do_brownian(T, {Args{(const double4 *)pos, (const double4 *)force, (double4 *)vel}}, group, N, kT, dt, timestep,
                       lanczos_m);
}

// The two halves of a Brownian evaluation on their own (a FUNCTIONAL split for two GPUs): parts = 1 the real-space
// half M_real.F + sqrt(2kT/dt) M_real^{1/2} psi, parts = 2 the wave-space half M_wave.F + the k-space noise; the halves add up to
// what pse_brownian_velocity returns (the reference computes them in one call and shares its inverse FFT and gather between the
// deterministic and the stochastic wave part as this does, PSEv1/Brownian.cu:772-923).
extern "C" int pse_brownian_velocity_part(pse_handle *h, const pse_double4 *pos, const pse_double4 *force, pse_double4 *vel,
                                          const unsigned *group, unsigned N, double kT, double dt, unsigned timestep, int parts,
                                          int *lanczos_m) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    if (parts < 1 || parts > 3) return fail(PSE_ERR_INVALID, "parts must select real (1), wave (2) or both (3)");
    if (!pos || !force || !vel) return fail(PSE_ERR_INVALID, "null array");
    pse_team T;
    TRY(team_of_one(h, T));
    return do_brownian(T, {Args{(const double4 *)pos, (const double4 *)force, (double4 *)vel}}, group, N, kT, dt, timestep, lanczos_m, parts);
}

// K15 alone (gpu_stokes_step_one_kernel, PSEv1/Stokes.cu:137-192): the Euler update + wrap for velocities the caller has put together
extern "C" int pse_integrate(pse_handle *h, pse_double4 *pos, const pse_double4 *vel, pse_double3 *accel, pse_int3 *image,
                             const pse_double4 *net_force, const unsigned *group, unsigned N, double dt, double shear_rate) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    if (!pos || !vel || !accel || !image || !net_force) return fail(PSE_ERR_INVALID, "null array");
    if (!(dt > 0)) return fail(PSE_ERR_INVALID, "need dt > 0");
    TRY(check_n(h, N));
    HIPCHK(hipSetDevice(h->device));
    launch_integrate((double4 *)pos, (const double4 *)vel, (double3 *)accel, (int3 *)image, (const double4 *)net_force, group, (int)N, h->dbox, dt,
                     shear_rate, h->stream);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int pse_step(pse_handle *h, pse_double4 *pos, pse_double4 *vel, pse_double3 *accel, pse_int3 *image,
                        const pse_double4 *net_force, const unsigned *group, unsigned N, double kT, double dt,
                        unsigned timestep, double shear_rate, int *lanczos_m) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    pse_team T;
    TRY(team_of_one(h, T));
    return do_step(T, {StepArgs{(double4 *)pos, (double4 *)vel, (double3 *)accel, (int3 *)image, (const double4 *)net_force}},
                   group, N, kT, dt, timestep, shear_rate, lanczos_m);
}

extern "C" int pse_sqrt_mreal(pse_handle *h, const pse_double4 *pos, const pse_double4 *psi, pse_double4 *out,
                              const unsigned *group, unsigned N, double tol, int *lanczos_m) {
    TRY(check_n(h, N));
    if (!pos || !psi || !out) return fail(PSE_ERR_INVALID, "null array");
    pse_team T;
    TRY(team_of_one(h, T));
    TRY(prepare(h, (const double4 *)pos, (const double4 *)psi, group, (int)N));   // f_s <- psi in sorted order
    HIPCHK(hipMemcpyAsync(h->psi_s, h->f_s, (size_t)N * sizeof(double4), hipMemcpyDeviceToDevice, h->stream));
    TRY(lanczos(T, (int)N, tol, 1.0, lanczos_m));
    launch_scatter_sum(h->ub_s, nullptr, nullptr, h->tag_s, (int)N, (double4 *)out, h->stream);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int pse_pair_repulsion(pse_handle *h, const pse_double4 *pos, pse_double4 *force, const unsigned *group, unsigned N,
                                  double k, double sigma, int accumulate) {
    TRY(check_n(h, N));
    if (!pos || !force) return fail(PSE_ERR_INVALID, "null array");
    if (!(sigma > 0.0) || sigma > h->d.rcut)
        return fail(PSE_ERR_INVALID, "repulsion range %.4f outside (0, rcut = %.4f]: the cell list is built for the hydrodynamic cutoff",
                    sigma, h->d.rcut);
    TRY(prepare(h, (const double4 *)pos, nullptr, group, (int)N, false, true));
    launch_pair_repulsion(h->pos_s, h->tag_s, (int)N, h->cell_off, h->dbox, h->nc, k, sigma, accumulate, (double4 *)force, h->stream);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int pse_random_psi(pse_handle *h, pse_double4 *psi, const unsigned *group, unsigned N, unsigned timestep) {
    TRY(check_n(h, N));
    if (!psi) return fail(PSE_ERR_INVALID, "null array");
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
    (void)hipFree(buf);
    if (e != hipSuccess) return fail(PSE_ERR_HIP, "pse_eval_realspace: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int pse_debug_matvec_ms(pse_handle *h, int reps, float *ms_per_launch) {
    if (!h || !ms_per_launch || reps < 1) return fail(PSE_ERR_INVALID, "pse_debug_matvec_ms: bad argument");
    HIPCHK(hipSetDevice(h->device));
    if (h->n_slabs != 1 || !h->nb_valid || h->nb.cap <= 0 || h->sorted_N <= 0)
        return fail(PSE_ERR_INVALID, "pse_debug_matvec_ms: needs the pair list of a Brownian call on a single-GPU engine (call it right after one)");
    const size_t stride = h->n_pad;
    const int N = h->sorted_N;
    RowMap rm{};
    rm.n = 1; rm.lo[0] = 0; rm.hi[0] = N;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto run = [&](int n) {   // the launch of a Lanczos iteration j >= 1 (vector V[1], its mirror, the sums against V[0] = psi), the kernel alone
        for (int r = 0; r < n; ++r)
            launch_mreal_lanczos(h->pos_s, h->V + stride, h->w_s, rm, h->cell_off, h->dbox, h->nc, h->d.rcut, h->d.self, h->coef, h->nb,
                                 LzFuse{h->psi_s, h->partials, h->npart_cap, nullptr, nullptr, nullptr}, h->scal, nullptr, nullptr, h->stream,
                                 VerletList{}, 1, nullptr, DevRowArgs{}, h->vq, true);
    };
    run(2);
    HIPCHK(hipEventRecord(e0, h->stream));
    run(reps);
    HIPCHK(hipEventRecord(e1, h->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    HIPCHK(hipGetLastError());
    *ms_per_launch = ms / (float)reps;
    return 0;
}
extern "C" int pse_debug_vq_roundtrip(int n, const double *rows_host, double *out_host) {
    if (n < 0 || (n > 0 && (!rows_host || !out_host))) return fail(PSE_ERR_INVALID, "pse_debug_vq_roundtrip: null argument");
    if (n == 0) return 0;
    double *d_in = nullptr, *d_out = nullptr;
    hipError_t e = hipMalloc((void **)&d_in, (size_t)3 * n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, (size_t)3 * n * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(d_in, rows_host, (size_t)3 * n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) { launch_vq_roundtrip(d_in, d_out, n, nullptr); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpy(out_host, d_out, (size_t)3 * n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_in); (void)hipFree(d_out);
    if (e != hipSuccess) return fail(PSE_ERR_HIP, "pse_debug_vq_roundtrip: %s", hipGetErrorString(e));
    return 0;
}
extern "C" int pse_debug_grid_placement(pse_handle *h, int *tried, float *ms_first, float *ms_kept) {
    if (!h || !tried || !ms_first || !ms_kept) return fail(PSE_ERR_INVALID, "pse_debug_grid_placement: null argument");
    *tried = h->place_tried; *ms_first = h->place_ms_first; *ms_kept = h->place_ms_kept;
    return 0;
}
extern "C" int pse_debug_kvector(pse_handle *h, int n, const int *ijk_host, double *out_host) {
    if (!h || !ijk_host || !out_host || n <= 0) return fail(PSE_ERR_INVALID, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    for (int t = 0; t < n; ++t)
        if (ijk_host[3 * t] < 0 || ijk_host[3 * t] >= h->G.Nx || ijk_host[3 * t + 1] < 0 || ijk_host[3 * t + 1] >= h->G.Ny ||
            ijk_host[3 * t + 2] < 0 || ijk_host[3 * t + 2] >= h->G.Nz)
            return fail(PSE_ERR_INVALID, "node %d outside the %d x %d x %d grid", t, h->G.Nx, h->G.Ny, h->G.Nz);
    int *d_ijk = nullptr;
    double *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_ijk, (size_t)3 * n * sizeof(int)));
    hipError_t e = hipMalloc((void **)&d_out, (size_t)5 * n * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(d_ijk, ijk_host, (size_t)3 * n * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_debug_kop(d_ijk, n, h->G, h->dbox, h->d.xi, h->d.eta, d_out, h->stream);
        e = hipStreamSynchronize(h->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out_host, d_out, (size_t)5 * n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_ijk);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(PSE_ERR_HIP, "pse_debug_kvector: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int pse_debug_spread(pse_handle *h, const pse_double4 *pos, const pse_double4 *force, const unsigned *group, unsigned N) {
    TRY(check_n(h, N));
    if (!pos || !force) return fail(PSE_ERR_INVALID, "null array");
    if (h->n_slabs > 1) return fail(PSE_ERR_INVALID, "this handle is a slab rank: drive it through a pse_team");
    TRY(prepare(h, (const double4 *)pos, (const double4 *)force, group, (int)N));
    const DGrid &G = h->G;
    const size_t nr = (size_t)(G.nxl + G.hl + G.nhalo) * G.Ny * G.Nz;
    if (spread_needs_zero(G)) HIPCHK(hipMemsetAsync(h->rgrid, 0, 3 * nr * sizeof(double), h->stream));
    HIPCHK(launch_far_records(h->pos_s, h->f_s, (int)N, G, h->dbox, h->sw, h->stream));
    HIPCHK(launch_spread(h->pos_s, h->f_s, (int)N, h->rgrid, h->rgrid + nr, h->rgrid + 2 * nr, G, h->dbox, h->sw, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int pse_debug_copy_grid(pse_handle *h, int stage, double *host_out) {
    if (!h || !host_out) return fail(PSE_ERR_INVALID, "bad argument");
    (void)stage;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->side) HIPCHK(hipStreamSynchronize(h->side));
    const size_t plane = (size_t)h->G.Ny * h->G.Nz, full = plane * (h->G.nxl + h->G.hl + h->G.nhalo);
    for (int c = 0; c < 3; ++c)   // the slab's own planes of each component (halo planes are not copied)
        HIPCHK(hipMemcpy(host_out + (size_t)c * plane * h->G.nxl, h->rgrid + c * full + plane * h->G.hl, plane * h->G.nxl * sizeof(double),
                         hipMemcpyDeviceToHost));
    return 0;
}

// ---- team C-ABI (multi-GPU) --------------------------------------------------------------------------------------
extern "C" int pse_team_unique_id(void *id128_host) {
    if (!id128_host) return fail(PSE_ERR_INVALID, "null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    memcpy(id128_host, &id, sizeof id);
    return 0;
}

extern "C" int pse_team_destroy(pse_team *T);
// One double round the ring through the team's own exchange path (RCCL group or the host program's transport): every rank sends its
// number to the right neighbour and must receive the left neighbour's.  A transport that does not deliver -- a rank that never
// joined, crossed peers -- is reported here, at creation, not as a hang or as wrong physics in the first step.
static int team_ring_test(pse_team &T) {
    if (T.G < 2 || !remote(T)) return 0;
    pse_handle *h = T.m[0];
    HIPCHK(hipSetDevice(h->device));
    double *buf = nullptr;
    HIPCHK(hipMalloc((void **)&buf, 2 * sizeof(double)));
    const double mine = 1000.0 + h->slab_rank;
    double got = -1.0;
    int rc = 0;
    hipError_t e = hipMemcpy(buf, &mine, sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const int L = (h->slab_rank + T.G - 1) % T.G, R = (h->slab_rank + 1) % T.G;
        rc = team_exchange(T, {Xfer{buf, 1, R, buf + 1, 1, L}}, false);
        if (!rc) e = hipStreamSynchronize(h->stream);
        if (!rc && e == hipSuccess && T.comm) e = hipStreamSynchronize(T.comm);
        if (!rc && e == hipSuccess) e = hipMemcpy(&got, buf + 1, sizeof(double), hipMemcpyDeviceToHost);
        if (!rc && e == hipSuccess && got != 1000.0 + L)
            rc = fail(PSE_ERR_COMM, "team ring test: rank %d expected %g from its left neighbour %d and received %g -- the transport does not "
                      "connect the ranks as a ring of %d", h->slab_rank, 1000.0 + L, L, got, T.G);
    }
    (void)hipFree(buf);
    if (e != hipSuccess && !rc) rc = fail(PSE_ERR_COMM, "team ring test failed: %s", hipGetErrorString(e));
    return rc;
}
// RCCL communicator of a one-member-per-process team (in-process teams have nothing to connect)
static int team_connect(pse_team *T, const void *id128_host) {
    if (T->m.size() != 1 || !id128_host) return 0;
    ncclUniqueId id;
    memcpy(&id, id128_host, sizeof id);
    HIPCHK(hipSetDevice(T->m[0]->device));
    {   // a rank that never joins leaves ncclCommInitRank waiting for ever: wait for it on a thread, with a deadline (PSE_TEAM_CONNECT_TIMEOUT, s)
        const char *te_ = getenv("PSE_TEAM_CONNECT_TIMEOUT");
        const int limit = te_ ? std::max(1, atoi(te_)) : 300;
        // The thread writes into a slot of its OWN (shared with this function): on a timeout the caller destroys the team, and a
        // peer that joins late -- or a bootstrap that fails later -- must not find freed memory behind the pointer.
        struct Slot { std::promise<ncclResult_t> done; ncclComm_t comm = nullptr; std::atomic<bool> abandoned{false}; };
        auto slot = std::make_shared<Slot>();
        std::future<ncclResult_t> fut = slot->done.get_future();
        const int G_ = T->G, r_ = T->m[0]->slab_rank, dev_ = T->m[0]->device;
        std::thread([slot, G_, id, r_, dev_]() {
            (void)hipSetDevice(dev_);
            const ncclResult_t r = ncclCommInitRank(&slot->comm, G_, id, r_);
            if (slot->abandoned.load() && r == ncclSuccess && slot->comm) { (void)ncclCommAbort(slot->comm); slot->comm = nullptr; }   // nobody will ever use it
            slot->done.set_value(r);
        }).detach();
        if (fut.wait_for(std::chrono::seconds(limit)) != std::future_status::ready) {
            slot->abandoned.store(true);
            return fail(PSE_ERR_COMM, "RCCL communicator of %d ranks did not form within %d s: a peer is missing (rank %d waited; "
                        "PSE_TEAM_CONNECT_TIMEOUT sets the limit)", G_, limit, r_);
        }
        const ncclResult_t r0 = fut.get();
        if (r0 != ncclSuccess) return fail(PSE_ERR_COMM, "ncclCommInitRank failed: %s", ncclGetErrorString(r0));
        T->nccl = slot->comm;
    }
    // Two compute lanes + a communication stream of their own are OPT-IN for an RCCL team (PSE_TEAM_LANES=1): no multi-GPU node has
    // run that path yet.  Default: one stream carries the kernels and the RCCL calls, in program order -- nothing can interleave.
    {
        const char *e = getenv("PSE_TEAM_LANES");
        T->lanes = e && atoi(e) > 0;
    }
    if (T->G > 1 && T->lanes) {   // the one stream every RCCL call of the team is issued on (see pse_team)
        HIPCHK(hipStreamCreateWithFlags(&T->comm, hipStreamNonBlocking));
        T->evs.resize(32);
        for (hipEvent_t &e : T->evs) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    return team_ring_test(*T);
}

extern "C" int pse_team_create(pse_handle **members, int n_members, const void *id128_host, pse_team **out) {
    if (!members || n_members < 1 || !out) return fail(PSE_ERR_INVALID, "bad argument");
    *out = nullptr;
    if (!members[0]) return fail(PSE_ERR_INVALID, "null member");
    const int G = members[0]->n_slabs;
    std::vector<char> seen(G, 0);
    for (int i = 0; i < n_members; ++i) {
        pse_handle *h = members[i];
        if (!h || h->n_slabs != G) return fail(PSE_ERR_INVALID, "members disagree on n_slabs");
        if (h->d.Nx != members[0]->d.Nx || h->d.Ny != members[0]->d.Ny || h->d.Nz != members[0]->d.Nz || h->n_pad != members[0]->n_pad)
            return fail(PSE_ERR_INVALID, "members disagree on grid or capacity");
        if (seen[h->slab_rank]) return fail(PSE_ERR_INVALID, "slab rank %d appears twice", h->slab_rank);
        seen[h->slab_rank] = 1;
    }
    if (n_members == 1 && (G > 1 || id128_host)) {
        if (!id128_host) return fail(PSE_ERR_INVALID, "a one-rank-per-process team needs the RCCL unique id of rank 0");
    } else if (n_members != G) {
        return fail(PSE_ERR_INVALID, "an in-process team must hold all %d slab ranks (got %d)", G, n_members);
    }
    pse_team *T = new pse_team();
    T->m.assign(members, members + n_members);
    T->G = G;
    if (const char *e = getenv("PSE_TEAM_LANES")) T->lanes = atoi(e) > 0;   // in-process and host-staged teams: two lanes unless switched off
    if (n_members > 1)   // in-process team: one side stream for all members, so every lane is ordered by its stream alone
        for (int i = 1; i < n_members; ++i) members[i]->side = members[0]->side;
    int rc = team_connect(T, id128_host);
    if (rc) { std::string keep = error_text(); pse_team_destroy(T); error_text() = keep; return rc; }
    *out = T;
    return 0;
}

extern "C" int pse_team_create_transport(pse_handle *member, const pse_transport *transport, pse_team **out) {
    if (!member || !transport || !out || !transport->exchange) return fail(PSE_ERR_INVALID, "bad argument");
    *out = nullptr;
    if (member->n_slabs < 2) return fail(PSE_ERR_INVALID, "a team needs handles created with n_slabs >= 2");
    pse_team *T = new pse_team();
    T->m = {member};
    T->G = member->n_slabs;
    T->cb = *transport;
    T->has_cb = true;
    if (const char *e = getenv("PSE_TEAM_LANES")) T->lanes = atoi(e) > 0;
    if (int rc = team_ring_test(*T)) { std::string keep = error_text(); pse_team_destroy(T); error_text() = keep; return rc; }
    *out = T;
    return 0;
}

extern "C" int pse_team_destroy(pse_team *T) {
    if (!T) return 0;
    for (pse_handle *h : T->m) {   // members take their own side stream back (an in-process team shared member 0's)
        if (h->side != h->side_owned) {
            h->side = h->side_owned; h->side_on = false; h->wstream = h->stream;
            // the rocFFT execution infos were bound to member 0's side stream: rebind, or the next call with the wave chain on
            // the main stream finds (side_on, wstream) already as it wants them and the transforms run on the stale stream
            if (h->info_fwd) (void)rocfft_execution_info_set_stream(h->info_fwd, h->wstream);
            if (h->info_inv) (void)rocfft_execution_info_set_stream(h->info_inv, h->wstream);
        }
    }
    if (T->comm) { (void)hipStreamSynchronize(T->comm); }
    if (T->nccl) ncclCommDestroy(T->nccl);
    for (hipEvent_t e : T->evs) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : T->diag.ev) if (e) (void)hipEventDestroy(e);
    if (T->comm) (void)hipStreamDestroy(T->comm);
    if (T->stage) (void)hipHostFree(T->stage);
    delete T;
    return 0;
}

extern "C" int pse_team_debug_solo(pse_team *T, int slab_rank) {
    if (!T) return fail(PSE_ERR_INVALID, "null team");
    if (slab_rank < 0) { T->solo = -1; T->solo_m.clear(); return 0; }
    if (!loopback(*T)) return fail(PSE_ERR_INVALID, "pse_team_debug_solo needs an in-process team");
    for (pse_handle *h : T->m)
        if (h->slab_rank == slab_rank) { T->solo = slab_rank; T->solo_m = {h}; return 0; }
    return fail(PSE_ERR_INVALID, "no member with slab rank %d", slab_rank);
}

extern "C" int pse_team_mobility(pse_team *T, const pse_double4 *const *pos, const pse_double4 *const *force,
                                 pse_double4 *const *vel, const unsigned *group, unsigned N, int parts) {
    if (!T || !pos || !force || !vel) return fail(PSE_ERR_INVALID, "null argument");
    std::vector<Args> a;
    for (size_t r = 0; r < T->m.size(); ++r) a.push_back(Args{(const double4 *)pos[r], (const double4 *)force[r], (double4 *)vel[r]});
    return do_mobility(*T, a, group, N, parts);
}

extern "C" int pse_team_brownian_velocity(pse_team *T, const pse_double4 *const *pos, const pse_double4 *const *force,
                                          pse_double4 *const *vel, const unsigned *group, unsigned N, double kT, double dt,
                                          unsigned timestep, int *lanczos_m) {
    if (!T || !pos || !force || !vel) return fail(PSE_ERR_INVALID, "null argument");
    std::vector<Args> a;
    for (size_t r = 0; r < T->m.size(); ++r) a.push_back(Args{(const double4 *)pos[r], (const double4 *)force[r], (double4 *)vel[r]});
    return do_brownian(*T, a, group, N, kT, dt, timestep, lanczos_m);
}

extern "C" int pse_team_step(pse_team *T, pse_double4 *const *pos, pse_double4 *const *vel, pse_double3 *const *accel,
                             pse_int3 *const *image, const pse_double4 *const *net_force, const unsigned *group, unsigned N,
                             double kT, double dt, unsigned timestep, double shear_rate, int *lanczos_m) {
    if (!T || !pos || !vel || !accel || !image || !net_force) return fail(PSE_ERR_INVALID, "null argument");
    std::vector<StepArgs> a;
    for (size_t r = 0; r < T->m.size(); ++r)
        a.push_back(StepArgs{(double4 *)pos[r], (double4 *)vel[r], (double3 *)accel[r], (int3 *)image[r], (const double4 *)net_force[r]});
    return do_step(*T, a, group, N, kT, dt, timestep, shear_rate, lanczos_m);
}

extern "C" int pse_team_step_local(pse_team *T, pse_double4 *const *pos, pse_double4 *const *vel, pse_double3 *const *accel, pse_int3 *const *image,
                                   const pse_double4 *const *net_force, unsigned int *const *tag, unsigned int *const *n_local, double kT,
                                   double dt, unsigned int timestep, double shear_rate, int integrate, int *lanczos_m) {
    if (!T || !pos || !vel || !accel || !image || !net_force || !tag || !n_local) return fail(PSE_ERR_INVALID, "null argument");
    std::vector<LocalCaller> ca;
    for (size_t r = 0; r < T->m.size(); ++r)
        ca.push_back(LocalCaller{(double4 *)pos[r], (double4 *)vel[r], (double3 *)accel[r], (int3 *)image[r], (const double4 *)net_force[r], tag[r], n_local[r]});
    return local_call(*T, ca, kT, dt, timestep, shear_rate, integrate, lanczos_m);
}

// Every particle to the rank that owns it under the CURRENT box (after pse_set_box has taken the tilt through a Lees-Edwards flip):
// what HOOMD's domain decomposition does when the box is re-mapped (PSEv1/VariantShearFunction.cc:34-43 drives the flip).  Two
// exchanges of the team's own transfer list around three kernels (pse_local.hip); the host sizes the second exchange from the count
// rows of the first, so this call WAITS twice -- it runs once per unit of strain.
static int redistribute_local(pse_team &T, const std::vector<LocalCaller> &ca) {
    if (T.G < 2) return fail(PSE_ERR_INVALID, "pse_team_redistribute_local needs a team of >= 2 ranks");
    if (T.solo >= 0) return fail(PSE_ERR_INVALID, "pse_team_redistribute_local: not in solo mode");
    const int G = T.G, row = (G + 2 + 1) & ~1;      // ints per count row: G counts, capacity, particle count (+ padding to whole doubles)
    for (size_t r = 0; r < T.m.size(); ++r) {
        pse_handle *h = T.m[r];
        if (!h->loc.on) return fail(PSE_ERR_INVALID, "pse_team_redistribute_local: member %zu was not created with local_rows", r);
        if (h->loc.err_host[0]) return fail(PSE_ERR_INVALID, "owned-particle rank %d: an earlier step failed on the device (flags %d)", h->slab_rank, h->loc.err_host[0]);
        const LocalCaller &c = ca[r];
        if (!c.pos || !c.vel || !c.accel || !c.image || !c.force || !c.tag || !c.n_local) return fail(PSE_ERR_INVALID, "null array");
        HIPCHK(hipSetDevice(h->device));
        if (!h->loc.rd_send) {
            const size_t nrec = (size_t)h->loc.g.c_own * LOCAL_REC;
            TRY(dmalloc(h, &h->loc.rd_send, nrec)); TRY(dmalloc(h, &h->loc.rd_recv, nrec));
            TRY(dmalloc(h, &h->loc.rd_rows, (size_t)G * row)); TRY(dmalloc(h, &h->loc.rd_off, (size_t)2 * G));
            HIPCHK(hipHostMalloc((void **)&h->loc.rd_host, ((size_t)G * row + 2 * G) * sizeof(int)));
            h->info.device_bytes = h->bytes;
        }
    }
    // (1) new owners and the count row of every rank
    for (size_t r = 0; r < T.m.size(); ++r) {
        pse_handle *h = T.m[r];
        HIPCHK(hipSetDevice(h->device));
        int *mine = h->loc.rd_rows + (size_t)h->slab_rank * row;
        HIPCHK(hipMemsetAsync(h->loc.rd_rows, 0, (size_t)G * row * sizeof(int), h->stream));
        launch_redist_count(ca[r], h->loc.g, h->dbox, h->nc, h->keys, mine, h->loc.err, h->stream);
        HIPCHK(hipMemcpyAsync(mine + G, &h->loc.g.c_own, sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(mine + G + 1, ca[r].n_local, sizeof(int), hipMemcpyDeviceToDevice, h->stream));
    }
    TRY(team_run_exchange(T, [&](pse_handle *h) {
        std::vector<Xfer> ops;
        for (int p = 0; p < G; ++p) {
            if (p == h->slab_rank) continue;
            ops.push_back(Xfer{(const double *)(h->loc.rd_rows + (size_t)h->slab_rank * row), (size_t)row / 2, p,
                               (double *)(h->loc.rd_rows + (size_t)p * row), (size_t)row / 2, p});
        }
        return ops; }, false, DIAG_GHOST));
    for (pse_handle *h : T.m) {
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipMemcpyAsync(h->loc.rd_host, h->loc.rd_rows, (size_t)G * row * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(h->loc.err_host, h->loc.err, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    }
    for (pse_handle *h : T.m) { HIPCHK(hipSetDevice(h->device)); HIPCHK(hipStreamSynchronize(h->stream)); }
    // (2) every rank holds every row: the same verdict everywhere -- nothing moves unless everything fits
    const int *M = T.m[0]->loc.rd_host;
    for (pse_handle *h : T.m)
        if (h->loc.err_host[0]) return fail(PSE_ERR_INVALID, "owned-particle rank %d: flags %d while counting (16: n_local above capacity)", h->slab_rank, h->loc.err_host[0]);
    for (int d = 0; d < G; ++d) {
        long in = 0;
        for (int src = 0; src < G; ++src) in += M[(size_t)src * row + d];
        if (in > M[(size_t)d * row + G])
            return fail(PSE_ERR_INVALID, "redistribution: rank %d would own %ld particles, its arrays hold %d (pse_local_layout rows_own): nothing was moved", d, in, M[(size_t)d * row + G]);
    }
    // (3) pack by destination, exchange exactly those records, unpack into the caller's arrays
    for (size_t r = 0; r < T.m.size(); ++r) {
        pse_handle *h = T.m[r];
        HIPCHK(hipSetDevice(h->device));
        int *off = h->loc.rd_host + (size_t)G * row;     // [G] send offsets, then [G] zeros for the fill counters
        const int *mine = h->loc.rd_host + (size_t)h->slab_rank * row;
        int acc = 0;
        for (int d = 0; d < G; ++d) { off[d] = acc; acc += mine[d]; off[G + d] = 0; }
        HIPCHK(hipMemcpyAsync(h->loc.rd_off, off, 2 * G * sizeof(int), hipMemcpyHostToDevice, h->stream));
        launch_redist_pack(ca[r], h->loc.g, h->keys, h->loc.rd_off, h->loc.rd_off + G, h->loc.rd_send, h->stream);
    }
    TRY(team_run_exchange(T, [&](pse_handle *h) {
        const int me = h->slab_rank;
        const int *Mh = h->loc.rd_host;
        std::vector<Xfer> ops;
        size_t so = 0, ro = 0;
        std::vector<size_t> soff(G), roff(G);
        for (int p = 0; p < G; ++p) { soff[p] = so; so += (size_t)Mh[(size_t)me * row + p]; roff[p] = ro; ro += (size_t)Mh[(size_t)p * row + me]; }
        for (int p = 0; p < G; ++p) {   // (the diagonal block is a local copy in every transport: team_exchange)
            const size_t ns = (size_t)Mh[(size_t)me * row + p] * LOCAL_REC, nr = (size_t)Mh[(size_t)p * row + me] * LOCAL_REC;
            if (ns || nr) ops.push_back(Xfer{h->loc.rd_send + soff[p] * LOCAL_REC, ns, p, h->loc.rd_recv + roff[p] * LOCAL_REC, nr, p});
        }
        return ops; }, false, DIAG_FIRST));
    for (size_t r = 0; r < T.m.size(); ++r) {
        pse_handle *h = T.m[r];
        HIPCHK(hipSetDevice(h->device));
        long in = 0;
        for (int src = 0; src < G; ++src) in += h->loc.rd_host[(size_t)src * row + h->slab_rank];
        launch_redist_unpack(h->loc.rd_recv, (int)in, ca[r], h->stream);
        h->sorted_N = 0; h->nb_valid = false; h->vl_valid = false;
        HIPCHK(hipGetLastError());
    }
    return 0;
}

extern "C" int pse_team_redistribute_local(pse_team *T, pse_double4 *const *pos, pse_double4 *const *vel, pse_double3 *const *accel, pse_int3 *const *image,
                                           pse_double4 *const *net_force, unsigned int *const *tag, unsigned int *const *n_local) {
    if (!T || !pos || !vel || !accel || !image || !net_force || !tag || !n_local) return fail(PSE_ERR_INVALID, "null argument");
    std::vector<LocalCaller> ca;
    for (size_t r = 0; r < T->m.size(); ++r)
        ca.push_back(LocalCaller{(double4 *)pos[r], (double4 *)vel[r], (double3 *)accel[r], (int3 *)image[r], (const double4 *)net_force[r], tag[r], n_local[r]});
    return redistribute_local(*T, ca);
}

extern "C" int pse_local_layout(pse_handle *h, int *rows_own, int *rows_ghost, int *records, int *layers, int *layers_per_rank) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    if (!h->loc.on) return fail(PSE_ERR_INVALID, "not an owned-particle handle (pse_params.local_rows)");
    if (rows_own) *rows_own = h->loc.g.c_own;
    if (rows_ghost) *rows_ghost = h->loc.g.c_g;
    if (records) *records = h->loc.g.c_x;
    if (layers) *layers = h->loc.g.nx;
    if (layers_per_rank) *layers_per_rank = h->loc.g.per;
    return 0;
}

extern "C" int pse_set_lanczos_extra(pse_handle *h, int extra) {
    if (!h) return fail(PSE_ERR_INVALID, "null handle");
    if (extra > 32) return fail(PSE_ERR_INVALID, "at most 32 extra Lanczos iterations");
    h->lz_extra = extra < 0 ? -1 : extra;
    return 0;
}
extern "C" int pse_team_set_lanczos_extra(pse_team *T, int extra) {
    if (!T) return fail(PSE_ERR_INVALID, "null team");
    if (extra > 32) return fail(PSE_ERR_INVALID, "at most 32 extra Lanczos iterations");
    T->lz_extra = extra < 0 ? -1 : extra;
    return 0;
}

extern "C" int pse_team_local_status(pse_team *T, int *flags) {
    if (!T) return fail(PSE_ERR_INVALID, "null team");
    int any = 0;
    for (size_t r = 0; r < T->m.size(); ++r) {
        pse_handle *h = T->m[r];
        if (!h->loc.on) return fail(PSE_ERR_INVALID, "member %zu is not an owned-particle handle", r);
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->side) HIPCHK(hipStreamSynchronize(h->side));
        int f = 0;
        HIPCHK(hipMemcpy(&f, h->loc.err, sizeof(int), hipMemcpyDeviceToHost));
        h->loc.err_host[0] = f;
        if (flags) flags[r] = f;
        any |= f;
    }
    if (any) return fail(PSE_ERR_INVALID, "owned-particle step failed on the device (flags %d: 1 own rows, 2 ghost rows, 4 message capacity exceeded, "
                         "8 a particle moved beyond the neighbour, 16 n_local above capacity)", any);
    return 0;
}

extern "C" int pse_team_set_diag(pse_team *T, int enabled) {
    if (!T) return fail(PSE_ERR_INVALID, "null team");
    HIPCHK(hipSetDevice(T->m[0]->device));
    if (enabled && T->diag.ev.empty()) {
        T->diag.ev.resize(2 * PSE_DIAG_MAX + 4);
        for (hipEvent_t &e : T->diag.ev) HIPCHK(hipEventCreate(&e));
    }
    T->diag.on = enabled != 0;
    T->diag.n = 0;
    return 0;
}
extern "C" int pse_team_get_diag(pse_team *T, pse_team_diag *out) {
    if (!T || !out) return fail(PSE_ERR_INVALID, "null argument");
    memset(out, 0, sizeof *out);
    if (!T->diag.on) return fail(PSE_ERR_INVALID, "diagnosis is off (pse_team_set_diag)");
    pse_handle *h = T->solo >= 0 ? T->solo_m[0] : T->m[0];
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->side) HIPCHK(hipStreamSynchronize(h->side));
    out->n_exchanges = T->diag.n;
    for (int q = 0; q < T->diag.n; ++q) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, T->diag.ev[2 * q], T->diag.ev[2 * q + 1]) != hipSuccess) ms = 0.0f;
        out->kind[q] = T->diag.kind[q]; out->lane[q] = T->diag.lane[q]; out->device_us[q] = 1e3 * ms; out->host_us[q] = T->diag.host_us[q];
        out->bytes[q] = T->diag.bytes[q];
    }
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, T->diag.ev[2 * PSE_DIAG_MAX], T->diag.ev[2 * PSE_DIAG_MAX + 1]) == hipSuccess) out->main_lane_ms = ms;
    if (T->diag.side_used && hipEventElapsedTime(&ms, T->diag.ev[2 * PSE_DIAG_MAX + 2], T->diag.ev[2 * PSE_DIAG_MAX + 3]) == hipSuccess) out->side_lane_ms = ms;
    out->critical_path_ms = out->main_lane_ms;
    return 0;
}
