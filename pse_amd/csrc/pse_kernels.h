// Launch wrappers of the HIP kernels (defined in pse_kernels.hip).  All take the stream explicitly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pse_device.h"

namespace pse {

// ---- particle binning ------------------------------------------------------------------------------------
// counting sort by cell, equal to a stable sort by key: perm[slot] = input index, cell_off[c] = first slot of cell c.
// keys, rank, slots: N unsigned each; cnt: ncell + 1 ints; tmp: cell_sort_temp_bytes(ncell)
size_t cell_sort_temp_bytes(size_t ncell);
// A slab rank of a team works on its own cell layers and the neighbouring ghost layer on either side only: up to three ranges
// [c0, c1) of storage cells (x is the slowest cell index, so a range of layers is a range of cells and -- through cell_off -- a
// range of rows).  n = 0: everything (single GPU).  EVERY particle is still counted (the row offsets are global); the ordering
// inside the cells, the gather into cell order, the far-field records and the random vector are done for these rows alone.
struct CellRanges {
    int n, c0[3], c1[3];
    __device__ __forceinline__ bool cell(int c) const {
        if (n == 0) return true;
        for (int q = 0; q < n; ++q) if (c >= c0[q] && c < c1[q]) return true;
        return false;
    }
    __device__ __forceinline__ bool row(int s, const int *__restrict__ cell_off) const {
        if (n == 0) return true;
        for (int q = 0; q < n; ++q) if (s >= cell_off[c0[q]] && s < cell_off[c1[q]]) return true;
        return false;
    }
};
// Rows a near-field pass works on: up to three ranges [lo, hi) of sorted rows (a slab rank's own rows FIRST, then the ghost cell
// layers of its neighbours -- the last layers of the box wrap round to rank 0, hence three), laid out back to back in "list rows":
// range k starts at list row base[k], a multiple of 256, so that no workgroup straddles two ranges.  The per-step pair list is
// stored by list row; a pass over the own rows alone (n = 1) therefore reads the list a pass over all ranges wrote.
struct RowMap {
    int n, lo[3], hi[3], base[3];
    __host__ __device__ int list_rows() const { return n ? base[n - 1] + ((hi[n - 1] - lo[n - 1] + 255) & ~255) : 0; }
    // sorted row of list row lr, or -1 (padding)
    __device__ __forceinline__ int first_row() const { return hi[0] > lo[0] ? lo[0] : (n > 1 && hi[1] > lo[1] ? lo[1] : lo[2]); }
    __device__ __forceinline__ int row(int lr) const {   // (constant indices only: the struct then lives in scalar registers, also when it was loaded from memory)
        int l = lo[0], h = hi[0], b = base[0];
        if (n > 1 && lr >= base[1]) { l = lo[1]; h = hi[1]; b = base[1]; }
        if (n > 2 && lr >= base[2]) { l = lo[2]; h = hi[2]; b = base[2]; }
        const int i = l + (lr - b);
        return i < h ? i : -1;
    }
};
struct RowRanges { int n, lo[3], hi[3]; };   // up to three disjoint row ranges (own rows and the two ghost regions)
// Row space of an owned-particle rank (pse_team_step_local), written ON THE DEVICE by the cell sort of the step (k_local_scatter):
// the host never learns these numbers inside a step -- kernels are launched for the capacities and read their rows here.
//   own rows [0, n_own) | left ghosts [c_own, c_own + n_gl) | right ghosts [c_own + c_g, c_own + c_g + n_gr)      (c_own, c_g: capacities)
struct LocalRows {
    int n_own, n_gl, n_gr;
    int first_end;                // own rows [0, first_end): the first `depth` cell layers -- the left neighbour's right ghosts
    int last_begin;               // own rows [last_begin, n_own): the last `depth` layers -- the right neighbour's left ghosts
    int ok;                       // 0: a capacity was exceeded (flagged): every range here is empty
    RowMap own, own1;             // own rows; own rows + the ADJACENT ghost layer on either side (list-row bases as RowMap wants them)
    RowRanges all;                // own + all ghost rows
};
// rows of a near-field / vector pass given on the device (rm nullable: the RowMap passed by value counts), + where a mat-vec
// parks the rows of the last `depth` layers on their way to the right neighbour (their first row is known on the device only)
struct DevRowArgs {
    const RowMap *rm;             // device
    const LocalRows *lr;          // device (nullable): for stage_hi
    double4 *stage_hi;            // rows [lr->last_begin, lr->n_own) of the result also go to stage_hi[row - last_begin]
    int rows_cap;                 // list rows to launch for (>= what *rm will say)
    int stage_cap;                // rows stage_hi holds (the ghost capacity c_g)
};
// a RowMap in scalars: what a kernel works with when the map may come from device memory (assigning a loaded struct to the by-value
// kernel argument put it in scratch memory: 44 bytes per lane and 20 % on the pair-list mat-vec)
struct RowMapRegs {
    int n, l0, l1, l2, h0, h1, h2, b1, b2;
    __device__ __forceinline__ RowMapRegs(const RowMap &m, const RowMap *dev) {
        n = m.n; l0 = m.lo[0]; l1 = m.lo[1]; l2 = m.lo[2]; h0 = m.hi[0]; h1 = m.hi[1]; h2 = m.hi[2]; b1 = m.base[1]; b2 = m.base[2];
        if (dev) { n = dev->n; l0 = dev->lo[0]; l1 = dev->lo[1]; l2 = dev->lo[2]; h0 = dev->hi[0]; h1 = dev->hi[1]; h2 = dev->hi[2]; b1 = dev->base[1]; b2 = dev->base[2]; }
    }
    __device__ __forceinline__ int list_rows() const { return n == 0 ? 0 : (n == 1 ? ((h0 - l0 + 255) & ~255) : (n == 2 ? b1 + ((h1 - l1 + 255) & ~255) : b2 + ((h2 - l2 + 255) & ~255))); }
    __device__ __forceinline__ int first_row() const { return h0 > l0 ? l0 : (n > 1 && h1 > l1 ? l1 : l2); }
    __device__ __forceinline__ int row(int lr) const {
        int l = l0, h = h0, b = 0;
        if (n > 1 && lr >= b1) { l = l1; h = h1; b = b1; }
        if (n > 2 && lr >= b2) { l = l2; h = h2; b = b2; }
        const int i = l + (lr - b);
        return i < h ? i : -1;
    }
};
inline RowMap row_map(int lo, int hi) { return RowMap{1, {lo, 0, 0}, {hi, 0, 0}, {0, 0, 0}}; }
inline RowMap row_map(const int (*rg)[2], int n) {
    RowMap m{};
    m.n = n;
    int base = 0;
    for (int k = 0; k < n; ++k) { m.lo[k] = rg[k][0]; m.hi[k] = rg[k][1]; m.base[k] = base; base += (rg[k][1] - rg[k][0] + 255) & ~255; }
    return m;
}

// A launch that runs only on one outcome of a decision taken ON THE DEVICE earlier in the stream (asynchronous mode: whether the kept
// neighbour list may be reused): every workgroup reads the word and leaves at once when the gate is closed.  word = null: always open.
struct Gate {
    const int *word;
    int want;   // 1: run if *word != 0 (rebuild), 0: run if *word == 0 (reuse)
    __device__ __forceinline__ bool closed() const { return word != nullptr && ((*word != 0) != (want != 0)); }
};
// gate word <- flags[0] | flags[1] (moved beyond the skin | a row overflowed at the build); a rebuild starts with flags[1] cleared
void launch_gate_decide(int *flags, int *word, hipStream_t s);
void launch_gate_zero(Gate g, int *a, size_t na, int *b, size_t nb, hipStream_t s);          // two int ranges zeroed if the gate is open
void launch_gate_copy(Gate g, double4 *dst, const double4 *src, size_t n, hipStream_t s);    // dst <- src if the gate is open
// n > 0 (a slab rank that keeps only some cell layers): the particles of cells it does not keep are counted per slab of
// cells_per_slab storage cells, on cell book[slab] -- the first cell of that slab the rank does not keep
struct SlabBook { int n, cells_per_slab, spread, book[64]; };   // spread: the counts go to cells book[slab] .. book[slab] + spread - 1 (one layer)
hipError_t cell_sort(const double4 *pos, const unsigned *group, int N, DBox box, DCells nc, unsigned *keys, unsigned *rank,
                     unsigned *slots, int *cnt, int ncell, void *tmp, size_t tmp_bytes, int *cell_off, unsigned *perm, hipStream_t s,
                     CellRanges need = CellRanges{}, SlabBook sb = SlabBook{}, bool cnt_is_zero = false,   // cnt[0 .. ncell] already zeroed by the caller
                     Gate gate = Gate{});
// exclusive prefix sum of n ints (the cell counts -> row offsets), same scratch as cell_sort
hipError_t launch_cell_scan(const int *cnt, int *out, int n, void *tmp, size_t tmp_bytes, hipStream_t s);
// pos_s[i] = wrapped position of particle perm[i], vec_s[i] = vec[tag].xyz, tag_s[i] = its index in the caller's arrays
// pos_build (nullable): the sorted positions the neighbour list was built at; a particle that has moved more than
// sqrt(half_skin2) from there (minimum image) sets flags[0] -- HOOMD's NeighborList distance check (r_buff / 2)
void launch_permute(const double4 *pos, const double4 *vec, const unsigned *group, const unsigned *perm, int N, DBox box,
                    double4 *pos_s, float4 *posf_s, double2 *pv, double4 *vec_s, unsigned *tag_s, hipStream_t s,
                    const double4 *pos_build = nullptr, double half_skin2 = 0.0, int *flags = nullptr,
                    CellRanges need = CellRanges{}, const int *cell_off = nullptr,    // a slab rank: the needed rows only
                    const struct FarBinArgs *far = nullptr,                           // rank the particles in their far-field bins
                    double4 *psi_s = nullptr, uint32_t seed = 0, uint32_t timestep = 0,    // draw the particle noise of this step (K14)
                    Gate gate = Gate{}, const uint32_t *ts_off = nullptr);             // ts_off (nullable): ... at timestep + *ts_off
void launch_permute_vec(const double4 *vec, const unsigned *tag_s, int N, double4 *vec_s, hipStream_t s);

// ---- near field (K9) -------------------------------------------------------------------------------------
// Per-step pair list, 16 B per pair, in wave-blocked ELL layout: four slots of the 64 rows a wavefront owns form one
// contiguous 4096-byte group [4 slots][64 lanes] of records (neighbour row 27 bits | sign of h | fr 26-bit fixed point | s = d sqrt|h| as
// three 22-bit mantissas under one exponent: nb_pack in pse_kernels.hip; pair term f v + sgn (s.v) s), groups of one wave back to back
// -- a wave streams one contiguous region with 16-byte loads only.  Read by the Lanczos mat-vecs only.
// Row r = i - lo (lo: first row of the rank, fixed within a step); record (r / 64) * cap + slot.
struct NbList {
    char *data;
    int *cnt;             // neighbour count per particle; -1 if it exceeded cap (dense cluster): that row walks the cells
    int cap;
};
// Neighbour (Verlet) list kept ACROSS steps, as the reference keeps HOOMD's NeighborListGPUBinned(rcut, r_buff = 0.4) with a
// distance check every step (PSEv1/integrate.py:60,79; Stokes.cc:433): every pair closer than rcut + skin when it was built.
// The same wave-blocked layout, entries only: groups of four slots [64 lanes][4 x u32 neighbour slot] = 1024 bytes.
struct VerletList {
    unsigned *idx;
    int *cnt;             // entries per row; -1: the row did not fit (flags[1] is set: the list is not reused)
    int cap;              // slots per row, a multiple of 4
    int *flags;           // [0] a particle moved beyond skin / 2 since the build, [1] a row overflowed at the build
    double rskin;         // rcut + skin
};
__host__ __device__ inline size_t verlet_list_bytes(size_t rows, int cap) { return ((rows + 63) / 64) * (size_t)(cap / 4) * 1024; }
enum { VL_NONE = 0, VL_WRITE = 1, VL_USE = 2 };
constexpr size_t NB_REC = 64 * 16;   // 64 rows x one 16-byte record
__host__ __device__ inline size_t nb_list_bytes(size_t rows, int cap) { return ((rows + 63) / 64) * (size_t)cap * NB_REC; }
enum { MREAL_CELLS = 0, MREAL_BUILD_LIST = 1, MREAL_USE_LIST = 2 };
// Lanczos sums fused into the pair-list mat-vec (see k_lz_update): with x = vec and y = M x the kernel also leaves the
// per-block partial sums of x.x, x.y and x.x_{j-1}
struct LzFuse {
    const double4 *vprev;   // x_{j-1}, unnormalised (null for j = 0)
    double *partials;       // [LZ_NGRAM][npart_cap]
    int npart_cap;
    const double4 *q, *p, *u;   // two-step blocks of a team (sums = 2, 3): v_j, v_{j-1} (nullable), M v_{j-1} (nullable)
};
// slots of the Gram sums of a two-step block (k_lz_block): products of q = v_j, p = v_{j-1}, u = M p, w1 = M q, w2 = M w1
// -- the eight that M = M^T leaves independent: q.w2 = w1.w1, u.w1 = p.w2, q.u = p.w1 (the pair list holds every pair twice with
// bit-identical coefficients, so the products agree to summation rounding), and u.u is carried over from the block that made u
enum { LZG_QW1 = 0, LZG_W1W1, LZG_W1W2, LZG_W2W2, LZG_PW1, LZG_PW2, LZG_UW2, LZG_QQ, LZ_NGRAM };
// out = M_real . vec (+ self). mode: cells only / cells + write the pair list / use the pair list
// rows [lo, hi) of the mat-vec (the whole vector is read; multi-GPU ranks each take a row range)
void launch_mreal(const double4 *pos_s, const float4 *posf_s, const double4 *vec_s, double4 *out_s, RowMap rows,
                  const int *cell_off, DBox box, DCells nc, double rcut, double self, const double *coef, int ncoef, NbList nb,
                  int mode, hipStream_t s, const double4 *vec2_s = nullptr, double4 *out2_s = nullptr,   // BUILD_LIST: a second vector rides along
                  VerletList vl = VerletList{}, int vl_mode = VL_NONE,   // VL_WRITE: the cell pass also writes the neighbour list; VL_USE: no cell walk
                  const double2 *pv = nullptr,                           // packed (position, vec_s) records for the drain's gathers
                  double *sums0 = nullptr, int sums0_cap = 0, double *scal = nullptr,    // ... and the sums vec2.vec2, vec2.out2 are left in scal[LZ_TMP ..] (Lanczos iteration 0)
                  Gate gate = Gate{},                                    // cell pass / kept-list pass without a pair list: run on one outcome of the device-side list check only
                  DevRowArgs dr = DevRowArgs{});                         // owned-particle ranks: the rows come from device memory (cell passes only; stage_hi takes out2)
bool mreal_table_in_lds(int ncoef);   // the neighbour list across steps needs the LDS copy of the table
// pair-list mat-vec + Lanczos sums; leaves the three reduced sums in scal[LZ_TMP .. LZ_TMP + 2]
void launch_mreal_lanczos(const double4 *pos_s, const double4 *vec_s, double4 *w, RowMap rows, const int *cell_off,
                          DBox box, DCells nc, double rcut, double self, const double *coef, NbList nb, LzFuse lz,
                          double *scal, hipEvent_t ev_begin, hipEvent_t ev_end, hipStream_t s,   // events (nullable) bracket the mat-vec kernel
                          VerletList vl = VerletList{},   // in use this step: rows that overflowed the pair list walk it instead of the cells
                          int sums = 1,                   // 0 none, 1 one-step Lanczos sums, 2 Gram sums of a two-step block, 3 single step of that driver
                          const int *stop = nullptr,      // nullable: leave at once if *stop != 0 (the device-side Lanczos decision)
                          DevRowArgs dr = DevRowArgs{},   // owned-particle ranks: rows from device memory
                          const void *vec_q = nullptr,    // (sums == 1) the 16-byte mirror of vec_s (launch_lz_update's xq): the neighbours' rows in ONE gather per pair
                          bool no_reduce = false);        // the mat-vec kernel only: no reduction of its partial sums behind it
int mreal_partials_needed(int rows);
void launch_lz_reduce3(const double *partials, int npart, int cap, double *scal, hipStream_t s);

// ---- far field (K2-K8) -----------------------------------------------------------------------------------
// scratch of the fast far-field path (rebuilt every call): particles binned by the 8^3 block of nodes their support
// origin lies in, and their records written in bin order
struct FarBins {
    int nbx, nby, nbz;          // bins per axis (set by the launchers from the grid)
    int *cnt, *off;             // [nbins], [nbins + 1]
    int *rank_s;                // [N] rank of a particle inside its bin (-1: not needed by this slab rank)
    void *tmp; size_t tmp_bytes;   // scan scratch
};
size_t bin_scan_temp_bytes(size_t nbins);
struct FarRec;                  // 64-byte bin-ordered particle record (pse_farfield.hip)
struct SpreadWork {
    FarBins fb;
    FarRec *rec_t;              // [N] bin order: origin, sorted index (bit 31: owned by another slab rank), offset, prefac * force
    int force_tz, force_nw;     // tuning switches of the handle (0: automatic): z depth of a spread block, waves per block
    int force_bz;               // ... bins along z per gather workgroup (PSE_GATHER_BZ=1|2)
    CellRanges need;            // a slab rank: rows outside hold no particle data (their support cannot reach the slab)
    const int *cell_off;
    int rows_local;             // 1: N counts the rows of ONE slab rank (owned-particle teams), not the particles of the suspension
};
// per-step constants of the separable Gaussian weights: step ratios r_t = exp(-c h^2 (2t+1)) (y with the (1 + xy^2) of the
// sheared lattice), ln K = -2 c xy hx hy, the tilt
constexpr int FAR_PMAX = 14;   // largest support of the block far field (error 1e-6 gives P = 13)
struct GaussConsts { double rx[FAR_PMAX - 1], ry[FAR_PMAX - 1], rz[FAR_PMAX - 1], lnk, s; };
size_t farfield_bins(const DGrid &G);
// true if the caller must zero the grids first (atomic fallback: P outside 4..8 or a grid smaller than two tiles)
bool spread_needs_zero(const DGrid &G);
// The far field of a step: (1) the pass that gathers the particles into cell order (launch_permute) ranks every particle inside
// its bin (FarBinArgs; the bin counts are zeroed by the caller before it); (2) launch_far_records: bin offsets + the 64-byte
// records in bin order; (3) launch_spread; ... (4) launch_gather, which reads the same records.
struct FarBinArgs { bool on; DGrid G; FarBins fb; };
FarBinArgs far_bin_args(const DGrid &G, const SpreadWork &w);
size_t far_bin_count(const DGrid &G);   // ints of fb.cnt to zero
hipError_t launch_far_records(const double4 *pos_s, const double4 *f_s, int N, DGrid G, DBox box, SpreadWork w, hipStream_t s);
hipError_t launch_spread(const double4 *pos_s, const double4 *f_s, int N, double *gx, double *gy, double *gz, DGrid G,
                   DBox box, SpreadWork w, hipStream_t s);
struct ScaleArgs {
    double xi, eta;
    int noise;               // add k-space Brownian noise (K6)
    double noise_fac;        // sqrt(2 kT / (dt h^3))
    uint32_t seed, timestep;
    const uint32_t *ts_off;  // nullable: the noise is drawn at timestep + *ts_off, read on the device (pse_set_timestep_offset)
    int transposed;          // layout [Nx][ny_local][Nzh] (slab mode, after the transpose) instead of [nx_local][Ny][Nzh]
    int y0, nyl;             // slab of y rows in transposed layout
    int runtime_plan;        // PSE_XMIX=1: the runtime radix plan also where a compile-time one exists (A/B)
    int wide_small;          // PSE_XFFT_SMALL_KB=8: eight kz columns per workgroup also on small grids (A/B)
};
void launch_scale(double2 *X, double2 *Y, double2 *Z, DGrid G, DBox box, ScaleArgs a, hipStream_t s);
// debug: kx, ky, kz, w sinc^2, sqrt(w) sinc of n nodes (i, j, k)
void launch_debug_kop(const int *ijk, int n, DGrid G, DBox box, double xi, double eta, double *out, hipStream_t s);
// fused forward-x FFT + scale + inverse-x FFT on [3][Nx][Ny][Nzh] (Nx a power of two, 16..512); tw[m] = exp(-2 pi i m/Nx)
bool xfuse_supported(int Nx);
// own in-place y transforms of the half spectra [3 nxl][Ny][Nzp] (Ny = 2^a 3^b 5^c, not a power of two); tw[m] = exp(-2 pi i m / Ny);
// kb: consecutive kz per workgroup (2, 4 or 8)
bool yfft_supported(int Ny);
bool yfft_regs_supported(int Ny, int Nz, bool own_z = false);   // the register y pass (k_yfft_regs) instead of rocFFT's strided pass: 256 (Nz <= 256), and 256 / 512 with the own z pass
// a slab rank's y transforms with the reordering of the all-to-all blocks folded in: forward reads the planes [3][nxl][Ny][Nzp] and writes
// [3][G][nxl][nyl][Nzp]; inverse the other way round (any Ny = 2^a 3^b 5^c in 16..512: yfft_possible)
bool yfft_possible(int Ny);
void launch_yfft_slab(double2 *cgrid, double2 *blocks, DGrid G, int nyl, bool inverse, const double2 *tw, hipStream_t s,
                      bool regs = true);   // regs: Ny = 256, 512 by the register pass (k_yfft_regs with the block map) instead of k_fft_cols
void launch_yfft(double2 *spectra, DGrid G, bool inverse, const double2 *tw, hipStream_t s, int kb = 4);
// own z pass (k_zfft_rows): `rows` real rows of Nz doubles per component <-> spectrum rows of Nzp complex numbers; tw[m] = exp(-2 pi i m / Nz)
bool zfft_supported(int Nz);
void launch_zfft(double *const real[3], double2 *const spec[3], int rows, int Nz, int Nzp, bool inverse, const double2 *tw, hipStream_t s);
void launch_xfft_scale(double2 *X, double2 *Y, double2 *Z, DGrid G, DBox box, ScaleArgs a, const double2 *tw, hipStream_t s);
// the fast path reads the records written by launch_spread of the same step
hipError_t launch_gather(const double4 *pos_s, SpreadWork w, int N, const double *gx, const double *gy, const double *gz, DGrid G,
                   DBox box, double4 *u_s, hipStream_t s);

// ---- slab decomposition helpers
void launch_slab_pack(double2 *cgrid, double2 *buf, int nxl, int Ny, int Nzh, int nyl, int unpack, hipStream_t s);
void launch_add_inplace(double *a, const double *b, size_t n, hipStream_t s);
// in-process teams: the device copies that stand for one exchange, as ONE launch (a message transport posts one group too)
struct CopyList { int n; const double *src[40]; double *dst[40]; unsigned cnt[40]; };   // counts in doubles
void launch_copy_list(const CopyList &l, hipStream_t s);

// ---- vector kernels (K10-K14) ----------------------------------------------------------------------------
void launch_psi(double4 *psi_s, const unsigned *tag_s, int N, uint32_t seed, uint32_t timestep, hipStream_t s,
                CellRanges need = CellRanges{}, const int *cell_off = nullptr);
// scal layout (device doubles): [0..127] alpha, [128..255] beta, [256] psi norm, [257] scratch
constexpr int LZ_ALPHA = 0, LZ_BETA = 128, LZ_NORM = 256, LZ_TMP = 257, LZ_UU = 272, LZ_NSCAL = 400;   // TMP: 3 sums (one-step) or the LZ_NGRAM sums of a two-step block; UU[j] = |M v_{j-1}|^2 of the block that starts at j
constexpr int LZ_NPART = 1024;  // partial-sum slots
// out_s = scale * sum_q t[q] X[q], X[0] = x0, X[q] = V[q] (the UNNORMALISED Lanczos vectors: t carries the 1 / |x_q|)
// rows [lo, hi)
struct BasisCoef { double t[104]; };   // the m <= 100 coefficients of the final combination, passed by value
// sink.on: the result is not stored in out_s but added to add_a + add_b (nullable) of the row and sent on: vel[tag].xyz (single GPU) or
// rows[i] = (sum, tag) (a team's all-gathered array)
struct CombineSink { bool on; const double4 *add_a, *add_b; const unsigned *tag_s; double4 *vel; double4 *rows; };
void launch_basis_combine(const double4 *x0, const double4 *V, size_t stride, const BasisCoef &t, int m, const double *scal,
                          double scale, int use_norm, double4 *out_s, int lo, int hi, hipStream_t s, CombineSink sink = CombineSink{},
                          const struct LzState *st = nullptr);   // st: m and the coefficients come from the device-side decision
// Lanczos iteration on the own rows [lo, hi) (see k_lz_update in pse_kernels.hip).  dots: sums of x.x, x.y, x.vprev ->
// scal[LZ_TMP..+2] (y = null: x.x only), for mat-vecs that did not fuse them; [all-reduce by the caller when sharded];
// update: alpha_j, beta_j -> scal, x_{j+1} -> xnext
void launch_lz_dots(const double4 *x, const double4 *y, const double4 *vprev, int lo, int hi, double *partials, int cap,
                    double *scal, hipStream_t s);
// one block of the two-iterations-per-exchange Lanczos of a team (k_lz_block in pse_kernels.hip)
struct LzBlockArgs {
    const double4 *q, *p, *w1, *w2;   // p null for j = 0; w2 null: a single step (scalars alpha_j, beta_{j+1} only)
    double4 *u;                       // in: M p (not read for j = 0); out: M v_{j+1} (two steps) or M v_j = w1 (one step)
    double4 *v1, *v2;                 // V[j + 1], V[j + 2]
    int j;
};
// sums_all / nranks (teams): [nranks][LZ_NGRAM] partial sums of all ranks, added in rank order by the kernel; nranks = 0: scal[LZ_TMP ..]
// sch: the host's copy of the scalars (device pointer of mapped pinned memory, nullable): alpha, beta, the norm are written there too
void launch_lz_block(const LzBlockArgs &a, bool full, double *scal, const int (*row_ranges)[2], int n_ranges, hipStream_t s,
                     const double *sums_all = nullptr, int nranks = 0, double *sch = nullptr,
                     const RowRanges *rg_dev = nullptr, int rows_cap = 0,   // owned-particle ranks: the ranges come from device memory (vectors_off: scalars only)
                     bool vectors_off = false, const int *stop = nullptr);
void launch_vq_roundtrip(const double *in, double *out, int n, hipStream_t s);   // debug: vq_unpack(vq_pack(row)) of n rows of three doubles
void launch_lz_update(const double4 *xin, const double4 *y, const double4 *xprev, double4 *xnext, int j,
                      double *scal, const int (*row_ranges)[2], int n_ranges, hipStream_t s,   // up to three disjoint row ranges in one launch
                      const double *sums_all = nullptr, int nranks = 0, double *sch = nullptr, const int *stop = nullptr,
                      void *xq = nullptr);   // nullable: [rows] 16-byte mirror of xnext (vq_pack, pse_device.h), written on the same rows
// ---- the convergence decision of the Lanczos iteration ON THE DEVICE (queue-only Brownian calls: pse_set_async) ----------------
// Replaces, for calls that may not wait for the host, what the host driver does between batches of iterations (the reference's
// LAPACKE_spteqr + host loops, PSEv1/Brownian.cu:540-582, and its step-norm test, PSEv1/Brownian.cu:673-724): one workgroup solves
// the tridiagonal eigenproblems of the sizes to be checked (one wavefront each, implicit QL as tridiag_eigen in pse_params.cpp),
// walks m upward exactly as the host does and, once the step norm is below the tolerance, closes the gate `done` that every later
// kernel of the iteration reads, and leaves the coefficients of the final combination for k_basis_combine.
struct LzState {
    int done;            // the iteration has ended: m_final and coef are valid; later iteration kernels leave at once
    int m_final;
    int checked;         // t_prev holds T^{1/2} e_1 of this size (0: nothing yet)
    int status;          // 0 converged, 1 the queued iterations ran out first (result from the last size checked), 2 non-finite coefficient / eigen-solve failure
    double stepnorm;
    double t_prev[104];
    double coef[104];    // coefficients of the basis vectors in the final combination (t divided by the norms the basis carries)
};
struct LzDecide {
    int m_lo, m_hi;      // sizes to check, at most two (m_lo = m_hi - 1 or m_hi)
    int done_iters;      // iterations whose alpha exists; one-step driver: beta[done_iters] does not exist yet
    int have_last_beta;  // 1: beta[done_iters] exists as well (two-step driver of a team)
    int pending_beta;    // > 0: first test beta[pending_beta] for breakdown (it arrived with this batch)
    int first;           // first decision of a call: reset the state
    int last;            // last queued decision: if still open, end the iteration at the last size checked (status 1)
    int normalised;      // the basis holds normalised v_q for q >= 1 (two-step driver); else unnormalised x_q (divide by beta_q)
    int m_max;
    double tol;
};
constexpr int LZ_HOST_M = 380, LZ_HOST_STEPNORM = 381, LZ_HOST_STATUS = 382, LZ_HOST_SEQ = 383, LZ_HOST_OPEN = 384;   // (OPEN: calls so far that ended with status != 0)   // slots of the host mirror (sch) the decision writes
void launch_lz_decide(const LzDecide &d, double *scal, LzState *st, double *sch, double seq, hipStream_t s);
bool lz_decide_supported(int m_hi);   // the eigenvectors of both sizes fit the LDS of one workgroup
// tag_s (nullable): out[i].w = the particle's index in the caller's arrays, so the rows can be scattered by ranks that did not sort them
void launch_sum_rows(const double4 *a, const double4 *b, const double4 *c, double4 *out, int lo, int hi, hipStream_t s,
                     const unsigned *tag_s = nullptr);
void launch_pick(const int *cell_off, const int *idx, int n, int *out, hipStream_t s);
// vel[tag].xyz = a + b + c (each may be null), keep w;  tag_s = null: the tag travels in a[s].w (launch_sum_rows)
void launch_scatter_sum(const double4 *a, const double4 *b, const double4 *c, const unsigned *tag_s, int N,
                        double4 *vel, hipStream_t s);
void launch_integrate(double4 *pos, const double4 *vel, double3 *accel, int3 *image, const double4 *force,
                      const unsigned *group, int N, DBox box, double dt, double shear_rate, hipStream_t s);
void launch_eval_fg(const double *r, int n, const double *coef, double *f, double *g, hipStream_t s);
// soft pair repulsion from the cell list, scattered to the caller's order (force provider)
void launch_pair_repulsion(const double4 *pos_s, const unsigned *tag_s, int N, const int *cell_off, DBox box, DCells nc,
                           double k, double sigma, int accumulate, double4 *force, hipStream_t s);

}  // namespace pse
