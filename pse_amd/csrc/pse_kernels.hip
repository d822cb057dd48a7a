// HIP kernels of the PSE engine for gfx950 (CDNA4, wave64).  Each kernel names the reference kernel it
// replaces.  All arithmetic is fp64.
#include "pse_kernels.h"
#include "pse_dft.h"
#include "pse_farbin.h"

#include <hipcub/hipcub.hpp>
#include <atomic>

namespace pse {

constexpr int TPB = 256;
constexpr double TWO_PI = 6.283185307179586476925286766559;

static inline int nblocks(long n, int tpb) { return (int)((n + tpb - 1) / tpb); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: every launcher keeps, per device ordinal, the largest size it has
// raised the attribute to (a process-wide flag would leave a second device, or a second thread's first launch, without it).  Racing
// threads at worst set the same attribute twice.
struct LdsAttr {
    std::atomic<size_t> have[32];
    bool need(size_t lds) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 32) return true;
        if (have[dev].load(std::memory_order_relaxed) >= lds) return false;
        have[dev].store(lds, std::memory_order_relaxed);
        return true;
    }
};

// ------------------------------------------------------------------------------------------------ reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over a 256-thread block; result valid in all threads
__device__ __forceinline__ double block_sum(double v, double *sh /* >= 4 doubles */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
// every block reduces the same partial array in the same order -> identical value everywhere
__device__ __forceinline__ double reduce_partials(const double *partials, int n, double *sh) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += TPB) v += partials[i];
    return block_sum(v, sh);
}

// ------------------------------------------------------------------------------------------------ binning
// (replaces HOOMD CellListGPU used through NeighborListGPUBinned, PSEv1/integrate.py:58-83)
// Counting sort by cell (rocprim's sort_pairs runs ten merge passes at N = 1e6: 0.15 ms).  Arrival ranks from atomics are
// not reproducible, and every slab rank must end up with the SAME order (rows are exchanged by position), so a last
// pass orders each cell by original index: the result equals a stable sort by key.  cell_off[c] = first slot of cell c
// (c = 0..ncell): cell c owns [cell_off[c], cell_off[c+1]) and any run of consecutive cells is one contiguous slot range.
constexpr unsigned KEY_FOREIGN = 0xFFFFFFFFu;   // a particle a slab rank counts but does not order
// (Counting the foreign particles per x layer in an LDS histogram instead -- one global add per layer and workgroup, booked on
// the layer's first cell -- was slower: 58 us against 45 us; the kernel is not bound by its global atomics.)
__global__ void k_cell_keys(const double4 *__restrict__ pos, const unsigned *__restrict__ group, int N, DBox box,
                            DCells nc, unsigned *__restrict__ keys, unsigned *__restrict__ rank, int *__restrict__ cnt, CellRanges need,
                            SlabBook sb, Gate gate) {
    if (gate.closed()) return;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = g < N;
    unsigned key = 0;
    bool mine = false;
    if (live) {
        const unsigned idx = group ? group[g] : (unsigned)g;
        const double4 p = pos[idx];
        double fx, fy, fz;
        frac_coords(box, p.x, p.y, p.z, fx, fy, fz);
        const int cx = cell_coord(fx, nc.nx), cy = cell_coord(fy, nc.ny), cz = cell_coord(fz, nc.nz);
        const int zb = cz / nc.bz;
        key = (unsigned)cell_slot(nc, cx, cy, zb, cz - zb * nc.bz);
        mine = need.cell((int)key);
        if (mine) {
            keys[g] = key;
            rank[g] = (unsigned)atomicAdd(&cnt[key], 1);
        } else {
            keys[g] = KEY_FOREIGN;
            if (sb.n == 0) atomicAdd(&cnt[key], 1);   // counted (the row offsets are global), not ranked
        }
    }
    if (sb.n > 0) {
        // A slab rank needs the row offsets of its own and its ghost cells and of every slab boundary -- not of the cells in
        // between.  The particles of another rank's cells are therefore counted per SLAB, on the first cell of that slab the rank
        // does not keep (all of them lie before or after the kept layers of that slab, so every offset that is used comes out
        // right), one atomic per distinct slab of a wavefront: device-scope atomics run at ~20 G/s however they are addressed
        // (MI355X_MICROARCH.md), and 80 % of the particles are foreign to a rank of eight.
        const int slab = live && !mine ? (int)(key / (unsigned)sb.cells_per_slab) : -1;
        unsigned long long todo = __ballot(slab >= 0);
        const int lane = threadIdx.x & 63;
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            const int s0 = __shfl(slab, src, 64);
            const unsigned long long m = __ballot(slab == s0) & todo;
            // spread over `spread` cells of that layer by workgroup: one word takes ~88 atomics per microsecond, and a step has
            // 15 000 wavefronts x 7 foreign slabs (all on ONE cell per slab the kernel took 195 us instead of 45)
            if (lane == src) atomicAdd(&cnt[sb.book[s0] + (int)((blockIdx.x * 4u + (threadIdx.x >> 6)) % (unsigned)sb.spread)], __popcll(m));
            todo &= ~m;
        }
    }
}
__global__ void k_cell_scatter(const unsigned *__restrict__ keys, const unsigned *__restrict__ rank,
                               const int *__restrict__ cell_off, int N, unsigned *__restrict__ slots, Gate gate) {
    if (gate.closed()) return;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < N && keys[g] != KEY_FOREIGN) slots[cell_off[keys[g]] + rank[g]] = (unsigned)g;
}
// one thread per slot: rank its particle among the members of its cell (a handful: the loads of a cell's threads are the same
// few words), write it in order.  (One wave per cell kept five of 64 lanes busy: 33 us at N = 1e6; this way 10.)
__global__ void __launch_bounds__(TPB)
k_cell_order(const int *__restrict__ cell_off, const unsigned *__restrict__ keys, int N, const unsigned *__restrict__ slots,
             unsigned *__restrict__ perm, CellRanges need, Gate gate) {
    if (gate.closed()) return;
    const int s = blockIdx.x * TPB + threadIdx.x;
    if (s >= N || !need.row(s, cell_off)) return;
    const unsigned v = slots[s];
    const int c = (int)keys[v], a = cell_off[c], n = cell_off[c + 1] - a;
    int smaller = 0;
    for (int t = 0; t < n; ++t) smaller += slots[a + t] < v ? 1 : 0;
    perm[a + smaller] = v;
}
size_t cell_sort_temp_bytes(size_t ncell) {
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const int *)nullptr, (int *)nullptr, (int)(ncell + 1));
    return bytes;
}
hipError_t launch_cell_scan(const int *cnt, int *out, int n, void *tmp, size_t tmp_bytes, hipStream_t s) {
    return hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, cnt, out, n, s);
}
hipError_t cell_sort(const double4 *pos, const unsigned *group, int N, DBox box, DCells nc, unsigned *keys, unsigned *rank,
                     unsigned *slots, int *cnt, int ncell, void *tmp, size_t tmp_bytes, int *cell_off, unsigned *perm, hipStream_t s,
                     CellRanges need, SlabBook sb, bool cnt_is_zero, Gate gate) {
    hipError_t e = cnt_is_zero ? hipSuccess : hipMemsetAsync(cnt, 0, (size_t)(ncell + 1) * sizeof(int), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_cell_keys, dim3(nblocks(N, TPB)), dim3(TPB), 0, s, pos, group, N, box, nc, keys, rank, cnt, need, sb, gate);
    // cnt[ncell] = 0: cell_off[ncell] = N.  A failed scan (scratch too small for ncell) would leave garbage offsets: reported
    e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, cnt, cell_off, ncell + 1, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_cell_scatter, dim3(nblocks(N, TPB)), dim3(TPB), 0, s, keys, rank, cell_off, N, slots, gate);
    hipLaunchKernelGGL(k_cell_order, dim3(nblocks(N, TPB)), dim3(TPB), 0, s, cell_off, keys, N, slots, perm, need, gate);
    return hipGetLastError();
}

// Gather into cell order: wrapped sorted positions (+ the float copy and the packed records of the near field), the vector, the tags
// -- and, since round 4, what used to be two more launches over the same rows: the rank of every particle inside its far-field bin
// (far.on; was k_support) and the particle noise of the step (psi_s; K14 gpu_stokes_BrownianGenerate_kernel, PSEv1/Brownian.cu:99-130).
__global__ void __launch_bounds__(TPB)
k_permute(const double4 *__restrict__ pos, const double4 *__restrict__ vec,
                          const unsigned *__restrict__ group, const unsigned *__restrict__ perm, int N, DBox box,
                          double4 *__restrict__ pos_s, float4 *__restrict__ posf_s, double2 *__restrict__ pv,
                          double4 *__restrict__ vec_s, unsigned *__restrict__ tag_s, const double4 *__restrict__ pos_build,
                          double half_skin2, int *__restrict__ flags, CellRanges need, const int *__restrict__ cell_off,
                          FarBinArgs far, double4 *__restrict__ psi_s, uint32_t seed, uint32_t timestep, Gate gate,
                          const uint32_t *__restrict__ ts_off) {
    if (gate.closed()) return;
    if (ts_off) timestep += *ts_off;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s - (int)(threadIdx.x & 63) >= N) return;            // whole wave past the end
    const bool live = s < N && need.row(s, cell_off);        // a slab rank holds particle data for its own and its ghost rows only
    bool bin_need = false;
    int bin = -1;
    if (live) {
        const unsigned g = perm[s];
        const unsigned idx = group ? group[g] : g;
        const double4 p = pos[idx];
        // wrap into the primary cell (keeps sheared images consistent: y images shift x by xy*Ly)
        double fx, fy, fz;
        frac_coords(box, p.x, p.y, p.z, fx, fy, fz);
        const double y = (fy - 0.5) * box.Ly;
        double4 q;
        q.x = (fx - 0.5) * box.Lx + box.xy * y;
        q.y = y;
        q.z = (fz - 0.5) * box.Lz;
        q.w = 0.0;
        pos_s[s] = q;
        posf_s[s] = make_float4((float)q.x, (float)q.y, (float)q.z, 0.0f);   // single-precision copy for the cutoff pre-filter
        if (pv) {   // packed 48-byte (position, vector) records of the near-field passes' drain: the position half
            pv[3 * (size_t)s] = make_double2(q.x, q.y);
            ((double *)&pv[3 * (size_t)s + 1])[0] = q.z;
        }
        tag_s[s] = idx;
        if (vec) {
            double4 v = vec[idx];
            v.w = 0.0;
            vec_s[s] = v;
            if (pv) {   // the vector half: the neighbour-list pass gathers (position, force) as one record
                ((double *)&pv[3 * (size_t)s + 1])[1] = v.x;
                pv[3 * (size_t)s + 2] = make_double2(v.y, v.z);
            }
        }
        if (pos_build) {   // distance check of the kept neighbour list
            const double4 b = pos_build[s];
            double dx = q.x - b.x, dy = q.y - b.y, dz = q.z - b.z;
            min_image(box, dx, dy, dz);
            if (dx * dx + dy * dy + dz * dz > half_skin2) flags[0] = 1;
        }
        if (psi_s) {       // K14, keyed by the particle's global index
            uint32_t r[4];
            philox4x32(idx, 0u, timestep, DOMAIN_PARTICLE, seed, PHILOX_KEY1, r);
            const double c = 1.7320508075688772;  // sqrt(3): variance 1
            psi_s[s] = make_double4(uniform_pm(r[0], c), uniform_pm(r[1], c), uniform_pm(r[2], c), 0.0);
        }
        if (far.on) {      // the far-field bin of the particle, from the fractional coordinates of the STORED position (as k_far_records
                           // recomputes them: both must name the same bin)
            double gx, gy, gz;
            frac_coords(box, q.x, q.y, q.z, gx, gy, gz);
            int4 o;
            double4 d;
            far_support(gx, gy, gz, far.G, o, d);
            bin_need = true;
            if (far.G.nxl < far.G.Nx) bin_need = wrapi(o.w - (far.G.x0 - far.G.P), far.G.Nx) < far.G.nxl + 2 * far.G.P;   // within a support of the slab's planes
            bin = bin_index(o.x, o.y, o.z, far.fb);
        }
    }
    if (far.on) {
        const int rk = far_bin_rank(bin_need, bin, far.fb.cnt);
        if (s < N) far.fb.rank_s[s] = rk;
    }
}

__global__ void k_permute_vec(const double4 *__restrict__ vec, const unsigned *__restrict__ tag_s, int N,
                              double4 *__restrict__ vec_s) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    double4 v = vec[tag_s[s]];
    v.w = 0.0;
    vec_s[s] = v;
}

void launch_permute(const double4 *pos, const double4 *vec, const unsigned *group, const unsigned *perm, int N, DBox box,
                    double4 *pos_s, float4 *posf_s, double2 *pv, double4 *vec_s, unsigned *tag_s, hipStream_t s,
                    const double4 *pos_build, double half_skin2, int *flags, CellRanges need, const int *cell_off,
                    const FarBinArgs *far, double4 *psi_s, uint32_t seed, uint32_t timestep, Gate gate, const uint32_t *ts_off) {
    hipLaunchKernelGGL(k_permute, dim3(nblocks(N, TPB)), dim3(TPB), 0, s, pos, vec, group, perm, N, box, pos_s, posf_s, pv, vec_s, tag_s,
                       pos_build, half_skin2, flags, need, cell_off, far ? *far : FarBinArgs{}, psi_s, seed, timestep, gate, ts_off);
}
__global__ void k_gate_decide(int *__restrict__ flags, int *__restrict__ word) {
    const int f = flags[0] | flags[1];
    *word = f;
    if (f) flags[1] = 0;   // the build that follows sets it again if a row overflows
}
void launch_gate_decide(int *flags, int *word, hipStream_t s) { hipLaunchKernelGGL(k_gate_decide, dim3(1), dim3(1), 0, s, flags, word); }
__global__ void k_gate_zero(Gate g, int *__restrict__ a, size_t na, int *__restrict__ b, size_t nb) {
    if (g.closed()) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += (size_t)gridDim.x * blockDim.x) {
        if (i < na) a[i] = 0; else b[i - na] = 0;
    }
}
void launch_gate_zero(Gate g, int *a, size_t na, int *b, size_t nb, hipStream_t s) {
    hipLaunchKernelGGL(k_gate_zero, dim3(std::max(1, std::min(1024, nblocks((long)(na + nb), TPB)))), dim3(TPB), 0, s, g, a, na, b, nb);
}
__global__ void k_gate_copy(Gate g, double4 *__restrict__ dst, const double4 *__restrict__ src, size_t n) {
    if (g.closed()) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void launch_gate_copy(Gate g, double4 *dst, const double4 *src, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_gate_copy, dim3(std::max(1, std::min(4096, nblocks((long)n, TPB)))), dim3(TPB), 0, s, g, dst, src, n);
}
void launch_permute_vec(const double4 *vec, const unsigned *tag_s, int N, double4 *vec_s, hipStream_t s) {
    hipLaunchKernelGGL(k_permute_vec, dim3(nblocks(N, TPB)), dim3(TPB), 0, s, vec, tag_s, N, vec_s);
}

// ------------------------------------------------------------------------------------------------ near field
// K9 gpu_stokes_Mreal_kernel (PSEv1/Mobility.cu:594-687): u_i = self F_i + sum_j [f (I - rr) + g rr] F_j over
// minimum-image neighbours with r < rcut.  The neighbours come from the build's own cell list (HOOMD's
// NeighborListGPUBinned is not available, PSEv1/integrate.py:58-83).
//
// Two phases per thread so the wave does not diverge on the ~15 % of candidates that pass the cutoff: (1) scan the
// 27 neighbour cells (z-runs merged) and push the slots that pass r < rcut into a per-thread LDS queue -- cheap, no
// transcendental work; (2) drain the queue densely: every lane evaluates f, g for a real neighbour.  With LIST the
// drained pairs (j, f, (g-f)/r^2, r) are also written to a per-step ELL pair list that the Lanczos mat-vecs reuse
// (positions do not change inside a step, PSEv1/Brownian.cu:473-521 recomputes them every iteration).
// 44 entries per thread: 44 KB of queue + the 8 KB table of the metric point let THREE workgroups share a CU (48: two)
constexpr int QCAP = 44;
constexpr unsigned JMASK = (1u << 27) - 1;

// Four slots of the 64 rows of a wave form one 4096-byte group of [slot][lane] 16-byte records: a mat-vec reads a group with four
// 16-byte loads per lane (a coalesced dword or dwordx2 load occupies the texture addresser as long as a dwordx4 load,
// tools/microbench/stream_widths: 8 / 16 / 16 clk -- the twelve narrow loads of the round-2 (entry | f | h) planes cost three times that).
//
// What a record holds (round 6): the neighbour's row and the pair's term of the mat-vec, f v + h (d.v) d with d = x_i - x_j (minimum
// image) and h = (g - f) / r^2, as FOUR ROUNDED NUMBERS: fr = f and s = d sqrt|h| (the sign of h in a bit: f v + sgn (s.v) s) -- fr a
// signed 26-bit integer in units of 2^-24 (|f| < 2), s three signed 22-bit mantissas under the exponent es of its largest component
// (value = m 2^(es - 21)): absolute errors <= 3e-8 (fr) and 2^-22 |s|_max, what single precision gives for numbers of order one, in 100
// bits instead of 128 (rounds 6a: four floats + the entry = 20 bytes per pair; the mat-vec is bound by this stream).  The mat-vecs then
// gather ONE thing per pair -- the neighbour's vector row -- instead of the 48-byte (position, vector) record, and neither subtract
// positions nor look up image shifts; the list is only ever read by the Lanczos iteration of M_real^{1/2} psi (tolerance `error`,
// 1e-3 by default; the deterministic M.F of the pass that builds the list stays fp64 -- the reference's whole path is fp32,
// SURVEY.md 2.4-1).  EVERY application of the operator inside a Lanczos iteration uses these rounded numbers -- the vector that
// rides along with the build pass, the list mat-vecs, the rows that did not fit the list -- so the iteration sees ONE symmetric
// matrix: d_ji = -d_ij exactly and rounding to nearest is odd, so both directions of a pair round alike.  (The tests' CPU restatement
// of the algorithm repeats this rounding operation by operation: include/pse_amd.h, the precision note.)
struct PairCoef { int f, mx, my, mz, es; unsigned neg; };
__device__ __forceinline__ PairCoef pair_coef(double f, double h, double dx, double dy, double dz) {
    const double hs = (double)sqrtf((float)fabs(h));
    const double sx = dx * hs, sy = dy * hs, sz = dz * hs;
    const double m = fmax(fabs(sx), fmax(fabs(sy), fabs(sz)));
    int e = 0;
    (void)frexp(m, &e);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    const double sc = __hiloint2double((21 - e + 1023) << 20, 0);   // 2^(21 - e): |component| sc < 2^21
    const double lim = 2097151.0;                                   // 2^21 - 1
    PairCoef p;
    p.mx = (int)fmax(-lim, fmin(lim, rint(sx * sc))); p.my = (int)fmax(-lim, fmin(lim, rint(sy * sc))); p.mz = (int)fmax(-lim, fmin(lim, rint(sz * sc)));
    p.es = e;
    p.f = (int)fmax(-33554431.0, fmin(33554431.0, rint(f * 16777216.0)));
    p.neg = h < 0.0 ? 1u : 0u;
    return p;
}
__device__ __forceinline__ void pair_apply(const PairCoef &p, double vx, double vy, double vz, double &ux, double &uy, double &uz) {
    const double sc = __hiloint2double((p.es - 21 + 1023) << 20, 0);
    const double fr = (double)p.f * 5.9604644775390625e-08, sx = (double)p.mx * sc, sy = (double)p.my * sc, sz = (double)p.mz * sc;   // (2^-24)
    double sd = sx * vx + sy * vy + sz * vz;
    if (p.neg) sd = -sd;
    ux += fr * vx + sd * sx; uy += fr * vy + sd * sy; uz += fr * vz + sd * sz;
}
// the 16-byte record: x = row (27) | neg (1) | fr[3:0] (4); y = fr[25:4] (22) | mx[9:0] (10); z = mx[21:10] (12) | my[19:0] (20);
// w = my[21:20] (2) | mz (22) | es + 128 (8)
typedef unsigned nb4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ nb4 nb_pack(unsigned j, const PairCoef &p) {
    const unsigned f = (unsigned)p.f & 0x3FFFFFFu, mx = (unsigned)p.mx & 0x3FFFFFu, my = (unsigned)p.my & 0x3FFFFFu, mz = (unsigned)p.mz & 0x3FFFFFu;
    nb4 w;
    w.x = j | (p.neg ? (1u << 27) : 0u) | (f << 28);
    w.y = (f >> 4) | (mx << 22);
    w.z = (mx >> 10) | (my << 12);
    w.w = (my >> 20) | (mz << 2) | ((unsigned)(p.es + 128) << 24);
    return w;
}
__device__ __forceinline__ PairCoef nb_unpack(nb4 w, bool ok) {
    PairCoef p;
    p.neg = w.x & (1u << 27);
    p.f = ok ? (int)((((w.y & 0x3FFFFFu) << 4) | (w.x >> 28)) << 6) >> 6 : 0;
    p.mx = ok ? (int)((((w.z & 0xFFFu) << 10) | (w.y >> 22)) << 10) >> 10 : 0;
    p.my = ok ? (int)((((w.w & 0x3u) << 20) | (w.z >> 12)) << 10) >> 10 : 0;
    p.mz = ok ? (int)(((w.w >> 2) & 0x3FFFFFu) << 10) >> 10 : 0;
    p.es = ok ? (int)(w.w >> 24) - 128 : 0;
    return p;
}
template <bool STREAM = false>
__device__ __forceinline__ void nb_store(char *rec, int slot, int lane, unsigned j, const PairCoef &p) {
    char *r = rec + (size_t)(slot >> 2) * (4 * NB_REC);
    const nb4 w = nb_pack(j, p);
    if (STREAM) __builtin_nontemporal_store(w, (nb4 *)r + (slot & 3) * 64 + lane);
    else ((nb4 *)r)[(slot & 3) * 64 + lane] = w;
}
constexpr unsigned NB_NEG = 1u << 27;   // entry = neighbour row | NB_NEG if h < 0

__device__ __forceinline__ void vl_store(char *vrec, int slot, int lane, unsigned j) {
    ((unsigned *)(vrec + (size_t)(slot >> 2) * 1024))[lane * 4 + (slot & 3)] = j;
}

// CL: the f, g coefficient table is copied to LDS first (ncoef doubles of dynamic shared memory).  Every neighbour reads
// 20 coefficients of its own interval: from global memory that was 20 of the 24 L1 accesses per pair and kept the
// texture addresser busy for the whole kernel (rocprofv3: TA_BUSY = duration, 425 M cache accesses = 20 x 21.3 M pairs).
// TWO: a second vector rides along (out2 = M_real vec2): the pass that builds the pair list for M.F also delivers
// M.psi, the first Lanczos mat-vec, for one more gather per neighbour.
// VL: the pass also writes the neighbour list kept across steps -- every pair closer than vl.rskin = rcut + skin (the
// pre-filter and the queue work with that radius; f, g and the pair list still stop at rcut).
// DEV: an owned-particle rank -- the rows come from device memory (DevRowArgs); a separate instantiation, so that the single-GPU
// pass keeps its 166 registers (with the few extra scalars it spilled)
// PK: the drain reads a neighbour's position AND vector from the packed 48-byte record pv[j] (three 16-byte gathers from one or two
// lines instead of four from two arrays): the pass is bound by the drain's round trips through the texture addresser, not by its scan.
template <bool LIST, bool CL, bool TWO, bool VL, bool DEV = false, bool PK = false>
__global__ void __launch_bounds__(TPB, ((LIST && CL && !VL) ? 3 : 1))   // the list-building pass of every step: <= 168 VGPRs (it takes 166)
k_mreal_cells(const double4 *__restrict__ pos_s, const float4 *__restrict__ posf_s, const double4 *__restrict__ vec_s,
              double4 *__restrict__ out_s, RowMap rm_arg, const int *__restrict__ cell_off, DBox box, DCells nc, double rcut2,
              float rcut2_pre, double self, const double *__restrict__ coef_g, int ncoef, NbList nb,
              const double4 *__restrict__ vec2_s, double4 *__restrict__ out2_s, VerletList vl,
              double *__restrict__ sums0, int sums0_cap, Gate gate, DevRowArgs dr, const double2 *__restrict__ pvin) {
    if (gate.closed()) return;
    const RowMapRegs rm(rm_arg, DEV ? dr.rm : nullptr);
    if (!DEV) nc.xpad = 0;   // (only owned-particle ranks pad their x layers: a constant here, the term folds away)
    int nb_live = gridDim.x;
    if (DEV) {   // an owned-particle rank: the rows of this pass are known on the device only; the launch covers the capacity
        nb_live = (rm.list_rows() + TPB - 1) / TPB;
        if ((int)blockIdx.x >= nb_live) return;
    }
    // the pass that also writes the kept neighbour list queues every pair within rcut + r_buff (28 per row instead of 21): a deeper
    // queue, or half of the waves would stop for an extra, poorly filled drain in the middle of the walk
    constexpr int QC = VL ? 64 : QCAP;
    __shared__ unsigned queue[QC * TPB];
    extern __shared__ double scoef[];
    const int tid = threadIdx.x;
    if (CL) {
        // intervals padded from 20 to 21 doubles: lanes in different intervals then read different banks (unpadded, interval k
        // and k + 8 collide: two thirds of this kernel's LDS cycles were bank conflicts)
        for (int q = tid; q < ncoef; q += TPB) scoef[(q / (2 * RS_NCOEF)) * (2 * RS_NCOEF + 1) + q % (2 * RS_NCOEF)] = coef_g[q];
        __syncthreads();
    }
    const double *coef = CL ? scoef : coef_g;
    const int lr = xcd_block(blockIdx.x, nb_live) * TPB + tid;       // list row -> sorted row (own rows, then ghost layers)
    const int i_own = rm.row(lr);
    // a padding lane repeats the first row (identical stores, into the same places or into padding list rows) instead of leaving:
    // the wave-level sums at the end then see every lane
    const bool padding = i_own < 0;
    if (padding && !(TWO && sums0)) return;
    const int i = padding ? rm.first_row() : i_own;
    const double4 pi = pos_s[i];
    const double4 vi = vec_s[i];
    double ux = self * vi.x, uy = self * vi.y, uz = self * vi.z;
    double wx = 0.0, wy = 0.0, wz = 0.0;
    if (TWO) { const double4 v2 = vec2_s[i]; wx = self * v2.x; wy = self * v2.y; wz = self * v2.z; }
    double fx, fy, fz;
    frac_coords(box, pi.x, pi.y, pi.z, fx, fy, fz);
    const int cx = cell_coord(fx, nc.nx), cy = cell_coord(fy, nc.ny), cz = cell_coord(fz, nc.nz);
    const bool shift_only = nc.nx > 1 && nc.ny > 1 && nc.nz > 1;   // otherwise finish with the rint minimum image
    int qn = 0, total = 0, vtotal = 0;
    const int lane = lr & 63;
    char *rec = LIST ? nb.data + (size_t)(lr >> 6) * nb.cap * NB_REC : nullptr;
    char *vrec = VL ? (char *)vl.idx + (size_t)(lr >> 6) * (vl.cap / 4) * 1024 : nullptr;
    const double rq2 = VL ? vl.rskin * vl.rskin : rcut2;   // what enters the queue

    // DU queue entries per iteration: their load -> distance -> table -> force chains are independent (one entry at a time the phase
    // was latency-bound: 57 % of its wave-cycles were s_waitcnt; three or four buy nothing more).  Written as loops over small arrays:
    // the compiler then keeps the pass at 166 VGPRs (the hand-interleaved form it replaces took 200), which together with the
    // 44-entry queue is what lets a third workgroup onto the CU: 0.64 -> 0.56 ms.
    auto drain = [&]() {
        constexpr int DU = 2;
        for (int q = 0; q < qn; q += DU) {
            unsigned e[DU]; int j[DU]; bool live[DU];
            double4 p[DU], F[DU], G[DU];
#pragma unroll
            for (int u = 0; u < DU; ++u) {
                live[u] = q + u < qn;
                e[u] = queue[(live[u] ? q + u : q) * TPB + tid];
                j[u] = (int)(e[u] & JMASK);
            }
#pragma unroll
            for (int u = 0; u < DU; ++u) {
                if (PK) {
                    const double2 *r = pvin + 3 * (size_t)j[u];
                    const double2 a = r[0], b = r[1], c = r[2];
                    p[u] = make_double4(a.x, a.y, b.x, 0.0); F[u] = make_double4(b.y, c.x, c.y, 0.0);
                } else { p[u] = pos_s[j[u]]; F[u] = vec_s[j[u]]; }
                if (TWO) G[u] = vec2_s[j[u]];
            }
#pragma unroll
            for (int u = 0; u < DU; ++u) {
                double sx, sy, sz;
                image_shift(e[u] >> 27, box, sx, sy, sz);
                double dx = pi.x - p[u].x - sx, dy = pi.y - p[u].y - sy, dz = pi.z - p[u].z - sz;
                if (!shift_only) min_image(box, dx, dy, dz);
                const double r2 = dx * dx + dy * dy + dz * dz;
                if (VL) {
                    if (live[u] && r2 < rq2 && r2 > 0.0) { if (vtotal < vl.cap) vl_store(vrec, vtotal, lane, (unsigned)j[u]); ++vtotal; }
                }
                const double t = VL ? fmin(r2, rcut2) : r2;   // the table ends at rcut
                double f, h;
                if (CL) eval_fg<2 * RS_NCOEF + 1>(t, coef, f, h);
                else eval_fg(t, coef, f, h);
                const bool in = live[u] && r2 < rcut2 && r2 > 0.0;   // the fp64 cutoff decides
                if (!in) { f = 0.0; h = 0.0; }
                // A pass that writes the pair list serves a Lanczos iteration: the vector of that iteration (vec2 when F rides in
                // front, else vec itself) sees the rounded pair coefficients the list carries (pair_coef, nb_store); F never does.
                if (!LIST || TWO) {
                    const double rd = (dx * F[u].x + dy * F[u].y + dz * F[u].z) * h;
                    ux += f * F[u].x + rd * dx; uy += f * F[u].y + rd * dy; uz += f * F[u].z + rd * dz;
                }
                if (LIST) {
                    const PairCoef pc = pair_coef(f, h, dx, dy, dz);
                    if (TWO) pair_apply(pc, G[u].x, G[u].y, G[u].z, wx, wy, wz);
                    else pair_apply(pc, F[u].x, F[u].y, F[u].z, ux, uy, uz);
                    if (in) {   // 16 B per pair: row | sign | fr | s
                        if (total < nb.cap) nb_store<true>(rec, total, lane, (unsigned)j[u], pc);
                        ++total;
                    }
                } else if (TWO) {
                    const double sd = (dx * G[u].x + dy * G[u].y + dz * G[u].z) * h;
                    wx += f * G[u].x + sd * dx; wy += f * G[u].y + sd * dy; wz += f * G[u].z + sd * dz;
                }
            }
        }
        qn = 0;
    };

    // Scan in single precision on a float copy of the positions against a cutoff enlarged by the rounding bound, eight
    // candidates per round trip (a float4 is half the registers of a double4: the scan is bound by dependent round
    // trips at three waves per SIMD); the drain repeats the test in double precision, so the neighbour set is exactly
    // the fp64 one.
    constexpr int SU = 8;
    if (!shift_only) {   // fewer than three cells along some axis: minimum-image search in double precision
        for_each_run(nc, cell_off, cx, cy, cz, [&](int jb, int je, unsigned code) {
            for (int j = jb; j < je; ++j) {
                const double4 pj = pos_s[j];
                double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                min_image(box, dx, dy, dz);
                const double r2 = dx * dx + dy * dy + dz * dz;
                if (r2 < rq2 && j != i && r2 > 0.0) {
                    queue[qn * TPB + tid] = (unsigned)j | (code << 27);
                    ++qn;
                }
                if (__any(qn > QC - SU)) drain();
            }
        });
    } else {
        // Every lane walks ITS runs at its own pace, SU candidates of one run per round (the bounds of the run after the current
        // one are already loaded when the current one ends).  Stepping through "run k of all 64 lanes" together cost the longest
        // run k of the wave nine times over: 288 load instructions per wave for 137 candidates per lane; this way ~200.
        // the runs of this lane, one at a time (the order of for_each_run): up to three per (x, y) column.  Written out in place --
        // one site, no closure: as a lambda called from two places the iterator state went to scratch memory.
        const int zb0 = cz / nc.bz, zi0 = cz - zb0 * nc.bz;
        int ox = -1, oy = -2, rk = 0, nparts = 0;
        int f0 = 0, l0 = 0, f1 = 0, l1 = 0, f2 = 0, l2 = 0;
        unsigned c0 = 13u, c1 = 13u, c2 = 13u;
        int j = 0, je = 0, jn = 0, jen = 0;
        unsigned code = 13u, code_n = 13u;
        float qx = 0.f, qy = 0.f, qz = 0.f;
        bool have_n = true;
        int pend = 2;   // first round: prime the prefetch, then make the first run current
        while (true) {
            do {
                if (j >= je && have_n) {   // the current run is used up: the prefetched one becomes current (an empty one is skipped next round)
                    j = jn; je = jen; code = code_n;
                    double sx, sy, sz;
                    image_shift(code, box, sx, sy, sz);
                    qx = (float)(pi.x - sx); qy = (float)(pi.y - sy); qz = (float)(pi.z - sz);
                    if (rk >= nparts) {    // next (x, y) column
                        if (++oy > 1) { oy = -1; ++ox; }
                        if (ox > 1) have_n = false;
                        else {
                            int ax = cx + ox, wx = 0, ay = cy + oy, wy = 0;
                            if (ax < 0) { ax += nc.nx; wx = -1; } else if (ax >= nc.nx) { ax -= nc.nx; wx = 1; }
                            if (ay < 0) { ay += nc.ny; wy = -1; } else if (ay >= nc.ny) { ay -= nc.ny; wy = 1; }
                            const unsigned cxy = (unsigned)((wx + 1) * 9 + (wy + 1) * 3);
                            int s0, s1, s2; unsigned d0, d1, d2;
                            z_neighbour_slot(nc, ax, ay, cz, zb0, zi0, -1, cxy, s0, d0);
                            z_neighbour_slot(nc, ax, ay, cz, zb0, zi0, 0, cxy, s1, d1);
                            z_neighbour_slot(nc, ax, ay, cz, zb0, zi0, 1, cxy, s2, d2);
                            const bool m01 = s1 == s0 + 1 && d1 == d0, m12 = s2 == s1 + 1 && d2 == d1;
                            f0 = s0; c0 = d0;
                            if (m01 && m12) { l0 = s2; nparts = 1; }
                            else if (m01) { l0 = s1; f1 = l1 = s2; c1 = d2; nparts = 2; }
                            else if (m12) { l0 = s0; f1 = s1; l1 = s2; c1 = d1; nparts = 2; }
                            else { l0 = s0; f1 = l1 = s1; c1 = d1; f2 = l2 = s2; c2 = d2; nparts = 3; }
                            rk = 0;
                        }
                    }
                    if (have_n) {
                        const int sa = rk == 0 ? f0 : (rk == 1 ? f1 : f2), sb = (rk == 0 ? l0 : (rk == 1 ? l1 : l2)) + 1;
                        code_n = rk == 0 ? c0 : (rk == 1 ? c1 : c2);
                        ++rk;
                        jn = cell_off[sa]; jen = cell_off[sb];   // consumed at the next switch
                    }
                }
            } while (--pend > 0);
            pend = 1;
            const bool work = j < je;
            if (!__any(work || have_n)) break;
            float4 pj[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) pj[u] = posf_s[work ? min(j + u, je - 1) : i];   // independent loads in flight
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int jj = j + u;
                const float dx = qx - pj[u].x, dy = qy - pj[u].y, dz = qz - pj[u].z;
                const float r2 = dx * dx + dy * dy + dz * dz;
                if (jj < je && r2 < rcut2_pre && jj != i) {
                    queue[qn * TPB + tid] = (unsigned)jj | (code << 27);
                    ++qn;
                }
            }
            j += SU;
            // drain together: a lane-private "queue full" branch would serialise the wave once per lane
            if (__any(qn > QC - SU)) drain();
        }
    }
    drain();
    out_s[i] = make_double4(ux, uy, uz, 0.0);
    if (TWO) {
        out2_s[i] = make_double4(wx, wy, wz, 0.0);
        if (DEV && dr.stage_hi) {   // rows of the last layers: parked for the exchange with the right neighbour
            const int lb = dr.lr->last_begin, no = dr.lr->n_own;
            if (i >= lb && i < no && i - lb < dr.stage_cap) dr.stage_hi[i - lb] = make_double4(wx, wy, wz, 0.0);   // (k_local_scatter refuses a step whose last layers exceed the capacity; the bound is the belt to its braces)
        }
        if (sums0) {    // the sums of Lanczos iteration 0 (x = psi, y = M psi): x.x and x.y per wavefront (was k_lz_dots: one more pass over both)
            const double4 v2 = vec2_s[i];
            const double a = padding ? 0.0 : v2.x * v2.x + v2.y * v2.y + v2.z * v2.z, b = padding ? 0.0 : v2.x * wx + v2.y * wy + v2.z * wz;
            const double sa = wave_sum(a), sb = wave_sum(b);
            if ((tid & 63) == 0) {
                const int slot = blockIdx.x * (TPB / 64) + (tid >> 6);
                sums0[slot] = sa; sums0[sums0_cap + slot] = sb; sums0[2 * sums0_cap + slot] = 0.0;
            }
        }
    }
    if (LIST) {
        if (total > nb.cap) total = -1;   // the row did not fit: the mat-vecs of this step walk the cells for it
        nb.cnt[i] = total;
    }
    if (VL) {
        if (vtotal > vl.cap) { vtotal = -1; vl.flags[1] = 1; }
        vl.cnt[i] = vtotal;
    }
}

// The near-field pass of a step that REUSES the neighbour list: no cell walk -- every row reads its entries (groups of four,
// one 16-byte load), gathers the neighbours' (position, vec) records (+ vec2), takes the minimum image itself (particles may
// have crossed the periodic boundary since the build) and evaluates f, g densely; pairs beyond rcut (about a quarter at
// skin = 0.4) contribute zero.  LIST: writes the step's pair list for the Lanczos mat-vecs, exactly as the cell pass does.
// PK: pv holds (position, vec_s) records (three 16-byte gathers per neighbour instead of four).
template <bool LIST, bool TWO, bool PK>
__global__ void __launch_bounds__(TPB)
k_mreal_verlet(const double4 *__restrict__ pos_s, const double2 *__restrict__ pv, const double4 *__restrict__ vec_s,
               double4 *__restrict__ out_s, int N, DBox box, double rcut2, double self, const double *__restrict__ coef_g, int ncoef,
               NbList nb, VerletList vl, const double4 *__restrict__ vec2_s, double4 *__restrict__ out2_s, Gate gate) {
    if (gate.closed()) return;
    extern __shared__ double scoef[];
    const int tid = threadIdx.x;
    for (int q = tid; q < ncoef; q += TPB) scoef[(q / (2 * RS_NCOEF)) * (2 * RS_NCOEF + 1) + q % (2 * RS_NCOEF)] = coef_g[q];
    __syncthreads();
    const int i = xcd_block(blockIdx.x, gridDim.x) * TPB + tid;
    if (i >= N) return;
    const double4 pi = pos_s[i];
    const double4 vi = vec_s[i];
    double ux = self * vi.x, uy = self * vi.y, uz = self * vi.z;
    double wx = 0.0, wy = 0.0, wz = 0.0;
    if (TWO) { const double4 v2 = vec2_s[i]; wx = self * v2.x; wy = self * v2.y; wz = self * v2.z; }
    const int lane = i & 63, cnt = vl.cnt[i];
    const char *vrec = (const char *)vl.idx + (size_t)(i >> 6) * (vl.cap / 4) * 1024;
    char *rec = LIST ? nb.data + (size_t)(i >> 6) * nb.cap * NB_REC : nullptr;
    int total = 0;
    constexpr int U = 4;
    for (int s0 = 0; s0 < cnt; s0 += U) {
        const uint4 e4 = ((const uint4 *)(vrec + (size_t)(s0 >> 2) * 1024))[lane];
        unsigned e[U] = {e4.x, e4.y, e4.z, e4.w};
        double2 a[U], b[U], c[U];
        double4 G[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u && s0 + u >= cnt) e[u] = e[0];          // slots past the row's count were never written
            if (PK) {
                const double2 *r = pv + 3 * (size_t)e[u];
                a[u] = r[0]; b[u] = r[1]; c[u] = r[2];
            } else {
                const double4 pj = pos_s[e[u]], Fj = vec_s[e[u]];
                a[u] = make_double2(pj.x, pj.y); b[u] = make_double2(pj.z, Fj.x); c[u] = make_double2(Fj.y, Fj.z);
            }
            if (TWO) G[u] = vec2_s[e[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double dx = pi.x - a[u].x, dy = pi.y - a[u].y, dz = pi.z - b[u].x;
            const double ny = rint(dy * box.iLy);
            dy -= ny * box.Ly; dx -= ny * box.xy * box.Ly;
            const double nx = rint(dx * box.iLx), nz = rint(dz * box.iLz);
            dx -= nx * box.Lx; dz -= nz * box.Lz;
            const double r2 = dx * dx + dy * dy + dz * dz;
            const bool in = s0 + u < cnt && r2 < rcut2 && r2 > 0.0;
            double f, h;
            eval_fg<2 * RS_NCOEF + 1>(fmin(r2, rcut2), scoef, f, h);
            if (!in) { f = 0.0; h = 0.0; }
            const double Fx = b[u].y, Fy = c[u].x, Fz = c[u].y;
            if (!LIST || TWO) {
                const double rd = (dx * Fx + dy * Fy + dz * Fz) * h;
                ux += f * Fx + rd * dx; uy += f * Fy + rd * dy; uz += f * Fz + rd * dz;
            }
            if (LIST) {   // (as the cell pass: the vector of the Lanczos iteration sees the coefficients the list carries)
                const PairCoef pc = pair_coef(f, h, dx, dy, dz);
                if (TWO) pair_apply(pc, G[u].x, G[u].y, G[u].z, wx, wy, wz);
                else pair_apply(pc, Fx, Fy, Fz, ux, uy, uz);
                if (in) {
                    if (total < nb.cap) nb_store(rec, total, lane, e[u], pc);
                    ++total;
                }
            } else if (TWO) {
                const double sd = (dx * G[u].x + dy * G[u].y + dz * G[u].z) * h;
                wx += f * G[u].x + sd * dx; wy += f * G[u].y + sd * dy; wz += f * G[u].z + sd * dz;
            }
        }
    }
    out_s[i] = make_double4(ux, uy, uz, 0.0);
    if (TWO) out2_s[i] = make_double4(wx, wy, wz, 0.0);
    if (LIST) nb.cnt[i] = total > nb.cap ? -1 : total;   // -1: the mat-vecs of this step walk the neighbour list for this row
}

// eval_fg with the Horner steps as a rolled loop (the same operations in the same order: bit-identical results): for the rare rows of
// the pair-list mat-vec that did not fit the list -- unrolled, its twenty coefficient loads in flight set the register count of the
// whole kernel (126), which the list loop itself does not need
__device__ __forceinline__ void eval_fg_lean(double r2, const double *__restrict__ coef, double &f, double &gmf_r2) {
    const double ir = rsqrt(r2), r = r2 * ir, ir2 = ir * ir;
    double f0, g0;
    if (r > 2.0) { const double ir3 = ir * ir2; f0 = 0.75 * ir + 0.5 * ir3; g0 = 1.5 * ir - ir3; }
    else { f0 = 1.0 - 0.28125 * r; g0 = 1.0 - 0.1875 * r; }
    const double s = r * RS_PER_UNIT;
    const int k = (int)s;
    const double t = 2.0 * (s - k) - 1.0;
    const double *c = coef + (size_t)k * (2 * RS_NCOEF);
    double fw = c[RS_DEG], gw = c[RS_NCOEF + RS_DEG];
#pragma unroll 1
    for (int q = RS_DEG - 1; q >= 0; --q) {
        fw = fma(fw, t, c[q]);
        gw = fma(gw, t, c[RS_NCOEF + q]);
    }
    f = f0 - fw;
    gmf_r2 = ((g0 - gw) - f) * ir2;
}

// mat-vec from the pair list (one thread per particle, ELL layout: slot-major so a wave reads contiguous rows).
// Per pair 16 B of list from HBM -- row | sign | fr | s, see nb_pack -- plus the neighbour's vector row gathered through L2 (24 bytes
// of doubles, or 16 from its mirror: VQ): no positions, no image shifts.  FUSE adds the Lanczos epilogue (see LzFuse).
// WSP = 4: the four waves of a workgroup share ONE block of 64 rows and take every fourth group of slots each (partial sums
// through LDS): the waves resident on a CU then gather from a quarter as many neighbourhoods (the kernel is bound by L1 misses).
// FUSE: 0 none; 1 the three sums of the one-step iteration; 2 the Gram sums of a two-step block (this launch is its SECOND
// mat-vec: vec = w1 = M q, result w2); 3 the sums of a single step in the two-step driver (vec = q, result w1).
// VQ: the neighbours' rows come from the 16-byte mirror of the vector (vq_pack, pse_device.h): ONE gather per pair instead of two.
template <int FUSE, int UNROLL, int NT, int WSP = 1, bool VQ = false>
__global__ void __launch_bounds__(NT, (WSP == 4 ? 5 : 1))   // the split kernel: <= 96 VGPRs, five workgroups per CU (it is bound by the round trips its waves have in flight)
k_mreal_list(const double4 *__restrict__ pos_s, const double4 *__restrict__ vec_s, double4 *__restrict__ out_s, RowMap rm_arg,
             DBox box, double self, NbList nb, LzFuse lz,
             const int *__restrict__ cell_off, DCells nc, double rcut2, const double *__restrict__ coef, VerletList vl,
             const int *__restrict__ stop, DevRowArgs dr, const vq4 *__restrict__ vec_q = nullptr) {
    if (stop && *stop) return;   // the Lanczos iteration has ended (device-side decision)
    const RowMapRegs rm(rm_arg, dr.rm);
    int nb_live = gridDim.x;
    if (dr.rm) {   // an owned-particle rank: rows from device memory; workgroups past the last one still own a slot of the partial sums
        nb_live = (rm.list_rows() + (WSP > 1 ? 64 : NT) - 1) / (WSP > 1 ? 64 : NT);
        if ((int)blockIdx.x >= nb_live) {
            if (FUSE && threadIdx.x == 0)
                for (int t = 0; t < (FUSE >= 2 ? (int)LZ_NGRAM : 3); ++t) lz.partials[(size_t)t * lz.npart_cap + blockIdx.x] = 0.0;
            return;
        }
    }
    __shared__ double sh[4];
    static_assert(WSP == 1 || NT == 64 * WSP, "split rows: one wave per slot phase");
    const int wv = WSP > 1 ? (int)(threadIdx.x >> 6) : 0;
    const int lr = WSP > 1 ? xcd_block(blockIdx.x, nb_live) * 64 + (int)(threadIdx.x & 63)
                           : xcd_block(blockIdx.x, nb_live) * NT + (int)threadIdx.x;
    const int i = rm.row(lr);
    const bool active = i >= 0;
    double ux = 0.0, uy = 0.0, uz = 0.0;
    double4 vi = make_double4(0.0, 0.0, 0.0, 0.0);
    if (active) {
        vi = vec_s[i];
        const int cnt = nb.cnt[i];
        if (cnt >= 0) {
            if (wv == 0) { ux = self * vi.x; uy = self * vi.y; uz = self * vi.z; }
            const int lane = lr & 63;
            const char *rec = nb.data + (size_t)(lr >> 6) * nb.cap * NB_REC;
            // software-pipelined by hand: the list entries of UNROLL slots, then their gathers, then the arithmetic -- left to the
            // compiler every slot waited for its own load -> gather chain (the kernel was latency-bound at 1.9 TB/s).  Branch-free:
            // slots past cnt re-read the last valid one with zero coefficients.
            static_assert(UNROLL == 4, "one list group per iteration");
            for (int s0 = UNROLL * wv; s0 < cnt; s0 += UNROLL * WSP) {
                const char *grp = rec + (size_t)(s0 >> 2) * (4 * NB_REC);
                // streamed once: non-temporal, so the list does not evict the neighbour rows the gathers reuse from L1
                unsigned e[UNROLL];
                nb4 c[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    c[u] = __builtin_nontemporal_load((const nb4 *)grp + u * 64 + lane);
                    e[u] = c[u].x;
                    if (u && s0 + u >= cnt) e[u] = e[0];          // slots past the row's count were never written
                }
                double2 vxy[UNROLL];
                double vz[UNROLL];
                vq4 vq[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    if (VQ) { vq[u] = vec_q[e[u] & JMASK]; continue; }   // the row in 16 bytes: one gather
                    const double4 *vj = vec_s + (e[u] & JMASK);   // 24 of the row's 32 bytes: a 16- and an 8-byte gather from one line
                    vxy[u] = *reinterpret_cast<const double2 *>(vj);
                    vz[u] = vj->z;
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const bool ok = s0 + u < cnt;
                    const PairCoef pc = nb_unpack(c[u], ok);
                    if (VQ) { vq_unpack(vq[u], vxy[u].x, vxy[u].y, vz[u]); }
                    pair_apply(pc, vxy[u].x, vxy[u].y, vz[u], ux, uy, uz);
                }
            }
        } else if (wv != 0) {
            // overflow rows are computed whole by the first wave
        } else if (vl.idx) {
            // the row did not fit the pair list and the step runs on the kept neighbour list (the cells are those of its build)
            const double4 pi = pos_s[i];
            ux = self * vi.x; uy = self * vi.y; uz = self * vi.z;
            const int lane = i & 63, vc = vl.cnt[i];
            const char *vrec = (const char *)vl.idx + (size_t)(i >> 6) * (vl.cap / 4) * 1024;
            for (int sl = 0; sl < vc; ++sl) {
                const unsigned j = ((const unsigned *)(vrec + (size_t)(sl >> 2) * 1024))[lane * 4 + (sl & 3)];
                const double4 pj = pos_s[j];
                double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                min_image(box, dx, dy, dz);
                const double r2 = dx * dx + dy * dy + dz * dz;
                if (r2 < rcut2 && r2 > 0.0) {
                    double f, h;
                    eval_fg_lean(r2, coef, f, h);
                    const double4 Fj = vec_s[j];
                    pair_apply(pair_coef(f, h, dx, dy, dz), Fj.x, Fj.y, Fj.z, ux, uy, uz);   // (the coefficients the list would have carried)
                }
            }
        } else {
            // the row did not fit the list (dense cluster): walk the cells, as the pass that built the list did
            const double4 pi = pos_s[i];
            ux = self * vi.x; uy = self * vi.y; uz = self * vi.z;
            double fx, fy, fz;
            frac_coords(box, pi.x, pi.y, pi.z, fx, fy, fz);
            const int cx = cell_coord(fx, nc.nx), cy = cell_coord(fy, nc.ny), cz = cell_coord(fz, nc.nz);
            for_each_run(nc, cell_off, cx, cy, cz, [&](int jb, int je, unsigned) {
                for (int j = jb; j < je; ++j) {
                    const double4 pj = pos_s[j];
                    double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                    min_image(box, dx, dy, dz);
                    const double r2 = dx * dx + dy * dy + dz * dz;
                    if (r2 < rcut2 && j != i && r2 > 0.0) {
                        double f, h;
                        eval_fg_lean(r2, coef, f, h);
                        const double4 Fj = vec_s[j];
                        pair_apply(pair_coef(f, h, dx, dy, dz), Fj.x, Fj.y, Fj.z, ux, uy, uz);
                    }
                }
            });
        }
    }
    if (WSP > 1) {   // the other waves' shares of the rows
        __shared__ double red[WSP > 1 ? (WSP - 1) * 3 * 64 : 1];
        const int ln = threadIdx.x & 63;
        if (wv > 0) { red[((wv - 1) * 3 + 0) * 64 + ln] = ux; red[((wv - 1) * 3 + 1) * 64 + ln] = uy; red[((wv - 1) * 3 + 2) * 64 + ln] = uz; }
        __syncthreads();
        if (wv > 0) return;
        for (int w = 0; w < WSP - 1; ++w) { ux += red[(w * 3 + 0) * 64 + ln]; uy += red[(w * 3 + 1) * 64 + ln]; uz += red[(w * 3 + 2) * 64 + ln]; }
    }
    if (FUSE >= 2) {   // two-step driver: the sums the block scalars are derived from (see k_lz_block), slots LZG_*
        double g[LZ_NGRAM];
#pragma unroll
        for (int t = 0; t < LZ_NGRAM; ++t) g[t] = 0.0;
        if (active) {
            auto dot = [](const double4 &x, double yx, double yy, double yz) { return x.x * yx + x.y * yy + x.z * yz; };
            const double4 z4 = make_double4(0.0, 0.0, 0.0, 0.0);
            const double4 p = lz.p ? lz.p[i] : z4;
            if (FUSE == 2) {   // vi = w1, (ux, uy, uz) = w2
                const double4 q = lz.q[i], u = lz.u ? lz.u[i] : z4;
                g[LZG_QW1] = dot(q, vi.x, vi.y, vi.z); g[LZG_W1W1] = dot(vi, vi.x, vi.y, vi.z); g[LZG_W1W2] = dot(vi, ux, uy, uz);
                g[LZG_W2W2] = ux * ux + uy * uy + uz * uz; g[LZG_PW1] = dot(p, vi.x, vi.y, vi.z);
                g[LZG_PW2] = dot(p, ux, uy, uz); g[LZG_UW2] = dot(u, ux, uy, uz); g[LZG_QQ] = dot(q, q.x, q.y, q.z);
            } else {           // vi = q, (ux, uy, uz) = w1
                g[LZG_QW1] = dot(vi, ux, uy, uz); g[LZG_W1W1] = ux * ux + uy * uy + uz * uz; g[LZG_PW1] = dot(p, ux, uy, uz);
                g[LZG_QQ] = dot(vi, vi.x, vi.y, vi.z);
            }
        }
        static_assert(FUSE < 2 || NT == 64 || WSP > 1, "Gram sums: one wave per block of rows");
#pragma unroll
        for (int t = 0; t < LZ_NGRAM; ++t) {
            if (FUSE == 3 && t != LZG_QW1 && t != LZG_W1W1 && t != LZG_PW1 && t != LZG_QQ) { if (threadIdx.x == 0) lz.partials[(size_t)t * lz.npart_cap + blockIdx.x] = 0.0; continue; }   // (compile-time: the loop is unrolled)
            const double v = wave_sum(g[t]);
            if (threadIdx.x == 0) lz.partials[(size_t)t * lz.npart_cap + blockIdx.x] = v;
        }
    } else if (FUSE) {   // x = vec (unnormalised Lanczos vector), y = M x: partial sums of x.x, x.y, x.x_{j-1}
        double a = 0.0, b = 0.0, c = 0.0;
        if (active) {
            a = vi.x * vi.x + vi.y * vi.y + vi.z * vi.z;
            b = vi.x * ux + vi.y * uy + vi.z * uz;
            if (lz.vprev) {
                const double4 m = lz.vprev[i];
                c = vi.x * m.x + vi.y * m.y + vi.z * m.z;
            }
        }
        if (NT == 64 || WSP > 1) { a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); }
        else { a = block_sum(a, sh); __syncthreads(); b = block_sum(b, sh); __syncthreads(); c = block_sum(c, sh); }
        if (threadIdx.x == 0) {
            lz.partials[blockIdx.x] = a; lz.partials[lz.npart_cap + blockIdx.x] = b; lz.partials[2 * lz.npart_cap + blockIdx.x] = c;
        }
    }
    if (active) {
        out_s[i] = make_double4(ux, uy, uz, 0.0);
        if (dr.stage_hi) {   // rows of the last layers: parked for the exchange with the right neighbour
            const int lb = dr.lr->last_begin, no = dr.lr->n_own;
            if (i >= lb && i < no && i - lb < dr.stage_cap) dr.stage_hi[i - lb] = make_double4(ux, uy, uz, 0.0);
        }
    }
}

__global__ void __launch_bounds__(1024) k_lz_reduce(const double *__restrict__ partials, int npart, int cap, int nsum, double *__restrict__ scal, const int *__restrict__ stop = nullptr);
static size_t mreal_lds_bytes(int ncoef) { return (size_t)(ncoef / (2 * RS_NCOEF)) * (2 * RS_NCOEF + 1) * sizeof(double); }   // padded copy
bool mreal_table_in_lds(int ncoef) { return mreal_lds_bytes(ncoef) <= 14 * 1024; }   // with the 44 KB queue: three workgroups per CU up to 9 KB of table, two beyond

void launch_mreal(const double4 *pos_s, const float4 *posf_s, const double4 *vec_s, double4 *out_s, RowMap rm,
                  const int *cell_off, DBox box, DCells nc, double rcut, double self, const double *coef, int ncoef, NbList nb,
                  int mode, hipStream_t s, const double4 *vec2_s, double4 *out2_s, VerletList vl, int vl_mode, const double2 *pv,
                  double *sums0, int sums0_cap, double *scal, Gate gate, DevRowArgs dr) {
    const int rows = dr.rm ? dr.rows_cap : rm.list_rows();
    if (rows <= 0) return;
    const dim3 g(nblocks(rows, TPB)), b(TPB);
    const size_t cb = mreal_lds_bytes(ncoef);
    const bool cl = mreal_table_in_lds(ncoef);
    if (mode == MREAL_USE_LIST) {
        hipLaunchKernelGGL((k_mreal_list<0, 4, TPB>), g, b, 0, s, pos_s, vec_s, out_s, rm, box, self, nb, LzFuse{}, cell_off, nc, rcut * rcut, coef,
                           vl_mode == VL_USE ? vl : VerletList{}, nullptr, DevRowArgs{});
        return;
    }
    const bool list = mode == MREAL_BUILD_LIST, two = list && vec2_s != nullptr;
    if (vl_mode == VL_USE) {   // rows [0, N) of a single rank; the table is in LDS (the caller checked mreal_table_in_lds)
        const int hi = rm.hi[0];
#define PSE_VERLET(L, T) do { if (pv) hipLaunchKernelGGL((k_mreal_verlet<L, T, true>), g, b, cb, s, pos_s, pv, vec_s, out_s, hi, box, rcut * rcut, self, coef, ncoef, nb, vl, two ? vec2_s : nullptr, out2_s, gate); \
        else hipLaunchKernelGGL((k_mreal_verlet<L, T, false>), g, b, cb, s, pos_s, pv, vec_s, out_s, hi, box, rcut * rcut, self, coef, ncoef, nb, vl, two ? vec2_s : nullptr, out2_s, gate); } while (0)
        if (list && two) PSE_VERLET(true, true);
        else if (list) PSE_VERLET(true, false);
        else PSE_VERLET(false, false);
#undef PSE_VERLET
        return;
    }
    // pre-filter cutoff: coordinates (and image-shifted coordinates) are below 1.5 (Lx + |xy| Ly + Ly + Lz), rounded to
    // 2^-24 relative a few times on the way to a separation component
    const bool wr = vl_mode == VL_WRITE && cl;
    const double cmax = 1.5 * (box.Lx + std::fabs(box.xy) * box.Ly + box.Ly + box.Lz);
    const double rpre = (wr ? vl.rskin : rcut) + 16.0 * cmax * 5.97e-8;
    const float rcut2_pre = (float)(rpre * rpre * (1.0 + 1e-6));
#define PSE_CELLS(L, C, T, V) hipLaunchKernelGGL((k_mreal_cells<L, C, T, V>), g, b, (C) ? cb : 0, s, pos_s, posf_s, vec_s, out_s, rm, cell_off, box, nc, rcut * rcut, rcut2_pre, self, coef, ncoef, nb, two ? vec2_s : nullptr, out2_s, vl, (two && (C)) ? sums0 : nullptr, sums0_cap, gate, dr, nullptr)
#define PSE_CELLS_PK(L, T, D) hipLaunchKernelGGL((k_mreal_cells<L, true, T, false, D, true>), g, b, cb, s, pos_s, posf_s, vec_s, out_s, rm, cell_off, box, nc, rcut * rcut, rcut2_pre, self, coef, ncoef, nb, two ? vec2_s : nullptr, out2_s, vl, (two && !(D)) ? sums0 : nullptr, sums0_cap, gate, dr, pv)
    if (dr.rm) {   // owned-particle ranks (table in LDS, no kept list, packed records): the two passes of pse_team_step_local
        if (list && two) PSE_CELLS_PK(true, true, true); else PSE_CELLS_PK(false, false, true);
        return;
    }
    if (pv && cl && !wr && vl_mode != VL_USE && (two || !list)) {   // the hot passes of a single GPU: packed (position, vector) records
        if (list) PSE_CELLS_PK(true, true, false); else PSE_CELLS_PK(false, false, false);
    } else if (list) {
        if (cl && two) { if (wr) PSE_CELLS(true, true, true, true); else PSE_CELLS(true, true, true, false); }
        else if (cl) { if (wr) PSE_CELLS(true, true, false, true); else PSE_CELLS(true, true, false, false); }
        else PSE_CELLS(true, false, false, false);
    } else if (cl) { if (wr) PSE_CELLS(false, true, false, true); else PSE_CELLS(false, true, false, false); }
    else PSE_CELLS(false, false, false, false);
#undef PSE_CELLS_PK
#undef PSE_CELLS
    if (sums0 && list && two && cl)   // one partial per wavefront of the pass -> scal[LZ_TMP .. LZ_TMP + 2]
        hipLaunchKernelGGL(k_lz_reduce, dim3(3), dim3(1024), 0, s, sums0, (int)g.x * (TPB / 64), sums0_cap, 3, scal, nullptr);
}

int mreal_partials_needed(int rows) { return nblocks(std::max(rows, 1), 64); }   // one per 64 rows (the split mat-vec); the plain one uses a quarter
void launch_lz_reduce3(const double *partials, int npart, int cap, double *scal, hipStream_t s) {
    hipLaunchKernelGGL(k_lz_reduce, dim3(3), dim3(1024), 0, s, partials, npart, cap, 3, scal, nullptr);
}
void launch_mreal_lanczos(const double4 *pos_s, const double4 *vec_s, double4 *w, RowMap rm, const int *cell_off,
                          DBox box, DCells nc, double rcut, double self, const double *coef, NbList nb, LzFuse lz,
                          double *scal, hipEvent_t ev_begin, hipEvent_t ev_end, hipStream_t s, VerletList vl,
                          int sums, const int *stop, DevRowArgs dr, const void *vec_q, bool no_reduce) {
    const int rows = std::max(dr.rm ? dr.rows_cap : rm.list_rows(), 1);
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    // Four waves per block of 64 rows, each taking every fourth group of slots: 0.166 ms against 0.191 with four blocks of rows
    // per workgroup (two waves: 0.182, eight: 0.196).
    const int nb64 = nblocks(rows, 64);
#define PSE_LIST(F) hipLaunchKernelGGL((k_mreal_list<F, 4, 256, 4>), dim3(nb64), dim3(256), 0, s, pos_s, vec_s, w, rm, box, self, nb, lz, cell_off, nc, rcut * rcut, coef, vl, stop, dr)
    if (sums == 1 && vec_q)
        hipLaunchKernelGGL((k_mreal_list<1, 4, 256, 4, true>), dim3(nb64), dim3(256), 0, s, pos_s, vec_s, w, rm, box, self, nb, lz, cell_off, nc, rcut * rcut, coef, vl, stop, dr,
                           (const vq4 *)vec_q);
    else if (sums == 0) PSE_LIST(0); else if (sums == 1) PSE_LIST(1); else if (sums == 2) PSE_LIST(2); else PSE_LIST(3);
#undef PSE_LIST
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (no_reduce) return;   // (pse_debug_matvec_ms: the mat-vec kernel alone, back to back)
    if (sums == 1) hipLaunchKernelGGL(k_lz_reduce, dim3(3), dim3(1024), 0, s, lz.partials, nb64, lz.npart_cap, 3, scal, stop);
    else if (sums >= 2) hipLaunchKernelGGL(k_lz_reduce, dim3(LZ_NGRAM), dim3(1024), 0, s, lz.partials, nb64, lz.npart_cap, LZ_NGRAM, scal, stop);
}

__global__ void k_eval_fg(const double *__restrict__ r, int n, const double *__restrict__ coef, double *f, double *g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double ff, h;
    const double r2 = r[i] * r[i];
    eval_fg(r2, coef, ff, h);
    f[i] = ff;
    g[i] = ff + h * r2;
}
void launch_eval_fg(const double *r, int n, const double *coef, double *f, double *g, hipStream_t s) {
    hipLaunchKernelGGL(k_eval_fg, dim3(nblocks(n, TPB)), dim3(TPB), 0, s, r, n, coef, f, g);
}

// (far field: spread, gather and their particle records live in pse_farfield.hip)

// K1+K5+K6 fused: gpu_stokes_SetGridk_kernel (PSEv1/Helper.cu:285-332), gpu_stokes_Green_kernel
// (PSEv1/Mobility.cu:264-299) and gpu_stokes_BrownianGridGenerate_kernel (PSEv1/Brownian.cu:153-345) on the
// real-to-complex half spectrum.  Wave vectors are computed in registers (no gridk array).
struct KOp {
    double kx, ky, kz, ik2, B, c;  // 1 / k^2, B = w sinc^2 (deterministic), c = noise_fac sqrt(w) sinc
};
__device__ __forceinline__ KOp make_kop(int i, int j, int k, const DGrid &G, const DBox &box, double xi, double eta,
                                        double noise_fac) {
    KOp o;
    const int ki = i < (G.Nx + 1) / 2 ? i : i - G.Nx;     // FFT index folding, PSEv1/Helper.cu:307-312
    const int kj = j < (G.Ny + 1) / 2 ? j : j - G.Ny;
    const int kk = k < (G.Nz + 1) / 2 ? k : k - G.Nz;
    o.kx = TWO_PI * ki * box.iLx;
    o.ky = TWO_PI * (kj * box.iLy - box.xy * ki * box.iLx);   // sheared reciprocal lattice, Helper.cu:308
    o.kz = TWO_PI * kk * box.iLz;
    // one reciprocal square root serves 1 / k^2, |k| and 1 / |k|; the loop-invariant reciprocals are hoisted by the compiler
    const double k2 = o.kx * o.kx + o.ky * o.ky + o.kz * o.kz;
    const double rk = rsqrt(k2), kn = k2 * rk;
    o.ik2 = rk * rk;
    const double q = k2 * (1.0 / (4.0 * xi * xi));
    const double ing = 1.0 / ((double)G.Nx * (double)G.Ny * (double)G.Nz);
    const double w = 6.0 * M_PI * (1.0 + q) * exp_neg(-(1.0 - eta) * q) * (o.ik2 * ing);   // Helper.cu:326
    const double sinc = sin_lean(kn) * rk;                                            // Mobility.cu:290 (a = 1)
    o.B = w * sinc * sinc;
    o.c = noise_fac != 0.0 ? noise_fac * sqrt(w) * sinc : 0.0;                        // Brownian.cu:197,274-276
    return o;
}
// out += B (I - kk) f + c (I - kk) psi    (complex 3-vectors)
__device__ __forceinline__ void apply_kop(const KOp &o, const double2 f[3], const double2 psi[3], bool noise,
                                          double scale, double2 out[3]) {
    const double ik2 = o.ik2;
    double2 v[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        v[a].x = o.B * f[a].x;
        v[a].y = o.B * f[a].y;
        if (noise) { v[a].x += o.c * psi[a].x; v[a].y += o.c * psi[a].y; }
    }
    const double dr = (o.kx * v[0].x + o.ky * v[1].x + o.kz * v[2].x) * ik2;
    const double di = (o.kx * v[0].y + o.ky * v[1].y + o.kz * v[2].y) * ik2;
    const double kv[3] = {o.kx, o.ky, o.kz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        out[a].x += scale * (v[a].x - kv[a] * dr);
        out[a].y += scale * (v[a].y - kv[a] * di);
    }
}

// What K5 + K6 do to one node (i,j,k) of the half spectrum: f -> B (I - kk) f + c (I - kk) psi.
__device__ __forceinline__ void scale_node(int i, int j, int k, const double2 f[3], const DGrid &G, const DBox &box,
                                           const ScaleArgs &a, double2 out[3]) {
    out[0] = out[1] = out[2] = make_double2(0, 0);
    if (i == 0 && j == 0 && k == 0) return;   // k = 0 mode is dropped (Helper.cu:321-323, Mobility.cu:287)
    // Hermitian bookkeeping.  The reference fills the FULL complex grid, every node with the folded wave vector of its own index
    // (PSEv1/Helper.cu:307-315), transforms back with a C2C FFT and keeps the real part (PSEv1/Mobility.cu:447) -- i.e. the
    // Hermitian part 1/2 (G(k) + conj G(partner)) of what it wrote.  The equivalent on the half spectrum of a C2R inverse is the
    // symmetrised operator (S(k_node) + S(k_partner)) / 2, k_partner the folded wave vector of index (-i, -j, -k) mod N.
    // Wherever an index is a Nyquist index (even N) the folding gives node and partner the SAME sign in that component, so the
    // two operators differ (a whole x line for j = Ny/2, a node per line for i = Nx/2, the plane kz = Nz/2); elsewhere
    // k_partner = -k_node, S is even in k, and one evaluation serves.  Held to the reference's kernels by
    // tests/golden/reference_kernels.json.gz (Green and noise, even / odd / mixed grids).
    const bool plane = (k == 0) || ((G.Nz % 2 == 0) && (k == G.Nz / 2));
    const int ip = (G.Nx - i) % G.Nx, jp = (G.Ny - j) % G.Ny;
    double2 psi[3] = {{0, 0}, {0, 0}, {0, 0}};
    if (a.noise) {
        // K6: complex psi with Re, Im ~ U(-sqrt(3/2), sqrt(3/2)) (variance 1/2 each, Brownian.cu:178-189), keyed by the
        // canonical member of each conjugate pair so both members reconstruct the same draw; self-conjugate nodes are
        // real with variance 1 (x sqrt2, Brownian.cu:255-268).
        const unsigned long long own = ((unsigned long long)i * G.Ny + j) * G.Nz + k;
        const unsigned long long par = ((unsigned long long)ip * G.Ny + jp) * G.Nz + k;
        const bool selfc = plane && par == own;
        const bool flip = plane && par < own;
        const unsigned long long canon = (plane && par < own) ? par : own;
        uint32_t ra[4], rb[4];
        const uint32_t ts = a.ts_off ? a.timestep + *a.ts_off : a.timestep;
        philox4x32((uint32_t)canon, (uint32_t)(canon >> 32), ts, DOMAIN_GRID_A, a.seed, PHILOX_KEY1, ra);
        philox4x32((uint32_t)canon, (uint32_t)(canon >> 32), ts, DOMAIN_GRID_B, a.seed, PHILOX_KEY1, rb);
        const double s = 1.2247448713915890;  // sqrt(3/2)
        const double re[3] = {uniform_pm(ra[0], s), uniform_pm(ra[1], s), uniform_pm(ra[2], s)};
        const double im[3] = {uniform_pm(ra[3], s), uniform_pm(rb[0], s), uniform_pm(rb[1], s)};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (selfc) psi[c] = make_double2(1.4142135623730951 * re[c], 0.0);
            else       psi[c] = make_double2(re[c], flip ? -im[c] : im[c]);
        }
    }
    const KOp o1 = make_kop(i, j, k, G, box, a.xi, a.eta, a.noise_fac);
    const bool nyq = plane || ((G.Nx % 2 == 0) && (i == G.Nx / 2)) || ((G.Ny % 2 == 0) && (j == G.Ny / 2));
    if (nyq) {
        const KOp o2 = make_kop(ip, jp, (G.Nz - k) % G.Nz, G, box, a.xi, a.eta, a.noise_fac);
        apply_kop(o1, f, psi, a.noise, 0.5, out);
        apply_kop(o2, f, psi, a.noise, 0.5, out);
    } else {
        apply_kop(o1, f, psi, a.noise, 1.0, out);
    }
}

// debug: the k-space operator of given nodes (i, j, k) as the scaling kernels evaluate it: kx, ky, kz, B = w sinc^2,
// c / noise_fac = sqrt(w) sinc  (what K1 gpu_stokes_SetGridk_kernel tabulates and K5 / K6 apply, PSEv1/Helper.cu:300-327)
__global__ void k_debug_kop(const int *__restrict__ ijk, int n, DGrid G, DBox box, double xi, double eta, double *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int i = ijk[3 * t], j = ijk[3 * t + 1], k = ijk[3 * t + 2];
    KOp o = make_kop(i, j, k, G, box, xi, eta, 1.0);
    if (i == 0 && j == 0 && k == 0) { o.B = 0.0; o.c = 0.0; }
    out[5 * t] = o.kx; out[5 * t + 1] = o.ky; out[5 * t + 2] = o.kz; out[5 * t + 3] = o.B; out[5 * t + 4] = o.c;
}
void launch_debug_kop(const int *ijk, int n, DGrid G, DBox box, double xi, double eta, double *out, hipStream_t s) {
    hipLaunchKernelGGL(k_debug_kop, dim3(nblocks(n, TPB)), dim3(TPB), 0, s, ijk, n, G, box, xi, eta, out);
}

__global__ void __launch_bounds__(TPB)
k_scale(double2 *__restrict__ X, double2 *__restrict__ Y, double2 *__restrict__ Z, DGrid G, DBox box, ScaleArgs a) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t rows = a.transposed ? (size_t)a.nyl * G.Nx : (size_t)G.nxl * G.Ny;
    if (tid >= rows * G.Nzp) return;
    const int k = (int)(tid % G.Nzp);
    if (k >= G.Nzh) return;                                  // padding of the row
    const size_t row = tid / G.Nzp;
    int i, j;
    if (a.transposed) { i = (int)(row / a.nyl); j = a.y0 + (int)(row % a.nyl); }   // [Nx][ny_local][Nzh]
    else              { i = G.x0 + (int)(row / G.Ny); j = (int)(row % G.Ny); }
    const double2 f[3] = {X[tid], Y[tid], Z[tid]};
    double2 out[3];
    scale_node(i, j, k, f, G, box, a, out);
    X[tid] = out[0]; Y[tid] = out[1]; Z[tid] = out[2];
}

void launch_scale(double2 *X, double2 *Y, double2 *Z, DGrid G, DBox box, ScaleArgs a, hipStream_t s) {
    const size_t rows = a.transposed ? (size_t)a.nyl * G.Nx : (size_t)G.nxl * G.Ny;
    hipLaunchKernelGGL(k_scale, dim3(nblocks((long)(rows * G.Nzp), TPB)), dim3(TPB), 0, s, X, Y, Z, G, box, a);
}

// ---- fused x pass -----------------------------------------------------------------------------------------------
// For a power-of-two Nx the last forward axis pass, the k-space scaling (+ noise) and the first inverse axis pass are one
// kernel: a workgroup owns KB consecutive kz columns of one y row for all three components ([3][KB][Nx] complex in LDS),
// runs the forward x transforms (Stockham radix-4/2, in place through registers), applies scale_node, runs the inverse
// transforms and stores -- the spectra cross HBM once instead of three times (rocFFT x pass, k_scale, rocFFT x pass).
// Both transforms are unnormalised: the 1/Ng lives in the scale factor (Helper.cu:325-326).

template <int LOGN, int KB, int NTH, bool INVERSE>
__device__ __forceinline__ void lds_fft_x(double2 *__restrict__ d, const double2 *__restrict__ tw) {
    constexpr int N = 1 << LOGN, NCOL = 3 * KB, CS = N + 1;   // padded column stride: columns start on different banks
    const int tid = threadIdx.x;
    int ns = 1;
#pragma unroll
    for (int done = 0; done < LOGN;) {
        if (LOGN - done >= 2) {
            constexpr int R = 4, NB = NCOL * (N / R), PER = (NB + NTH - 1) / NTH;
            double2 v[PER][R];
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int bfly = tid + q * NTH;
                if (bfly < NB) {
                    const int col = bfly / (N / R), jj = bfly - col * (N / R), kk = jj & (ns - 1);
                    const double2 *src = d + col * CS + jj;
                    const int tstep = kk * (N / (ns * R));
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        double2 x = src[r * (N / R)];
                        if (r) { double2 w = tw[r * tstep]; if (INVERSE) w.y = -w.y; x = cmul(x, w); }
                        v[q][r] = x;
                    }
                    const double2 a0 = make_double2(v[q][0].x + v[q][2].x, v[q][0].y + v[q][2].y);
                    const double2 a1 = make_double2(v[q][0].x - v[q][2].x, v[q][0].y - v[q][2].y);
                    const double2 a2 = make_double2(v[q][1].x + v[q][3].x, v[q][1].y + v[q][3].y);
                    const double2 t = make_double2(v[q][1].x - v[q][3].x, v[q][1].y - v[q][3].y);
                    const double2 a3 = INVERSE ? make_double2(-t.y, t.x) : make_double2(t.y, -t.x);   // +-i (v1 - v3)
                    v[q][0] = make_double2(a0.x + a2.x, a0.y + a2.y);
                    v[q][1] = make_double2(a1.x + a3.x, a1.y + a3.y);
                    v[q][2] = make_double2(a0.x - a2.x, a0.y - a2.y);
                    v[q][3] = make_double2(a1.x - a3.x, a1.y - a3.y);
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int bfly = tid + q * NTH;
                if (bfly < NB) {
                    const int col = bfly / (N / R), jj = bfly - col * (N / R), kk = jj & (ns - 1);
                    double2 *dst = d + col * CS + (jj - kk) * R + kk;
#pragma unroll
                    for (int r = 0; r < R; ++r) dst[r * ns] = v[q][r];
                }
            }
            __syncthreads();
            ns *= 4; done += 2;
        } else {
            constexpr int R = 2, NB = NCOL * (N / R), PER = (NB + NTH - 1) / NTH;
            double2 v[PER][R];
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int bfly = tid + q * NTH;
                if (bfly < NB) {
                    const int col = bfly / (N / R), jj = bfly - col * (N / R), kk = jj & (ns - 1);
                    const double2 *src = d + col * CS + jj;
                    double2 w = tw[kk * (N / (ns * R))];
                    if (INVERSE) w.y = -w.y;
                    const double2 x0 = src[0], x1 = cmul(src[N / R], w);
                    v[q][0] = make_double2(x0.x + x1.x, x0.y + x1.y);
                    v[q][1] = make_double2(x0.x - x1.x, x0.y - x1.y);
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int bfly = tid + q * NTH;
                if (bfly < NB) {
                    const int col = bfly / (N / R), jj = bfly - col * (N / R), kk = jj & (ns - 1);
                    double2 *dst = d + col * CS + (jj - kk) * R + kk;
                    dst[0] = v[q][0];
                    dst[ns] = v[q][1];
                }
            }
            __syncthreads();
            ns *= 2; done += 1;
        }
    }
}

template <int LOGN, int KB, int NTH>
__global__ void __launch_bounds__(NTH, (NTH == 256 ? 3 : 1))   // 256 threads: <= 168 VGPRs so that three workgroups share a CU
k_xfft_scale(double2 *__restrict__ X, double2 *__restrict__ Y, double2 *__restrict__ Z, DGrid G, DBox box, ScaleArgs a,
             const double2 *__restrict__ twiddle) {
    constexpr int N = 1 << LOGN, CS = N + 1;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double2 *d = reinterpret_cast<double2 *>(smem_raw);      // [3][KB][N + 1]
    double2 *tw = d + 3 * KB * CS;                            // [N]  exp(-2 pi i m / N)
    const int tid = threadIdx.x;
    const int nkb = (G.Nzh + KB - 1) / KB;
    // layout [Nx][rows][Nzh] per component: all Ny rows on a single GPU, the nyl rows [y0, y0 + nyl) of this rank after
    // the slab transpose
    const int rows = a.transposed ? a.nyl : G.Ny;
    // neighbouring kz blocks of a row share 128-byte lines (a block's pieces are 64 or 32 bytes): keep them on one XCD, so the
    // second one finds the line in that XCD's L2 (round-robin placement fetched every line from memory twice)
    const int bid = xcd_block(blockIdx.x, gridDim.x);
    const int jl = bid / nkb, k0 = (bid - jl * nkb) * KB;
    const int j = a.transposed ? a.y0 + jl : jl;
    const int kv = min(KB, G.Nzh - k0);
    double2 *comp[3] = {X, Y, Z};
    const size_t xstride = (size_t)rows * G.Nzp, base = (size_t)jl * G.Nzp + k0;
    for (int e = tid; e < N; e += NTH) tw[e] = twiddle[e];
    {   // every load of a lane in flight before the first is parked in LDS (a load -> store loop pays the latency per element)
        constexpr int PER = (3 * N * KB + NTH - 1) / NTH;
        double2 v[PER];
#pragma unroll
        for (int it = 0; it < PER; ++it) {                    // KB consecutive kz are contiguous in memory (128 B at KB = 8)
            const int e = tid + it * NTH, c = e / (N * KB), r = e - c * (N * KB), x = r / KB, q = r - x * KB;
            v[it] = make_double2(0, 0);
            if (e < 3 * N * KB && q < kv) v[it] = comp[c][(size_t)x * xstride + base + q];
        }
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int e = tid + it * NTH, c = e / (N * KB), r = e - c * (N * KB), x = r / KB, q = r - x * KB;
            if (e < 3 * N * KB) d[(c * KB + q) * CS + x] = v[it];
        }
    }
    __syncthreads();
    lds_fft_x<LOGN, KB, NTH, false>(d, tw);
    for (int e = tid; e < N * KB; e += NTH) {
        const int x = e / KB, q = e - x * KB;
        if (q < kv) {
            const double2 f[3] = {d[q * CS + x], d[(KB + q) * CS + x], d[(2 * KB + q) * CS + x]};
            double2 out[3];
            scale_node(x, j, k0 + q, f, G, box, a, out);
            d[q * CS + x] = out[0]; d[(KB + q) * CS + x] = out[1]; d[(2 * KB + q) * CS + x] = out[2];
        }
    }
    __syncthreads();
    lds_fft_x<LOGN, KB, NTH, true>(d, tw);
    for (int e = tid; e < 3 * N * KB; e += NTH) {
        const int c = e / (N * KB), r = e - c * (N * KB), x = r / KB, q = r - x * KB;
        if (q < kv) comp[c][(size_t)x * xstride + base + q] = d[(c * KB + q) * CS + x];
    }
}


// ---- any Nx = 2^a 3^b 5^c (the grids the reference's rule produces, PSEv1/Stokes.cc:147-199) ---------------------------------
// Mixed-radix Stockham passes (radix 5, 4, 3, 2) between two LDS buffers: a butterfly reads its R points from one buffer and
// writes them to the other, so nothing is held across the barrier and the number of butterflies per lane may be anything.
struct FftPlanX { int n, nstage, radix[10]; unsigned mns[10], mnr[10]; };   // + reciprocals ceil(2^32 / ns), ceil(2^32 / (n / R)) of every stage
__device__ __forceinline__ int fast_quot(int a, unsigned m) { return (int)__umulhi((unsigned)a, m); }   // a / d for a d < 2^31, m = ceil(2^32 / d)

// one pass over NCOL columns of n points (column stride cs): in -> out
template <int R, bool INVERSE>
__device__ __forceinline__ void fft_pass(const double2 *__restrict__ in, double2 *__restrict__ out, const double2 *__restrict__ tw,
                                         int n, int cs, int ncol, int ns, int nth, unsigned mns, unsigned mnr) {
    const int nr = n / R, nb = ncol * nr, tstep = n / (ns * R);
    for (int bfly = threadIdx.x; bfly < nb; bfly += nth) {
        const int col = fast_quot(bfly, mnr), jj = bfly - col * nr, kk = ns == 1 ? 0 : jj - fast_quot(jj, mns) * ns;   // no runtime divisions (ns = 1: the reciprocal 2^32 does not fit)
        const double2 *src = in + col * cs + jj;
        double2 v[9];
        // W^{r kk tstep} for r = 1 .. R - 1 as powers of ONE table entry: a read of the twiddle table per point was a third of the
        // LDS traffic of a pass, and the passes are what bounds the mixed-radix kernels (error: R - 2 <= 7 roundings)
        double2 w1 = tw[kk * tstep];                                 // kk * tstep < n
        if (INVERSE) w1.y = -w1.y;
        double2 w = w1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double2 x = src[r * nr];
            if (r) {
                x = cmul(x, w);
                if (r + 1 < R) w = cmul(w, w1);
            }
            v[r] = x;
        }
        dft_small<R, INVERSE>(v);
        double2 *dst = out + col * cs + (jj - kk) * R + kk;
#pragma unroll
        for (int r = 0; r < R; ++r) dst[r * ns] = v[r];
    }
}

template <bool INVERSE>
__device__ __forceinline__ double2 *fft_mixed(double2 *a, double2 *b, const double2 *tw, const FftPlanX &pl, int cs, int ncol, int nth) {
    int ns = 1;
    for (int st = 0; st < pl.nstage; ++st) {
        const int R = pl.radix[st];
        if (R == 9) fft_pass<9, INVERSE>(a, b, tw, pl.n, cs, ncol, ns, nth, pl.mns[st], pl.mnr[st]);
        else if (R == 8) fft_pass<8, INVERSE>(a, b, tw, pl.n, cs, ncol, ns, nth, pl.mns[st], pl.mnr[st]);
        else if (R == 5) fft_pass<5, INVERSE>(a, b, tw, pl.n, cs, ncol, ns, nth, pl.mns[st], pl.mnr[st]);
        else if (R == 4) fft_pass<4, INVERSE>(a, b, tw, pl.n, cs, ncol, ns, nth, pl.mns[st], pl.mnr[st]);
        else if (R == 3) fft_pass<3, INVERSE>(a, b, tw, pl.n, cs, ncol, ns, nth, pl.mns[st], pl.mnr[st]);
        else fft_pass<2, INVERSE>(a, b, tw, pl.n, cs, ncol, ns, nth, pl.mns[st], pl.mnr[st]);
        __syncthreads();
        double2 *t = a; a = b; b = t;
        ns *= R;
    }
    return a;                                                        // the buffer that holds the result
}

// The same passes with the length and the radices known at compile time (the sizes the reference's rule picks at the BASELINE
// configurations: 360, 270, 180, 375, 500, ...).  The fused x pass is bound by vector-instruction issue (57 % of the cycles, 3 170
// instructions per wave at 360: profiles/r04_x360_counters.txt); here the index arithmetic is constant-folded, the butterflies of a
// lane are unrolled so that the loads of the second are in flight during the first, the first pass (all twiddles 1) multiplies
// nothing, and later passes read their twiddles from the table (no product chain: arithmetic is what this kernel is short of).
template <int N_, int R0, int R1 = 1, int R2 = 1, int R3 = 1>
struct CtPlan { static constexpr int N = N_, NST = 1 + (R1 > 1) + (R2 > 1) + (R3 > 1); };
struct RtPlan { static constexpr int N = 0; };

template <int R, bool INVERSE, int N, int NS, int NCOL, int NTH>
__device__ __forceinline__ void fft_pass_ct(const double2 *__restrict__ in, double2 *__restrict__ out, const double2 *__restrict__ tw) {
    constexpr int NR = N / R, NB = NCOL * NR, TSTEP = N / (NS * R), CS = N + 1, ITER = (NB + NTH - 1) / NTH;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int bfly = (int)threadIdx.x + it * NTH;
        if (NB % NTH == 0 || bfly < NB) {
            const int col = bfly / NR, jj = bfly - col * NR, kk = NS == 1 ? 0 : jj % NS;
            const double2 *src = in + col * CS + jj;
            double2 v[9];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double2 x = src[r * NR];
                if (r && NS > 1) {
                    double2 w = tw[r * kk * TSTEP];              // r kk TSTEP < R NS TSTEP = N
                    if (INVERSE) w.y = -w.y;
                    x = cmul(x, w);
                }
                v[r] = x;
            }
            dft_small<R, INVERSE>(v);
            double2 *dst = out + col * CS + (jj - kk) * R + kk;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r * NS] = v[r];
        }
    }
}
template <int R, bool INVERSE, int N, int NS, int NCOL, int NTH>
__device__ __forceinline__ void fft_stage_ct(double2 *&a, double2 *&b, const double2 *tw) {
    if constexpr (R > 1) {
        fft_pass_ct<R, INVERSE, N, NS, NCOL, NTH>(a, b, tw);
        __syncthreads();
        double2 *t = a; a = b; b = t;
    }
}
template <class P> struct CtRadices;
template <int N_, int R0, int R1, int R2, int R3>
struct CtRadices<CtPlan<N_, R0, R1, R2, R3>> { static constexpr int r0 = R0, r1 = R1, r2 = R2, r3 = R3; };
template <class P, bool INVERSE, int NCOL, int NTH>
__device__ __forceinline__ double2 *fft_mixed_ct(double2 *a, double2 *b, const double2 *tw) {
    using Rd = CtRadices<P>;
    fft_stage_ct<Rd::r0, INVERSE, P::N, 1, NCOL, NTH>(a, b, tw);
    fft_stage_ct<Rd::r1, INVERSE, P::N, Rd::r0, NCOL, NTH>(a, b, tw);
    fft_stage_ct<Rd::r2, INVERSE, P::N, Rd::r0 * Rd::r1, NCOL, NTH>(a, b, tw);
    fft_stage_ct<Rd::r3, INVERSE, P::N, Rd::r0 * Rd::r1 * Rd::r2, NCOL, NTH>(a, b, tw);
    return a;
}

// ---- N = R0 x R1 x R2 with ONE component in LDS at a time (512 = 8 x 8 x 8, 360 = 10 x 6 x 6, 256 = 4 x 8 x 8) ---------------------
// The kernels above keep the three components of a block of columns in LDS, which caps a 512-point block at two kz columns
// (32-byte pieces of every 128-byte line: 2.4 TB/s; the 256-point kernel loses the same 30 % when it is given two columns) and
// sends a point through LDS four times per transform.  Here the data live in registers, LDS holds one component of the block
// while it changes hands, and a workgroup owns KB = 4 kz columns (64-byte pieces; 8 = whole 128-byte pieces needs the kernel in 128
// registers, which it is not: 225 spilled, 3.3 ms at 512^3):
//   layout A (global memory side): thread (q, n) = (tid % KB, tid / KB), n < N / R0, holds x[n + (N / R0) r] of column q -- the
//            loads and stores of KB lanes cover one contiguous piece; stage 1 (decimation in frequency) is the radix-R0 transform over r;
//   layout B: wave w = column w; lanes (k0, n'') run stage 2 (radix R1 inside the N / R0-point transform k0), lanes (k0, k1) stage 3
//            (radix R2), with one exchange in between that stays inside the wave (LDS instructions of a wave execute in order: no
//            barrier).  Stage 3 leaves X[k0 + R0 k1 + R0 R1 k2] in register k2 of lane R1 k0 + k1 for all three components, so the
//            k-space operator works on registers (PARK: the third component waits in the lane's own LDS slots meanwhile -- the
//            operator needs ~100 registers of its own), and the inverse (decimation in time) runs the same stages backwards from
//            that digit-reversed order.
// Per transform a point crosses LDS twice; 10 barriers per block; every load of the block is in flight at once and a stage's
// twiddles are fetched once for the three components (stage by stage over the components: a variant that took a component
// through all stages before the next -- fewer live registers -- was 15 % slower at 512^3: three times the twiddle fetches).
// Every stage has registers of its own, defined in ALL lanes: where only some lanes take part in a layout (360: 144 of 256
// threads in A, 60 of 64 lanes in B) values left in the others would stay alive across the whole kernel (291 registers spilled).
// Global addresses are a uniform base plus a 32-bit byte offset per lane, recomputed at the stores (the launcher checks < 4 GiB
// per component): 64-bit addresses kept from the loads to the stores were spilled too.
// Positions in a column: B[k0][n] at P0 k0 + n, C[k0][k1][n] at P0 k0 + P1 k1 + n -- every access is a per-lane base plus a
// compile-time offset; P0, P1 and the column stride CS come from tools/debug/lds_banks_xcols.py (every access conflict-free but the
// layout A read of the inverse, two-way).
// 512^3: 1.5 ms = 4.3 TB/s in some processes, 1.77 ms in others (both kernels of this file show the two modes, whatever the
// plane stride: 128 bytes to 64 KB appended to every x plane changed nothing) against 2.77 ms; 360^3: 0.80 ms against 1.05.
// 256 = 8 x 8 x 4 through this kernel (one column per wave, 64-byte pieces): 0.25 - 0.27 ms at 256^3, no better than k_xfft_scale256;
// 256 = 4 x 8 x 8 with two columns per wave (CPW = 2, 128-byte pieces): 0.201 ms against 0.226 -- dispatched.
// Also measured and dropped: a table of the operator's scalars per node (what the reference keeps in gridk.w) instead of the
// exponential, sine and reciprocal square root per node: 256^3 0.25 -> 0.30 ms, 512^3 +20 % (a dependent load in the middle of
// the block costs more than the 150 instructions it saves).
template <int N, int R0, int R1, int KB, int WPS, bool PARK, int P0, int P1, int CS, int CPW = 1>
__global__ void __launch_bounds__(64 * KB / CPW, WPS)
k_xfft_scale_cols(double2 *__restrict__ X, double2 *__restrict__ Y, double2 *__restrict__ Z, DGrid G, DBox box, ScaleArgs a,
                  const double2 *__restrict__ twiddle) {
    // CPW = 2 (256 = 4 x 8 x 8): TWO columns per wave in layout B (32 lanes each), so that a workgroup of four waves owns eight kz
    // columns = whole 128-byte pieces; a thread of layout A then runs NBA = 2 butterflies (n and n + NTH / KB)
    constexpr int M1 = N / R0, R2 = M1 / R1, L2 = R0 * R2, L3 = R0 * R1, TA = KB * M1, LC = 64 / CPW, NTH = 64 * KB / CPW;
    constexpr int NBA = (TA + NTH - 1) / NTH, NSTEP = NTH / KB;
    static_assert(R0 * R1 * R2 == N && L2 <= LC && L3 <= LC && NTH % KB == 0 && (NBA == 1 || TA == NBA * NTH) && CS >= LC * R2, "plan");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double2 *buf = reinterpret_cast<double2 *>(smem_raw);    // [KB][CS]: one component of the block
    double2 *twS = buf + KB * CS;                            // exp(-2 pi i m / M1), m < M1
    const int tid = threadIdx.x;
    const int nkb = (G.Nzh + KB - 1) / KB;
    const int rows = a.transposed ? a.nyl : G.Ny;
    const int bid = xcd_block(blockIdx.x, gridDim.x);        // rows are not multiples of 128 bytes: neighbouring blocks share lines
    const int jl = bid / nkb, kz0 = (bid - jl * nkb) * KB;
    const int j = a.transposed ? a.y0 + jl : jl;
    const int kv = min(KB, G.Nzh - kz0);
    // global addresses: a component's block base (uniform) + a 32-bit byte offset per lane (the launcher checks that a component is
    // below 4 GiB), recomputed where it is used: 64-bit addresses of every point kept from the loads to the stores were most of
    // the register spills of this kernel
    const size_t xstride = (size_t)rows * G.Nzp, base = (size_t)jl * G.Nzp + kz0;
    char *comp[3] = {reinterpret_cast<char *>(X + base), reinterpret_cast<char *>(Y + base), reinterpret_cast<char *>(Z + base)};
    const unsigned xs16 = (unsigned)(xstride * sizeof(double2));
    const int q = tid % KB;                                  // layout A
    const bool actA = TA >= NTH || tid < TA;
    const int n1 = actA ? tid / KB : 0;                      // (+ NSTEP for the second butterfly)
    const int w = (tid >> 6) * CPW + (tid & 63) / LC, l = (tid & 63) % LC;   // layout B: column, lane of the column
    const bool act2 = L2 == LC || l < L2, act3 = L3 == LC || l < L3;
    const int kp2 = act2 ? l / R2 : 0, nn = l % R2, kp3 = act3 ? l / R1 : 0, k1 = l % R1;
    double2 *pA = buf + q * CS + n1;                         // + P0 k0 (+ NSTEP)
    double2 *pB = buf + w * CS + P0 * kp2 + nn;              // + R2 s (B), + P1 k1 (C)
    double2 *pC = buf + w * CS + P0 * kp3 + P1 * k1;         // + n'' (C of lane (k0, k1))
    double2 *pP = buf + w * CS + l;                          // + LC k2: the lane's own slots
    double2 ld[3][NBA][R0], v[3][R2];                        // layout A points of a component; its final values (every stage has registers of
                                                             // its own, defined in all lanes: what a stage leaves behind must not stay alive)
    const double2 zero = make_double2(0, 0);
    auto offset0 = [&]() __attribute__((always_inline)) {
        unsigned o = (unsigned)n1 * xs16 + (unsigned)q * (unsigned)sizeof(double2);
        asm volatile("" : "+v"(o));                           // not a common subexpression of the other uses
        return o;
    };
    auto load = [&](int c) __attribute__((always_inline)) {
        const unsigned o = offset0();
#pragma unroll
        for (int u = 0; u < NBA; ++u)
#pragma unroll
            for (int r = 0; r < R0; ++r) {
                ld[c][u][r] = make_double2(0, 0);
                if (actA && q < kv) ld[c][u][r] = *reinterpret_cast<const double2 *>(comp[c] + (size_t)(o + (unsigned)(M1 * r + NSTEP * u) * xs16));
            }
    };
    load(0); load(1); load(2);                               // every load of the block in flight at once
    if (tid < M1) twS[tid] = twiddle[R0 * tid];
    // ---- forward: stage 1 over r -> k0, times W_N^{n k0}
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int u = 0; u < NBA; ++u) dft_small<R0, false>(ld[c][u]);
#pragma unroll
    for (int u = 0; u < NBA; ++u)
#pragma unroll
        for (int k = 1; k < R0; ++k) {
            const double2 t = twiddle[(n1 + NSTEP * u) * k];
#pragma unroll
            for (int c = 0; c < 3; ++c) ld[c][u][k] = cmul(ld[c][u][k], t);
        }
    double2 b[3][R1];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (c) __syncthreads();                               // every wave has read the previous component
        if (actA) {
#pragma unroll
            for (int u = 0; u < NBA; ++u)
#pragma unroll
                for (int k = 0; k < R0; ++k) pA[P0 * k + NSTEP * u] = ld[c][u][k];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < R1; ++s) b[c][s] = act2 ? pB[R2 * s] : zero;
    }
    // stage 2 over s -> k1, times W_M1^{n'' k1}
#pragma unroll
    for (int c = 0; c < 3; ++c) dft_small<R1, false>(b[c]);
#pragma unroll
    for (int k = 1; k < R1; ++k) {
        const double2 t = twS[nn * k];
#pragma unroll
        for (int c = 0; c < 3; ++c) b[c][k] = cmul(b[c][k], t);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {                             // inside the wave's own column: no barrier
        if (act2) {
#pragma unroll
            for (int k = 0; k < R1; ++k) pB[P1 * k] = b[c][k];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n = 0; n < R2; ++n) v[c][n] = act3 ? pC[n] : zero;
        __builtin_amdgcn_wave_barrier();
        dft_small<R2, false>(v[c]);                           // stage 3 over n'' -> k2
    }
    // ---- the k-space operator on X[k0 + R0 k1 + R0 R1 k2]
    if (PARK) {                                               // the third component waits in the lane's own slots: room for the operator's temporaries
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) pP[LC * k2] = v[2][k2];
    }
    if (w < kv && act3) {
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) {
            const double2 f[3] = {v[0][k2], v[1][k2], PARK ? pP[LC * k2] : v[2][k2]};
            double2 out[3];
            scale_node(kp3 + R0 * k1 + R0 * R1 * k2, j, kz0 + w, f, G, box, a, out);
            v[0][k2] = out[0]; v[1][k2] = out[1];
            if (PARK) pP[LC * k2] = out[2]; else v[2][k2] = out[2];
            __builtin_amdgcn_sched_barrier(0);                // one node at a time
        }
    }
    if (PARK) {
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) v[2][k2] = pP[LC * k2];
        __builtin_amdgcn_wave_barrier();
    }
    // ---- inverse: stage 3 backwards over k2 -> n'', times conj W_M1^{n'' k1}
#pragma unroll
    for (int c = 0; c < 3; ++c) dft_small<R2, true>(v[c]);
#pragma unroll
    for (int n = 1; n < R2; ++n) {
        double2 t = twS[n * k1]; t.y = -t.y;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c][n] = cmul(v[c][n], t);
    }
    double2 bi[3][R1];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (act3) {
#pragma unroll
            for (int n = 0; n < R2; ++n) pC[n] = v[c][n];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < R1; ++k) bi[c][k] = act2 ? pB[P1 * k] : zero;
        __builtin_amdgcn_wave_barrier();
        dft_small<R1, true>(bi[c]);                           // stage 2 backwards over k1 -> s
    }
#pragma unroll
    for (int s = 0; s < R1; ++s) {                            // times conj W_N^{(R2 s + n'') k0}
        double2 t = twiddle[(R2 * s + nn) * kp2]; t.y = -t.y;
#pragma unroll
        for (int c = 0; c < 3; ++c) bi[c][s] = cmul(bi[c][s], t);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (c) __syncthreads();                               // the layout A reads of the previous component are done
        if (act2) {
#pragma unroll
            for (int s = 0; s < R1; ++s) pB[R2 * s] = bi[c][s];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NBA; ++u) {
            double2 st[R0];
#pragma unroll
            for (int k = 0; k < R0; ++k) st[k] = actA ? pA[P0 * k + NSTEP * u] : zero;
            dft_small<R0, true>(st);                          // stage 1 backwards over k0 -> r
            if (actA && q < kv) {
                const unsigned o = offset0();
#pragma unroll
                for (int r = 0; r < R0; ++r) *reinterpret_cast<double2 *>(comp[c] + (size_t)(o + (unsigned)(M1 * r + NSTEP * u) * xs16)) = st[r];
            }
        }
    }
}

template <int N, int R0, int R1, int KB, int WPS, bool PARK, int P0, int P1, int CS, int CPW = 1>
static void launch_xfft_cols(double2 *X, double2 *Y, double2 *Z, DGrid G, DBox box, ScaleArgs a, const double2 *tw, hipStream_t s) {
    const size_t lds = (size_t)(KB * CS + N / R0) * sizeof(double2);
    static LdsAttr attr;
    if (attr.need(lds)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_xfft_scale_cols<N, R0, R1, KB, WPS, PARK, P0, P1, CS, CPW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int nkb = (G.Nzh + KB - 1) / KB;
    const int rows = a.transposed ? a.nyl : G.Ny;
    hipLaunchKernelGGL((k_xfft_scale_cols<N, R0, R1, KB, WPS, PARK, P0, P1, CS, CPW>), dim3(rows * nkb), dim3(64 * KB / CPW), lds, s, X, Y, Z, G, box, a, tw);
}

template <int KB, int NTH, class PLAN = RtPlan>
__global__ void __launch_bounds__(NTH)
k_xfft_scale_mixed(double2 *__restrict__ X, double2 *__restrict__ Y, double2 *__restrict__ Z, DGrid G, DBox box, ScaleArgs a,
                   const double2 *__restrict__ twiddle, FftPlanX pl) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr bool CT = PLAN::N > 0;
    const int N = CT ? PLAN::N : pl.n, CS = N + 1;
    constexpr int NCOL = 3 * KB;
    double2 *bufa = reinterpret_cast<double2 *>(smem_raw), *bufb = bufa + NCOL * CS;   // [3][KB][N + 1] each
    double2 *tw = bufb + NCOL * CS;                                                     // [N]
    const int tid = threadIdx.x;
    const int nkb = (G.Nzh + KB - 1) / KB;
    const int rows = a.transposed ? a.nyl : G.Ny;
    // neighbouring kz blocks of a row share 128-byte lines (a block's pieces are 64 or 32 bytes): keep them on one XCD, so the
    // second one finds the line in that XCD's L2 (round-robin placement fetched every line from memory twice)
    const int bid = xcd_block(blockIdx.x, gridDim.x);
    const int jl = bid / nkb, k0 = (bid - jl * nkb) * KB;
    const int j = a.transposed ? a.y0 + jl : jl;
    const int kv = min(KB, G.Nzh - k0);
    double2 *comp[3] = {X, Y, Z};
    const size_t xstride = (size_t)rows * G.Nzp, base = (size_t)jl * G.Nzp + k0;
    for (int e = tid; e < N; e += NTH) tw[e] = twiddle[e];
    const int total = 3 * N * KB;
    for (int e0 = 0; e0 < total; e0 += 4 * NTH) {             // four loads of a lane in flight at a time
        double2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * NTH + tid, c = e / (N * KB), r = e - c * (N * KB), x = r / KB, q = r - x * KB;
            v[u] = make_double2(0, 0);
            if (e < total && q < kv) v[u] = comp[c][(size_t)x * xstride + base + q];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * NTH + tid, c = e / (N * KB), r = e - c * (N * KB), x = r / KB, q = r - x * KB;
            if (e < total) bufa[(c * KB + q) * CS + x] = v[u];
        }
    }
    __syncthreads();
    double2 *d;
    if constexpr (CT) d = fft_mixed_ct<PLAN, false, NCOL, NTH>(bufa, bufb, tw);
    else d = fft_mixed<false>(bufa, bufb, tw, pl, CS, NCOL, NTH);
    for (int e = tid; e < N * KB; e += NTH) {
        const int x = e / KB, q = e - x * KB;
        if (q < kv) {
            const double2 f[3] = {d[q * CS + x], d[(KB + q) * CS + x], d[(2 * KB + q) * CS + x]};
            double2 out[3];
            scale_node(x, j, k0 + q, f, G, box, a, out);
            d[q * CS + x] = out[0]; d[(KB + q) * CS + x] = out[1]; d[(2 * KB + q) * CS + x] = out[2];
        }
    }
    __syncthreads();
    if constexpr (CT) d = fft_mixed_ct<PLAN, true, NCOL, NTH>(d, d == bufa ? bufb : bufa, tw);
    else d = fft_mixed<true>(d, d == bufa ? bufb : bufa, tw, pl, CS, NCOL, NTH);
    for (int e = tid; e < total; e += NTH) {
        const int c = e / (N * KB), r = e - c * (N * KB), x = r / KB, q = r - x * KB;
        if (q < kv) comp[c][(size_t)x * xstride + base + q] = d[(c * KB + q) * CS + x];
    }
}

static bool plan_x(int n, FftPlanX &pl) {
    pl.n = n; pl.nstage = 0;
    int m = n;
    for (int r : {9, 8, 5, 4, 3, 2})   // few, wide passes: 360 = 9 8 5
        while (m % r == 0) { if (pl.nstage == 10) return false; pl.radix[pl.nstage++] = r; m /= r; }
    // the first pass scatters with a stride of R points: free of LDS bank conflicts for odd R only -- lead with an odd radix if there is one
    for (int st = 1; st < pl.nstage && (pl.radix[0] & 1) == 0; ++st)
        if (pl.radix[st] & 1) std::swap(pl.radix[0], pl.radix[st]);
    unsigned ns = 1;
    for (int st = 0; st < pl.nstage; ++st) {
        const unsigned nr = (unsigned)n / (unsigned)pl.radix[st];
        pl.mns[st] = (unsigned)((0x100000000ull + ns - 1) / ns);
        pl.mnr[st] = (unsigned)((0x100000000ull + nr - 1) / nr);
        ns *= (unsigned)pl.radix[st];
    }
    return m == 1;
}

template <int KB, int NTH, class PLAN = RtPlan>
static void launch_xfft_mixed(double2 *X, double2 *Y, double2 *Z, DGrid G, DBox box, ScaleArgs a, const double2 *tw, const FftPlanX &pl, hipStream_t s) {
    const size_t lds = (size_t)(2 * 3 * KB * (pl.n + 1) + pl.n) * sizeof(double2);
    static LdsAttr attr;
    if (lds > 48 * 1024 && attr.need(lds)) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_xfft_scale_mixed<KB, NTH, PLAN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int nkb = (G.Nzh + KB - 1) / KB;
    const int rows = a.transposed ? a.nyl : G.Ny;
    hipLaunchKernelGGL((k_xfft_scale_mixed<KB, NTH, PLAN>), dim3(rows * nkb), dim3(NTH), lds, s, X, Y, Z, G, box, a, tw, pl);
}

// ---- own y pass (round 4): complex transforms along y of the half spectra, in place, for grids 2^a 3^b 5^c that are not powers of
// two.  rocFFT's strided pass over y is its slow one there (360^2: 0.93 ms of the 1.40 ms of its 2-D transform, the z pass takes
// 0.48 ms = what its bytes cost); here a workgroup takes KB consecutive kz of one x plane (KB * 16-byte pieces, neighbouring blocks on
// one XCD as in the x pass) and all Ny rows, through the same mixed-radix passes as the x pass.  The z passes stay rocFFT's (1-D).
// MAP (slab ranks of a team): the transform also does the reordering between the plane layout [3][nxl][Ny][Nzp] and the layout of the
// all-to-all blocks [3][G][nxl][nyl][Nzp] (y = q nyl + jl goes to rank q) -- 1: plane layout in, block layout out (forward: what
// k_slab_pack did in a pass of its own); 2: block layout in, plane layout out (inverse: the unpack).  0: in place, plane layout.
template <int KB, int NTH, bool INVERSE, int MAP = 0, class PLAN = RtPlan>
__global__ void __launch_bounds__(NTH)
k_fft_cols(double2 *__restrict__ data, FftPlanX pl, const double2 *__restrict__ twiddle, int nkb, int Nzh, int Nzp, size_t plane_stride,
           double2 *__restrict__ other = nullptr, int nxl = 0, int nyl = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr bool CT = PLAN::N > 0;
    const int N = CT ? PLAN::N : pl.n, CS = N + 1;
    double2 *bufa = reinterpret_cast<double2 *>(smem_raw), *bufb = bufa + KB * CS;   // [KB][N + 1] each
    double2 *tw = bufb + KB * CS;                                                     // [N]
    const int tid = threadIdx.x;
    const int bid = xcd_block(blockIdx.x, gridDim.x);
    const int plane = bid / nkb, k0 = (bid - plane * nkb) * KB;
    const int kv = min(KB, Nzh - k0);
    double2 *base = data + (size_t)plane * plane_stride + k0;
    // the same (plane, y) in the block layout: component c = plane / nxl, plane lx = plane % nxl of this rank, block q = y / nyl
    const int pc = MAP ? plane / nxl : 0, plx = MAP ? plane - pc * nxl : 0;
    auto blk = [&](int y) -> size_t {
        const int q = y / nyl, jl = y - q * nyl;
        return (size_t)pc * nxl * plane_stride + (((size_t)q * nxl + plx) * nyl + jl) * Nzp + k0;
    };
    for (int e = tid; e < N; e += NTH) tw[e] = twiddle[e];
    const int total = N * KB;
    constexpr int U = 6;                                       // loads of a lane in flight at a time
    for (int e0 = 0; e0 < total; e0 += U * NTH) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * NTH + tid, y = e / KB, q = e - y * KB;
            v[u] = make_double2(0, 0);
            if (e < total && q < kv) v[u] = MAP == 2 ? other[blk(y) + q] : base[(size_t)y * Nzp + q];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * NTH + tid, y = e / KB, q = e - y * KB;
            if (e < total) bufa[q * CS + y] = v[u];
        }
    }
    __syncthreads();
    const double2 *d;
    if constexpr (CT) d = fft_mixed_ct<PLAN, INVERSE, KB, NTH>(bufa, bufb, tw);
    else d = fft_mixed<INVERSE>(bufa, bufb, tw, pl, CS, KB, NTH);
    for (int e = tid; e < total; e += NTH) {
        const int y = e / KB, q = e - y * KB;
        if (q < kv) {
            if (MAP == 1) other[blk(y) + q] = d[q * CS + y];
            else base[(size_t)y * Nzp + q] = d[q * CS + y];
        }
    }
}
bool yfft_possible(int Ny) {    // any 2^a 3^b 5^c, 16..512 (slab ranks: the pass also packs / unpacks the all-to-all blocks)
    FftPlanX pl;
    return Ny >= 16 && Ny <= 512 && plan_x(Ny, pl);
}
bool yfft_supported(int Ny) {   // 2^a 3^b 5^c, 16..512, not a power of two (rocFFT's 2-D kernels are good at those)
    FftPlanX pl;
    return Ny >= 16 && Ny <= 512 && (Ny & (Ny - 1)) != 0 && plan_x(Ny, pl);
}
template <int KB, int NTH, class PLAN = RtPlan>
static void launch_fft_cols(double2 *data, const FftPlanX &pl, const double2 *tw, int nplanes, int Nzh, int Nzp, size_t plane_stride,
                            bool inverse, hipStream_t s) {
    const size_t lds = (size_t)(2 * KB * (pl.n + 1) + pl.n) * sizeof(double2);
    static LdsAttr attr2[2];
    if (lds > 48 * 1024 && attr2[inverse].need(lds)) {
        if (inverse) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fft_cols<KB, NTH, true, 0, PLAN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fft_cols<KB, NTH, false, 0, PLAN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int nkb = (Nzh + KB - 1) / KB;
    const dim3 g(nplanes * nkb), b(NTH);
    if (inverse) hipLaunchKernelGGL((k_fft_cols<KB, NTH, true, 0, PLAN>), g, b, lds, s, data, pl, tw, nkb, Nzh, Nzp, plane_stride);
    else hipLaunchKernelGGL((k_fft_cols<KB, NTH, false, 0, PLAN>), g, b, lds, s, data, pl, tw, nkb, Nzh, Nzp, plane_stride);
}
// slab ranks: forward = planes (cgrid) -> transformed blocks (blocks); inverse = blocks -> transformed planes
static bool yfft_slab_regs(double2 *cgrid, double2 *blocks, DGrid G, int nyl, bool inverse, const double2 *tw, hipStream_t s);   // Ny = 256, 512: the register pass
void launch_yfft_slab(double2 *cgrid, double2 *blocks, DGrid G, int nyl, bool inverse, const double2 *tw, hipStream_t s, bool regs) {
    if (regs && yfft_slab_regs(cgrid, blocks, G, nyl, inverse, tw, s)) return;
    FftPlanX pl;
    plan_x(G.Ny, pl);
    constexpr int KB = 4, NTH = 256;
    const size_t lds = (size_t)(2 * KB * (pl.n + 1) + pl.n) * sizeof(double2);
    static LdsAttr attr2[2];
    if (lds > 48 * 1024 && attr2[inverse].need(lds)) {
        if (inverse) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fft_cols<KB, NTH, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fft_cols<KB, NTH, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int nkb = (G.Nzh + KB - 1) / KB, nplanes = 3 * G.nxl;
    const size_t ps = (size_t)G.Ny * G.Nzp;
    const dim3 g(nplanes * nkb), b(NTH);
    if (inverse) hipLaunchKernelGGL((k_fft_cols<KB, NTH, true, 2>), g, b, lds, s, cgrid, pl, tw, nkb, G.Nzh, G.Nzp, ps, blocks, G.nxl, nyl);
    else hipLaunchKernelGGL((k_fft_cols<KB, NTH, false, 1>), g, b, lds, s, cgrid, pl, tw, nkb, G.Nzh, G.Nzp, ps, blocks, G.nxl, nyl);
}
// ---- the y pass of Ny = 256 = 8 x 8 x 4 with the data in registers ------------------------------------------------------------------
// The stages of k_xfft_scale_cols without the operator: one component, one direction, natural order in and out -- after stage 3
// the wave parks X[k0 + 8 k1 + 64 k2] at its natural position of the column (padded by one per eight: the strided writes and the
// layout A reads are conflict-free), one more barrier, and layout A stores whole pieces.  Three trips through LDS and two barriers
// per block; eight kz columns = 128-byte pieces at ~60 registers.  (Inverse = the same decimation in frequency with conjugate
// twiddles, unnormalised like rocFFT's.)
// MAP (slab ranks, as k_fft_cols): 1 the forward pass stores into the all-to-all blocks `other`, 2 the inverse pass loads from them
template <int N, int R0, int R1, int KB, bool INVERSE, int P0, int P1, int CS, int MAP = 0>
__global__ void __launch_bounds__(64 * KB)
k_yfft_regs(double2 *__restrict__ data, const double2 *__restrict__ twiddle, int nkb, int Nzh, int Nzp, size_t plane_stride,
            double2 *__restrict__ other = nullptr, int nxl = 0, int nyl = 0) {
    constexpr int M1 = N / R0, R2 = M1 / R1, L2 = R0 * R2, L3 = R0 * R1, TA = KB * M1, PN = M1 + M1 / 8;
    static_assert(R0 == 8 && R1 == 8 && R0 * R1 * R2 == N && L2 <= 64 && TA <= 64 * KB && CS >= N + N / 8, "plan");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double2 *buf = reinterpret_cast<double2 *>(smem_raw);    // [KB][CS]
    double2 *twS = buf + KB * CS;                            // exp(-2 pi i m / M1), m < M1
    const int tid = threadIdx.x;
    const int bid = xcd_block(blockIdx.x, gridDim.x);
    const int plane = bid / nkb, kz0 = (bid - plane * nkb) * KB, kv = min(KB, Nzh - kz0);
    char *base = reinterpret_cast<char *>(data + (size_t)plane * plane_stride + kz0);
    const unsigned row16 = (unsigned)Nzp * (unsigned)sizeof(double2);
    const int q = tid % KB;
    const bool actA = TA == 64 * KB || tid < TA;
    const int n1 = actA ? tid / KB : 0;
    const int w = tid >> 6, l = tid & 63;
    const bool act2 = L2 == 64 || l < L2;
    const int kp2 = act2 ? l / R2 : 0, nn = l % R2, kp3 = l / R1, k1 = l % R1;
    double2 *pA = buf + q * CS + n1;                         // + P0 k0
    double2 *pB = buf + w * CS + P0 * kp2 + nn;              // + R2 s (B), + P1 k1 (C)
    double2 *pC = buf + w * CS + P0 * kp3 + P1 * k1;         // + n''
    double2 *pN = buf + w * CS + kp3 + 9 * k1;               // + 72 k2: natural position ky + ky / 8 of ky = k0 + 8 k1 + 64 k2
    double2 *pO = buf + q * CS + n1 + (n1 >> 3);             // + PN r: natural position of y = n + M1 r
    const double2 zero = make_double2(0, 0);
    const unsigned o0 = (unsigned)n1 * row16 + (unsigned)q * (unsigned)sizeof(double2);
    // the same (plane, y) in the block layout: component pc = plane / nxl, plane plx of this rank, block y / nyl
    const int pc = MAP ? plane / nxl : 0, plx = MAP ? plane - pc * nxl : 0;
    auto blk = [&](int y) -> double2 * {
        const int qb = y / nyl, jl = y - qb * nyl;
        return other + (size_t)pc * nxl * plane_stride + (((size_t)qb * nxl + plx) * nyl + jl) * Nzp + kz0 + q;
    };
    double2 a[R0];
#pragma unroll
    for (int r = 0; r < R0; ++r) {
        a[r] = zero;
        if (actA && q < kv) a[r] = MAP == 2 ? *blk(n1 + M1 * r) : *reinterpret_cast<const double2 *>(base + (size_t)(o0 + (unsigned)(M1 * r) * row16));
    }
    if (tid < M1) twS[tid] = twiddle[R0 * tid];
    dft_small<R0, INVERSE>(a);                                // stage 1 over r -> k0, times W_N^{n k0}
#pragma unroll
    for (int k = 1; k < R0; ++k) { double2 t = twiddle[n1 * k]; if (INVERSE) t.y = -t.y; a[k] = cmul(a[k], t); }
    if (actA) {
#pragma unroll
        for (int k = 0; k < R0; ++k) pA[P0 * k] = a[k];
    }
    __syncthreads();
    double2 b[R1];
#pragma unroll
    for (int s = 0; s < R1; ++s) b[s] = act2 ? pB[R2 * s] : zero;
    dft_small<R1, INVERSE>(b);                                // stage 2 over s -> k1, times W_M1^{n'' k1}
#pragma unroll
    for (int k = 1; k < R1; ++k) { double2 t = twS[nn * k]; if (INVERSE) t.y = -t.y; b[k] = cmul(b[k], t); }
    if (act2) {
#pragma unroll
        for (int k = 0; k < R1; ++k) pB[P1 * k] = b[k];
    }
    __builtin_amdgcn_wave_barrier();
    double2 v[R2];
#pragma unroll
    for (int n = 0; n < R2; ++n) v[n] = pC[n];
    __builtin_amdgcn_wave_barrier();
    dft_small<R2, INVERSE>(v);                                // stage 3 over n'' -> k2
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) pN[72 * k2] = v[k2];
    __syncthreads();
    if (actA && q < kv) {
#pragma unroll
        for (int r = 0; r < R0; ++r) {
            if (MAP == 1) *blk(n1 + M1 * r) = pO[PN * r];
            else *reinterpret_cast<double2 *>(base + (size_t)(o0 + (unsigned)(M1 * r) * row16)) = pO[PN * r];
        }
    }
}
template <int N, int R0, int R1, int KB, int P0, int P1, int CS>
static void launch_yfft_regs_slab(double2 *cgrid, double2 *blocks, DGrid G, int nyl, bool inverse, const double2 *tw, hipStream_t s) {
    const size_t lds = (size_t)(KB * CS + N / R0) * sizeof(double2);
    static LdsAttr attr;
    if (attr.need(lds)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_yfft_regs<N, R0, R1, KB, true, P0, P1, CS, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_yfft_regs<N, R0, R1, KB, false, P0, P1, CS, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int nkb = (G.Nzh + KB - 1) / KB, nplanes = 3 * G.nxl;
    const size_t ps = (size_t)G.Ny * G.Nzp;
    const dim3 g(nplanes * nkb), b(64 * KB);
    if (inverse) hipLaunchKernelGGL((k_yfft_regs<N, R0, R1, KB, true, P0, P1, CS, 2>), g, b, lds, s, cgrid, tw, nkb, G.Nzh, G.Nzp, ps, blocks, G.nxl, nyl);
    else hipLaunchKernelGGL((k_yfft_regs<N, R0, R1, KB, false, P0, P1, CS, 1>), g, b, lds, s, cgrid, tw, nkb, G.Nzh, G.Nzp, ps, blocks, G.nxl, nyl);
}
template <int N, int R0, int R1, int KB, int P0, int P1, int CS>
static void launch_yfft_regs(double2 *data, const double2 *tw, int nplanes, int Nzh, int Nzp, size_t plane_stride, bool inverse, hipStream_t s) {
    const size_t lds = (size_t)(KB * CS + N / R0) * sizeof(double2);
    static LdsAttr attr;
    if (attr.need(lds)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_yfft_regs<N, R0, R1, KB, true, P0, P1, CS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_yfft_regs<N, R0, R1, KB, false, P0, P1, CS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int nkb = (Nzh + KB - 1) / KB;
    const dim3 g(nplanes * nkb), b(64 * KB);
    if (inverse) hipLaunchKernelGGL((k_yfft_regs<N, R0, R1, KB, true, P0, P1, CS>), g, b, lds, s, data, tw, nkb, Nzh, Nzp, plane_stride);
    else hipLaunchKernelGGL((k_yfft_regs<N, R0, R1, KB, false, P0, P1, CS>), g, b, lds, s, data, tw, nkb, Nzh, Nzp, plane_stride);
}
// 256^3: forward 0.30 against 0.31 - 0.32 ms, inverse 0.32 against 0.34 (rocFFT's 2-D plan).  Not at 512: the pass itself equals rocFFT's
// strided one there (2.6 ms per direction either way), and rocFFT's 1-D real forward transform of 512 points, which would replace
// the z half of its 2-D plan, is slow (3.7 ms per direction at 512^3 instead of 2.6).
// Round 5: with the own z pass (k_zfft_rows) the register y pass also runs at 512 (8 x 8 x 8, four columns per workgroup).
bool yfft_regs_supported(int Ny, int Nz, bool own_z) { return (Ny == 256 && Nz <= 256) || (own_z && (Ny == 256 || Ny == 512)); }

static bool yfft_slab_regs(double2 *cgrid, double2 *blocks, DGrid G, int nyl, bool inverse, const double2 *tw, hipStream_t s) {
    if (G.Ny == 256) { launch_yfft_regs_slab<256, 8, 8, 8, 44, 5, 359>(cgrid, blocks, G, nyl, inverse, tw, s); return true; }
    if (G.Ny == 512) { launch_yfft_regs_slab<512, 8, 8, 4, 72, 9, 578>(cgrid, blocks, G, nyl, inverse, tw, s); return true; }
    return false;
}
bool zfft_supported(int Nz);
// all three components: [3 Nx] planes of [Ny][Nzp]; tw[m] = exp(-2 pi i m / Ny)
void launch_yfft(double2 *spectra, DGrid G, bool inverse, const double2 *tw, hipStream_t s, int kb) {
    FftPlanX pl;
    plan_x(G.Ny, pl);
    const int nplanes = 3 * G.nxl;
    const size_t ps = (size_t)G.Ny * G.Nzp;
    if (G.Ny == 256) { launch_yfft_regs<256, 8, 8, 8, 44, 5, 359>(spectra, tw, nplanes, G.Nzh, G.Nzp, ps, inverse, s); return; }   // (four columns: +3 %)
    if (G.Ny == 512 && yfft_regs_supported(G.Ny, G.Nz, zfft_supported(G.Nz))) { launch_yfft_regs<512, 8, 8, 4, 72, 9, 578>(spectra, tw, nplanes, G.Nzh, G.Nzp, ps, inverse, s); return; }
    if (kb == 8) launch_fft_cols<8, 256>(spectra, pl, tw, nplanes, G.Nzh, G.Nzp, ps, inverse, s);
    else if (kb == 2) launch_fft_cols<2, 256>(spectra, pl, tw, nplanes, G.Nzh, G.Nzp, ps, inverse, s);
    else if (kb == 4 && G.Ny == 360) launch_fft_cols<4, 256, CtPlan<360, 9, 8, 5>>(spectra, pl, tw, nplanes, G.Nzh, G.Nzp, ps, inverse, s);   // compile-time plans, as in the x pass
    else if (kb == 4 && G.Ny == 270) launch_fft_cols<4, 256, CtPlan<270, 9, 5, 3, 2>>(spectra, pl, tw, nplanes, G.Nzh, G.Nzp, ps, inverse, s);
    else if (kb == 4 && G.Ny == 375) launch_fft_cols<4, 256, CtPlan<375, 5, 5, 5, 3>>(spectra, pl, tw, nplanes, G.Nzh, G.Nzp, ps, inverse, s);
    else if (kb == 4 && G.Ny == 500) launch_fft_cols<4, 256, CtPlan<500, 5, 5, 5, 4>>(spectra, pl, tw, nplanes, G.Nzh, G.Nzp, ps, inverse, s);
    else launch_fft_cols<4, 256>(spectra, pl, tw, nplanes, G.Nzh, G.Nzp, ps, inverse, s);
}

// ---- the z pass: real rows <-> half spectra, one wavefront per row (round 5) -----------------------------------------------------
// rocFFT's 1-D real transforms take two kernels per direction (a complex transform of half the length + an r2c / c2r step) and,
// at 512 points, run well below the rate of its 2-D plan.  Here a wavefront owns a row: its N reals are N / 2 = R0 x 8 x 8 complex
// points z[n] = x[2n] + i x[2n + 1], lane l holds z[l + 64 r] (coalesced 16-byte loads of the row as it lies), three register
// stages with two trips through the wave's own LDS column (no workgroup barrier anywhere), then the real <-> half-spectrum step on
// the natural order and coalesced stores: the row crosses HBM once each way.  Unnormalised both ways, like rocFFT's real plans;
// tools/debug/zfft_model.py is the index algebra in NumPy, checked against numpy.fft.
//   forward:  X[k] = (Z[k] + conj Z[NC - k]) / 2 - i/2 W_N^k (Z[k] - conj Z[NC - k]),  k = 0 .. NC      (Z[NC] = Z[0])
//   inverse:  Z[k] = (X[k] + conj X[NC - k]) + i conj W_N^k (X[k] - conj X[NC - k]),   k = 0 .. NC - 1;  x = N x the true inverse
struct ZRows { double *real[3]; double2 *spec[3]; int rows; int Nz, Nzp; };   // `rows` rows per component, consecutive in both arrays
// A wavefront transforms RW = 64 / (8 R0) rows AT ONCE -- stages 2 and 3 are 8 R0 radix-8 butterflies per row, and with one row per
// wave three quarters of the lanes (256 points: half) did arithmetic for nothing: the kernel ran at 3.3 TB/s bound by vector
// instructions, not by memory -- and RPW such groups one after the other with the twiddles of its lanes loaded once.
template <int NC, int R0, bool INVERSE, int RPW>
__global__ void __launch_bounds__(256)
k_zfft_rows(ZRows zr, const double2 *__restrict__ tw /* exp(-2 pi i m / N), m < N = 2 NC */) {
    // a row's column in LDS: the stage layouts and, once stage 3 has read its points, the natural order on top of them
    constexpr int P0 = 72, P1 = 9, CSW = P0 * (R0 - 1) + P1 * 7 + 8, WB = CSW > NC ? CSW : NC, RW = 64 / (8 * R0);
    static_assert(NC == R0 * 64 && (R0 == 2 || R0 == 4) && RPW % RW == 0, "NC = R0 x 8 x 8");
    __shared__ __attribute__((aligned(16))) double2 lds[4 * RW * WB];
    __shared__ __attribute__((aligned(16))) double2 tw2[64];   // W_64^{nn k1} at [nn * 8 + k1]
    typedef double d2v __attribute__((ext_vector_type(2)));
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (threadIdx.x < 64) { double2 t = tw[(2 * NC / 64) * (l >> 3) * (l & 7)]; if (INVERSE) t.y = -t.y; tw2[l] = t; }
    __syncthreads();
    const long total = 3L * zr.rows, row0 = ((long)blockIdx.x * 4 + wv) * RPW;
    if (row0 >= total) return;                                 // (whole waves; nothing below synchronises across waves)
    const int nrow = (int)min((long)RPW, total - row0);
    double2 *const col = lds + wv * RW * WB;                   // + j WB: row j of the group
    double2 t1[R0], tp[R0];                                     // W_NC^{l k0} (stage 1), W_N^{l + 64 q} (the real <-> half-spectrum step)
#pragma unroll
    for (int q = 0; q < R0; ++q) {
        t1[q] = tw[2 * l * q]; tp[q] = tw[l + 64 * q];
        if (INVERSE) { t1[q].y = -t1[q].y; tp[q].y = -tp[q].y; }
    }
    auto rowptrs = [&](long row, double *&xr, double2 *&xs) __attribute__((always_inline)) {
        const int c = (int)(row / zr.rows);
        const long r = row - (long)c * zr.rows;
        xr = zr.real[c] + r * zr.Nz; xs = zr.spec[c] + r * zr.Nzp;
    };
    // stages 2 and 3: lane -> (row of the group, k0, low index)
    const int jrow = l / (8 * R0), tl = l % (8 * R0), k0 = tl >> 3, lo = tl & 7;
    double2 *const mycol = col + jrow * WB;
    for (int i = 0; i < nrow; i += RW) {
        const int ng = min(RW, nrow - i);                       // rows of this group (the last may be short; wave-uniform)
        double2 a[RW][R0];
        if (!INVERSE) {
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                double *xr; double2 *xs;
                rowptrs(row0 + i + min(j, ng - 1), xr, xs);
#pragma unroll
                for (int q = 0; q < R0; ++q) { const d2v t = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(xr) + l + 64 * q); a[j][q] = make_double2(t.x, t.y); }
            }
        } else {
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                double *xr; double2 *xs;
                rowptrs(row0 + i + min(j, ng - 1), xr, xs);
#pragma unroll
                for (int q = 0; q < R0; ++q) {
                    const int k = l + 64 * q;                    // (k = 0 pairs with the Nyquist entry)
                    const d2v t = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(xs) + k), u = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(xs) + (NC - k));
                    const double2 sm = make_double2(t.x + u.x, t.y - u.y), df = make_double2(t.x - u.x, t.y + u.y);   // xk +- conj xc
                    const double2 w = cmul(tp[q], df);           // conj W_N^k (xk - conj xc)
                    a[j][q] = make_double2(sm.x - w.y, sm.y + w.x);   // sm + i w
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            dft_small<R0, INVERSE>(a[j]);                        // stage 1 over r -> k0, times W_NC^{l k0}
#pragma unroll
            for (int q = 1; q < R0; ++q) a[j][q] = cmul(a[j][q], t1[q]);
#pragma unroll
            for (int q = 0; q < R0; ++q) col[j * WB + P0 * q + l] = a[j][q];
        }
        __builtin_amdgcn_wave_barrier();
        double2 b[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) b[s] = mycol[P0 * k0 + lo + 8 * s];
        __builtin_amdgcn_wave_barrier();
        dft_small<8, INVERSE>(b);                                // stage 2 over s -> k1, times W_64^{nn k1}
#pragma unroll
        for (int k1 = 1; k1 < 8; ++k1) b[k1] = cmul(b[k1], tw2[lo * 8 + k1]);
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) mycol[P0 * k0 + P1 * k1 + lo] = b[k1];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n = 0; n < 8; ++n) b[n] = mycol[P0 * k0 + P1 * lo + n];   // lane (row, k0, k1 = lo)
        __builtin_amdgcn_wave_barrier();                         // (the natural order overwrites the stage layout)
        dft_small<8, INVERSE>(b);                                // stage 3 over nn -> k2: Z[k0 + R0 k1 + 8 R0 k2]
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) mycol[k0 + R0 * lo + 8 * R0 * k2] = b[k2];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            if (j >= ng) break;
            double *xr; double2 *xs;
            rowptrs(row0 + i + j, xr, xs);
            const double2 *nat = col + j * WB;
            if (!INVERSE) {
#pragma unroll
                for (int q = 0; q < R0; ++q) {
                    const int k = l + 64 * q;
                    const double2 zk = nat[k], zc0 = nat[(NC - k) & (NC - 1)];
                    const double2 sm = make_double2(zk.x + zc0.x, zk.y - zc0.y), df = make_double2(zk.x - zc0.x, zk.y + zc0.y);   // zk +- conj zc
                    const double2 t = cmul(tp[q], df);           // W_N^k (zk - conj zc)
                    d2v o; o.x = 0.5 * (sm.x + t.y); o.y = 0.5 * (sm.y - t.x);   // (sm - i t) / 2
                    __builtin_nontemporal_store(o, reinterpret_cast<d2v *>(xs) + k);
                }
                if (l == 0) { const double2 z0 = nat[0]; xs[NC] = make_double2(z0.x - z0.y, 0.0); }
            } else {
                d2v *z = reinterpret_cast<d2v *>(xr);
#pragma unroll
                for (int q = 0; q < R0; ++q) { const double2 t = nat[l + 64 * q]; d2v o; o.x = t.x; o.y = t.y; __builtin_nontemporal_store(o, z + l + 64 * q); }
            }
        }
        __builtin_amdgcn_wave_barrier();                         // (the next group's stage 1 overwrites the columns)
    }
}
constexpr int ZFFT_RPW = 8;
bool zfft_g_supported(int Nz);   // pse_zfft.hip: NC = R0 x R1 x R2 (360, 270, 180, ...)
void launch_zfft_g(double *const real[3], double2 *const spec[3], int rows, int Nz, int Nzp, bool inverse, const double2 *tw, hipStream_t s);
bool zfft_supported(int Nz) { return Nz == 256 || Nz == 512 || zfft_g_supported(Nz); }
void launch_zfft(double *const real[3], double2 *const spec[3], int rows, int Nz, int Nzp, bool inverse, const double2 *tw, hipStream_t s) {
    if (Nz != 256 && Nz != 512) { launch_zfft_g(real, spec, rows, Nz, Nzp, inverse, tw, s); return; }
    ZRows zr{};
    for (int c = 0; c < 3; ++c) { zr.real[c] = real[c]; zr.spec[c] = spec[c]; }
    zr.rows = rows; zr.Nz = Nz; zr.Nzp = Nzp;
    const dim3 g((unsigned)((3L * rows + 4 * ZFFT_RPW - 1) / (4 * ZFFT_RPW))), b(256);
    if (Nz == 512) {
        if (inverse) hipLaunchKernelGGL((k_zfft_rows<256, 4, true, ZFFT_RPW>), g, b, 0, s, zr, tw);
        else hipLaunchKernelGGL((k_zfft_rows<256, 4, false, ZFFT_RPW>), g, b, 0, s, zr, tw);
    } else {
        if (inverse) hipLaunchKernelGGL((k_zfft_rows<128, 2, true, ZFFT_RPW>), g, b, 0, s, zr, tw);
        else hipLaunchKernelGGL((k_zfft_rows<128, 2, false, ZFFT_RPW>), g, b, 0, s, zr, tw);
    }
}

bool xfuse_supported(int Nx) {   // 2^a 3^b 5^c, 16..512
    FftPlanX pl;
    return Nx >= 16 && Nx <= 512 && plan_x(Nx, pl);
}

template <int LOGN, int KB, int NTH>
static void launch_xfft_t(double2 *X, double2 *Y, double2 *Z, DGrid G, DBox box, ScaleArgs a, const double2 *tw, hipStream_t s) {
    constexpr int N = 1 << LOGN;
    const size_t lds = (size_t)(3 * KB * (N + 1) + N) * sizeof(double2);
    static LdsAttr attr;
    if (attr.need(lds)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_xfft_scale<LOGN, KB, NTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int nkb = (G.Nzh + KB - 1) / KB;
    const int rows = a.transposed ? a.nyl : G.Ny;
    hipLaunchKernelGGL((k_xfft_scale<LOGN, KB, NTH>), dim3(rows * nkb), dim3(NTH), lds, s, X, Y, Z, G, box, a, tw);
}

void launch_xfft_scale(double2 *X, double2 *Y, double2 *Z, DGrid G, DBox box, ScaleArgs a, const double2 *tw, hipStream_t s) {
    // (k_xfft_scale_cols addresses a component with 32-bit byte offsets: 512 x 512 x 264 complex numbers, the largest grid, are 1.1 GB)
    if (G.Nx & (G.Nx - 1)) {   // not a power of two: mixed-radix passes; 4 kz columns per workgroup while two buffers fit
        FftPlanX pl;
        plan_x(G.Nx, pl);
        // (one kz column per workgroup, 128 or 256 threads: 1.34 / 1.35 ms at 360^3 against 1.34 with two columns and 256 threads; two
        // columns with 128 / 192 threads: 1.90 / 1.52 ms -- the pass is bound by instruction issue, not occupancy; variants removed)
        if (!a.runtime_plan) {   // compile-time plans for the sizes of the reference's rule at the BASELINE configurations (PSE_XMIX=1: runtime plan)
            switch (G.Nx) {
                case 360:   // 0.80 ms at 360^3 (10 x 6 x 6, no spills at two waves per SIMD; three waves: 66 spilled, 0.97); the passes in LDS: 1.05
                    launch_xfft_cols<360, 10, 6, 4, 2, false, 54, 9, 538>(X, Y, Z, G, box, a, tw, s);
                    return;
                case 270: launch_xfft_mixed<2, 256, CtPlan<270, 9, 5, 3, 2>>(X, Y, Z, G, box, a, tw, pl, s); return;
                case 375: launch_xfft_mixed<2, 256, CtPlan<375, 5, 5, 5, 3>>(X, Y, Z, G, box, a, tw, pl, s); return;
                case 500: launch_xfft_mixed<2, 256, CtPlan<500, 5, 5, 5, 4>>(X, Y, Z, G, box, a, tw, pl, s); return;
                case 180: launch_xfft_mixed<4, 256, CtPlan<180, 9, 5, 4>>(X, Y, Z, G, box, a, tw, pl, s); return;
                default: break;
            }
        }
        if (G.Nx <= 200) launch_xfft_mixed<4, 256>(X, Y, Z, G, box, a, tw, pl, s);
        else launch_xfft_mixed<2, 256>(X, Y, Z, G, box, a, tw, pl, s);   // four columns, 512 threads, one workgroup per CU: 1.61 against 1.38 ms at 360^3
        return;
    }
    // small grids (BASELINE config 2: 64^3): with eight kz columns per workgroup the launch is a single, partly filled wave of
    // workgroups and its duration is one workgroup's latency chain (34 us at 64^3); two columns per workgroup give four times as
    // many, shorter chains (the 32-byte pieces of a line are fetched by neighbouring workgroups of one XCD)
    const int rows_ = a.transposed ? a.nyl : G.Ny;
    if (G.Nx <= 128 && (long)rows_ * ((G.Nzh + 7) / 8) < 1024 && !a.wide_small) {
        switch (G.Nx) {
            case 16: launch_xfft_t<4, 2, 64>(X, Y, Z, G, box, a, tw, s); return;
            case 32: launch_xfft_t<5, 2, 64>(X, Y, Z, G, box, a, tw, s); return;
            case 64: launch_xfft_t<6, 2, 128>(X, Y, Z, G, box, a, tw, s); return;
            case 128: launch_xfft_t<7, 2, 256>(X, Y, Z, G, box, a, tw, s); return;
            default: break;
        }
    }
    switch (G.Nx) {   // 8 kz columns per workgroup = 128-byte pieces; LDS = 3*KB*(N+1)*16 B
        case 16: launch_xfft_t<4, 8, 256>(X, Y, Z, G, box, a, tw, s); break;
        case 32: launch_xfft_t<5, 8, 256>(X, Y, Z, G, box, a, tw, s); break;
        case 64: launch_xfft_t<6, 8, 256>(X, Y, Z, G, box, a, tw, s); break;
        case 128: launch_xfft_t<7, 8, 512>(X, Y, Z, G, box, a, tw, s); break;
        // 64-byte pieces (four kz): 3.87 TB/s at 256 x 512 x 512; 32-byte pieces (two kz, three workgroups per CU) 2.73; 128-byte pieces with
        // all three components in LDS (eight kz, one workgroup per CU) 2.71
        case 256:
            // 256 = 4 x 8 x 8 with TWO columns per wave: eight kz columns = 128-byte pieces at four waves per workgroup; 256^3: 0.201 against
            // 0.226 ms, with noise 0.214 against 0.243 (256 = 8 x 8 x 4 with one column per wave and 64-byte pieces only tied: 0.25 - 0.27)
            launch_xfft_cols<256, 4, 8, 8, 3, true, 72, 9, 295, 2>(X, Y, Z, G, box, a, tw, s);
            break;
        // 512: radix 16, 16, 2; two kz columns: three workgroups per CU (3.3 ms at 512^3; four columns, one workgroup: 3.8; radix 4/2 in LDS: 5.2)
        default:
            // one component in LDS at a time, four kz columns, three workgroups per CU: 1.50 ms at 512^3 (4.3 TB/s); eight columns at four
            // waves per SIMD (128 registers: 225 spilled) 3.3 ms, six columns 2.9 ms, four columns without parking the third component
            // during the operator (107 spilled) 2.27 ms; all three components in LDS, two columns (k_xfft_scale256, removed in round 5): 2.77 ms
            launch_xfft_cols<512, 8, 8, 4, 3, true, 72, 9, 578>(X, Y, Z, G, box, a, tw, s);
            break;
    }
}

// ------------------------------------------------------------------------------------------------ slab transposes
// Slab-decomposed far field (new design; the reference is single-GPU, PSEv1/Stokes.cc:104).  After the local 2-D
// transforms rank r holds [3][nxl][Ny][Nzh]; the block of y rows [q*nyl, (q+1)*nyl) goes to rank q.  pack lays the
// half spectrum out as [3][G][nxl][nyl][Nzh] so each destination's block is contiguous; on the receiving side the blocks
// of all source ranks line up as [3][Nx][nyl][Nzh] (x-major) with no further copy.  unpack is the inverse map.
__global__ void k_slab_pack(const double2 *__restrict__ cgrid, double2 *__restrict__ buf, int nxl, int Ny, int Nzh,
                            int nyl, int unpack) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)nxl * Ny * Nzh;
    if (tid >= 3 * per) return;
    const int c = (int)(tid / per);
    size_t r = tid - (size_t)c * per;
    const int k = (int)(r % Nzh); r /= Nzh;
    const int j = (int)(r % Ny);
    const int lx = (int)(r / Ny);
    const int q = j / nyl, jl = j - q * nyl;
    const size_t o = (size_t)c * per + (((size_t)q * nxl + lx) * nyl + jl) * Nzh + k;
    if (unpack) ((double2 *)cgrid)[tid] = buf[o];
    else buf[o] = cgrid[tid];
}
void launch_slab_pack(double2 *cgrid, double2 *buf, int nxl, int Ny, int Nzh, int nyl, int unpack, hipStream_t s) {
    const size_t n = (size_t)3 * nxl * Ny * Nzh;
    hipLaunchKernelGGL(k_slab_pack, dim3(nblocks((long)n, TPB)), dim3(TPB), 0, s, cgrid, buf, nxl, Ny, Nzh, nyl, unpack);
}
__global__ void k_axpy_inplace(double *__restrict__ a, const double *__restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] += b[i];
}
__global__ void __launch_bounds__(TPB) k_copy_list(CopyList l) {
    const int it = blockIdx.y;
    const double *__restrict__ src = l.src[it];
    double *__restrict__ dst = l.dst[it];
    const unsigned n = l.cnt[it];
    if ((((size_t)src | (size_t)dst) & 15) == 0) {   // 16-byte pieces where both ends allow
        const unsigned n2 = n >> 1;
        for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < n2; i += gridDim.x * TPB) ((double2 *)dst)[i] = ((const double2 *)src)[i];
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = src[n - 1];
    } else {
        for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) dst[i] = src[i];
    }
}
void launch_copy_list(const CopyList &l, hipStream_t s) {
    if (l.n <= 0) return;
    unsigned mx = 0;
    for (int i = 0; i < l.n; ++i) mx = std::max(mx, l.cnt[i]);
    const int gx = std::max(1, std::min(256, nblocks((long)(mx / 2 + 1), TPB)));
    hipLaunchKernelGGL(k_copy_list, dim3(gx, l.n), dim3(TPB), 0, s, l);
}
void launch_add_inplace(double *a, const double *b, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_axpy_inplace, dim3(std::min<long>(2048, nblocks((long)n, TPB))), dim3(TPB), 0, s, a, b, n);
}

// ------------------------------------------------------------------------------------------------ vectors
// K14 gpu_stokes_BrownianGenerate_kernel (PSEv1/Brownian.cu:99-130), keyed by the particle's global index
__global__ void k_psi(double4 *__restrict__ psi_s, const unsigned *__restrict__ tag_s, int N, uint32_t seed,
                      uint32_t timestep, CellRanges need, const int *__restrict__ cell_off) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N || !need.row(s, cell_off)) return;
    uint32_t r[4];
    philox4x32(tag_s[s], 0u, timestep, DOMAIN_PARTICLE, seed, PHILOX_KEY1, r);
    const double q = 1.7320508075688772;  // sqrt(3): variance 1
    psi_s[s] = make_double4(uniform_pm(r[0], q), uniform_pm(r[1], q), uniform_pm(r[2], q), 0.0);
}
void launch_psi(double4 *psi_s, const unsigned *tag_s, int N, uint32_t seed, uint32_t timestep, hipStream_t s, CellRanges need,
                const int *cell_off) {
    hipLaunchKernelGGL(k_psi, dim3(nblocks(N, TPB)), dim3(TPB), 0, s, psi_s, tag_s, N, seed, timestep, need, cell_off);
}

static inline int vec_grid(int N) { return std::min(LZ_NPART, std::max(1, nblocks(N, TPB))); }

// Lanczos (PSEv1/Brownian.cu:440-521) with device-resident scalars: no host round trip inside an iteration.
// ---- Lanczos iteration (PSEv1/Brownian.cu:440-521) with deferred normalisation --------------------------------------
// Each rank owns rows [lo, hi) of every vector (a single GPU owns them all).  The mat-vec runs on the UNNORMALISED vector
// x_j (v_j = x_j / beta_j, beta_j = |x_j|): with y = M x_j the three sums  s1 = x_j.x_j, s2 = x_j.y, s3 = x_j.x_{j-1}
// give  beta_j = sqrt(s1)  and  alpha_j = v_j.(M v_j - beta_j v_{j-1}) = s2/s1 - s3/beta_{j-1}  -- the reference's K10-K12
// sequence (w = M v - beta v_prev; alpha = v.w; w -= alpha v; beta' = |w|) with ONE reduction per iteration (one 3-scalar
// all-reduce when sharded) and one vector pass: x_{j+1} = (y - alpha_j x_j)/beta_j - (beta_j/beta_{j-1}) x_{j-1}.
// The normalised v_j are never stored (round 3): the basis keeps the x_j and the final combination divides by beta_j -- one
// 32-byte write per row and iteration less.  beta_0 = |psi| is the norm the result is rescaled with (Brownian.cu:440-452,739).
__global__ void __launch_bounds__(TPB)
k_lz_dots(const double4 *__restrict__ x, const double4 *__restrict__ y, const double4 *__restrict__ vprev, int lo, int hi,
          double *__restrict__ partials, int cap) {
    __shared__ double sh[4];
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = lo + blockIdx.x * TPB + threadIdx.x; i < hi; i += gridDim.x * TPB) {
        const double4 p = x[i];
        a += p.x * p.x + p.y * p.y + p.z * p.z;
        if (y) { const double4 q = y[i]; b += p.x * q.x + p.y * q.y + p.z * q.z; }
        if (vprev) { const double4 m = vprev[i]; c += p.x * m.x + p.y * m.y + p.z * m.z; }
    }
    a = block_sum(a, sh);
    __syncthreads();
    b = block_sum(b, sh);
    __syncthreads();
    c = block_sum(c, sh);
    if (threadIdx.x == 0) { partials[blockIdx.x] = a; partials[cap + blockIdx.x] = b; partials[2 * cap + blockIdx.x] = c; }
}
// this rank's partial sums -> scal[LZ_TMP .. LZ_TMP + nsum) (then all-reduced over the ranks)
__global__ void __launch_bounds__(1024)
k_lz_reduce(const double *__restrict__ partials, int npart, int cap, int nsum, double *__restrict__ scal, const int *__restrict__ stop) {
    if (stop && *stop) return;
    __shared__ double sh[16];
    const int q = blockIdx.x;                         // one workgroup per sum, fixed summation order
    const double *p = partials + (size_t)q * cap;
    double v = 0.0;
    for (int i = threadIdx.x; i < npart; i += 1024) v += p[i];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += sh[w];
        scal[LZ_TMP + q] = t;
    }
}
// alpha_j, beta_j from the reduced sums; x_{j+1} on the given rows
__global__ void __launch_bounds__(TPB)
k_lz_update(const double4 *__restrict__ xin, const double4 *__restrict__ y, const double4 *__restrict__ xprev,
            double4 *__restrict__ xnext, int j, double *__restrict__ scal, RowRanges rg,
            const double *__restrict__ sums_all, int nranks, double *__restrict__ sch, const int *__restrict__ stop, vq4 *__restrict__ xq) {
    if (stop && *stop) return;
    // the three sums: this GPU's (single GPU), or the ranks' partial sums added in rank order -- every rank holds all of them
    // (they travel with the ghost rows: no separate all-reduce) and adds them in the same order: identical scalars everywhere
    double s1 = 0.0, s2 = 0.0, s3raw = 0.0;
    if (nranks > 0) for (int r = 0; r < nranks; ++r) { s1 += sums_all[r * LZ_NGRAM]; s2 += sums_all[r * LZ_NGRAM + 1]; s3raw += sums_all[r * LZ_NGRAM + 2]; }
    else { s1 = scal[LZ_TMP]; s2 = scal[LZ_TMP + 1]; s3raw = scal[LZ_TMP + 2]; }
    const double bprev = j == 1 ? scal[LZ_NORM] : (j > 1 ? scal[LZ_BETA + j - 1] : 0.0);   // |x_{j-1}|
    const double ibprev = bprev > 0.0 ? 1.0 / bprev : 0.0;
    const double beta = s1 > 0.0 ? sqrt(s1) : 0.0;
    const double inv = beta > 0.0 ? 1.0 / beta : 0.0;
    const double alpha = s1 > 0.0 ? s2 / s1 - s3raw * ibprev : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scal[LZ_ALPHA + j] = alpha;
        if (j == 0) { scal[LZ_NORM] = beta; scal[LZ_BETA] = 0.0; } else scal[LZ_BETA + j] = beta;
        if (sch) {   // the host's copy (mapped pinned memory): what the convergence check reads -- no device-to-host copy is queued
            sch[LZ_ALPHA + j] = alpha;
            if (j == 0) { sch[LZ_NORM] = beta; sch[LZ_BETA] = 0.0; } else sch[LZ_BETA + j] = beta;
        }
    }
    const double cp = beta * ibprev;   // beta_j / beta_{j-1}
    const int n0 = rg.hi[0] - rg.lo[0], n1 = rg.n > 1 ? rg.hi[1] - rg.lo[1] : 0, n2 = rg.n > 2 ? rg.hi[2] - rg.lo[2] : 0;
    for (int t = blockIdx.x * TPB + threadIdx.x; t < n0 + n1 + n2; t += gridDim.x * TPB) {
        const int i = t < n0 ? rg.lo[0] + t : (t < n0 + n1 ? rg.lo[1] + (t - n0) : rg.lo[2] + (t - n0 - n1));
        const double4 p = xin[i], q = y[i];
        double nx = (q.x - alpha * p.x) * inv, ny = (q.y - alpha * p.y) * inv, nz = (q.z - alpha * p.z) * inv;
        if (xprev) { const double4 m = xprev[i]; nx -= cp * m.x; ny -= cp * m.y; nz -= cp * m.z; }
        xnext[i] = make_double4(nx, ny, nz, 0.0);
        if (xq) xq[i] = vq_pack(nx, ny, nz);   // the 16-byte mirror the next pair-list mat-vec gathers its neighbours from
    }
}
// ---- two Lanczos iterations per exchange (teams) ---------------------------------------------------------------------------
// A slab rank pays one exchange (an all-reduce of the sums + the ghost rows of the mat-vec results) per Lanczos iteration, and at
// 1/8 of the rows an exchange costs as much as the mat-vec.  The two-step block needs ONE per two iterations: with q = v_j,
// p = v_{j-1}, u = M p known on the own rows and TWO ghost cell layers,
//     w1 = M q   on the own rows and one ghost layer (redundantly: the neighbours compute those rows too),
//     w2 = M w1  on the own rows,
// the Gram sums of {p, q, u, w1, w2} over the own rows (one all-reduce of LZ_NGRAM numbers) give alpha_j, beta_{j+1}, alpha_{j+1},
// beta_{j+2} in closed form (M is symmetric: q.w2 = w1.w1, p.w2 = u.w1, ...; every product is summed explicitly all the same):
//     r1 = w1 - alpha q - beta p,            beta'^2 = w1.w1 - alpha^2 - 2 beta p.w1 + beta^2,        v' = r1 / beta'
//     z1 = M v' = (w2 - alpha w1 - beta u) / beta',   alpha' = v'.z1,
//     r2 = z1 - alpha' v' - beta' q,         beta''^2 = z1.z1 - alpha'^2 - 2 beta' q.z1 + beta'^2,    v'' = r2 / beta''
// and with the ghost rows of w1, w2 (the same exchange) every rank forms v', v'', z1 on its own AND its ghost rows -- the state
// of the next block (p = v', q = v'', u = z1).  In exact arithmetic these are the alpha, beta of PSEv1/Brownian.cu:440-521; the
// sums cost a relative 1e-14 in beta (cancellation against O(1) terms).  q of block 0 is psi unnormalised: s = 1 / |q| from q.q.
// The basis V holds the NORMALISED v_j here (the one-step path keeps unnormalised x_j).
template <bool FULL>
__global__ void __launch_bounds__(TPB)
k_lz_block(LzBlockArgs a, double *__restrict__ scal, RowRanges rg, const double *__restrict__ sums_all, int nranks, double *__restrict__ sch,
           const RowRanges *__restrict__ rg_dev, int vectors_off, const int *__restrict__ stop) {
    if (stop && *stop) return;
    if (rg_dev) { rg = *rg_dev; if (vectors_off) rg.n = 0; }   // an owned-particle rank: the ranges are known on the device only
    __shared__ double sG[LZ_NGRAM];   // the ranks' partial sums added in rank order (they came with the ghost rows: see k_lz_update):
    if (threadIdx.x < LZ_NGRAM) {     // one lane per sum, once per workgroup
        double v = 0.0;
        for (int r = 0; r < nranks; ++r) v += sums_all[r * LZ_NGRAM + threadIdx.x];
        sG[threadIdx.x] = nranks > 0 ? v : scal[LZ_TMP + threadIdx.x];
    }
    __syncthreads();
    double G[LZ_NGRAM];
#pragma unroll
    for (int t = 0; t < LZ_NGRAM; ++t) G[t] = sG[t];
    const int j = a.j;
    const double n0 = G[LZG_QQ], s2 = n0 > 0.0 ? 1.0 / n0 : 0.0, sc = sqrt(s2);
    const double beta = j > 0 ? scal[LZ_BETA + j] : 0.0, alpha_prev = j > 0 ? scal[LZ_ALPHA + j - 1] : 0.0;
    const double ga = G[LZG_QW1] * s2, gb = G[LZG_W1W1] * s2, gf = G[LZG_PW1] * sc;   // q.w1, w1.w1, p.w1 (= q.u) of the normalised q
    const double alpha = ga;
    const double bp2 = gb - alpha * alpha - 2.0 * beta * gf + beta * beta;
    const double bp = bp2 > 0.0 ? sqrt(bp2) : 0.0, ibp = bp > 1e-12 ? 1.0 / bp : 0.0;
    double alpha1 = 0.0, bpp = 0.0, ibpp = 0.0, uu_next = 0.0;
    if (FULL) {
        const double gc = G[LZG_W1W2] * s2, gd = G[LZG_W2W2] * s2, ge = gb /* q.w2 = w1.w1 */, gg = G[LZG_PW2] * sc, gg2 = gg /* u.w1 = p.w2 */,
                     gh = G[LZG_UW2] * sc, gk = j > 0 ? scal[LZ_UU + j] : 0.0, gf2 = gf /* q.u = p.w1 */;
        const double r1w2 = gc - alpha * ge - beta * gg, r1w1 = gb - alpha * ga - beta * gf, r1u = gg2 - alpha * gf2 - beta * alpha_prev;
        alpha1 = (r1w2 - alpha * r1w1 - beta * r1u) * ibp * ibp;
        const double zz = (gd + alpha * alpha * gb + beta * beta * gk - 2.0 * alpha * gc - 2.0 * beta * gh + 2.0 * alpha * beta * gg2) * ibp * ibp;
        const double qz = (ge - alpha * ga - beta * gf2) * ibp;
        const double bpp2 = zz - alpha1 * alpha1 - 2.0 * bp * qz + bp * bp;
        bpp = bpp2 > 0.0 ? sqrt(bpp2) : 0.0;
        ibpp = bpp > 1e-12 ? 1.0 / bpp : 0.0;
        uu_next = zz;                              // |M v_{j+1}|^2: the u.u of the block that starts at j + 2
    } else {
        uu_next = gb;                              // a single step leaves u = M v_j = w1
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (j == 0) { scal[LZ_NORM] = n0 > 0.0 ? sqrt(n0) : 0.0; scal[LZ_BETA] = 0.0; }
        scal[LZ_ALPHA + j] = alpha;
        scal[LZ_BETA + j + 1] = bp;
        if (FULL) { scal[LZ_ALPHA + j + 1] = alpha1; scal[LZ_BETA + j + 2] = bpp; }
        scal[LZ_UU + j + (FULL ? 2 : 1)] = uu_next;   // another slot than the one this launch reads
        if (sch) {   // the host's copy (mapped pinned memory)
            if (j == 0) { sch[LZ_NORM] = n0 > 0.0 ? sqrt(n0) : 0.0; sch[LZ_BETA] = 0.0; }
            sch[LZ_ALPHA + j] = alpha;
            sch[LZ_BETA + j + 1] = bp;
            if (FULL) { sch[LZ_ALPHA + j + 1] = alpha1; sch[LZ_BETA + j + 2] = bpp; }
        }
    }
    const int n0r = rg.n > 0 ? rg.hi[0] - rg.lo[0] : 0, n1r = rg.n > 1 ? rg.hi[1] - rg.lo[1] : 0, n2r = rg.n > 2 ? rg.hi[2] - rg.lo[2] : 0;
    for (int t = blockIdx.x * TPB + threadIdx.x; t < n0r + n1r + n2r; t += gridDim.x * TPB) {
        const int i = t < n0r ? rg.lo[0] + t : (t < n0r + n1r ? rg.lo[1] + (t - n0r) : rg.lo[2] + (t - n0r - n1r));
        const double4 q = a.q[i], w1 = a.w1[i];
        double4 p = make_double4(0.0, 0.0, 0.0, 0.0), u = p;
        if (j > 0) { p = a.p[i]; if (FULL) u = a.u[i]; }
        const double qx = sc * q.x, qy = sc * q.y, qz_ = sc * q.z, ax = sc * w1.x, ay = sc * w1.y, az = sc * w1.z;
        const double v1x = (ax - alpha * qx - beta * p.x) * ibp, v1y = (ay - alpha * qy - beta * p.y) * ibp, v1z = (az - alpha * qz_ - beta * p.z) * ibp;
        a.v1[i] = make_double4(v1x, v1y, v1z, 0.0);
        double nx = v1x, ny = v1y, nz = v1z;
        if (FULL) {
            const double4 w2 = a.w2[i];
            const double z1x = (sc * w2.x - alpha * ax - beta * u.x) * ibp, z1y = (sc * w2.y - alpha * ay - beta * u.y) * ibp,
                         z1z = (sc * w2.z - alpha * az - beta * u.z) * ibp;
            nx = (z1x - alpha1 * v1x - bp * qx) * ibpp; ny = (z1y - alpha1 * v1y - bp * qy) * ibpp; nz = (z1z - alpha1 * v1z - bp * qz_) * ibpp;
            a.v2[i] = make_double4(nx, ny, nz, 0.0);
            a.u[i] = make_double4(z1x, z1y, z1z, 0.0);
        } else {
            a.u[i] = make_double4(ax, ay, az, 0.0);   // M v_j: the u of a block that starts at j + 1
        }
    }
}
void launch_lz_block(const LzBlockArgs &a, bool full, double *scal, const int (*rg)[2], int nrg, hipStream_t s, const double *sums_all, int nranks, double *sch,
                     const RowRanges *rg_dev, int rows_cap, bool vectors_off, const int *stop) {
    RowRanges r{};
    r.n = nrg;
    int total = 0;
    for (int q = 0; q < nrg && q < 3; ++q) { r.lo[q] = rg[q][0]; r.hi[q] = rg[q][1]; total += rg[q][1] - rg[q][0]; }
    if (rg_dev) total = vectors_off ? 1 : rows_cap;
    const dim3 g(vec_grid(std::max(1, total)));
    if (full) hipLaunchKernelGGL(k_lz_block<true>, g, dim3(TPB), 0, s, a, scal, r, sums_all, nranks, sch, rg_dev, vectors_off ? 1 : 0, stop);
    else hipLaunchKernelGGL(k_lz_block<false>, g, dim3(TPB), 0, s, a, scal, r, sums_all, nranks, sch, rg_dev, vectors_off ? 1 : 0, stop);
}


// ---- the Lanczos decision on the device (LzState / LzDecide in pse_kernels.h) ------------------------------------------------------
// One wavefront: eigen-decomposition of the m x m tridiagonal (alpha[0..m), beta[1..m)) by implicit QL with Wilkinson shifts -- the
// algorithm of tridiag_eigen (pse_params.cpp; it replaces LAPACKE_spteqr, PSEv1/Brownian.cu:540), statement by statement -- and
// t = Z sqrt(Lambda) Z^T e_1 (PSEv1/Brownian.cu:563-582).  The scalar recurrence runs redundantly in every lane (identical values);
// lane 0 stores d and e; row k of the eigenvector matrix belongs to lane k mod 64 (columns in LDS, stride m).  m <= 2 x 64.
__device__ __forceinline__ void lz_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// d and e live in REGISTERS, entry i in lane i mod 64 of one of two registers (m <= 128): the recurrence reads them with v_readlane
// at a wave-uniform index and the owning lane stores -- no LDS round trip on the serial chain (with d, e in LDS the decision of
// sizes 6 and 7 took 21 us, most of it waiting for ds_read).  Only the eigenvector columns are in LDS; their update is off the chain.
struct LzReg {
    double lo, hi;   // entries lane and 64 + lane
    __device__ __forceinline__ double get(int i) const {
        const double v = i < 64 ? lo : hi;
        const int l = i & 63;
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
    }
    __device__ __forceinline__ void set(int i, double v) {
        const int lane = threadIdx.x & 63;
        if (lane == (i & 63)) { if (i < 64) lo = v; else hi = v; }
    }
};
__device__ bool lz_sqrt_e1(int m, const double *__restrict__ alpha, const double *__restrict__ beta, double *z, double *t_out) {
    const int lane = threadIdx.x & 63;
    LzReg d, e;
    d.lo = lane < m ? alpha[lane] : 0.0; d.hi = lane + 64 < m ? alpha[lane + 64] : 0.0;
    e.lo = lane + 1 < m ? beta[lane + 1] : 0.0; e.hi = lane + 65 < m ? beta[lane + 65] : 0.0;
    for (int c = 0; c < m; ++c)
        for (int k = lane; k < m; k += 64) z[c * m + k] = c == k ? 1.0 : 0.0;
    bool ok = true;
    for (int l = 0; l < m && ok; ++l) {
        int iter = 0, mm;
        do {
            for (mm = l; mm < m - 1; ++mm) {
                const double dd = fabs(d.get(mm)) + fabs(d.get(mm + 1));
                if (fabs(e.get(mm)) <= 2.3e-16 * dd) break;
            }
            if (mm != l) {
                if (iter++ == 200) { ok = false; break; }
                const double dl = d.get(l), el = e.get(l);
                double g = (d.get(l + 1) - dl) / (2.0 * el);
                double r = sqrt(g * g + 1.0);
                g = d.get(mm) - dl + el / (g + (g >= 0 ? r : -r));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = mm - 1; i >= l; --i) {
                    const double ei = e.get(i);
                    double f = s * ei, b = c * ei;
                    r = sqrt(f * f + g * g);
                    e.set(i + 1, r);
                    if (r == 0.0) { d.set(i + 1, d.get(i + 1) - p); e.set(mm, 0.0); break; }
                    const double ir = 1.0 / r;
                    s = f * ir; c = g * ir;
                    g = d.get(i + 1) - p;
                    r = (d.get(i) - g) * s + 2.0 * c * b;
                    p = s * r;
                    d.set(i + 1, g + p);
                    g = c * r - b;
                    for (int k = lane; k < m; k += 64) {
                        const double zk1 = z[(i + 1) * m + k], zk0 = z[i * m + k];
                        z[(i + 1) * m + k] = s * zk0 + c * zk1;
                        z[i * m + k] = c * zk0 - s * zk1;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d.set(l, d.get(l) - p); e.set(l, g); e.set(mm, 0.0);
            }
        } while (mm != l);
    }
    lz_wave_sync();
    if (!ok) return false;
    for (int i = lane; i < m; i += 64) {
        double t = 0.0;
        for (int j = 0; j < m; ++j) t += z[j * m + i] * (sqrt(fmax(d.get(j), 0.0)) * z[j * m]);
        t_out[i] = t;
    }
    lz_wave_sync();
    return true;
}
__global__ void __launch_bounds__(128)
k_lz_decide(LzDecide a, double *__restrict__ scal, LzState *__restrict__ st, double *__restrict__ sch, double seq) {
    extern __shared__ double lds[];
    __shared__ double tbuf[2][104];
    __shared__ int okf[2];
    if (a.first) { if (threadIdx.x == 0) { st->done = 0; st->m_final = 0; st->checked = 0; st->status = 0; st->stepnorm = 1.0; } }
    else if (st->done) return;
    __syncthreads();
    const int wv = threadIdx.x >> 6, nm = a.m_hi - a.m_lo + 1;
    const double *alpha = scal + LZ_ALPHA, *beta = scal + LZ_BETA;
    const double norm = scal[LZ_NORM];
    const bool dead = !(norm > 0.0) || !isfinite(norm);          // psi == 0: the result is zero
    if (wv < nm && !dead) {
        const int m = a.m_lo + wv;
        double *base = lds + (wv == 0 ? 0 : (size_t)a.m_lo * a.m_lo);
        const bool ok = lz_sqrt_e1(m, alpha, beta, base, tbuf[wv]);
        if ((threadIdx.x & 63) == 0) okf[wv] = ok ? 1 : 0;
    }
    // what the walk below reads, fetched by all lanes at once: thread 0 alone paid a dependent global round trip per coefficient
    // (alpha, beta, the previous size's t: ~20 of them in a row, a third of the kernel's 24-28 us)
    __shared__ double s_alpha[104], s_beta[104], s_prev[104];
    __shared__ int s_checked;
    __shared__ double s_stepnorm;
    for (int q = threadIdx.x; q <= min(a.m_hi, 103); q += blockDim.x) { s_alpha[q] = scal[LZ_ALPHA + q]; s_beta[q] = scal[LZ_BETA + q]; s_prev[q] = st->t_prev[q]; }
    if (threadIdx.x == 0) { s_checked = a.first ? 0 : st->checked; s_stepnorm = a.first ? 1.0 : st->stepnorm; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    alpha = s_alpha; beta = s_beta;
    int m_final = 0, status = 0, checked = s_checked;
    double stepnorm = s_stepnorm;
    const double *t_fin = nullptr;
    if (dead) { m_final = -1; }
    else if (a.pending_beta > 0 && beta[a.pending_beta] < 1e-8) {   // |x_m| of the vector the previous batch ended on: invariant subspace
        m_final = a.pending_beta; stepnorm = 0.0; t_fin = s_prev;
    } else {
        for (int w = 0; w < nm && !m_final; ++w) {               // walk m upward as the reference's while loop does (PSEv1/Brownian.cu:606-724)
            const int m = a.m_lo + w;
            const bool have_beta = m < a.done_iters || a.have_last_beta;
            if (!isfinite(alpha[m - 1]) || (have_beta && !isfinite(beta[m])) || !okf[w]) { m_final = max(checked, 1); status = 2; t_fin = checked ? s_prev : tbuf[w]; break; }
            const double *tc = tbuf[w];
            if (m < a.done_iters && beta[m] < 1e-8) { m_final = m; stepnorm = 0.0; t_fin = tc; break; }   // invariant subspace (Brownian.cu:503)
            if (checked == m - 1 && checked > 0) {
                double s2 = tc[m - 1] * tc[m - 1];
                for (int q = 0; q < m - 1; ++q) { const double dq = tc[q] - s_prev[q]; s2 += dq * dq; }
                stepnorm = sqrt(s2 / alpha[0]);                   // Brownian.cu:719-724; psi.M.psi / |psi|^2 = alpha_0
                if (stepnorm <= a.tol || m >= a.m_max) { m_final = m; t_fin = tc; break; }
            }
            if (a.have_last_beta && m == a.done_iters && beta[m] < 1e-8) { m_final = m; stepnorm = 0.0; t_fin = tc; break; }
            for (int q = 0; q < m; ++q) { st->t_prev[q] = tc[q]; s_prev[q] = tc[q]; }
            checked = m;
        }
        if (!m_final && (a.last || a.done_iters >= a.m_max)) {     // nothing more is queued: end with what there is
            m_final = checked; t_fin = s_prev; status = a.done_iters >= a.m_max ? 0 : 1;
        }
    }
    st->checked = checked;
    st->stepnorm = stepnorm;
    if (m_final) {
        const int m = max(m_final, 0);
        for (int q = 0; q < m; ++q) st->coef[q] = t_fin[q] / (q == 0 ? norm : (a.normalised ? 1.0 : beta[q]));
        st->m_final = m; st->status = status;
        st->done = 1;                                            // (read by LATER launches only: the kernel boundary orders it)
        if (sch) {
            sch[LZ_HOST_M] = (double)m; sch[LZ_HOST_STEPNORM] = stepnorm; sch[LZ_HOST_STATUS] = (double)status; sch[LZ_HOST_SEQ] = seq;
            if (status != 0) sch[LZ_HOST_OPEN] = sch[LZ_HOST_OPEN] + 1.0;   // (sticky: a loop that reads pse_info only now and then still learns of it)
        }
    }
}
static size_t lz_decide_lds(int m_lo, int m_hi) {
    size_t n = (size_t)m_lo * m_lo;
    if (m_hi != m_lo) n += (size_t)m_hi * m_hi;
    return n * sizeof(double);
}
bool lz_decide_supported(int m_hi) { return m_hi >= 1 && m_hi <= 100 && lz_decide_lds(std::max(1, m_hi - 1), m_hi) <= 150 * 1024; }
void launch_lz_decide(const LzDecide &d, double *scal, LzState *st, double *sch, double seq, hipStream_t s) {
    const size_t lds = lz_decide_lds(d.m_lo, d.m_hi);
    static LdsAttr attr;
    if (lds > 48 * 1024 && attr.need(150 * 1024)) (void)hipFuncSetAttribute((const void *)k_lz_decide, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(k_lz_decide, dim3(1), dim3(128), lds, s, d, scal, st, sch, seq);
}

void launch_lz_dots(const double4 *x, const double4 *y, const double4 *vprev, int lo, int hi, double *partials, int cap,
                    double *scal, hipStream_t s) {
    const int g = vec_grid(std::max(1, hi - lo));
    hipLaunchKernelGGL(k_lz_dots, dim3(g), dim3(TPB), 0, s, x, y, vprev, lo, hi, partials, cap);
    hipLaunchKernelGGL(k_lz_reduce, dim3(y ? 3 : 1), dim3(1024), 0, s, partials, g, cap, y ? 3 : 1, scal, nullptr);
}
void launch_lz_update(const double4 *xin, const double4 *y, const double4 *xprev, double4 *xnext, int j,
                      double *scal, const int (*rg)[2], int nrg, hipStream_t s, const double *sums_all, int nranks, double *sch,
                      const int *stop, void *xq) {
    RowRanges r{};
    r.n = nrg;
    int total = 0;
    for (int q = 0; q < nrg && q < 3; ++q) { r.lo[q] = rg[q][0]; r.hi[q] = rg[q][1]; total += rg[q][1] - rg[q][0]; }
    hipLaunchKernelGGL(k_lz_update, dim3(vec_grid(std::max(1, total))), dim3(TPB), 0, s, xin, y, xprev, xnext, j, scal, r, sums_all, nranks, sch, stop, (vq4 *)xq);
}
__global__ void k_vq_roundtrip(const double *__restrict__ in, double *__restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x, y, z;
    vq_unpack(vq_pack(in[3 * i], in[3 * i + 1], in[3 * i + 2]), x, y, z);
    out[3 * i] = x; out[3 * i + 1] = y; out[3 * i + 2] = z;
}
void launch_vq_roundtrip(const double *in, double *out, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_vq_roundtrip, dim3(nblocks(n, TPB)), dim3(TPB), 0, s, in, out, n);
}
// out[i] = a[i] + b[i] + c[i] on rows [lo, hi)  (each may be null)
__global__ void k_sum_rows(const double4 *__restrict__ a, const double4 *__restrict__ b, const double4 *__restrict__ c,
                           double4 *__restrict__ out, int lo, int hi, const unsigned *__restrict__ tag_s) {
    const int i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    double x = 0, y = 0, z = 0;
    if (a) { const double4 v = a[i]; x += v.x; y += v.y; z += v.z; }
    if (b) { const double4 v = b[i]; x += v.x; y += v.y; z += v.z; }
    if (c) { const double4 v = c[i]; x += v.x; y += v.y; z += v.z; }
    out[i] = make_double4(x, y, z, tag_s ? (double)tag_s[i] : 0.0);   // an index below 2^32 is exact in a double
}
void launch_sum_rows(const double4 *a, const double4 *b, const double4 *c, double4 *out, int lo, int hi, hipStream_t s,
                     const unsigned *tag_s) {
    if (hi > lo) hipLaunchKernelGGL(k_sum_rows, dim3(nblocks(hi - lo, TPB)), dim3(TPB), 0, s, a, b, c, out, lo, hi, tag_s);
}
// row boundaries of the cell slabs: out[r] = cell_off[r * stride] for r = 0..n-1
__global__ void k_pick(const int *__restrict__ cell_off, const int *__restrict__ idx, int n, int *__restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) out[r] = cell_off[idx[r]];
}
void launch_pick(const int *cell_off, const int *idx, int n, int *out, hipStream_t s) {
    hipLaunchKernelGGL(k_pick, dim3(nblocks(n, TPB)), dim3(TPB), 0, s, cell_off, idx, n, out);
}

// K13 gpu_stokes_MatVecMultiply_kernel (PSEv1/Helper.cu:251-279) + the final rescale (PSEv1/Brownian.cu:739) -- and, round 4, what used to
// be the next launch: with a sink the Brownian part is added to the far-field and near-field velocities of the row on the fly and
// goes straight to where the step wants the sum (a team: the row of the all-gathered array, tag in .w; a single GPU: vel[tag].xyz),
// so ub_s is neither written nor read back (K10 gpu_stokes_LinearCombination_kernel, PSEv1/Helper.cu:113-133)
__global__ void __launch_bounds__(TPB)
k_basis_combine(const double4 *__restrict__ x0, const double4 *__restrict__ V, size_t stride, BasisCoef tc, int m,
                const double *__restrict__ scal, double scale, int use_norm, double4 *__restrict__ out, int lo, int N, CombineSink sink,
                const LzState *__restrict__ st) {
    const double *t = tc.t;   // kernel arguments: no host-to-device copy whose source the host would have to keep alive
    if (st) { m = st->m_final; t = st->coef; }   // ... or what the device-side decision left (queue-only calls)
    const double sc = use_norm ? scale * scal[LZ_NORM] : scale;
    for (int i = lo + blockIdx.x * TPB + threadIdx.x; i < N; i += gridDim.x * TPB) {
        double x = 0, y = 0, z = 0;
        for (int q = 0; q < m; ++q) {
            const double4 v = q == 0 ? x0[i] : V[(size_t)q * stride + i];   // x_0 = psi was never copied into the basis
            const double tq = t[q];
            x += tq * v.x; y += tq * v.y; z += tq * v.z;
        }
        x *= sc; y *= sc; z *= sc;
        if (!sink.on) { out[i] = make_double4(x, y, z, 0.0); continue; }
        // a + b + c in the order the separate pass added them: far field, near field, Brownian
        double sx = 0.0, sy = 0.0, sz = 0.0;
        if (sink.add_a) { const double4 v = sink.add_a[i]; sx += v.x; sy += v.y; sz += v.z; }
        if (sink.add_b) { const double4 v = sink.add_b[i]; sx += v.x; sy += v.y; sz += v.z; }
        sx += x; sy += y; sz += z;
        const unsigned idx = sink.tag_s[i];
        if (sink.vel) { double4 o = sink.vel[idx]; o.x = sx; o.y = sy; o.z = sz; sink.vel[idx] = o; }
        else sink.rows[i] = make_double4(sx, sy, sz, (double)idx);
    }
}
void launch_basis_combine(const double4 *x0, const double4 *V, size_t stride, const BasisCoef &t_dev, int m, const double *scal,
                          double scale, int use_norm, double4 *out_s, int lo, int hi, hipStream_t s, CombineSink sink, const LzState *st) {
    hipLaunchKernelGGL(k_basis_combine, dim3(std::min(2048, std::max(1, nblocks(hi - lo, TPB)))), dim3(TPB), 0, s, x0, V, stride,
                       t_dev, m, scal, scale, use_norm, out_s, lo, hi, sink, st);
}

// Force provider next to the path (the step consumes net_force, PSEv1/Stokes.cc:447): soft repulsion
// F_i = sum_j k (sigma - r) (r_i - r_j)/r over pairs closer than sigma, from the engine's own cell list.  One thread per
// particle; the result is added to (or stored in) the caller's force array in the caller's order.
__global__ void __launch_bounds__(TPB)
k_pair_repulsion(const double4 *__restrict__ pos_s, const unsigned *__restrict__ tag_s, int N, const int *__restrict__ cell_off,
                 DBox box, DCells nc, double k, double sigma, int accumulate, double4 *__restrict__ force) {
    const int i = xcd_block(blockIdx.x, gridDim.x) * TPB + threadIdx.x;
    if (i >= N) return;
    const double4 pi = pos_s[i];
    double fx, fy, fz;
    frac_coords(box, pi.x, pi.y, pi.z, fx, fy, fz);
    const int cx = cell_coord(fx, nc.nx), cy = cell_coord(fy, nc.ny), cz = cell_coord(fz, nc.nz);
    const double s2 = sigma * sigma;
    double Fx = 0.0, Fy = 0.0, Fz = 0.0;
    for_each_run(nc, cell_off, cx, cy, cz, [&](int jb, int je, unsigned) {
        for (int j = jb; j < je; ++j) {
            const double4 pj = pos_s[j];
            double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            min_image(box, dx, dy, dz);
            const double r2 = dx * dx + dy * dy + dz * dz;
            if (r2 < s2 && j != i && r2 > 0.0) {
                const double r = sqrt(r2), c = k * (sigma - r) / r;
                Fx += c * dx; Fy += c * dy; Fz += c * dz;
            }
        }
    });
    const unsigned idx = tag_s[i];
    double4 f = force[idx];
    if (accumulate) { f.x += Fx; f.y += Fy; f.z += Fz; } else { f.x = Fx; f.y = Fy; f.z = Fz; }
    force[idx] = f;
}
void launch_pair_repulsion(const double4 *pos_s, const unsigned *tag_s, int N, const int *cell_off, DBox box, DCells nc,
                           double k, double sigma, int accumulate, double4 *force, hipStream_t s) {
    hipLaunchKernelGGL(k_pair_repulsion, dim3(nblocks(N, TPB)), dim3(TPB), 0, s, pos_s, tag_s, N, cell_off, box, nc, k, sigma,
                       accumulate, force);
}

// K10 gpu_stokes_LinearCombination_kernel (PSEv1/Helper.cu:113-133) as the final un-sort: vel.xyz = a + b + c, keep w
__global__ void k_scatter_sum(const double4 *__restrict__ a, const double4 *__restrict__ b,
                              const double4 *__restrict__ c, const unsigned *__restrict__ tag_s, int N,
                              double4 *__restrict__ vel) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    double x = 0, y = 0, z = 0;
    if (a) { const double4 v = a[s]; x += v.x; y += v.y; z += v.z; }
    if (b) { const double4 v = b[s]; x += v.x; y += v.y; z += v.z; }
    if (c) { const double4 v = c[s]; x += v.x; y += v.y; z += v.z; }
    const unsigned idx = tag_s ? tag_s[s] : (unsigned)a[s].w;
    double4 o = vel[idx];
    o.x = x; o.y = y; o.z = z;
    vel[idx] = o;
}
void launch_scatter_sum(const double4 *a, const double4 *b, const double4 *c, const unsigned *tag_s, int N,
                        double4 *vel, hipStream_t s) {
    hipLaunchKernelGGL(k_scatter_sum, dim3(nblocks(N, TPB)), dim3(TPB), 0, s, a, b, c, tag_s, N, vel);
}

// K15 gpu_stokes_step_one_kernel (PSEv1/Stokes.cu:137-192)
__global__ void k_integrate(double4 *__restrict__ pos, const double4 *__restrict__ vel, double3 *__restrict__ accel,
                            int3 *__restrict__ image, const double4 *__restrict__ force,
                            const unsigned *__restrict__ group, int N, DBox box, double dt, double shear_rate) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const unsigned idx = group ? group[g] : (unsigned)g;
    double4 p = pos[idx];
    const double4 v = vel[idx];
    const double4 F = force[idx];
    p.x += (v.x + shear_rate * p.y) * dt;   // Stokes.cu:164-171
    p.y += v.y * dt;
    p.z += v.z * dt;
    int3 im = image[idx];
    // triclinic wrap (HOOMD BoxDim::wrap): z, then y (a y image shifts x by xy*Ly), then x
    double n = floor(p.z * box.iLz + 0.5);
    p.z -= n * box.Lz; im.z += (int)n;
    n = floor(p.y * box.iLy + 0.5);
    p.y -= n * box.Ly; p.x -= n * box.xy * box.Ly; im.y += (int)n;
    n = floor((p.x - box.xy * p.y) * box.iLx + 0.5);
    p.x -= n * box.Lx; im.x += (int)n;
    const double im_ = 1.0 / v.w;            // mass in vel.w (Stokes.cu:160,174)
    accel[idx] = make_double3(F.x * im_, F.y * im_, F.z * im_);
    pos[idx] = p;
    image[idx] = im;
}
void launch_integrate(double4 *pos, const double4 *vel, double3 *accel, int3 *image, const double4 *force,
                      const unsigned *group, int N, DBox box, double dt, double shear_rate, hipStream_t s) {
    hipLaunchKernelGGL(k_integrate, dim3(nblocks(N, TPB)), dim3(TPB), 0, s, pos, vel, accel, image, force, group, N,
                       box, dt, shear_rate);
}

}  // namespace pse
