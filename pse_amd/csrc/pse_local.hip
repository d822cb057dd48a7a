// Kernels of the owned-particle team step (pse_local.h).  gfx950, wave64, fp64.
#include "pse_local.h"
#include "pse_farbin.h"

namespace pse {

constexpr int TPB = 256;
constexpr unsigned KEY_FOREIGN = 0xFFFFFFFFu;
static inline int nblocks(long n, int tpb) { return (int)((n + tpb - 1) / tpb); }

// cell layer cx relative to the first layer of the rank's slab, nearest periodic image around the slab's centre:
// 0 .. per - 1 = own layers, negative = towards the left neighbour, >= per = towards the right one
__device__ __forceinline__ int rel_layer(int cx, const LocalGeom &g) {
    int s2 = 2 * (cx - g.rank * g.per) - g.per;            // twice the distance from the slab's centre, minus a half layer
    const int n2 = 2 * g.nx;
    s2 %= n2;
    if (s2 < -g.nx) s2 += n2; else if (s2 >= g.nx) s2 -= n2;
    return (s2 + g.per) / 2;                               // (even: exact)
}
__device__ __forceinline__ unsigned cell_of(const DBox &box, const DCells &nc, double x, double y, double z, int &cx) {
    double fx, fy, fz;
    frac_coords(box, x, y, z, fx, fy, fz);
    cx = cell_coord(fx, nc.nx);
    const int cy = cell_coord(fy, nc.ny), cz = cell_coord(fz, nc.nz), zb = cz / nc.bz;
    return (unsigned)cell_slot(nc, cx, cy, zb, cz - zb * nc.bz);
}
// slot of this thread's item in a list that grows by ONE atomic per workgroup (-1: the thread has none): a word takes ~88 atomics
// per microsecond (MI355X_MICROARCH.md), and the particles of the boundary layers sit in a third of the workgroups
__device__ __forceinline__ int block_append(bool want, int *counter, int *sh /* TPB / 64 + 1 ints */) {
    const unsigned long long m = __ballot(want);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sh[wv] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < TPB / 64; ++w) { const int c = sh[w]; sh[w] = tot; tot += c; }
        sh[TPB / 64] = tot ? atomicAdd(counter, tot) : 0;
    }
    __syncthreads();
    const int slot = sh[TPB / 64] + sh[wv] + __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    return want ? slot : -1;
}
__device__ __forceinline__ void write_record(double *records, int slot, const double4 &p, const double4 &f, double mass, const int3 &im, unsigned tag) {
    double2 *r = reinterpret_cast<double2 *>(records + (size_t)slot * LOCAL_REC);
    r[0] = make_double2(p.x, p.y); r[1] = make_double2(p.z, p.w);
    r[2] = make_double2(f.x, f.y); r[3] = make_double2(f.z, mass);
    const int4 w = make_int4(im.x, im.y, im.z, (int)tag);      // bit patterns: the transports move bytes
    r[4] = *reinterpret_cast<const double2 *>(&w);
}

// layer_cnt[slot][cx] += (kept particles of this workgroup in layer cx): a wave counts its lanes per distinct layer with ballots (the
// rows come in the cell order of the last step: one or two layers per wave), the workgroup adds them up in LDS, and ONE device atomic
// per workgroup and layer goes to one of LOCAL_LAYER_SLOTS copies of the counters, 2 KB apart.  (One atomic per wave on the ten
// adjacent words of one copy took the classification from 16 to 36 us: ~3 400 atomics on one cache line at ~88 per microsecond.)
__device__ __forceinline__ void count_layers(int cx, bool keep, int *__restrict__ layer_cnt, int *sh_hist /* LOCAL_MAX_LAYERS ints of LDS */) {
    for (int k = threadIdx.x; k < LOCAL_MAX_LAYERS; k += TPB) sh_hist[k] = 0;
    __syncthreads();
    unsigned long long todo = __ballot(keep);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int cc = __shfl(cx, leader);
        const unsigned long long m = __ballot(keep && cx == cc);
        if (lane == leader) atomicAdd(&sh_hist[cc], __popcll(m));
        todo &= ~m;
    }
    __syncthreads();
    int *mine = layer_cnt + (blockIdx.x % LOCAL_LAYER_SLOTS) * LOCAL_MAX_LAYERS;
    for (int k = threadIdx.x; k < LOCAL_MAX_LAYERS; k += TPB) { const int v = sh_hist[k]; if (v) atomicAdd(&mine[k], v); }
}
__device__ __forceinline__ int layer_count(const int *__restrict__ layer_cnt, int layer) {
    int t = 0;
#pragma unroll
    for (int q = 0; q < LOCAL_LAYER_SLOTS; ++q) t += layer_cnt[q * LOCAL_MAX_LAYERS + layer];
    return t;
}

__global__ void __launch_bounds__(TPB)
k_local_classify(LocalCaller c, LocalGeom g, DBox box, DCells nc, LocalPool pool, double *__restrict__ send_l, double *__restrict__ send_r,
                 int *__restrict__ counters, int *__restrict__ err, int *__restrict__ layer_cnt) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    const unsigned n_raw = *c.n_local;
    const int n = (int)min(n_raw, (unsigned)g.c_own);
    if (i == 0 && n_raw > (unsigned)g.c_own) atomicOr(err, LOCAL_ERR_COUNT);
    bool to_l = false, to_r = false;
    double4 p = make_double4(0, 0, 0, 0), f = p;
    double mass = 0.0;
    int3 im = make_int3(0, 0, 0);
    unsigned tag = 0;
    int cx = 0;
    bool kept = false;
    if (i < n) {
        p = c.pos[i];
        const unsigned key = cell_of(box, nc, p.x, p.y, p.z, cx);
        const int s = rel_layer(cx, g);
        // what the neighbours need: their ghosts (my first / last `depth` layers) and the particles that have left the slab on their side
        to_l = s < g.depth;
        to_r = s >= g.per - g.depth;
        if (g.G == 2 && s < 0) to_r = false;              // two ranks: both sides are the same neighbour -- a migrant travels once
        if (g.G == 2 && s >= g.per) to_l = false;
        // beyond the layers the neighbour shares with ITS other neighbour: more than a slab (less the ghost depth) in one step
        if (s < -(g.per - g.depth) || s >= 2 * g.per - g.depth) atomicOr(err, LOCAL_ERR_FAR);
        const bool keep = s >= -g.depth && s < g.per + g.depth;
        kept = keep;
        if (keep) {
            pool.keys[i] = key;
            pool.rank[i] = (unsigned)atomicAdd(&pool.cnt[key], 1);
            pool.ptag[i] = c.tag[i];
        } else {
            pool.keys[i] = KEY_FOREIGN;
        }
        if (to_l || to_r) { f = c.force[i]; mass = c.vel[i].w; im = c.image[i]; tag = c.tag[i]; }
    } else if (i < g.c_own) {
        pool.keys[i] = KEY_FOREIGN;
    }
    __shared__ int sh[TPB / 64 + 1];
    __shared__ int sh_hist[LOCAL_MAX_LAYERS];
    count_layers(cx, kept, layer_cnt, sh_hist);
    const int sl = block_append(to_l, &counters[0], sh), sr = block_append(to_r, &counters[1], sh);
    // (the numbers of records travel as they are: the two counters are one more transfer of the exchange)
    if (to_l) { if (sl < g.c_x) write_record(send_l + LOCAL_HDR, sl, p, f, mass, im, tag); else atomicOr(err, LOCAL_ERR_MSG); }
    if (to_r) { if (sr < g.c_x) write_record(send_r + LOCAL_HDR, sr, p, f, mass, im, tag); else atomicOr(err, LOCAL_ERR_MSG); }
}
void launch_local_classify(const LocalCaller &c, const LocalGeom &g, DBox box, DCells nc, LocalPool pool, double *send_l, double *send_r,
                           int *counters, int *err, int *layer_cnt, hipStream_t s) {
    hipLaunchKernelGGL(k_local_classify, dim3(nblocks(g.c_own, TPB)), dim3(TPB), 0, s, c, g, box, nc, pool, send_l, send_r, counters, err, layer_cnt);
}

__device__ __forceinline__ bool bin_one(int q, int &cx, const double *__restrict__ recv_l, const double *__restrict__ recv_r, const LocalGeom &g, const DBox &box,
                                        const DCells &nc, const LocalPool &pool, int *__restrict__ err) {
    if (q >= 2 * g.c_x) return false;
    const int side = q / g.c_x, k = q - side * g.c_x;
    const double *m = side == 0 ? recv_l : recv_r;
    const int pid = g.c_own + q;
    // the sender's two counters (records in ITS left, right message) arrived in front of the records: what came from the left
    // neighbour is its right message, what came from the right neighbour its left one
    const int cnt = reinterpret_cast<const int *>(m)[side == 0 ? 1 : 0];
    if (cnt < 0 || cnt > g.c_x) { if (k == 0) atomicOr(err, LOCAL_ERR_MSG); }   // (the sender flags its own overflow; it sent c_x records)
    if (k >= min(cnt, g.c_x)) { pool.keys[pid] = KEY_FOREIGN; return false; }
    const double2 *r = reinterpret_cast<const double2 *>(m + LOCAL_HDR + (size_t)k * LOCAL_REC);
    const double2 a = r[0], b = r[1];
    const unsigned key = cell_of(box, nc, a.x, a.y, b.x, cx);
    const int s = rel_layer(cx, g);
    if (s >= -g.depth && s < g.per + g.depth) {
        pool.keys[pid] = key;
        pool.rank[pid] = (unsigned)atomicAdd(&pool.cnt[key], 1);
        const double2 w2 = r[4];
        pool.ptag[pid] = (unsigned)reinterpret_cast<const int4 *>(&w2)->w;
        return true;
    }
    pool.keys[pid] = KEY_FOREIGN;
    return false;
}
__global__ void __launch_bounds__(TPB)
k_local_bin_incoming(const double *__restrict__ recv_l, const double *__restrict__ recv_r, LocalGeom g, DBox box, DCells nc, LocalPool pool,
                     int *__restrict__ err, int *__restrict__ layer_cnt) {
    const int q = blockIdx.x * TPB + threadIdx.x;
    int cx = 0;
    const bool kept = bin_one(q, cx, recv_l, recv_r, g, box, nc, pool, err);
    __shared__ int sh_hist[LOCAL_MAX_LAYERS];
    count_layers(cx, kept, layer_cnt, sh_hist);
}
void launch_local_bin_incoming(const double *recv_l, const double *recv_r, const LocalGeom &g, DBox box, DCells nc, LocalPool pool, int *err,
                               int *layer_cnt, hipStream_t s) {
    hipLaunchKernelGGL(k_local_bin_incoming, dim3(nblocks(2 * g.c_x, TPB)), dim3(TPB), 0, s, recv_l, recv_r, g, box, nc, pool, err, layer_cnt);
}

// Row offsets of the cells a rank keeps, ONE launch without a grid-wide scan (was: rocPRIM's two-kernel scan over ALL cells of the box --
// 1.1 M at config 4 -- and the region arithmetic in the scatter): classify / bin_incoming have counted the kept particles per x LAYER
// (a handful of atomics per wave), so the first row of every layer is a sum of at most per + 2 depth numbers, and a workgroup per
// kept layer scans that layer's cells on its own.  Block 0 also derives the row ranges of the step (LocalRows) from the layer counts.
constexpr int OFF_TPB = 1024;
__global__ void __launch_bounds__(OFF_TPB)
k_local_offsets(const int *__restrict__ cell_cnt, const int *__restrict__ layer_cnt, int *__restrict__ cell_off, LocalGeom g, LocalRegions rg, int lc,
                LocalRows *__restrict__ rows, int *__restrict__ err) {
    // the kept layers in region order: own (per), left ghosts (depth), right ghosts (depth); layer index of block b, its region q
    const int b = blockIdx.x;
    const int q = b < g.per ? 0 : (b < g.per + g.depth ? 1 : 2);
    const int in_q = b - (q == 0 ? 0 : (q == 1 ? g.per : g.per + g.depth));
    const int l0[3] = {rg.c0[0] / lc, rg.c0[1] / lc, rg.c0[2] / lc};        // first layer of every region (regions are whole layers)
    const int nl[3] = {g.per, g.depth, g.depth};
    int nq[3];
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        int t = 0;
        for (int k = 0; k < nl[r]; ++k) t += layer_count(layer_cnt, l0[r] + k);
        nq[r] = t;
        ok = ok && t <= rg.cap[r];
    }
    // The boundary layers of the own rows are what the NEIGHBOURS hold as ghosts, and what the Lanczos blocks send them in messages of
    // c_g rows (the first `depth` layers as they lie, the last ones parked in stage_w1 / stage_w2, c_g rows each): more than c_g rows
    // in either is the neighbour's ghost overflow seen from this side -- the step must not run here either (the mat-vecs would park
    // rows beyond their staging buffers).
    int first_end = 0, last_begin = 0;
    for (int k = 0; k < g.depth; ++k) first_end += layer_count(layer_cnt, l0[0] + k);
    for (int k = 0; k < g.per - g.depth; ++k) last_begin += layer_count(layer_cnt, l0[0] + k);
    const bool edge_ok = first_end <= rg.cap[1] && nq[0] - last_begin <= rg.cap[2];
    ok = ok && edge_ok;
    if (b == 0 && threadIdx.x == 0) {
        LocalRows r{};
        if (!ok) {
            atomicOr(err, (nq[0] > rg.cap[0] ? LOCAL_ERR_OWN : 0) | ((nq[1] > rg.cap[1] || nq[2] > rg.cap[2] || !edge_ok) ? LOCAL_ERR_GHOST : 0));
            r.own = RowMap{1, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}}; r.own1 = r.own;
        } else {
            r.ok = 1;
            r.n_own = nq[0]; r.n_gl = nq[1]; r.n_gr = nq[2];
            r.first_end = first_end;
            r.last_begin = last_begin;
            int gl_adj = rg.base[1];                              // first row of the left ghosts' layer next to the slab (their last layer)
            for (int k = 0; k < g.depth - 1; ++k) gl_adj += layer_count(layer_cnt, l0[1] + k);
            const int gr_adj_end = rg.base[2] + layer_count(layer_cnt, l0[2]);  // end of the right ghosts' layer next to the slab (their first)
            r.own = RowMap{1, {0, 0, 0}, {nq[0], 0, 0}, {0, 0, 0}};
            const int b1 = (nq[0] + 255) & ~255, len1 = rg.base[1] + nq[1] - gl_adj;
            r.own1 = RowMap{3, {0, gl_adj, rg.base[2]}, {nq[0], rg.base[1] + nq[1], gr_adj_end}, {0, b1, b1 + ((len1 + 255) & ~255)}};
            r.all = RowRanges{3, {0, rg.base[1], rg.base[2]}, {nq[0], rg.base[1] + nq[1], rg.base[2] + nq[2]}};
        }
        *rows = r;
    }
    if (!ok) return;
    int base = rg.base[q];
    for (int k = 0; k < in_q; ++k) base += layer_count(layer_cnt, l0[q] + k);
    // exclusive scan of this layer's lc cell counts (the last storage cell of a layer is empty: its offset ends the layer's rows)
    const int c_first = (l0[q] + in_q) * lc;
    const int per_t = (lc + OFF_TPB - 1) / OFF_TPB, t0 = threadIdx.x * per_t, t1 = min(lc, t0 + per_t);
    int mine = 0;
    for (int k = t0; k < t1; ++k) mine += cell_cnt[c_first + k];
    __shared__ int wsum[OFF_TPB / 64];
    int incl = mine;                                              // inclusive scan across the lanes of a wave ...
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int woff = 0;                                                 // ... and across the waves
    for (int w = 0; w < wv; ++w) woff += wsum[w];
    int run = base + woff + incl - mine;
    for (int k = t0; k < t1; ++k) { cell_off[c_first + k] = run; run += cell_cnt[c_first + k]; }
}
void launch_local_offsets(const int *cell_cnt, const int *layer_cnt, int *cell_off, const LocalGeom &g, const LocalRegions &rg, int layer_cells,
                          LocalRows *rows, int *err, hipStream_t s) {
    hipLaunchKernelGGL(k_local_offsets, dim3(g.per + 2 * g.depth), dim3(OFF_TPB), 0, s, cell_cnt, layer_cnt, cell_off, g, rg, layer_cells, rows, err);
}

// the slot of every kept particle of the pool: the first row of its cell + its arrival rank there
__global__ void __launch_bounds__(TPB)
k_local_scatter(const int *__restrict__ cell_off, LocalGeom g, LocalPool pool, unsigned *__restrict__ slots, const LocalRows *__restrict__ rows) {
    if (!rows->ok) return;
    const int v = blockIdx.x * TPB + threadIdx.x;
    if (v >= g.c_own + 2 * g.c_x) return;
    const unsigned key = pool.keys[v];
    if (key == KEY_FOREIGN) return;
    slots[cell_off[key] + (int)pool.rank[v]] = (unsigned)v;
}
void launch_local_scatter(const int *cell_off, const LocalGeom &g, LocalPool pool, unsigned *slots, const LocalRows *rows, hipStream_t s) {
    hipLaunchKernelGGL(k_local_scatter, dim3(nblocks(g.c_own + 2 * g.c_x, TPB)), dim3(TPB), 0, s, cell_off, g, pool, slots, rows);
}

// One thread per row of the row space.  A live row s holds pool member v = slots[s] of cell c; its final row is the cell's first
// row + the number of members with a smaller tag (the order BOTH ranks that hold the cell arrive at).  Then what k_permute does
// for a replicated rank: wrapped position (+ float copy, packed records), force, tag, noise, far-field bin rank -- and the
// particle state the end of the step needs (position as given, mass, image).
__global__ void __launch_bounds__(TPB)
k_local_permute(LocalCaller c, const double *__restrict__ recv_l, const double *__restrict__ recv_r, LocalGeom g, DBox box,
                const int *__restrict__ cell_off, LocalPool pool, const unsigned *__restrict__ slots, const LocalRows *__restrict__ rows,
                LocalSorted o, FarBinArgs far, uint32_t seed, uint32_t timestep, const uint32_t *__restrict__ ts_off) {
    const int s = blockIdx.x * TPB + threadIdx.x;
    const int rows_cap = g.c_own + 2 * g.c_g;
    if (s - (int)(threadIdx.x & 63) >= rows_cap) return;
    const LocalRows R = *rows;
    const bool live = (s < R.n_own) || (s >= g.c_own && s < g.c_own + R.n_gl) || (s >= g.c_own + g.c_g && s < g.c_own + g.c_g + R.n_gr);
    if (ts_off) timestep += *ts_off;
    bool bin_need = false;
    int bin = -1, row = s;
    if (live) {
        const unsigned v = slots[s];
        const int cl = (int)pool.keys[v], a = cell_off[cl], n = cell_off[cl + 1] - a;
        const unsigned tg = pool.ptag[v];
        int smaller = 0;
        for (int t = 0; t < n; ++t) smaller += pool.ptag[slots[a + t]] < tg ? 1 : 0;
        row = a + smaller;
        double4 p, f;
        double mass;
        int3 im;
        if ((int)v < g.c_own) {
            p = c.pos[v]; f = c.force[v]; mass = c.vel[v].w; im = c.image[v];
        } else {
            const int q = (int)v - g.c_own, side = q / g.c_x, k = q - side * g.c_x;
            const double2 *r = reinterpret_cast<const double2 *>((side == 0 ? recv_l : recv_r) + LOCAL_HDR + (size_t)k * LOCAL_REC);
            const double2 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4];
            p = make_double4(r0.x, r0.y, r1.x, r1.y); f = make_double4(r2.x, r2.y, r3.x, 0.0); mass = r3.y;
            const int4 w = *reinterpret_cast<const int4 *>(&r4);
            im = make_int3(w.x, w.y, w.z);
        }
        // wrap into the primary cell (k_permute, pse_kernels.hip)
        double fx, fy, fz;
        frac_coords(box, p.x, p.y, p.z, fx, fy, fz);
        const double y = (fy - 0.5) * box.Ly;
        const double4 q = make_double4((fx - 0.5) * box.Lx + box.xy * y, y, (fz - 0.5) * box.Lz, 0.0);
        o.pos_s[row] = q;
        o.posf_s[row] = make_float4((float)q.x, (float)q.y, (float)q.z, 0.0f);
        o.pv[3 * (size_t)row] = make_double2(q.x, q.y);
        o.pv[3 * (size_t)row + 1] = make_double2(q.z, f.x);
        o.pv[3 * (size_t)row + 2] = make_double2(f.y, f.z);
        o.f_s[row] = make_double4(f.x, f.y, f.z, 0.0);
        o.tag_s[row] = tg;
        if (row < g.c_own) { o.porig_s[row] = p; o.mass_s[row] = mass; o.image_s[row] = im; }
        if (o.psi_s) {
            uint32_t r[4];
            philox4x32(tg, 0u, timestep, DOMAIN_PARTICLE, seed, PHILOX_KEY1, r);
            const double cc = 1.7320508075688772;
            o.psi_s[row] = make_double4(uniform_pm(r[0], cc), uniform_pm(r[1], cc), uniform_pm(r[2], cc), 0.0);
        }
        if (far.on) {
            double gx, gy, gz;
            frac_coords(box, q.x, q.y, q.z, gx, gy, gz);
            int4 og;
            double4 d;
            far_support(gx, gy, gz, far.G, og, d);
            bin_need = wrapi(og.w - (far.G.x0 - far.G.P), far.G.Nx) < far.G.nxl + 2 * far.G.P;   // within a support of the slab's planes
            bin = bin_index(og.x, og.y, og.z, far.fb);
        }
    }
    if (far.on) {
        const int rk = far_bin_rank(bin_need, bin, far.fb.cnt);
        if (live) far.fb.rank_s[row] = rk;
        else if (s < rows_cap) far.fb.rank_s[s] = -1;       // a row of the capacity that holds no particle this step
    }
}
void launch_local_permute(const LocalCaller &c, const double *recv_l, const double *recv_r, const LocalGeom &g, DBox box, const int *cell_off,
                          LocalPool pool, const unsigned *slots, const LocalRows *rows, LocalSorted out, const FarBinArgs *far, uint32_t seed,
                          uint32_t timestep, const uint32_t *ts_off, hipStream_t s) {
    hipLaunchKernelGGL(k_local_permute, dim3(nblocks(g.c_own + 2 * g.c_g, TPB)), dim3(TPB), 0, s, c, recv_l, recv_r, g, box, cell_off, pool,
                       slots, rows, out, far ? *far : FarBinArgs{}, seed, timestep, ts_off);
}

__global__ void __launch_bounds__(TPB)
k_local_finish(LocalFinish a, LocalCaller c, DBox box) {
    const int n = a.rows->n_own;
    for (int k = blockIdx.x * TPB + threadIdx.x; k < a.n_zero; k += gridDim.x * TPB) a.zero[k] = 0;   // (everything that counted into them has been consumed: both lanes have joined)
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.rows->ok) *c.n_local = (unsigned)n;   // (a step that overflowed leaves the caller's state alone)
    int m = 0;
    const double *t = nullptr;
    double sc = 0.0;
    if (a.st) { m = a.st->m_final; t = a.st->coef; sc = a.scale * a.scal[LZ_NORM]; }
    for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
        double x = 0, y = 0, z = 0;
        for (int q = 0; q < m; ++q) {                        // K13 + the final rescale (PSEv1/Helper.cu:251-279, PSEv1/Brownian.cu:739)
            const double4 v = q == 0 ? a.psi_s[i] : a.V[(size_t)q * a.stride + i];
            const double tq = t[q];
            x += tq * v.x; y += tq * v.y; z += tq * v.z;
        }
        x *= sc; y *= sc; z *= sc;
        double ux = 0.0, uy = 0.0, uz = 0.0;                 // far field + near field + Brownian, in that order (k_basis_combine's sink)
        if (a.uw_s) { const double4 v = a.uw_s[i]; ux += v.x; uy += v.y; uz += v.z; }
        if (a.ur_s) { const double4 v = a.ur_s[i]; ux += v.x; uy += v.y; uz += v.z; }
        ux += x; uy += y; uz += z;
        double4 p = a.porig_s[i];
        const double mass = a.mass_s[i];
        int3 im = a.image_s[i];
        const double4 F = a.f_s[i];
        if (a.integrate) {                                   // K15 gpu_stokes_step_one_kernel (PSEv1/Stokes.cu:137-192), as k_integrate
            p.x += (ux + a.shear_rate * p.y) * a.dt;
            p.y += uy * a.dt;
            p.z += uz * a.dt;
            double w = floor(p.z * box.iLz + 0.5);
            p.z -= w * box.Lz; im.z += (int)w;
            w = floor(p.y * box.iLy + 0.5);
            p.y -= w * box.Ly; p.x -= w * box.xy * box.Ly; im.y += (int)w;
            w = floor((p.x - box.xy * p.y) * box.iLx + 0.5);
            p.x -= w * box.Lx; im.x += (int)w;
        }
        const double im_ = 1.0 / mass;
        c.pos[i] = p;
        c.vel[i] = make_double4(ux, uy, uz, mass);
        c.accel[i] = make_double3(F.x * im_, F.y * im_, F.z * im_);
        c.image[i] = im;
        c.tag[i] = a.tag_s[i];
    }
}
void launch_local_finish(const LocalFinish &a, const LocalCaller &c, DBox box, int rows_cap, hipStream_t s) {
    hipLaunchKernelGGL(k_local_finish, dim3(std::min(2048, std::max(1, nblocks(rows_cap, TPB)))), dim3(TPB), 0, s, a, c, box);
}

// ---- redistribution after a Lees-Edwards flip (pse_team_redistribute_local) ------------------------------------------------------------
// A flip of the tilt (xy + 0.5 -> - 0.5, PSEv1/VariantShearFunction.cc:34-43) re-maps the fractional x of every particle by its
// fractional y: any particle may belong to any rank afterwards, which the one-neighbour migration of a step cannot follow.  Three
// kernels around two exchanges of the team's transfer list: (1) the new owner of every particle and how many go where; [the ranks
// exchange their count rows; the host sizes the messages]; (2) the records (the wire format of the step's first exchange), packed
// by destination; [the records travel: exact sizes]; (3) what arrived becomes the caller's arrays.
__device__ __forceinline__ int wave_append(int dest, bool live, int *counters, int stride_words = 1) {
    // slot of this lane among the particles of its destination: one atomic per distinct destination of the wave
    int slot = -1;
    unsigned long long todo = __ballot(live);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int dd = __shfl(dest, leader);
        const unsigned long long m = __ballot(live && dest == dd);
        int base = 0;
        if (lane == leader) base = atomicAdd(&counters[dd * stride_words], __popcll(m));
        base = __shfl(base, leader);
        if (live && dest == dd) slot = base + __popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
    }
    return slot;
}
__global__ void __launch_bounds__(TPB)
k_redist_count(LocalCaller c, LocalGeom g, DBox box, DCells nc, unsigned *__restrict__ dest, int *__restrict__ counts, int *__restrict__ err) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    const unsigned n_raw = *c.n_local;
    const int n = (int)min(n_raw, (unsigned)g.c_own);
    if (i == 0 && n_raw > (unsigned)g.c_own) atomicOr(err, LOCAL_ERR_COUNT);
    int d = 0;
    const bool live = i < n;
    if (live) {
        const double4 p = c.pos[i];
        int cx;
        (void)cell_of(box, nc, p.x, p.y, p.z, cx);
        d = min(cx / g.per, g.G - 1);
        dest[i] = (unsigned)d;
    }
    (void)wave_append(d, live, counts);
}
__global__ void __launch_bounds__(TPB)
k_redist_pack(LocalCaller c, LocalGeom g, const unsigned *__restrict__ dest, const int *__restrict__ send_off, int *__restrict__ fill,
              double *__restrict__ records) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    const int n = (int)min(*c.n_local, (unsigned)g.c_own);
    const bool live = i < n;
    const int d = live ? (int)dest[i] : 0;
    const int k = wave_append(d, live, fill);
    if (live) write_record(records, send_off[d] + k, c.pos[i], c.force[i], c.vel[i].w, c.image[i], c.tag[i]);
}
__global__ void __launch_bounds__(TPB)
k_redist_unpack(const double *__restrict__ records, int total, LocalCaller c) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i == 0) *c.n_local = (unsigned)total;
    if (i >= total) return;
    const double2 *r = reinterpret_cast<const double2 *>(records + (size_t)i * LOCAL_REC);
    const double2 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4];
    const int4 w = *reinterpret_cast<const int4 *>(&r4);
    c.pos[i] = make_double4(r0.x, r0.y, r1.x, r1.y);
    const_cast<double4 *>(c.force)[i] = make_double4(r2.x, r2.y, r3.x, 0.0);      // (the caller's force provider would recompute it: the rows have a new order)
    c.vel[i] = make_double4(0.0, 0.0, 0.0, r3.y);
    c.accel[i] = make_double3(0.0, 0.0, 0.0);
    c.image[i] = make_int3(w.x, w.y, w.z);
    c.tag[i] = (unsigned)w.w;
}
void launch_redist_count(const LocalCaller &c, const LocalGeom &g, DBox box, DCells nc, unsigned *dest, int *counts, int *err, hipStream_t s) {
    hipLaunchKernelGGL(k_redist_count, dim3(nblocks(g.c_own, TPB)), dim3(TPB), 0, s, c, g, box, nc, dest, counts, err);
}
void launch_redist_pack(const LocalCaller &c, const LocalGeom &g, const unsigned *dest, const int *send_off, int *fill, double *records, hipStream_t s) {
    hipLaunchKernelGGL(k_redist_pack, dim3(nblocks(g.c_own, TPB)), dim3(TPB), 0, s, c, g, dest, send_off, fill, records);
}
void launch_redist_unpack(const double *records, int total, const LocalCaller &c, hipStream_t s) {
    hipLaunchKernelGGL(k_redist_unpack, dim3(std::max(1, nblocks(total, TPB))), dim3(TPB), 0, s, records, total, c);
}

}  // namespace pse
