// Near field of the PSE engine on cell blocks with LDS-staged particle tiles (gfx950, fp64).
//
// K9 gpu_stokes_Mreal_kernel (PSEv1/Mobility.cu:594-687): u_i = self F_i + sum_j [f (I - rr) + g rr] F_j over minimum-image
// neighbours with r < rcut, applied once for M.F and once per Lanczos iteration (PSEv1/Brownian.cu:473-521).
//
// A workgroup owns a block of bx x by x bz cells of the engine's cell list.  It copies the (position, vector) records of
// the block's cells AND of the one-cell halo around it into LDS -- contiguous runs of the cell-sorted arrays, read with
// coalesced loads, periodic images resolved while copying -- so that everything a pair needs afterwards is an LDS
// access instead of three scattered 16-byte global gathers per pair (the round-1 pair-list mat-vec kept the texture
// addresser 99.5 % busy with those).  Two phases share the tile:
//   build   every row scans the staged particles of its 27 cells (merged into 9 z runs) against the cutoff in double
//           precision, evaluates f(r), g(r) for the hits from the coefficient table (in LDS, intervals padded to 21 doubles
//           so that lanes in different intervals read different banks) and writes the per-step pair list: a 2-byte index
//           INTO THE TILE plus (f, (g - f)/r^2), slot-major so that the lanes of a wave read and write contiguous bytes.
//   apply   every row walks its list: entry (prefetched four slots ahead) -> tile record -> separation -> accumulate.
// Positions do not change inside a step, so the Lanczos mat-vecs run `apply` alone: a coalesced 18-byte stream per pair and
// LDS reads, no gathers from global memory at all.  With TPR > 1 (few, long rows: large cutoffs) TPR adjacent lanes share a
// row, each with its own sub-list.
//
// Blocks whose tile does not fit (dense clusters) and rows whose list overflows fall back to walking the cells in global
// memory; results are the same up to summation order.
#include "pse_kernels.h"

namespace pse {

constexpr int NB_NT = 256;              // threads per workgroup = most rows x TPR of a block
constexpr int CSTRIDE = 2 * RS_NCOEF + 1;

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
    return v;
}

// one row from the cells in global memory (tile overflow / list overflow): what round 1's cell kernel did
template <bool TWO>
__device__ __noinline__ void row_walk_cells(int i, const double4 *__restrict__ pos_s, const double4 *__restrict__ vec_s,
                                            const double4 *__restrict__ vec2_s, const int *__restrict__ cell_off, const DBox &box,
                                            const DCells &nc, double rcut2, const double *__restrict__ coef, double (&u)[3], double (&w)[3]) {
    const double4 pi = pos_s[i];
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
                eval_fg(r2, coef, f, h);
                const double4 Fj = vec_s[j];
                const double rdF = (dx * Fj.x + dy * Fj.y + dz * Fj.z) * h;
                u[0] += f * Fj.x + rdF * dx; u[1] += f * Fj.y + rdF * dy; u[2] += f * Fj.z + rdF * dz;
                if (TWO) {
                    const double4 Gj = vec2_s[j];
                    const double rdG = (dx * Gj.x + dy * Gj.y + dz * Gj.z) * h;
                    w[0] += f * Gj.x + rdG * dx; w[1] += f * Gj.y + rdG * dy; w[2] += f * Gj.z + rdG * dz;
                }
            }
        }
    });
}

// BUILD: scan the tile, evaluate and write the pair list, apply it on the way.  TWO: a second vector rides along (out2 = M_real
// vec2).  FUSE: the Lanczos sums of LzFuse (x = vec, y = M x: partial sums of x.x, x.y, x.v_{j-1} per block).  TPR: lanes per row.
template <bool BUILD, bool TWO, bool FUSE, int TPR>
__global__ void __launch_bounds__(NB_NT)
k_mreal_blocks(const double4 *__restrict__ pos_s, const double4 *__restrict__ vec_s, double4 *__restrict__ out_s,
               const double4 *__restrict__ vec2_s, double4 *__restrict__ out2_s, const int *__restrict__ cell_off, DBox box, DCells nc,
               double rcut2, double self, const double *__restrict__ coef_g, int nint, NbList nb, LzFuse lz, int nbby, int nbbz) {
    constexpr int NT = NB_NT, RPB = NT / TPR, REC = TWO ? 9 : 6;     // rows per block, doubles per staged record
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const NbBlocks &B = nb.blk;
    double *S = smem;                                                // [cap_st][REC]: x, y, z (image applied), vector(s)
    double *scoef = S + (size_t)B.cap_st * REC;                      // BUILD: [nint][CSTRIDE]
    int *s_src = reinterpret_cast<int *>(scoef + (BUILD ? (size_t)nint * CSTRIDE : 0));   // [cap_st] row | image code << 27
    int *s_coff = s_src + B.cap_st;                                  // [NH + 1] first tile index of each halo cell
    int *s_rpre = s_coff + 260;                                      // [NI + 1] first block row of each inner cell
    int *s_rhid = s_rpre + 68;                                       // [NI] its halo-cell index
    int *s_ws = s_rhid + 68;                                         // [8] scratch
    __shared__ double s_red[12];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = xcd_block(blockIdx.x, gridDim.x);
    const int ibz = b % nbbz; b /= nbbz;
    const int iby = b % nbby; b /= nbby;
    const int ibx = b;
    const int c0[3] = {B.cx0 + ibx * B.bx, iby * B.by, ibz * B.bz};
    const int ex = min(B.bx, B.cx0 + B.ncx - c0[0]), ey = min(B.by, nc.ny - c0[1]), ez = min(B.bz, nc.nz - c0[2]);
    const int hy = ey + 2, hz = ez + 2, NH = (ex + 2) * hy * hz, NI = ex * ey * ez;

    // ---- the halo cells: rows in the sorted arrays, image, first index in the tile (exclusive scan of the counts)
    int my_start = 0, my_code = 0, my_cnt = 0, my_off = 0;
    {
        if (tid < NH) {
            const int qz = tid % hz, r = tid / hz, qy = r % hy, qx = r / hy;
            int ax = c0[0] + qx - 1, ay = c0[1] + qy - 1, az = c0[2] + qz - 1, wx = 0, wy = 0, wz = 0;
            if (ax < 0) { ax += nc.nx; wx = -1; } else if (ax >= nc.nx) { ax -= nc.nx; wx = 1; }
            if (ay < 0) { ay += nc.ny; wy = -1; } else if (ay >= nc.ny) { ay -= nc.ny; wy = 1; }
            if (az < 0) { az += nc.nz; wz = -1; } else if (az >= nc.nz) { az -= nc.nz; wz = 1; }
            const int c = (ax * nc.ny + ay) * nc.nz + az;
            my_start = cell_off[c];
            my_cnt = cell_off[c + 1] - my_start;
            my_code = (wx + 1) * 9 + (wy + 1) * 3 + (wz + 1);
        }
        const int inc = wave_incl_scan(my_cnt, lane);
        if (lane == 63) s_ws[wave] = inc;
        __syncthreads();
        int pre = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) pre += w < wave ? s_ws[w] : 0;
        my_off = pre + inc - my_cnt;
        if (tid < NH) s_coff[tid] = my_off;
        if (tid == NH - 1) s_coff[NH] = pre + inc;
    }
    if (BUILD)
        for (int q = tid; q < nint * 2 * RS_NCOEF; q += NT) scoef[(q / (2 * RS_NCOEF)) * CSTRIDE + q % (2 * RS_NCOEF)] = coef_g[q];
    __syncthreads();
    const int nst = s_coff[NH];
    // ---- the block's rows: inner cells in (x, y, z) order
    if (wave == 0) {
        int cnt = 0, hid = 0;
        if (lane < NI) {
            const int iz = lane % ez, r = lane / ez, iy = r % ey, ix = r / ey;
            hid = ((ix + 1) * hy + iy + 1) * hz + iz + 1;
            cnt = s_coff[hid + 1] - s_coff[hid];
            s_rhid[lane] = hid;
        }
        const int inc = wave_incl_scan(cnt, lane);
        if (lane < NI) s_rpre[lane] = inc - cnt;
        if (lane == NI - 1) s_rpre[NI] = inc;
    }
    const bool tile_ok = nst <= B.cap_st;
    // where every tile entry comes from: one lane per halo cell writes its particles' (row, image) words
    if (tile_ok && tid < NH)
        for (int k = 0; k < my_cnt; ++k) s_src[my_off + k] = (my_start + k) | (my_code << 27);
    __syncthreads();
    const int nrows = s_rpre[NI];
    const bool fits = tile_ok && nrows <= RPB;
    const int rr = tid / TPR, q = tid % TPR;
    const bool has_row = fits && rr < nrows;
    int own = 0, i = 0, hid = 0;
    if (has_row) {
        int lo = 0, hi = NI;                                          // largest inner cell with s_rpre <= rr
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_rpre[mid] <= rr) lo = mid; else hi = mid; }
        hid = s_rhid[lo];
        const int k = rr - s_rpre[lo];
        own = s_coff[hid] + k;
        i = (s_src[own] & 0x7FFFFFF);
    }
    double u[3] = {0.0, 0.0, 0.0}, w2[3] = {0.0, 0.0, 0.0};
    double4 xi = make_double4(0.0, 0.0, 0.0, 0.0), x2 = xi;
    if (fits) {
        // ---- stage the tile: all loads of a lane in flight together
        constexpr int SIT = 5;                                        // cap_st <= SIT * NT
        double4 sp[SIT], sv[SIT];
        unsigned ssrc[SIT];                                           // row | image code << 27 (codes are below 27); ~0: none
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int e = tid + it * NT;
            ssrc[it] = e < nst ? (unsigned)s_src[e] : ~0u;
        }
#pragma unroll
        for (int it = 0; it < SIT; ++it)
            if (ssrc[it] != ~0u) { const int j = ssrc[it] & 0x7FFFFFF; sp[it] = pos_s[j]; sv[it] = vec_s[j]; }
#pragma unroll
        for (int it = 0; it < SIT; ++it)
            if (ssrc[it] != ~0u) {
                double sx, sy, sz;
                image_shift(ssrc[it] >> 27, box, sx, sy, sz);
                double *rec = S + (size_t)(tid + it * NT) * REC;
                rec[0] = sp[it].x + sx; rec[1] = sp[it].y + sy; rec[2] = sp[it].z + sz;
                rec[3] = sv[it].x; rec[4] = sv[it].y; rec[5] = sv[it].z;
                if (TWO) { const double4 g = vec2_s[ssrc[it] & 0x7FFFFFF]; rec[6] = g.x; rec[7] = g.y; rec[8] = g.z; }
            }
        __syncthreads();
        const size_t lbase = ((size_t)blockIdx.x * B.cap) * NT + tid;                  // [slot][thread]
        unsigned short *myidx = B.list + lbase;
        double2 *myfh = B.fh + lbase;
        int cnt = 0;
        double pix = 0.0, piy = 0.0, piz = 0.0;
        if (has_row) {
            const double *me = S + (size_t)own * REC;
            pix = me[0]; piy = me[1]; piz = me[2];
            xi = make_double4(me[3], me[4], me[5], 0.0);
            if (TWO) x2 = make_double4(me[6], me[7], me[8], 0.0);
        }
        if (BUILD) {
            if (has_row) {
                const int hx_ = hid / (hy * hz), r = hid - hx_ * hy * hz, hy_ = r / hz, hz_ = r - hy_ * hz;
                for (int dx = -1; dx <= 1; ++dx)
                    for (int dy = -1; dy <= 1; ++dy) {
                        const int cb = ((hx_ + dx) * hy + hy_ + dy) * hz + hz_;       // the column's cell at this z
                        const int ja = s_coff[cb - 1], jb = s_coff[cb + 2];            // three z cells: one run of the tile
                        for (int j = ja + q; j < jb; j += TPR) {
                            const double *pj = S + (size_t)j * REC;
                            const double ddx = pix - pj[0], ddy = piy - pj[1], ddz = piz - pj[2];
                            const double r2 = ddx * ddx + ddy * ddy + ddz * ddz;
                            if (r2 < rcut2 && j != own && r2 > 0.0) {
                                if (cnt < B.cap) myidx[(size_t)cnt * NT] = (unsigned short)j;
                                ++cnt;
                            }
                        }
                    }
            }
            // the row's counts, 8 bits per lane of the row; -1: some sub-list overflowed (the row walks the cells)
            int packed = cnt << (8 * q);
            bool over = cnt > B.cap;
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) { packed |= __shfl_xor(packed, o, 64); over = over || __shfl_xor((int)over, o, 64); }
            if (over) { packed = -1; cnt = -1; }
            if (has_row && q == 0) nb.cnt[i] = packed;
            if (has_row && cnt >= 0) {
                // evaluate every pair once (dense: every lane has a real pair), keep (f, (g - f)/r^2) for the step, apply
                for (int s0 = 0; s0 < cnt; s0 += 2) {
                    const bool two = s0 + 1 < cnt;
                    const int j0 = myidx[(size_t)s0 * NT], j1 = myidx[(size_t)(two ? s0 + 1 : s0) * NT];
                    const double *p0 = S + (size_t)j0 * REC, *p1 = S + (size_t)j1 * REC;
                    const double d0x = pix - p0[0], d0y = piy - p0[1], d0z = piz - p0[2];
                    const double d1x = pix - p1[0], d1y = piy - p1[1], d1z = piz - p1[2];
                    double f0, h0, f1, h1;
                    eval_fg<CSTRIDE>(d0x * d0x + d0y * d0y + d0z * d0z, scoef, f0, h0);
                    eval_fg<CSTRIDE>(d1x * d1x + d1y * d1y + d1z * d1z, scoef, f1, h1);
                    myfh[(size_t)s0 * NT] = make_double2(f0, h0);
                    if (two) myfh[(size_t)(s0 + 1) * NT] = make_double2(f1, h1); else { f1 = 0.0; h1 = 0.0; }
                    const double r0 = (d0x * p0[3] + d0y * p0[4] + d0z * p0[5]) * h0, r1 = (d1x * p1[3] + d1y * p1[4] + d1z * p1[5]) * h1;
                    u[0] += f0 * p0[3] + r0 * d0x + f1 * p1[3] + r1 * d1x;
                    u[1] += f0 * p0[4] + r0 * d0y + f1 * p1[4] + r1 * d1y;
                    u[2] += f0 * p0[5] + r0 * d0z + f1 * p1[5] + r1 * d1z;
                    if (TWO) {
                        const double t0 = (d0x * p0[6] + d0y * p0[7] + d0z * p0[8]) * h0, t1 = (d1x * p1[6] + d1y * p1[7] + d1z * p1[8]) * h1;
                        w2[0] += f0 * p0[6] + t0 * d0x + f1 * p1[6] + t1 * d1x;
                        w2[1] += f0 * p0[7] + t0 * d0y + f1 * p1[7] + t1 * d1y;
                        w2[2] += f0 * p0[8] + t0 * d0z + f1 * p1[8] + t1 * d1z;
                    }
                }
            }
        } else {
            if (has_row) {
                const int packed = nb.cnt[i];
                cnt = packed < 0 ? -1 : (packed >> (8 * q)) & 0xFF;
            }
            // ---- apply: the list streams in four slots ahead of its use (coalesced: slot-major), everything else is LDS
            constexpr int PF = 4;
            int jn[PF];
            double2 fn[PF];
            const int nloop = has_row ? max(cnt, 0) : 0;
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int s = min(k, nloop - 1);
                if (nloop > 0) { jn[k] = myidx[(size_t)s * NT]; fn[k] = myfh[(size_t)s * NT]; }
            }
            for (int s0 = 0; s0 < nloop; s0 += PF) {
                int jc[PF];
                double2 fc[PF];
#pragma unroll
                for (int k = 0; k < PF; ++k) { jc[k] = jn[k]; fc[k] = fn[k]; }
#pragma unroll
                for (int k = 0; k < PF; ++k) {                       // the next four, clamped to the last valid slot
                    const int s = min(s0 + PF + k, nloop - 1);
                    jn[k] = myidx[(size_t)s * NT]; fn[k] = myfh[(size_t)s * NT];
                }
#pragma unroll
                for (int k = 0; k < PF; ++k) {
                    const double *pj = S + (size_t)jc[k] * REC;
                    const bool ok = s0 + k < nloop;
                    const double fk = ok ? fc[k].x : 0.0, hk = ok ? fc[k].y : 0.0;
                    const double dx = pix - pj[0], dy = piy - pj[1], dz = piz - pj[2];
                    const double rd = (dx * pj[3] + dy * pj[4] + dz * pj[5]) * hk;
                    u[0] += fk * pj[3] + rd * dx; u[1] += fk * pj[4] + rd * dy; u[2] += fk * pj[5] + rd * dz;
                }
            }
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { u[a] += __shfl_xor(u[a], o, 64); if (TWO) w2[a] += __shfl_xor(w2[a], o, 64); }
        }
        if (has_row && q == 0 && cnt < 0) {     // list overflow: the whole row from the cells
            u[0] = u[1] = u[2] = 0.0; w2[0] = w2[1] = w2[2] = 0.0;
            row_walk_cells<TWO>(i, pos_s, vec_s, vec2_s, cell_off, box, nc, rcut2, coef_g, u, w2);
        }
        if (has_row && q == 0) {
            out_s[i] = make_double4(self * xi.x + u[0], self * xi.y + u[1], self * xi.z + u[2], 0.0);
            if (TWO) out2_s[i] = make_double4(self * x2.x + w2[0], self * x2.y + w2[1], self * x2.z + w2[2], 0.0);
        }
    }
    double sa = 0.0, sb = 0.0, sc = 0.0;
    if (fits) {
        if (FUSE && has_row && q == 0) {
            sa = xi.x * xi.x + xi.y * xi.y + xi.z * xi.z;
            sb = xi.x * (self * xi.x + u[0]) + xi.y * (self * xi.y + u[1]) + xi.z * (self * xi.z + u[2]);
            if (lz.vprev) { const double4 m = lz.vprev[i]; sc = xi.x * m.x + xi.y * m.y + xi.z * m.z; }
        }
    } else {
        // ---- the tile does not fit (a dense cluster): every row of the block from the cells in global memory
        for (int r = tid; r < nrows; r += NT) {
            int lo = 0, hi = NI;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_rpre[mid] <= r) lo = mid; else hi = mid; }
            const int hh = s_rhid[lo];
            const int qz = hh % hz, r3 = hh / hz, qy = r3 % hy, qx = r3 / hy;          // an inner cell: no wrap
            const int c = ((c0[0] + qx - 1) * nc.ny + c0[1] + qy - 1) * nc.nz + c0[2] + qz - 1;
            const int ii = cell_off[c] + (r - s_rpre[lo]);
            double uu[3] = {0.0, 0.0, 0.0}, ww[3] = {0.0, 0.0, 0.0};
            row_walk_cells<TWO>(ii, pos_s, vec_s, vec2_s, cell_off, box, nc, rcut2, coef_g, uu, ww);
            const double4 x = vec_s[ii];
            const double yx = self * x.x + uu[0], yy = self * x.y + uu[1], yz = self * x.z + uu[2];
            out_s[ii] = make_double4(yx, yy, yz, 0.0);
            if (TWO) { const double4 g = vec2_s[ii]; out2_s[ii] = make_double4(self * g.x + ww[0], self * g.y + ww[1], self * g.z + ww[2], 0.0); }
            if (BUILD) nb.cnt[ii] = -1;
            if (FUSE) {
                sa += x.x * x.x + x.y * x.y + x.z * x.z;
                sb += x.x * yx + x.y * yy + x.z * yz;
                if (lz.vprev) { const double4 m = lz.vprev[ii]; sc += x.x * m.x + x.y * m.y + x.z * m.z; }
            }
        }
    }
    if (FUSE) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); sc += __shfl_xor(sc, o, 64); }
        if (lane == 0) { s_red[wave] = sa; s_red[4 + wave] = sb; s_red[8 + wave] = sc; }
        __syncthreads();
        if (tid == 0) {
            const int bi = blockIdx.x;
            lz.partials[bi] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
            lz.partials[lz.npart_cap + bi] = s_red[4] + s_red[5] + s_red[6] + s_red[7];
            lz.partials[2 * lz.npart_cap + bi] = s_red[8] + s_red[9] + s_red[10] + s_red[11];
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------
static size_t lds_fixed_bytes(int cap_st) { return (size_t)cap_st * sizeof(int) + (260 + 68 * 2 + 8) * sizeof(int); }
size_t nb_blocks_lds_bytes(const NbBlocks &B, int nint, bool build, bool two) {
    return (size_t)B.cap_st * (two ? 9 : 6) * sizeof(double) + (build ? (size_t)nint * CSTRIDE * sizeof(double) : 0) + lds_fixed_bytes(B.cap_st);
}

// block shape for a grid of nc cells holding n particles: as many rows per workgroup as it has lanes for (counting the
// partly filled blocks at the edges), within the tile capacity (with room for fluctuations)
void nb_blocks_plan(NbBlocks &B, const DCells &nc, int ncx, double n, int slots_needed, int nint) {
    B.on = 0;
    if (nc.nx < 3 || nc.ny < 3 || nc.nz < 3) return;     // fewer than three cells along an axis: minimum-image search (legacy kernels)
    // the apply pass (one per Lanczos iteration) holds the tile only: three workgroups per CU
    const int budget = 160 * 1024 / 3 - 64;
    B.cap_st = std::min(5 * NB_NT, (int)((budget - (260 + 68 * 2 + 8) * (int)sizeof(int)) / (6 * sizeof(double) + sizeof(int)))) & ~15;
    const double per_cell = n / ((double)nc.nx * nc.ny * nc.nz);
    double best = 0.0;
    B.bx = B.by = B.bz = 1;
    for (int bx = 1; bx <= 4; ++bx)
        for (int by = 1; by <= 4; ++by)
            for (int bz = 1; bz <= 4; ++bz) {
                const double halo = (bx + 2.0) * (by + 2.0) * (bz + 2.0) * per_cell, rows = bx * by * bz * per_cell;
                if (halo + 5.0 * std::sqrt(halo) + 16 > B.cap_st || rows + 3.5 * std::sqrt(rows) + 4 > NB_NT) continue;
                if ((bx + 2) * (by + 2) * (bz + 2) > 256 || bx * by * bz > 64) continue;
                const double nblk = (double)((ncx + bx - 1) / bx) * ((nc.ny + by - 1) / by) * ((nc.nz + bz - 1) / bz);
                // rows a workgroup really gets, discounted by the tile it has to stage for them
                const double score = (n * ncx / nc.nx / nblk) / std::sqrt((bx + 2.0) * (by + 2.0) * (bz + 2.0) / (bx * by * bz)) + 0.01 * bz;
                if (score > best) { best = score; B.bx = bx; B.by = by; B.bz = bz; }
            }
    if (best == 0.0) { if (27.0 * per_cell + 5.0 * std::sqrt(27.0 * per_cell) + 16 > B.cap_st) return; }
    const double rows = B.bx * B.by * B.bz * per_cell;
    B.tpr = 1;
    while (B.tpr < 4 && (rows + 3.5 * std::sqrt(rows) + 4) * (2 * B.tpr) <= NB_NT) B.tpr *= 2;
    B.cap = std::min(255, (slots_needed + B.tpr - 1) / B.tpr + (B.tpr > 1 ? 8 : 0));
    B.on = 1;
}
int nb_blocks_count(const NbBlocks &B, const DCells &nc, int ncx) {
    return ((ncx + B.bx - 1) / B.bx) * ((nc.ny + B.by - 1) / B.by) * ((nc.nz + B.bz - 1) / B.bz);
}

template <bool BUILD, bool TWO, bool FUSE>
static void launch_blocks_t(const double4 *pos_s, const double4 *vec_s, double4 *out_s, const double4 *vec2_s, double4 *out2_s,
                            const int *cell_off, DBox box, DCells nc, double rcut, double self, const double *coef, int nint,
                            NbList nb, LzFuse lz, hipStream_t s) {
    const NbBlocks &B = nb.blk;
    const int nbby = (nc.ny + B.by - 1) / B.by, nbbz = (nc.nz + B.bz - 1) / B.bz, nblk = nb_blocks_count(B, nc, B.ncx);
    const size_t lds = nb_blocks_lds_bytes(B, nint, BUILD, TWO);
    auto go = [&](auto kern) {
        static size_t attr_lds = 48 * 1024;   // raise the dynamic LDS limit of this instantiation when a launch needs more
        if (lds > attr_lds) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_lds = lds; }
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(NB_NT), lds, s, pos_s, vec_s, out_s, vec2_s, out2_s, cell_off, box, nc, rcut * rcut, self,
                           coef, nint, nb, lz, nbby, nbbz);
    };
    switch (B.tpr) {
        case 1: go(k_mreal_blocks<BUILD, TWO, FUSE, 1>); break;
        case 2: go(k_mreal_blocks<BUILD, TWO, FUSE, 2>); break;
        default: go(k_mreal_blocks<BUILD, TWO, FUSE, 4>); break;
    }
}

// out = M_real vec on the cell blocks.  build: scan and (re)write the pair list first; vec2/out2: a second vector in the same
// pass (build only); lz.partials: fuse the Lanczos sums (apply only)
void launch_mreal_blocks(const double4 *pos_s, const double4 *vec_s, double4 *out_s, const double4 *vec2_s, double4 *out2_s,
                         const int *cell_off, DBox box, DCells nc, double rcut, double self, const double *coef, int nint, NbList nb,
                         bool build, LzFuse lz, hipStream_t s) {
    if (build) {
        if (vec2_s) launch_blocks_t<true, true, false>(pos_s, vec_s, out_s, vec2_s, out2_s, cell_off, box, nc, rcut, self, coef, nint, nb, lz, s);
        else launch_blocks_t<true, false, false>(pos_s, vec_s, out_s, nullptr, nullptr, cell_off, box, nc, rcut, self, coef, nint, nb, lz, s);
    } else if (lz.partials) {
        launch_blocks_t<false, false, true>(pos_s, vec_s, out_s, nullptr, nullptr, cell_off, box, nc, rcut, self, coef, nint, nb, lz, s);
    } else {
        launch_blocks_t<false, false, false>(pos_s, vec_s, out_s, nullptr, nullptr, cell_off, box, nc, rcut, self, coef, nint, nb, lz, s);
    }
}

}  // namespace pse
