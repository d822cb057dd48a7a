// Far field of the PSE engine on gfx950: particle -> grid spreading (K2+K3 gpu_stokes_ZeroGrid/Spread_kernel,
// PSEv1/Helper.cu:87-97, PSEv1/Mobility.cu:114-252) and grid -> particle gathering (K8 gpu_stokes_Contract_kernel,
// PSEv1/Mobility.cu:325-477), fp64.
//
// Both work from 64-byte particle records written once per step in BIN order (bin = the 8^3 block of nodes the support
// origin lies in): origin, sorted index, offset of the origin from the particle in grid units, prefac * force.  Nothing
// else about a particle crosses HBM: the separable Gaussian weights are rebuilt where they are used from six
// exponentials per particle (Gaussian recurrence).
//
// Separable weights under shear.  Node (tx,ty,tz) of a support sits at  ex = u + hx tx + s hy ty,  ey = Y0 + hy ty,
// ez = Z0 + hz tz  from the particle (s = xy, u = hx d0x + s hy d0y; PSEv1/Mobility.cu:223-230), so
//   exp(-c (ex^2 + ey^2 + ez^2)) = ax[tx] ay[ty] az[tz] K[tx][ty]
//   ax[t] = exp(-c (u + hx t)^2)                       ay[t] = exp(-c ((Y0 + hy t)^2 + s^2 hy^2 t^2 + 2 u s hy t))
//   az[t] = exp(-c (Z0 + hz t)^2)                      K[t][v] = exp(-2 c s hx hy t v)   (the same for every particle)
// and each of ax, ay, az obeys E(t+1) = E(t) q r_t with r_t independent of the particle (GaussConsts).
#include "pse_kernels.h"
#include "pse_farbin.h"

#include <hipcub/hipcub.hpp>
#include <type_traits>

namespace pse {

static inline int nblocks(long n, int tpb) { return (int)((n + tpb - 1) / tpb); }


struct FarRec {
    int ox, oy, oz;          // support origin (first node per axis), wrapped into the grid
    unsigned idx;            // index in the cell-sorted arrays; bit 31: owned by another slab rank
    double d0x, d0y, d0z;    // origin - particle, grid units
    double fx, fy, fz;       // prefac * force
};
static_assert(sizeof(FarRec) == 64, "record is four 16-byte loads");

// one axis: E(t) for t = 0..P-1 from E(0) = exp(e0), ratio exp(lq) r_t
template <int P>
__device__ __forceinline__ void gauss_axis(double e0, double lq, const double *__restrict__ r, double (&a)[P]) {
    double e = exp_lean(e0);
    const double q = exp_lean(lq);
#pragma unroll
    for (int t = 0; t < P; ++t) { a[t] = e; if (t + 1 < P) e *= q * r[t]; }
}

template <int P>
__device__ __forceinline__ void gauss_tables(double d0x, double d0y, double d0z, const DGrid &G, double s, const GaussConsts &gc,
                                             double (&ax)[P], double (&ay)[P], double (&az)[P]) {
    const double c = G.expfac;
    const double Y0 = G.hy * d0y, Z0 = G.hz * d0z, u = G.hx * d0x + s * Y0;
    double e = exp_lean(-c * u * u), q = exp_lean(-2.0 * c * G.hx * u);
#pragma unroll
    for (int t = 0; t < P; ++t) { ax[t] = e; if (t + 1 < P) e *= q * gc.rx[t]; }
    e = exp_lean(-c * Y0 * Y0); q = exp_lean(-2.0 * c * G.hy * (Y0 + s * u));
#pragma unroll
    for (int t = 0; t < P; ++t) { ay[t] = e; if (t + 1 < P) e *= q * gc.ry[t]; }
    e = exp_lean(-c * Z0 * Z0); q = exp_lean(-2.0 * c * G.hz * Z0);
#pragma unroll
    for (int t = 0; t < P; ++t) { az[t] = e; if (t + 1 < P) e *= q * gc.rz[t]; }
}

GaussConsts gauss_consts(const DGrid &G, double xy) {
    GaussConsts gc;
    const double c = G.expfac;
    for (int t = 0; t < FAR_PMAX - 1; ++t) {
        gc.rx[t] = std::exp(-c * G.hx * G.hx * (2 * t + 1));
        gc.ry[t] = std::exp(-c * G.hy * G.hy * (1.0 + xy * xy) * (2 * t + 1));
        gc.rz[t] = std::exp(-c * G.hz * G.hz * (2 * t + 1));
    }
    gc.lnk = -2.0 * c * xy * G.hx * G.hy;
    gc.s = xy;
    return gc;
}

// n / d and n % d for n d < 2^31 by one multiply-high with a host-made reciprocal (a runtime divisor costs ~30 scalar
// instructions per division; every workgroup decodes its block index with two of them)
struct FastDiv { unsigned m, d; };
static FastDiv fast_div(int d) { return FastDiv{(unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d), (unsigned)d}; }
__device__ __forceinline__ int fdiv(int n, FastDiv q, int &rem) {
    const int k = (int)__umulhi((unsigned)n, q.m);
    rem = n - k * (int)q.d;
    return k;
}

// ---- binning: far_bin_particle (pse_farbin.h), called by the gather-into-cell-order pass k_permute -------------------------
size_t bin_scan_temp_bytes(size_t nbins) {
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const int *)nullptr, (int *)nullptr, (int)(nbins + 1));
    return bytes;
}

bool farfield_fast_path(const DGrid &G);
// the records, in bin order (the support is recomputed from the sorted position: nothing was parked for this kernel)
__global__ void __launch_bounds__(256)
k_far_records(const double4 *__restrict__ pos_s, const double4 *__restrict__ f_s, int N, DGrid G, DBox box,
              FarBins fb, FarRec *__restrict__ rec) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const int rank = fb.rank_s[p];
    if (rank < 0) return;
    const double4 q = pos_s[p];
    double fx, fy, fz;
    frac_coords(box, q.x, q.y, q.z, fx, fy, fz);
    int4 sp;
    double4 d0;
    far_support(fx, fy, fz, G, sp, d0);
    const double4 F = f_s[p];
    const int slot = fb.off[bin_index(sp.x, sp.y, sp.z, fb)] + rank;
    const bool owned = G.nxl == G.Nx || wrapi(sp.w - G.x0, G.Nx) < G.nxl;
    int4 h = make_int4(sp.x, sp.y, sp.z, (int)((unsigned)p | (owned ? 0u : 0x80000000u)));
    double2 *o = reinterpret_cast<double2 *>(rec + slot);
    o[0] = *reinterpret_cast<double2 *>(&h);
    o[1] = make_double2(d0.x, d0.y);
    o[2] = make_double2(d0.z, G.prefac * F.x);
    o[3] = make_double2(G.prefac * F.y, G.prefac * F.z);
}

// bin offsets + records; the bin counts and ranks come from the gather-into-cell-order pass (far_bin_args / k_permute)
hipError_t launch_far_records(const double4 *pos_s, const double4 *f_s, int N, DGrid G, DBox box, SpreadWork w, hipStream_t s) {
    if (!farfield_fast_path(G) || !w.rec_t) return hipSuccess;
    FarBins fb = w.fb;
    fb.nbx = bins_of(G.Nx); fb.nby = bins_of(G.Ny); fb.nbz = bins_of(G.Nz);
    const int nbins = fb.nbx * fb.nby * fb.nbz;
    size_t tb = fb.tmp_bytes;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(fb.tmp, tb, fb.cnt, fb.off, nbins + 1, s);   // cnt[nbins] = 0: off[nbins] = total
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_far_records, dim3(nblocks(N, 256)), dim3(256), 0, s, pos_s, f_s, N, G, box, fb, w.rec_t);
    return hipGetLastError();
}
// what the gather-into-cell-order pass needs to rank the particles in their bins (on = false: the generic far field, no bins)
FarBinArgs far_bin_args(const DGrid &G, const SpreadWork &w) {
    FarBinArgs a{};
    a.on = farfield_fast_path(G) && w.rec_t != nullptr;
    a.G = G;
    a.fb = w.fb;
    a.fb.nbx = bins_of(G.Nx); a.fb.nby = bins_of(G.Ny); a.fb.nbz = bins_of(G.Nz);
    return a;
}
size_t far_bin_count(const DGrid &G) { return (size_t)bins_of(G.Nx) * bins_of(G.Ny) * bins_of(G.Nz) + 1; }

// ---- spread ------------------------------------------------------------------------------------------------------
// Register-accumulating spread: a wavefront owns an 8 x 8 x TZ block of grid nodes -- lane = (x,y) column, TZ x 3
// accumulators per lane in registers -- and streams past it every particle whose support reaches the block.  No atomics of
// any kind (the LDS ds_add_f64 version of round 1 retired 3-7 lanes per clock and CU, 90 % of its time), every node is
// written exactly once with plain stores, so there is no ZeroGrid pass either.
//
// Per chunk of 64 candidate records (lane = candidate): clip against the block, rebuild the separable Gaussian tables
// of the survivors (six exponentials each) and park them in LDS: ax, ay as zero-padded columns (index-major, stride 65:
// bank-conflict free both for the lane-per-particle writes and the lane-per-column reads), az and the force as plain
// columns.  Then, particle by particle (wave-uniform): each lane looks up ax[lx - ox], ay[ly - oy] (zero outside the
// support: no branch), and the z extent of the overlap -- known at compile time inside each case of a switch on the
// particle's z offset -- is accumulated with 3 FMAs per node against broadcast reads of az.
// compile-time loop: body(integral_constant<int, I>) for I = 0..N-1 -- register arrays are only ever indexed by constants
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int LO, int HI, class F>
__device__ __forceinline__ void uniform_dispatch(int k, F &&f) {
    if constexpr (HI - LO == 1) {
        f(std::integral_constant<int, LO>{});
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (k < MID) uniform_dispatch<LO, MID>(k, f);
        else uniform_dispatch<MID, HI>(k, f);
    }
}

// NW > 1 (small grids: fewer blocks than SIMDs): NW waves share a block and split its candidate chunks among themselves, each
// with its own tables; their accumulators are summed through LDS at the end (in wave order: the result does not depend on timing).
template <int P, int TZ, bool SHEAR, int NW>
__global__ void __launch_bounds__(64 * NW)
k_spread_tiles(const FarRec *__restrict__ rec, FarBins fb, double *__restrict__ gx, double *__restrict__ gy,
               double *__restrict__ gz, DGrid G, GaussConsts gc, FastDiv dz, FastDiv dy) {
    constexpr int TX = 8, TY = 8, PT = P + 1, LS = 65, RMAX = P <= 8 ? 20 : 32;   // runs: bins in x times bins in y times z parts
    constexpr int UB = (P + 4) & ~1;          // doubles per particle of the wave-uniform block: az[P], force[3], pad to 16 bytes
    constexpr int WB = 2 * PT * LS + UB * 64 + ((2 * PT * LS) & 1);   // doubles of one wave's tables (s_u 16-byte aligned)
    constexpr int RED = (NW - 1) * TZ * 3 * 64;                        // the other waves' accumulators, parked over the tables
    __shared__ __attribute__((aligned(16))) double s_tab[NW * WB > RED ? NW * WB : RED];
    constexpr int KS = PT <= 8 ? 8 : 16;      // row stride of the shear table (columns 0 .. P): 448 bytes at P = 6, so that the sheared
                                              // kernel keeps the twelve workgroups per CU of the unsheared one
    __shared__ double s_k[SHEAR ? PT * KS : 1];
    __shared__ int s_rb[RMAX], s_ro[RMAX + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lx = lane & 7, ly = lane >> 3;
    double *const s_ax = s_tab + wv * WB, *const s_ay = s_ax + PT * LS, *const s_u = s_tab + wv * WB + (WB - UB * 64);
    // tables are private to a wave: inside the chunk loop a wave only has to order its own LDS accesses (they execute in issue
    // order); a workgroup barrier there would also deadlock, the waves run different numbers of chunks
    auto wave_sync = [] () __attribute__((always_inline)) {
        if (NW == 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    };
    int tz_, ty_;
    const int tx_ = fdiv(fdiv(xcd_block(blockIdx.x, gridDim.x), dz, tz_), dy, ty_);
    const int t0[3] = {G.x0 + tx_ * TX, ty_ * TY, tz_ * TZ};
    const int ext[3] = {min(TX, G.x0 + G.nxl - t0[0]), min(TY, G.Ny - t0[1]), min(TZ, G.Nz - t0[2])};
    const int Nn[3] = {G.Nx, G.Ny, G.Nz};
    const int nb[3] = {fb.nbx, fb.nby, fb.nbz};

    if (SHEAR)   // K[t][v] = exp(-2 c s hx hy t v) on the padded index grid (the pad row / column P multiplies a zero)
        for (int e = threadIdx.x; e < PT * KS; e += 64 * NW) s_k[e] = exp_lean(gc.lnk * (double)((e / KS) * (e % KS)));
    // bins whose origins [t0 - P + 1, t0 + ext - 1] (cyclic) can reach the block; consecutive z bins are one record range
    int blo[3], bcnt[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        blo[a] = wrapi(t0[a] - (P - 1), Nn[a]) / BIN;
        bcnt[a] = min(nb[a], wrapi((t0[a] + ext[a] - 1) / BIN - blo[a], nb[a]) + 1);
    }
    const int zparts = blo[2] + bcnt[2] > nb[2] ? 2 : 1;
    const int nr = bcnt[0] * bcnt[1] * zparts;
    if ((int)threadIdx.x < nr) {
        const int zp = lane % zparts, r = lane / zparts, iy = r % bcnt[1], ix = r / bcnt[1];
        const int row = (((blo[0] + ix) % nb[0]) * nb[1] + (blo[1] + iy) % nb[1]) * nb[2];
        int z0 = blo[2], z1 = blo[2] + bcnt[2];                      // [z0, z1) bins, before the wrap
        if (zparts == 2) { if (zp == 0) z1 = nb[2]; else { z0 = 0; z1 = blo[2] + bcnt[2] - nb[2]; } }
        const int o = fb.off[row + z0];
        s_rb[lane] = o;
        s_ro[lane + 1] = fb.off[row + z1] - o;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        s_ro[0] = 0;
        for (int r = 0; r < nr; ++r) { run += s_ro[r + 1]; s_ro[r + 1] = run; }
    }
    __syncthreads();
    const int total = s_ro[nr];

    double acc[TZ][3];
    static_for<TZ>([&](auto zc) __attribute__((always_inline)) { constexpr int z = decltype(zc)::value; acc[z][0] = acc[z][1] = acc[z][2] = 0.0; });

    const unsigned lxb = (unsigned)(lx * LS * 8), lyb = (unsigned)(ly * LS * 8);   // byte offset of row lx / ly
    // record of candidate k: the runs are few, a scan of independent (broadcast) LDS reads finds its run
    double2 n0, n1, n2, n3;
    auto fetch = [&](int c0) __attribute__((always_inline)) {
        const int k = c0 + lane;
        int r = 0;
        for (int q = 1; q < nr; ++q) r = s_ro[q] <= k ? q : r;
        const size_t slot = k < total ? (size_t)(s_rb[r] + (k - s_ro[r])) : (size_t)s_rb[0];
        const double2 *rp = reinterpret_cast<const double2 *>(rec + slot);
        n0 = rp[0]; n1 = rp[1]; n2 = rp[2]; n3 = rp[3];
    };
    if (64 * wv < total) fetch(64 * wv);
    for (int c0 = 64 * wv; c0 < total; c0 += 64 * NW) {
        const bool valid = c0 + lane < total;
        const double2 q0 = n0, q1 = n1, q2 = n2, q3 = n3;
        if (c0 + 64 * NW < total) fetch(c0 + 64 * NW);               // the next chunk's records are in flight during this one
        const int4 hd = *reinterpret_cast<const int4 *>(&q0);
        const int o[3] = {hd.x, hd.y, hd.z};
        int rel[3];
        bool hit = valid;
#pragma unroll
        for (int a = 0; a < 3; ++a) {   // support origin relative to the block (nearest image)
            int d = o[a] - t0[a];
            if (d < -Nn[a] / 2) d += Nn[a]; else if (d >= Nn[a] - Nn[a] / 2) d -= Nn[a];
            rel[a] = d;
            hit = hit && d + P > 0 && d < ext[a];
        }
        unsigned long long mask = __ballot(hit);
        if (mask == 0ull) continue;
        wave_sync();                                                  // the previous chunk's tables are no longer read
        {   // the survivors' separable weights, axis by axis (six exponentials per particle), parked in LDS as they come
            const double c = G.expfac;
            const double Y0 = G.hy * q1.y, Z0 = G.hz * q2.x, u = G.hx * q1.x + gc.s * Y0;
            double a[P];
            gauss_axis<P>(-c * u * u, -2.0 * c * G.hx * u, gc.rx, a);
            if (hit) {
                s_ax[P * LS + lane] = 0.0;
#pragma unroll
                for (int t = 0; t < P; ++t) s_ax[t * LS + lane] = a[t];
            }
            gauss_axis<P>(-c * Y0 * Y0, -2.0 * c * G.hy * (Y0 + gc.s * u), gc.ry, a);
            if (hit) {
                s_ay[P * LS + lane] = 0.0;
#pragma unroll
                for (int t = 0; t < P; ++t) s_ay[t * LS + lane] = a[t];
            }
            gauss_axis<P>(-c * Z0 * Z0, -2.0 * c * G.hz * Z0, gc.rz, a);
            if (hit) {
                double ub[UB];
#pragma unroll
                for (int t = 0; t < UB; ++t) ub[t] = t < P ? a[t] : 0.0;
                ub[P] = q2.y; ub[P + 1] = q3.x; ub[P + 2] = q3.y;
                double2 *up = reinterpret_cast<double2 *>(s_u + lane * UB);
#pragma unroll
                for (int t = 0; t < UB / 2; ++t) up[t] = make_double2(ub[2 * t], ub[2 * t + 1]);
            }
        }
        // byte offsets of the particle's table rows relative to lane 0's view: row (lx - ox) of column `lane`
        const int metax = rel[0] * (LS * 8) - lane * 8, metay = rel[1] * (LS * 8) - lane * 8;
        const int kzl = hit ? rel[2] + P - 1 : -1;                    // z offset class of this lane's particle
        wave_sync();
        // One loop per z offset class: inside it the z extent of the overlap is a compile-time range, and the accumulators are
        // loop-carried values updated in place (a switch inside one loop made the compiler copy all TZ x 3 accumulators at the
        // merge of its cases: ~90 v_mov_b64 per particle).
        static_for<TZ + P - 1>([&](auto kc) __attribute__((always_inline)) {
            constexpr int KZ = decltype(kc)::value, OZ = KZ - (P - 1);   // z of the support's first node inside the block
            unsigned long long m = __ballot(kzl == KZ);
            while (m) {
                const int p = __ffsll((long long)m) - 1;
                m &= m - 1ull;
                // byte address of this lane's row in the particle's column: rows 0 .. P - 1 hold the weights, row P is the zero pad; a
                // row below 0 wraps to a huge unsigned and is clamped to the pad like a row above P - 1 (one pad row: the table of a
                // wave is 1 KB smaller than with a pad at either end, and twelve workgroups share a CU instead of eleven)
                const unsigned mx = (unsigned)__builtin_amdgcn_readlane(metax, p), my = (unsigned)__builtin_amdgcn_readlane(metay, p);
                const unsigned cap = (unsigned)(P * LS * 8 + p * 8);
                const unsigned bxo = min(lxb - mx, cap), byo = min(lyb - my, cap);
                double w = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(s_ax) + bxo) *
                           *reinterpret_cast<const double *>(reinterpret_cast<const char *>(s_ay) + byo);
                if (SHEAR) w *= s_k[(bxo - p * 8) / (LS * 8) * KS + (byo - p * 8) / (LS * 8)];
                // the particle's wave-uniform block: 16-byte broadcast reads, only the pairs this class needs
                const double2 *up = reinterpret_cast<const double2 *>(s_u + p * UB);
                double ub[UB];
                static_for<UB / 2>([&](auto hc) __attribute__((always_inline)) {
                    constexpr int h = decltype(hc)::value;
                    constexpr bool need = (2 * h + 1 >= P) || (OZ + 2 * h + 1 >= 0 && OZ + 2 * h < TZ);
                    if constexpr (need) { const double2 v = up[h]; ub[2 * h] = v.x; ub[2 * h + 1] = v.y; }
                });
                const double wx = w * ub[P], wy = w * ub[P + 1], wz = w * ub[P + 2];
                static_for<P>([&](auto tc) __attribute__((always_inline)) {
                    constexpr int t = decltype(tc)::value, z = OZ + t;
                    if constexpr (z >= 0 && z < TZ) {
                        const double a = ub[t];
                        acc[z][0] = fma(wx, a, acc[z][0]);
                        acc[z][1] = fma(wy, a, acc[z][1]);
                        acc[z][2] = fma(wz, a, acc[z][2]);
                    }
                });
            }
        });
    }
    if (NW > 1) {
        __syncthreads();                                              // every wave is through its chunks: the tables are dead
        if (wv > 0)
            static_for<TZ * 3>([&](auto ec) __attribute__((always_inline)) {
                constexpr int e = decltype(ec)::value;
                s_tab[((wv - 1) * TZ * 3 + e) * 64 + lane] = acc[e / 3][e % 3];
            });
        __syncthreads();
        if (wv > 0) return;
        for (int w = 0; w < NW - 1; ++w)
            static_for<TZ * 3>([&](auto ec) __attribute__((always_inline)) {
                constexpr int e = decltype(ec)::value;
                acc[e / 3][e % 3] += s_tab[(w * TZ * 3 + e) * 64 + lane];
            });
    }
    if (lx < ext[0] && ly < ext[1]) {
        const size_t base = ((size_t)(t0[0] - G.x0 + G.hl + lx) * G.Ny + (t0[1] + ly)) * G.Nz + t0[2];
        if ((G.Nz & 1) == 0) {
            static_for<TZ / 2>([&](auto zc) __attribute__((always_inline)) {
                constexpr int z = 2 * decltype(zc)::value;
                if (z + 1 < ext[2]) {
                    *reinterpret_cast<double2 *>(gx + base + z) = make_double2(acc[z][0], acc[z + 1][0]);
                    *reinterpret_cast<double2 *>(gy + base + z) = make_double2(acc[z][1], acc[z + 1][1]);
                    *reinterpret_cast<double2 *>(gz + base + z) = make_double2(acc[z][2], acc[z + 1][2]);
                } else if (z < ext[2]) {
                    gx[base + z] = acc[z][0]; gy[base + z] = acc[z][1]; gz[base + z] = acc[z][2];
                }
            });
        } else {
            static_for<TZ>([&](auto zc) __attribute__((always_inline)) {
                constexpr int z = decltype(zc)::value;
                if (z < ext[2]) { gx[base + z] = acc[z][0]; gy[base + z] = acc[z][1]; gz[base + z] = acc[z][2]; }
            });
        }
    }
}

// v0 / generic support: one wave per particle, hardware fp64 atomics into the three real grids (zeroed by the caller).
__global__ void __launch_bounds__(256)
k_spread_atomic(const double4 *__restrict__ pos_s, const double4 *__restrict__ f_s, int N, double *__restrict__ gx,
                double *__restrict__ gy, double *__restrict__ gz, DGrid G, DBox box, CellRanges rows,
                const int *__restrict__ cell_off) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= N || !rows.row(p, cell_off)) return;   // a slab rank holds particle data for its own and its ghost rows only
    const double4 pp = pos_s[p];
    const double4 F = f_s[p];
    double fx, fy, fz;
    frac_coords(box, pp.x, pp.y, pp.z, fx, fy, fz);
    int sx, sy, sz;
    double d0x, d0y, d0z;
    support_start(fx, G.Nx, G.P, sx, d0x);
    support_start(fy, G.Ny, G.P, sy, d0y);
    support_start(fz, G.Nz, G.P, sz, d0z);
    const int P = G.P, P2 = P * P, P3 = P2 * P;
    for (int n = lane; n < P3; n += 64) {
        const int tx = n / P2, ty = (n - tx * P2) / P, tz = n - tx * P2 - ty * P;
        int ix = sx + tx; ix = ix < 0 ? ix + G.Nx : (ix >= G.Nx ? ix - G.Nx : ix);
        const int lx = ix - G.x0;
        if (lx < 0 || lx >= G.nxl) continue;
        int iy = sy + ty; iy = iy < 0 ? iy + G.Ny : (iy >= G.Ny ? iy - G.Ny : iy);
        int iz = sz + tz; iz = iz < 0 ? iz + G.Nz : (iz >= G.Nz ? iz - G.Nz : iz);
        const double ey = G.hy * (d0y + ty);
        const double ex = G.hx * (d0x + tx) + box.xy * ey;   // sheared lattice (PSEv1/Mobility.cu:230)
        const double ez = G.hz * (d0z + tz);
        const double w = G.prefac * exp_neg(-G.expfac * (ex * ex + ey * ey + ez * ez));
        const size_t idx = ((size_t)(lx + G.hl) * G.Ny + iy) * G.Nz + iz;
        unsafeAtomicAdd(&gx[idx], w * F.x);
        unsafeAtomicAdd(&gy[idx], w * F.y);
        unsafeAtomicAdd(&gz[idx], w * F.z);
    }
}

static int spread_tz(const DGrid &G, int force) {
    if (force == 8 || force == 16) return G.Nz >= 2 * force ? force : 8;
    // a wave per block: small grids need the smaller blocks to occupy the chip at all
    const long blocks16 = (long)((G.nxl + 7) / 8) * ((G.Ny + 7) / 8) * ((G.Nz + 15) / 16);
    return G.Nz >= 64 && blocks16 >= 2048 ? 16 : 8;
}

bool farfield_fast_path(const DGrid &G) {
    // the block kernels resolve a support to its nearest image of the block (needs N >= 2 max(block, support) per axis)
    const int need = 2 * std::max(8, G.P);
    return G.P >= 4 && G.P <= FAR_PMAX && G.Nx >= need && G.Ny >= need && G.Nz >= need;
}
bool spread_needs_zero(const DGrid &G) { return !farfield_fast_path(G); }
size_t farfield_bins(const DGrid &G) { return (size_t)bins_of(G.Nx) * bins_of(G.Ny) * bins_of(G.Nz); }

template <int P, int TZ, int NW>
static void launch_spread_pt(const FarRec *rec, FarBins fb, double *gx, double *gy, double *gz, const DGrid &G, const GaussConsts &gc,
                             hipStream_t s) {
    const int ntx = (G.nxl + 7) / 8, nty = (G.Ny + 7) / 8, ntz = (G.Nz + TZ - 1) / TZ;
    const dim3 g(ntx * nty * ntz), b(64 * NW);
    const FastDiv dz = fast_div(ntz), dy = fast_div(nty);
    if (gc.s != 0.0) hipLaunchKernelGGL((k_spread_tiles<P, TZ, true, NW>), g, b, 0, s, rec, fb, gx, gy, gz, G, gc, dz, dy);
    else hipLaunchKernelGGL((k_spread_tiles<P, TZ, false, NW>), g, b, 0, s, rec, fb, gx, gy, gz, G, gc, dz, dy);
}
constexpr int SPREAD_NW = 4;   // waves per block on small grids
template <int P>
static void launch_spread_p(const FarRec *rec, FarBins fb, double *gx, double *gy, double *gz, const DGrid &G, const GaussConsts &gc,
                            int force_tz, int nw_env, hipStream_t s) {
    const long blocks8 = (long)((G.nxl + 7) / 8) * ((G.Ny + 7) / 8) * ((G.Nz + 7) / 8);
    if (spread_tz(G, force_tz) == 16) launch_spread_pt<P, 16, 1>(rec, fb, gx, gy, gz, G, gc, s);
    else if (P <= 8 && (nw_env ? nw_env > 1 : blocks8 < 1024)) launch_spread_pt<P, 8, SPREAD_NW>(rec, fb, gx, gy, gz, G, gc, s);   // fewer blocks than SIMDs
    else launch_spread_pt<P, 8, 1>(rec, fb, gx, gy, gz, G, gc, s);
}

hipError_t launch_spread(const double4 *pos_s, const double4 *f_s, int N, double *gx, double *gy, double *gz, DGrid G,
                         DBox box, SpreadWork w, hipStream_t s) {
    if (!farfield_fast_path(G) || !w.rec_t) {
        hipLaunchKernelGGL(k_spread_atomic, dim3(nblocks(N, 4)), dim3(256), 0, s, pos_s, f_s, N, gx, gy, gz, G, box, w.need, w.cell_off);
        return hipGetLastError();
    }
    FarBins fb = w.fb;   // the records of this step are in place (launch_far_records)
    fb.nbx = bins_of(G.Nx); fb.nby = bins_of(G.Ny); fb.nbz = bins_of(G.Nz);
    const GaussConsts gc = gauss_consts(G, box.xy);
    switch (G.P) {
#define PSE_SPREAD_CASE(PV) case PV: launch_spread_p<PV>(w.rec_t, fb, gx, gy, gz, G, gc, w.force_tz, w.force_nw, s); break;
        PSE_SPREAD_CASE(4) PSE_SPREAD_CASE(5) PSE_SPREAD_CASE(6) PSE_SPREAD_CASE(7) PSE_SPREAD_CASE(8) PSE_SPREAD_CASE(9)
        PSE_SPREAD_CASE(10) PSE_SPREAD_CASE(11) PSE_SPREAD_CASE(12) PSE_SPREAD_CASE(13)
        default: launch_spread_p<14>(w.rec_t, fb, gx, gy, gz, G, gc, w.force_tz, w.force_nw, s); break;
#undef PSE_SPREAD_CASE
    }
    return hipGetLastError();
}

// ---- gather ------------------------------------------------------------------------------------------------------
// K8 gpu_stokes_Contract_kernel (PSEv1/Mobility.cu:325-477): u_p = h^3 sum_nodes prefac exp(-expfac r^2) u_grid.
// A workgroup takes one bin: the 8^3 nodes plus the P - 1 node halo on the high side of each axis (every support that
// starts in the bin lies inside) of all three velocity components are staged in LDS with 16-byte loads of whole rows (z rows
// padded to an even length; all loads are in flight before the first is consumed), while one lane per particle rebuilds
// the separable weights (six exponentials) into LDS.  Then four lanes per particle, each with two z offsets: the z window
// starts at an even node (a zero weight in front when the support starts at an odd one), so every LDS access is an aligned
// 16-byte read -- ax[tx] ay[ty] (K[tx][ty] under shear) against one read per (x, y) offset and component, az applied
// once at the end, reduction inside the quad by DPP -- instead of the reference's block per particle with a shared-memory
// tree over P^3 threads (PSEv1/Mobility.cu:456-470).  (P >= 8: eight lanes per particle.)
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v) {   // the value another lane of the quad holds
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// BZ (round 4): bins along z a workgroup takes together (their records are consecutive: z is the fastest bin index).  The cost of
// this kernel follows the GRID -- 0.30 ms at 256^3 with 30 particles per bin, 0.76 ms at 360^3 with 11: it is bound by the rate at
// which regions are loaded -- so where bins hold few particles two bins share a region of 13 x 13 x 22 nodes (0.79 of the loads per
// bin, 30 KB: five workgroups per CU).  At 30 particles per bin the single-bin shape wins (0.337 against 0.352 ms, round 2).
template <int P, bool SHEAR, int BZ = 1>
__global__ void __launch_bounds__(256)
k_gather_bins(const FarRec *__restrict__ rec, FarBins fb, FastDiv dz, FastDiv dy, int bx0, const double *__restrict__ gx,
              const double *__restrict__ gy, const double *__restrict__ gz, DGrid G, GaussConsts gc, double4 *__restrict__ u_s) {
    constexpr int NT = 256, E = BIN + P - 1, BINZ = BIN * BZ;
    constexpr int ZPL = 2, LPP = P <= 7 ? 4 : 8, ZW = ZPL * LPP;      // z offsets per lane, lanes per particle, z window (>= P + 1)
    constexpr int EZ = (BINZ + P - 1 > BINZ - 2 + ZW ? BINZ + P : BINZ - 2 + ZW) & ~1, HZ = EZ / 2, ROW = E * EZ, E3 = E * ROW, NW = 2 * P + ZW;
    // particles per pass: 40, not the 64 the lanes could take -- a bin holds ~30, and the 3.8 KB of weight tables this saves bring the
    // workgroup under 26 KB, so SIX share a CU instead of five (0.319 -> 0.297 ms; 48 still allocates for five; the 4 % of bins with
    // more than 40 particles take a second pass)
    constexpr int PPP = (NT / LPP) < 40 ? (NT / LPP) : 40;
    constexpr int NPC = E * E * HZ, ITER = (NPC + NT - 1) / NT;       // 16-byte pieces of the region, per thread
    static_assert(EZ >= BINZ + P - 1 && BINZ - 2 + ZW <= EZ, "z window inside the padded row");
    __shared__ __attribute__((aligned(16))) double reg[E3];
    __shared__ double s_w[NW * PPP];          // ax[P], ay[P], z-window weights[ZW] of the pass's particles: [t][particle]
    __shared__ int s_o[PPP];                  // row of the window's first node inside the region
    __shared__ unsigned s_id[PPP];            // sorted index | not-owned flag
    constexpr int KS = P <= 8 ? 8 : 16;   // row stride of the shear table K[tx][ty] (ty < P)
    __shared__ double s_k[SHEAR ? P * KS : 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // One (bin, component) per workgroup: a 19 KB region instead of 57 KB, so eight workgroups share a CU and their
    // load -> wait -> compute phases overlap (what a CU can have in flight is bounded by the LDS it can fill).
    int b = xcd_block(blockIdx.x, gridDim.x), bz, by;
    const int cmp = b % 3; b /= 3;
    b = fdiv(b, dz, bz);
    b = fdiv(b, dy, by);
    int bx = bx0 + b; if (bx >= fb.nbx) bx -= fb.nbx;
    const int bin = (bx * fb.nby + by) * fb.nbz + bz * BZ;            // dz counts groups of BZ bins along z (the last may be short)
    const int base = fb.off[bin], n = fb.off[bin + min(BZ, fb.nbz - bz * BZ)] - base;
    if (n == 0) return;
    const double *g = cmp == 0 ? gx : (cmp == 1 ? gy : gz);
    const int t0[3] = {bx * BIN, by * BIN, bz * BINZ};
    static_assert(P <= 16 && P * KS <= NT, "one table entry per thread");
    if (SHEAR && tid < P * KS) s_k[tid] = exp_lean(gc.lnk * (double)((tid / KS) * (tid % KS)));
    // Piece e of the region = 16 bytes (qx, qy, 2 hz .. 2 hz + 1), stored in that order.
    const bool windowed = G.nxl < G.Nx;
    // first plane of the region in the stored array (a slab rank stores planes x0 - hl .. x0 + nxl + nhalo - 1)
    const int px0 = windowed ? wrapi(t0[0] - (G.x0 - G.hl), G.Nx) : t0[0];
    if (px0 + E <= (windowed ? G.nxl + G.hl + G.nhalo : G.Nx) && t0[1] + E <= G.Ny && t0[2] + EZ <= G.Nz) {
        // straight into LDS (no registers): wave-uniform base + per-lane offset
        const char *src = reinterpret_cast<const char *>(g + ((size_t)px0 * G.Ny + t0[1]) * G.Nz + t0[2]);
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int e0 = it * NT + wave * 64, e = e0 + lane, hz = e % HZ, r = e / HZ, qy = r % E, qx = r / E;
            const unsigned rel = (unsigned)((((size_t)qx * G.Ny + qy) * G.Nz + 2 * hz) * sizeof(double));
            if (e < NPC) __builtin_amdgcn_global_load_lds((glb_void_t *)(src + rel), (lds_void_t *)(reg + 2 * e0), 16, 0, 0);
        }
    } else {
        // region crossing the periodic boundary (or a slab's window of stored planes): wrap every piece
        const int xs = G.x0 - G.hl, nstored = G.nxl + G.hl + G.nhalo;
        const size_t plane = (size_t)G.Ny * G.Nz;
        double2 v[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int e = tid + it * NT, hz = e % HZ, r = e / HZ, qy = r % E, qx = r / E;
            int ix = t0[0] + qx; if (ix >= G.Nx) ix -= G.Nx;
            int iy = t0[1] + qy; if (iy >= G.Ny) iy -= G.Ny;
            int iz = t0[2] + 2 * hz; if (iz >= G.Nz) iz -= G.Nz;       // Nz even: a piece never straddles the wrap
            bool ok = e < NPC;
            if (windowed) { ix = wrapi(ix - xs, G.Nx); ok = ok && ix < nstored; }   // stored plane index; other slabs' planes read as 0
            v[it] = make_double2(0.0, 0.0);
            if (ok) v[it] = *reinterpret_cast<const double2 *>(g + (size_t)ix * plane + (size_t)iy * G.Nz + iz);
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int e = tid + it * NT;
            if (e < NPC) *reinterpret_cast<double2 *>(reg + 2 * e) = v[it];
        }
    }
    const int grp = tid / LPP, tl = tid % LPP;
    const double cw = G.prefac * G.hx * G.hy * G.hz;                  // PSEv1/Brownian.cu:872
    for (int h0 = 0; h0 < n; h0 += PPP) {
        const int np = min(PPP, n - h0);
        if (h0) __syncthreads();                                      // the previous pass's tables are no longer read
        // the pass's particles, one lane each; the three axes on three waves: two exponentials and P - 1 products each
        if (wave < 3 && lane < np) {
            const double2 *rp = reinterpret_cast<const double2 *>(rec + base + h0 + lane);
            const double2 q0 = rp[0], q1 = rp[1], q2 = rp[2];
            const int4 hd = *reinterpret_cast<const int4 *>(&q0);
            const double c = G.expfac;
            const double Y0 = G.hy * q1.y, Z0 = G.hz * q2.x, u = G.hx * q1.x + gc.s * Y0;
            double a[P];
            if (wave == 0) {
                gauss_axis<P>(-c * u * u, -2.0 * c * G.hx * u, gc.rx, a);
#pragma unroll
                for (int t = 0; t < P; ++t) s_w[t * PPP + lane] = a[t];
            } else if (wave == 1) {
                gauss_axis<P>(-c * Y0 * Y0, -2.0 * c * G.hy * (Y0 + gc.s * u), gc.ry, a);
#pragma unroll
                for (int t = 0; t < P; ++t) s_w[(P + t) * PPP + lane] = a[t];
            } else {
                gauss_axis<P>(-c * Z0 * Z0, -2.0 * c * G.hz * Z0, gc.rz, a);
                const int oz = hd.z - t0[2], odd = ZPL == 2 ? (oz & 1) : 0;
#pragma unroll
                for (int jw = 0; jw < ZW; ++jw) {         // window weight jw belongs to support node jw - odd
                    double wj = 0.0;
#pragma unroll
                    for (int t = 0; t < P; ++t) wj = (jw - odd == t) ? a[t] : wj;
                    s_w[(2 * P + jw) * PPP + lane] = wj;
                }
                s_o[lane] = (hd.x - t0[0]) * ROW + (hd.y - t0[1]) * EZ + (oz - odd);
                s_id[lane] = (unsigned)hd.w;
            }
        }
        __syncthreads();                                              // tables written, region landed (the barrier waits for vmcnt(0))
        double u0 = 0.0;
        if (grp < np) {
            const double *r0 = reg + s_o[grp] + ZPL * tl;
            double ax[P], ay[P];
#pragma unroll
            for (int t = 0; t < P; ++t) { ax[t] = s_w[t * PPP + grp]; ay[t] = s_w[(P + t) * PPP + grp]; }
            double x0 = 0.0, x1 = 0.0;
#pragma unroll
            for (int tx = 0; tx < P; ++tx) {
                double y0 = 0.0, y1 = 0.0;                            // the x weight multiplies once per row of y offsets
#pragma unroll
                for (int ty = 0; ty < P; ++ty) {
                    double w = ay[ty];
                    if (SHEAR) w *= s_k[tx * KS + ty];
                    const double *r = r0 + tx * ROW + ty * EZ;
                    if (ZPL == 2) {
                        const double2 a = *reinterpret_cast<const double2 *>(r);
                        y0 = fma(w, a.x, y0); y1 = fma(w, a.y, y1);
                    } else {
                        y0 = fma(w, r[0], y0);
                    }
                }
                x0 = fma(ax[tx], y0, x0); x1 = fma(ax[tx], y1, x1);
            }
            u0 = s_w[(2 * P + ZPL * tl) * PPP + grp] * cw * x0;
            if (ZPL == 2) u0 = fma(s_w[(2 * P + ZPL * tl + 1) * PPP + grp] * cw, x1, u0);
        }
        u0 += dpp_quad<0xB1>(u0);   // quad_perm [1,0,3,2]
        u0 += dpp_quad<0x4E>(u0);   // quad_perm [2,3,0,1]
        if (LPP == 8) u0 += __shfl_xor(u0, 4, 64);
        if (grp < np && tl == 0) {
            const unsigned id = s_id[grp];
            if (!(id & 0x80000000u)) reinterpret_cast<double *>(u_s + id)[cmp] = u0;   // bit 31: owned by another slab rank
        }
    }
}

// generic support size: one exponential per node
__global__ void __launch_bounds__(256)
k_gather(const double4 *__restrict__ pos_s, int N, const double *__restrict__ gx, const double *__restrict__ gy,
         const double *__restrict__ gz, DGrid G, DBox box, double4 *__restrict__ u_s, CellRanges rows,
         const int *__restrict__ cell_off) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= N) return;
    if (!rows.row(p, cell_off)) {                   // no particle data on this rank for the row (another slab's, beyond the ghost layers)
        if (lane == 0) u_s[p] = make_double4(0.0, 0.0, 0.0, 0.0);
        return;
    }
    const double4 pp = pos_s[p];
    double fx, fy, fz;
    frac_coords(box, pp.x, pp.y, pp.z, fx, fy, fz);
    int sx, sy, sz;
    double d0x, d0y, d0z;
    support_start(fx, G.Nx, G.P, sx, d0x);
    support_start(fy, G.Ny, G.P, sy, d0y);
    support_start(fz, G.Nz, G.P, sz, d0z);
    // owned by the rank whose slab holds the particle's own plane; its support then lies inside the stored planes
    int own = (int)(fx * G.Nx) - G.x0; own %= G.Nx; if (own < 0) own += G.Nx;
    if (own >= G.nxl) {
        if (lane == 0) u_s[p] = make_double4(0.0, 0.0, 0.0, 0.0);
        return;
    }
    int rel0 = sx - (G.x0 - G.hl); rel0 %= G.Nx; if (rel0 < 0) rel0 += G.Nx;
    const int P = G.P, P2 = P * P, P3 = P2 * P;
    double ux = 0, uy = 0, uz = 0;
    for (int n = lane; n < P3; n += 64) {
        const int tx = n / P2, ty = (n - tx * P2) / P, tz = n - tx * P2 - ty * P;
        int lx = rel0 + tx; if (G.nxl == G.Nx && lx >= G.Nx) lx -= G.Nx;
        int iy = sy + ty; iy = iy < 0 ? iy + G.Ny : (iy >= G.Ny ? iy - G.Ny : iy);
        int iz = sz + tz; iz = iz < 0 ? iz + G.Nz : (iz >= G.Nz ? iz - G.Nz : iz);
        const double ey = G.hy * (d0y + ty);
        const double ex = G.hx * (d0x + tx) + box.xy * ey;
        const double ez = G.hz * (d0z + tz);
        const double w = exp_neg(-G.expfac * (ex * ex + ey * ey + ez * ez));
        const size_t idx = ((size_t)lx * G.Ny + iy) * G.Nz + iz;
        ux += w * gx[idx];
        uy += w * gy[idx];
        uz += w * gz[idx];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ux += __shfl_xor(ux, o, 64); uy += __shfl_xor(uy, o, 64); uz += __shfl_xor(uz, o, 64); }
    if (lane == 0) {
        const double c = G.prefac * G.hx * G.hy * G.hz;   // PSEv1/Brownian.cu:872
        u_s[p] = make_double4(c * ux, c * uy, c * uz, 0.0);
    }
}

template <int P>
static void launch_gather_p(const FarRec *rec, FarBins fb, int bx0, int nbx_l, const double *gx, const double *gy, const double *gz,
                            const DGrid &G, const GaussConsts &gc, double4 *u_s, hipStream_t s, int bz) {
    if constexpr (P <= 7) {
        if (bz == 2) {   // few particles per bin: two bins along z share a region (four: 51 KB, three workgroups per CU -- 0.79 ms
                         // against 0.63 at 360^3, 2.33 against 1.92 at 512^3: measured, not kept)
            const int ngz = (fb.nbz + 1) / 2;
            const dim3 g(3 * nbx_l * fb.nby * ngz), b(256);
            const FastDiv dz = fast_div(ngz), dy = fast_div(fb.nby);
            if (gc.s != 0.0) hipLaunchKernelGGL((k_gather_bins<P, true, 2>), g, b, 0, s, rec, fb, dz, dy, bx0, gx, gy, gz, G, gc, u_s);
            else hipLaunchKernelGGL((k_gather_bins<P, false, 2>), g, b, 0, s, rec, fb, dz, dy, bx0, gx, gy, gz, G, gc, u_s);
            return;
        }
    }
    const dim3 g(3 * nbx_l * fb.nby * fb.nbz), b(256);
    const FastDiv dz = fast_div(fb.nbz), dy = fast_div(fb.nby);
    if (gc.s != 0.0) hipLaunchKernelGGL((k_gather_bins<P, true>), g, b, 0, s, rec, fb, dz, dy, bx0, gx, gy, gz, G, gc, u_s);
    else hipLaunchKernelGGL((k_gather_bins<P, false>), g, b, 0, s, rec, fb, dz, dy, bx0, gx, gy, gz, G, gc, u_s);
}

hipError_t launch_gather(const double4 *pos_s, SpreadWork w, int N, const double *gx, const double *gy, const double *gz, DGrid G,
                         DBox box, double4 *u_s, hipStream_t s) {
    if (!farfield_fast_path(G) || !w.rec_t || (G.Nz & 1)) {
        hipLaunchKernelGGL(k_gather, dim3(nblocks(N, 4)), dim3(256), 0, s, pos_s, N, gx, gy, gz, G, box, u_s, w.need, w.cell_off);
        return hipGetLastError();
    }
    FarBins fb = w.fb;
    fb.nbx = bins_of(G.Nx); fb.nby = bins_of(G.Ny); fb.nbz = bins_of(G.Nz);
    int bx0 = 0, nbx_l = fb.nbx;
    if (G.nxl < G.Nx) {
        // a slab rank gathers the particles of its own planes (zeros elsewhere): their origins lie in [x0 - hl - 1, x0 + nxl)
        if (hipError_t e = hipMemsetAsync(u_s, 0, (size_t)N * sizeof(double4), s); e != hipSuccess) return e;
        const int olo = ((G.x0 - G.hl - 1) % G.Nx + G.Nx) % G.Nx, ohi = (G.x0 + G.nxl - 1) % G.Nx;
        bx0 = olo / BIN;
        nbx_l = std::min(fb.nbx, ((ohi / BIN - bx0) % fb.nbx + fb.nbx) % fb.nbx + 1);
    }
    const GaussConsts gc = gauss_consts(G, box.xy);
    // bins per workgroup along z: two where a bin holds fewer than ~20 particles on average (PSE_GATHER_BZ=1|2 overrides)
    const int bz_env = w.force_bz;
    const double per_bin = (double)N / ((double)(w.rows_local ? nbx_l : fb.nbx) * fb.nby * fb.nbz);
    const int bz = (G.Nz >= 32 && (bz_env ? bz_env == 2 : per_bin < 20.0)) ? 2 : 1;
    switch (G.P) {
#define PSE_GATHER_CASE(PV) case PV: launch_gather_p<PV>(w.rec_t, fb, bx0, nbx_l, gx, gy, gz, G, gc, u_s, s, bz); break;
        PSE_GATHER_CASE(4) PSE_GATHER_CASE(5) PSE_GATHER_CASE(6) PSE_GATHER_CASE(7) PSE_GATHER_CASE(8) PSE_GATHER_CASE(9)
        PSE_GATHER_CASE(10) PSE_GATHER_CASE(11) PSE_GATHER_CASE(12) PSE_GATHER_CASE(13)
        default: launch_gather_p<14>(w.rec_t, fb, bx0, nbx_l, gx, gy, gz, G, gc, u_s, s, bz); break;
#undef PSE_GATHER_CASE
    }
    return hipGetLastError();
}

}  // namespace pse
