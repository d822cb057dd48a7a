// Device-side helpers of the PSE engine (gfx950): triclinic box algebra, Philox4x32-10, real-space
// pair functions.  These restate the HOOMD device helpers the reference relies on but does not ship
// (BoxDim::makeFraction/minImage/wrap, detail::Saru, texFetchScalar4 -- SURVEY.md 8(a15)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pse_host.h"

namespace pse {

struct DBox {
    double Lx, Ly, Lz, xy;
    double iLx, iLy, iLz;
};

struct DGrid {
    int Nx, Ny, Nz, Nzh;   // Nzh = Nz/2 + 1 (half spectrum)
    int Nzp;               // elements a spectrum row is stored with: Nzh rounded up to 8 (128-byte aligned rows)
    int P;
    int x0, nxl;           // this rank's slab of x planes [x0, x0 + nxl)
    int nhalo;             // planes stored past the slab (copies of the next slab's first planes) for the gather
    int hl;                // planes stored before the slab (copies of the previous slab's last planes); storage starts at x0 - hl
    double hx, hy, hz;
    double prefac, expfac; // (2 xi^2/(pi eta))^{3/2}, 2 xi^2/eta   (PSEv1/Brownian.cu:828-829)
};

// Storage order of the cells: x slowest, then blocks of bz cells along z, then y, then z inside the block --
// slot = ((cx nzb + cz / bz) ny + cy) bz + cz % bz.  With bz = nz this is the plain (x, y, z) order (PSE_CELL_BZ=0).  Blocks
// (default: six cells) make 64 consecutive particles (a wavefront's rows) a squat 1 x 2 x 6-cell brick instead of a 12-cell needle
// along z, so the neighbour records a wave gathers come from ~95 cells instead of ~130; x stays slowest, so a rank's cell
// slab is still one contiguous row range.  The last block of a line is padded with empty cells when bz does not divide nz.
struct DCells {
    int nx, ny, nz;        // cells per dimension (1 or >= 3)
    int bz, nzb;           // block height along z and blocks per z line (nzb = ceil(nz / bz))
    int xpad;              // 1: every x layer ends with one EMPTY storage cell (owned-particle teams: the rows of a layer range then end at
                           //    cell_off[its last cell + 1] even where the next layer's rows live in another region of the row space); else 0
};
__host__ __device__ inline int layer_cells(const DCells &nc) { return nc.nzb * nc.ny * nc.bz + nc.xpad; }   // storage cells of one x layer
__host__ __device__ inline int cells_total(const DCells &nc) { return nc.nx * layer_cells(nc); }
__host__ __device__ inline int cell_slot(const DCells &nc, int cx, int cy, int zb, int zi) { return ((cx * nc.nzb + zb) * nc.ny + cy) * nc.bz + zi + cx * nc.xpad; }

// fractional coordinates in [0,1): f = ((x - xy*y)/Lx + 1/2, y/Ly + 1/2, z/Lz + 1/2)
__device__ __forceinline__ void frac_coords(const DBox &b, double x, double y, double z, double &fx, double &fy, double &fz) {
    fx = (x - b.xy * y) * b.iLx + 0.5;
    fy = y * b.iLy + 0.5;
    fz = z * b.iLz + 0.5;
    fx -= floor(fx); fy -= floor(fy); fz -= floor(fz);
    // guard against f == 1.0 after rounding
    if (fx >= 1.0) fx = 0.0;
    if (fy >= 1.0) fy = 0.0;
    if (fz >= 1.0) fz = 0.0;
}

// minimum image of a displacement (valid for |r| below half the smallest perpendicular box width)
__device__ __forceinline__ void min_image(const DBox &b, double &dx, double &dy, double &dz) {
    const double ny = rint(dy * b.iLy);
    dy -= ny * b.Ly;
    dx -= ny * b.xy * b.Ly;
    dx -= rint(dx * b.iLx) * b.Lx;
    dz -= rint(dz * b.iLz) * b.Lz;
}

__device__ __forceinline__ int cell_coord(double f, int n) {
    int c = (int)(f * n);
    c = c < 0 ? 0 : c;            // (a NaN coordinate converts to INT_MIN: garbage in must not become an address out of bounds)
    return c >= n ? n - 1 : c;
}

// Workgroups b and b + 8 share an XCD (round-robin dispatch, /opt/skills/guides/MI355X_MICROARCH.md): hand each XCD a
// contiguous range of logical blocks, so the neighbour data a block gathers is what the other blocks of its XCD
// gather too and stays in that XCD's 4 MiB L2.  Bijective for any block count; placement affects speed only.
__device__ __forceinline__ int xcd_block(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, x = b & 7, k = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// ---- a Lanczos vector row in 16 bytes: three 40-bit mantissas under one exponent ------------------------------------------------
// The pair-list mat-vec gathers its neighbours' vector rows, and a wavefront's gather of 64 scattered addresses costs the texture
// path the same 64 cycles whether it fetches 16 or 8 bytes per lane: the 24 bytes of (x, y, z) in doubles are TWO such instructions
// per pair, and the kernel is bound by them (TA 78 % busy).  One 16-byte word per row holds the three components as signed 40-bit
// integers scaled by the power of two above the largest of them: absolute error <= 2^-39 of the row's largest component (1.8e-12),
// four orders below the rounding of the pair coefficients the same mat-vec reads (pse_kernels.hip PairCoef).  w = (x lo, y lo, z lo, x hi | y hi << 8 |
// z hi << 16 | (e + 128) << 24).  Written by k_lz_update beside the double row it mirrors; the row itself stays the truth (diagonal
// term, sums, basis combination).
typedef unsigned vq4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ vq4 vq_pack(double x, double y, double z) {
    const double m = fmax(fabs(x), fmax(fabs(y), fabs(z)));
    int e = 0;
    (void)frexp(m, &e);                                   // m = f 2^e, 0.5 <= f < 1  (m = 0: e = 0)
    e = e < -100 ? -100 : (e > 127 ? 127 : e);
    const double sc = __hiloint2double((39 - e + 1023) << 20, 0);   // 2^(39 - e): |component| sc < 2^39
    const double lim = 549755813887.0;                    // 2^39 - 1 (f -> 1 may round up to 2^39)
    const long long ix = (long long)fmax(-lim, fmin(lim, rint(x * sc))), iy = (long long)fmax(-lim, fmin(lim, rint(y * sc))),
                    iz = (long long)fmax(-lim, fmin(lim, rint(z * sc)));
    vq4 w;
    w.x = (unsigned)ix; w.y = (unsigned)iy; w.z = (unsigned)iz;
    w.w = ((unsigned)(ix >> 32) & 0xffu) | (((unsigned)(iy >> 32) & 0xffu) << 8) | (((unsigned)(iz >> 32) & 0xffu) << 16) | ((unsigned)(e + 128) << 24);
    return w;
}
__device__ __forceinline__ void vq_unpack(vq4 w, double &x, double &y, double &z) {
    const int e = (int)(w.w >> 24) - 128;
    const double sc = __hiloint2double((e - 39 + 1023) << 20, 0);   // 2^(e - 39)
    const int hx = (int)(w.w << 24) >> 24, hy = (int)(w.w << 16) >> 24, hz = (int)(w.w << 8) >> 24;   // sign-extended bits 39..32
    x = fma((double)hx, 4294967296.0, (double)w.x) * sc;
    y = fma((double)hy, 4294967296.0, (double)w.y) * sc;
    z = fma((double)hz, 4294967296.0, (double)w.z) * sc;
}

__device__ __forceinline__ int wrapi(int a, int n) { a %= n; return a < 0 ? a + n : a; }

// ---- Philox4x32-10 ------------------------------------------------------------------------------------
constexpr uint32_t PHILOX_KEY1 = 0x50534531u;  // 'PSE1'
constexpr uint32_t DOMAIN_PARTICLE = 0, DOMAIN_GRID_A = 1, DOMAIN_GRID_B = 2;

__host__ __device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                    uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// uint32 -> uniform on (-s, s)
__host__ __device__ __forceinline__ double uniform_pm(uint32_t x, double s) {
    return (((double)x + 0.5) * 2.3283064365386963e-10 * 2.0 - 1.0) * s;
}

// exp(x) for |x| < 700 -- the exponentials on the path are Gaussian weights exp(-c r^2), their step ratios and the
// k-space factor.  ~18 VALU instructions instead of the library's ~40: no overflow/NaN paths; underflow goes to 0
// through ldexp.  |error| < 2 ulp: Cody-Waite reduction x = n ln2 + r, |r| <= ln2/2, Taylor to r^13 (remainder 4e-18).
__device__ __forceinline__ double exp_lean(double x) {
    const double n = rint(x * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = fma(p, r, 2.08767569878681e-09);
    p = fma(p, r, 2.505210838544172e-08);
    p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 1.984126984126984e-04);
    p = fma(p, r, 1.388888888888889e-03);
    p = fma(p, r, 8.333333333333333e-03);
    p = fma(p, r, 4.1666666666666664e-02);
    p = fma(p, r, 1.6666666666666666e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}
__device__ __forceinline__ double exp_neg(double x) { return exp_lean(x); }   // call sites whose argument is <= 0

// sin(x) for 0 <= x < ~1e5 (wave numbers |k| a of the far-field grid): two-term reduction by pi, odd Taylor series to x^21 on
// [-pi/2, pi/2] (truncation 2.5e-16 relative to x) -- a fifth of the instructions of the library routine, which pays for
// arguments of any size.
__device__ __forceinline__ double sin_lean(double x) {
    const double n = rint(x * 0.31830988618379067154);
    double r = fma(-n, 3.14159265358979311600, x);
    r = fma(-n, 1.2246467991473531772e-16, r);
    const double r2 = r * r;
    double p = -1.9572941063391261231e-20;                       // -1/21!
    p = fma(p, r2, 8.2206352466243297170e-18);                   //  1/19!
    p = fma(p, r2, -2.8114572543455207632e-15);                  // -1/17!
    p = fma(p, r2, 7.6471637318198164759e-13);                   //  1/15!
    p = fma(p, r2, -1.6059043836821614599e-10);                  // -1/13!
    p = fma(p, r2, 2.5052108385441718775e-08);                   //  1/11!
    p = fma(p, r2, -2.7557319223985890653e-06);                  // -1/9!
    p = fma(p, r2, 1.9841269841269841270e-04);                   //  1/7!
    p = fma(p, r2, -8.3333333333333333333e-03);                  // -1/5!
    p = fma(p, r2, 1.6666666666666666667e-01);                   //  1/3!
    const double sr = fma(-r * r2, p, r);
    return ((long long)n & 1) ? -sr : sr;
}

// ---- real-space pair functions ------------------------------------------------------------------------
// f(r), g(r) of M_real = f (I - rr) + g rr  (replaces the fp32 linear table PSEv1/Stokes.cc:334-422 and its
// lookup PSEv1/Mobility.cu:661-670): analytic free-space RPY minus the tabulated smooth wave part.
// Returns f and (g - f)/r^2.
template <int STRIDE = 2 * RS_NCOEF>   // doubles per interval (a copy in LDS pads it to 21: intervals then start on different banks)
__device__ __forceinline__ void eval_fg(double r2, const double *__restrict__ coef, double &f, double &gmf_r2) {
    const double ir = rsqrt(r2), r = r2 * ir, ir2 = ir * ir;   // one reciprocal square root instead of a square root and a division
    double f0, g0;
    if (r > 2.0) {
        const double ir3 = ir * ir2;
        f0 = 0.75 * ir + 0.5 * ir3;
        g0 = 1.5 * ir - ir3;
    } else {
        f0 = 1.0 - 0.28125 * r;
        g0 = 1.0 - 0.1875 * r;
    }
    const double s = r * RS_PER_UNIT;
    const int k = (int)s;
    const double t = 2.0 * (s - k) - 1.0;
    const double *c = coef + (size_t)k * STRIDE;
    double fw = c[RS_DEG], gw = c[RS_NCOEF + RS_DEG];
#pragma unroll
    for (int q = RS_DEG - 1; q >= 0; --q) {
        fw = fma(fw, t, c[q]);
        gw = fma(gw, t, c[RS_NCOEF + q]);
    }
    f = f0 - fw;
    gmf_r2 = ((g0 - gw) - f) * ir2;
}

// ---- cell walk ---------------------------------------------------------------------------------------
__device__ __forceinline__ void image_shift(unsigned code, const DBox &b, double &sx, double &sy, double &sz) {
    const int wx = (int)(code / 9) - 1, wy = (int)((code / 3) % 3) - 1, wz = (int)(code % 3) - 1;
    sx = wx * b.Lx + wy * b.xy * b.Ly;
    sy = wy * b.Ly;
    sz = wz * b.Lz;
}

struct CellWalk {   // the neighbour cells of one particle as (slot range, image code) runs
    int cx, cy, cz, rx, ry, rz;
};

template <class F>
__device__ __forceinline__ void for_each_run(const DCells &nc, const int *__restrict__ cell_off, int cx, int cy, int cz,
                                             F &&body) {
    const int rx = nc.nx > 1 ? 1 : 0, ry = nc.ny > 1 ? 1 : 0;
    const int zb0 = cz / nc.bz, zi0 = cz - zb0 * nc.bz;
    for (int ox = -rx; ox <= rx; ++ox) {
        int ax = cx + ox, wx = 0;
        if (ax < 0) { ax += nc.nx; wx = -1; } else if (ax >= nc.nx) { ax -= nc.nx; wx = 1; }
        for (int oy = -ry; oy <= ry; ++oy) {
            int ay = cy + oy, wy = 0;
            if (ay < 0) { ay += nc.ny; wy = -1; } else if (ay >= nc.ny) { ay -= nc.ny; wy = 1; }
            const unsigned cxy = (unsigned)((wx + 1) * 9 + (wy + 1) * 3);
            if (nc.nz == 1) {
                const int c = cell_slot(nc, ax, ay, 0, 0);
                body(cell_off[c], cell_off[c + 1], cxy + 1u);
                continue;
            }
            // the three z cells, as maximal runs of consecutive slots with one image code (same block, no wrap in between)
            int s[3]; unsigned code[3];
#pragma unroll
            for (int oz = -1; oz <= 1; ++oz) {
                int az = cz + oz, wz = 0;
                if (az < 0) { az += nc.nz; wz = -1; } else if (az >= nc.nz) { az -= nc.nz; wz = 1; }
                int zb = zb0, zi = zi0 + oz;                           // block and offset of az, from those of cz
                if (wz != 0 || zi < 0 || zi >= nc.bz) { zb = az / nc.bz; zi = az - zb * nc.bz; }
                s[oz + 1] = cell_slot(nc, ax, ay, zb, zi);
                code[oz + 1] = cxy + (unsigned)(wz + 1);
            }
            int k = 0;
            while (k < 3) {
                int e = k;
                while (e + 1 < 3 && s[e + 1] == s[e] + 1 && code[e + 1] == code[k]) ++e;
                body(cell_off[s[k]], cell_off[s[e] + 1], code[k]);
                k = e + 1;
            }
        }
    }
}


// slot and image code of the z neighbour cz + oz of cell (ax, ay, cz); zb0, zi0: block and offset of cz
__device__ __forceinline__ void z_neighbour_slot(const DCells &nc, int ax, int ay, int cz, int zb0, int zi0, int oz, unsigned cxy,
                                                 int &slot, unsigned &code) {
    int az = cz + oz, wz = 0;
    if (az < 0) { az += nc.nz; wz = -1; } else if (az >= nc.nz) { az -= nc.nz; wz = 1; }
    int zb = zb0, zi = zi0 + oz;
    if (wz != 0 || zi < 0 || zi >= nc.bz) { zb = az / nc.bz; zi = az - zb * nc.bz; }
    slot = cell_slot(nc, ax, ay, zb, zi);
    code = cxy + (unsigned)(wz + 1);
}

}  // namespace pse
