// The z half of K4 / K7 (the reference's cuFFT C2C transforms, PSEv1/Brownian.cu:844-846,867-869) at the grid sizes of the
// reference's own rule that are not 256 or 512 (PSEv1/Stokes.cc:143-199: next 2^a 3^b 5^c -- 360 at the metric point, 270 / 360 at
// BASELINE config 3, 500 at config 4): k_zfft_rows (pse_kernels.hip) generalised from NC = R0 x 8 x 8 to NC = R0 x R1 x R2.
// A wavefront owns RW rows at a time; a row's N reals are NC = N / 2 complex points z[n] = x[2n] + i x[2n + 1]; lane l < L = R1 R2
// holds z[l + L r], r < R0 (coalesced 16-byte loads of the row as it lies); stage 1 is a radix-R0 butterfly in registers, stages 2
// (radix R1 on R0 R2 lanes per row) and 3 (radix R2 on R0 R1 lanes per row) trade their points through the wave's own LDS columns
// (no workgroup barrier anywhere); then the real <-> half-spectrum step on the natural order and coalesced non-temporal stores:
// the row crosses HBM once each way, rocFFT's 1-D real plans (a complex transform + an r2c / c2r step, two kernels per direction)
// are off the path.  Unnormalised both ways, like rocFFT's.  tools/debug/zfft_model.py is the index algebra in NumPy, checked
// against numpy.fft for every size instantiated here.
//   forward:  X[k] = (Z[k] + conj Z[NC - k]) / 2 - i/2 W_N^k (Z[k] - conj Z[NC - k]),  k = 0 .. NC      (Z[NC] = Z[0])
//   inverse:  Z[k] = (X[k] + conj X[NC - k]) + i conj W_N^k (X[k] - conj X[NC - k]),   k = 0 .. NC - 1;  x = N x the true inverse
#include "pse_kernels.h"
#include "pse_dft.h"

namespace pse {

struct ZRowsG { double *real[3]; double2 *spec[3]; int rows; int Nz, Nzp; };   // `rows` rows per component, consecutive in both arrays

template <int NC, int R0, int R1, int R2, bool INVERSE, int RPW>
__global__ void __launch_bounds__(256)
k_zfft_rows_g(ZRowsG zr, const double2 *__restrict__ tw /* exp(-2 pi i m / N), m < N = 2 NC */) {
    constexpr int L = R1 * R2, B2 = R0 * R2, B3 = R0 * R1, BM = B2 > B3 ? B2 : B3, RW = 64 / BM >= 4 ? 4 : (64 / BM >= 2 ? 2 : 1);   // rows per group: a power of two
    static_assert(NC == R0 * L && L <= 64 && BM <= 64 && RW >= 1 && RPW % RW == 0, "NC = R0 x R1 x R2, a row's stage on one wave");
    // a row's column in LDS: the stage layouts (k0 planes P0 apart, k1 groups P1 apart: odd, so that the lanes of stage 3 start on
    // different banks) and, once stage 3 has read its points, the natural order on top of them
    constexpr int P1 = R2 | 1, P0 = R1 * P1, CSW = P0 * (R0 - 1) + P1 * (R1 - 1) + R2, WB = CSW > NC ? CSW : NC, NQ = (NC + 63) / 64;
    __shared__ __attribute__((aligned(16))) double2 lds[4 * RW * WB];
    __shared__ __attribute__((aligned(16))) double2 tw2[R1 * R2];   // W_L^{lo k1} at [lo * R1 + k1]
    typedef double d2v __attribute__((ext_vector_type(2)));
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (threadIdx.x < R1 * R2) { double2 t = tw[2 * R0 * (threadIdx.x / R1) * (threadIdx.x % R1)]; if (INVERSE) t.y = -t.y; tw2[threadIdx.x] = t; }
    __syncthreads();
    const long total = 3L * zr.rows, row0 = ((long)blockIdx.x * 4 + wv) * RPW;
    if (row0 >= total) return;                                 // (whole waves; nothing below synchronises across waves)
    const int nrow = (int)min((long)RPW, total - row0);
    double2 *const col = lds + wv * RW * WB;                   // + j WB: row j of the group
    const bool lane1 = l < L;                                  // lanes of stage 1 (and of its loads / the inverse's stores)
    const int ll = lane1 ? l : 0;
    double2 t1[R0], tpl[R0], tpn[NQ];                          // W_NC^{l k0}; W_N^{l + L r} (inverse: the step in front of stage 1); W_N^{l + 64 q} (forward: behind stage 3)
#pragma unroll
    for (int q = 0; q < R0; ++q) {
        t1[q] = tw[2 * ll * q]; tpl[q] = tw[ll + L * q];
        if (INVERSE) { t1[q].y = -t1[q].y; tpl[q].y = -tpl[q].y; }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) tpn[q] = tw[min(l + 64 * q, NC - 1)];
    auto rowptrs = [&](long row, double *&xr, double2 *&xs) __attribute__((always_inline)) {
        const int c = (int)(row / zr.rows);
        const long r = row - (long)c * zr.rows;
        xr = zr.real[c] + r * zr.Nz; xs = zr.spec[c] + r * zr.Nzp;
    };
    // stage 2: lane -> (row of the group, k0, lo); stage 3: lane -> (row of the group, k0, k1)
    const bool lane2 = l < RW * B2, lane3 = l < RW * B3;
    const int j2 = lane2 ? l / B2 : 0, u2 = lane2 ? l % B2 : 0, k0_2 = u2 / R2, lo = u2 % R2;
    const int j3 = lane3 ? l / B3 : 0, u3 = lane3 ? l % B3 : 0, k0_3 = u3 / R1, k1_3 = u3 % R1;
    double2 *const col2 = col + j2 * WB + P0 * k0_2, *const col3 = col + j3 * WB;
    for (int i = 0; i < nrow; i += RW) {
        const int ng = min(RW, nrow - i);                       // rows of this group (the last may be short; wave-uniform)
        double2 a[RW][R0];
        if (!INVERSE) {
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                double *xr; double2 *xs;
                rowptrs(row0 + i + min(j, ng - 1), xr, xs);
#pragma unroll
                for (int q = 0; q < R0; ++q) { const d2v t = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(xr) + ll + L * q); a[j][q] = make_double2(t.x, t.y); }
            }
        } else {
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                double *xr; double2 *xs;
                rowptrs(row0 + i + min(j, ng - 1), xr, xs);
#pragma unroll
                for (int q = 0; q < R0; ++q) {
                    const int k = ll + L * q;                    // (k = 0 pairs with the Nyquist entry)
                    const d2v t = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(xs) + k), u = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(xs) + (NC - k));
                    const double2 sm = make_double2(t.x + u.x, t.y - u.y), df = make_double2(t.x - u.x, t.y + u.y);   // xk +- conj xc
                    const double2 w = cmul(tpl[q], df);          // conj W_N^k (xk - conj xc)
                    a[j][q] = make_double2(sm.x - w.y, sm.y + w.x);   // sm + i w
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            dft_small<R0, INVERSE>(a[j]);                        // stage 1 over r -> k0, times W_NC^{l k0}
#pragma unroll
            for (int q = 1; q < R0; ++q) a[j][q] = cmul(a[j][q], t1[q]);
            if (lane1) {
#pragma unroll
                for (int q = 0; q < R0; ++q) col[j * WB + P0 * q + l] = a[j][q];
            }
        }
        __builtin_amdgcn_wave_barrier();
        {
            double2 b[R1];
#pragma unroll
            for (int s = 0; s < R1; ++s) b[s] = col2[lo + R2 * s];
            __builtin_amdgcn_wave_barrier();
            dft_small<R1, INVERSE>(b);                           // stage 2 over s -> k1, times W_L^{lo k1}
#pragma unroll
            for (int k1 = 1; k1 < R1; ++k1) b[k1] = cmul(b[k1], tw2[lo * R1 + k1]);
            if (lane2) {
#pragma unroll
                for (int k1 = 0; k1 < R1; ++k1) col2[P1 * k1 + lo] = b[k1];
            }
        }
        __builtin_amdgcn_wave_barrier();
        {
            double2 b[R2];
#pragma unroll
            for (int n = 0; n < R2; ++n) b[n] = col3[P0 * k0_3 + P1 * k1_3 + n];   // lane (row, k0, k1)
            __builtin_amdgcn_wave_barrier();                     // (the natural order overwrites the stage layout)
            dft_small<R2, INVERSE>(b);                           // stage 3 over lo -> k2: Z[k0 + R0 k1 + R0 R1 k2]
            if (lane3) {
#pragma unroll
                for (int k2 = 0; k2 < R2; ++k2) col3[k0_3 + R0 * k1_3 + R0 * R1 * k2] = b[k2];
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            if (j >= ng) break;
            double *xr; double2 *xs;
            rowptrs(row0 + i + j, xr, xs);
            const double2 *nat = col + j * WB;
            if (!INVERSE) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int k = l + 64 * q;
                    if (k < NC) {
                        const double2 zk = nat[k], zc0 = nat[k ? NC - k : 0];
                        const double2 sm = make_double2(zk.x + zc0.x, zk.y - zc0.y), df = make_double2(zk.x - zc0.x, zk.y + zc0.y);   // zk +- conj zc
                        const double2 t = cmul(tpn[q], df);      // W_N^k (zk - conj zc)
                        d2v o; o.x = 0.5 * (sm.x + t.y); o.y = 0.5 * (sm.y - t.x);   // (sm - i t) / 2
                        __builtin_nontemporal_store(o, reinterpret_cast<d2v *>(xs) + k);
                    }
                }
                if (l == 0) { const double2 z0 = nat[0]; xs[NC] = make_double2(z0.x - z0.y, 0.0); }
            } else {
                d2v *z = reinterpret_cast<d2v *>(xr);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int k = l + 64 * q;
                    if (k < NC) { const double2 t = nat[k]; d2v o; o.x = t.x; o.y = t.y; __builtin_nontemporal_store(o, z + k); }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                         // (the next group's stage 1 overwrites the columns)
    }
}

constexpr int ZFFT_G_RPW = 8;
template <int NC, int R0, int R1, int R2>
static void launch_g(const ZRowsG &zr, bool inverse, const double2 *tw, hipStream_t s) {
    const dim3 g((unsigned)((3L * zr.rows + 4 * ZFFT_G_RPW - 1) / (4 * ZFFT_G_RPW))), b(256);
    if (inverse) hipLaunchKernelGGL((k_zfft_rows_g<NC, R0, R1, R2, true, ZFFT_G_RPW>), g, b, 0, s, zr, tw);
    else hipLaunchKernelGGL((k_zfft_rows_g<NC, R0, R1, R2, false, ZFFT_G_RPW>), g, b, 0, s, zr, tw);
}
// Nz -> (R0, R1, R2): the even sizes of the reference's rule between 180 and 500 (an odd Nz has no half-length complex form: rocFFT
// keeps those, as it keeps every size not listed); tools/debug/zfft_model.py SIZES is the same table
#define PSE_ZFFT_SIZES(X) X(360, 3, 6, 10) X(270, 3, 5, 9) X(180, 2, 5, 9) X(240, 2, 6, 10) X(300, 3, 5, 10) X(320, 4, 4, 10) X(384, 3, 8, 8) \
                          X(400, 4, 5, 10) X(450, 5, 5, 9) X(480, 4, 6, 10) X(500, 5, 5, 10)
bool zfft_g_supported(int Nz) {
#define PSE_Z_CASE(N, A, B, C) if (Nz == N) return true;
    PSE_ZFFT_SIZES(PSE_Z_CASE)
#undef PSE_Z_CASE
    return false;
}
void launch_zfft_g(double *const real[3], double2 *const spec[3], int rows, int Nz, int Nzp, bool inverse, const double2 *tw, hipStream_t s) {
    ZRowsG zr{};
    for (int c = 0; c < 3; ++c) { zr.real[c] = real[c]; zr.spec[c] = spec[c]; }
    zr.rows = rows; zr.Nz = Nz; zr.Nzp = Nzp;
#define PSE_Z_CASE(N, A, B, C) if (Nz == N) { launch_g<N / 2, A, B, C>(zr, inverse, tw, s); return; }
    PSE_ZFFT_SIZES(PSE_Z_CASE)
#undef PSE_Z_CASE
}

}  // namespace pse
